#!/bin/bash
# A/B of two builds of the library on the same box, alternating: bash tools/ab_build.sh <a> <b> [probe arguments]
# runs tools/tail_window_probe.py under haphic_amd/libhaphic_hip_<a>.so and ..._<b>.so (HAPHIC_HIP_SO), twice each; the probe prints the
# digest of the whole result, so "the same bits" is checked between the builds and not only between the knob values of one build.
mkdir -p gpurun_out
A=${1:-head}; B=${2:-new}; shift 2
ARGS=${@:-1.1 row_order 1,1}
for v in $A $B $A $B; do
  HAPHIC_HIP_SO=$PWD/haphic_amd/libhaphic_hip_$v.so timeout 300 python tools/tail_window_probe.py $ARGS > gpurun_out/ab_$v.json.tmp 2> gpurun_out/ab_$v.err; echo "$v rc=$?"
  cat gpurun_out/ab_$v.json.tmp >> gpurun_out/ab_$v.jsonl; cut -c1-700 gpurun_out/ab_$v.json.tmp
done
rm -f gpurun_out/*.tmp
