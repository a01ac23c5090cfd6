#!/bin/bash
# GPU pass 2: kernel trace of the C2 bench, then the C3 (100k contigs / 500M pairs) probe
mkdir -p gpurun_out
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_c2 -o c2 -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/prof_c2.log 2>&1; echo "rc=$?" >> gpurun_out/prof_c2.log
ls -R gpurun_out/prof_c2 | head -30
timeout 900 python bench.py --contigs 100000 --pairs 500000000 --nchrs 24 --mean-len 30000 --steps 1 --warmup 0 --no-cpu-baseline > gpurun_out/bench_c3.log 2>&1; echo "rc=$?" >> gpurun_out/bench_c3.log
tail -c 6000 gpurun_out/bench_c3.log
