#!/bin/bash
# HBM reads and L2 hits of the generic-stream window kernel over the C3 tail at inflation 1.1, rows as listed (row_order 0) and in min-hash order (1):
# rocprofv3 --pmc passes of tools/tail_window_probe.py, summed over the k_expand_window launches (FETCH_SIZE in KB, x 2 on gfx950: tools/pmc_summary.py)
mkdir -p gpurun_out; export TMPDIR=/tmp
INFL=${1:-1.1}
for order in 0 1; do
  for spec in "rd:FETCH_SIZE" "l2:TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum"; do
    name="${spec%%:*}"; ctrs="${spec#*:}"; rm -rf gpurun_out/tpmc_${order}_$name
    timeout 900 rocprofv3 --kernel-trace --pmc $ctrs --output-format csv -d gpurun_out/tpmc_${order}_$name -o p -- python tools/tail_window_probe.py $INFL row_order $order > gpurun_out/tpmc_${order}_$name.json 2> gpurun_out/tpmc_${order}_$name.err; echo "tail_pmc order $order $name rc=$?"
  done
done
# the SQ counters of the same launches in min-hash order (what a wave spends its cycles on): three more passes
for spec in "insts:SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM" "sq:SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" "lds:SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA"; do
    name="${spec%%:*}"; ctrs="${spec#*:}"; rm -rf gpurun_out/tpmc_1_sq$name
    timeout 900 rocprofv3 --kernel-trace --pmc $ctrs --output-format csv -d gpurun_out/tpmc_1_sq$name -o p -- python tools/tail_window_probe.py $INFL row_order 1 > gpurun_out/tpmc_1_sq$name.json 2> gpurun_out/tpmc_1_sq$name.err; echo "tail_pmc sq $name rc=$?"
done
python - <<'PYEOF'
import collections, csv, glob, json, os
out = {}
for order in (0, 1):
    agg = collections.defaultdict(float)
    n = 0
    for f in glob.glob('gpurun_out/tpmc_%d_*/p_counter_collection.csv' % order):
        seen = set()
        for r in csv.DictReader(open(f)):
            k = r['Kernel_Name'].replace('(anonymous namespace)::', '').split('(')[0].replace('void ', '')
            if not k.startswith('k_expand_window<0, '):
                continue
            agg[r['Counter_Name']] += float(r['Counter_Value'])
            seen.add(r['Dispatch_Id'])
        n = max(n, len(seen))
    probe = json.load(open('gpurun_out/tpmc_%d_rd.json' % order))
    out['row_order_%d' % order] = {'window_launches': n, 'hbm_read_bytes': agg['FETCH_SIZE'] * 1024.0 * 2.0, 'tcc_miss_x_128B': agg['TCC_MISS_sum'] * 128.0,
                                   'l2_hit_rate': agg['TCC_HIT_sum'] / agg['TCC_REQ_sum'] if agg['TCC_REQ_sum'] else None,
                                   'probe_under_pmc': {k: v for k, v in probe.items() if k.startswith('row_order')}}
sq = collections.defaultdict(float)
for f in glob.glob('gpurun_out/tpmc_1_sq*/p_counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'].replace('(anonymous namespace)::', '').split('(')[0].replace('void ', '')
        if k.startswith('k_expand_window<0, '):
            sq[r['Counter_Name']] += float(r['Counter_Value'])
out['row_order_1']['sq_counters_summed_over_the_window_launches'] = dict(sq)
json.dump(out, open('gpurun_out/tail_pmc.json', 'w'), indent=1)
print(json.dumps(out))
PYEOF
