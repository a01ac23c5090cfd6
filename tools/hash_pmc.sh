#!/bin/bash
# SQ counters of k_expand_hash (and the window kernel beside it) over one bench step: three rocprofv3 --pmc passes, summed per kernel
mkdir -p gpurun_out; export TMPDIR=/tmp
ARGS="--steps 1 --warmup 0 --no-cpu-baseline --no-parity --text-lines 0 --no-seam --no-run-e2e --sweep 0"
for spec in "insts:SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM" "sq:SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" "lds:SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA"; do
    name="${spec%%:*}"; ctrs="${spec#*:}"; rm -rf gpurun_out/hpmc_$name
    timeout 600 rocprofv3 --kernel-trace --pmc $ctrs --output-format csv -d gpurun_out/hpmc_$name -o p -- python bench.py $ARGS > gpurun_out/hpmc_$name.log 2>&1; echo "hash_pmc $name rc=$?"
done
python - <<'PYEOF'
import collections, csv, glob, json
agg = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.defaultdict(int)
for f in glob.glob('gpurun_out/hpmc_*/p_counter_collection.csv'):
    seen = collections.defaultdict(set)
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'].replace('(anonymous namespace)::', '').split('(')[0].replace('void ', '')
        if not (k.startswith('k_expand_hash') or k.startswith('k_expand_window<0, ')):
            continue
        agg[k][r['Counter_Name']] += float(r['Counter_Value'])
        seen[k].add(r['Dispatch_Id'])
    for k, s_ in seen.items():
        cnt[k] = max(cnt[k], len(s_))
out = {k: dict(v, launches=cnt[k]) for k, v in agg.items()}
json.dump(out, open('gpurun_out/hash_pmc.json', 'w'), indent=1)
for k, v in out.items():
    print(k, json.dumps(v))
PYEOF
