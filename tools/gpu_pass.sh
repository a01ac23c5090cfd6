#!/bin/bash
# usage (through gpurun): bash tools/gpu_pass.sh [tests] [c2] [c3] [prof_c2] [prof_c3]
mkdir -p gpurun_out
ulimit -c 0                        # a GPU memory fault must not spend minutes writing a core file
export TMPDIR=/tmp
cd ${GRAFT_REPO_ROOT:-.}
C3="--contigs 100000 --pairs 500000000 --nchrs 24 --mean-len 30000"
for what in "$@"; do
  case $what in
    seam) timeout 900 python -m pytest tests/test_seam_containers.py tests/test_gpu_pipeline.py tests/test_bam.py -m gpu -x -q --durations=5 > gpurun_out/pytest_seam.log 2>&1; echo "seam rc=$?"; grep -E "passed|failed|Error|^E " gpurun_out/pytest_seam.log | tail -12; grep -A6 "slowest" gpurun_out/pytest_seam.log | head -7;;
    seamleg) python -c "
import json
for l in open('gpurun_out/bench_full.log'):
    if l.startswith('{'): print(json.dumps(json.loads(l).get('seam_e2e'))[:3000])
";;
    lowtails) timeout 1500 python tools/lowtails.py $LOWTAIL_ARGS > gpurun_out/lowtails.jsonl 2> gpurun_out/lowtails.err; echo "lowtails rc=$?"; python -c "
import json
for l in open('gpurun_out/lowtails.jsonl'):
    r = json.loads(l); r.pop('sampled_iterations', None); print(json.dumps(r)[:700])
"; tail -3 gpurun_out/lowtails.err;;
    rest) timeout 900 python -m pytest tests/test_gpu_scale.py tests/test_plot.py tests/test_seam_containers.py -m gpu -x -q -k "not c3 and not c2" --durations=5 > gpurun_out/pytest_rest.log 2>&1; echo "rest rc=$?"; grep -E "passed|failed|Error|^E " gpurun_out/pytest_rest.log | tail -12;;
    c1run) timeout 600 python tools/c1_run.py > gpurun_out/c1_run.json 2> gpurun_out/c1_run.err; echo "c1run rc=$?"; cat gpurun_out/c1_run.json | cut -c1-1500; tail -3 gpurun_out/c1_run.err;;
    host8c3) timeout 1100 python bench.py --gpus 8 --transport host --steps 1 --warmup 0 --no-cpu-baseline --no-seam --sweep 20 --sharded-sweep-timeout 900 --check-sweep > gpurun_out/bench_host8_c3.log 2>&1; echo "host8c3 rc=$?"; python tools/bench_brief.py gpurun_out/bench_host8_c3.log | head -3; python -c "
import json
for l in open('gpurun_out/bench_host8_c3.log'):
    if l.startswith('{'): print(json.dumps(json.loads(l).get('sweep_sharded'))[:1200])
"; tail -3 gpurun_out/bench_host8_c3.log | cut -c1-300;;
    grouppmc) # SQ counters of the generic-stream window kernel and of the re-use kernel on the same tail (24k contigs, inflation 1.1)
        for spec in "insts:SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM" "sq:SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" "lds:SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA"; do
            name="${spec%%:*}"; ctrs="${spec#*:}"; rm -rf gpurun_out/gpmc_$name
            timeout 600 rocprofv3 --kernel-trace --pmc $ctrs --output-format csv -d gpurun_out/gpmc_$name -o p -- python tools/lowtails.py --configs ${GROUP_CFG:-k24} --inflations 1.1 --reuse-ab > gpurun_out/gpmc_$name.jsonl 2> gpurun_out/gpmc_$name.err; echo "grouppmc $name rc=$?"
        done
        python - <<'PYEOF'
import collections, csv, glob, json
agg = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.defaultdict(int)
for f in glob.glob('gpurun_out/gpmc_*/p_counter_collection.csv'):
    seen = collections.defaultdict(set)
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'].replace('(anonymous namespace)::', '').split('(')[0].replace('void ', '')
        if not (k.startswith('k_expand_group') or k.startswith('k_expand_window<0, ') or k.startswith('k_expand_hash')):
            continue
        agg[k][r['Counter_Name']] += float(r['Counter_Value'])
        seen[k].add(r['Dispatch_Id'])
    for k, s_ in seen.items():
        cnt[k] = max(cnt[k], len(s_))
out = {k: dict(v, launches=cnt[k]) for k, v in agg.items()}
json.dump(out, open('gpurun_out/group_pmc.json', 'w'), indent=1)
for k, v in out.items():
    print(k, json.dumps(v))
PYEOF
        ;;
    tests) timeout 1500 python -m pytest tests -m gpu -x -q --durations=6 > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log; grep -A8 "slowest" gpurun_out/pytest_gpu.log | head -9; tail -2 gpurun_out/pytest_gpu.log;;
    c2) timeout 600 python bench.py --contigs 10000 --pairs 50000000 --nchrs 16 --mean-len 50000 --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/bench_c2.log 2>&1; echo "rc=$?" >> gpurun_out/bench_c2.log; python tools/bench_brief.py gpurun_out/bench_c2.log;;
    c3) timeout 900 python bench.py $C3 --steps 1 --warmup 0 --no-cpu-baseline > gpurun_out/bench_c3.log 2>&1; echo "rc=$?" >> gpurun_out/bench_c3.log; grep "hhx expand" gpurun_out/bench_c3.log | head -3; python tools/bench_brief.py gpurun_out/bench_c3.log;;
    c3w) HHX_DEBUG=1 timeout 900 python bench.py $C3 --steps 1 --warmup 1 --no-cpu-baseline --text-lines 0 > gpurun_out/bench_c3w.log 2>&1; echo "rc=$?" >> gpurun_out/bench_c3w.log; grep "hhx expand" gpurun_out/bench_c3w.log | cut -c150-420 | tail -26 | head -4; python tools/bench_brief.py gpurun_out/bench_c3w.log | head -2;;
    sharded1) timeout 900 python bench.py --steps 1 --warmup 1 --no-cpu-baseline --force-sharded > gpurun_out/bench_sh.log 2>&1; echo "rc=$?" >> gpurun_out/bench_sh.log; python tools/bench_brief.py gpurun_out/bench_sh.log; tail -3 gpurun_out/bench_sh.log | cut -c1-300;;
    pushes4) timeout 900 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --pushes 4 > gpurun_out/bench_p4.log 2>&1; echo "rc=$?" >> gpurun_out/bench_p4.log; python tools/bench_brief.py gpurun_out/bench_p4.log | head -4;;
    text) timeout 600 python -m pytest tests -m gpu -x -q -k "pairs_text" 2>&1 | tail -3; timeout 600 python bench.py --contigs 10000 --pairs 20000000 --nchrs 16 --mean-len 50000 --steps 1 --warmup 0 --no-cpu-baseline > gpurun_out/bench_text.log 2>&1; echo "rc=$?"; python -c "
import json
for l in open('gpurun_out/bench_text.log'):
    if l.startswith('{'): print(json.dumps(json.loads(l)['ingest'].get('text')))
";;
    pcie) timeout 600 python tools/pcie_rate.py 100000000 2>&1 | tail -4;;
    full) timeout 900 python bench.py > gpurun_out/bench_full.log 2>&1; echo "rc=$?" >> gpurun_out/bench_full.log; python tools/bench_brief.py gpurun_out/bench_full.log; tail -c 600 gpurun_out/bench_full.log | grep -v "^{";;
    prof) rm -rf gpurun_out/prof; timeout 1200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof -o r -- python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-parity --text-lines 0 $PMC_ARGS > gpurun_out/prof.log 2>&1; echo "rc=$?" >> gpurun_out/prof.log; ls -R gpurun_out/prof | head; tail -2 gpurun_out/prof.log | cut -c1-400;;
    pmc:*) # pmc:NAME:COUNTER1,COUNTER2  -> one rocprofv3 --pmc pass of the default bench (1 step, no warmup)
        spec="${what#pmc:}"; name="${spec%%:*}"; ctrs="${spec#*:}"; rm -rf gpurun_out/pmc_$name
        timeout 900 rocprofv3 --kernel-trace --pmc ${ctrs//,/ } --output-format csv -d gpurun_out/pmc_$name -o p -- python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-parity --text-lines 0 $PMC_ARGS > gpurun_out/pmc_$name.log 2>&1; echo "pmc $name rc=$?"; ls gpurun_out/pmc_$name | head -5;;
    probe) timeout 1200 python tools/expand_probe.py $PROBE_ARGS > gpurun_out/expand_probe.log 2>&1; echo "probe rc=$?"; cut -c1-330 gpurun_out/expand_probe.log | tail -20;;
    ktests) timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -x -q > gpurun_out/pytest_k.log 2>&1; echo "ktests rc=$?"; tail -5 gpurun_out/pytest_k.log;;
    calib) # known-byte microkernels under the PMC counters (tools/pmc_calib.hip), one pass per counter group
        ./tools/pmc_calib > gpurun_out/calib_plain.log 2>&1; cat gpurun_out/calib_plain.log
        for spec in "a:FETCH_SIZE TCC_MISS_sum" "b:TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum" "c:WRITE_SIZE TCC_EA0_WRREQ_sum" "d:TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_64B_sum TCC_READ_sum TCC_WRITE_sum"; do
            name="${spec%%:*}"; ctrs="${spec#*:}"; rm -rf gpurun_out/calib_$name
            timeout 300 rocprofv3 --kernel-trace --pmc $ctrs --output-format csv -d gpurun_out/calib_$name -o p -- ./tools/pmc_calib > gpurun_out/calib_$name.log 2>&1; echo "calib $name rc=$?"
        done
        python tools/pmc_calib_summary.py gpurun_out | tee gpurun_out/calib_summary.txt;;
    sbench) timeout 600 ./tools/stream_bench > gpurun_out/stream_bench.log 2>&1; echo "sbench rc=$?"; cat gpurun_out/stream_bench.log;;
    ldsbench) timeout 600 ./tools/lds_atomic_bench > gpurun_out/lds_atomic_bench.log 2>&1; echo "ldsbench rc=$?"; cat gpurun_out/lds_atomic_bench.log;;
    scale) timeout 1500 python -m pytest tests/test_gpu_scale.py -m gpu -x -q --durations=5 > gpurun_out/pytest_scale.log 2>&1; echo "scale rc=$?"; tail -12 gpurun_out/pytest_scale.log;;
    flips) timeout 1200 python tools/flip_count.py > gpurun_out/flip_count_c2.json 2> gpurun_out/flip_count.err; echo "flips rc=$?"; cut -c1-1500 gpurun_out/flip_count_c2.json; tail -3 gpurun_out/flip_count.err;;
    multirank) timeout 1500 python -m pytest tests/test_gpu_multirank.py -m gpu -x -q --durations=3 > gpurun_out/pytest_multirank.log 2>&1; echo "multirank rc=$?"; tail -15 gpurun_out/pytest_multirank.log;;
    plot) timeout 600 python -m pytest tests/test_plot.py -m gpu -x -q > gpurun_out/pytest_plot.log 2>&1; echo "plot rc=$?"; grep -E "passed|failed|Error|error" gpurun_out/pytest_plot.log | tail -8;;
    cprobe) timeout 600 python tools/contact_probe.py > gpurun_out/contact_probe.jsonl 2> gpurun_out/contact_probe.err; echo "cprobe rc=$?"; cat gpurun_out/contact_probe.jsonl; tail -3 gpurun_out/contact_probe.err;;
    iprobe) timeout 600 python tools/ingest_probe.py > gpurun_out/ingest_probe.jsonl 2> gpurun_out/ingest_probe.err; echo "iprobe rc=$?"; cat gpurun_out/ingest_probe.jsonl; tail -3 gpurun_out/ingest_probe.err;;
    k:*) timeout 900 python -m pytest tests -m gpu -x -q -k "${what#k:}" > gpurun_out/pytest_sel.log 2>&1; echo "sel rc=$?"; grep -E "passed|failed|Error|^E " gpurun_out/pytest_sel.log | tail -12;;
    rccl) timeout 600 python tools/rccl_probe.py > gpurun_out/rccl_probe.jsonl 2> gpurun_out/rccl_probe.err; echo "rccl rc=$?"; cat gpurun_out/rccl_probe.jsonl; tail -3 gpurun_out/rccl_probe.err;;
    host2) timeout 900 python bench.py --gpus 2 --transport host --contigs 10000 --pairs 50000000 --nchrs 16 --mean-len 50000 --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/bench_host2.log 2>&1; echo "host2 rc=$?"; python tools/bench_brief.py gpurun_out/bench_host2.log; tail -3 gpurun_out/bench_host2.log | cut -c1-400;;
    host8) timeout 1200 python bench.py --gpus 8 --transport host --contigs 10000 --pairs 50000000 --nchrs 16 --mean-len 50000 --steps 1 --warmup 0 --no-cpu-baseline --sweep 8 > gpurun_out/bench_host8.log 2>&1; echo "host8 rc=$?"; python tools/bench_brief.py gpurun_out/bench_host8.log | head -3; python -c "
import json
for l in open('gpurun_out/bench_host8.log'):
    if l.startswith('{'): print(json.dumps(json.loads(l).get('sweep_sharded'))[:900])
";;
    c5) timeout 1500 python bench.py --contigs 200000 --pairs 2000000000 --pushes 4 --steps 1 --warmup 1 --no-cpu-baseline --text-lines 0 --sweep 0 > gpurun_out/bench_c5.log 2>&1; echo "c5 rc=$?"; python tools/bench_brief.py gpurun_out/bench_c5.log; tail -2 gpurun_out/bench_c5.log | cut -c1-300;;
    asan) # the host side of the library under AddressSanitizer (haphic_amd/build.py build_asan), a subset of the gpu tests
        RT=$(python -c "from haphic_amd import build; print(build.asan_runtime())")
        LD_PRELOAD="$RT /usr/lib/x86_64-linux-gnu/libstdc++.so.6" ASAN_OPTIONS=detect_leaks=0:protect_shadow_gap=0:verify_asan_link_order=0:halt_on_error=1 LD_LIBRARY_PATH=$(python -c "import torch, os; print(os.path.join(os.path.dirname(torch.__file__), 'lib'))"):$LD_LIBRARY_PATH HAPHIC_HIP_SO=${GRAFT_REPO_ROOT:-$PWD}/haphic_amd/libhaphic_hip_asan.so \
          timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_pipeline.py tests/test_seam_containers.py tests/test_bam.py tests/test_plot.py -m gpu -q -p no:cacheprovider > gpurun_out/pytest_asan.log 2>&1
        echo "asan rc=$?"; grep -E "passed|failed|ERROR: AddressSanitizer|SUMMARY" gpurun_out/pytest_asan.log | tail -5; tail -3 gpurun_out/pytest_asan.log | cut -c1-300;;
    tails) rm -rf gpurun_out/prof_tails; timeout 1200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_tails -o t -- python tools/tail_probe.py $TAIL_ARGS > gpurun_out/tail_probe.jsonl 2> gpurun_out/tail_probe.err; echo "tails rc=$?"; cut -c1-400 gpurun_out/tail_probe.jsonl; tail -3 gpurun_out/tail_probe.err; find gpurun_out/prof_tails -name "*kernel_stats.csv" | head -1 | xargs -r head -12 | cut -c1-200;;
    tailpmc) # fabric bytes of the tail kernels at inflation 1.2 (generic-stream window class, hash class): two PMC passes of the first iterations
        for spec in "rd:FETCH_SIZE" "wr:WRITE_SIZE"; do
            name="${spec%%:*}"; ctr="${spec#*:}"; rm -rf gpurun_out/tailpmc_$name
            timeout 600 rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d gpurun_out/tailpmc_$name -o p -- python tools/tail_probe.py --inflations 1.2 --max-iter 12 > gpurun_out/tailpmc_$name.jsonl 2> gpurun_out/tailpmc_$name.err; echo "tailpmc $name rc=$?"
        done
        python - <<'PY'
import collections, csv, json
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for name in ('rd', 'wr'):
    try:
        for r in csv.DictReader(open('gpurun_out/tailpmc_%s/p_counter_collection.csv' % name)):
            k = r['Kernel_Name'].replace('(anonymous namespace)::', '').split('(')[0].replace('void ', '')
            agg[k][r['Counter_Name']].append(float(r['Counter_Value']))
    except OSError as e:
        print(e)
out = {}
for k, cs in agg.items():
    if 'k_expand' in k or 'k_dense' in k:
        rd = sum(cs.get('FETCH_SIZE', [])) * 1024 * 2
        wr = sum(cs.get('WRITE_SIZE', [])) * 1024
        out[k] = {'launches': max(len(v) for v in cs.values()), 'read_bytes': rd, 'write_bytes': wr}
probe = [json.loads(l) for l in open('gpurun_out/tailpmc_rd.jsonl') if l.startswith('{')]
res = {'what': 'rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes; FETCH_SIZE x 1024 x 2 on gfx950) of tools/tail_probe.py --inflations 1.2 --max-iter 12',
       'kernels': out, 'probe': probe}
json.dump(res, open('gpurun_out/tail_pmc_summary.json', 'w'), indent=1)
for k, v in sorted(out.items(), key=lambda kv: -kv[1]['read_bytes'])[:8]:
    print(k, v)
PY
        ;;
    oscale) timeout 600 python tools/oracle_scaling.py > gpurun_out/oracle_scaling.jsonl 2> gpurun_out/oracle_scaling.err; echo "oscale rc=$?"; cat gpurun_out/oracle_scaling.jsonl; tail -2 gpurun_out/oracle_scaling.err;;
    listpmc) rocprofv3 -L > gpurun_out/pmc_list.txt 2>&1; grep -c . gpurun_out/pmc_list.txt;;
    env:*) export "${what#env:}"; echo "set ${what#env:}";;
    *) echo "unknown $what";;
  esac
done
