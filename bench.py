#!/usr/bin/env python3
"""bench.py — the hot path of BASELINE.json on synthetic data, inputs resident in HBM.

One "step" = one pass of the hot path over one batch: ingest P Hi-C read pairs into the link tables
(hash-table kernel + insertion-order finalisation), dict_to_matrix, L1 normalise, pre-expansion (M^2)
and mcl() at inflation 2.0 (:2026-2062), attractor read-out.

metric/value  : Hi-C pairs/s through the link-matrix build (ingest + finalise + dict_to_matrix), the
                first half of BASELINE.json's metric; the second half (MCL iterations/s) is reported
                in the "mcl" object of the same JSON line.
workload (N=1): the configuration BASELINE.json's metric is quoted on ("@100k contigs"), configs[2]:
                100k contigs / 500M synthetic pairs, inflation 2.0 — it fits one MI355X (8 GB of pairs,
                ~45 GB of tables).  configs[1] (10k / 50M): --contigs 10000 --pairs 50000000 --nchrs 16
                --mean-len 50000.  N>1: STRONG scaling of the same job ("1->8 MI355X row-block shard"):
                rank r ingests the r-th contiguous chunk of the 500M-pair stream, the link tables are
                merged with one exchange, and the MCL row blocks are all-gathered every iteration
                (haphic_amd/sharded.py).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

HBM_PEAK_GBS = 8000.0       # MI355X_MICROARCH.md: HBM3E 8 TB/s peak (6.3 TB/s achievable)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=2)
    ap.add_argument('--warmup', type=int, default=1)
    ap.add_argument('--contigs', type=int, default=100000)
    ap.add_argument('--pairs', type=int, default=500_000_000, help='total pairs of the job (split over ranks)')
    ap.add_argument('--nchrs', type=int, default=24)
    ap.add_argument('--mean-len', type=int, default=30_000)
    ap.add_argument('--inflation', type=float, default=2.0)
    ap.add_argument('--sharded-sweep-timeout', type=int, default=240, help='N > 1: seconds the sharded inflation sweep (after the timed region) may take before the line is printed without it')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-reference-python', action='store_true', help='cpu_baseline: do not time the reference checkout even if it is present (carry the stored measurement)')
    ap.add_argument('--pushes', type=int, default=1, help='hand the pairs over in this many batches (streaming ingest: one aggregated run per batch, merged at finalize)')
    ap.add_argument('--force-sharded', action='store_true', help='take the multi-GPU code path on one rank (1-rank nccl group)')
    ap.add_argument('--cpu-sample-pairs', type=int, default=20_000_000)
    ap.add_argument('--text-lines', type=int, default=1_000_000, help='a1 leg: lines of .pairs text formatted on the host (0 = skip)')
    ap.add_argument('--text-file-tile', type=int, default=110, help='a1 leg: copies of that text written to the .pairs file of the end-to-end figure (110 x 96 MB = a 10.6 GB file; reduced automatically when the temporary directory is short of space)')
    ap.add_argument('--no-parity', action='store_true', help='skip the oracle check of sampled rows of iteration 0')
    ap.add_argument('--text-tile', type=int, default=16, help='a1 leg: copies of that text concatenated in HBM')
    ap.add_argument('--sweep', type=int, default=20, help='after the timed region: the inflation sweep of run_mcl_clustering :2155-2158 (1.1, 1.2, ... this many '
                    'values) with ONE expansion — iteration 0 of every inflation from the dense row blocks of M^2 (0 = skip)')
    ap.add_argument('--sweep-tail-seconds', type=float, default=20.0, help='sweep leg: wall-time budget for the mcl() tails (run from the highest inflation down)')
    ap.add_argument('--check-sweep', action='store_true', help='N > 1: rank 0 also runs the one-GPU sweep on the all-gathered link matrix and compares every inflation (functional runs)')
    ap.add_argument('--no-seam', action='store_true', help='skip the seam_e2e leg (the reference\'s own operator sequence S5 -> filter_fragments -> S4 on host id arrays, after the timed region)')
    ap.add_argument('--no-run-e2e', action='store_true', help='skip the run_e2e leg (the whole seam sequence of run() :2829-2945 from the .pairs FILE, 20-inflation sweep and every file included; tools/c3_run.py)')
    ap.add_argument('--no-seam-files', action='store_true', help='seam_e2e: skip the three files run() writes between S5 and S4 (HT_links.pkl, paired_links.clm, full_links.pkl)')
    ap.add_argument('--transport', choices=('rccl', 'host'), default='rccl',
                    help='rccl: one rank per GPU over RCCL / xGMI (the product path).  host: the same ranks and the same exchanges, every collective '
                         'staged through host memory over gloo (haphic_amd.host_transport) — runs N ranks on ONE GPU, a functional proof of '
                         'the multi-rank path on a box without xGMI; its timings say nothing about scaling')
    ap.add_argument('--master-port', type=int, default=0, help='rendezvous port of the self-launch (0: pick a free one)')
    return ap.parse_args()


def self_launch(args):
    """`python bench.py --gpus N` with no rendezvous in the environment: become `python -m torch.distributed.run --nnodes=1
    --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py <same arguments>` (the command the driver itself uses).
    Under torchrun (RANK / WORLD_SIZE set) this is a no-op."""
    if args.gpus <= 1 or 'WORLD_SIZE' in os.environ:
        return
    port = args.master_port
    if not port:
        import socket
        with socket.socket() as sk:
            sk.bind(('127.0.0.1', 0))
            port = sk.getsockname()[1]
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(args.gpus), '--master-addr', '127.0.0.1',
           '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
    sys.stdout.flush()
    os.execv(sys.executable, cmd)


def main():
    args = parse()
    self_launch(args)
    import torch
    import torch.distributed as dist
    from haphic_amd import _lib, synth
    from haphic_amd.cluster import FragTable

    rank = int(os.environ.get('RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if args.gpus != world:
        raise SystemExit('--gpus %d but WORLD_SIZE is %d' % (args.gpus, world))
    n_dev = torch.cuda.device_count()
    if args.transport == 'rccl' and world > n_dev:
        raise SystemExit('%d ranks but %d GPU(s): RCCL needs one device per rank (use --transport host for a functional run on one GPU)' % (world, n_dev))
    gpu = local_rank % max(1, n_dev)
    torch.cuda.set_device(gpu)
    _lib.check(_lib.load().hhx_set_device(gpu))
    dev = 'cuda:%d' % gpu
    sharded_path = world > 1 or args.force_sharded
    if sharded_path:
        if world == 1:
            os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
            os.environ.setdefault('MASTER_PORT', '29533')
        if args.transport == 'host':
            from haphic_amd import host_transport as _ht
            dist.init_process_group('gloo', rank=rank, world_size=world)
            raw_dist, dist = dist, _ht.HostStagedCollectives(dist)
            dist.destroy_process_group = raw_dist.destroy_process_group
        elif world == 1:
            dist.init_process_group('nccl', rank=0, world_size=1, device_id=torch.device(dev))
        else:
            dist.init_process_group('nccl', device_id=torch.device(dev))

    per_chr = max(1, args.contigs // args.nchrs)
    gen = synth.make_genome(args.nchrs, per_chr * args.mean_len, args.mean_len, seed=12345)
    n = gen.n
    lex = gen.lexical_rank()
    table = FragTable.for_contigs(lex, gen.length, np.ones(n, np.uint8))
    flank = 500_000                                   # --flank 500 (kb), HapHiC_cluster.py default
    # this rank's shard of the read-pair stream, generated straight into HBM
    local_pairs = args.pairs // world
    # generated in slices of 500 M (the sampler's temporaries are ~80 B per pair) and torch's cache handed back, so
    # that the library's own allocator sees the whole HBM
    parts = [synth.sample_pairs(gen, min(500_000_000, local_pairs - lo), seed=12345 + rank + 1000 * k, device=dev)
             for k, lo in enumerate(range(0, local_pairs, 500_000_000))]
    id1, p1, id2, p2 = [torch.cat([q[c] for q in parts]) if len(parts) > 1 else parts[0][c] for c in range(4)]
    del parts
    torch.cuda.synchronize()
    torch.cuda.empty_cache()
    in_set = np.ones(n, np.uint8)

    if sharded_path:
        from haphic_amd import sharded
    state = {}

    def step():
        t0 = time.perf_counter()
        ing = _lib.Ingest(table, flank, bins=False, skip_intra=True)
        ing.set_ordinal_base(rank * local_pairs)
        bounds = [local_pairs * k // args.pushes for k in range(args.pushes + 1)]
        for lo, hi in zip(bounds[:-1], bounds[1:]):
            ing.push_device(hi - lo, id1[lo:hi].data_ptr(), p1[lo:hi].data_ptr(), id2[lo:hi].data_ptr(), p2[lo:hi].data_ptr())
        n_full, n_flank = ing.finalize()
        if sharded_path:
            # all-reduce(min) of the first positions + all-to-all(v) of the matrix entries by row owner: every rank ends
            # up with ITS row block of the link matrix (no rank ever holds another rank's table)
            state['shard_ms'] = {}
            m, _fi, n_linked, shape = sharded.build_link_matrix_sharded(sharded.HipEngine(dev), ing, in_set, dist, timings=state['shard_ms'])
        else:
            # dict_to_matrix fused onto the device-resident table; link-less contigs get trailing indices
            m, fidx, n_linked = ing.link_matrix(in_set)
        _lib.check(_lib.load().hhx_synchronize())
        t1 = time.perf_counter()
        state['n_full'], state['n_flank'], state['nnz_link'] = n_full, n_flank, m.nnz
        ing.destroy()
        # ---- run_mcl_clustering :2144-2158 at one inflation
        t2 = time.perf_counter()
        if sharded_path:
            # the raw row blocks are all-gathered once; normalisation (:2144) is row-local, iteration 0 = class stream
            res, n_iter, conv, stats = sharded.mcl_sharded(None, 2, args.inflation, 200, 1e-4, dist, dev, local_links=m, n=shape)
        else:
            # normalisation (:2144) and pre-expansion (:2146-2147) fused into iteration 0: the 10^8..10^10-entry
            # M^2 never exists, and iteration 0 streams the link matrix as 16-bit counts
            res, n_iter, conv, stats = _lib.mcl(m, 2, args.inflation, 200, 1e-4, want_stats=True, links=True)
        att, ptr, mem = _lib.interpret(res)
        state['t_mcl'] = time.perf_counter() - t2
        state['clusters'] = len(att)
        t_pre = 0.0
        t3 = time.perf_counter()
        state.update(t_ingest=t1 - t0, t_pre=t_pre, n_iter=n_iter, conv=conv, stats=stats, t_total=t3 - t0)
        res.free()
        if state.get('keep_matrix'):
            state['matrix'] = m                      # the parity leg (after the timed region) re-uses the last link matrix
        elif state.get('keep_block'):
            state['block'], state['shape'] = m, shape        # the N-rank sweep leg re-uses this rank's row block of the link matrix
        else:
            m.free()

    def barrier():
        torch.cuda.synchronize()
        if sharded_path:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    _lib.profile_reset()
    _lib.profile_enable(True)
    if sharded_path:
        sharded.STAGES = {}            # per-stage clocks of the row-block MCL (raw all-gather, iteration 0, every exchange, the replicated tail): a few stream syncs per iteration
    barrier()
    t0 = time.perf_counter()
    t_ing = t_mcl = t_pre = 0.0
    iters = 0
    for k_step in range(args.steps):
        state['keep_matrix'] = (k_step == args.steps - 1) and world == 1 and not sharded_path and not (args.no_parity and args.sweep <= 1)
        state['keep_block'] = (k_step == args.steps - 1) and sharded_path and world > 1 and args.sweep > 1
        step()
        t_ing += state['t_ingest']; t_mcl += state['t_mcl']; t_pre += state['t_pre']; iters += state['n_iter']
    barrier()
    elapsed = time.perf_counter() - t0
    _lib.profile_enable(False)
    mcl_stages = None
    if sharded_path:
        mcl_stages, sharded.STAGES = sharded.STAGES, None
    tm = torch.tensor([elapsed, t_ing, t_mcl, t_pre], dtype=torch.float64, device=dev)
    if sharded_path:
        dist.all_reduce(tm, op=dist.ReduceOp.MAX)
    elapsed, t_ing, t_mcl, t_pre = tm.tolist()

    final_out = None
    if rank == 0:
        K = args.steps
        pairs_total = local_pairs * world * K
        value = pairs_total / t_ing
        stats = np.asarray(state['stats'])
        pg = _lib.profile_get
        # ---- dominant kernel of the step: k_expand_window (the B-row stream of the fused expansion), iteration 0.
        # Algorithmic bytes = what the kernel's stream format moves per launch: 2 B per product of a value-uniform
        # sub-segment (16-bit window-local column; the count-1 entries of the link matrix), 6 B per other product
        # (column + float32 value; SURVEY §8d counts 8 B for an int32 + float32 entry: `achieved_survey_8B_model`)
        # + 24 B per staged A entry and window (column, value, 16-byte segment record); survivors are written by the
        # finalize kernel.  Time: HIP events on the launch stream around the n_win launches of every call.
        win_ms, win_n = pg('expand_window')
        F_w = _lib.profile_counter('expand_window_products')
        F_u = _lib.profile_counter('expand_window_uniform_products')
        A_w = _lib.profile_counter('expand_window_a_reads')
        traffic = pmc_traffic(n, local_pairs)
        roofline = None
        if win_n:
            X_b = _lib.profile_counter('expand_window_explicit_bytes')     # 4 B per explicit product in the integer arithmetic (count << 16 | column), 6 B otherwise
            alg = 2.0 * F_u + float(X_b) + 24.0 * A_w
            ach = alg / (win_ms * 1e-3) / 1e9
            roofline = {'kernel': 'k_expand_window', 'bound': 'hbm (fabric behind L2: Infinity Cache + HBM, not separable by the TCC counters)', 'achieved': ach, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                        'frac': ach / HBM_PEAK_GBS, 'traffic': pick(traffic, 'k_expand_window<0'),
                        'alg_bytes_per_launch': alg / win_n, 'avg_launch_ms': win_ms / win_n, 'launches_per_step': win_n / K,
                        'products_per_step': F_w / K, 'uniform_products_per_step': F_u / K, 'bytes_per_product': alg / F_w if F_w else None,
                        'products_per_s': F_w / (win_ms * 1e-3),
                        'achieved_survey_8B_model': (8.0 * F_w + 16.0 * A_w) / (win_ms * 1e-3) / 1e9,
                        'measured_stream_ceiling_GBs': 6290.0, 'frac_of_measured_ceiling': ach / 6290.0,
                        'lds_atomic_ceiling_products_per_s': 256 * 2.4e9 * 64 / 12.77,
                        'full_products_per_step': float(stats[0, 3]) if len(stats) else None,
                        'note': 'iteration 0 (the link matrix is the operand; long segments) in the integer arithmetic of the link matrix: S = L D^-1 L is '
                                'exactly symmetric, so the launches walk only the blocks J >= I (products_per_step of full_products_per_step) into a dense '
                                'float32 block; the rest is transposed (dense_transpose) and the rows are finished by dense_epilogue.  The launches of later iterations are in '
                                'mcl.kernel_ms_per_step.expand_window_short.  The stream is served by the fabric behind L2 (HBM + the 256 MB '
                                'Infinity Cache: one column-window slice of the class stream is ~200 MB), so the HBM peak is the contract '
                                'denominator and the float4-copy ceiling of MI355X_MICROARCH.md (6.29 TB/s) the practical one; the second '
                                'ceiling is the LDS atomic rate (ds_add_u64 on random slots: 12.8 clk per wave instruction, '
                                'profiles/r02_lds_atomic_bench.jsonl).  traffic = fabric bytes per launch from rocprofv3 PMC (profiles/), '
                                'null if no profile matches this workload'}
        # whole link-matrix build against SURVEY §8d's B_ingest = 16 P + 12 (K_full + K_flank) + 4 n
        b_ingest = 16.0 * local_pairs + 12.0 * (state['n_full'] + state['n_flank']) + 4.0 * n
        # no single kernel bounds `value` (map 5.0, three radix levels 10.8, aggregation 7.9, matrix 14.4 of ~40 ms): the roofline
        # of the link-matrix build is the WHOLE build against B_ingest; traffic = the sum over its kernels
        build_kernels = ('k_map_records', 'k_part_count', 'k_part_scatter', 'k_aggregate', 'k_run_stats', 'k_row_emit', 'k_min_over_xcc',
                         'k_index_from_sorted', 'k_len_from_base', 'k_iota_u64', 'k_init_counts')
        build_traffic = sum(v for name, v in traffic.items() if any(bk in name for bk in build_kernels)) or None
        ach_build = b_ingest * K / t_ing / 1e9
        ing_roofline = {'kernel': 'link-matrix build, all kernels (k_map_records, k_part_count/scatter x5, k_aggregate, k_row_emit, ...)', 'bound': 'hbm',
                        'achieved': ach_build, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': ach_build / HBM_PEAK_GBS,
                        'traffic': build_traffic, 'alg_bytes_per_launch': b_ingest, 'avg_launch_ms': t_ing / K * 1e3,
                        'note': 'achieved = SURVEY 8d B_ingest (16 B per pair + 12 B per key + 4 B per fragment) / wall time of the build; the group-by '
                                'moves several times B_ingest by construction (map record + 3 radix levels + aggregated run + 2 matrix levels)'}
        cmp_ms, _ = pg('expand_compact')
        fin_ms, _ = pg('expand_finalize')
        tiny_ms, _ = pg('expand_tiny')
        cvg_ms, _ = pg('convergence')
        # B_iter of SURVEY §8d summed over the iterations of one mcl() call (fused: C is never written or re-read)
        b_iter = float((8 * (stats[:, 0] + stats[:, 3]) + 8 * stats[:, 2] + 16 * stats[:, 2] + 12 * n).sum()) if len(stats) else 0.0
        mcl = {'iters_per_s': iters / t_mcl if t_mcl else None, 'n': int(n), 'iterations': int(state['n_iter']),
               'converged': bool(state['conv']), 'inflation': args.inflation, 'ms_per_mcl': t_mcl / K * 1e3,
               'pre_expansion': 'fused into iteration 0', 'clusters': state.get('clusters'),
               'alg_bytes_per_mcl_survey': b_iter, 'alg_GBs_survey': b_iter * K / t_mcl / 1e9 if t_mcl else None,
               'stats_nnzA_nnzC_nnzP_F': stats.tolist(),
               'kernel_ms_per_step': {'expand_window': win_ms / K, 'expand_window_short': pg('expand_window_short')[0] / K, 'expand_hash': pg('expand_hash')[0] / K,
                                      'dense_transpose': pg('dense_transpose')[0] / K, 'dense_epilogue': pg('dense_epilogue')[0] / K,
                                      'expand_finalize': fin_ms / K, 'expand_compact': cmp_ms / K,
                                      'expand_tiny': tiny_ms / K,
                                      'convergence': cvg_ms / K}}
        ingest = {'pairs_per_s': value, 'ms_per_step': t_ing / K * 1e3, 'alg_bytes_survey': b_ingest,
                  'alg_GBs_survey': b_ingest * K / t_ing / 1e9,
                  'kernels_ms_per_step': {k: pg(k)[0] / K for k in ('ingest', 'map', 'part_count1', 'part_scatter1', 'part_count2',
                                                                   'part_scatter2', 'part_count3', 'part_scatter3', 'aggregate',
                                                                   'ingest_merge', 'link_matrix', 'd2m_count1', 'd2m_scatter1', 'd2m_count2',
                                                                   'd2m_scatter2', 'd2m_rank', 'd2m_emit')},
                  'roofline': ing_roofline}
        out = {'metric': 'Hi-C pairs/s ingested (link-matrix build) + MCL iters/s', 'value': value, 'unit': 'pairs/s',
               'n_gpus': world, 'steps': K, 'warmup': args.warmup, 'ms_per_step': elapsed / K * 1e3,
               'transport': ('RCCL' if args.transport == 'rccl' else 'host-staged gloo, %d rank(s) on %d GPU(s): functional run, not a scaling figure' % (world, n_dev)) if sharded_path else None,
               'higher_is_better': True, 'scaling': 'strong', 'vs_baseline': None,
               'dtype': 'u64 keys + u32 counts (ingest); f32 values, exact u64 fixed-point accumulation (MCL)',
               'data': 'synthetic',
               'config': {'workload': '%d contigs / %d pairs (whole job, %d per GPU), %d chr, mean contig %d bp, inflation %.1f, no MFMA dense-block expansion (iteration 0 accumulates M^2 into a dense float32 block in HBM)'
                                      % (n, local_pairs * world, local_pairs, args.nchrs, args.mean_len, args.inflation),
                          'contigs': int(n), 'pairs_per_gpu': local_pairs, 'full_keys': int(state['n_full']),
                          'flank_keys': int(state['n_flank']), 'link_matrix_nnz': int(state['nnz_link'])},
               'mcl_iters_per_s': mcl['iters_per_s'], 'ingest': ingest, 'mcl': mcl, 'roofline': roofline}
        if world == 1 and args.text_lines:
            out['ingest']['text'] = text_leg(args, gen, id1, p1, id2, p2, dev)
        if sharded_path:
            out['ingest']['sharded_build_ms_last_step'] = state.get('shard_ms')
            # rank 0's clocks per step, to lay next to DESIGN.md 5.2: [ms, bytes received by this rank, calls]
            out['mcl']['sharded_stage_ms_bytes_calls_per_step'] = {k_: [v[0] / K, v[1] / K, v[2] / K] for k_, v in (mcl_stages or {}).items()}
        if state.get('matrix') is not None and args.sweep > 1:
            try:
                out['sweep'] = sweep_leg(args, state['matrix'], t_mcl / K)
            except RuntimeError as e:
                out['sweep'] = {'error': str(e)[:300]}
        if state.get('matrix') is not None and not args.no_parity:
            out['parity'] = parity_leg(args, state.pop('matrix'))
        elif state.get('matrix') is not None:
            state.pop('matrix').free()
        if world == 1 and not sharded_path and not args.no_seam:
            try:
                out['seam_e2e'] = seam_leg(args, gen, id1, p1, id2, p2)
            except Exception as e:                       # noqa: BLE001 — a leg after the timed region must never take the line down
                out['seam_e2e'] = {'error': '%s: %s' % (type(e).__name__, str(e)[:300])}
        if not args.no_cpu_baseline and world == 1:          # a reported baseline of the N = 1 line only
            out['cpu_baseline'] = cpu_baseline(args, gen, table, flank, id1, p1, id2, p2, state)
        if world == 1 and not sharded_path and not args.no_run_e2e:
            # ONE wall clock over run()'s own sequence from the .pairs file (a1 + alignments.bed -> S5 -> HT_links.pkl -> paired_links.clm ->
            # filter_fragments -> full_links.pkl -> dict_to_matrix -> run_mcl_clustering with its 20 inflation directories), after everything else
            try:
                sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), 'tools'))
                import c3_run
                torch.cuda.empty_cache()
                _lib.check(_lib.load().hhx_pool_trim())
                out['run_e2e'] = c3_run.run_job(pairs=local_pairs, nchrs=args.nchrs, gen=gen, arrays=[id1, p1, id2, p2], device=dev)
            except Exception as e:                       # noqa: BLE001 — a leg after the timed region must never take the line down
                out['run_e2e'] = {'error': '%s: %s' % (type(e).__name__, str(e)[:300])}
        final_out = out
    # ---- outside the timed region, every rank: run_mcl_clustering's inflation sweep (:2155-2158) shared out over the ranks
    # (sharded.sweep_sharded: ONE expansion across the ranks, the heavy iterations of the low inflations row-sharded, the light
    # remainders dealt by predicted cost) — the N-rank counterpart of the one-GPU `sweep` leg
    # The line of the timed region is complete at this point.  The leg below runs collectives that no multi-GPU node has carried yet:
    # a watchdog prints the line without it and ends the process if it does not come back (every rank runs the same timer).
    sweep_sh = None
    if state.get('block') is not None:
        import threading

        def give_up():
            if final_out is not None:
                final_out['sweep_sharded'] = {'error': 'no result within %d s' % args.sharded_sweep_timeout}
                print(json.dumps(final_out), flush=True)
            os._exit(0)
        watchdog = threading.Timer(args.sharded_sweep_timeout, give_up)
        watchdog.daemon = True
        watchdog.start()
        from decimal import Decimal
        eng = sharded.HipEngine(dev)
        infl = [float(Decimal('1.1') + Decimal('0.1') * k) for k in range(args.sweep)]
        try:
            full = sharded.allgather_rows(eng, state.pop('block'), state['shape'], dist)
            barrier()
            sharded.STAGES = {}
            ts = time.perf_counter()
            res_sw = sharded.sweep_sharded(eng, full, infl, 200, 1e-4, dist)
            barrier()
            sweep_stages, sharded.STAGES = sharded.STAGES, None
            sweep_sh = {'seconds': time.perf_counter() - ts, 'stage_ms_bytes_calls': {k_: [round(v[0], 3), v[1], v[2]] for k_, v in sweep_stages.items()}, 'inflations': infl, 'iterations': [r[4] for r in res_sw], 'converged': [bool(r[5]) for r in res_sw],
                        'clusters': [int(len(r[0])) for r in res_sw], 'ranks': world,
                        'what': 'sharded.sweep_sharded: one expansion shared by the ranks (each holds its rows of M^2), dense epilogue + exchange per '
                                'inflation, heavy iterations row-sharded (>= %.0e products), light remainders dealt by predicted cost' % sharded.SWEEP_SHARD_PRODUCTS}
            if args.check_sweep and rank == 0:
                # functional runs (--transport host): the SAME all-gathered link matrix through the one-GPU sweep — iteration counts,
                # convergence flags and the attractor arrays (= the cluster sets) of every inflation must be those of the ranks' sweep
                from haphic_amd import cluster as _cl
                tc = time.perf_counter()
                sw1 = _cl.DenseSweep(full, 1e-4)
                same = []
                try:
                    for k_, first in enumerate(sw1.first_iterations(infl)):
                        r1, n1, c1 = _lib.mcl_resume(first, 1, 2, infl[k_], 200, 1e-4)
                        first.free()
                        a1 = _lib.interpret(r1)
                        r1.free()
                        rs = res_sw[k_]
                        same.append(bool(n1 == rs[4] and bool(c1) == bool(rs[5]) and all(np.array_equal(x, y) for x, y in zip(a1, rs[:3]))))
                finally:
                    sw1.close()
                sweep_sh['equals_one_gpu_sweep_on_the_same_matrix'] = same
                sweep_sh['all_equal'] = all(same)
                sweep_sh['one_gpu_sweep_seconds'] = time.perf_counter() - tc
            full.free()
        except Exception as e:                           # noqa: BLE001 — this leg must never take the line of the timed region down
            sweep_sh = {'error': '%s: %s' % (type(e).__name__, str(e)[:300])}
        watchdog.cancel()

    if rank == 0:
        if sweep_sh is not None:
            final_out['sweep_sharded'] = sweep_sh
        print(json.dumps(final_out))
    if sharded_path:
        dist.barrier()
        dist.destroy_process_group()


def text_leg(args, gen, id1, p1, id2, p2, dev):
    """a1 beside the headline: .pairs TEXT already in HBM -> id / position arrays (hhx_pairs_parse), with and without
    the alignments.bed bytes.  The text is text_lines real lines of this workload, tiled to text_tile copies."""
    import torch
    from haphic_amd import _lib
    k = min(args.text_lines, id1.numel())
    names = list(gen.names)
    h = [a[:k].cpu().numpy() for a in (id1, p1, id2, p2)]
    raw = ''.join('r%d\t%s\t%d\t%s\t%d\t+\t-\n' % (i, names[a], x + 1, names[b], y + 1)
                  for i, (a, x, b, y) in enumerate(zip(*[v.tolist() for v in h]))).encode()
    text = torch.frombuffer(bytearray(raw), dtype=torch.uint8).to(dev).repeat(args.text_tile)
    n_bytes, n_lines = text.numel(), k * args.text_tile
    ps = _lib.PairsParser(names)
    res = {'lines': n_lines, 'bytes': n_bytes, 'bytes_per_line': n_bytes / n_lines}
    for tag, bed in (('parse', False), ('parse_bed', True)):
        best = None
        for _ in range(3):
            _lib.check(_lib.load().hhx_synchronize())
            t0 = time.perf_counter()
            assert ps.parse(None, want_bed=bed, device_ptr=text.data_ptr(), n_bytes=n_bytes) == n_lines
            _lib.check(_lib.load().hhx_synchronize())
            dt = time.perf_counter() - t0
            best = dt if best is None else min(best, dt)
        res[tag] = {'ms': best * 1e3, 'text_GBs': n_bytes / best / 1e9, 'pairs_per_s': n_lines / best}
        if bed:
            res[tag]['bed_bytes'] = ps.bed_bytes
    _lib.profile_reset()
    _lib.profile_enable(True)
    ps.parse(None, want_bed=True, device_ptr=text.data_ptr(), n_bytes=n_bytes)
    _lib.check(_lib.load().hhx_synchronize())
    _lib.profile_enable(False)
    res['kernel_ms'] = {kname: _lib.profile_get(kname)[0] for kname in ('text_breaks', 'text_starts', 'text_parse', 'text_bed')}
    # algorithmic bytes of the parse kernel: the text once + 16 B of arrays per line
    res['roofline'] = {'kernel': 'k_parse_lines', 'bound': 'hbm', 'achieved': (n_bytes + 16 * n_lines) / (res['kernel_ms']['text_parse'] * 1e-3) / 1e9,
                       'peak': 8000.0, 'unit': 'GB/s'}
    res['roofline']['frac'] = res['roofline']['achieved'] / 8000.0
    # end to end from a FILE: .pairs text on disk (page cache) -> mmap -> device tokeniser -> group-by -> link matrix
    import shutil
    import tempfile
    from haphic_amd import cluster
    with tempfile.TemporaryDirectory() as td:
        path = os.path.join(td, 'bench.pairs')
        # the file + its alignments.bed (1.35 x the text) must fit the temporary directory with room to spare
        room = shutil.disk_usage(td).free
        tile = max(1, min(args.text_file_tile, int(room * 0.6 / (2.4 * len(raw)))))
        with open(path, 'wb') as f:
            for _ in range(tile):
                f.write(raw)
        n_file = k * tile
        res['file_tile'] = tile
        tb = cluster.FragTable.for_contigs(gen.lexical_rank(), gen.length, np.ones(gen.n, np.uint8), names=names)
        for tag, bed_path in (('file_to_link_matrix', None), ('file_to_link_matrix_with_bed', os.path.join(td, 'alignments.bed'))):
            best = None
            for _ in range(2):
                t0 = time.perf_counter()
                aln = cluster.pairs_generator_inter_ctgs(path, 'pairs')
                aln.bed_path = bed_path
                ing = _lib.Ingest(tb, 500_000, bins=False, skip_intra=True)
                for parser, kk in aln.batches(names):
                    ing.push_device(kk, *parser.device_arrays()[:4])
                ing.finalize()
                mm, _fidx, _nl = ing.link_matrix(np.ones(gen.n, np.uint8))
                _lib.check(_lib.load().hhx_synchronize())
                dt = time.perf_counter() - t0
                _lib.files_join()                    # alignments.bed is written behind the stage (hhx_byte_sink); its completion is timed by run_e2e
                mm.free()
                ing.destroy()
                best = dt if best is None else min(best, dt)
            res[tag] = {'lines': n_file, 'file_bytes': len(raw) * tile, 'ms': best * 1e3, 'pairs_per_s': n_file / best,
                        'text_GBs': len(raw) * tile / best / 1e9}
            if bed_path:
                res[tag]['bed_bytes'] = os.path.getsize(bed_path)
    out = ps.fetch()                                 # every tile must reproduce the sample it was formatted from
    for c in range(4):
        assert np.array_equal(out[c].reshape(args.text_tile, k), np.broadcast_to(h[c], (args.text_tile, k))), 'text leg mismatch'
    ps.destroy()
    return res


def seam_leg(args, gen, id1, p1, id2, p2):
    """The drop-in as run() :2869-2935 drives it, through the reference's own operator API (the mirrors of haphic_amd/cluster.py
    that patch_reference binds over HapHiC_cluster): the read pairs as HOST id arrays (what a generator hands over; PCIe included)
    -> parse_alignments_for_ctgs (S5: the six containers, CLM side records kept) -> filter_fragments (f1, the reference's default
    thresholds) -> dict_to_matrix (S4), and beside it the three files run() writes on the way (HT_links.pkl :2879, paired_links.clm
    :2888, full_links.pkl :2929).  The containers are array-backed (haphic_amd/containers.py): no Python object per key."""
    import shutil
    import tempfile
    import types
    import torch
    from haphic_amd import _lib, cluster
    t_prep = time.perf_counter()
    host = [a.cpu().numpy() for a in (id1, p1, id2, p2)]
    names = list(gen.names)
    lengths = gen.length.tolist()
    fa_dict = {nm: [None, int(ln), int(ln) // 256 + 1] for nm, ln in zip(names, lengths)}      # GATC density of uniform ACGT
    ctg_len_dict = {nm: v[1] for nm, v in fa_dict.items()}
    re_dict = {nm: v[2] for nm, v in fa_dict.items()}
    nx = set(names)
    aln = cluster.IdArrays(names, *host)
    aln.inter_only = True                       # pairs_generator_inter_ctgs :1582: ref == mref never reaches the loop
    a = types.SimpleNamespace(flank=500, remove_allelic_links=0, remove_concentrated_links=False, max_read_pairs=200, nwindows=50)
    prep_s = time.perf_counter() - t_prep
    P = len(aln)
    import logging
    lg = logging.getLogger('HapHiC_cluster')
    level = lg.level
    lg.setLevel(logging.WARNING)
    sync = lambda: _lib.check(_lib.load().hhx_synchronize())            # noqa: E731
    files = None
    try:
        t0 = time.perf_counter()
        full, flank, HT, clm, frag_link, coord = cluster.parse_alignments_for_ctgs(aln, fa_dict, a, ctg_len_dict, nx, 'int32', 'int32')
        sync()
        t1 = time.perf_counter()
        kept = cluster.filter_fragments(nx, re_dict, 5, frag_link, '0.2X', '1.9X', 10, '1.5X', 0, flank, {}, '1.5X', set())
        sync()
        t2 = time.perf_counter()
        m, idx = cluster.dict_to_matrix(flank, kept, dense_matrix=False, add_self_loops=True)
        sync()
        t3 = time.perf_counter()
        frozen = all(c.frozen for c in (full, flank, HT, clm))
        dev = m.take_device()
        shape = dev.shape3
        dev.free()
        out = {'pairs': P, 'pairs_per_s': P / (t3 - t0), 'seconds': t3 - t0,
               's5_parse_alignments_for_ctgs_s': t1 - t0, 'filter_fragments_s': t2 - t1, 'dict_to_matrix_s': t3 - t2,
               'fragments_kept': len(kept), 'matrix_order': int(shape[0]), 'matrix_nnz': int(shape[2]), 'containers_still_frozen': frozen,
               'host_prep_s_untimed': prep_s,
               'what': 'cluster.parse_alignments_for_ctgs(IdArrays on the HOST, all %d pairs, CLM / HT side records on) -> cluster.filter_fragments '
                       '(reference defaults) -> cluster.dict_to_matrix(add_self_loops): the seam sequence of run() :2869-2935 as patch_reference binds it; '
                       'wall clock, PCIe included' % P}
        if not args.no_seam_files:
            # where the three files go: a RAM disk when it has the room (the pipeline, not the box's disk, is what is measured; said so)
            need = 70 * P                       # ~52 B of CLM text + ~10 B of pickles per pair, with margin
            where = None
            try:
                import psutil
                ram_ok = psutil.virtual_memory().available > 3 * need       # a RAM disk's pages count as this process's memory
            except ImportError:
                ram_ok = False
            for cand in (['/dev/shm'] if ram_ok else []) + [tempfile.gettempdir()]:
                try:
                    if shutil.disk_usage(cand).free > need + (4 << 30):
                        where = cand
                        break
                except OSError:
                    pass
            if where is None:
                out['files'] = {'skipped': 'no directory with %.0f GB free' % (need / 1e9)}
            else:
                d = tempfile.mkdtemp(prefix='hhx_seam_', dir=where)
                cwd = os.getcwd()
                os.chdir(d)
                try:
                    f0 = time.perf_counter()
                    cluster.output_pickle(HT, 'HT_link_dict', 'HT_links.pkl')
                    cluster.output_clm(clm)
                    cluster.output_pickle(full, 'full_link_dict', 'full_links.pkl')
                    f1 = time.perf_counter()                 # the three calls return once the files are queued on the library's writer thread
                    _lib.files_join()
                    f3 = time.perf_counter()
                    size = {f: os.path.getsize(f) for f in ('HT_links.pkl', 'paired_links.clm', 'full_links.pkl')}
                    files = {'directory': where, 'three_calls_s': f1 - f0, 'files_complete_s': f3 - f0, 'bytes': size,
                             'full_keys': len(full), 'written_GBs': sum(size.values()) / (f3 - f0) / 1e9,
                             'containers_still_frozen': all(c.frozen for c in (full, flank, clm))}
                    out['files'] = files
                    out['run_shaped_pairs_per_s'] = P / ((t3 - t0) + (f1 - f0))       # what the caller of the seams waits for
                    out['run_shaped_seconds'] = (t3 - t0) + (f1 - f0)
                    out['run_shaped_seconds_until_the_files_are_complete'] = (t3 - t0) + (f3 - f0)
                finally:
                    os.chdir(cwd)
                    shutil.rmtree(d, ignore_errors=True)
        del full, flank, HT, clm, coord, frag_link
        # what the generic dict path costs when something thaws a table (remove_allelic_HiC_links, --remove_concentrated_links, user
        # code): measured on a 20 M-pair prefix of the stream — thawing is linear in the keys, and 1.2e8 tuples would take ~25 GB of host RAM
        small = cluster.IdArrays(names, *[h[:20_000_000] for h in host])
        small.inter_only = True
        full, flank, HT, clm, frag_link, coord = cluster.parse_alignments_for_ctgs(small, fa_dict, a, ctg_len_dict, nx, 'int32', 'int32')
        th0 = time.perf_counter()
        n_keys = len(flank) + len(full)
        flank._thaw()
        full._thaw()
        th = time.perf_counter() - th0
        out['thaw'] = {'keys': n_keys, 'seconds': th, 'keys_per_s': n_keys / th if th else None, 'sample_pairs': len(small),
                       'what': 'full_link_dict + flank_link_dict of a 20 M-pair prefix turned into real dicts (dict.update(zip(...)) at C speed)'}
        del full, flank, HT, clm, coord
        return out
    finally:
        lg.setLevel(level)
        torch.cuda.empty_cache()


def sweep_leg(args, m, one_mcl_s):
    """Outside the timed region: run_mcl_clustering's inflation sweep (:2155-2158 — every inflation restarts mcl() from the matrix
    pre-expanded at :2146-2147) on the link matrix of the last step, with ONE expansion: the rows of M^2 are stored as float32
    row blocks by the window kernel's dense mode (cluster.DenseSweep / hhx_expand_links_dense), iteration 0 of each inflation is
    the epilogue over them (hhx_dense_inflate_prune), the loop resumes with hhx_mcl_resume.  Low inflations prune little: their
    tails are long by nature (the reference's too), so the tails run from the highest inflation down within a time budget."""
    from decimal import Decimal
    from haphic_amd import _lib, cluster
    sync = lambda: _lib.check(_lib.load().hhx_synchronize())
    inflations = [Decimal('1.1') + Decimal('0.1') * k for k in range(args.sweep)]
    n = m.shape3[0]
    t0 = time.perf_counter()
    sw = cluster.DenseSweep(m, 1e-4)
    order = inflations[::-1]                                         # highest inflation first: the small matrices first
    out = {'inflations': [str(x) for x in order], 'row_blocks': len(sw.bounds) - 1}
    firsts, errors = [], {}
    ta = time.perf_counter()
    if len(sw.bounds) == 2:                                          # M^2 fits as one resident block (n = 100k: 40 GB)
        warm = sw._prewarm(min(cluster.DenseSweep.GROUP, len(inflations)))      # as DenseSweep.first_iterations does: the first group's pools, allocated while the expansion runs
        blk = _lib.DenseRows(m, 0, n)                                # the expansion, once
        sw.n_products = blk.n_products
        sync()
        tb = time.perf_counter()
        _lib.profile_reset()
        _lib.profile_enable(True)
        ep = {}
        G = cluster.DenseSweep.GROUP                                 # inflations per pass over the block (hhx_dense_inflate_prune_multi), as run_mcl_clustering takes them
        for lo in range(0, len(inflations), G):                      # ascending: the pools sized by the previous (larger) demand fit
            grp = inflations[lo:lo + G]
            te = time.perf_counter()
            try:
                firsts.extend(blk.inflate_prune_multi([float(x) for x in grp], 1e-4))
            except RuntimeError as e:                                # e.g. a first iteration of more than 2^31 entries at inflation 1.1
                for infl in grp:                                     # one by one: only the inflation that fails is lost
                    try:
                        firsts.append(blk.inflate_prune(float(infl), 1e-4))
                    except RuntimeError as e1:
                        firsts.append(None)
                        errors[str(infl)] = str(e1)[:200]
            sync()
            ep[' '.join(str(x) for x in grp)] = (time.perf_counter() - te) * 1e3
        if warm is not None:
            warm.join()                                              # (it has long finished: ~0.4 s of allocations started with the expansion)
        firsts = firsts[::-1]
        out['epilogue_group_ms'] = ep
        out['pools_prewarmed_during_the_expansion'] = warm is not None
        out['fresh_pool_GB_during_the_epilogues'] = _lib.profile_counter('pool_fresh_bytes') / 1e9
        sync()
        tc = time.perf_counter()
        _lib.profile_enable(False)
        blk.free()
        out['expansion_ms'] = (tb - ta) * 1e3
        out['epilogues_all_inflations_ms'] = (tc - tb) * 1e3
    else:                                                            # several row blocks, each expanded once and used by every inflation
        _lib.profile_reset()
        _lib.profile_enable(True)
        firsts = list(sw.first_iterations(order))
        sync()
        tc = time.perf_counter()
        _lib.profile_enable(False)
    ep_ms, ep_n = _lib.profile_get('dense_epilogue')
    out.update(iteration0_all_inflations_ms=(tc - ta) * 1e3, dense_epilogue_kernel_ms_avg=ep_ms / ep_n if ep_n else None,
               dense_bytes=4.0 * n * n, t1_nnz=[f.nnz if f is not None else None for f in firsts], products_walked=sw.n_products,
               one_fused_mcl_ms=one_mcl_s * 1e3, errors=errors,
               note='iteration 0 of all inflations = ONE pass over the products (expansion_ms) + one epilogue per inflation; the per-inflation '
                    'alternative (hhx_mcl_links at every inflation) walks the products %d times' % args.sweep)
    # iteration 0 at 2.0 must be the fused kernel's bits
    if Decimal('2.0') in inflations:
        k = inflations[::-1].index(Decimal('2.0'))
        one = _lib.mcl(m, 2, 2.0, 1, 1e-4, links=True)[0]
        out['iteration0_bit_identical_to_fused_at_2.0'] = bool(firsts[k] is not None and all(np.array_equal(x, y) for x, y in zip(one.to_arrays(), firsts[k].to_arrays())))
        one.free()
    tails = {}
    t_budget = time.perf_counter()
    for infl, first in zip(order, firsts):
        if first is None:
            continue
        if time.perf_counter() - t_budget > args.sweep_tail_seconds or first.nnz > 300_000_000:      # (a first iteration that large: minutes of tail)
            first.free()
            continue
        t1 = time.perf_counter()
        try:
            res, n_iter, conv = _lib.mcl_resume(first, 1, 2, float(infl), 200, 1e-4)
        except RuntimeError as e:
            errors['tail ' + str(infl)] = str(e)[:200]
            first.free()
            continue
        first.free()
        att, ptr, mem = _lib.interpret(res)
        res.free()
        tails[str(infl)] = {'ms': (time.perf_counter() - t1) * 1e3, 'iterations': int(n_iter), 'converged': bool(conv), 'clusters': int(len(att))}
    sw.close()
    out['tails'] = tails
    out['tails_skipped_time_or_size_budget'] = [str(x) for x in order if str(x) not in tails and str(x) not in errors and 'tail ' + str(x) not in errors]
    out['seconds'] = time.perf_counter() - t0
    return out


def parity_leg(args, m, rows=64):
    """Outside the timed region: iteration 0 of the MCL on the link matrix of the last step (the kernel instantiation the
    roofline is quoted on) against the oracle — the C restatement of the reference's expand / inflate / prune in the
    kernels' exact fixed-point specification — on `rows` sampled rows of the real operand.  Bit equality expected."""
    from haphic_amd import _lib
    from oracle import oracle as orc
    orc.set_threads(0)                                   # all host cores: this is the checker, not the timed baseline
    t0 = time.perf_counter()
    integer_dev, layout = _lib.links_plan(m)             # what iteration 0 does with this matrix on this device (label below)
    one = _lib.mcl(m, 2, args.inflation, 1, 1e-4, links=True)[0]
    gp, gj, gx = one.to_arrays()
    # iteration 1 (T1 x T1, the hash class) on sampled rows of the real T1
    n1 = one.shape3[0]
    rows1 = np.sort(np.random.default_rng(6).choice(n1, min(rows, n1), replace=False))
    bad1 = 0
    prod1 = 0
    for r in rows1:
        blk = one.row_block(int(r), int(r) + 1)
        got1 = _lib.expand_inflate_prune(blk, one, args.inflation, 1e-4)[0].to_arrays()
        blk.free()
        lo, hi = gp[r], gp[r + 1]
        c1 = orc.spgemm((np.array([0, hi - lo], np.int32), gj[lo:hi], gx[lo:hi]), (gp, gj, gx), n_cols=n1, mode=1, fx_shift=52)
        w1 = orc.prune((c1[0], c1[1], orc.normalize_l1(c1[0], orc.power(c1[2], args.inflation))), 1e-4)
        prod1 += int(np.diff(gp)[gj[lo:hi]].sum())
        if not (np.array_equal(got1[1], w1[1]) and np.array_equal(got1[2], w1[2])):
            bad1 += 1
    one.free()
    mp, mj, mx = m.to_arrays()
    n = m.shape3[0]
    m.free()
    pick_rows = np.sort(np.random.default_rng(5).choice(n, min(rows, n), replace=False))
    integer = orc.links_shift((mp, mj, mx)) > 0           # the kernels took the integer arithmetic (symmetric counts, row sums < 2^18)
    if integer:
        c = orc.expand_links((mp, mj, mx), rows=pick_rows)
    else:
        norm = orc.normalize_l1(mp, mx)
        sub_p = np.zeros(len(pick_rows) + 1, np.int32)
        sub_p[1:] = np.cumsum(mp[pick_rows + 1] - mp[pick_rows])
        take = np.concatenate([np.arange(mp[r], mp[r + 1]) for r in pick_rows])
        c = orc.spgemm((sub_p, mj[take], norm[take]), (mp, mj, norm), n_cols=n, mode=1, fx_shift=52)
    x = orc.normalize_l1(c[0], orc.power(c[2], args.inflation))
    want = orc.prune((c[0], c[1], x), 1e-4)
    bad = 0
    for k, r in enumerate(pick_rows):
        lo, hi, wl, wh = gp[r], gp[r + 1], want[0][k], want[0][k + 1]
        if not (np.array_equal(gj[lo:hi], want[1][wl:wh]) and np.array_equal(gx[lo:hi], want[2][wl:wh])):
            bad += 1
    path = ('integer arithmetic, ' if integer_dev else 'float arithmetic, ') + {
        0: 'every row walks all its products into the fused epilogue',
        1: 'symmetric half into the square dense block + transposition; inflate + prune from the dense rows',
        2: 'symmetric half into the upper block triangle alone; inflate + prune block row by block row'}[layout]
    return {'what': 'iteration 0 (pre-expansion: %s) vs oracle, sampled rows of the real operand' % path,
            'rows_checked': int(len(pick_rows)), 'rows_differing': int(bad), 'bit_identical': bad == 0,
            'products_checked': int(sum(int(np.diff(mp)[mj[mp[r]:mp[r + 1]]].sum()) for r in pick_rows)),
            'specification': 'integer (S = L D^-1 L exact, orc_expand_links)' if integer else 'fixed point 2^-52 (orc_spgemm mode 1)',
            'iteration1_hash_class': {'rows_checked': int(len(rows1)), 'rows_differing': int(bad1), 'bit_identical': bad1 == 0, 'products_checked': prod1},
            'oracle_threads': orc.get_threads(), 'seconds': time.perf_counter() - t0,
            'full_size_tests': 'tests/test_gpu_scale.py: C2 whole ingest + mcl() bit equal; C3 whole ingest (500 M pairs) + dict_to_matrix, EVERY row of '
                               'iteration 0 + the whole tail continued by the oracle to convergence, bit equal, sweep inflations 1.1 / 1.4 / 3.0; C5 (4 pushes) 1k rows + whole tail + '
                               'cross-push ingest prefix; C4 at 40k contigs: containers vs oracle, cluster files vs the reference run'}


def pmc_traffic(n_contigs, pairs):
    """HBM/fabric bytes per launch measured with rocprofv3 --pmc on this same command line (tools/pmc_summary.py
    writes profiles/pmc_traffic.json); bench.py cannot read PMC counters itself."""
    path = os.path.join(ROOT, 'profiles', 'pmc_traffic.json')
    try:
        with open(path) as f:
            d = json.load(f)
    except (OSError, ValueError):
        return {}
    if d.get('contigs') != int(n_contigs) or d.get('pairs_per_gpu') != int(pairs):
        return {}
    from haphic_amd import build
    if d.get('kernel_source_sha16') != build.source_hash():      # counters of other kernels than the ones running now: not reported
        return {}
    return d.get('bytes_per_launch', {})


def pick(traffic, *needles):
    """bytes per launch of the profiled kernel whose name contains every needle (the heaviest one if several do)"""
    hits = [v for name, v in traffic.items() if all(x in name for x in needles)]
    return max(hits) if hits else None


def cpu_baseline(args, gen, table, flank, id1, p1, id2, p2, state):
    """The oracle (a scalar C port of the reference algorithm, 1 thread) timed on this host on a bounded
    sample of the same workload: the first S pairs for ingest, and mcl() on the link matrix those
    pairs produce at 1/4 of the contigs (so that the CPU leg stays within ~30 s)."""
    from oracle import oracle as orc
    orc.set_threads(1)                                   # the scalar port: "cores": 1 below
    S = min(args.cpu_sample_pairs, id1.numel())
    h = [a[:S].cpu().numpy() for a in (id1, p1, id2, p2)]
    keep = h[0] != h[2]
    t = orc.FragTable(table.ctg_rank, table.ctg_len, table.ctg_frag0, table.ctg_split, 0, table.frag_rank,
                      table.frag_len, table.frag_nx)
    a = (h[0][keep], h[1][keep].astype(np.int64), h[2][keep], h[3][keep].astype(np.int64))
    t0 = time.perf_counter()
    r = orc.ingest(t, a[0], a[1], a[2], a[3], flank)
    dt = time.perf_counter() - t0
    # MCL leg: contigs of the first quarter of the chromosomes only
    nq = int((gen.chrom < max(1, gen.nchrs // 4)).sum())
    sel = (r['flank_i'] < nq) & (r['flank_j'] < nq)
    in_set = np.zeros(gen.n, np.uint8)
    in_set[:nq] = 1
    linked = np.zeros(gen.n, bool)
    linked[r['flank_i'][sel]] = True
    linked[r['flank_j'][sel]] = True
    p, j, x, fidx, nl = orc.dict_to_matrix(r['flank_i'], r['flank_j'], r['flank_cnt'].astype(np.float64), gen.n, in_set,
                                           int(nq - linked.sum()))
    xn = orc.normalize_l1(p, x)
    pre = orc.spgemm((p, j, xn), (p, j, xn), mode=0)
    t1 = time.perf_counter()
    res = orc.mcl(pre, 2, args.inflation, 200, 1e-4)
    dm = time.perf_counter() - t1
    # the same mcl() with the expansion and every row-local step spread over ALL host cores (OpenMP; the reference's own SpGEMM —
    # sparse_dot_mkl — is multithreaded too, its Python prune() is not): same result, bit for bit
    orc.set_threads(0)
    all_threads = orc.get_threads()
    t1 = time.perf_counter()
    res_all = orc.mcl(pre, 2, args.inflation, 200, 1e-4)
    dm_all = time.perf_counter() - t1
    orc.set_threads(1)
    assert res_all[3] == res[3] and np.array_equal(res_all[2], res[2])
    # a1 beside it: the reference's tokeniser loop (pairs_generator :1539-1559, restated in oracle.parse_pairs_text) on
    # 300 k lines of this workload's .pairs text
    names = list(gen.names)
    nt = min(300_000, S)
    text = ''.join('r%d\t%s\t%d\t%s\t%d\t+\t-\n' % (i, names[u], x + 1, names[v], y + 1)
                   for i, (u, x, v, y) in enumerate(zip(h[0][:nt].tolist(), h[1][:nt].tolist(), h[2][:nt].tolist(), h[3][:nt].tolist()))).encode()
    t2 = time.perf_counter()
    orc.parse_pairs_text(text, names)
    dtok = time.perf_counter() - t2
    # the reference's OWN Python: measured in this run when its checkout is here (the dev container), otherwise the stored
    # measurement of tools/reference_cpu_baseline.py is carried along and labelled as such
    ref_py = None
    if os.path.exists('/root/reference/scripts/HapHiC_cluster.py') and not args.no_reference_python:
        import subprocess
        try:
            r_ = subprocess.run([sys.executable, os.path.join(ROOT, 'tools', 'reference_cpu_baseline.py'), '3000', '2000000'], capture_output=True,
                                text=True, timeout=900, env=dict(os.environ, PYTHONHASHSEED='0'))
            ref_py = json.loads([l for l in r_.stdout.splitlines() if l.startswith('{')][-1])
            ref_py['measured_where'] = 'THIS run, this host (the reference checkout is present)'
            ref_py['stored'] = False
        except Exception as e:                           # noqa: BLE001 — a baseline leg must not take the bench line down
            ref_py = {'error': str(e)[:200]}
    if ref_py is None or 'error' in ref_py:
        try:
            with open(os.path.join(ROOT, 'profiles', 'r02_reference_python_baseline.json')) as f:
                ref_py = json.load(f)
            ref_py['measured_where'] = 'STORED: dev container, round 2 (the reference checkout does not exist on the GPU box): tools/reference_cpu_baseline.py'
            ref_py['stored'] = True
        except (OSError, ValueError):
            pass
    # the unmodified `haphic cluster` on configs[0] (tools/reference_c1_baseline.py, dev container) beside the same command
    # through the mirrors on one MI355X (tools/c1_run.py) — both stored measurements
    if isinstance(ref_py, dict):
        for key, fn in (('c1_haphic_cluster', 'r05_reference_c1_baseline.json'), ('c1_mirrors_mi355x', 'r05_c1_run_mi355x.json')):
            try:
                with open(os.path.join(ROOT, 'profiles', fn)) as f:
                    c1 = json.load(f)
                c1.pop('inflation_dirs', None)
                c1.pop('log_tail', None)
                c1['stored'] = True
                ref_py[key] = c1
            except (OSError, ValueError):
                pass
    return {'value': S / dt, 'unit': 'pairs/s', 'cores': 1, 'kind': 'port', 'reference_python': ref_py,
            'sample': 'ingest: first %d pairs of the rank-0 shard through the C oracle (hash-map port of '
                      'parse_alignments_for_ctgs); mcl: oracle mcl() on the %d-contig sub-assembly (first quarter '
                      'of the chromosomes) built from those pairs' % (S, nq),
            'ingest_seconds': dt, 'mcl_iters_per_s': res[3] / dm, 'mcl_n': int(nq), 'mcl_iterations': int(res[3]),
            'mcl_all_cores': {'iters_per_s': res_all[3] / dm_all, 'threads': all_threads, 'what': 'the same oracle mcl(), expansion and row-local steps '
                              'row-parallel over all host cores (OpenMP), bit-identical result'},
            'text_tokeniser_pairs_per_s': nt / dtok, 'text_tokeniser_lines': nt, 'host_cpus': os.cpu_count(),
            'cpus_usable': orc.effective_cpus()}      # (affinity mask capped by the cgroup CPU quota: what "all cores" means for this process)


if __name__ == '__main__':
    main()
