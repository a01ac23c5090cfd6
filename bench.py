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
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--cpu-sample-pairs', type=int, default=20_000_000)
    return ap.parse_args()


def main():
    args = parse()
    import torch
    import torch.distributed as dist
    from haphic_amd import _lib, synth
    from haphic_amd.cluster import FragTable

    rank = int(os.environ.get('RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            raise SystemExit('launch with torch.distributed.run --nproc-per-node %d' % args.gpus)
    torch.cuda.set_device(local_rank)
    _lib.check(_lib.load().hhx_set_device(local_rank))
    dev = 'cuda:%d' % local_rank
    if world > 1:
        dist.init_process_group('nccl', device_id=torch.device(dev))

    per_chr = max(1, args.contigs // args.nchrs)
    gen = synth.make_genome(args.nchrs, per_chr * args.mean_len, args.mean_len, seed=12345)
    n = gen.n
    lex = gen.lexical_rank()
    table = FragTable.for_contigs(lex, gen.length, np.ones(n, np.uint8))
    flank = 500_000                                   # --flank 500 (kb), HapHiC_cluster.py default
    # this rank's shard of the read-pair stream, generated straight into HBM
    local_pairs = args.pairs // world
    id1, p1, id2, p2 = synth.sample_pairs(gen, local_pairs, seed=12345 + rank, device=dev)
    torch.cuda.synchronize()
    in_set = np.ones(n, np.uint8)

    if world > 1:
        from haphic_amd import sharded
    state = {}

    def step():
        t0 = time.perf_counter()
        ing = _lib.Ingest(table, flank, bins=False, skip_intra=True)
        ing.set_ordinal_base(rank * local_pairs)
        ing.push_device(id1.numel(), id1.data_ptr(), p1.data_ptr(), id2.data_ptr(), p2.data_ptr())
        n_full, n_flank = ing.finalize()
        if world > 1:
            m, n_linked, merged = sharded.merge_flank_and_build(ing, table, flank, False, in_set, dist, dev)
            n_full, n_flank = merged.n_full, merged.n_flank
            merged.destroy()
        else:
            # dict_to_matrix fused onto the device-resident table; link-less contigs get trailing indices
            m, fidx, n_linked = ing.link_matrix(in_set)
        _lib.check(_lib.load().hhx_synchronize())
        t1 = time.perf_counter()
        state['n_full'], state['n_flank'], state['nnz_link'] = n_full, n_flank, m.nnz
        ing.destroy()
        # ---- run_mcl_clustering :2144-2158 at one inflation
        t2 = time.perf_counter()
        if world > 1:
            _lib.normalize_l1(m)                                                # :2144
            res, n_iter, conv, stats = sharded.mcl_sharded(m, 2, args.inflation, 200, 1e-4, dist, dev)
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
        m.free()

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    _lib.profile_reset()
    _lib.profile_enable(True)
    barrier()
    t0 = time.perf_counter()
    t_ing = t_mcl = t_pre = 0.0
    iters = 0
    for _ in range(args.steps):
        step()
        t_ing += state['t_ingest']; t_mcl += state['t_mcl']; t_pre += state['t_pre']; iters += state['n_iter']
    barrier()
    elapsed = time.perf_counter() - t0
    _lib.profile_enable(False)
    tm = torch.tensor([elapsed, t_ing, t_mcl, t_pre], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(tm, op=dist.ReduceOp.MAX)
    elapsed, t_ing, t_mcl, t_pre = tm.tolist()

    if rank == 0:
        K = args.steps
        pairs_total = local_pairs * world * K
        value = pairs_total / t_ing
        stats = np.asarray(state['stats'])
        # ---- roofline of the dominant kernel of the headline metric: k_ingest
        ing_ms, ing_n = _lib.profile_get('ingest')
        # 16 B read per pair + 12 B written per distinct key of each table + 4 B per fragment
        alg_bytes = 16.0 * local_pairs + 12.0 * (state['n_full'] + state['n_flank']) + 4.0 * n
        ach = alg_bytes / (ing_ms / max(ing_n, 1) * 1e-3) / 1e9 if ing_ms else None
        roofline = {'kernel': 'k_ingest', 'bound': 'hbm', 'achieved': ach, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                    'frac': (ach / HBM_PEAK_GBS) if ach else None, 'traffic': None,
                    'alg_bytes_per_launch': alg_bytes, 'avg_launch_ms': ing_ms / max(ing_n, 1)}
        # ---- MCL: expansion kernel roofline with SURVEY §8d's byte model, 8*(nnz_A + F) read + 8*nnz_C written
        num_ms, num_n = _lib.profile_get('expand_window')
        sym_ms, sym_n = _lib.profile_get('expand_compact')
        infl_ms, _ = _lib.profile_get('inflate_stats')
        prw_ms, _ = _lib.profile_get('prune_write')
        cvg_ms, _ = _lib.profile_get('convergence')
        # fused expand+inflate+prune launches: read 8 B per entry of A and per product (gather of B rows),
        # write 8 B per surviving entry; the expanded matrix C never touches HBM
        sp_bytes = float((8 * (stats[:, 0] + stats[:, 3]) + 8 * stats[:, 2]).sum()) if len(stats) else 0.0
        sp_ach = sp_bytes * K / ((num_ms + sym_ms) * 1e-3) / 1e9 if (num_ms + sym_ms) else None
        # B_iter of SURVEY §8d summed over the iterations of one mcl() call
        b_iter = float((8 * (stats[:, 0] + stats[:, 3]) + 8 * stats[:, 1] + 8 * stats[:, 1] + 8 * stats[:, 2]
                        + 16 * stats[:, 2] + 12 * n).sum()) if len(stats) else 0.0
        mcl = {'iters_per_s': iters / t_mcl if t_mcl else None, 'n': int(n), 'iterations': int(state['n_iter']),
               'converged': bool(state['conv']), 'inflation': args.inflation, 'ms_per_mcl': t_mcl / K * 1e3,
               'pre_expansion': 'fused into iteration 0', 'clusters': state.get('clusters'),
               'alg_bytes_per_mcl': b_iter, 'alg_GBs': b_iter * K / t_mcl / 1e9 if t_mcl else None,
               'frac_hbm': b_iter * K / t_mcl / 1e9 / HBM_PEAK_GBS if t_mcl else None,
               'stats_nnzA_nnzC_nnzP_F': stats.tolist(),
               'kernel_ms_per_step': {'expand_compact': sym_ms / K, 'expand_window': num_ms / K, 'inflate_stats': infl_ms / K,
                                      'prune_write': prw_ms / K, 'convergence': cvg_ms / K},
               'roofline_expand': {'kernels': 'k_expand_window + k_expand_compact', 'bound': 'hbm', 'achieved': sp_ach, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                                           'frac': sp_ach / HBM_PEAK_GBS if sp_ach else None, 'traffic': None,
                                           'launches_per_step': (num_n + sym_n) / K}}
        out = {'metric': 'Hi-C pairs/s ingested (link-matrix build) + MCL iters/s', 'value': value, 'unit': 'pairs/s',
               'n_gpus': world, 'steps': K, 'warmup': args.warmup, 'ms_per_step': elapsed / K * 1e3,
               'higher_is_better': True, 'scaling': 'strong', 'vs_baseline': None, 'dtype': 'int32 keys / u64 fixed-point + f32 values',
               'data': 'synthetic',
               'config': {'workload': '%d contigs / %d pairs (whole job, %d per GPU), %d chr, mean contig %d bp, inflation %.1f, dense-block off'
                                      % (n, local_pairs * world, local_pairs, args.nchrs, args.mean_len, args.inflation),
                          'contigs': int(n), 'pairs_per_gpu': local_pairs, 'full_keys': int(state['n_full']),
                          'flank_keys': int(state['n_flank']), 'link_matrix_nnz': int(state['nnz_link'])},
               'ingest_ms_per_step': t_ing / K * 1e3,
               'ingest_kernels_ms': {k: _lib.profile_get(k)[0] / K for k in ('ingest', 'part_count1', 'part_scatter1', 'part_count2', 'part_scatter2', 'aggregate', 'compact', 'ingest_merge', 'link_matrix')},
               'mcl': mcl, 'roofline': roofline}
        if not args.no_cpu_baseline:
            out['cpu_baseline'] = cpu_baseline(args, gen, table, flank, id1, p1, id2, p2, state)
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def cpu_baseline(args, gen, table, flank, id1, p1, id2, p2, state):
    """The oracle (a scalar C port of the reference algorithm, 1 thread) timed on this host on a bounded
    sample of the same workload: the first S pairs for ingest, and mcl() on the link matrix those
    pairs produce at 1/4 of the contigs (so that the CPU leg stays within ~30 s)."""
    from oracle import oracle as orc
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
    return {'value': S / dt, 'unit': 'pairs/s', 'cores': 1, 'kind': 'port',
            'sample': 'ingest: first %d pairs of the rank-0 shard through the C oracle (hash-map port of '
                      'parse_alignments_for_ctgs); mcl: oracle mcl() on the %d-contig sub-assembly (first quarter '
                      'of the chromosomes) built from those pairs' % (S, nq),
            'ingest_seconds': dt, 'mcl_iters_per_s': res[3] / dm, 'mcl_n': int(nq), 'mcl_iterations': int(res[3]),
            'host_cpus': os.cpu_count()}


if __name__ == '__main__':
    main()
