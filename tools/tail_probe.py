"""The mcl() tails of the low inflations of run_mcl_clustering's sweep (:2155-2158) on the BASELINE configs[2] link matrix, iteration by
iteration (VERDICT r03 "weak" #8: 14 of the 18.5 s of the 20-inflation sweep are the tails at 1.1-1.3 and nothing there was profiled).
One JSON line per inflation: per iteration the wall time, entries of the operand, products, entries of the expanded rows, survivors,
and the kernel-class times of the library's own profile (hhx_profile_get); run under rocprofv3 --kernel-trace --stats for the kernel
shares (tools/gpu_pass.sh tails)."""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

KERNELS = ('expand_hash', 'expand_window', 'expand_window_short', 'expand_finalize', 'expand_compact', 'expand_tiny', 'class_layout',
           'convergence', 'inflate_stats', 'prune_write', 'dense_epilogue')


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--contigs', type=int, default=100000)
    ap.add_argument('--pairs', type=int, default=500_000_000)
    ap.add_argument('--nchrs', type=int, default=24)
    ap.add_argument('--mean-len', type=int, default=30000)
    ap.add_argument('--inflations', default='1.1,1.2,1.3')
    ap.add_argument('--max-iter', type=int, default=200)
    args = ap.parse_args()
    import torch
    from haphic_amd import _lib, synth
    from haphic_amd.cluster import FragTable
    dev = 'cuda:0'
    gen = synth.make_genome(args.nchrs, max(1, args.contigs // args.nchrs) * args.mean_len, args.mean_len, seed=12345)
    n = gen.n
    table = FragTable.for_contigs(gen.lexical_rank(), gen.length, np.ones(n, np.uint8))
    id1, p1, id2, p2 = synth.sample_pairs(gen, args.pairs, seed=12345, device=dev)
    torch.cuda.synchronize()
    ing = _lib.Ingest(table, 500_000, bins=False, skip_intra=True)
    ing.push_device(id1.numel(), id1.data_ptr(), p1.data_ptr(), id2.data_ptr(), p2.data_ptr())
    ing.finalize()
    m, _fidx, _nl = ing.link_matrix(np.ones(n, np.uint8))
    ing.destroy()
    del id1, p1, id2, p2
    torch.cuda.empty_cache()
    sync = lambda: _lib.check(_lib.load().hhx_synchronize())
    t0 = time.perf_counter()
    blk = _lib.DenseRows(m, 0, m.shape3[0])
    sync()
    print(json.dumps({'n': n, 'nnz': m.nnz, 'expansion_ms': (time.perf_counter() - t0) * 1e3, 'products': blk.n_products}), flush=True)
    for infl in [float(x) for x in args.inflations.split(',')]:
        t0 = time.perf_counter()
        cur = blk.inflate_prune(infl, 1e-4)
        sync()
        rows = [{'it': 0, 'ms': (time.perf_counter() - t0) * 1e3, 'survivors': cur.nnz}]
        t_all = time.perf_counter()
        _lib.profile_enable(True)
        conv = False
        for it in range(1, args.max_iter):
            _lib.profile_reset()
            t1 = time.perf_counter()
            nxt, f, z = _lib.expand_inflate_prune(cur, cur, infl, 1e-4)
            sync()
            t2 = time.perf_counter()
            d = _lib.convergence_stat(nxt, cur) if it > 1 else 1.0
            t3 = time.perf_counter()
            k = {name: round(_lib.profile_get(name)[0], 2) for name in KERNELS if _lib.profile_get(name)[1]}
            rows.append({'it': it, 'ms': round((t2 - t1) * 1e3, 2), 'conv_ms': round((t3 - t2) * 1e3, 2), 'nnz_a': cur.nnz, 'products': f, 'nnz_c': z,
                         'survivors': nxt.nnz, 'kernels_ms': k, 'window_products': _lib.profile_counter('expand_window_products') +
                         _lib.profile_counter('expand_window_short_products')})
            cur.free()
            cur = nxt
            if it > 1 and d <= 1e-8:
                conv = True
                break
        _lib.profile_enable(False)
        total = (time.perf_counter() - t_all) * 1e3
        att = _lib.interpret(cur)[0]
        cur.free()
        prod = sum(r.get('products', 0) for r in rows)
        print(json.dumps({'inflation': infl, 'iterations': len(rows), 'converged': conv, 'clusters': int(len(att)), 'tail_ms': round(total, 1),
                          'products': prod, 'products_per_s': prod / (total / 1e3), 'per_iteration': rows}), flush=True)
    blk.free()


if __name__ == '__main__':
    main()
