"""Where the time of k_expand_group (the re-use kernel, hhx_tune("reuse", 4)) goes: ONE iteration of a low-inflation tail — the operand
after `--iteration` iterations at inflation 1.1 — expanded repeatedly, optionally with the LDS atomics of the re-use kernel switched off (HHX_GROUP_PROBE=1: garbage results, timing only;
one process per setting because the library reads the variable once).  Prints one JSON line: kernel ms per call of the class kernels."""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CONFIGS = {'c2': (10_000, 50_000_000, 16, 50_000), 'k24': (24_000, 120_000_000, 24, 30_000), 'c3': (100_000, 500_000_000, 24, 30_000)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--config', default='k24')
    ap.add_argument('--iteration', type=int, default=5)
    ap.add_argument('--reps', type=int, default=5)
    args = ap.parse_args()
    import torch
    from haphic_amd import _lib, synth
    from haphic_amd.cluster import FragTable
    contigs, pairs, nchrs, mean_len = CONFIGS[args.config]
    gen = synth.make_genome(nchrs, max(1, contigs // nchrs) * mean_len, mean_len, seed=12345)
    table = FragTable.for_contigs(gen.lexical_rank(), gen.length, np.ones(gen.n, np.uint8))
    id1, p1, id2, p2 = synth.sample_pairs(gen, pairs, seed=12345, device='cuda:0')
    torch.cuda.synchronize()
    ing = _lib.Ingest(table, 500_000, bins=False, skip_intra=True)
    ing.push_device(id1.numel(), id1.data_ptr(), p1.data_ptr(), id2.data_ptr(), p2.data_ptr())
    ing.finalize()
    m, _f, _n = ing.link_matrix(np.ones(gen.n, np.uint8))
    ing.destroy()
    del id1, p1, id2, p2
    torch.cuda.empty_cache()
    blk = _lib.DenseRows(m, 0, m.shape3[0])
    cur = blk.inflate_prune(1.1, 1e-4)
    blk.free()
    for _ in range(1, args.iteration):                       # the real iterations before the probed one (re-use off: exact results)
        nxt = _lib.expand_inflate_prune(cur, cur, 1.1, 1e-4)[0]
        cur.free()
        cur = nxt
    sync = lambda: _lib.check(_lib.load().hhx_synchronize())          # noqa: E731
    out = {'config': args.config, 'iteration': args.iteration, 'nnz': int(cur.nnz), 'probe': os.environ.get('HHX_GROUP_PROBE', '0')}
    for label, R in (('one_row_per_walk', 0), ('four_rows_per_walk', 4)):
        _lib.tune('reuse', R)
        res = _lib.expand_inflate_prune(cur, cur, 1.1, 1e-4)             # warm-up
        res[0].free()
        _lib.profile_reset()
        _lib.profile_enable(True)
        sync()
        t0 = time.perf_counter()
        for _ in range(args.reps):
            res = _lib.expand_inflate_prune(cur, cur, 1.1, 1e-4)
            res[0].free()
        sync()
        dt = (time.perf_counter() - t0) / args.reps
        _lib.profile_enable(False)
        out[label] = {'wall_ms': round(dt * 1e3, 2), 'products': res[1],
                      'kernel_ms': {k: round(_lib.profile_get(k)[0] / args.reps, 2) for k in ('expand_group', 'group_build', 'expand_window_short', 'expand_window', 'expand_hash', 'expand_finalize', 'class_layout') if _lib.profile_get(k)[1]}}
        _lib.tune('reuse', None)
    print(json.dumps(out), flush=True)


if __name__ == '__main__':
    main()
