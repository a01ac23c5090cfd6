"""the tail of mcl() at inflation 1.1 (C3) takes 7.4 s in tools/lowtails.py and 9.0 s inside bench.py's sweep leg.  What differs is the
state of the library's memory pool: this probe runs the SAME tail (hhx_mcl_resume from the same iteration-0 matrix) with the pool's cache empty, warm
(the cache the previous run of the same tail left), and in the state bench.py's sweep leaves it in (the twenty first-iteration matrices resident, the
cache shaped by twenty epilogues), and counts the fresh device allocations of each run (bytes, calls, time on the caller's thread).
    python tools/tail_alloc_probe.py [inflation]          one JSON object"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import torch
    from haphic_amd import _lib, synth, cluster
    from haphic_amd.cluster import FragTable
    infl = float(sys.argv[1]) if len(sys.argv) > 1 else 1.1
    _lib.check(_lib.load().hhx_set_device(0))
    gen = synth.make_genome(24, 100_000 // 24 * 30_000, 30_000, seed=12345)
    n = gen.n
    table = FragTable.for_contigs(gen.lexical_rank(), gen.length, np.ones(n, np.uint8))
    parts = [synth.sample_pairs(gen, 250_000_000, seed=12345 + 1 + 1000 * k, device='cuda:0') for k in range(2)]
    ing = _lib.Ingest(table, 500_000, bins=False, skip_intra=True)
    for q in parts:
        ing.push_device(q[0].numel(), *[t.data_ptr() for t in q])
    ing.finalize()
    m, _fidx, _nl = ing.link_matrix(np.ones(n, np.uint8))
    ing.destroy()
    del parts
    torch.cuda.empty_cache()
    sync = lambda: _lib.check(_lib.load().hhx_synchronize())            # noqa: E731
    blk = _lib.DenseRows(m, 0, n)
    first = blk.inflate_prune(infl, 1e-4)
    out = {'inflation': infl, 'n': int(n), 'first_iteration_nnz': int(first.nnz)}

    def run(tag, before=None):
        if before:
            before()
        sync()
        _lib.profile_reset()
        _lib.profile_enable(True)
        f = first.copy()
        t = time.perf_counter()
        res, n_iter, conv = _lib.mcl_resume(f, 1, 2, infl, 200, 1e-4)
        sync()
        dt = time.perf_counter() - t
        _lib.profile_enable(False)
        f.free()
        res.free()
        pc = _lib.profile_counter
        out[tag] = {'seconds': dt, 'iterations': n_iter, 'fresh_GB': pc('pool_fresh_bytes') / 1e9, 'fresh_calls': pc('pool_fresh_calls'), 'fresh_alloc_s': pc('pool_fresh_us') / 1e6,
                    'trims_on_failure': pc('pool_trims_on_failure'), 'pool_retries': pc('expand_pool_retries'), 'pool_cached_GB_after': _lib.pool_cached_bytes() / 1e9, 'free_GB_after': _lib.mem_info()[0] / 1e9,
                    'window_kernel_ms': _lib.profile_get('expand_window')[0] + _lib.profile_get('expand_window_short')[0], 'hash_kernel_ms': _lib.profile_get('expand_hash')[0]}
    run('cache_empty', lambda: _lib.check(_lib.load().hhx_pool_trim()))
    run('cache_warm')
    run('cache_warm_again')
    # the state of bench.py's sweep leg: iteration 0 of all twenty inflations formed first (their matrices stay resident), then the tails from 3.0 down
    from decimal import Decimal
    infls = [float(Decimal('1.1') + Decimal('0.1') * k) for k in range(20)]
    _lib.check(_lib.load().hhx_pool_trim())
    firsts = []
    for lo in range(0, 20, 5):
        firsts += blk.inflate_prune_multi(infls[lo:lo + 5], 1e-4)
    for k in range(19, 0, -1):                                   # the other nineteen tails, highest inflation first
        r, _n, _c = _lib.mcl_resume(firsts[k], 1, 2, infls[k], 200, 1e-4)
        r.free()
    run('after_the_other_19_tails_with_20_first_iterations_resident')
    for f in firsts:
        f.free()
    first.free()
    blk.free()
    m.free()
    print(json.dumps(out))


if __name__ == '__main__':
    main()
