"""Does a narrower column window — a slice of the operand stream that fits the 256 MB Infinity Cache — pay in the heavy iterations of a low-inflation tail?  The tail at
inflation 1.1 (C3) under hhx_tune("cache_slice_mb", v): v = 0 (the default plan: 5 windows), then slices that force 6, 8 and 10 windows; every variant must give the same bits.
    python tools/tail_window_probe.py [inflation] [knob] [v1,v2,...]
With a knob named (any hhx_tune switch, e.g. row_order 0,1,2,0) the same tail is timed under each of its values instead."""
import hashlib
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import torch
    from haphic_amd import _lib, synth
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
    blk.free()
    out = {'inflation': infl, 'n': int(n), 'first_iteration_nnz': int(first.nnz)}
    ref = None
    knob = sys.argv[2] if len(sys.argv) > 2 else 'cache_slice_mb'
    values = [int(v) for v in sys.argv[3].split(',')] if len(sys.argv) > 3 else (0, 280, 230, 180, 130, 0)
    out['knob'] = knob
    for slice_mb in values:
        _lib.tune(knob, slice_mb if slice_mb or knob != 'cache_slice_mb' else None)
        f = first.copy()
        sync()
        _lib.profile_reset()
        _lib.profile_enable(True)
        t = time.perf_counter()
        res, n_iter, conv = _lib.mcl_resume(f, 1, 2, infl, 200, 1e-4)
        sync()
        dt = time.perf_counter() - t
        _lib.profile_enable(False)
        digest = hashlib.sha256(b''.join(np.ascontiguousarray(a).tobytes() for a in res.to_arrays())).hexdigest()[:16]
        ref = ref or digest
        key = '%s_%d' % (knob if knob != 'cache_slice_mb' else 'slice_mb', slice_mb)
        while key in out:
            key += '_again'
        out[key] = {'seconds': dt, 'iterations': n_iter, 'window_kernel_ms': _lib.profile_get('expand_window')[0] + _lib.profile_get('expand_window_short')[0],
                    'window_launches': _lib.profile_get('expand_window')[1] + _lib.profile_get('expand_window_short')[1], 'hash_kernel_ms': _lib.profile_get('expand_hash')[0], 'order_ms': _lib.profile_get('row_order')[0], 'same_bits': digest == ref}
        f.free()
        res.free()
    _lib.tune(knob, None)
    out['digest'] = ref                                  # of the whole result (indptr, indices, data): equal between two builds = the same bits
    first.free()
    m.free()
    print(json.dumps(out))


if __name__ == '__main__':
    main()
