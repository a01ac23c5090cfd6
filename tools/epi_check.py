"""Dense-epilogue check at the bench workload: ONE expansion of the link matrix into the dense float32 block (hhx_expand_links_dense),
then hhx_dense_inflate_prune at a few inflations; prints one JSON line with the time of every call and a SHA-256 of the pruned
matrix (indptr | indices | data).  Run it twice — HHX_DENSE_EPI_SW=0 (k_dense_epilogue, the plain slot layout and true
divisions) and default (k_dense_epilogue_sw) — and compare the digests: the two kernels must give the same bits for all rows.

  python tools/epi_check.py [--contigs 100000 --pairs 500000000 --nchrs 24 --mean-len 30000] [--inflations 2.0,1.4,3.0]
"""
import argparse
import hashlib
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--contigs', type=int, default=100_000)
    ap.add_argument('--pairs', type=int, default=500_000_000)
    ap.add_argument('--nchrs', type=int, default=24)
    ap.add_argument('--mean-len', type=int, default=30_000)
    ap.add_argument('--inflations', default='2.0,1.4,3.0')
    ap.add_argument('--repeat', type=int, default=2)
    a = ap.parse_args()
    import torch
    from haphic_amd import _lib, synth
    from haphic_amd.cluster import FragTable
    gen = synth.make_genome(a.nchrs, max(1, a.contigs // a.nchrs) * a.mean_len, a.mean_len, seed=12345)
    n = gen.n
    table = FragTable.for_contigs(gen.lexical_rank(), gen.length, np.ones(n, np.uint8))
    dev = synth.sample_pairs(gen, a.pairs, seed=12345, device='cuda:0')
    ing = _lib.Ingest(table, 500_000, bins=False, skip_intra=True)
    ing.push_device(a.pairs, *[x.data_ptr() for x in dev])
    torch.cuda.synchronize()
    ing.finalize()
    del dev
    torch.cuda.empty_cache()
    m, _, _ = ing.link_matrix(np.ones(n, np.uint8))
    ing.destroy()
    sync = lambda: _lib.check(_lib.load().hhx_synchronize())
    out = {'kernel': 'k_dense_epilogue' if os.environ.get('HHX_DENSE_EPI_SW') == '0' else 'k_dense_epilogue_sw', 'n': int(m.shape3[0]),
           'nnz': int(m.nnz), 'calls': []}
    t0 = time.perf_counter()
    blk = _lib.DenseRows(m, 0, m.shape3[0])
    sync()
    out['expansion_ms'] = (time.perf_counter() - t0) * 1e3
    for infl in [float(x) for x in a.inflations.split(',')]:
        ms = []
        for _ in range(a.repeat):
            sync()
            t0 = time.perf_counter()
            r = blk.inflate_prune(infl, 1e-4)
            sync()
            ms.append((time.perf_counter() - t0) * 1e3)
            arrays = r.to_arrays()
            r.free()
        h = hashlib.sha256()
        for x in arrays:
            h.update(np.ascontiguousarray(x).tobytes())
        out['calls'].append({'inflation': infl, 'ms': [round(v, 2) for v in ms], 'nnz': int(arrays[0][-1]), 'sha256': h.hexdigest()})
    print(json.dumps(out))


if __name__ == '__main__':
    main()
