#!/usr/bin/env python3
"""how should run_mcl_clustering restart the inflations?  materialise M^2 once (hhx_spgemm) + mcl per inflation, or
fused iteration 0 per inflation (hhx_mcl_links)?  C2-sized matrix."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from haphic_amd import _lib, synth
from haphic_amd.cluster import FragTable
n_ctg, pairs = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (10000, 50_000_000)
gen = synth.make_genome(16, max(1, n_ctg // 16) * 50_000, 50_000, seed=12345)
table = FragTable.for_contigs(gen.lexical_rank(), gen.length, np.ones(gen.n, np.uint8))
a = synth.sample_pairs(gen, pairs, seed=1, device='cuda:0')
ing = _lib.Ingest(table, 500_000, bins=False, skip_intra=True)
ing.push_device(pairs, *[x.data_ptr() for x in a])
torch.cuda.synchronize()
m, _, _ = ing.link_matrix(np.ones(gen.n, np.uint8))
sync = lambda: _lib.check(_lib.load().hhx_synchronize())
for rep in range(2):
    norm = m.copy(); _lib.normalize_l1(norm); sync()
    t0 = time.perf_counter(); pre = _lib.spgemm(norm, norm, fx_shift=52); sync(); t1 = time.perf_counter()
    r, it, cv = _lib.mcl(pre, 2, 2.0, 200, 1e-4); sync(); t2 = time.perf_counter()
    r2, it2, cv2 = _lib.mcl(m, 2, 2.0, 200, 1e-4, links=True); sync(); t3 = time.perf_counter()
    print('n=%d nnz=%d: spgemm %.1f ms (nnz %d), mcl(pre) %.1f ms (%d it), fused mcl(links) %.1f ms (%d it)' % (
        gen.n, m.nnz, (t1 - t0) * 1e3, pre.nnz, (t2 - t1) * 1e3, it, (t3 - t2) * 1e3, it2))
    pre.free(); r.free(); r2.free(); norm.free()
