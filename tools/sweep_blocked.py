#!/usr/bin/env python3
"""inflation sweep at n^2 >= 2^31: M^2 kept in HBM as row blocks (built once) + resume per inflation, against the
fused iteration 0 per inflation (hhx_mcl_links).  usage: sweep_blocked.py [contigs pairs]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from haphic_amd import _lib, cluster, synth
from haphic_amd.cluster import FragTable
n_ctg, pairs = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (100000, 500_000_000)
gen = synth.make_genome(24, max(1, n_ctg // 24) * 30_000, 30_000, seed=12345)
table = FragTable.for_contigs(gen.lexical_rank(), gen.length, np.ones(gen.n, np.uint8))
a = synth.sample_pairs(gen, pairs, seed=12345, device='cuda:0')
ing = _lib.Ingest(table, 500_000, bins=False, skip_intra=True)
ing.push_device(pairs, *[x.data_ptr() for x in a]); ing.finalize()
del a
torch.cuda.empty_cache()
m, _, _ = ing.link_matrix(np.ones(gen.n, np.uint8))
ing.destroy()
sync = lambda: _lib.check(_lib.load().hhx_synchronize())
n = m.shape3[0]
print('n', n, 'nnz', m.nnz, 'free GB', _lib.mem_info()[0] / 1e9, flush=True)
for infl in (2.0, 1.6):
    sync(); t0 = time.perf_counter()
    r, it, cv = _lib.mcl(m, 2, infl, 200, 1e-4, links=True); sync()
    print('fused per inflation: inflation %.1f  %d it  %.3f s' % (infl, it, time.perf_counter() - t0), flush=True)
    ref = r.to_arrays() if infl == 2.0 else None
    r.free()
    if infl == 2.0:
        ref2 = ref
norm = m.copy(); _lib.normalize_l1(norm); sync()
rows_per = (2 ** 31 - 1) // n
t0 = time.perf_counter()
blocks = []
for r0 in range(0, n, rows_per):
    blk = norm.row_block(r0, min(n, r0 + rows_per))
    blocks.append(_lib.spgemm(blk, norm, fx_shift=52)); blk.free()
sync()
print('M^2 as %d row blocks: %.3f s, %d entries, %.1f GB; free GB %.1f' % (len(blocks), time.perf_counter() - t0, sum(b.nnz for b in blocks),
      sum(b.nnz for b in blocks) * 8 / 1e9, _lib.mem_info()[0] / 1e9), flush=True)
for infl in (2.0, 1.6):
    sync(); t0 = time.perf_counter()
    res = cluster.mcl_device_blocked(blocks, 2, infl, 200, 1e-4); sync()
    print('blocked: inflation %.1f  %.3f s' % (infl, time.perf_counter() - t0), flush=True)
    if infl == 2.0:
        print('equal to the fused run:', all(np.array_equal(x, y) for x, y in zip(res.to_arrays(), ref2)), flush=True)
    res.free()
