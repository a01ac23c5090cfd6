#!/usr/bin/env python3
"""PCIe-inclusive ingest rate: the same link-matrix build with the read pairs handed over as HOST arrays
(hhx_ingest_push(on_device=0), 16 B per pair over PCIe from pageable numpy memory).  DESIGN.md quotes the result;
bench.py's `value` is the HBM-resident rate."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
from haphic_amd import _lib, synth  # noqa: E402
from haphic_amd.cluster import FragTable  # noqa: E402

n_pairs = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000_000
gen = synth.make_genome(24, 4161 * 30_000, 30_000, seed=12345)
table = FragTable.for_contigs(gen.lexical_rank(), gen.length, np.ones(gen.n, np.uint8))
dev = synth.sample_pairs(gen, n_pairs, seed=1, device='cuda:0')
host = [t.cpu().numpy() for t in dev]
in_set = np.ones(gen.n, np.uint8)
for rep in range(3):
    t0 = time.perf_counter()
    ing = _lib.Ingest(table, 500_000, bins=False, skip_intra=True)
    ing.push(*host)
    ing.finalize()
    m, _, _ = ing.link_matrix(in_set)
    _lib.check(_lib.load().hhx_synchronize())
    dt = time.perf_counter() - t0
    print('host arrays -> link matrix: %d pairs in %.1f ms = %.3g pairs/s (%.1f GB/s of pair bytes)' % (n_pairs, dt * 1e3, n_pairs / dt, 16 * n_pairs / dt / 1e9))
    ing.destroy(); m.free()
