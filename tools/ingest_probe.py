#!/usr/bin/env python3
"""Link-matrix build alone at the bench's C3 workload (500 M synthetic pairs, 100k contigs): wall time and the HIP-event
time of every kernel class of the ingest and of hhx_ingest_link_matrix.  One JSON line.
usage: ingest_probe.py [pairs] [contigs] [steps]"""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from haphic_amd import _lib, synth  # noqa: E402
from haphic_amd.cluster import FragTable  # noqa: E402

NAMES = ('ingest', 'map', 'part_count1', 'part_scatter1', 'part_count2', 'part_scatter2', 'part_count3', 'part_scatter3', 'aggregate',
         'link_matrix', 'd2m_first', 'd2m_count1', 'd2m_scatter1', 'd2m_count2', 'd2m_scatter2', 'd2m_row_first', 'd2m_rank', 'd2m_emit')


def main():
    pairs = int(sys.argv[1]) if len(sys.argv) > 1 else 500_000_000
    contigs = int(sys.argv[2]) if len(sys.argv) > 2 else 100_000
    steps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
    dev = 'cuda:0'
    _lib.check(_lib.load().hhx_set_device(0))
    gen = synth.make_genome(24, max(1, contigs // 24) * 30_000, 30_000, seed=12345)
    table = FragTable.for_contigs(gen.lexical_rank(), gen.length, np.ones(gen.n, np.uint8))
    id1, p1, id2, p2 = synth.sample_pairs(gen, pairs, seed=12346, device=dev)
    torch.cuda.synchronize()
    torch.cuda.empty_cache()
    in_set = np.ones(gen.n, np.uint8)
    wall = []
    for s in range(steps + 1):
        if s == 1:
            _lib.profile_reset()
            _lib.profile_enable(True)
        t0 = time.perf_counter()
        ing = _lib.Ingest(table, 500_000, bins=False, skip_intra=True)
        ing.push_device(pairs, id1.data_ptr(), p1.data_ptr(), id2.data_ptr(), p2.data_ptr())
        ing.finalize()
        m, _fidx, _nl = ing.link_matrix(in_set)
        _lib.check(_lib.load().hhx_synchronize())
        if s:
            wall.append((time.perf_counter() - t0) * 1e3)
        nnz = m.nnz
        ing.destroy()
        del m
    _lib.profile_enable(False)
    print(json.dumps({'probe': 'ingest', 'pairs': pairs, 'contigs': int(gen.n), 'nnz': int(nnz), 'wall_ms': wall, 'wall_ms_min': min(wall),
                      'kernel_ms': {k: round(_lib.profile_get(k)[0] / steps, 3) for k in NAMES if _lib.profile_get(k)[1]}}))


if __name__ == '__main__':
    main()
