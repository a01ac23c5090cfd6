"""End-to-end a1 + a2 from a .pairs TEXT FILE: file -> chunks -> hhx_pairs_parse -> hhx_ingest_push(device) ->
hhx_ingest_link_matrix (the CPU baseline of the tokeniser is in bench.py's cpu_baseline).
usage: python tools/text_e2e.py [lines] [contigs]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import torch
    from haphic_amd import _lib, cluster, synth
    lines = int(sys.argv[1]) if len(sys.argv) > 1 else 8_000_000
    contigs = int(sys.argv[2]) if len(sys.argv) > 2 else 10000
    gen = synth.make_genome(16, contigs // 16 * 50_000, 50_000, seed=12345)
    names = list(gen.names)
    id1, p1, id2, p2 = [a.tolist() for a in synth.sample_pairs(gen, lines, seed=7)]
    path = '/tmp/e2e.pairs'
    t0 = time.perf_counter()
    with open(path, 'w') as f:
        f.write('## pairs format v1.0\n')
        step = 1 << 20
        for s in range(0, lines, step):
            f.write(''.join('r%d\t%s\t%d\t%s\t%d\t+\t-\n' % (s + i, names[a], x + 1, names[b], y + 1) for i, (a, x, b, y) in
                            enumerate(zip(id1[s:s + step], p1[s:s + step], id2[s:s + step], p2[s:s + step]))))
    size = os.path.getsize(path)
    print('wrote %d lines, %.2f GB in %.1f s' % (lines, size / 1e9, time.perf_counter() - t0))
    table = cluster.FragTable.for_contigs(gen.lexical_rank(), gen.length, np.ones(gen.n, np.uint8), names=names)
    os.chdir('/tmp')
    for want_bed in (True, False):
        for rep in range(2):
            t0 = time.perf_counter()
            aln = cluster.pairs_generator_inter_ctgs(path, 'pairs')
            if not want_bed:
                aln.bed_path = None
            ing = _lib.Ingest(table, 500_000, bins=False, skip_intra=True)
            t_parse = 0.0
            for parser, k in aln.batches(names):
                ing.push_device(k, *parser.device_arrays()[:4])
            ing.finalize()
            m, fidx, n_linked = ing.link_matrix(np.ones(gen.n, np.uint8))
            _lib.check(_lib.load().hhx_synchronize())
            dt = time.perf_counter() - t0
            nnz = m.nnz
            m.free()
            ing.destroy()
        print('device path, bed=%s: %.3f s  %.1f M pairs/s  %.2f GB/s of text  (link matrix nnz %d)' % (want_bed, dt, lines / dt / 1e6, size / dt / 1e9, nnz))


if __name__ == '__main__':
    main()
