"""The reference's OWN Python on the host CPU (BASELINE.md §4 / SURVEY §8d "Timing the reference CPU path"): run in the
dev container only (needs /root/reference; pysam / portion are stubbed as in SURVEY App. B, MKL is absent so
dot_product_mkl is scipy's float32 product).  Times
    ingest : parse_alignments_for_ctgs(pairs_generator_inter_ctgs(file, 'pairs'), ...)  — single-threaded by construction
    mcl    : run_mcl_clustering's normalise + pre-expansion, then mcl(M2, 2, 2.0, 200, 1e-4, dense_matrix=False)
on a synthetic assembly of the BASELINE configs[2] contig model (mean 30 kb) small enough to finish in minutes, and
writes one JSON object (committed as profiles/r02_reference_python_baseline.json; bench.py carries it in
cpu_baseline.reference_python)."""
import json
import os
import sys
import tempfile
import time
import types

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
for name, attrs in (('pysam', {'set_verbosity': lambda *a, **k: None, 'AlignmentFile': None}), ('portion', {'closed': None, 'empty': None})):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
sys.path.insert(0, '/root/reference/scripts')
import HapHiC_cluster as H  # noqa: E402

H.dot_product_mkl = lambda a, b, **k: (a @ b).tocsc()
H.INTEL_MKL = True
H.logger.setLevel('WARNING')
from haphic_amd import synth  # noqa: E402


class Args:
    flank = 500
    remove_allelic_links = 0
    remove_concentrated_links = False
    max_read_pairs = 200
    nwindows = 50


def main():
    contigs, pairs = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (4000, 3_000_000)
    nchrs = 8
    gen = synth.make_genome(nchrs, max(1, contigs // nchrs) * 30_000, 30_000, seed=12345)
    id1, p1, id2, p2 = [a.numpy() for a in synth.sample_pairs(gen, pairs, seed=12345)]
    names = list(gen.names)
    fa_dict = {n: [None, int(l), int(r)] for n, l, r in zip(names, gen.length, gen.re_sites)}
    out = {'host': os.uname().nodename, 'host_cpus': os.cpu_count(), 'cores_used': 1, 'contigs': gen.n, 'pairs': int(pairs),
           'python': sys.version.split()[0], 'note': 'reference = /root/reference/scripts/HapHiC_cluster.py, unmodified; MKL absent: '
           'dot_product_mkl = scipy float32 `@` (SURVEY §8c)'}
    with tempfile.TemporaryDirectory() as td:
        path = os.path.join(td, 'hic.pairs')
        with open(path, 'w') as f:
            f.write('## pairs format v1.0\n')
            for k, (a, x, b, y) in enumerate(zip(id1.tolist(), p1.tolist(), id2.tolist(), p2.tolist())):
                f.write('r%d\t%s\t%d\t%s\t%d\t+\t-\n' % (k, names[a], x + 1, names[b], y + 1))
        cwd = os.getcwd()
        os.chdir(td)
        try:
            ctg_len = {n: fa_dict[n][1] for n in names}
            t0 = time.perf_counter()
            full, flank, HT, clm, frag_link, coord = H.parse_alignments_for_ctgs(
                H.pairs_generator_inter_ctgs(path, 'pairs'), fa_dict, Args(), ctg_len, set(names), 'int32', 'int32')
            dt = time.perf_counter() - t0
        finally:
            os.chdir(cwd)
    out['ingest'] = {'seconds': dt, 'pairs_per_s': pairs / dt, 'what': 'parse_alignments_for_ctgs over pairs_generator_inter_ctgs (.pairs text, '
                     'alignments.bed written inside the loop as the reference does)', 'full_keys': len(full), 'flank_keys': len(flank)}
    t0 = time.perf_counter()
    mat, fidx = H.dict_to_matrix(flank, set(names), dense_matrix=False, add_self_loops=True)
    out['dict_to_matrix_seconds'] = time.perf_counter() - t0
    t0 = time.perf_counter()
    norm = H.normalize(mat, norm='l1', axis=0)
    pre = H.mkl_matrix_power(norm, 2)
    out['pre_expansion_seconds'] = time.perf_counter() - t0
    iters = []
    orig_prune = H.prune

    def counting_prune(*a, **k):
        iters.append(time.perf_counter())
        return orig_prune(*a, **k)
    H.prune = counting_prune
    t0 = time.perf_counter()
    res = H.mcl(pre, 2, 2.0, 200, 1e-4, dense_matrix=False)
    dm = time.perf_counter() - t0
    H.prune = orig_prune
    out['mcl'] = {'seconds': dm, 'iterations': len(iters), 'iters_per_s': len(iters) / dm, 'n': int(mat.shape[0]), 'link_matrix_nnz': int(mat.nnz),
                  'pre_expanded_nnz': int(pre.nnz), 'clusters': len(H.interpret_result(res, dense_matrix=False) or []),
                  'what': 'mcl(M2, 2, 2.0, 200, 1e-4, dense_matrix=False) on the pre-expanded matrix, scipy SpGEMM stand-in'}
    print(json.dumps(out))


if __name__ == '__main__':
    main()
