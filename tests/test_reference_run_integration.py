"""The whole of the reference's own `haphic cluster` entry point, HapHiC_cluster.run(args), twice on the same FASTA +
.pairs input: once untouched (sparse mode with SURVEY's scipy stand-in for MKL), once with every seam re-bound by
haphic_amd.patch.patch_reference — and every file it writes compared (cluster and group files of every inflation, paired_links.clm,
alignments.bed byte for byte; full_links.pkl and HT_links.pkl as unpickled dicts, order and value types included).  This is the drop-in claim end to end: the
reference's CLI code drives the mirrors.  It needs the reference checkout, so it runs only in the dev container; there
is no GPU there, so tests/oracle_lib.py stands in for the HIP library (what is pinned here is the seam wiring and the
host mirrors inside run(); the kernels are pinned by the -m gpu tests against the same oracle)."""
import os
import sys
import types

import numpy as np
import pytest

REF = '/root/reference/scripts'


def _load_reference():
    for name, attrs in (('pysam', {'set_verbosity': lambda *a, **k: None, 'AlignmentFile': None}),
                        ('portion', {'closed': None, 'empty': None})):
        if name not in sys.modules:
            m = types.ModuleType(name)
            m.__dict__.update(attrs)
            sys.modules[name] = m
    sys.path.insert(0, REF)
    try:
        import HapHiC_cluster as H
    finally:
        sys.path.remove(REF)
    return H


def _write_inputs(d, split, messy=False, poly=False):
    from haphic_amd import synth
    rng = np.random.default_rng(17)
    gen = synth.make_genome(3, 1_200_000, 30_000 if not split else 150_000, cv=0.4, min_len=6000, seed=23)
    base_n = gen.n
    if poly:                                           # autotetraploid: four collinear haplotypes, 8 % allelic contacts
        gen = synth.make_polyploid(synth.make_genome(1, 900_000, 45_000, cv=0.3, min_len=8000, seed=23), 4)
        base_n = gen.n // 4
    with open(os.path.join(d, 'asm.fa'), 'w') as f:
        for nm, ln in zip(gen.names, gen.length.tolist()):
            seq = ''.join(rng.choice(list('ACGT'), ln))
            width = 80
            if messy:                                  # soft-masked runs, N gaps, GATC straddling line ends, header text
                a, b = sorted(rng.integers(0, ln, 2).tolist())
                seq = seq[:a] + seq[a:b].lower() + seq[b:]
                g0 = int(rng.integers(0, max(1, ln - 500)))
                seq = seq[:g0] + 'N' * 300 + seq[g0 + 300:]
                width = int(rng.choice([60, 61, 70, 80, 100]))
            f.write('>%s%s\n' % (nm, ' len=%d some description' % ln if messy else ''))
            for k in range(0, ln, width):
                f.write(seq[k:k + width] + '\n')
            if messy and rng.random() < 0.3:
                f.write('\n')
    id1, p1, id2, p2 = [t.numpy() for t in synth.sample_pairs(gen, 60_000, seed=29, cis=0.9)]
    if poly:
        id1, p1, id2, p2 = synth.add_allelic_pairs(gen, base_n, 4, id1, p1, id2, p2, 0.08, 31)
    with open(os.path.join(d, 'hic.pairs'), 'w') as f:
        f.write('## pairs format v1.0\n#columns: readID chr1 pos1 chr2 pos2 strand1 strand2\n')
        for k, (a, x, b, y) in enumerate(zip(id1.tolist(), p1.tolist(), id2.tolist(), p2.tolist())):
            na, nb = gen.names[a], gen.names[b]
            end = '\n'
            if messy:                                  # names that are not in the FASTA, CRLF ends, extra columns, blank / comment lines
                r = rng.random()
                if r < 0.01:
                    na = 'unplaced_scaffold_%d' % (k % 7)
                elif r < 0.02:
                    nb = 'chrUn'
                elif r < 0.05:
                    end = '\r\n'
                elif r < 0.06:
                    f.write('\n' if k % 2 else '#note\n')
            f.write('r%d\t%s\t%d\t%s\t%d\t+\t-%s%s' % (k, na, x + 1, nb, y + 1, '\tmapq=60' if messy and k % 11 == 0 else '', end))
    return gen


def _run(H, d, extra, pairs='../hic.pairs', nchrs=3):
    argv = sys.argv
    cwd = os.getcwd()
    os.makedirs(d, exist_ok=True)
    os.chdir(d)
    try:
        sys.argv = ['haphic', '../asm.fa', pairs, str(nchrs), '--min_inflation', '1.2', '--max_inflation', '2.4', '--inflation_step', '0.4',
                    '--Nx', '100', '--flank', '20'] + extra
        args = H.parse_arguments()
        H.run(args)
    finally:
        sys.argv = argv
        os.chdir(cwd)


def _tree(d):
    out = {}
    for root, _dirs, files in os.walk(d):
        for f in files:
            if f.endswith(('.log', '.pdf', '.png')):          # logs carry times, matplotlib stamps a creation date
                continue
            with open(os.path.join(root, f), 'rb') as fh:
                out[os.path.relpath(os.path.join(root, f), d)] = fh.read()
    return out


@pytest.mark.skipif(not os.path.isdir(REF), reason='reference checkout not present')
@pytest.mark.parametrize('split,extra', [(False, []), (True, ['--bin_size', '40']), (False, ['--remove_allelic_links', '2', '--max_read_pairs', '60']),
                                         (False, ['--normalize_by_nlinks']),
                                         (True, ['--bin_size', '40', '--remove_allelic_links', '2', '--remove_concentrated_links', '--max_read_pairs', '40']),
                                         (False, ['--remove_concentrated_links', '--max_read_pairs', '40']),
                                         (False, ['--RE', 'GATC,GANTC', '--gz']),
                                         (False, ['--dense_matrix']),
                                         (False, ['--expansion', '3']),
                                         (False, ['--messy']),
                                         (False, ['--poly', '--remove_allelic_links', '4', '--max_read_pairs', '40']),
                                         (False, ['--poly', '--remove_allelic_links', '4', '--remove_concentrated_links', '--normalize_by_nlinks']),
                                         (True, ['--messy', '--bin_size', '40', '--RE', 'GATC,GANTC']),
                                         (False, ['--max_iter', '3']),
                                         (True, ['--bin_size', '40', '--Nx', '60', '--flank', '0']),
                                         (False, ['--quick_view']),
                                         (False, ['--skip_clustering']),
                                         (True, ['--bin_size', '40', '--density_lower', '0.3X', '--rank_sum_hard_cutoff', '500', '--gz'])])
def test_reference_run_with_and_without_the_seams(tmp_path, monkeypatch, split, extra):
    import haphic_amd
    from haphic_amd import cluster, patch
    from tests import oracle_lib
    H = _load_reference()
    poly_case = '--poly' in extra
    _write_inputs(str(tmp_path), split, messy='--messy' in extra, poly=poly_case)
    extra = [e for e in extra if e not in ('--messy', '--poly')]
    pairs = '../hic.pairs'
    if '--gz' in extra:                                      # bgzipped_pairs input (:1544)
        import gzip
        extra = [e for e in extra if e != '--gz']
        with open(tmp_path / 'hic.pairs', 'rb') as fi, gzip.open(tmp_path / 'hic.pairs.gz', 'wb') as fo:
            fo.write(fi.read())
        pairs = '../hic.pairs.gz'
    # ---- the reference as it is (sparse mode: dot_product_mkl = scipy's float32 product)
    monkeypatch.setattr(H, 'dot_product_mkl', lambda a, b, **k: (a @ b).tocsc(), raising=False)
    monkeypatch.setattr(H, 'INTEL_MKL', True, raising=False)
    nchrs = 4 if poly_case else 3
    import logging
    seen = {'ref': [], 'ours': []}

    class Collect(logging.Handler):
        def __init__(self, sink):
            super().__init__(logging.DEBUG)
            self.sink = sink

        def emit(self, record):
            self.sink.append(record.getMessage())
    h_ref = Collect(seen['ref'])
    H.logger.addHandler(h_ref)
    try:
        _run(H, str(tmp_path / 'ref'), extra, pairs, nchrs)
    finally:
        H.logger.removeHandler(h_ref)
    # ---- the same entry point with the seams re-bound
    monkeypatch.setattr(haphic_amd, '_lib', oracle_lib)
    monkeypatch.setattr(cluster, '_lib', oracle_lib)
    monkeypatch.setattr(patch, '_lib', oracle_lib, raising=False)
    saved = patch.patch_reference(H)
    h_ours = Collect(seen['ours'])
    H.logger.addHandler(h_ours)
    # which of the array-backed S5 containers (haphic_amd/containers.py) had to become real dicts during run()
    from haphic_amd import containers
    thawed = []
    real_thaw = containers._Frozen._thaw
    monkeypatch.setattr(containers._Frozen, '_thaw', lambda self: (thawed.append(self._kind), real_thaw(self))[1])
    try:
        _run(H, str(tmp_path / 'ours'), extra, pairs, nchrs)
    finally:
        H.logger.removeHandler(h_ours)
        patch.unpatch_reference(H, saved)
    # run() itself never needs a Python object per key: with the default options the seams (output_pickle, output_clm, normalize_by_nlinks,
    # filter_fragments, dict_to_matrix) work on the arrays.  What thaws a table is the reference's own dict code: remove_allelic_HiC_links
    # :474-689 (full, flank, the coordinate lists) and the --remove_concentrated_links loop :2899-2902; output_statistics :2279 gets its
    # per-group link sums (parse_link_dict :2252) from the arrays too
    expected = set()
    if '--remove_allelic_links' in extra:
        expected |= {'full', 'flank', 'crd'}
    if '--remove_concentrated_links' in extra:
        expected |= {'full', 'crd'}
    assert set(thawed) == expected, (thawed, expected)
    # the log is an interface (users read it, HapHiC_pipeline.py:385 parses it): every message of the filtering and
    # clustering stages must come out of the mirrors word for word, in the same order (timings aside)
    def stable(msgs):
        return [m for m in msgs if m.startswith('[') or 'You could try inflation' in m or 'bin_size is' in m or 'The matrix' in m
                or 'Normalizing' in m or 'Reducing' in m or 'missing / redundant' in m]
    assert stable(seen['ours']) == stable(seen['ref'])
    want, got = _tree(str(tmp_path / 'ref')), _tree(str(tmp_path / 'ours'))
    assert sorted(want) == sorted(got), (sorted(want), sorted(got))
    assert 'HT_links.pkl' in want
    if '--quick_view' not in extra and '--skip_clustering' not in extra:
        assert any(k.endswith('.clusters.txt') for k in want) and 'full_links.pkl' in want and 'paired_links.clm' in want
    import pickle
    for k in want:
        if k.endswith('.pkl'):
            # same dict, same insertion order, same value types; the BYTES differ only through pickle's memo (the reference
            # keys hold a fresh str object per parsed line, the mirror re-uses one object per contig name)
            a, b = pickle.loads(want[k]), pickle.loads(got[k])
            assert type(a) is type(b) and list(a.items()) == list(b.items()), 'pickle differs: ' + k
            assert [type(v) for v in a.values()] == [type(v) for v in b.values()], 'value types differ: ' + k
        else:
            assert want[k] == got[k], 'file differs: ' + k


@pytest.mark.skipif(not os.path.isdir(REF), reason='reference checkout not present')
def test_dict_to_matrix_as_reassign_calls_it(monkeypatch):
    """HapHiC_reassign.py:538 — group names as keys, float link densities as values, dense result, a group without links"""
    from haphic_amd import cluster
    from tests import oracle_lib
    H = _load_reference()
    monkeypatch.setattr(cluster, '_lib', oracle_lib)
    rng = np.random.default_rng(4)
    groups = ['group%d' % k for k in range(1, 13)]
    d = {}
    for a in range(11):
        for b in range(a + 1, 11):
            if rng.random() < 0.6:
                d[(groups[a], groups[b])] = float(rng.random() * 3.7)
    for dense, loops in ((True, False), (False, True), (False, False)):
        want, widx = H.dict_to_matrix(dict(d), set(groups), dense_matrix=dense, add_self_loops=loops)
        got, gidx = cluster.dict_to_matrix(dict(d), set(groups), dense_matrix=dense, add_self_loops=loops)
        assert widx == gidx and list(widx) == list(gidx)
        if dense:
            assert got.dtype == want.dtype and np.array_equal(got, want)
        else:
            want.sort_indices()
            got.sort_indices()
            assert got.dtype == want.dtype and np.array_equal(got.indptr, want.indptr) and np.array_equal(got.indices, want.indices) \
                and np.array_equal(got.data, want.data)
