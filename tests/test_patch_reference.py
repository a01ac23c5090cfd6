"""CPU: the re-binding of the reference's seams (haphic_amd/patch.py).  Needs the reference checkout,
which only exists in the dev container — skipped elsewhere (nothing in -m gpu reads /root/reference)."""
import os
import sys
import types

import pytest

REF = '/root/reference/scripts'


@pytest.mark.skipif(not os.path.isdir(REF), reason='reference checkout not present')
def test_patch_binds_every_seam():
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
    from haphic_amd import cluster, patch
    import inspect
    # the mirror keeps the reference's signatures (same names, same positional order)
    for name in ('mcl', 'prune', 'interpret_result', 'dict_to_matrix', 'parse_alignments_for_ctgs', 'parse_alignments',
                 'run_mcl_clustering', 'mkl_matrix_power', 'parse_fasta', 'stat_fragments', 'count_RE_sites', 'filter_fragments',
                 'pairs_generator', 'pairs_generator_inter_ctgs', 'bam_generator', 'normalize_by_nlinks', 'normalize_by_length',
                 'reduce_inter_hap_HiC_links', 'check_param', 'cal_concordance_ratio', 'cal_concentration_adj_ratio', 'parse_RE_sites'):
        ref_params = list(inspect.signature(getattr(H, name)).parameters)
        our_params = [p for p in inspect.signature(getattr(cluster, name)).parameters if not p.startswith('_') and p != 'outdir_root']
        assert our_params[:len(ref_params)] == ref_params, (name, ref_params, our_params)
    saved = patch.patch_reference(H, ingest=True, matrix_build=True)
    try:
        assert H.mcl.__wrapped__ is cluster.mcl and H.run_mcl_clustering.__wrapped__ is cluster.run_mcl_clustering
        assert H.dot_product_mkl is cluster.dot_product_mkl and H.INTEL_MKL is True
        assert H.parse_alignments_for_ctgs is cluster.parse_alignments_for_ctgs
        assert H.bam_generator is cluster.bam_generator and H.normalize_by_nlinks is cluster.normalize_by_nlinks
    finally:
        patch.unpatch_reference(H, saved)
    assert H.mcl is not cluster.mcl and not hasattr(H.mcl, '__wrapped__')


def test_patch_reassign_binds_parse_link_dict():
    """f3: HapHiC_reassign.parse_link_dict is re-bound; float / normalised links go back to the reference function"""
    if not os.path.isdir(REF):
        pytest.skip('reference checkout not present')
    for name, attrs in (('pysam', {'set_verbosity': lambda *a, **k: None, 'AlignmentFile': None}), ('portion', {'closed': None, 'empty': None})):
        if name not in sys.modules:
            m = types.ModuleType(name)
            m.__dict__.update(attrs)
            sys.modules[name] = m
    sys.path.insert(0, REF)
    try:
        import HapHiC_reassign as R
    finally:
        sys.path.remove(REF)
    import inspect
    from haphic_amd import cluster, patch
    ref_params = list(inspect.signature(R.parse_link_dict).parameters)
    assert [p for p in inspect.signature(cluster.parse_link_dict).parameters if not p.startswith('_')] == ref_params
    saved = patch.patch_reassign(R)
    try:
        assert R.parse_link_dict.__wrapped__ is cluster.parse_link_dict
        d = {('a', 'b'): 0.5, ('a', 'c'): 1.5}
        grp = {'a': 'g1', 'b': 'g1', 'c': 'ungrouped'}
        got = R.parse_link_dict(dict(d), grp)                      # float links: the original function, through the seam
        want = saved['parse_link_dict'](dict(d), grp)
        assert {k: dict(v) for k, v in got[0].items()} == {k: dict(v) for k, v in want[0].items()} and got[1] == want[1]
    finally:
        R.parse_link_dict = saved['parse_link_dict']


@pytest.mark.skipif(not os.path.isdir(REF), reason='reference checkout not present')
def test_dict_to_matrix_folds_reversed_keys_like_the_reference(monkeypatch):
    """a link dict holding both (a, b) and (b, a) — never produced by the reference's parsers, but allowed by
    the public S4 seam — must give the matrix coo_matrix(...).tocsc() gives (duplicates summed, :368); a self key is
    refused.  Checked against the reference's own dict_to_matrix."""
    import numpy as np
    for name, attrs in (('pysam', {'set_verbosity': lambda *a, **k: None, 'AlignmentFile': None}), ('portion', {'closed': None, 'empty': None})):
        if name not in sys.modules:
            m = types.ModuleType(name)
            m.__dict__.update(attrs)
            sys.modules[name] = m
    sys.path.insert(0, REF)
    try:
        import HapHiC_cluster as H
    finally:
        sys.path.remove(REF)
    from haphic_amd import cluster
    from tests import oracle_lib
    monkeypatch.setattr(cluster, '_lib', oracle_lib)
    rng = np.random.default_rng(4)
    names = ['f%02d' % k for k in range(30)]
    d = {}
    for _ in range(200):
        a, b = rng.choice(30, 2, replace=False)
        d[(names[a], names[b])] = d.get((names[a], names[b]), 0) + int(rng.integers(1, 9))      # both orientations occur
    assert any((b, a) in d for (a, b) in d)
    frag_set = set(names[:27])
    want, want_idx = H.dict_to_matrix(dict(d), set(frag_set), dense_matrix=False, add_self_loops=True)
    got, got_idx = cluster.dict_to_matrix(dict(d), set(frag_set), dense_matrix=False, add_self_loops=True)
    assert got_idx == want_idx
    want = want.tocsc(); want.sum_duplicates(); want.sort_indices()
    got = got.tocsc(); got.sort_indices()
    assert np.array_equal(got.indptr, want.indptr) and np.array_equal(got.indices, want.indices) and np.array_equal(got.data, want.data)
    with pytest.raises(ValueError, match='links a fragment with itself'):
        cluster.dict_to_matrix({(names[0], names[0]): 3}, frag_set, dense_matrix=False)


def test_run_is_rebound_to_join_the_file_writers(monkeypatch, tmp_path):
    """round 6: output_pickle / output_clm / the .pairs front end only QUEUE their files (csrc/hhx_jobs.hip), so patch_reference re-binds run() :2738 itself:
    the writers are joined — and a writer's failure raised — before run() returns to main() :2967 / HapHiC_pipeline.py:358; a failure of run() itself
    is the one that is raised, after the queue has been drained.  (A stand-in module: the re-binding needs no reference checkout.)"""
    import numpy as np
    from haphic_amd import _lib, patch
    _lib.load()
    calls = []
    H = types.ModuleType('HapHiC_cluster_stand_in')

    def run(args, log_file=None):
        """the reference's run()"""
        calls.append(('run', args, log_file))
        if args == 'boom':
            raise ValueError('the run failed')
        # what a seam of the real run() does: queue a file on the library's writer lanes
        i = np.arange(5, dtype=np.int32)
        _lib.write_link_pickle_async(str(tmp_path / 'x.pkl') if args != 'full' else '/dev/full', i, i[::-1].copy(), np.arange(5), ['c%d' % k for k in range(5)])
        return 'done'
    H.run = run
    saved = patch.patch_reference(H, ingest=True, matrix_build=True)
    try:
        assert H.run.__wrapped__ is run and H.run.__doc__ == run.__doc__ and saved['run'] is run
        assert H.run('args', log_file='l') == 'done' and calls[-1] == ('run', 'args', 'l')
        assert _lib.files_pending()[0] == 0 and (tmp_path / 'x.pkl').stat().st_size > 0       # complete when run() returns
        with pytest.raises(RuntimeError, match='/dev/full'):                                   # a writer's failure surfaces from run()
            H.run('full')
        with pytest.raises(ValueError, match='the run failed'):                                # run()'s own failure wins; nothing is left queued
            H.run('boom')
        assert _lib.files_pending()[0] == 0
    finally:
        patch.unpatch_reference(H, saved)
    assert H.run is run
