"""f4, second half: the `haphic plot` contact-map binning (haphic_amd/csrc/hhx_contacts.hip behind haphic_amd/plot.py)
against oracle/plot_oracle.py, itself pinned to contact matrices the reference's own parse_agp / generate_contact_matrix /
parse_pairs produced (tests/golden/plot.npz, make_golden.py gen_plot)."""
import os

import numpy as np
import pytest

from oracle import plot_oracle as po
from tests import bam_fixture as bf
from tests import plot_fixture as pf
from tests.conftest import load_golden


@pytest.fixture(scope='module')
def golden_plot():
    return load_golden('plot.npz')


def _text(g, key):
    return g[key].tobytes().decode()


def _containers(g, name):
    bin_size, min_len_bp = (int(x) for x in g[name + '__params'])
    cd, cad, sizes, frags, gfd = po.parse_agp(_text(g, name + '__agp'), bin_size)
    mat, to_total, group_list, ctg_set = po.generate_contact_matrix(sizes, frags, gfd, bin_size, min_len_bp / 1000000, _text(g, name + '__specified') or None)
    return bin_size, cd, cad, mat, to_total, group_list, ctg_set


@pytest.mark.parametrize('name', list(pf.CASES))
def test_oracle_against_reference_matrices(golden_plot, name):
    g = golden_plot
    bin_size, cd, cad, mat, to_total, group_list, ctg_set = _containers(g, name)
    assert ','.join(group_list) == _text(g, name + '__groups') and ','.join(sorted(ctg_set)) == _text(g, name + '__ctg_set')
    assert mat.shape == g[name + '__matrix'].shape
    error = _text(g, name + '__error')
    if error:
        with pytest.raises(Exception) as e:
            po.bin_pairs(po.pairs_records(_text(g, name + '__pairs')), cd, cad, bin_size, mat, to_total, group_list, ctg_set)
        assert error.startswith(str(e.value))            # the reference appends its hint about the file type
        return
    got = po.bin_pairs(po.pairs_records(_text(g, name + '__pairs')), cd, cad, bin_size, mat, to_total, group_list, ctg_set)
    assert np.array_equal(got, g[name + '__matrix']) and got.sum() > 0


def _flat_inputs(table, records, pos_offset=0):
    cid = {n: i for i, n in enumerate(table.names)}
    id1 = np.array([cid.get(r[0], -1) for r in records], np.int32)
    id2 = np.array([cid.get(r[2], -1) for r in records], np.int32)
    p1 = np.array([r[1] - pos_offset for r in records], np.int32)
    p2 = np.array([r[3] - pos_offset for r in records], np.int32)
    return id1, p1, id2, p2


@pytest.mark.parametrize('name', list(pf.CASES))
def test_flattened_tables_keep_the_lookup(golden_plot, name):
    """host logic of the product (ContactTable: reference dicts -> the arrays of hhx_contact_map_create) through the numpy
    restatement of the kernel's loop: same matrix as the dict walk, same first offending pair"""
    from haphic_amd.plot import ContactTable
    g = golden_plot
    bin_size, cd, cad, mat, to_total, group_list, ctg_set = _containers(g, name)
    table = ContactTable(cd, cad, bin_size, to_total, group_list, ctg_set, mat.shape[0])
    records = list(po.pairs_records(_text(g, name + '__pairs')))
    got, bad = po.bin_flat(*table.arrays, bin_size, mat.shape[0], *_flat_inputs(table, records))
    if _text(g, name + '__error'):
        assert bad >= 0
        r = records[bad >> 1]
        assert 'Cannot find alignment position: {}:{} in'.format(*(r[2:] if bad & 1 else r[:2])) in _text(g, name + '__error')
    else:
        assert bad == -1 and np.array_equal(got, g[name + '__matrix'])


def test_abi_declares_contact_map():
    from haphic_amd import _lib
    lib = _lib.load()
    for sym in ('hhx_contact_map_create', 'hhx_contact_map_push', 'hhx_contact_map_fetch', 'hhx_contact_map_device', 'hhx_contact_map_destroy'):
        assert hasattr(lib, sym)


@pytest.mark.gpu
@pytest.mark.parametrize('name', list(pf.CASES))
def test_parse_pairs_mirror_on_reference_cases(golden_plot, name, tmp_path):
    from haphic_amd import plot
    g = golden_plot
    bin_size, cd, cad, mat, to_total, group_list, ctg_set = _containers(g, name)
    error = _text(g, name + '__error')
    for suffix in ('.pairs', '.pairs.gz'):
        path = tmp_path / ('hic' + suffix)
        if suffix == '.pairs':
            path.write_bytes(g[name + '__pairs'].tobytes())
        else:
            import gzip
            path.write_bytes(gzip.compress(g[name + '__pairs'].tobytes()))
        fresh = np.zeros_like(mat)
        if error:
            with pytest.raises(Exception) as e:
                plot.parse_pairs(str(path), cd, cad, bin_size, fresh, to_total, group_list, ctg_set)
            assert str(e.value) == error
        else:
            got = plot.parse_pairs(str(path), cd, cad, bin_size, fresh, to_total, group_list, ctg_set)
            assert got is fresh and got.dtype == mat.dtype and np.array_equal(got, g[name + '__matrix'])


@pytest.mark.gpu
def test_parse_bam_mirror(tmp_path):
    """parse_bam :205-245: read1 records only, reference_start + 1, unmapped mates and references outside the AGP skipped"""
    from haphic_amd import plot
    case = pf.make_case(seed=11, n_pairs=20000, broken=4, partial=3)
    bin_size = case['bin_size']
    cd, cad, sizes, frags, gfd = po.parse_agp(case['agp'], bin_size)
    mat, to_total, group_list, ctg_set = po.generate_contact_matrix(sizes, frags, gfd, bin_size, 0, None)
    rng = np.random.default_rng(3)
    refs = [(n, l) for n, l in case['contigs'].items()] + [('ghost_a', 50000), ('ghost_b', 50000)]
    rng.shuffle(refs)                                     # BAM reference order != AGP order
    rid = {n: k for k, (n, _) in enumerate(refs)}
    recs, want_records = [], []
    for r, p, m, q in case['records']:
        flag = 0x41 if rng.random() < 0.5 else 0x81       # read1 / read2
        unmapped_mate = rng.random() < 0.05
        recs.append((rid[r], p - 1, -1 if unmapped_mate else rid[m], -1 if unmapped_mate else q - 1, flag))
        if flag & 0x40 and not unmapped_mate:
            want_records.append((r, p, m, q))
    path = tmp_path / 'hic.bam'
    path.write_bytes(bf.bam_bytes(refs, recs, block_payload=4000))
    want = po.bin_pairs(want_records, cd, cad, bin_size, mat.copy(), to_total, group_list, ctg_set)
    got = plot.parse_bam(str(path), cd, cad, bin_size, mat.copy(), to_total, group_list, ctg_set, 2)
    assert np.array_equal(got, want) and got.sum() > 5000
    # accumulation into a matrix that already holds counts (the reference adds in place)
    twice = plot.parse_bam(str(path), cd, cad, bin_size, got.copy(), to_total, group_list, ctg_set, 2)
    assert np.array_equal(twice, 2 * want)
    # a position outside the AGP: the reference's message, with the 1-based position
    bad_case = pf.make_case(seed=12, n_pairs=2000, partial=5, out_of_agp=True)
    cd, cad, sizes, frags, gfd = po.parse_agp(bad_case['agp'], bin_size)
    mat, to_total, group_list, ctg_set = po.generate_contact_matrix(sizes, frags, gfd, bin_size, 0, None)
    refs = [(n, l) for n, l in bad_case['contigs'].items()] + [('ghost_a', 50000), ('ghost_b', 50000)]
    rid = {n: k for k, (n, _) in enumerate(refs)}
    path.write_bytes(bf.bam_bytes(refs, [(rid[r], p - 1, rid[m], q - 1, 0x41) for r, p, m, q in bad_case['records']]))
    with pytest.raises(Exception) as want_error:
        po.bin_pairs(bad_case['records'], cd, cad, bin_size, mat.copy(), to_total, group_list, ctg_set)
    with pytest.raises(Exception) as e:
        plot.parse_bam(str(path), cd, cad, bin_size, mat.copy(), to_total, group_list, ctg_set, 1)
    assert str(e.value) == str(want_error.value) + '. Please check whether the input AGP and BAM files match'


@pytest.mark.gpu
def test_contact_map_kernel_large_random():
    """2 M read pairs with the Hi-C shape (most pairs close to the diagonal: the LDS pre-aggregation path) and ids / positions
    of every kind, against the numpy restatement over the same flattened tables; several pushes accumulate"""
    from haphic_amd.plot import ContactTable
    case = pf.make_case(seed=21, n_scaffolds=12, ctgs_per=40, bin_size=20000, n_pairs=10, broken=30, partial=10, short_scaffolds=5)
    bin_size = case['bin_size']
    cd, cad, sizes, frags, gfd = po.parse_agp(case['agp'], bin_size)
    mat, to_total, group_list, ctg_set = po.generate_contact_matrix(sizes, frags, gfd, bin_size, 0.03, None)
    table = ContactTable(cd, cad, bin_size, to_total, group_list, ctg_set, mat.shape[0])
    rng = np.random.default_rng(5)
    n, n_ctg = 2_000_000, len(table.names)
    lens = np.array([case['contigs'][c] for c in table.names], np.int64)
    id1 = rng.integers(-2, n_ctg, n).astype(np.int32)
    id2 = np.where(rng.random(n) < 0.7, id1, rng.integers(-1, n_ctg + 1, n)).astype(np.int32)        # n_ctg: out of range, skipped
    pos1 = (rng.random(n) * lens[np.clip(id1, 0, n_ctg - 1)]).astype(np.int32) + 1
    pos2 = np.where(id1 == id2, np.clip(pos1 + rng.integers(-3000, 3000, n), 1, None),
                    (rng.random(n) * lens[np.clip(id2, 0, n_ctg - 1)]).astype(np.int64) + 1).astype(np.int32)
    # keep partially placed contigs inside their alignment bins here (the error path has its own check below)
    last = np.array([(len(cad[c]) and (max(cad[c]) + 1) * bin_size) or 1 for c in table.names], np.int64)
    pos1 = np.minimum(pos1, last[np.clip(id1, 0, n_ctg - 1)]).astype(np.int32)
    pos2 = np.minimum(pos2, last[np.clip(id2, 0, n_ctg - 1)]).astype(np.int32)
    want, bad = po.bin_flat(*table.arrays, bin_size, mat.shape[0], id1, pos1, id2, pos2)
    assert bad == -1 and want.sum() > n // 3 and np.trace(want) > want.sum() // 4
    cm = table.device()
    try:
        cut = [0, 700_001, 700_002, 1_500_000, n]
        for a, b in zip(cut, cut[1:]):
            assert cm.push(id1[a:b], pos1[a:b], id2[a:b], pos2[a:b]) == -1
        assert np.array_equal(cm.fetch(), want)
        assert cm.push(id1[:0], pos1[:0], id2[:0], pos2[:0]) == -1             # empty batch
        # 0-based input + pos_offset 1 (the BAM convention)
        cm2 = table.device()
        assert cm2.push(id1, pos1 - 1, id2, pos2 - 1, pos_offset=1) == -1 and np.array_equal(cm2.fetch(), want)
        cm2.destroy()
        # first offending pair: position 0, a position behind the last alignment bin (first end, then mate only)
        k = next(int(c) for c in np.flatnonzero((id1 >= 0) & (id2 >= 0) & (id2 < n_ctg))[1000:]
                 if po.bin_flat(*table.arrays, bin_size, mat.shape[0], id1[c:c + 1], pos1[c:c + 1], id2[c:c + 1], pos2[c:c + 1])[0].sum() == 1)
        for side, value in ((0, 0), (0, int(last[id1[k]]) + 1), (1, int(last[id2[k]]) + 1)):
            p1, p2 = pos1.copy(), pos2.copy()
            (p2 if side else p1)[k] = value
            want_bad = po.bin_flat(*table.arrays, bin_size, mat.shape[0], id1, p1, id2, p2)[1]
            assert cm.push(id1, p1, id2, p2) == want_bad and want_bad >= 0
    finally:
        cm.destroy()


@pytest.mark.skipif(not os.path.isdir('/root/reference/scripts'), reason='needs the reference checkout (dev container only)')
def test_patch_plot_rebinds_the_reference_seams():
    """signatures of the mirrors == the reference's (HapHiC_plot.py :153 :205); nothing under /root/reference is edited"""
    import inspect
    import sys
    import types
    for name, attrs in (('pysam', {'set_verbosity': lambda *a, **k: None, 'AlignmentFile': None}), ('portion', {'closed': po.Closed})):
        if name not in sys.modules:
            m = types.ModuleType(name)
            m.__dict__.update(attrs)
            sys.modules[name] = m
    if getattr(sys.modules['portion'], 'closed', None) is None:
        sys.modules['portion'].closed = po.Closed
    sys.path.insert(0, '/root/reference/scripts')
    try:
        import HapHiC_plot as P
    finally:
        sys.path.remove('/root/reference/scripts')
    from haphic_amd import plot
    # the reference's own containers (parse_agp :41-103, generate_contact_matrix :106-150) == the oracle restatement's,
    # contents and dict orders, on every fixture case: what plot.ContactTable flattens is what the reference would hand it
    import tempfile
    P.logger.setLevel('CRITICAL')
    for name, (kw, min_len, specified) in pf.CASES.items():
        case = pf.make_case(**kw)
        with tempfile.NamedTemporaryFile('w', suffix='.agp') as f:
            f.write(case['agp'])
            f.flush()
            r_cd, r_cad, r_sizes, r_frags, r_gfd = P.parse_agp(f.name, case['bin_size'])
        o_cd, o_cad, o_sizes, o_frags, o_gfd = po.parse_agp(case['agp'], case['bin_size'])
        assert list(r_cd) == list(o_cd) and all(list(r_cd[c].items()) == list(o_cd[c].items()) for c in r_cd), name
        assert list(r_cad) == list(o_cad) and all(list(r_cad[c].items()) == list(o_cad[c].items()) for c in r_cad), name
        assert list(r_sizes.items()) == list(o_sizes.items()) and r_frags == o_frags and dict(r_gfd) == dict(o_gfd), name
        r_out = P.generate_contact_matrix(r_sizes, r_frags, r_gfd, case['bin_size'], min_len, specified)
        o_out = po.generate_contact_matrix(o_sizes, o_frags, o_gfd, case['bin_size'], min_len, specified)
        assert r_out[0].shape == o_out[0].shape and r_out[0].dtype == o_out[0].dtype
        assert list(r_out[1].items()) == list(o_out[1].items()) and r_out[2] == o_out[2] and r_out[3] == o_out[3], name
        t_ref = plot.ContactTable(r_cd, r_cad, case['bin_size'], r_out[1], r_out[2], r_out[3], r_out[0].shape[0])
        t_orc = plot.ContactTable(o_cd, o_cad, case['bin_size'], o_out[1], o_out[2], o_out[3], o_out[0].shape[0])
        assert t_ref.names == t_orc.names and all(np.array_equal(a, b) for a, b in zip(t_ref.arrays, t_orc.arrays)), name
    want = {n: list(inspect.signature(getattr(P, n)).parameters) for n in ('parse_pairs', 'parse_bam')}
    saved = plot.patch_plot(P)
    try:
        assert P.parse_pairs is plot.parse_pairs and P.parse_bam is plot.parse_bam
        for n in want:
            assert list(inspect.signature(getattr(P, n)).parameters) == want[n]
    finally:
        P.parse_pairs, P.parse_bam = saved['parse_pairs'], saved['parse_bam']
