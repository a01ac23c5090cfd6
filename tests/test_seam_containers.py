"""The seam sequence of the reference's run() :2869-2935 on array-backed containers (haphic_amd/containers.py):
parse_alignments* -> output_pickle(HT) -> output_clm -> [normalize_by_nlinks] -> filter_fragments -> output_pickle(full) ->
dict_to_matrix -> run_mcl_clustering.  Checked here: (1) while nothing but the seams touches them the containers stay frozen and
every seam gives what it gives on the real dicts; (2) any other access (lookups, iteration, mutation, deletion — what
remove_allelic_HiC_links :474-689 does) thaws them into dicts equal to the reference's, after which the generic paths take over;
(3) pickles are plain `defaultdict(int)` pickles; (4) the library's host-side pickle writer against pickle itself.
The bodies run on CPU with tests/oracle_lib.py standing in for the HIP library and, marked gpu, on the library itself."""
import os
import pickle
import types
from collections import defaultdict

import numpy as np
import pytest

from haphic_amd import cluster, containers, synth
from oracle import oracle as orc
from tests import oracle_lib


def _case(n_pairs=40_000, seed=5, split=False):
    gen = synth.make_genome(3, 900_000, 30_000 if not split else 140_000, cv=0.4, min_len=6000, seed=seed)
    id1, p1, id2, p2 = [t.numpy() for t in synth.sample_pairs(gen, n_pairs, seed=seed + 1, cis=0.9)]
    if not split:                                        # what pairs_generator_inter_ctgs :1562-1583 hands to parse_alignments_for_ctgs
        keep = id1 != id2
        id1, p1, id2, p2 = id1[keep], p1[keep], id2[keep], p2[keep]
    fa_dict = {nm: [None, int(ln), int(ln) // 256 + 1] for nm, ln in zip(gen.names, gen.length.tolist())}
    return gen, fa_dict, cluster.IdArrays(gen.names, id1, p1, id2, p2)


def _args(**kw):
    a = dict(flank=20, remove_allelic_links=0, remove_concentrated_links=False, max_read_pairs=200, nwindows=50)
    a.update(kw)
    return types.SimpleNamespace(**a)


def _s5(fa_dict, aln, args, split=False):
    if split:
        bin_size, frag_len_dict, split_set = 40_000, {}, set()              # the bin table of stat_fragments :225-247
        for c, (_seq, ln, _re) in fa_dict.items():
            if ln > bin_size:
                split_set.add(c)
                nb = -(-ln // bin_size)
                for k in range(nb):
                    frag_len_dict['{}_bin{}'.format(c, k + 1)] = bin_size if k < nb - 1 else ln - bin_size * (nb - 1)
            else:
                frag_len_dict[c] = ln
        nx = set(frag_len_dict)
        return cluster.parse_alignments(aln, fa_dict, args, bin_size, frag_len_dict, nx, split_set, 'int32', 'int32') + (frag_len_dict, nx)
    ctg_len = {c: v[1] for c, v in fa_dict.items()}
    return cluster.parse_alignments_for_ctgs(aln, fa_dict, args, ctg_len, set(fa_dict), 'int32', 'int32') + (ctg_len, set(fa_dict))


def _same_matrix(a, b):
    a, b = a.to_arrays(), b.to_arrays()
    return all(np.array_equal(x, y) for x, y in zip(a, b))


def body_frozen_seams_equal_dict_seams(tmp_path, monkeypatch, normalize):
    """every seam of run() on the frozen containers == the same seam on the thawed copies"""
    monkeypatch.chdir(tmp_path)
    gen, fa_dict, aln = _case()
    args = _args()
    full, flank, HT, clm, frag_link, coord, ctg_len, nx = _s5(fa_dict, aln, args)
    assert all(c.frozen for c in (full, flank, HT, clm)) and not getattr(coord, 'frozen', False) and len(coord) == 0
    # the same containers as real dicts, from a second run that is thawed at once
    full2, flank2, HT2, clm2, frag_link2, _coord2, _, _ = _s5(fa_dict, aln, args)
    for c in (full2, flank2, HT2, clm2):
        c._thaw()
        assert type(c) is containers.Thawed and not c.frozen
    assert dict(frag_link) == dict(frag_link2)
    assert len(full) == len(full2) and len(flank) == len(flank2) and len(HT) == len(HT2) and len(clm) == len(clm2) and bool(full)
    assert all(c.frozen for c in (full, flank, HT, clm)), 'len() must not thaw'
    # output_pickle: HT, full
    for name, a, b in (('HT', HT, HT2), ('full', full, full2)):
        cluster.output_pickle(a, name, name + '_a.pkl')
        cluster.output_pickle(b, name, name + '_b.pkl')
        cluster._lib.files_join()                                   # a frozen table's file is queued on the library's writer thread
        da, db = pickle.load(open(name + '_a.pkl', 'rb')), pickle.load(open(name + '_b.pkl', 'rb'))
        assert type(da) is defaultdict and type(db) is defaultdict and da.default_factory is int
        assert list(da.items()) == list(db.items()) and all(type(v) is int for v in da.values())
        assert pickle.loads(pickle.dumps(a)) == db and type(pickle.loads(pickle.dumps(a))) is defaultdict      # plain pickle of a frozen table
    # output_clm
    cluster.output_clm(clm)
    cluster._lib.files_join()
    os.rename('paired_links.clm', 'a.clm')
    cluster.output_clm(clm2)
    assert open('a.clm', 'rb').read() == open('paired_links.clm', 'rb').read() and os.path.getsize('a.clm') > 1000
    if normalize:
        cluster.normalize_by_nlinks(flank, frag_link)
        cluster.normalize_by_nlinks(flank2, frag_link2)
    assert all(c.frozen for c in (full, flank, HT, clm))
    # filter_fragments + dict_to_matrix
    re_dict = {c: v[2] for c, v in fa_dict.items()}
    fargs = (nx, re_dict, 5, None, '0.2X', '1.9X', 10, '1.5X', 0, None, {}, '1.5X', set())

    def filt(fl, fr):
        a = list(fargs)
        a[3], a[9] = fr, fl
        return cluster.filter_fragments(*a)
    kept, kept2 = filt(flank, frag_link), filt(flank2, frag_link2)
    assert kept == kept2 and 5 < len(kept) < len(nx)
    m, idx = cluster.dict_to_matrix(flank, kept, dense_matrix=False, add_self_loops=True)
    m2, idx2 = cluster.dict_to_matrix(flank2, kept2, dense_matrix=False, add_self_loops=True, _device=True)
    assert isinstance(m, cluster.ResidentMatrix) and flank.frozen
    assert idx == idx2 and list(idx) == list(idx2)
    csc = m2.to_scipy_csc()
    assert m.shape == csc.shape and m.nnz == csc.nnz                        # answered by the scipy matrix the reference would have got
    dev = m.take_device()
    assert _same_matrix(dev, m2)
    if normalize:
        vals = np.array(list(flank2.values()))
        assert vals.dtype == np.float64 and np.array_equal(flank.arrays()[2], vals)
    dev.free()
    m2.free()
    # the dense form (filters :603, reassign)
    d1, i1 = cluster.dict_to_matrix(flank, kept, dense_matrix=True, add_self_loops=False)
    d2, i2 = cluster.dict_to_matrix(flank2, kept, dense_matrix=True, add_self_loops=False)
    assert isinstance(d1, np.ndarray) and np.array_equal(d1, d2) and i1 == i2
    # a fragment set with a name the table does not know takes the generic path (and gives the reference's numbering)
    odd = set(list(kept)[:50]) | {'not_a_contig'}
    m3, i3 = cluster.dict_to_matrix(flank, odd, dense_matrix=False, add_self_loops=True, _device=True)
    m4, i4 = cluster.dict_to_matrix(flank2, odd, dense_matrix=False, add_self_loops=True, _device=True)
    assert i3 == i4 and list(i3) == list(i4) and 'not_a_contig' in i3 and _same_matrix(m3, m4)
    m3.free()
    m4.free()
    assert flank.frozen and full.frozen
    # parse_link_dict :2252-2268 (output_statistics): per-group link sums from the arrays == the reference's loop on the dict,
    # outer and inner insertion orders included (ties at :2362 are broken by them); some contigs 'ungrouped'
    names = list(fa_dict)
    rng = np.random.default_rng(2)
    ctg_group = {c: ('ungrouped' if rng.random() < 0.2 else int(rng.integers(0, 7))) for c in names}
    got = cluster.group_link_dict(full, ctg_group)
    want = defaultdict(dict)
    for (ci, cj), links in full2.items():                    # the loop of :2263-2266
        for ctg, group in ((ci, ctg_group[cj]), (cj, ctg_group[ci])):
            if group != 'ungrouped':
                want[ctg][group] = want[ctg].get(group, 0) + links
    assert full.frozen and type(got) is defaultdict and got == want and list(got) == list(want)
    assert all(list(got[c].items()) == list(want[c].items()) and all(type(x) is int for x in got[c].values()) for c in want)
    assert cluster.group_link_dict(full2, ctg_group) == want


def body_mutation_thaws_and_falls_back(tmp_path, monkeypatch):
    """what remove_allelic_HiC_links / the --remove_concentrated_links loop do to the dicts: lookups, deletions, scaling"""
    monkeypatch.chdir(tmp_path)
    gen, fa_dict, aln = _case(seed=9)
    args = _args(remove_allelic_links=2, remove_concentrated_links=True, max_read_pairs=30)
    full, flank, HT, clm, frag_link, coord, ctg_len, nx = _s5(fa_dict, aln, args)
    assert coord.frozen and full.frozen
    ref = _s5(fa_dict, aln, args)
    for c in ref[:4] + (ref[5],):
        c._thaw()
    full2, flank2, HT2, clm2, _fl2, coord2 = ref[:6]
    # reads that thaw
    some = next(iter(dict.keys(full2)))
    assert full[some] == full2[some] and not full.frozen and type(full) is containers.Thawed
    assert full == full2 and list(full.items()) == list(full2.items())
    assert ('nope', 'nope') not in flank and not flank.frozen and flank == flank2
    assert coord == coord2 and list(coord) == list(coord2) and not coord.frozen
    assert any(isinstance(v, list) for v in coord.values()), 'no contig pair reached max_read_pairs: test too small'
    assert sum(1 for _ in clm.items()) == len(clm2) and not clm.frozen and clm == clm2
    assert HT.frozen
    # mutations on a table that is still frozen
    full3, flank3 = _s5(fa_dict, aln, args)[:2]
    keys = list(dict.keys(flank2))
    for k in keys[::7]:
        del flank3[k]
        del flank2[k]
    flank3[keys[1]] *= 0.5
    flank2[keys[1]] *= 0.5
    assert not flank3.frozen and flank3 == flank2 and list(flank3) == list(flank2)
    assert flank3[('x', 'y')] == 0 and ('x', 'y') in flank3                 # still a defaultdict(int)
    del flank3[('x', 'y')]
    kept = set(list(nx)[::2])
    m, idx = cluster.dict_to_matrix(flank3, kept, dense_matrix=False, add_self_loops=True, _device=True)
    m2, idx2 = cluster.dict_to_matrix(dict(flank2), kept, dense_matrix=False, add_self_loops=True, _device=True)
    assert idx == idx2 and list(idx) == list(idx2) and _same_matrix(m, m2)
    m.free()
    m2.free()
    for pair, data in coord.items():                                         # :2899-2902
        if isinstance(data, list):
            full3[pair] *= data[1]
            full2[pair] *= data[1]
    assert not full3.frozen and full3 == full2 and [type(v) for v in full3.values()] == [type(v) for v in full2.values()]
    cluster.output_pickle(full3, 'full', 'f3.pkl')
    got = pickle.load(open('f3.pkl', 'rb'))
    assert type(got) is defaultdict and got == full2 and [type(v) for v in got.values()] == [type(v) for v in full2.values()]
    # the handle goes with the last container
    import gc
    import weakref
    full4, flank4, HT4, clm4, _f, coord4 = _s5(fa_dict, aln, args)[:6]
    w = weakref.ref(full4._session)
    del full4, flank4, HT4, clm4, coord4
    gc.collect()
    assert w() is None


def body_split_contigs(tmp_path, monkeypatch):
    monkeypatch.chdir(tmp_path)
    gen, fa_dict, aln = _case(n_pairs=30_000, seed=13, split=True)
    args = _args(remove_allelic_links=2, max_read_pairs=40)
    out = _s5(fa_dict, aln, args, split=True)
    full, flank, HT, clm, frag_link, coord, pair_to_frag = out[:7]
    assert full.frozen and flank.frozen and any('_bin' in a for a, _b in pair_to_frag[next(iter(pair_to_frag))])
    ref = _s5(fa_dict, aln, args, split=True)
    for c in ref[:4] + (ref[5],):
        c._thaw()
    assert any('_bin' in a or '_bin' in b for a, b in ref[1])
    cluster.output_clm(clm)
    cluster._lib.files_join()
    os.rename('paired_links.clm', 'a.clm')
    cluster.output_clm(ref[3])
    assert open('a.clm', 'rb').read() == open('paired_links.clm', 'rb').read()
    kept = set(out[8])
    m, idx = cluster.dict_to_matrix(flank, kept, dense_matrix=False, add_self_loops=True, _device=True)
    m2, idx2 = cluster.dict_to_matrix(ref[1], kept, dense_matrix=False, add_self_loops=True, _device=True)
    assert idx == idx2 and list(idx) == list(idx2) and _same_matrix(m, m2)
    m.free()
    m2.free()
    assert flank.frozen
    assert flank == ref[1] and full == ref[0] and HT == ref[2] and coord == ref[5] and list(HT) == list(ref[2])


@pytest.fixture
def host_only(monkeypatch):
    import haphic_amd
    monkeypatch.setattr(cluster, '_lib', oracle_lib)
    monkeypatch.setattr(haphic_amd, '_lib', oracle_lib)


@pytest.mark.parametrize('normalize', [False, True])
def test_frozen_seams_equal_dict_seams_cpu(host_only, tmp_path, monkeypatch, normalize):
    body_frozen_seams_equal_dict_seams(tmp_path, monkeypatch, normalize)


def test_mutation_thaws_and_falls_back_cpu(host_only, tmp_path, monkeypatch):
    body_mutation_thaws_and_falls_back(tmp_path, monkeypatch)


def test_split_contigs_cpu(host_only, tmp_path, monkeypatch):
    body_split_contigs(tmp_path, monkeypatch)


@pytest.mark.gpu
@pytest.mark.parametrize('normalize', [False, True])
def test_frozen_seams_equal_dict_seams(tmp_path, monkeypatch, normalize):
    body_frozen_seams_equal_dict_seams(tmp_path, monkeypatch, normalize)


@pytest.mark.gpu
def test_mutation_thaws_and_falls_back(tmp_path, monkeypatch):
    body_mutation_thaws_and_falls_back(tmp_path, monkeypatch)


@pytest.mark.gpu
def test_split_contigs(tmp_path, monkeypatch):
    body_split_contigs(tmp_path, monkeypatch)


@pytest.mark.gpu
def test_device_clm_and_pickles_against_the_oracle(tmp_path):
    """hhx_ingest_write_clm / hhx_write_link_pickle against the oracle's restatement of output_clm :376-392 / output_pickle :710-715,
    on a stream with groups of one read pair (skipped), groups of thousands, equal distances, names of every length"""
    from haphic_amd import _lib
    rng = np.random.default_rng(3)
    n = 60
    names = ['c%d' % k if k % 3 else 'contig_with_a_rather_long_name_%d_%s' % (k, 'x' * (k * 7 % 300)) for k in range(n)]
    length = rng.integers(20_000, 3_000_000, n)
    order = sorted(range(n), key=names.__getitem__)
    rank = np.empty(n, np.int32)
    rank[order] = np.arange(n, dtype=np.int32)
    t = orc.FragTable(rank, length, np.arange(n, dtype=np.int32), np.zeros(n, np.uint8), 0, rank, length, np.ones(n, np.uint8))
    npairs = 300_000
    id1 = rng.integers(0, n, npairs).astype(np.int32)
    id2 = np.where(rng.random(npairs) < 0.5, (id1 + 1) % n, (id1 + 1 + rng.integers(0, n - 1, npairs)) % n).astype(np.int32)   # some heavy groups; never id1 (what
    assert (id1 != id2).all()                                                # pairs_generator_inter_ctgs :1582 hands to the loop)
    p1 = (rng.random(npairs) * length[id1]).astype(np.int32)
    p2 = (rng.random(npairs) * length[id2]).astype(np.int32)
    p2[::5] = p1[::5] % np.maximum(length[id2[::5]], 1)                     # ties
    want = orc.ingest(t, id1, p1, id2, p2, 10_000, bins=False, want_clm=True)
    ing = _lib.Ingest(t, 10_000)
    ing.keep_pairs()
    for s in range(0, npairs, 77_777):
        ing.push(id1[s:s + 77_777], p1[s:s + 77_777], id2[s:s + 77_777], p2[s:s + 77_777])
    ing.finalize()
    lines, nbytes = ing.write_clm(str(tmp_path / 'x.clm'), names)
    text = orc.clm_text(names, want['full_i'], want['full_j'], want['clm_ptr'], want['clm'])
    got = open(tmp_path / 'x.clm', 'rb').read()
    assert nbytes == len(got) and lines == got.count(b'\n')
    assert got == text
    out = ing.fetch()
    # HT_link_dict's items, ordered on the device, against the host ordering of the stream positions
    ni, nj, cnt = ing.fetch_ht_items()
    first = ing.fetch_ht_order()
    k, q = np.nonzero(out['ht_cnt'])
    order = np.argsort(first[k, q], kind='stable')
    k, q = k[order], q[order]
    assert ing.n_ht_items() == len(k) == len(ni) and np.array_equal(out['ht_cnt'], want['ht_cnt'])
    assert np.array_equal(ni, 2 * out['full_i'][k] + (q >> 1)) and np.array_equal(nj, 2 * out['full_j'][k] + (q & 1)) and np.array_equal(cnt, out['ht_cnt'][k, q])
    assert np.array_equal(first, orc.ht_first(t, id1, p1, id2, p2, want['full_i'], want['full_j']))
    _lib.write_link_pickle(str(tmp_path / 'f.pkl'), out['full_i'], out['full_j'], out['full_cnt'], names)
    assert pickle.load(open(tmp_path / 'f.pkl', 'rb')) == pickle.loads(orc.link_pickle(names, want['full_i'], want['full_j'], want['full_cnt']))
    ing.destroy()
    # nothing to write: an empty file, like the reference
    ing = _lib.Ingest(t, 10_000, skip_intra=True)
    ing.keep_pairs()
    ing.push(id1[:4], p1[:4], id1[:4], p2[:4])                               # intra-contig pairs only: dropped
    ing.finalize()
    assert ing.write_clm(str(tmp_path / 'e.clm'), names) == (0, 0) and os.path.getsize(tmp_path / 'e.clm') == 0
    ing.destroy()


@pytest.mark.gpu
def test_queued_files_equal_the_files_written_in_place(tmp_path, monkeypatch):
    """output_pickle / output_clm on frozen containers only queue the file (hhx_jobs.hip: one writer thread, its own stream and
    pool arena) and the seams that follow run meanwhile.  The queued files are byte for byte the files the same calls write in place
    (HAPHIC_SYNC_FILES=1), the seams in between see the same tables, the kept read pairs leave HBM with paired_links.clm, and what a
    writer cannot do surfaces as RuntimeError — at the call when it can be known there, at files_join() otherwise."""
    from haphic_amd import _lib
    monkeypatch.chdir(tmp_path)
    gen, fa_dict, aln = _case(n_pairs=400_000, seed=21)
    args = _args()
    re_dict = {c: v[2] for c, v in fa_dict.items()}

    def sequence(tag):
        os.makedirs(tag)
        os.chdir(tag)
        full, flank, HT, clm, frag_link, coord, ctg_len, nx = _s5(fa_dict, aln, args)
        cluster.output_pickle(HT, 'HT_link_dict', 'HT_links.pkl')
        cluster.output_clm(clm)
        kept = cluster.filter_fragments(nx, re_dict, 5, frag_link, '0.2X', '1.9X', 10, '1.5X', 0, flank, {}, '1.5X', set())
        cluster.output_pickle(full, 'full_link_dict', 'full_links.pkl')
        m, idx = cluster.dict_to_matrix(flank, kept, dense_matrix=False, add_self_loops=True)
        arrays = m.take_device().to_arrays()
        assert all(c.frozen for c in (full, flank, HT, clm))
        os.chdir('..')
        return kept, idx, arrays, (full, flank, HT, clm)

    monkeypatch.setenv('HAPHIC_SYNC_FILES', '1')
    kept0, idx0, arrays0, _c0 = sequence('in_place')
    assert _lib.files_pending()[0] == 0
    monkeypatch.delenv('HAPHIC_SYNC_FILES')
    done_before = _lib.files_pending()[1]
    kept1, idx1, arrays1, held = sequence('queued')
    _lib.files_join()
    assert _lib.files_pending() == (0, done_before + 3)
    assert kept0 == kept1 and idx0 == idx1 and all(np.array_equal(a, b) for a, b in zip(arrays0, arrays1))
    for f in ('HT_links.pkl', 'paired_links.clm', 'full_links.pkl'):
        a, b = open(os.path.join('in_place', f), 'rb').read(), open(os.path.join('queued', f), 'rb').read()
        assert a == b and len(a) > 1000, f
    # the read pairs went with the CLM file (no coordinate lists asked for, HT already queued): what needs them says so
    session = held[0]._session
    assert session.ing is not None
    with pytest.raises(RuntimeError, match='released'):
        session.ing.fetch_ht_order()
    assert held[0].frozen and len(held[0]) == len(pickle.load(open('queued/full_links.pkl', 'rb')))
    # failures: at the call (the directory does not exist) ...
    full, flank, HT, clm = _s5(fa_dict, aln, args)[:4]
    with pytest.raises(RuntimeError, match='cannot open'):
        cluster.output_pickle(HT, 'HT_link_dict', 'no_such_dir/HT_links.pkl')
    # ... and at the join (the device is full: /dev/full accepts the open and refuses every write)
    cluster.output_pickle(full, 'full_link_dict', '/dev/full')
    with pytest.raises(RuntimeError, match='/dev/full'):
        _lib.files_join()
    _lib.files_join()                                                # the failure was reported once
    # a handle is not destroyed under a queued job
    cluster.output_pickle(HT, 'HT_link_dict', 'late.pkl')
    del full, flank, HT, clm
    import gc
    gc.collect()
    _lib.files_join()
    assert type(pickle.load(open('late.pkl', 'rb'))) is defaultdict


def test_library_pickle_writer_queued(tmp_path):
    """hhx_write_link_pickle_async + hhx_files_join are host code: same bytes as the call in place, failures raised by the join"""
    from haphic_amd import _lib
    rng = np.random.default_rng(3)
    names = ['ctg%d' % k for k in range(900)]
    n = 700_000
    i, j = rng.integers(0, len(names), n).astype(np.int32), rng.integers(0, len(names), n).astype(np.int32)
    cnt = rng.integers(1, 1 << 33, n)
    _lib.write_link_pickle(str(tmp_path / 'a.pkl'), i, j, cnt, names)
    _lib.write_link_pickle_async(str(tmp_path / 'b.pkl'), i, j, cnt, names)
    _lib.write_link_pickle_async(str(tmp_path / 'c.pkl'), i[:10], j[:10], cnt[:10], names)
    _lib.files_join()
    assert (tmp_path / 'a.pkl').read_bytes() == (tmp_path / 'b.pkl').read_bytes()
    assert len(pickle.load(open(tmp_path / 'c.pkl', 'rb'))) == len(set(zip(i[:10].tolist(), j[:10].tolist())))
    with pytest.raises(RuntimeError, match='cannot open'):
        _lib.write_link_pickle_async(str(tmp_path / 'nope' / 'd.pkl'), i, j, cnt, names)
    _lib.write_link_pickle_async('/dev/full', i, j, cnt, names)
    _lib.write_link_pickle_async(str(tmp_path / 'e.pkl'), i, j, cnt, names)         # the queue goes on after a failed file
    with pytest.raises(RuntimeError, match='No space left'):
        _lib.files_join()
    assert (tmp_path / 'a.pkl').read_bytes() == (tmp_path / 'e.pkl').read_bytes() and _lib.files_pending()[0] == 0
    _lib.files_join()


def test_library_pickle_writer_against_pickle(tmp_path):
    """hhx_write_link_pickle is host code: every opcode path (memo below / above 256 slots, names of 256 bytes and more, empty and
    non-ASCII names, BININT1 / BININT2 / BININT / LONG1 values incl. negatives, batch boundaries, several encoder slices)"""
    from haphic_amd import _lib
    rng = np.random.default_rng(1)
    names = ['ctg%d' % k for k in range(700)] + ['x' * 300, 'é' * 140, '']
    for n in (0, 1, 999, 1000, 1001, 50_000, 1_300_000):
        i = rng.integers(0, len(names), n).astype(np.int32)
        j = rng.integers(0, len(names), n).astype(np.int32)
        _, first = np.unique(i.astype(np.int64) * 10_000 + j, return_index=True)
        first.sort()
        i, j = i[first], j[first]
        cnt = rng.choice([0, 1, 5, 255, 256, 65535, 65536, 2**31 - 1, 2**31, 2**40 + 7, -1, -129, -2**31, -2**31 - 1, 2**62], len(i)).astype(np.int64)
        path = str(tmp_path / 't.pkl')
        nb = _lib.write_link_pickle(path, i, j, cnt, names)
        assert nb == os.path.getsize(path)
        with open(path, 'rb') as fh:
            got = pickle.load(fh)
        want = pickle.loads(orc.link_pickle(names, i, j, cnt))
        assert type(got) is defaultdict and got.default_factory is int
        assert list(got.items()) == list(want.items()) and all(type(v) is int for v in got.values())
    with pytest.raises(RuntimeError):
        _lib.write_link_pickle(str(tmp_path / 'bad.pkl'), np.array([len(names)], np.int32), np.array([0], np.int32), np.array([1]), names)
