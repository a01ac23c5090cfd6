"""End-to-end parity of the host mirror (haphic_amd/cluster.py) on the GPU against outputs frozen from
the reference's own run (tests/golden/pipeline_toy.npz, made by tests/golden/make_golden.py):
parse_alignments_for_ctgs -> dict_to_matrix -> run_mcl_clustering, cluster files compared byte for byte
(the "integer contig -> group map" of BASELINE.json), plus the log line HapHiC_pipeline.py:385 parses."""
import logging
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


class Args:
    flank = 500
    remove_allelic_links = 0
    remove_concentrated_links = False
    max_read_pairs = 200
    nwindows = 50


@pytest.mark.parametrize('block_rows', [None, 37])
def test_cluster_files_byte_identical(golden_pipeline, tmp_path, block_rows):
    """block_rows: the inflation sweep on M^2 held as row blocks (the n^2 >= 2^31 regime) must write the same files"""
    from haphic_amd import cluster
    g = golden_pipeline
    names = [str(x) for x in g['names']]
    fa_dict = {n: [None, int(l), int(r)] for n, l, r in zip(names, g['length'], g['re_sites'])}
    aln = ((names[a], names[b], int(x), int(y)) for a, x, b, y in zip(g['id1'], g['pos1'], g['id2'], g['pos2']))
    frag_len_dict = {n: fa_dict[n][1] for n in names}
    Nx_set = set(names)
    full, flank, HT, clm, frag_link, coord = cluster.parse_alignments_for_ctgs(
        aln, fa_dict, Args(), frag_len_dict, Nx_set, 'int32', 'int32')
    assert sum(full.values()) == len(g['id1'])
    mat, fidx = cluster.dict_to_matrix(flank, Nx_set, dense_matrix=False, add_self_loops=True, _device=True)
    assert [fidx[n] for n in names] == g['frag_index'].tolist()
    records = []
    handler = logging.Handler()
    handler.emit = lambda rec: records.append(rec.getMessage())
    cluster.logger.addHandler(handler)
    cluster.logger.setLevel('INFO')
    try:
        res, nrounds = cluster.run_mcl_clustering(mat, set(), frag_len_dict, fidx, 2, 1.2, 2.0, 0.4, 200, 1e-4, fa_dict,
                                                  int(g['nchrs']), False, outdir_root=str(tmp_path), _block_rows=block_rows)
    finally:
        cluster.logger.removeHandler(handler)
    assert nrounds == 3
    for infl in g['inflations']:
        infl = str(infl)
        d = tmp_path / ('inflation_' + infl)
        got = (d / 'mcl_inflation_{}.clusters.txt'.format(infl)).read_text()
        assert got == str(g['clusters_txt_' + infl]), 'cluster file differs at inflation ' + infl
        groups = sorted(f for f in os.listdir(d) if f.startswith('group'))
        assert groups == [str(x) for x in g['group_files_' + infl]]
        assert (d / groups[0]).read_text() == str(g['group0_txt_' + infl])
    want = str(g['log_recommend'][0])
    if want:
        assert want in records, (want, [m for m in records if 'You could try' in m])


def test_sharded_hip_engine_world1_nccl():
    """the HIP engine of haphic_amd/sharded.py (torch views of library buffers, RCCL collectives, table
    merge) on one GPU through a world-size-1 nccl group: must equal the single-process kernels bit for bit"""
    import torch
    import torch.distributed as dist
    from haphic_amd import _lib, sharded, synth
    from oracle import oracle as orc
    from tests.test_gpu_kernels import clustered_stochastic
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    os.environ.setdefault('MASTER_PORT', '29577')
    dist.init_process_group('nccl', rank=0, world_size=1, device_id=torch.device('cuda:0'))
    try:
        A = clustered_stochastic(6000, 100, 12, 1, 5)
        d = _lib.DeviceCSR.from_arrays(*A)
        res, n_iter, conv, stats = sharded.mcl_sharded(d, 2, 2.0, 100, 1e-4, dist, 'cuda:0')
        ref, n2, c2, st2 = _lib.mcl(d, 2, 2.0, 100, 1e-4, want_stats=True, normalized=True)
        assert (n_iter, conv) == (n2, c2)
        assert all(np.array_equal(x, y) for x, y in zip(res.to_arrays(), ref.to_arrays()))
        assert np.array_equal(stats, st2)
        # sharded ingest exchange on one rank == plain finalize
        gen = synth.make_genome(3, 400_000, 10_000, seed=2)
        n = gen.n
        lex = gen.lexical_rank()
        t = orc.FragTable(lex, gen.length, np.arange(n, dtype=np.int32), np.zeros(n, np.uint8), 0, lex, gen.length,
                          np.ones(n, np.uint8))
        id1, p1, id2, p2 = synth.sample_pairs(gen, 60_000, seed=3, device='cuda:0')
        ing = _lib.Ingest(t, 3000, bins=False, skip_intra=True)
        ing.push_device(id1.numel(), id1.data_ptr(), p1.data_ptr(), id2.data_ptr(), p2.data_ptr())
        torch.cuda.synchronize()
        ing.finalize()
        in_set = np.ones(n, np.uint8)
        m1, fidx, nl1 = ing.link_matrix(in_set)
        b = ing.fetch()
        for full_tables in (False, True):
            m, n_linked, merged = sharded.merge_flank_and_build(ing, t, 3000, False, in_set, dist, 'cuda:0', full_tables=full_tables)
            assert merged.n_flank == ing.n_flank and nl1 == n_linked
            assert all(np.array_equal(x, y) for x, y in zip(m.to_arrays(), m1.to_arrays()))
            a = merged.fetch()
            for k in a:
                if full_tables or k.startswith('flank') or k == 'frag_links':
                    assert np.array_equal(a[k], b[k]), k
        # row-owner build (all-reduce(min) of first positions + all-to-all(v) of matrix entries), world 1
        in_set[::13] = 0
        m1, fidx1, nl1 = ing.link_matrix(in_set)
        eng = sharded.HipEngine('cuda:0')
        block, fi, n_linked, shape = sharded.build_link_matrix_sharded(eng, ing, in_set, dist)
        assert (n_linked, shape) == (nl1, m1.shape3[0]) and np.array_equal(fi[fidx1 >= 0], fidx1[fidx1 >= 0])
        assert all(np.array_equal(x, y) for x, y in zip(block.to_arrays(), m1.to_arrays()))
        # ... and MCL started from the row block == MCL on the replicated matrix
        _lib.normalize_l1(block)
        r_b, it_b, cv_b, _ = sharded.mcl_sharded(None, 2, 2.0, 100, 1e-4, dist, 'cuda:0', local_block=block, n=shape)
        _lib.normalize_l1(m1)
        r_f, it_f, cv_f, _ = sharded.mcl_sharded(m1, 2, 2.0, 100, 1e-4, dist, 'cuda:0')
        assert (it_b, cv_b) == (it_f, cv_f) and all(np.array_equal(x, y) for x, y in zip(r_b.to_arrays(), r_f.to_arrays()))
    finally:
        dist.destroy_process_group()


def test_row_owner_build_two_chunks_one_gpu():
    """the device half of the multi-GPU link-matrix build with TWO chunks of a stream on one GPU: per-chunk first
    positions reduced with min, one ranking, per-owner entry slices swapped by hand (what all-to-all(v) does), and
    the owners' row blocks stacked == the link matrix of the whole stream (counts of keys seen in both chunks add up)"""
    import torch
    from haphic_amd import _lib, sharded, synth
    from oracle import oracle as orc
    gen = synth.make_genome(4, 600_000, 9_000, seed=8)
    n = gen.n
    lex = gen.lexical_rank()
    t = orc.FragTable(lex, gen.length, np.arange(n, dtype=np.int32), np.zeros(n, np.uint8), 0, lex, gen.length, np.ones(n, np.uint8))
    id1, p1, id2, p2 = synth.sample_pairs(gen, 300_000, seed=9, device='cuda:0')
    torch.cuda.synchronize()
    in_set = np.ones(n, np.uint8)
    in_set[::7] = 0
    whole = _lib.Ingest(t, 4000, bins=False, skip_intra=True)
    whole.push_device(id1.numel(), id1.data_ptr(), p1.data_ptr(), id2.data_ptr(), p2.data_ptr())
    whole.finalize()
    want, widx, wl = whole.link_matrix(in_set)
    cut = 131_071
    chunks = []
    for lo, hi in ((0, cut), (cut, id1.numel())):
        ing = _lib.Ingest(t, 4000, bins=False, skip_intra=True)
        ing.set_ordinal_base(lo)
        ing.push_device(hi - lo, id1[lo:hi].data_ptr(), p1[lo:hi].data_ptr(), id2[lo:hi].data_ptr(), p2[lo:hi].data_ptr())
        ing.finalize()
        chunks.append(ing)
    eng = sharded.HipEngine('cuda:0')
    states = [eng.shard_open(c, in_set) for c in chunks]
    first = torch.minimum(eng.shard_first(states[0]), eng.shard_first(states[1]))
    fidx, n_linked = eng.rank_first(first)
    shape = int(in_set.sum())
    assert n_linked == wl and np.array_equal(fidx.cpu().numpy()[widx >= 0], widx[widx >= 0])
    for bounds in ([0, shape // 3, shape], [0, 0, shape], [0, shape, shape]):
        sent = [eng.shard_emit(st, fidx, bounds) for st in states]
        blocks = []
        for owner in range(2):
            parts0, parts1 = [], []
            for w0, w1, counts in sent:
                off = sum(counts[:owner])
                parts0.append(w0[off:off + counts[owner]])
                parts1.append(w1[off:off + counts[owner]])
            blocks.append(eng.rows_from_entries(torch.cat(parts0).contiguous(), torch.cat(parts1).contiguous(), bounds[owner], bounds[owner + 1], shape))
        a, b = [x.to_arrays() for x in blocks]
        wp, wj, wx = want.to_arrays()
        assert np.array_equal(np.concatenate([a[0][:-1], b[0] + a[0][-1]]), wp)
        assert np.array_equal(np.concatenate([a[1], b[1]]), wj) and np.array_equal(np.concatenate([a[2], b[2]]), wx)
        for x in blocks:
            x.free()
    for st in states:
        eng.shard_close(st)


def _check_files(g, tmp_path, records):
    for infl in g['inflations']:
        infl = str(infl)
        d = tmp_path / ('inflation_' + infl)
        got = (d / 'mcl_inflation_{}.clusters.txt'.format(infl)).read_text()
        assert got == str(g['clusters_txt_' + infl]), 'cluster file differs at inflation ' + infl
        groups = sorted(f for f in os.listdir(d) if f.startswith('group'))
        assert groups == [str(x) for x in g['group_files_' + infl]]
        for name, want in zip(groups, g['group_txt_' + infl]):
            assert (d / name).read_text() == str(want), name
    want = str(g['log_recommend'][0])
    if want:
        assert want in records, (want, [m for m in records if 'You could try' in m])


def _run_clustering(cluster, mat, bin_set, frag_len_dict, fidx, fa_dict, nchrs, infl, tmp_path):
    records = []
    handler = logging.Handler()
    handler.emit = lambda rec: records.append(rec.getMessage())
    cluster.logger.addHandler(handler)
    cluster.logger.setLevel('INFO')
    try:
        cluster.run_mcl_clustering(mat, bin_set, frag_len_dict, fidx, 2, infl[0], infl[1], infl[2], 200, 1e-4, fa_dict, nchrs, False,
                                   outdir_root=str(tmp_path))
    finally:
        cluster.logger.removeHandler(handler)
    return records


def test_cluster_files_with_split_contigs(tmp_path):
    """parse_alignments (contigs split into bins, :1658-1752) -> dict_to_matrix -> run_mcl_clustering with the
    bin -> contig vote (:2172-2194): files byte-identical to the reference's frozen run"""
    from haphic_amd import cluster
    from tests.conftest import load_golden
    g = load_golden('pipeline_bins.npz')
    names = [str(x) for x in g['names']]
    fa_dict = {n: [None, int(l), int(r)] for n, l, r in zip(names, g['length'], g['re_sites'])}
    frag_names = [str(x) for x in g['frag_names']]
    frag_len_dict = {f: int(l) for f, l in zip(frag_names, g['frag_len'])}
    Nx_frag_set = {f for f, x in zip(frag_names, g['frag_nx']) if x}
    bin_set = {f for f, x in zip(frag_names, g['frag_is_bin']) if x}
    split_ctg_set = {n for n, x in zip(names, g['split']) if x}
    aln = ((names[a], names[b], int(x), int(y)) for a, x, b, y in zip(g['id1'], g['pos1'], g['id2'], g['pos2']))

    class A(Args):
        flank = 50
    full, flank, HT, clm, frag_link, coord, _ = cluster.parse_alignments(
        aln, fa_dict, A(), int(g['bin_size']), frag_len_dict, Nx_frag_set, split_ctg_set, 'int32', 'int32')
    assert (len(full), len(flank), sum(full.values())) == (int(g['n_full']), int(g['n_flank']), int(g['full_total']))
    mat, fidx = cluster.dict_to_matrix(flank, Nx_frag_set, dense_matrix=False, add_self_loops=True, _device=True)
    assert [fidx.get(f, -1) for f in frag_names] == g['frag_index'].tolist()
    records = _run_clustering(cluster, mat, bin_set, frag_len_dict, fidx, fa_dict, int(g['nchrs']), (1.2, 2.4, 0.4), tmp_path)
    _check_files(g, tmp_path, records)


def test_parse_alignments_seven_containers_allelic(monkeypatch):
    """parse_alignments with --remove_allelic_links on split contigs (:1658-1752): all seven containers — link
    tables, HT, CLM, frag link counts, the first max_read_pairs coordinates handed to cal_concordance_ratio
    (:454-471) and ctg_pair_to_frag (:1731-1733) — equal to the reference's frozen output, dict order included"""
    from haphic_amd import cluster
    from tests.conftest import load_golden
    g = load_golden('ingest_bins.npz')
    names = [str(x) for x in g['names']]
    frag_names = [str(x) for x in g['frag_names']]
    fa_dict = {n: [None, int(l), 0] for n, l in zip(names, g['ctg_len'])}
    frag_len_dict = {f: int(l) for f, l in zip(frag_names, g['frag_len'])}
    Nx_frag_set = {f for f, x in zip(frag_names, g['frag_nx']) if x}
    split_ctg_set = {n for n, x in zip(names, g['ctg_split']) if x}

    def nm(i):
        return names[i] if i >= 0 else 'unplaced_scaffold'
    aln = ((nm(a), nm(b), int(x), int(y)) for a, x, b, y in zip(g['id1'], g['pos1'], g['id2'], g['pos2']))

    class A(Args):
        flank = int(g['flank']) // 1000
        remove_allelic_links = 4
        max_read_pairs = int(g['max_read_pairs'])
        nwindows = 50
    monkeypatch.setattr(cluster, 'cal_concordance_ratio', lambda coord_list, shorter_len, nwindows: tuple(coord_list))
    full, flank, HT, clm, frag_link, coord, c2f = cluster.parse_alignments(
        aln, fa_dict, A(), int(g['bin_size']), frag_len_dict, Nx_frag_set, split_ctg_set, 'int32', 'int32')
    assert list(full.items()) == [((names[i], names[j]), c) for i, j, c in zip(g['full_i'], g['full_j'], g['full_cnt'].tolist())]
    assert list(flank.items()) == [((frag_names[i], frag_names[j]), c)
                                   for i, j, c in zip(g['flank_i'], g['flank_j'], g['flank_cnt'].tolist())]
    assert [frag_link.get(f, 0) for f in frag_names] == g['frag_links'].tolist()
    cp, kp = g['clm_ptr'].tolist(), g['crd_ptr'].tolist()
    for k, pair in enumerate(full):
        assert list(clm[pair]) == g['clm'][cp[k]:cp[k + 1]].tolist()
        v = coord[pair]
        v = list(v[0]) if isinstance(v, list) else list(v)
        assert v == g['crd'][kp[k]:kp[k + 1]].tolist()
        if kp[k + 1] - kp[k] >= 2 * A.max_read_pairs:
            assert isinstance(coord[pair], list) and coord[pair][1] == 1
        ht = g['ht_cnt'][k]
        for q, (a, b) in enumerate((('H', 'H'), ('H', 'T'), ('T', 'H'), ('T', 'T'))):
            assert HT.get((pair[0] + '_' + a, pair[1] + '_' + b), 0) == ht[q]
    quad = (('_H', '_H'), ('_H', '_T'), ('_T', '_H'), ('_T', '_T'))
    assert list(HT) == [(names[g['full_i'][k]] + quad[q][0], names[g['full_j'][k]] + quad[q][1]) for k, q in g['ht_order'].tolist()]
    want = {}
    for ci, cj, fi, fj in g['c2f'].tolist():
        want.setdefault((names[ci], names[cj]), set()).add((frag_names[fi], frag_names[fj]))
    assert dict(c2f) == want


def test_cluster_files_c1_config(tmp_path):
    """BASELINE.json configs[0]: ~1k contigs / 1 M pairs / nchrs = 4 — the reference's own CPU-runnable case"""
    from haphic_amd import cluster, synth
    from tests.conftest import load_golden
    g = load_golden('pipeline_c1.npz')
    gen = synth.make_genome(4, 25_000_000, 100_000, cv=0.3, min_len=5000, seed=12345)
    names = list(gen.names)
    assert len(names) == int(g['n_contigs'])
    id1, p1, id2, p2 = [t.numpy() for t in synth.sample_pairs(gen, 1_000_000, seed=12345)]
    if int(id1.sum() + p1.sum() + id2.sum() + p2.sum()) != int(g['pairs_checksum']):
        pytest.skip('torch CPU generator differs from the one that made the fixture')
    fa_dict = {n: [None, int(l), int(r)] for n, l, r in zip(names, gen.length, gen.re_sites)}
    keep = id1 != id2
    aln = ((names[a], names[b], int(x), int(y)) for a, x, b, y in zip(id1[keep], p1[keep], id2[keep], p2[keep]))
    frag_len_dict = {n: fa_dict[n][1] for n in names}
    Nx_set = set(names)
    full, flank, HT, clm, frag_link, coord = cluster.parse_alignments_for_ctgs(aln, fa_dict, Args(), frag_len_dict, Nx_set, 'int32', 'int32')
    assert (len(full), len(flank), sum(full.values())) == (int(g['n_full']), int(g['n_flank']), int(g['full_total']))
    mat, fidx = cluster.dict_to_matrix(flank, Nx_set, dense_matrix=False, add_self_loops=True, _device=True)
    assert [fidx[n] for n in names] == g['frag_index'].tolist() and mat.nnz == int(g['matrix_nnz'])
    records = _run_clustering(cluster, mat, set(), frag_len_dict, fidx, fa_dict, 4, (1.4, 2.2, 0.4), tmp_path)
    _check_files(g, tmp_path, records)


def test_cluster_files_c4_allele_aware(tmp_path):
    """BASELINE.json configs[3] at test scale: autotetraploid, --remove_allelic_links 4.  The device ingest must hand
    remove_allelic_HiC_links (:474-689, the reference's own Python: cliques + Hungarian matching) exactly the
    ctg_coord_dict the reference loop builds — key order, concordance ratios of the collapsed entries (:460-465, host
    statistic on the first max_read_pairs coordinates), raw coordinate lists; with its frozen verdict applied (which
    contig pairs and fragments go), dict_to_matrix + the inflation sweep write byte-identical cluster files"""
    from haphic_amd import cluster, synth
    from tests.conftest import load_golden
    g = load_golden('pipeline_c4.npz')
    base = synth.make_genome(3, 3_000_000, 60_000, cv=0.3, min_len=8000, seed=4040)
    gen = synth.make_polyploid(base, 4)
    id1, p1, id2, p2 = [t.numpy() for t in synth.sample_pairs(gen, 400_000, seed=4041, cis=0.9)]
    id1, p1, id2, p2 = synth.add_allelic_pairs(gen, base.n, 4, id1, p1, id2, p2, 0.08, 4042)
    keep = id1 != id2
    id1, p1, id2, p2 = id1[keep], p1[keep], id2[keep], p2[keep]
    if int(id1.sum() + p1.sum() + id2.sum() + p2.sum()) != int(g['pairs_checksum']):
        pytest.skip('torch CPU generator differs from the one that made the fixture')
    names = list(gen.names)
    assert len(names) == int(g['n_contigs'])
    fa_dict = {n: [None, int(l), int(r)] for n, l, r in zip(names, gen.length, gen.re_sites)}
    aln = ((names[a], names[b], int(x), int(y)) for a, x, b, y in zip(id1, p1, id2, p2))

    class A(Args):
        remove_allelic_links = 4
        max_read_pairs = 40
        min_read_pairs = 20
        concordance_ratio_cutoff = 0.2
    frag_len_dict = {n: fa_dict[n][1] for n in names}
    full, flank, HT, clm, frag_link, coord = cluster.parse_alignments_for_ctgs(aln, fa_dict, A(), frag_len_dict, set(names), 'int32', 'int32')
    assert (len(full), len(flank)) == (int(g['n_full']), int(g['n_flank']))
    # ctg_coord_dict == the reference's, entry by entry and in order
    assert [(names[i], names[j]) for i, j in zip(g['coord_i'], g['coord_j'])] == list(coord)
    rp = g['coord_raw_ptr'].tolist()
    for k, pair in enumerate(coord):
        v = coord[pair]
        if g['coord_collapsed'][k]:
            assert isinstance(v, list) and v[1] == 1 and float(v[0]) == g['coord_ratio'][k], pair
        else:
            assert list(v) == g['coord_raw'][rp[k]:rp[k + 1]].tolist(), pair
    # the reference's remove_allelic_HiC_links verdict, applied to OUR dicts
    for k, gone in zip(list(full), g['full_removed']):
        if gone:
            del full[k]
    for k, gone in zip(list(flank), g['flank_removed']):
        if gone:
            del flank[k]
    remaining = {n for n, r in zip(names, g['remaining']) if r}
    mat, fidx = cluster.dict_to_matrix(flank, remaining, dense_matrix=False, add_self_loops=True, _device=True)
    assert [fidx.get(n, -1) for n in names] == g['frag_index'].tolist()
    records = _run_clustering(cluster, mat, set(), frag_len_dict, fidx, fa_dict, int(g['nchrs']), (1.4, 2.6, 0.4), tmp_path)
    _check_files(g, tmp_path, records)


def test_edge_cases():
    """empty stream, every pair dropped, a single key, link-less fragments only, unknown names"""
    from haphic_amd import _lib
    from oracle import oracle as orc
    n = 7
    rank = np.arange(n, dtype=np.int32)[::-1].copy()
    length = np.full(n, 10_000, np.int64)
    t = orc.FragTable(rank, length, np.arange(n, dtype=np.int32), np.zeros(n, np.uint8), 0, rank, length, np.ones(n, np.uint8))
    z = np.zeros(0, np.int32)
    ing = _lib.Ingest(t, 0, bins=False, skip_intra=True)
    ing.push(z, z, z, z)                                               # empty batch
    assert ing.finalize() == (0, 0)
    out = ing.fetch()
    assert out['full_i'].size == 0 and out['flank_cnt'].size == 0 and not out['frag_links'].any()
    m, fidx, nl = ing.link_matrix(np.ones(n, np.uint8))
    p, j, x = m.to_arrays()
    assert nl == 0 and (fidx == -1).all() and np.array_equal(p, np.arange(n + 1)) and np.array_equal(j, np.arange(n)) and (x == 1).all()
    res, n_iter, conv = _lib.mcl(m, 2, 2.0, 50, 1e-4, links=True)      # identity: n singleton clusters
    att, ptr, mem = _lib.interpret(res)
    assert conv and len(att) == n and np.array_equal(mem, np.arange(n))
    # every pair dropped: intra-contig, unknown names (-1), id out of range
    ing = _lib.Ingest(t, 0, bins=False, skip_intra=True)
    ing.push(np.array([1, -1, 3, 99], np.int32), np.array([5, 5, 5, 5], np.int32), np.array([1, 2, -1, 0], np.int32), np.array([9, 9, 9, 9], np.int32))
    assert ing.finalize() == (0, 0)
    # one key hit many times from both orientations, in two pushes
    ing = _lib.Ingest(t, 0, bins=False, skip_intra=True)
    a = np.array([2, 5] * 500, np.int32)
    b = np.array([5, 2] * 500, np.int32)
    pos = np.arange(1000, dtype=np.int32)
    ing.push(a[:300], pos[:300], b[:300], pos[:300])
    ing.push(a[300:], pos[300:], b[300:], pos[300:])
    assert ing.finalize() == (1, 1)
    out = ing.fetch()
    ref = orc.ingest(t, a, pos.astype(np.int64), b, pos.astype(np.int64), 0)
    for k in ('full_i', 'full_j', 'full_cnt', 'ht_cnt', 'flank_i', 'flank_j', 'flank_cnt', 'frag_links'):
        assert np.array_equal(out[k], ref[k]), k
    in_set = np.ones(n, np.uint8)
    m, fidx, nl = ing.link_matrix(in_set)
    rp, rj, rx, ridx, rl = orc.dict_to_matrix(ref['flank_i'], ref['flank_j'], ref['flank_cnt'].astype(np.float64), n, in_set, n - 2)
    assert nl == rl == 2 and np.array_equal(fidx, ridx)
    assert all(np.array_equal(u, v) for u, v in zip(m.to_arrays(), (rp, rj, rx)))
    # a fragment of the only key is not in frag_set: nothing is linked
    in2 = in_set.copy()
    in2[5] = 0
    m2, fidx2, nl2 = ing.link_matrix(in2)
    assert nl2 == 0 and m2.shape3[0] == n - 1 and m2.nnz == n - 1


def test_re_sites_and_stat_fragments():
    """a5: count_RE_sites (:75-84, non-overlapping str.count semantics incl. self-overlapping sites) and
    stat_fragments (:188-296: bins, flank-only counting, seeded shuffle, Nx set, whitelist) against the reference"""
    from haphic_amd import _lib, cluster
    from tests.conftest import load_golden
    g = load_golden('resites.npz')
    for RE in g['REs']:
        RE = str(RE)
        sites = [str(x).encode() for x in g['sites_' + RE]]
        assert [x.decode() for x in cluster._sites_of(RE)] == [str(x) for x in g['sites_' + RE]]
        got = _lib.count_re_sites(g['seq'], g['seg_off'], g['seg_len'], sites)
        assert np.array_equal(got, g['counts_' + RE]), RE
    assert cluster.count_RE_sites('GATCGATC' + 'AAAAA', 'GATC,AAAA') == 3
    names = [str(x) for x in g['sf_names']]
    seq = bytes(g['sf_seq']).decode()
    fa, pos = {}, 0
    for n, L, re_ in zip(names, g['sf_len'], g['sf_re']):
        fa[n] = [seq[pos:pos + int(L)], int(L), int(re_)]
        pos += int(L)
    assert all(cluster.count_RE_sites(v[0], 'GATC,GANTC') + 1 == v[2] for v in fa.values())
    r = cluster.stat_fragments(fa, 'GATC,GANTC', {}, {'ptg003l'}, nchrs=2, flank=4, Nx=70, bin_size=20)
    assert [f for f, _ in r[0]] == [str(x) for x in g['sf_sorted']]
    assert r[2] == int(g['sf_bin_size']) and sorted(r[1]) == [str(x) for x in g['sf_bins']] and sorted(r[6]) == [str(x) for x in g['sf_split']]
    assert list(r[3]) == [str(x) for x in g['sf_frags']] and list(r[3].values()) == g['sf_frag_len'].tolist()
    assert sorted(r[4]) == [str(x) for x in g['sf_nx']]
    assert list(r[5]) == [str(x) for x in g['sf_re_frags']] and list(r[5].values()) == g['sf_re_counts'].tolist()
    assert all(v[0] is None for v in fa.values())


def test_filter_fragments_and_rank_sums():
    """f1: filter_fragments (:741-940) with the rank-sum statistic on the device, against the reference's kept sets;
    hhx_rank_sums against the oracle on rows with fewer links than topN, ties, and n < topN"""
    from haphic_amd import _lib, cluster
    from oracle import oracle as orc
    from tests.conftest import load_golden
    from tests.test_oracle_golden import _filter_inputs
    g = load_golden('filter.npz')
    names, Nx_set, RE_site_dict, frag_link, flank = _filter_inputs(g)
    cluster.logger.setLevel('WARNING')
    for k in range(int(g['n_cases'])):
        cut, lo, up, topn, rsu, hard = [str(x) for x in g['case%d_params' % k]]
        wl = {str(x) for x in g['case%d_whitelist' % k]} or None
        kept = cluster.filter_fragments(set(Nx_set), RE_site_dict, int(cut), frag_link, lo, up, int(topn), rsu, int(hard), flank, {}, '1.5X', wl)
        assert sorted(kept) == [str(x) for x in g['case%d_kept' % k]], k
    rng = np.random.default_rng(3)
    for n, deg, topn in ((300, 4, 10), (500, 40, 10), (7, 2, 10), (64, 1, 3), (200, 0, 5)):
        iu = rng.integers(0, n, n * deg)
        ju = rng.integers(0, n, n * deg)
        ok = iu != ju
        key = np.unique(np.minimum(iu, ju)[ok] * n + np.maximum(iu, ju)[ok])
        a, b = key // n, key % n
        v = rng.integers(1, 4, key.size).astype(np.float32)            # few distinct values: many ties
        rows = np.concatenate([a, b]); cols = np.concatenate([b, a]); vals = np.concatenate([v, v])
        order = np.lexsort((cols, rows))
        indptr = np.zeros(n + 1, np.int32)
        np.add.at(indptr, rows + 1, 1)
        A = (np.cumsum(indptr).astype(np.int32), cols[order].astype(np.int32), vals[order])
        m = _lib.DeviceCSR.from_arrays(*A)
        assert np.array_equal(_lib.rank_sums(m, topn), orc.rank_sums(A, topn)), (n, deg, topn)


def test_link_weights_a6():
    """a6: normalize_by_nlinks :718-724, normalize_by_length :727-738, reduce_inter_hap_HiC_links :695-707 — the device
    kernel on arrays and the dict mirrors of haphic_amd/cluster.py against what the reference's own functions did to
    the same dict (tests/golden/weights.npz).  float64 values: length / haplotype modes bit exact; the nlinks mode
    is Python's `** 0.5` (C pow) against the device square root + division: two ulp of float64 are allowed (measured: 2 of
    6000 values differ, by one ulp each side), 99 % must be bit equal (the matrix is float32, :368: compared at 1e-6)."""
    from collections import defaultdict
    from haphic_amd import _lib, cluster
    from tests.conftest import load_golden
    g = load_golden('weights.npz')
    fi, fj, cnt = g['fi'], g['fj'], g['cnt']
    n_frag = len(g['links'])
    names = ['ctg%04d' % k for k in range(n_frag)]
    v = cnt.astype(np.float64)
    _lib.link_weights(fi, fj, v, 0, n_frag, per_frag=g['links'])
    np.testing.assert_allclose(v, g['nlinks'], rtol=4.5e-16, atol=0)
    assert (v != g['nlinks']).mean() < 0.01
    v = cnt.astype(np.float64)
    _lib.link_weights(fi, fj, v, 1, n_frag, per_frag=g['length'], param=2000 * int(g['flank_kb']))
    assert np.array_equal(v, g['by_length'])
    for tag_, w in (('w1', 1.0), ('w05', 0.5), ('w03', 0.3)):
        v = cnt.astype(np.float64)
        nz = _lib.link_weights(fi, fj, v, 2, n_frag, tag=g['hap'], param=w)
        kept = np.flatnonzero(~((v == 0) & (g['hap'][fi] != g['hap'][fj])))
        assert np.array_equal(kept, g['hap_%s_kept' % tag_]) and nz == len(cnt) - len(kept)
        assert np.array_equal(v[kept], g['hap_%s_values' % tag_])

    def as_dict():
        d = defaultdict(int)
        for i, j, c in zip(fi.tolist(), fj.tolist(), cnt.tolist()):
            d[(names[i], names[j])] = c
        return d
    cluster.logger.setLevel('WARNING')
    rdd = {names[k]: ('h%d' % g['hap'][k], 30.0) for k in range(n_frag)}
    order = {(names[i], names[j]): k for k, (i, j) in enumerate(zip(fi.tolist(), fj.tolist()))}
    for tag_, w in (('w1', 1.0), ('w05', 0.5), ('w03', 0.3)):
        d = as_dict()
        cluster.reduce_inter_hap_HiC_links(d, rdd, w)
        assert [order[k] for k in d] == g['hap_%s_kept' % tag_].tolist()
        assert [float(x) for x in d.values()] == g['hap_%s_values' % tag_].tolist()
        assert [isinstance(x, int) for x in d.values()] == g['hap_%s_is_int' % tag_].tolist()      # untouched counts stay int
    d = as_dict()
    cluster.normalize_by_length(d, {names[k]: int(g['length'][k]) for k in range(n_frag)}, int(g['flank_kb']))
    assert list(d.values()) == g['by_length'].tolist() and all(isinstance(x, float) for x in d.values())
    d = as_dict()
    cluster.normalize_by_nlinks(d, {names[k]: int(g['links'][k]) for k in range(n_frag)})
    np.testing.assert_allclose(list(d.values()), g['nlinks'], rtol=4.5e-16, atol=0)
    m, fidx = cluster.dict_to_matrix(d, set(names), dense_matrix=False, add_self_loops=True)        # :2895 then :2934
    m.sort_indices()
    assert [fidx[nm] for nm in names] == g['nl_fidx'].tolist()
    assert np.array_equal(m.indptr, g['nl_m_p']) and np.array_equal(m.indices, g['nl_m_j'])
    np.testing.assert_allclose(m.data, g['nl_m_x'], rtol=1e-6, atol=0)


def test_reassign_group_link_sums_f3():
    """f3: HapHiC_reassign.parse_link_dict :217-263 — the per-(contig, group) link sums behind reassign's link densities
    (device: hhx_group_link_sums) rebuilt into the reference's nested dicts, both dict orders included, and
    linked_ctg_dict, against what the reference's own function returned (tests/golden/reassign.npz)"""
    from haphic_amd import _lib, cluster
    from tests.conftest import load_golden
    g = load_golden('reassign.npz')
    n_ctg = len(g['group'])
    names = ['c%03d' % k for k in range(n_ctg)]
    sums, first = _lib.group_link_sums(g['fi'], g['fj'], g['links'], g['group'], int(g['n_groups']))
    want = np.zeros_like(sums)
    for c, grp, v in g['cells'].tolist():
        want[c, grp] = v
    assert np.array_equal(sums, want) and np.array_equal(first >= 0, want > 0)
    link_dict = {(names[i], names[j]): int(v) for i, j, v in zip(g['fi'].tolist(), g['fj'].tolist(), g['links'].tolist())}
    ctg_group_dict = {names[k]: ('ungrouped' if g['group'][k] < 0 else 'group%d' % g['group'][k]) for k in range(n_ctg)}
    cgl, linked = cluster.parse_link_dict(link_dict, ctg_group_dict)
    cid = {n: k for k, n in enumerate(names)}
    assert [cid[c] for c in cgl] == g['outer'].tolist()                                        # outer dict order
    assert [(cid[c], int(grp[5:]), v) for c, inner in cgl.items() for grp, v in inner.items()] == [tuple(r) for r in g['cells'].tolist()]
    assert all(isinstance(v, int) for inner in cgl.values() for v in inner.values())
    lp = g['linked_ptr']
    for k, nm in enumerate(names):
        assert sorted(cid[x] for x in linked.get(nm, ())) == g['linked'][lp[k]:lp[k + 1]].tolist()
    with pytest.raises(ValueError):
        cluster.parse_link_dict({k: float(v) for k, v in link_dict.items()}, ctg_group_dict)     # float links: the reference function
    with pytest.raises(ValueError):
        cluster.parse_link_dict(link_dict, ctg_group_dict, normalize_by_nlinks=True)


def test_full_size_properties():
    """BASELINE.json configs[1] at full size (10k contigs / 50 M pairs — the oracle cannot finish this in seconds),
    checked through size-independent properties: count conservation, chunked == whole (the exchange step), symmetry
    and structure of the link matrix, stochastic rows, a valid partition, idempotence of the converged matrix, and
    bit-identical results across repeated runs and across the sharded (row-block) driver."""
    import torch
    from haphic_amd import _lib, synth
    from haphic_amd.cluster import FragTable, _clusters_from_arrays
    gen = synth.make_genome(16, 624 * 50_000, 50_000, seed=12345)
    n = gen.n
    table = FragTable.for_contigs(gen.lexical_rank(), gen.length, np.ones(n, np.uint8))
    P = 50_000_000
    id1, p1, id2, p2 = synth.sample_pairs(gen, P, seed=12345, device='cuda:0')
    kept = int((id1 != id2).sum().item())
    whole = _lib.Ingest(table, 500_000, bins=False, skip_intra=True)
    whole.push_device(P, id1.data_ptr(), p1.data_ptr(), id2.data_ptr(), p2.data_ptr())
    torch.cuda.synchronize()
    n_full, n_flank = whole.finalize()
    # the same stream in three uneven chunks, each aggregated on its own, then merged (multi-GPU exchange on one GPU)
    cuts = [0, 17_000_003, 40_000_000, P]
    merged = _lib.Ingest(table, 500_000, bins=False, skip_intra=True)
    parts = []
    for lo, hi in zip(cuts[:-1], cuts[1:]):
        part = _lib.Ingest(table, 500_000, bins=False, skip_intra=True)
        part.set_ordinal_base(lo)
        a = [x[lo:hi] for x in (id1, p1, id2, p2)]
        part.push_device(hi - lo, *[x.data_ptr() for x in a])
        torch.cuda.synchronize()
        part.finalize()
        parts.append(part)
        merged.push_table(0, *part.table_device(0))
    assert merged.finalize() == (n_full, n_flank)
    a, b = whole.fetch(), merged.fetch()
    for k in a:
        assert np.array_equal(a[k], b[k]), k                      # incl. dict insertion order
    assert int(a['full_cnt'].sum()) == kept and int(a['ht_cnt'].sum()) == kept        # every kept pair counted once
    assert int(a['frag_links'].sum()) == 2 * int(a['flank_cnt'].sum())
    assert (a['full_i'] != a['full_j']).all() and len(np.unique(a['full_i'].astype(np.int64) * n + a['full_j'])) == n_full
    in_set = np.ones(n, np.uint8)
    m, fidx, n_linked = whole.link_matrix(in_set)
    m2, fidx2, _ = merged.link_matrix(in_set)
    ip, ix, dx = m.to_arrays()
    assert all(np.array_equal(u, v) for u, v in zip((ip, ix, dx), m2.to_arrays())) and np.array_equal(fidx, fidx2)
    import scipy.sparse as sp
    M = sp.csr_matrix((dx, ix, ip), shape=(len(ip) - 1,) * 2)
    assert (M != M.T).nnz == 0 and (M.diagonal() == 1).all() and M.nnz == 2 * n_flank + M.shape[0]
    assert np.array_equal(np.sort(fidx[fidx >= 0]), np.arange(n_linked))
    rows = np.repeat(np.arange(M.shape[0]), np.diff(ip))
    assert (np.diff(ix)[np.diff(rows) == 0] > 0).all()            # rows sorted by column, no duplicates
    # MCL: stochastic, valid partition, idempotent, reproducible, shard-invariant
    res, n_iter, conv = _lib.mcl(m, 2, 2.0, 200, 1e-4, links=True)
    assert conv
    rp, rj, rx = res.to_arrays()
    sums = np.add.reduceat(rx.astype(np.float64), rp[:-1])
    assert np.abs(sums - 1).max() < 1e-6
    att, ptr, mem = _lib.interpret(res)
    clusters = _clusters_from_arrays(att, ptr, mem, res.shape3[0])
    assert clusters is not None and sum(len(c) for c in clusters) == res.shape3[0]
    again, _, _ = _lib.expand_inflate_prune(res, res, 2.0, 1e-4)   # one more iteration of a converged matrix
    np.testing.assert_allclose(again.to_arrays()[2], rx, rtol=1e-5)
    assert np.array_equal(again.to_arrays()[1], rj)
    res2, n2, c2 = _lib.mcl(m, 2, 2.0, 200, 1e-4, links=True)
    assert n2 == n_iter and all(np.array_equal(u, v) for u, v in zip(res2.to_arrays(), (rp, rj, rx)))
    norm = m.copy()
    _lib.normalize_l1(norm)
    r0, r1 = 3000, 5200                                           # a row block, as a GPU of a sharded run computes it
    blk, _, _ = _lib.expand_inflate_prune(norm.row_block(r0, r1), norm, 2.0, 1e-4)
    full, _, _ = _lib.expand_inflate_prune(norm, norm, 2.0, 1e-4)
    fp, fj, fx = full.to_arrays()
    bp, bj, bx = blk.to_arrays()
    lo, hi = fp[r0], fp[r1]
    assert np.array_equal(bp, fp[r0:r1 + 1] - lo) and np.array_equal(bj, fj[lo:hi]) and np.array_equal(bx, fx[lo:hi])
