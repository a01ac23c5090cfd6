"""Parity tests proper: the HIP kernels (through the C ABI) against the oracle and the golden vectors.
Run on the GPU box:  python -m pytest tests -m gpu"""
import numpy as np
import pytest

from haphic_amd import _lib
from oracle import oracle as orc

pytestmark = pytest.mark.gpu


def tri(g, prefix):
    return g[prefix + '_p'], g[prefix + '_j'], g[prefix + '_x']


def assert_close_csr(a, b, rtol, what):
    assert np.array_equal(a[0], b[0]), what + ': indptr'
    assert np.array_equal(a[1], b[1]), what + ': indices'
    np.testing.assert_allclose(a[2], b[2], rtol=rtol, atol=0, err_msg=what)


def random_stochastic(n, deg, seed):
    """row-stochastic CSR with ~deg random entries per row plus the diagonal (sorted, no duplicates)"""
    rng = np.random.default_rng(seed)
    rows = np.repeat(np.arange(n, dtype=np.int64), deg)
    cols = rng.integers(0, n, rows.size)
    key = np.unique(np.concatenate([rows * n + cols, np.arange(n, dtype=np.int64) * (n + 1)]))
    r, c = key // n, key % n
    indptr = np.zeros(n + 1, np.int32)
    np.add.at(indptr, r + 1, 1)
    indptr = np.cumsum(indptr).astype(np.int32)
    x = rng.random(key.size).astype(np.float32) + np.float32(0.01)
    return indptr, c.astype(np.int32), orc.normalize_l1(indptr, x)


def clustered_stochastic(n, block, deg_in, deg_out, seed):
    """like random_stochastic but most entries stay inside diagonal blocks (a clusterable graph, the
    regime MCL is used in: expansion fill-in stays bounded by the block size)"""
    rng = np.random.default_rng(seed)
    rows = np.repeat(np.arange(n, dtype=np.int64), deg_in)
    cols = (rows // block) * block + rng.integers(0, block, rows.size)
    rows_o = np.repeat(np.arange(n, dtype=np.int64), deg_out)
    cols_o = rng.integers(0, n, rows_o.size)
    rr = np.concatenate([rows, rows_o, cols, cols_o, np.arange(n, dtype=np.int64)])
    cc = np.minimum(np.concatenate([cols, cols_o, rows, rows_o, np.arange(n, dtype=np.int64)]), n - 1)
    key = np.unique(rr * n + cc)
    r, c = key // n, key % n
    indptr = np.zeros(n + 1, np.int32)
    np.add.at(indptr, r + 1, 1)
    indptr = np.cumsum(indptr).astype(np.int32)
    x = np.where(r // block == c // block, 20.0, 1.0).astype(np.float32) * (rng.random(key.size).astype(np.float32) + np.float32(0.5))
    return indptr, c.astype(np.int32), orc.normalize_l1(indptr, x)


# tolerance stated by BASELINE.json north_star: normalised values within 1e-6 relative
RTOL = 1e-6


def test_normalize_l1(golden_mcl):
    g = golden_mcl
    for tag in g['cases']:
        p, j, x = tri(g, tag + '_link')
        m = _lib.DeviceCSR.from_arrays(p, j, x)
        _lib.normalize_l1(m)
        got = m.to_arrays()
        assert_close_csr(got, tri(g, tag + '_norm'), RTOL, 'normalize ' + str(tag))


def test_spgemm_bit_exact_vs_fixed_point_spec(golden_mcl):
    g = golden_mcl
    for tag in g['cases']:
        A = tri(g, tag + '_norm')
        d = _lib.DeviceCSR.from_arrays(*A)
        for shift in (60, 62, 40):
            c, f = _lib.spgemm(d, d, fx_shift=shift, want_products=True)
            ref = orc.spgemm(A, A, mode=1, fx_shift=shift)
            got = c.to_arrays()
            assert np.array_equal(got[0], ref[0]) and np.array_equal(got[1], ref[1])
            assert np.array_equal(got[2], ref[2]), 'fixed-point spgemm not bit exact (shift %d)' % shift
        # and against the reference stand-in (scipy float32) within the stated tolerance
        c = _lib.spgemm(d, d)
        assert_close_csr(c.to_arrays(), tri(g, tag + '_m2'), RTOL, 'spgemm vs reference ' + str(tag))


@pytest.mark.parametrize('n,deg,seed', [(1000, 8, 1), (5000, 40, 2), (3000, 300, 3), (70000, 3, 4), (257, 200, 5)])
def test_spgemm_random(n, deg, seed):
    A = random_stochastic(n, deg, seed)
    d = _lib.DeviceCSR.from_arrays(*A)
    c, f = _lib.spgemm(d, d, fx_shift=60, want_products=True)          # generic kernels (64-bit integer accumulation)
    ref = orc.spgemm(A, A, mode=1, fx_shift=60)
    got = c.to_arrays()
    assert np.array_equal(got[0], ref[0]) and np.array_equal(got[1], ref[1]) and np.array_equal(got[2], ref[2])
    c52, f52 = _lib.spgemm(d, d, fx_shift=52, want_products=True)      # stochastic operands: fused kernels, plain-product mode
    ref52 = orc.spgemm(A, A, mode=1, fx_shift=52)
    assert f52 == f and all(np.array_equal(x, y) for x, y in zip(c52.to_arrays(), ref52))
    # operands outside [0, 1] take the generic kernels at any shift
    A2 = (A[0], A[1], (A[2] * np.float32(-3.0)).astype(np.float32))
    d2 = _lib.DeviceCSR.from_arrays(*A2)
    g2 = _lib.spgemm(d2, d2, fx_shift=40).to_arrays()
    assert all(np.array_equal(x, y) for x, y in zip(g2, orc.spgemm(A2, A2, mode=1, fx_shift=40)))
    lens = np.diff(A[0])
    assert f == int(lens[A[1]].sum())
    ref0 = orc.spgemm(A, A, mode=0)
    np.testing.assert_allclose(got[2], ref0[2], rtol=2e-6)   # float32-accumulating reference stand-in


def test_spgemm_rectangular_row_block():
    A = random_stochastic(2000, 20, 7)
    d = _lib.DeviceCSR.from_arrays(*A)
    full = _lib.spgemm(d, d, fx_shift=60).to_arrays()
    blk = d.row_block(500, 1300)
    part = _lib.spgemm(blk, d, fx_shift=60).to_arrays()
    lo, hi = full[0][500], full[0][1300]
    assert np.array_equal(part[0], full[0][500:1301] - lo)
    assert np.array_equal(part[1], full[1][lo:hi]) and np.array_equal(part[2], full[2][lo:hi])


def test_inflate_prune_iterations(golden_mcl):
    """inflate -> normalise -> prune -> normalise (+ convergence) per iteration.  The expansion feeding
    each iteration is the reference's own float32 product (oracle mode 0 == scipy bit for bit), so this
    isolates the row-local kernels, whose arithmetic the reference specifies exactly."""
    g = golden_mcl
    for tag in g['cases']:
        infl = float(g[tag + '_inflation'])
        niter = int(g[tag + '_niter'])
        cur = tri(g, tag + '_m2')
        last = None
        for it in range(niter):
            if it:
                cur = orc.spgemm(cur, cur, mode=0)
            c = _lib.DeviceCSR.from_arrays(*cur)
            p = _lib.inflate_prune(c, infl, 1e-4)
            ref = tri(g, '%s_it%d' % (tag, it))
            got = p.to_arrays()
            assert_close_csr(got, ref, RTOL, '%s iter %d' % (tag, it))
            # the stand-alone seams agree with the fused kernel
            c2 = _lib.DeviceCSR.from_arrays(*cur)
            _lib.inflate(c2, infl)
            p2 = _lib.prune(c2, 1e-4).to_arrays()
            assert np.array_equal(p2[0], got[0]) and np.array_equal(p2[1], got[1]) and np.array_equal(p2[2], got[2])
            cur = ref          # re-anchor on the reference so knife-edge threshold flips cannot compound
            if it > 1:
                d = _lib.convergence_stat(_lib.DeviceCSR.from_arrays(*cur), _lib.DeviceCSR.from_arrays(*last))
                assert d == orc.convergence_stat(cur, last)
                assert (d <= np.float32(1e-8)) == (it == niter - 1)
            last = cur


def test_expansion_inside_iterations(golden_mcl):
    """the HIP expansion on the reference's own per-iteration matrices: bit exact against the fixed-point
    specification, and within float32 accumulation noise of the float32-accumulating stand-in (the
    reference's MKL accumulation order is unspecified, so it is itself only reproducible to that noise)."""
    g = golden_mcl
    for tag in g['cases']:
        for it in range(int(g[tag + '_niter']) - 1):
            A = tri(g, '%s_it%d' % (tag, it))
            d = _lib.DeviceCSR.from_arrays(*A)
            got = _lib.spgemm(d, d, fx_shift=60).to_arrays()
            spec = orc.spgemm(A, A, mode=1, fx_shift=60)
            assert all(np.array_equal(x, y) for x, y in zip(got, spec))
            assert_close_csr(got, orc.spgemm(A, A, mode=0), 2e-6, 'expansion %s it %d' % (tag, it))


def clusters_of(att, ptr, mem):
    return {tuple(mem[ptr[a]:ptr[a + 1]].tolist()) for a in range(len(att))}


def test_mcl_whole(golden_mcl):
    g = golden_mcl
    for tag in g['cases']:
        infl = float(g[tag + '_inflation'])
        niter = int(g[tag + '_niter'])
        pre = _lib.DeviceCSR.from_arrays(*tri(g, tag + '_m2'))
        res, n_iter, conv, stats = _lib.mcl(pre, 2, infl, 200, 1e-4, want_stats=True)
        assert conv and n_iter == niter
        assert_close_csr(res.to_arrays(), tri(g, '%s_it%d' % (tag, niter - 1)), 1e-6, str(tag) + ' final')     # north_star: 1e-6 relative
        want = {tuple(g[tag + '_clusters'][g[tag + '_clusters_ptr'][a]:g[tag + '_clusters_ptr'][a + 1]].tolist())
                for a in range(len(g[tag + '_clusters_ptr']) - 1)}
        assert clusters_of(*_lib.interpret(res)) == want
        o = orc.mcl(tri(g, tag + '_m2'), 2, infl, 200, 1e-4, spgemm_mode=1, fx_shift=52, want_stats=True)
        assert np.array_equal(o[5], stats)      # nnz_A, nnz_C, nnz_P, F per iteration
        # not converging is not an error (:2058-2062)
        res2, n2, c2 = _lib.mcl(pre, 2, infl, 3, 1e-4)
        assert n2 == 3 and not c2


def test_mcl_deterministic_and_larger():
    A = clustered_stochastic(20000, 200, 30, 1, 11)
    A2 = orc.spgemm(A, A, mode=1, fx_shift=52)
    pre = _lib.DeviceCSR.from_arrays(*A2)
    r1, n1, c1 = _lib.mcl(pre, 2, 2.0, 60, 1e-4)
    r2, n2, c2 = _lib.mcl(pre, 2, 2.0, 60, 1e-4)
    a1, a2 = r1.to_arrays(), r2.to_arrays()
    assert n1 == n2 and all(np.array_equal(x, y) for x, y in zip(a1, a2)), 'run-to-run bits differ'
    o = orc.mcl(A2, 2, 2.0, 60, 1e-4, spgemm_mode=1, fx_shift=52)
    assert o[3] == n1 and o[4] == c1
    assert clusters_of(*_lib.interpret(r1)) == clusters_of(*orc.interpret(o[:3]))


def table_of(g):
    return orc.FragTable(g['ctg_rank'], g['ctg_len'], g['ctg_frag0'], g['ctg_split'], int(g['bin_size']),
                         g['frag_rank'], g['frag_len'], g['frag_nx'])


@pytest.mark.parametrize('chunk', [None, 4096])
def test_ingest_golden(golden_ingest, chunk):
    g = golden_ingest
    t = table_of(g)
    ing = _lib.Ingest(t, int(g['flank']), bins=bool(g['bins']), expected_keys=0 if chunk else 20000)
    n = len(g['id1'])
    step = chunk or n
    for s in range(0, n, step):
        ing.push(g['id1'][s:s + step], g['pos1'][s:s + step], g['id2'][s:s + step], g['pos2'][s:s + step])
    out = ing.fetch()
    for k in ('full_i', 'full_j', 'full_cnt', 'ht_cnt', 'flank_i', 'flank_j', 'flank_cnt', 'frag_links'):
        assert np.array_equal(out[k], g[k]), k


@pytest.mark.parametrize('chunk', [None, 7001])
def test_ingest_pairs_side_products_golden(golden_ingest, chunk):
    """f2: CLM distances (update_clm_dict :395-401) and first coordinates (record_coord_pairs :454-459) per contig
    pair, in dict insertion order and stream order, single push and several pushes"""
    g = golden_ingest
    t = table_of(g)
    ing = _lib.Ingest(t, int(g['flank']), bins=bool(g['bins']))
    ing.keep_pairs()
    n = len(g['id1'])
    step = chunk or n
    for s in range(0, n, step):
        ing.push(g['id1'][s:s + step], g['pos1'][s:s + step], g['id2'][s:s + step], g['pos2'][s:s + step])
    out = ing.fetch()
    clm_ptr, clm, crd_ptr, crd = ing.fetch_pairs(int(g['max_read_pairs']), out['full_cnt'])
    assert np.array_equal(4 * clm_ptr, g['clm_ptr']) and np.array_equal(clm, g['clm'])
    assert np.array_equal(2 * crd_ptr, g['crd_ptr']) and np.array_equal(crd, g['crd'])
    # HT_link_dict's insertion order (:404-416) from the first stream position of every (contig pair, quadrant)
    first = ing.fetch_ht_order()
    k, q = np.nonzero(out['ht_cnt'])
    order = np.argsort(first[k, q], kind='stable')
    assert np.array_equal(np.stack([k[order], q[order]], 1), g['ht_order'])
    assert (first[out['ht_cnt'] == 0] == np.iinfo(np.int64).max).all()


def _random_bins_case(npairs, seed):
    """contigs above 60 kb split into 25 kb bins, names ranked lexically as the reference's sorted() would"""
    from haphic_amd import synth
    gen = synth.make_genome(3, 1_500_000, 40_000, cv=0.8, min_len=3000, seed=seed)
    names = list(gen.names)
    bin_size = 25_000
    frag_names, frag0, split, frag_len = [], [], [], []
    for nm, ln in zip(names, gen.length.tolist()):
        frag0.append(len(frag_names))
        if ln > 60_000:
            nb = -(-ln // bin_size)
            split.append(1)
            frag_names += ['{}_bin{}'.format(nm, k + 1) for k in range(nb)]
            frag_len += [bin_size] * (nb - 1) + [ln - bin_size * (nb - 1)]
        else:
            split.append(0)
            frag_names.append(nm)
            frag_len.append(ln)

    def rank(lst):
        order = sorted(range(len(lst)), key=lst.__getitem__)
        r = np.empty(len(lst), np.int32)
        r[order] = np.arange(len(lst), dtype=np.int32)
        return r
    t = orc.FragTable(rank(names), gen.length, np.array(frag0, np.int32), np.array(split, np.uint8), bin_size,
                      rank(frag_names), np.array(frag_len, np.int64), (np.arange(len(frag_names)) % 5 != 0).astype(np.uint8))
    id1, p1, id2, p2 = [x.numpy() for x in synth.sample_pairs(gen, npairs, seed=seed + 1, cis=0.9)]
    return t, id1.astype(np.int32), p1.astype(np.int64), id2.astype(np.int32), p2.astype(np.int64)


@pytest.mark.parametrize('chunk', [None, 7001])
def test_ingest_frag_pairs_golden(chunk):
    """ctg_pair_to_frag (:1731-1733): the distinct fragment pairs, single push and several pushes; the other
    tables are unaffected by the extra stream"""
    from tests.conftest import load_golden
    g = load_golden('ingest_bins.npz')
    ing = _lib.Ingest(table_of(g), int(g['flank']), bins=True)
    ing.keep_frag_pairs()
    n = len(g['id1'])
    step = chunk or n
    for s in range(0, n, step):
        ing.push(g['id1'][s:s + step], g['pos1'][s:s + step], g['id2'][s:s + step], g['pos2'][s:s + step])
    out = ing.fetch()
    for k in ('full_i', 'full_j', 'full_cnt', 'ht_cnt', 'flank_i', 'flank_j', 'flank_cnt', 'frag_links'):
        assert np.array_equal(out[k], g[k]), k
    fi, fj = ing.fetch_frag_pairs()
    got = np.unique(np.stack([fi, fj], 1), axis=0)
    assert len(got) == len(fi)                              # distinct
    assert np.array_equal(got, np.unique(g['c2f'][:, 2:], axis=0))
    # larger random stream against the oracle's restatement
    t, id1, p1, id2, p2 = _random_bins_case(40000, seed=77)
    ing = _lib.Ingest(t, 3000, bins=True)
    ing.keep_frag_pairs()
    ing.push(id1, p1, id2, p2)
    ing.fetch()
    fi, fj = ing.fetch_frag_pairs()
    wi, wj = orc.frag_pairs(t, id1, p1, id2, p2)
    assert np.array_equal(np.unique(np.stack([fi, fj], 1), axis=0), np.stack([wi, wj], 1)) and len(fi) == len(wi)


def test_pairs_text_tokeniser_golden():
    """a1 on the device: line k of the chunk -> row k (skipped lines and unknown names -1), BED bytes identical to
    the reference's alignments.bed; whole text and cut into chunks of whole lines"""
    from tests.conftest import load_golden
    g = load_golden('pairs_text.npz')
    names = [str(x) for x in g['names']]
    raw = g['text'].tobytes()
    w1, wp1, w2, wp2, wbed = orc.parse_pairs_text(raw, names)
    ps = _lib.PairsParser(names)
    assert ps.parse(raw, want_bed=True) == len(w1)
    i1, p1, i2, p2, bed = ps.fetch(want_bed=True)
    assert all(np.array_equal(a, b) for a, b in zip((i1, p1, i2, p2), (w1, wp1, w2, wp2)))
    assert bed == g['bed_all'].tobytes() == wbed
    # chunks: cut after a '\n' every ~5 KB
    got, beds, at = [], [], 0
    while at < len(raw):
        cut = raw.rfind(b'\n', at, at + 5000) + 1
        if cut <= at:
            cut = len(raw)
        ps.parse(raw[at:cut], want_bed=True)
        out = ps.fetch(want_bed=True)
        got.append(np.stack(out[:4], 1))
        beds.append(out[4])
        at = cut
    assert b''.join(beds) == wbed
    rows = np.concatenate(got)
    live = ~((rows[:, 0] == -1) & (rows[:, 2] == -1) & (rows[:, 1] == 0) & (rows[:, 3] == 0))
    assert np.array_equal(rows[live], g['all'])            # a lone '\r' before a cut is a line end in both
    # no BED wanted, empty input, a text without any line end
    assert ps.parse(b'') == 0
    assert ps.parse(b'r1 ctg1 5 ctg2 9') == 1
    assert [a.tolist() for a in ps.fetch()[:4]] == [[1], [4], [2], [8]]
    # a line far longer than the LDS window (40 kB read name) among ordinary ones: the HBM reader takes its block
    long_text = b'r0\tctg2\t11\tctg3\t12\n' + b'R' * 40000 + b'\tctg5\t100\tctg7\t200\textra\n' + b'r2\tctg1\t5\tctg4\t6\n'
    assert ps.parse(long_text, want_bed=True) == 3
    got = ps.fetch(want_bed=True)
    want_long = orc.parse_pairs_text(long_text, names)
    assert all(np.array_equal(x, y) for x, y in zip(got[:4], want_long[:4])) and got[4] == want_long[4]
    # malformed lines raise what the reference raises
    with pytest.raises(IndexError):
        ps.parse(b'r1\tctg1\t5\tctg2\n')
    with pytest.raises(ValueError):
        ps.parse(b'r1\tctg1\t5x\tctg2\t7\n')
    with pytest.raises(ValueError):
        ps.parse(b'#h\n\nr1\tctg1\t5\tctg2\t7_\n')
    ps.destroy()


def test_pairs_text_through_ingest(tmp_path, monkeypatch):
    """.pairs file -> cluster.pairs_generator_inter_ctgs -> cluster.parse_alignments_for_ctgs: the dicts equal the
    oracle's on the tuples the reference generator yielded; alignments.bed byte-identical; plain and gzipped,
    one chunk and many"""
    import gzip
    from haphic_amd import cluster
    from tests.conftest import load_golden
    g = load_golden('pairs_text.npz')
    names = [str(x) for x in g['names']]
    raw = g['text'].tobytes()
    (tmp_path / 'in.pairs').write_bytes(raw)
    with gzip.open(tmp_path / 'in.pairs.gz', 'wb') as f:
        f.write(raw)
    from tests import bam_fixture as bf
    (tmp_path / 'in.pairs.bgz').write_bytes(b''.join(bf.bgzf_block(raw[a:a + 3001]) for a in range(0, len(raw), 3001)) + bf.EOF_BLOCK)      # bgzip's container
    monkeypatch.chdir(tmp_path)
    rng = np.random.default_rng(5)
    lens = rng.integers(3_000_000, 6_000_000, len(names))
    fa_dict = {n: [None, int(l), 0] for n, l in zip(names, lens)}
    frag_len_dict = {n: fa_dict[n][1] for n in names}
    order = sorted(range(len(names)), key=names.__getitem__)
    rank = np.empty(len(names), np.int32)
    rank[order] = np.arange(len(names), dtype=np.int32)
    t = orc.FragTable(rank, lens, np.arange(len(names), dtype=np.int32), np.zeros(len(names), np.uint8), 0, rank, lens,
                      np.ones(len(names), np.uint8))
    ok = g['inter'][:, 1] >= 0                              # the oracle takes 0-based positions like the generator's
    want = orc.ingest(t, *[g['inter'][:, c] for c in range(4)], 500_000)

    class A:
        flank = 500
        remove_allelic_links = 0
        remove_concentrated_links = False
        max_read_pairs = 200
        nwindows = 50
    for fname, fmt, chunk in (('in.pairs', 'pairs', 256 << 20), ('in.pairs', 'pairs', 3000), ('in.pairs.gz', 'bgzipped_pairs', 10000),
                              ('in.pairs.bgz', 'bgzipped_pairs', 256 << 20), ('in.pairs.bgz', 'bgzipped_pairs', 70_000)):
        aln = cluster.pairs_generator_inter_ctgs(fname, fmt)
        aln.chunk_bytes = chunk
        full, flank, HT, clm, frag_link, coord = cluster.parse_alignments_for_ctgs(aln, fa_dict, A(), frag_len_dict, set(names),
                                                                                   'int32', 'int32')
        assert list(full.items()) == [((names[i], names[j]), c) for i, j, c in zip(want['full_i'], want['full_j'], want['full_cnt'].tolist())]
        assert list(flank.items()) == [((names[i], names[j]), c) for i, j, c in zip(want['flank_i'], want['flank_j'], want['flank_cnt'].tolist())]
        cluster._lib.files_join()                               # alignments.bed leaves HBM through the library's file-writer thread
        assert (tmp_path / 'alignments.bed').read_bytes() == g['bed_all'].tobytes()
    assert ok.any()


def test_text_reader_chunks_are_the_file(tmp_path):
    """hhx_text_reader (the native front of a1): the chunks, concatenated, are the file; every chunk but the last ends on a line break; sizes around
    the chunk size, files without a final newline, '\r\n' and lone '\r' endings, a line longer than a chunk, the empty file"""
    import ctypes
    rng = np.random.default_rng(11)
    for case, (n_lines, chunk, ending, final_newline) in enumerate([(0, 4096, b'\n', True), (1, 4096, b'\n', False), (5000, 4096, b'\n', True), (5000, 10_000, b'\r\n', False),
                                                                     (3000, 65_536, b'\r', True), (200_000, 1 << 20, b'\n', True), (70_000, 300_000, b'\n', False)]):
        lines = [b'r%d\tctg%d\t%d\tctg%d\t%d\t+\t-' % (k, rng.integers(0, 1000), rng.integers(1, 10**7), rng.integers(0, 1000), rng.integers(1, 10**7)) for k in range(n_lines)]
        if case == 5:
            lines[777] = b'#' + b'x' * 1_500_000                  # a comment line longer than the chunk: the chunk grows (up to twice its size)
        text = ending.join(lines) + (ending if final_newline and lines else b'')
        path = tmp_path / ('t%d.txt' % case)
        path.write_bytes(text)
        reader = _lib.TextReader(str(path), chunk, threads=3)
        got = []
        for host, n in reader:
            piece = ctypes.string_at(host, n)
            got.append(piece)
        reader.close()
        assert b''.join(got) == text, case
        assert all(p[-1:] in (b'\n', b'\r') for p in got[:-1]) and all(len(p) <= 2 * chunk + 4096 for p in got), case
        assert len(got) >= len(text) // (2 * chunk + 4096), case
    # bgzipped text (BGZF: SAM specification 4.1): blocks of any payload size up to 64 KiB, empty blocks, the EOF marker; inflated by the reader's threads
    from tests import bam_fixture as bf
    import gzip
    for case, (n_lines, chunk, payload_max, eof_block) in enumerate([(0, 4096, 100, True), (3, 4096, 7, False), (20_000, 100_000, 65_000, True), (150_000, 1 << 20, 30_000, True),
                                                                      (40_000, 50_000, 500, False)]):
        lines = [b'r%d\tctg%d\t%d\tctg%d\t%d\t+\t-' % (k, rng.integers(0, 1000), rng.integers(1, 10**7), rng.integers(0, 1000), rng.integers(1, 10**7)) for k in range(n_lines)]
        text = b'\n'.join(lines) + (b'\n' if lines else b'')
        blocks, at = [], 0
        while at < len(text):
            n = int(rng.integers(1, payload_max + 1))
            blocks.append(bf.bgzf_block(text[at:at + n]))
            if rng.random() < 0.02:
                blocks.append(bf.bgzf_block(b''))
            at += n
        data = b''.join(blocks) + (bf.EOF_BLOCK if eof_block else b'')
        path = tmp_path / ('z%d.txt.gz' % case)
        path.write_bytes(data)
        assert gzip.decompress(data) == text if data else True
        if not data:
            continue
        reader = _lib.TextReader(str(path), chunk, threads=5, bgzf=True)
        got = [ctypes.string_at(host, n) for host, n in reader]
        reader.close()
        assert b''.join(got) == text, ('bgzf', case)
        assert all(p[-1:] == b'\n' for p in got[:-1]), ('bgzf', case)
    plain = tmp_path / 'plain.gz'
    plain.write_bytes(gzip.compress(b'r1\tctg1\t5\tctg2\t9\t+\t-\n' * 100))
    with pytest.raises(RuntimeError, match='not a BGZF file'):
        _lib.TextReader(str(plain), 4096, bgzf=True)
    cut = tmp_path / 'cut.gz'
    cut.write_bytes(data[:len(data) // 2])                             # a file that ends inside a block
    reader = _lib.TextReader(str(cut), 1 << 20, bgzf=True)
    with pytest.raises(RuntimeError, match='BGZF'):
        list(reader)
    reader.close()
    with pytest.raises(RuntimeError, match='cannot open'):
        _lib.TextReader(str(tmp_path / 'absent.txt'))
    path = tmp_path / 'long.txt'
    path.write_bytes(b'y' * 50_000 + b'\n')
    reader = _lib.TextReader(str(path), 4096)
    with pytest.raises(RuntimeError, match='longer than a chunk'):
        list(reader)
    reader.close()


@pytest.mark.parametrize('seed,flank,npairs,chunk', [(101, 0, 5000, None), (102, 1000, 60000, 9973), (103, 30000, 200000, None),
                                                      (104, 7000, 1, None), (105, 12000, 300000, 65536)])
def test_ingest_bins_randomised_vs_oracle(seed, flank, npairs, chunk):
    """split contigs, random flank widths (0 = every position is flank :299-307), single pairs, ragged pushes: every
    table, the CLM distances, the first coordinates and the fused link matrix against the oracle"""
    t, id1, p1, id2, p2 = _random_bins_case(npairs, seed)
    rng = np.random.default_rng(seed)
    id1 = id1.copy()
    id1[rng.random(npairs) < 0.02] = -1                      # names that are not in the FASTA
    want = orc.ingest(t, id1, p1, id2, p2, flank, bins=True, want_clm=True, max_read_pairs=7)
    ing = _lib.Ingest(t, flank, bins=True)
    ing.keep_pairs()
    step = chunk or npairs
    for s in range(0, npairs, step):
        ing.push(id1[s:s + step], p1[s:s + step], id2[s:s + step], p2[s:s + step])
    out = ing.fetch()
    for k in ('full_i', 'full_j', 'full_cnt', 'ht_cnt', 'flank_i', 'flank_j', 'flank_cnt', 'frag_links'):
        assert np.array_equal(out[k], want[k]), k
    clm_ptr, clm, crd_ptr, crd = ing.fetch_pairs(7, out['full_cnt'])
    assert np.array_equal(4 * clm_ptr, want['clm_ptr']) and np.array_equal(clm, want['clm'])
    assert np.array_equal(2 * crd_ptr, want['crd_ptr']) and np.array_equal(crd, want['crd'])
    in_set = t.frag_nx.copy()
    in_set[rng.random(len(in_set)) < 0.1] = 0                # a filtered fragment set (filter_fragments' output)
    m, fidx, n_linked = ing.link_matrix(in_set)
    ok = in_set[want['flank_i']].astype(bool) & in_set[want['flank_j']].astype(bool)
    linked = np.zeros(len(in_set), bool)
    linked[want['flank_i'][ok]] = True
    linked[want['flank_j'][ok]] = True
    n_rest = int(in_set.sum() - linked.sum())
    rp, rj, rx, ridx, rl = orc.dict_to_matrix(want['flank_i'], want['flank_j'], want['flank_cnt'].astype(np.float64), len(in_set),
                                              in_set, n_rest)
    assert n_linked == rl and np.array_equal(fidx, ridx)
    assert all(np.array_equal(a, b) for a, b in zip(m.to_arrays(), (rp, rj, rx)))
    m.free()
    ing.destroy()


def test_dict_to_matrix_golden(golden_ingest):
    g = golden_ingest
    in_set = g['d2m_in_set']
    ok = in_set[g['flank_i']].astype(bool) & in_set[g['flank_j']].astype(bool)
    linked = np.zeros(len(in_set), bool)
    linked[g['flank_i'][ok]] = True
    linked[g['flank_j'][ok]] = True
    n_rest = int(in_set.sum() - linked.sum())
    m, fidx, n_linked = _lib.dict_to_matrix(g['flank_i'], g['flank_j'], g['flank_cnt'].astype(np.float64),
                                            len(in_set), in_set, n_rest)
    p, j, x = m.to_arrays()
    assert np.array_equal(p, g['d2m_p']) and np.array_equal(j, g['d2m_j']) and np.array_equal(x, g['d2m_x'])
    assert np.array_equal(fidx[linked], g['d2m_frag_index'][linked]) and (fidx[~linked] == -1).all()
    fl = g['frag_links'].astype(np.float64)
    val = g['flank_cnt'] / (fl[g['flank_i']] * fl[g['flank_j']]) ** 0.5
    m2, _, _ = _lib.dict_to_matrix(g['flank_i'], g['flank_j'], val, len(in_set), in_set, n_rest)
    assert np.array_equal(m2.to_arrays()[2], g['d2m_nlinks_x'])


def test_ingest_random_vs_oracle_and_device_path():
    import torch
    from haphic_amd import synth
    gen = synth.make_genome(6, 3_000_000, 20_000, seed=3)
    n = gen.n
    t = orc.FragTable(gen.lexical_rank(), gen.length, np.arange(n, dtype=np.int32), np.zeros(n, np.uint8), 0,
                      gen.lexical_rank(), gen.length, (np.arange(n) % 7 != 0).astype(np.uint8))
    id1, p1, id2, p2 = synth.sample_pairs(gen, 2_000_000, seed=9, device='cuda')
    ing = _lib.Ingest(t, 5000, bins=False, skip_intra=True, expected_keys=1 << 21)
    ing.push_device(id1.numel(), id1.data_ptr(), p1.data_ptr(), id2.data_ptr(), p2.data_ptr())
    torch.cuda.synchronize()
    out = ing.fetch()
    h = [a.cpu().numpy() for a in (id1, p1, id2, p2)]
    keep = h[0] != h[2]
    ref = orc.ingest(t, h[0][keep], h[1][keep].astype(np.int64), h[2][keep], h[3][keep].astype(np.int64), 5000)
    for k in ('full_i', 'full_j', 'full_cnt', 'ht_cnt', 'flank_i', 'flank_j', 'flank_cnt', 'frag_links'):
        assert np.array_equal(out[k], ref[k]), k
    # device-resident flank table -> dict_to_matrix without a host round trip
    in_set = t.frag_nx.copy()
    linked = np.zeros(n, bool)
    linked[ref['flank_i']] = True
    linked[ref['flank_j']] = True
    n_rest = int(in_set.sum() - (linked & in_set.astype(bool)).sum())
    fi, fj, fv = ing.flank_device()
    m, fidx, n_linked = _lib.dict_to_matrix(fi, fj, fv, n, in_set, n_rest, on_device=True, n_keys=ing.n_flank)
    rp, rj, rx, ridx, rl = orc.dict_to_matrix(ref['flank_i'], ref['flank_j'], ref['flank_cnt'].astype(np.float64), n,
                                              in_set, n_rest)
    p, j, x = m.to_arrays()
    assert n_linked == rl and np.array_equal(fidx, ridx)
    assert np.array_equal(p, rp) and np.array_equal(j, rj) and np.array_equal(x, rx)
    # fused path: straight from the unordered aggregated table + first-seen ordinals
    m2, fidx2, nl2 = ing.link_matrix(in_set, n_rest)
    assert nl2 == rl and np.array_equal(fidx2, ridx)
    assert all(np.array_equal(a, b) for a, b in zip(m2.to_arrays(), (rp, rj, rx)))
    m3, _, nl3 = ing.link_matrix(in_set)                       # n_rest < 0: computed from in_set
    assert nl3 == rl and all(np.array_equal(a, b) for a, b in zip(m3.to_arrays(), (rp, rj, rx)))


def test_ingest_table_exchange_roundtrip():
    """two handles ingest the two halves of a stream (second one with the ordinal base of its chunk); pushing
    both aggregated tables into a third handle reproduces the tables of the whole stream — the multi-GPU
    exchange step on one GPU"""
    import torch
    from haphic_amd import synth
    gen = synth.make_genome(5, 1_000_000, 15_000, seed=4)
    n = gen.n
    t = orc.FragTable(gen.lexical_rank(), gen.length, np.arange(n, dtype=np.int32), np.zeros(n, np.uint8), 0,
                      gen.lexical_rank(), gen.length, np.ones(n, np.uint8))
    id1, p1, id2, p2 = synth.sample_pairs(gen, 300_000, seed=5, device='cuda')
    half = 170_001
    parts = []
    for lo, hi in ((0, half), (half, 300_000)):
        ing = _lib.Ingest(t, 4000, bins=False, skip_intra=True)
        ing.set_ordinal_base(lo)
        a = [x[lo:hi].contiguous() for x in (id1, p1, id2, p2)]
        ing.push_device(hi - lo, *[x.data_ptr() for x in a])
        torch.cuda.synchronize()
        ing.finalize()
        parts.append(ing)
    merged = _lib.Ingest(t, 4000, bins=False, skip_intra=True)
    for ing in parts:
        nrows, *ptrs = ing.table_device(0)
        merged.push_table(0, nrows, *ptrs)
    got = merged.fetch()
    whole = _lib.Ingest(t, 4000, bins=False, skip_intra=True)
    whole.push_device(300_000, id1.data_ptr(), p1.data_ptr(), id2.data_ptr(), p2.data_ptr())
    torch.cuda.synchronize()
    ref = whole.fetch()
    for k in ('full_i', 'full_j', 'full_cnt', 'ht_cnt', 'flank_i', 'flank_j', 'flank_cnt', 'frag_links'):
        assert np.array_equal(got[k], ref[k]), k
    h = [a.cpu().numpy() for a in (id1, p1, id2, p2)]
    keep = h[0] != h[2]
    o = orc.ingest(t, h[0][keep], h[1][keep].astype(np.int64), h[2][keep], h[3][keep].astype(np.int64), 4000)
    for k in ('full_i', 'full_j', 'full_cnt', 'ht_cnt', 'flank_i', 'flank_j', 'flank_cnt', 'frag_links'):
        assert np.array_equal(got[k], o[k]), k


def oracle_fused(A, B, infl, pruning=1e-4, shift=52):
    c = orc.spgemm(A, B, n_cols=len(B[0]) - 1, mode=1, fx_shift=shift)
    x = orc.normalize_l1(c[0], orc.power(c[2], infl))
    return orc.prune((c[0], c[1], x), pruning), int(c[0][-1])


@pytest.mark.parametrize('hash_max', [0, 4_000_000])          # 0: the window / compact / tiny classes alone; default: rows try the LDS hash table first
@pytest.mark.parametrize('n,block,deg_in,deg_out,infl', [
    (3000, 100, 20, 1, 2.0),        # window mode, one window
    (40000, 400, 60, 1, 2.0),       # window mode, several column windows (n_cols > LDS capacity)
    (40000, 50, 4, 0, 2.0),         # compact (bitmap-rank) mode
    (5000, 100, 10, 2, 1.4),        # pow() path
    (70000, 10, 2, 1, 3.0),         # tiny rows
    (8000, 8000, 40, 0, 2.0),       # ~4500 distinct columns per row: leaves the hash class for the window class
    (20000, 10000, 40, 0, 2.0),     # the same, below the window class's product count: leaves it for the compact class
])
def test_fused_expand_inflate_prune(n, block, deg_in, deg_out, infl, hash_max):
    A = clustered_stochastic(n, block, deg_in, deg_out, 31)
    d = _lib.DeviceCSR.from_arrays(*A)
    _lib.tune('hash_max', hash_max)
    try:
        p, f, nnz_c = _lib.expand_inflate_prune(d, d, infl, 1e-4)
        ref, ref_nnz_c = oracle_fused(A, A, infl)
        got = p.to_arrays()
        assert nnz_c == ref_nnz_c
        assert f == int(np.diff(A[0])[A[1]].sum())
        assert_close_csr(got, ref, RTOL, 'fused iteration')
        # row-block (multi-GPU shard) gives the same rows
        r0, r1 = n // 3, n // 3 + n // 5
        pb, _, _ = _lib.expand_inflate_prune(d.row_block(r0, r1), d, infl, 1e-4)
        gb = pb.to_arrays()
        lo, hi = got[0][r0], got[0][r1]
        assert np.array_equal(gb[0], got[0][r0:r1 + 1] - lo)
        assert np.array_equal(gb[1], got[1][lo:hi]) and np.array_equal(gb[2], got[2][lo:hi])
        # deterministic, and the same bits whichever class took the rows
        p2, _, _ = _lib.expand_inflate_prune(d, d, infl, 1e-4)
        assert all(np.array_equal(x, y) for x, y in zip(p2.to_arrays(), got))
        _lib.tune('hash_max', 0 if hash_max else 4_000_000)
        p3, _, _ = _lib.expand_inflate_prune(d, d, infl, 1e-4)
        assert all(np.array_equal(x, y) for x, y in zip(p3.to_arrays(), got))
    finally:
        _lib.tune('hash_max', None)


def test_mcl_normalized_fuses_pre_expansion(golden_mcl):
    g = golden_mcl
    for tag in g['cases']:
        infl = float(g[tag + '_inflation'])
        niter = int(g[tag + '_niter'])
        norm = _lib.DeviceCSR.from_arrays(*tri(g, tag + '_norm'))
        res, n_iter, conv, stats = _lib.mcl(norm, 2, infl, 200, 1e-4, want_stats=True, normalized=True)
        assert conv and n_iter == niter
        assert_close_csr(res.to_arrays(), tri(g, '%s_it%d' % (tag, niter - 1)), 1e-6, str(tag) + ' final')     # north_star: 1e-6 relative
        want = {tuple(g[tag + '_clusters'][g[tag + '_clusters_ptr'][a]:g[tag + '_clusters_ptr'][a + 1]].tolist())
                for a in range(len(g[tag + '_clusters_ptr']) - 1)}
        assert clusters_of(*_lib.interpret(res)) == want
        assert stats[0, 0] == len(g[tag + '_norm_j']) and stats[0, 1] == len(g[tag + '_m2_j']) and stats[0, 3] > 0


def test_mcl_blocked_pre_expansion_resume(golden_mcl):
    """M^2 as row blocks (hhx_spgemm on hhx_csr_row_block), iteration 0 per block without touching the blocks
    (hhx_inflate_prune_keep), hhx_csr_vstack, hhx_mcl_resume == hhx_mcl on the whole pre-expanded matrix: same
    matrix bit for bit, same iteration count and convergence flag, for two inflations from the same blocks"""
    A = clustered_stochastic(6000, 100, 12, 1, 7)
    d = _lib.DeviceCSR.from_arrays(*A)
    pre = _lib.spgemm(d, d, fx_shift=52)
    n = d.shape3[0]
    blocks = []
    for r0 in range(0, n, 1300):
        a = d.row_block(r0, min(n, r0 + 1300))
        blocks.append(_lib.spgemm(a, d, fx_shift=52))
        a.free()
    stacked = _lib.vstack(blocks)
    assert all(np.array_equal(x, y) for x, y in zip(stacked.to_arrays(), pre.to_arrays()))
    keep = [b.to_arrays()[2].copy() for b in blocks]
    for infl in (2.0, 1.4):
        want, it_w, cv_w = _lib.mcl(pre, 2, infl, 100, 1e-4)
        parts = [_lib.inflate_prune_keep(b, infl, 1e-4) for b in blocks]
        first = _lib.vstack(parts)
        got, it_g, cv_g = _lib.mcl_resume(first, 1, 2, infl, 100, 1e-4)
        assert (it_g, cv_g) == (it_w, cv_w)
        assert all(np.array_equal(x, y) for x, y in zip(got.to_arrays(), want.to_arrays()))
        one, it_1, cv_1 = _lib.mcl(pre, 2, infl, 1, 1e-4)             # a single iteration == the stacked iteration 0
        assert all(np.array_equal(x, y) for x, y in zip(first.to_arrays(), one.to_arrays())) and (it_1, cv_1) == (1, False)
        for x in parts + [first, got, want, one]:
            x.free()
    assert all(np.array_equal(b.to_arrays()[2], k) for b, k in zip(blocks, keep))       # the blocks were not written
    free_b, total_b = _lib.mem_info()
    assert 0 < free_b <= total_b


def test_mcl_links_class_stream_iteration0():
    """hhx_mcl_links (normalisation fused, iteration 0 streaming the link matrix with the count-1 entries of every
    segment moved to the front: 16-bit window-local columns only for those, column + value for the rest) must give
    exactly the bits of hhx_normalize_l1 + hhx_mcl_normalized, on a matrix wide enough for several column windows,
    for every tile shape of the kernel; non-integer values must fall back to the generic stream."""
    import torch
    from haphic_amd import synth
    gen = synth.make_genome(8, 60_000_000, 20_000, seed=8)          # ~24k contigs -> 2 column windows
    n = gen.n
    lex = gen.lexical_rank()
    t = orc.FragTable(lex, gen.length, np.arange(n, dtype=np.int32), np.zeros(n, np.uint8), 0, lex, gen.length,
                      np.ones(n, np.uint8))
    id1, p1, id2, p2 = synth.sample_pairs(gen, 20_000_000, seed=9, device='cuda')
    ing = _lib.Ingest(t, 500_000, bins=False, skip_intra=True)
    ing.push_device(id1.numel(), id1.data_ptr(), p1.data_ptr(), id2.data_ptr(), p2.data_ptr())
    torch.cuda.synchronize()
    links, fidx, n_linked = ing.link_matrix(np.ones(n, np.uint8))
    assert links.shape3[0] > 17_500
    r1, n1, c1, st1 = _lib.mcl(links, 2, 2.0, 200, 1e-4, want_stats=True, links=True)      # default: integer arithmetic, symmetric half
    mp, mj, mx = links.to_arrays()
    assert orc.links_shift((mp, mj, mx)) > 0
    one_int = _lib.mcl(links, 2, 2.0, 1, 1e-4, links=True)[0].to_arrays()
    # iteration 0 against the oracle's integer specification on a slice of rows, through the row-block entry point of the
    # multi-GPU driver (no symmetry: all products) — and those rows of the symmetric one-GPU result are the same bits
    c = orc.expand_links((mp, mj, mx), rows=np.arange(100, 140))
    ref = orc.prune((c[0], c[1], orc.normalize_l1(c[0], orc.power(c[2], 2.0))), 1e-4)
    got, f_blk, z_blk = _lib.expand_links(links, 100, 140, 2.0, 1e-4)
    ga = got.to_arrays()
    got.free()
    assert all(np.array_equal(x, y) for x, y in zip(ga, ref)), 'integer iteration 0 (row block) vs oracle'
    lo, hi = one_int[0][100], one_int[0][140]
    assert np.array_equal(one_int[1][lo:hi], ref[1]) and np.array_equal(one_int[2][lo:hi], ref[2]), 'integer iteration 0 (symmetric) vs oracle'
    try:
        for sym, tile_u in ((0, 0), (1, 1), (1, 2), (0, 3), (1, 4), (1, 8)):          # without the symmetry; every explicit tile shape
            _lib.tune('links_sym', sym)
            _lib.tune('tile_u', tile_u)
            one = _lib.mcl(links, 2, 2.0, 1, 1e-4, links=True)[0].to_arrays()
            assert all(np.array_equal(x, y) for x, y in zip(one, one_int)), 'integer iteration 0 changed bits: %r' % ((sym, tile_u),)
        _lib.tune('links_sym', 1)
        _lib.tune('tile_u', 0)
        _lib.tune('links_integer', 0)                                # the float arithmetic: class stream, then generic stream
        _lib.tune('cls', 0)                                          # generic (column, value) stream
        r0, n0, c0 = _lib.mcl(links, 2, 2.0, 200, 1e-4, links=True)
        one0 = _lib.mcl(links, 2, 2.0, 1, 1e-4, links=True)[0].to_arrays()
        for cls, nc, tile_u in ((1, 1, 0), (1, 3, 0), (1, 2, 0), (1, 3, 1), (1, 3, 2), (1, 1, 3), (1, 3, 4), (1, 2, 8), (0, 3, 1), (0, 3, 2), (0, 3, 3),
                                (0, 3, 4)):
            _lib.tune('cls', cls)
            _lib.tune('cls_nc', nc)
            _lib.tune('tile_u', tile_u)
            one = _lib.mcl(links, 2, 2.0, 1, 1e-4, links=True)[0].to_arrays()
            assert all(np.array_equal(x, y) for x, y in zip(one, one0)), 'stream layout changed bits: %r' % ((cls, nc, tile_u),)
    finally:
        _lib.tune('cls', 1)
        _lib.tune('cls_nc', 1)
        _lib.tune('tile_u', 0)
        _lib.tune('links_integer', 1)
        _lib.tune('links_sym', 1)
    # integer vs float arithmetic: float32 round-off apart after one iteration, the same clusters at the end
    assert np.array_equal(one_int[0], one0[0]) and np.array_equal(one_int[1], one0[1])
    np.testing.assert_allclose(one_int[2], one0[2], rtol=2e-6, atol=0)
    assert clusters_of(*_lib.interpret(r1)) == clusters_of(*_lib.interpret(r0))
    norm = links.copy()
    _lib.normalize_l1(norm)
    r2, n2, c2, st2 = _lib.mcl(norm, 2, 2.0, 200, 1e-4, want_stats=True, normalized=True)
    assert (n0, c0) == (n2, c2)
    assert all(np.array_equal(x, y) for x, y in zip(r0.to_arrays(), r2.to_arrays())), 'class-stream iteration 0 (float arithmetic) changed bits'
    assert np.array_equal(st1[0, [0, 1, 3]], st2[0, [0, 1, 3]])          # entries, nnz of M^2, products of iteration 0
    # one fused iteration against the oracle on a slice of rows (the oracle is too slow for all of them)
    A = norm.to_arrays()
    blk = norm.row_block(100, 140)
    got = _lib.expand_inflate_prune(blk, norm, 2.0, 1e-4)[0].to_arrays()
    lo, hi = A[0][100], A[0][140]
    ref, _ = oracle_fused(((A[0][100:141] - lo).astype(np.int32), A[1][lo:hi], A[2][lo:hi]), A, 2.0)
    assert_close_csr(got, ref, RTOL, 'windowed fused iteration vs oracle')
    # values that are not integer counts: generic stream, same answer as the two-call path
    p, j, x = links.to_arrays()
    x2 = (x * np.float32(0.37)).astype(np.float32)
    l2 = _lib.DeviceCSR.from_arrays(p, j, x2)
    r3, n3, c3 = _lib.mcl(l2, 2, 2.0, 30, 1e-4, links=True)
    nn = l2.copy()
    _lib.normalize_l1(nn)
    r4, n4, c4 = _lib.mcl(nn, 2, 2.0, 30, 1e-4, normalized=True)
    assert (n3, c3) == (n4, c4) and all(np.array_equal(x, y) for x, y in zip(r3.to_arrays(), r4.to_arrays()))


def test_integer_arithmetic_applicability():
    """hhx_mcl_links takes the integer arithmetic only where its specification applies — symmetric integer counts, row sums up to
    2^18 — and the float arithmetic (bit-identical to hhx_normalize_l1 + hhx_mcl_normalized) everywhere else; the device's
    decision is the oracle's (orc.links_shift)."""
    import scipy.sparse as sp
    rng = np.random.default_rng(12)
    n = 3000
    a = sp.random(n, n, 0.01, random_state=5, format='coo')
    cnt = rng.geometric(0.4, size=a.nnz).astype(np.float32)
    base = sp.coo_matrix((cnt, (a.row, a.col)), shape=(n, n)).tocsr()
    sym = (base + base.T + sp.identity(n, dtype=np.float32, format='csr')).tocsr()
    sym.sort_indices()

    def triple(m):
        m = m.tocsr()
        m.sort_indices()
        return m.indptr.astype(np.int32), m.indices.astype(np.int32), m.data.astype(np.float32)

    def links_equal_normalized(L):
        d = _lib.DeviceCSR.from_arrays(*L)
        r1, n1, c1 = _lib.mcl(d, 2, 2.0, 60, 1e-4, links=True)
        nn = d.copy()
        _lib.normalize_l1(nn)
        r2, n2, c2 = _lib.mcl(nn, 2, 2.0, 60, 1e-4, normalized=True)
        same = (n1, c1) == (n2, c2) and all(np.array_equal(x, y) for x, y in zip(r1.to_arrays(), r2.to_arrays()))
        for x in (r1, r2, nn, d):
            x.free()
        return same
    cases = {}
    cases['symmetric'] = triple(sym)
    asym = sym.tolil()
    asym[5, 9] = asym[5, 9] + 3                                  # one count differs from its mirror image
    cases['asymmetric'] = triple(asym.tocsr())
    heavy = sym.tolil()
    for k in range(1, 7):                                        # a row sum beyond 2^18 (symmetric, counts <= 65535)
        heavy[0, k] = 60000
        heavy[k, 0] = 60000
    cases['heavy row'] = triple(heavy.tocsr())
    for name, L in cases.items():
        d = _lib.DeviceCSR.from_arrays(*L)
        ok = _lib.links_integer_ok(d)
        d.free()
        assert ok == (name == 'symmetric'), name
        if name != 'asymmetric':                                 # (the oracle's rule looks at the values only; symmetry is the caller's)
            assert (orc.links_shift(L) > 0) == ok, name
        if not ok:
            assert links_equal_normalized(L), name + ': the float arithmetic must be the two-call path, bit for bit'
    # where it applies: the oracle's integer pre-expansion, bit for bit, through the whole mcl()
    L = cases['symmetric']
    d = _lib.DeviceCSR.from_arrays(*L)
    r, n_it, cv = _lib.mcl(d, 2, 2.0, 60, 1e-4, links=True)
    o = orc.mcl(orc.expand_links(L), 2, 2.0, 60, 1e-4, spgemm_mode=1, fx_shift=52)
    assert (n_it, cv) == (o[3], o[4]) and all(np.array_equal(x, y) for x, y in zip(r.to_arrays(), o[:3]))
    r.free()
    d.free()


def test_dense_sweep_equals_fused_iteration0():
    """The inflation sweep with one expansion (hhx_expand_links_dense + hhx_dense_inflate_prune): rows of M^2 stored once as
    float32, iteration 0 of every inflation from them — bit for bit the first iteration of hhx_mcl_links at that inflation, with
    the rows in one block, in ragged blocks, through the generic stream (non-integer values), and against the oracle on a slice."""
    import torch
    from haphic_amd import cluster, synth
    gen = synth.make_genome(8, 60_000_000, 20_000, seed=8)          # ~24k contigs -> 2 column windows
    n = gen.n
    lex = gen.lexical_rank()
    t = orc.FragTable(lex, gen.length, np.arange(n, dtype=np.int32), np.zeros(n, np.uint8), 0, lex, gen.length, np.ones(n, np.uint8))
    id1, p1, id2, p2 = synth.sample_pairs(gen, 8_000_000, seed=9, device='cuda')
    ing = _lib.Ingest(t, 500_000, bins=False, skip_intra=True)
    ing.push_device(id1.numel(), id1.data_ptr(), p1.data_ptr(), id2.data_ptr(), p2.data_ptr())
    torch.cuda.synchronize()
    links, fidx, n_linked = ing.link_matrix(np.ones(n, np.uint8))
    ing.destroy()
    n = links.shape3[0]
    inflations = (2.0, 1.4, 3.0)
    want = {r: _lib.mcl(links, 2, r, 1, 1e-4, want_stats=True, links=True) for r in inflations}
    whole = _lib.DenseRows(links, 0, n)
    assert whole.n_products == want[2.0][3][0, 3] and whole.nnz_expanded == want[2.0][3][0, 1]
    for r in inflations:
        got = whole.inflate_prune(r, 1e-4)
        assert all(np.array_equal(x, y) for x, y in zip(got.to_arrays(), want[r][0].to_arrays())), 'dense sweep, one block, inflation %r' % r
        got.free()
    whole.free()
    # several inflations in ONE pass over the block (hhx_dense_inflate_prune_multi: x and log2 x once per entry): the same bits, in
    # every grouping — 2.0 (x * x) is routed through the one-inflation kernel
    whole = _lib.DenseRows(links, 0, n)
    extra = {r: whole.inflate_prune(r, 1e-4) for r in (1.1, 1.7)}
    for group in ((1.4, 3.0), (3.0, 2.0, 1.1, 1.4, 1.7), (2.0,), (1.7,)):
        got = whole.inflate_prune_multi(group, 1e-4)
        for r, g_ in zip(group, got):
            ref = want[r][0] if r in want else extra[r]
            assert all(np.array_equal(x, y) for x, y in zip(g_.to_arrays(), ref.to_arrays())), 'multi-inflation epilogue, group %r, inflation %r' % (group, r)
            g_.free()
    whole.free()
    # the block stored as its UPPER BLOCK TRIANGLE alone (what an order the square does not fit for takes: n = 200k on one GPU) —
    # the lower blocks are never written, the epilogue turns them one block row at a time: the same bits
    _lib.tune('dense_tri', 1)
    try:
        tri = _lib.DenseRows(links, 0, n)
        assert tri.n_products == want[2.0][3][0, 3] and tri.nnz_expanded == want[2.0][3][0, 1]
        for r in inflations:
            got = tri.inflate_prune(r, 1e-4)
            assert all(np.array_equal(x, y) for x, y in zip(got.to_arrays(), want[r][0].to_arrays())), 'dense sweep, upper block triangle, inflation %r' % r
            got.free()
        for r, got in zip((1.4, 3.0, 1.7), tri.inflate_prune_multi((1.4, 3.0, 1.7), 1e-4)):
            ref = want[r][0] if r in want else extra[r]
            assert all(np.array_equal(x, y) for x, y in zip(got.to_arrays(), ref.to_arrays())), 'multi-inflation epilogue over the triangle, inflation %r' % r
            got.free()
        tri.free()
        fused = _lib.mcl(links, 2, 2.0, 1, 1e-4, want_stats=True, links=True)
        assert all(np.array_equal(x, y) for x, y in zip(fused[0].to_arrays(), want[2.0][0].to_arrays())) and np.array_equal(fused[3], want[2.0][3])
        fused[0].free()
    finally:
        _lib.tune('dense_tri', None)
    # ragged row blocks through the host driver (cluster.DenseSweep), the pieces stacked
    sweep = cluster.DenseSweep(links, 1e-4, block_rows=7001)
    assert len(sweep.bounds) == 5
    for r, first in zip(inflations, sweep.first_iterations(inflations)):
        assert all(np.array_equal(x, y) for x, y in zip(first.to_arrays(), want[r][0].to_arrays())), 'dense sweep, 4 blocks, inflation %r' % r
        # ... and the loop resumed from it ends where the fused mcl() ends
        if r == 2.0:
            a, na, ca = _lib.mcl_resume(first, 1, 2, r, 200, 1e-4)
            b, nb, cb = _lib.mcl(links, 2, r, 200, 1e-4, links=True)
            assert (na, ca) == (nb, cb) and all(np.array_equal(x, y) for x, y in zip(a.to_arrays(), b.to_arrays()))
            a.free(); b.free()
        first.free()
    sweep.close()
    # against the oracle on a slice of rows (mode 1: the kernels' own specification)
    mp, mj, mx = links.to_arrays()
    r0, r1 = 5000, 5040
    blk = _lib.DenseRows(links, r0, r1)
    c = orc.expand_links((mp, mj, mx), rows=np.arange(r0, r1))
    for r in (2.0, 1.7):
        ref = orc.prune((c[0], c[1], orc.normalize_l1(c[0], orc.power(c[2], r))), 1e-4)
        got = blk.inflate_prune(r, 1e-4).to_arrays()
        assert np.array_equal(got[0], ref[0]) and np.array_equal(got[1], ref[1])
        if r == 2.0:
            assert np.array_equal(got[2], ref[2])
        else:
            np.testing.assert_allclose(got[2], ref[2], rtol=RTOL)      # powf (oracle) vs float(pow) (device)
    blk.free()
    # an empty row block, and values that are not integer counts (generic stream)
    empty = _lib.DenseRows(links, 100, 100)
    e = empty.inflate_prune(2.0, 1e-4)
    assert e.shape3 == (0, n, 0)
    x2 = (mx * np.float32(0.37)).astype(np.float32)
    l2 = _lib.DeviceCSR.from_arrays(mp, mj, x2)
    w2 = _lib.mcl(l2, 2, 2.0, 1, 1e-4, links=True)[0]
    d2 = _lib.DenseRows(l2, 0, n)
    g2 = d2.inflate_prune(2.0, 1e-4)
    assert all(np.array_equal(x, y) for x, y in zip(g2.to_arrays(), w2.to_arrays()))
    for x in (e, g2, w2, l2, links):
        x.free()
    d2.free()
    empty.free()


def test_pack_unpack_row_blocks_and_empty_blocks():
    """the packed exchange message of the row-block MCL (hhx_csr_pack_block / hhx_csr_unpack_blocks) — blocks of different
    sizes, an EMPTY block among them (balanced_ranges can leave a rank without rows) — and a fused iteration on a
    block without rows"""
    import torch
    A = clustered_stochastic(3000, 60, 6, 1, 3)
    d = _lib.DeviceCSR.from_arrays(*A)
    cuts = [0, 700, 700, 2100, 3000]
    blocks = [d.row_block(a, b) for a, b in zip(cuts[:-1], cuts[1:])]
    rows = np.array([b.shape3[0] for b in blocks], np.int64)
    nnzs = np.array([b.shape3[2] for b in blocks], np.int64)
    stride = int((rows + 2 * nnzs).max()) + 5
    packed = torch.zeros(len(blocks) * stride, dtype=torch.int32, device='cuda')
    for k, b in enumerate(blocks):
        _lib.check(_lib.load().hhx_csr_pack_block(b.h, _lib.C.c_void_p(packed[k * stride:].data_ptr()), stride))
    torch.cuda.synchronize()
    out = _lib.C.c_void_p()
    _lib.check(_lib.load().hhx_csr_unpack_blocks(len(blocks), rows.ctypes.data_as(_lib.c_i64p), nnzs.ctypes.data_as(_lib.c_i64p),
                                                 _lib.C.c_void_p(packed.data_ptr()), stride, 3000, _lib.C.byref(out)))
    back = _lib.DeviceCSR(out)
    assert all(np.array_equal(x, y) for x, y in zip(back.to_arrays(), d.to_arrays()))
    with pytest.raises(RuntimeError, match='buffer of'):
        _lib.check(_lib.load().hhx_csr_pack_block(blocks[0].h, _lib.C.c_void_p(packed.data_ptr()), 10))
    got, f, z = _lib.expand_inflate_prune(blocks[1], d, 2.0, 1e-4)        # the empty block through the fused iteration
    assert got.shape3 == (0, 3000, 0) and f == 0 and z == 0
    assert _lib.row_products(blocks[1], d).size == 0
    for x in blocks + [back, got, d]:
        x.free()


def test_wide_positions_int64_path(tmp_path):
    """Contigs of 2^31 bp and more (determine_int_type :116-147): the device tokeniser in wide mode (int64 positions,
    the same alignments.bed bytes), hhx_ingest_push64 through host arrays, device arrays and the .pairs file front end, the CLM /
    coordinate side records — all against the containers the reference itself produced on an assembly with a 3.0 Gb and a 2.3 Gb
    contig (tests/golden/ingest_wide.npz), and the mirror parse_alignments_for_ctgs with the reference's int types."""
    from array import array
    from haphic_amd import cluster
    from tests.conftest import load_golden
    g = load_golden('ingest_wide.npz')
    names = [str(x) for x in g['names']]
    n = len(names)
    table = cluster.FragTable.for_contigs(cluster.FragTable._rank(names), g['ctg_len'], np.ones(n, np.uint8), names=names)
    assert table.wide
    text = g['text'].tobytes()
    # tokeniser, wide mode
    ps = _lib.PairsParser(names)
    with pytest.raises(ValueError, match='int32'):
        ps.parse(text)                                           # positions beyond 2^31 without the wide mode: refused, loudly
    ps.set_wide(True)
    assert ps.parse(text, want_bed=True) == text.count(b'\n')
    i1, p1, i2, p2, bed = ps.fetch(want_bed=True)
    assert p1.dtype == np.int64 and bed == g['bed'].tobytes()
    oi1, op1, oi2, op2, _ = orc.parse_pairs_text(text, names, wide=True)
    assert all(np.array_equal(x, y) for x, y in zip((i1, p1, i2, p2), (oi1, op1, oi2, op2)))
    tup = g['tuples']
    tables = ('full_i', 'full_j', 'full_cnt', 'ht_cnt', 'flank_i', 'flank_j', 'flank_cnt', 'frag_links')

    def check(ing):
        ing.finalize()
        got = ing.fetch()
        for k in tables:
            assert np.array_equal(got[k], g[k]), k
        clm_ptr, clm, crd_ptr, crd = ing.fetch_pairs(int(g['max_read_pairs']), got['full_cnt'])
        assert np.array_equal(clm_ptr * 4, g['clm_ptr']) and np.array_equal(clm, g['clm'])
        assert np.array_equal(crd_ptr * 2, g['crd_ptr']) and np.array_equal(crd, g['crd'])
        ing.destroy()
    # host arrays (int64 positions -> hhx_ingest_push64), in two ragged pushes
    ing = _lib.Ingest(table, int(g['flank']), bins=False, skip_intra=True)
    ing.keep_pairs()
    cut = 1777
    for lo, hi in ((0, cut), (cut, len(tup))):
        ing.push(tup[lo:hi, 0].astype(np.int32), tup[lo:hi, 1], tup[lo:hi, 2].astype(np.int32), tup[lo:hi, 3])
    check(ing)
    # the width is decided by the VALUES whatever the dtype: uint32 positions of 2^31 and more take the 64-bit path too,
    # and an explicit wide=False refuses them instead of wrapping them negative
    assert int(tup[:, 1].max()) > 2 ** 31 and int(tup[:, 1].max()) < 2 ** 32
    ing = _lib.Ingest(table, int(g['flank']), bins=False, skip_intra=True)
    ing.keep_pairs()
    ing.push(tup[:, 0].astype(np.int32), tup[:, 1].astype(np.uint32), tup[:, 2].astype(np.int32), tup[:, 3].astype(np.uint32))
    check(ing)
    ing = _lib.Ingest(table, int(g['flank']), bins=False, skip_intra=True)
    with pytest.raises(ValueError, match='int32'):
        ing.push(tup[:, 0].astype(np.int32), tup[:, 1], tup[:, 2].astype(np.int32), tup[:, 3], wide=False)
    ing.destroy()
    # device arrays straight from the tokeniser (every line: headers and intra-contig pairs are dropped on the device)
    ing = _lib.Ingest(table, int(g['flank']), bins=False, skip_intra=True)
    ing.keep_pairs()
    ing.push_device(ps.n_lines, *ps.device_arrays()[:4], wide=True)
    check(ing)
    ps.destroy()


def test_wide_positions_file_front_end_and_mirror(tmp_path):
    """the .pairs file front end and the mirror of parse_alignments_for_ctgs on the same wide case, with the int types the
    reference's determine_int_type picks (int64 containers: array('l'))"""
    from array import array
    from haphic_amd import cluster
    from tests.conftest import load_golden
    g = load_golden('ingest_wide.npz')
    names = [str(x) for x in g['names']]
    text = g['text'].tobytes()
    path = tmp_path / 'wide.pairs'
    path.write_bytes(text)
    fa_dict = {nm: [None, int(l), 5] for nm, l in zip(names, g['ctg_len'])}

    class A:
        flank = 500
        remove_allelic_links = 4
        remove_concentrated_links = False
        max_read_pairs = int(g['max_read_pairs'])
        nwindows = 50
    aln = cluster.pairs_generator_inter_ctgs(str(path), 'pairs')
    aln.bed_path = str(tmp_path / 'alignments.bed')
    frag_len = {nm: fa_dict[nm][1] for nm in names}
    import haphic_amd.cluster as C_
    orig = C_.cal_concordance_ratio
    C_.cal_concordance_ratio = lambda coord_list, shorter_len, nwindows: tuple(coord_list)
    try:
        full, flank, HT, clm, frag_link, coord = cluster.parse_alignments_for_ctgs(aln, fa_dict, A(), frag_len, set(names), 'int64', 'int64')
    finally:
        C_.cal_concordance_ratio = orig
    cluster._lib.files_join()
    assert (tmp_path / 'alignments.bed').read_bytes() == g['bed'].tobytes()
    assert [(names.index(a), names.index(b)) for a, b in full] == list(zip(g['full_i'].tolist(), g['full_j'].tolist()))
    assert list(full.values()) == g['full_cnt'].tolist() and list(flank.values()) == g['flank_cnt'].tolist()
    k0 = next(iter(full))
    assert isinstance(clm[k0], array) and clm[k0].typecode == 'l' and max(max(v) for v in clm.values()) == int(g['clm'].max()) > 2 ** 32


def test_link_matrix_packed_and_wide_entries():
    """hhx_ingest_link_matrix moves 8-byte entries (row, column, count < 2^24) through its partition and 16-byte ones when
    a count needs more bits: both against the oracle's dict_to_matrix on the fetched tables, > 4096 fragments (the
    sort-ranked index assignment), then with one contig pair holding 17 M links (the automatic fall-back)"""
    import os
    import torch
    from haphic_amd import synth
    gen = synth.make_genome(5, 20_000_000, 20_000, seed=8)
    n = gen.n
    assert n > 4096
    t = orc.FragTable(gen.lexical_rank(), gen.length, np.arange(n, dtype=np.int32), np.zeros(n, np.uint8), 0,
                      gen.lexical_rank(), gen.length, (np.arange(n) % 11 != 0).astype(np.uint8))
    id1, p1, id2, p2 = synth.sample_pairs(gen, 3_000_000, seed=2, device='cuda')
    in_set = t.frag_nx.copy()

    def check(ing):
        out = ing.fetch()
        linked = np.zeros(n, bool)
        ok = in_set[out['flank_i']].astype(bool) & in_set[out['flank_j']].astype(bool)
        linked[out['flank_i'][ok]] = True
        linked[out['flank_j'][ok]] = True
        n_rest = int(in_set.sum() - linked.sum())
        want = orc.dict_to_matrix(out['flank_i'], out['flank_j'], out['flank_cnt'].astype(np.float64), n, in_set, n_rest)
        got = []
        for wide in (False, True):
            if wide:
                os.environ['HHX_D2M_WIDE'] = '1'
            try:
                m, fidx, nl = ing.link_matrix(in_set)
            finally:
                os.environ.pop('HHX_D2M_WIDE', None)
            assert nl == want[4] and np.array_equal(fidx, want[3])
            assert all(np.array_equal(a, b) for a, b in zip(m.to_arrays(), want[:3])), wide
            got.append(m)
        return out

    ing = _lib.Ingest(t, 5000, bins=False, skip_intra=True)
    ing.push_device(id1.numel(), id1.data_ptr(), p1.data_ptr(), id2.data_ptr(), p2.data_ptr())
    check(ing)
    ing.destroy()
    # one heavy contig pair: 17 M read pairs inside the flanks of contigs a, b (both in the Nx set)
    a, b = 1, 2
    heavy = 17_000_000
    ing = _lib.Ingest(t, 5000, bins=False, skip_intra=True)
    ing.push_device(id1.numel(), id1.data_ptr(), p1.data_ptr(), id2.data_ptr(), p2.data_ptr())
    ha = torch.full((heavy,), a, dtype=torch.int32, device='cuda')
    hb = torch.full((heavy,), b, dtype=torch.int32, device='cuda')
    hp = torch.full((heavy,), 10, dtype=torch.int32, device='cuda')
    ing.push_device(heavy, ha.data_ptr(), hp.data_ptr(), hb.data_ptr(), hp.data_ptr())
    out = check(ing)
    k = np.flatnonzero((out['flank_i'] == a) & (out['flank_j'] == b))
    assert len(k) == 1 and out['flank_cnt'][k[0]] > (1 << 24)
    ing.destroy()


def test_pool_blocks_taken_ahead_and_the_trim_that_keeps():
    """hhx_pool_prewarm / hhx_pool_trim / hhx_pool_trim_keep (include/haphic_hip.h): a block taken ahead of its use survives ONE trim; the keeping trim leaves the
    small blocks and, largest first, up to keep_bytes of the mid-size ones.  (Sizes are 64 MiB granules above 64 MiB: the cached byte counts are exact.)"""
    L = _lib.load()
    _lib.check(L.hhx_pool_trim())
    _lib.check(L.hhx_pool_trim())                 # twice: whatever an earlier test took ahead is gone too
    base = _lib.pool_cached_bytes()
    MiB = 1 << 20
    _lib.pool_prewarm([640 * MiB])
    assert _lib.pool_cached_bytes() == base + 640 * MiB
    _lib.check(L.hhx_pool_trim())
    assert _lib.pool_cached_bytes() == base + 640 * MiB, 'the block taken ahead did not survive the first trim'
    _lib.check(L.hhx_pool_trim())
    assert _lib.pool_cached_bytes() == base, 'an unused block taken ahead survived two trims'
    # the keeping trim: 3 mid-size blocks of 128 / 256 / 512 MiB and a small one; keep_bytes = 800 MiB keeps 512 + 256
    _lib.check(L.hhx_pool_trim())
    _lib.check(L.hhx_pool_trim())
    base = _lib.pool_cached_bytes()
    _lib.pool_prewarm([128 * MiB, 256 * MiB, 512 * MiB, 1 * MiB])
    _lib.check(L.hhx_pool_trim())                 # spends their one exemption: plain cached blocks from here on
    assert _lib.pool_cached_bytes() == base + 897 * MiB
    _lib.check(L.hhx_pool_trim_keep(800 * MiB))
    assert _lib.pool_cached_bytes() == base + (512 + 256 + 1) * MiB
    _lib.check(L.hhx_pool_trim_keep(0))           # keep_bytes 0 = hhx_pool_trim
    assert _lib.pool_cached_bytes() == base
    assert L.hhx_pool_trim_keep(-1) != 0
