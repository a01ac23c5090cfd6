"""The oracle (oracle/hhx_oracle.c) against golden vectors produced by the reference's own Python
functions (tests/golden/make_golden.py).  CPU only."""
import numpy as np

from oracle import oracle as orc


def tri(g, prefix):
    return g[prefix + '_p'], g[prefix + '_j'], g[prefix + '_x']


def assert_close_csr(a, b, rtol, what):
    assert np.array_equal(a[0], b[0]), what + ': indptr'
    assert np.array_equal(a[1], b[1]), what + ': indices'
    np.testing.assert_allclose(a[2], b[2], rtol=rtol, atol=0, err_msg=what)


def test_normalize_bit_exact(golden_mcl):
    g = golden_mcl
    for tag in g['cases']:
        p, j, x = tri(g, tag + '_link')
        assert np.array_equal(orc.normalize_l1(p, x), g[tag + '_norm_x'])


def test_spgemm_matches_scipy_standin(golden_mcl):
    g = golden_mcl
    for tag in g['cases']:
        A = tri(g, tag + '_norm')
        C = orc.spgemm(A, A, mode=0)
        # float32 accumulation in ascending-k order == scipy csr_matmat on canonical input: bit exact
        ref = tri(g, tag + '_m2')
        assert np.array_equal(C[0], ref[0]) and np.array_equal(C[1], ref[1])
        assert np.array_equal(C[2], ref[2])
        # fixed-point mode (the HIP kernel's specification) agrees to float32 round-off
        F = orc.spgemm(A, A, mode=1, fx_shift=62)
        assert_close_csr(F, ref, 1e-6, 'fixed-point spgemm')


def test_integer_pre_expansion_against_the_reference(golden_mcl):
    """The integer specification of the pre-expansion (orc.expand_links: S = L D^-1 L in exact integers, then one division per
    row — the symmetric form the HIP kernels evaluate) against the reference's own normalize + M @ M (:2144-2147) on every
    golden link matrix that holds integer counts: same pattern, values within float32 round-off of the float32 product, and the
    whole mcl() from it ends in the reference's clusters."""
    g = golden_mcl
    done = 0
    for tag in g['cases']:
        L = tri(g, tag + '_link')
        sym = (np.asarray(L[2]) == np.rint(L[2])).all()
        if not sym or orc.links_shift(L) < 0:
            continue
        import scipy.sparse as sp
        m = sp.csr_matrix((L[2], L[1], L[0]), shape=(len(L[0]) - 1,) * 2)
        if (m != m.T).nnz:
            continue
        done += 1
        ref = tri(g, tag + '_m2')
        C = orc.expand_links(L)
        assert_close_csr(C, ref, 2e-6, tag + ': integer pre-expansion vs the reference float32 product')
        F = orc.spgemm(tri(g, tag + '_norm'), tri(g, tag + '_norm'), mode=1, fx_shift=52)
        assert_close_csr(C, F, 3e-7, tag + ': integer specification vs the fixed-point one')       # the roundings of the normalised entries
        infl = float(g[tag + '_inflation'])
        p, j, x, n_iter, conv = orc.mcl(C, 2, infl, 200, 1e-4, spgemm_mode=1, fx_shift=52)
        att, ptr, mem = orc.interpret((p, j, x))
        cp, cl = g[tag + '_clusters_ptr'], g[tag + '_clusters']
        assert {tuple(mem[ptr[a]:ptr[a + 1]].tolist()) for a in range(len(att))} == {tuple(cl[cp[a]:cp[a + 1]].tolist()) for a in range(len(cp) - 1)}
        # rows on their own == rows of the whole
        rows = np.arange(0, len(L[0]) - 1, 7, dtype=np.int32)
        sub = orc.expand_links(L, rows=rows)
        for t, r in enumerate(rows):
            assert np.array_equal(sub[2][sub[0][t]:sub[0][t + 1]], C[2][C[0][r]:C[0][r + 1]])
        # the one-pass form the whole-matrix checks at C3 use (orc.links_iteration0) == the composition, bit for bit
        for r in (infl, 1.3):
            rows7 = np.arange(len(L[0]) - 1, dtype=np.int32)[::-1][::3].copy()
            sub7 = orc.expand_links(L, rows=rows7)
            want = orc.prune((sub7[0], sub7[1], orc.normalize_l1(sub7[0], orc.power(sub7[2], r))), 1e-4)
            got = orc.links_iteration0(L, rows7, r, 1e-4)
            assert got[3] == len(sub7[1]) and all(np.array_equal(u, v) for u, v in zip(got[:3], want))
    assert done >= 1, 'no golden case holds a symmetric integer link matrix'


def test_mcl_iterations(golden_mcl):
    g = golden_mcl
    for tag in g['cases']:
        infl = float(g[tag + '_inflation'])
        niter = int(g[tag + '_niter'])
        cur = tri(g, tag + '_m2')
        last = None
        for it in range(niter):
            if it:
                cur = orc.spgemm(cur, cur, mode=0)
            x = orc.normalize_l1(cur[0], orc.power(cur[2], infl))
            cur = orc.prune((cur[0], cur[1], x), 1e-4)
            ref = tri(g, '%s_it%d' % (tag, it))
            # power(): numpy's SIMD powf is within 1 ulp of libm's; everything else is exact
            assert_close_csr(cur, ref, 0 if infl == 2.0 else 5e-7, '%s iter %d' % (tag, it))
            if infl != 2.0:
                cur = ref      # re-anchor so that 1-ulp pow differences do not compound
            if it > 1:
                d = orc.convergence_stat(cur, last)
                assert (d <= np.float32(1e-8)) == (it == niter - 1), (tag, it, d)
            last = cur


def test_mcl_driver_and_interpret(golden_mcl):
    g = golden_mcl
    for tag in g['cases']:
        infl = float(g[tag + '_inflation'])
        niter = int(g[tag + '_niter'])
        p, j, x, n_iter, conv, stats = orc.mcl(tri(g, tag + '_m2'), 2, infl, 200, 1e-4, want_stats=True)
        assert conv and n_iter == niter
        ref = tri(g, '%s_it%d' % (tag, niter - 1))
        assert_close_csr((p, j, x), ref, 0 if infl == 2.0 else 1e-5, tag + ' final')
        assert stats[0, 3] == 0 and (stats[1:, 3] > 0).all()
        att, ptr, mem = orc.interpret((p, j, x))
        got = {tuple(mem[ptr[a]:ptr[a + 1]].tolist()) for a in range(len(att))}
        cp, cl = g[tag + '_clusters_ptr'], g[tag + '_clusters']
        want = {tuple(cl[cp[a]:cp[a + 1]].tolist()) for a in range(len(cp) - 1)}
        assert got == want
        # fixed-point spgemm mode gives the same partition
        p2, j2, x2, n2, c2 = orc.mcl(tri(g, tag + '_m2'), 2, infl, 200, 1e-4, spgemm_mode=1)
        att, ptr, mem = orc.interpret((p2, j2, x2))
        assert {tuple(mem[ptr[a]:ptr[a + 1]].tolist()) for a in range(len(att))} == want


def table_of(g):
    return orc.FragTable(g['ctg_rank'], g['ctg_len'], g['ctg_frag0'], g['ctg_split'], int(g['bin_size']),
                         g['frag_rank'], g['frag_len'], g['frag_nx'])


def test_ingest(golden_ingest):
    g = golden_ingest
    for chunk in (None, 997):
        out = orc.ingest(table_of(g), g['id1'], g['pos1'], g['id2'], g['pos2'], int(g['flank']),
                         bins=bool(g['bins']), want_clm=True, max_read_pairs=int(g['max_read_pairs']), chunk=chunk)
        for k in ('full_i', 'full_j', 'full_cnt', 'ht_cnt', 'flank_i', 'flank_j', 'flank_cnt', 'frag_links',
                  'clm_ptr', 'clm', 'crd_ptr', 'crd'):
            assert np.array_equal(out[k], g[k]), k


def _text_rows(g, which):
    return g[which][:, 0], g[which][:, 1], g[which][:, 2], g[which][:, 3]


def test_pairs_text_restatement_pinned():
    """a1: the tuples pairs_generator (:1539-1559) yields and the alignments.bed it writes, frozen from the
    reference on a text with headers, blank lines, CR / CRLF ends, odd separators and odd integers"""
    from tests.conftest import load_golden
    g = load_golden('pairs_text.npz')
    names = [str(x) for x in g['names']]
    i1, p1, i2, p2, bed = orc.parse_pairs_text(g['text'].tobytes(), names)
    live = ~((i1 == -1) & (i2 == -1) & (p1 == 0) & (p2 == 0))
    assert np.array_equal(np.stack([i1, p1, i2, p2], 1)[live], g['all'])
    assert bed == g['bed_all'].tobytes()


def _wide_case():
    from tests.conftest import load_golden
    g = load_golden('ingest_wide.npz')
    names = [str(x) for x in g['names']]
    n = len(names)
    order = sorted(range(n), key=names.__getitem__)
    rank = np.empty(n, np.int32)
    rank[order] = np.arange(n, dtype=np.int32)
    t = orc.FragTable(rank, g['ctg_len'], np.arange(n, dtype=np.int32), np.zeros(n, np.uint8), 0, rank, g['ctg_len'], np.ones(n, np.uint8))
    return g, names, t


def test_wide_positions_restatement_pinned():
    """contigs of 2^31 bp and more (determine_int_type :116-147 -> int64 containers): the tokeniser's tuples and alignments.bed,
    and every container of parse_alignments_for_ctgs, frozen from the reference on an assembly with a 3.0 Gb and a 2.3 Gb contig"""
    g, names, t = _wide_case()
    assert str(g['pos_int_type']) == 'int64' and g['tuples'][:, [1, 3]].max() > 2 ** 31
    i1, p1, i2, p2, bed = orc.parse_pairs_text(g['text'].tobytes(), names, wide=True)
    assert p1.dtype == np.int64 and bed == g['bed'].tobytes()
    text_names = [ln.split('\t')[1:4:2] for ln in g['text'].tobytes().decode().splitlines() if ln and not ln.startswith('#')]
    live = np.array([a != b for a, b in text_names])             # pairs_generator_inter_ctgs :1582 compares NAMES
    rows = np.stack([i1, p1, i2, p2], 1)[1:][live]                # line 0 is the header
    assert np.array_equal(rows, g['tuples'])
    tup = g['tuples']
    got = orc.ingest(t, tup[:, 0].astype(np.int32), tup[:, 1], tup[:, 2].astype(np.int32), tup[:, 3], int(g['flank']), want_clm=True,
                     max_read_pairs=int(g['max_read_pairs']))
    for k in ('full_i', 'full_j', 'full_cnt', 'ht_cnt', 'flank_i', 'flank_j', 'flank_cnt', 'frag_links', 'clm', 'crd'):
        assert np.array_equal(got[k], g[k]), k
    assert np.array_equal(got['clm_ptr'], g['clm_ptr']) and np.array_equal(got['crd_ptr'], g['crd_ptr'])


def test_frag_pairs_restatement_pinned():
    """ctg_pair_to_frag (:1731-1733), frozen from the reference's parse_alignments on split contigs"""
    from tests.conftest import load_golden
    g = load_golden('ingest_bins.npz')
    fi, fj = orc.frag_pairs(table_of(g), g['id1'], g['pos1'], g['id2'], g['pos2'])
    want = np.unique(g['c2f'][:, 2:], axis=0)
    assert len(want) == len(g['c2f'])                      # a fragment pair belongs to exactly one contig pair
    assert np.array_equal(np.stack([fi, fj], 1), want)


def test_ht_order_restatement_pinned(golden_ingest):
    """HT_link_dict's insertion order (:404-416), frozen from the reference: sorting the non-empty (contig pair,
    quadrant) entries by the first stream position the restatement finds reproduces it"""
    g = golden_ingest
    first = orc.ht_first(table_of(g), g['id1'], g['pos1'], g['id2'], g['pos2'], g['full_i'], g['full_j'])
    k, q = np.nonzero(g['ht_cnt'])
    assert (first[k, q] != np.iinfo(np.int64).max).all() and (first[g['ht_cnt'] == 0] == np.iinfo(np.int64).max).all()
    order = np.argsort(first[k, q], kind='stable')
    assert np.array_equal(np.stack([k[order], q[order]], 1), g['ht_order'])


def test_dict_to_matrix(golden_ingest):
    g = golden_ingest
    in_set = g['d2m_in_set']
    linked = np.zeros(len(in_set), bool)
    ok = in_set[g['flank_i']].astype(bool) & in_set[g['flank_j']].astype(bool)
    linked[g['flank_i'][ok]] = True
    linked[g['flank_j'][ok]] = True
    n_rest = int(in_set.sum() - linked.sum())
    p, j, x, fidx, n_linked = orc.dict_to_matrix(g['flank_i'], g['flank_j'], g['flank_cnt'].astype(np.float64),
                                                 len(in_set), in_set, n_rest)
    assert np.array_equal(p, g['d2m_p']) and np.array_equal(j, g['d2m_j']) and np.array_equal(x, g['d2m_x'])
    ref_idx = g['d2m_frag_index']
    assert np.array_equal(fidx[linked], ref_idx[linked])         # first-seen order of linked fragments
    assert (ref_idx[in_set.astype(bool) & ~linked] >= n_linked).all()
    # normalize_by_nlinks (:718-724): python-float division, then the float32 cast of :368
    fl = g['frag_links'].astype(np.float64)
    val = g['flank_cnt'] / (fl[g['flank_i']] * fl[g['flank_j']]) ** 0.5
    x2 = orc.dict_to_matrix(g['flank_i'], g['flank_j'], val, len(in_set), in_set, n_rest)[2]
    assert np.array_equal(x2, g['d2m_nlinks_x'])


def test_count_re_sites(request):
    from tests.conftest import load_golden
    g = load_golden('resites.npz')
    for RE in g['REs']:
        RE = str(RE)
        sites = [str(x).encode() for x in g['sites_' + RE]]
        got = orc.count_re_sites(g['seq'], g['seg_off'], g['seg_len'], sites)
        assert np.array_equal(got, g['counts_' + RE]), RE


def _filter_inputs(g):
    names = [str(x) for x in g['names']]
    Nx_set = {n for n, x in zip(names, g['nx']) if x}
    RE_site_dict = {n: int(r) for n, r in zip(names, g['re_sites'])}
    frag_link = {n: int(c) for n, c in zip(names, g['frag_links']) if c >= 0}
    flank = {(names[i], names[j]): int(c) for i, j, c in zip(g['flank_i'], g['flank_j'], g['flank_cnt'])}
    return names, Nx_set, RE_site_dict, frag_link, flank


def test_rank_sums_pin_filter_fragments(monkeypatch):
    """the oracle's rank-sum statistic, pinned through the reference's filter_fragments outputs: the host mirror
    (haphic_amd/cluster.py:filter_fragments) is run with the oracle standing in for the two device calls"""
    from haphic_amd import cluster, _lib
    from tests.conftest import load_golden
    g = load_golden('filter.npz')
    names, Nx_set, RE_site_dict, frag_link, flank = _filter_inputs(g)

    class HostMatrix:
        def __init__(self, tri):
            self.tri = tri

        def free(self):
            pass

    def d2m(link_dict, frag_set, dense_matrix=True, add_self_loops=False, _device=False):
        ids = {}
        fi, fj, fv = [], [], []
        for (a, b), v in link_dict.items():
            fi.append(ids.setdefault(a, len(ids))); fj.append(ids.setdefault(b, len(ids))); fv.append(v)
        for f in frag_set:
            ids.setdefault(f, len(ids))
        id_names = list(ids)
        in_set = np.array([f in frag_set for f in id_names], np.uint8)
        fi, fj = np.array(fi, np.int32), np.array(fj, np.int32)
        ok = in_set[fi].astype(bool) & in_set[fj].astype(bool)
        linked = np.zeros(len(id_names), bool)
        linked[fi[ok]] = True
        linked[fj[ok]] = True
        rest = [f for f in frag_set if not linked[ids[f]]]
        p, j, x, fidx, nl = orc.dict_to_matrix(fi, fj, np.array(fv, np.float64), len(id_names), in_set, len(rest), add_self_loops)
        frag_index = {id_names[i]: int(fidx[i]) for i in np.flatnonzero(linked)}
        for k, f in enumerate(rest):
            frag_index[f] = nl + k
        return HostMatrix((p, j, x)), frag_index

    monkeypatch.setattr(cluster, 'dict_to_matrix', d2m)
    monkeypatch.setattr(_lib, 'rank_sums', lambda m, topN: orc.rank_sums(m.tri, topN))
    cluster.logger.setLevel('WARNING')
    for k in range(int(g['n_cases'])):
        cut, lo, up, topn, rsu, hard = [str(x) for x in g['case%d_params' % k]]
        wl = {str(x) for x in g['case%d_whitelist' % k]} or None
        kept = cluster.filter_fragments(set(Nx_set), RE_site_dict, int(cut), frag_link, lo, up, int(topn), rsu, int(hard), flank, {}, '1.5X', wl)
        assert sorted(kept) == [str(x) for x in g['case%d_kept' % k]], k


def test_coordinate_statistics_mirrors_pinned():
    """the host statistics of --remove_allelic_links / --remove_concentrated_links (cal_concordance_ratio :419-428,
    cal_concentration_adj_ratio :431-451): cluster.py's mirrors return the reference's floats exactly"""
    from array import array
    from haphic_amd import cluster
    from tests.conftest import load_golden
    g = load_golden('coord_stats.npz')
    for k in range(len(g['shorter'])):
        c = array('i', g['coords'][g['ptr'][k]:g['ptr'][k + 1]].tolist())
        assert float(cluster.cal_concordance_ratio(c, int(g['shorter'][k]), 50)) == g['concordance'][k]
        assert float(cluster.cal_concentration_adj_ratio(c)) == g['adj'][k]


def test_link_weights_against_reference():
    """a6: the numpy restatement (oracle.link_weights) against the reference's dict rewrites, bit for bit"""
    from tests.conftest import load_golden
    g = load_golden('weights.npz')
    fi, fj, cnt = g['fi'], g['fj'], g['cnt'].astype(np.float64)
    assert np.array_equal(orc.link_weights(fi, fj, cnt, 0, per_frag=g['links']), g['nlinks'])
    assert np.array_equal(orc.link_weights(fi, fj, cnt, 1, per_frag=g['length'], param=2000 * int(g['flank_kb'])), g['by_length'])
    for tag_, w in (('w1', 1.0), ('w05', 0.5), ('w03', 0.3)):
        v = orc.link_weights(fi, fj, cnt, 2, tag=g['hap'], param=w)
        kept = np.flatnonzero(~((v == 0) & (g['hap'][fi] != g['hap'][fj])))
        assert np.array_equal(kept, g['hap_%s_kept' % tag_]) and np.array_equal(v[kept], g['hap_%s_values' % tag_])


def test_group_link_sums_against_reference():
    """f3: oracle.group_link_sums against HapHiC_reassign.parse_link_dict's nested dicts (values and first-contribution order)"""
    from tests.conftest import load_golden
    g = load_golden('reassign.npz')
    sums, first = orc.group_link_sums(g['fi'], g['fj'], g['links'], g['group'], int(g['n_groups']))
    rows, cols = np.nonzero(first >= 0)
    row_first = np.full(len(g['group']), np.iinfo(np.int64).max)
    np.minimum.at(row_first, rows, first[rows, cols])
    outer = [r for r in np.argsort(row_first, kind='stable').tolist() if row_first[r] != np.iinfo(np.int64).max]
    assert outer == g['outer'].tolist()
    cells = []
    for r in outer:
        for c in np.argsort(np.where(first[r] >= 0, first[r], np.iinfo(np.int64).max), kind='stable').tolist():
            if first[r, c] >= 0:
                cells.append((r, c, int(sums[r, c])))
    assert cells == [tuple(x) for x in g['cells'].tolist()]
