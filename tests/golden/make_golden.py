#!/usr/bin/env python3
"""Generate golden vectors by running the REFERENCE's own functions (HapHiC_cluster.py).

Run in the dev container only (needs /root/reference):
    PYTHONHASHSEED=0 python tests/golden/make_golden.py
The reference has no tests / golden vectors of its own, so these fixtures are what pins
the oracle (oracle/hhx_oracle.c) and, through it, the HIP kernels.  Only the two import stubs of
SURVEY Appendix B are applied (pysam / portion are absent here); MKL is absent too, so the sparse
code path runs with scipy's `@` standing in for sparse_dot_mkl.dot_product_mkl.
"""
import os
import sys
import types

import numpy as np
import scipy.sparse as sp

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

for name, attrs in (('pysam', {'set_verbosity': lambda *a, **k: None, 'AlignmentFile': None}),
                    ('portion', {'closed': None, 'empty': None})):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
sys.path.insert(0, '/root/reference/scripts')
import HapHiC_cluster as H  # noqa: E402

H.dot_product_mkl = lambda a, b: (a @ b).tocsc()
H.INTEL_MKL = True
H.logger.setLevel('WARNING')

from haphic_amd import synth  # noqa: E402


def canon(m):
    m = m.tocsc().copy()
    m.sum_duplicates()
    m.sort_indices()
    return m.indptr.astype(np.int32), m.indices.astype(np.int32), m.data.astype(np.float32)


def planted_matrix(n, ngroups, seed, p_in=0.5, p_out=0.02, max_cnt=40):
    rng = np.random.default_rng(seed)
    grp = rng.integers(0, ngroups, n)
    iu, ju = np.triu_indices(n, 1)
    p = np.where(grp[iu] == grp[ju], p_in, p_out)
    keep = rng.random(iu.size) < p
    iu, ju = iu[keep], ju[keep]
    cnt = np.where(grp[iu] == grp[ju], rng.integers(1, max_cnt, iu.size), rng.integers(1, 4, iu.size)).astype(np.float32)
    rows = np.concatenate([iu, ju, np.arange(n)])
    cols = np.concatenate([ju, iu, np.arange(n)])
    data = np.concatenate([cnt, cnt, np.ones(n, np.float32)])
    return sp.coo_matrix((data, (rows, cols)), shape=(n, n), dtype=np.float32).tocsc()


def gen_mcl(path):
    out = {}
    cases = [('a', 150, 4, 1, 2.0), ('b', 320, 6, 2, 2.0), ('c', 200, 5, 3, 1.7), ('d', 90, 3, 4, 3.0)]
    out['cases'] = np.array([c[0] for c in cases])
    for tag, n, ng, seed, infl in cases:
        link = planted_matrix(n, ng, seed)
        norm = H.normalize(link, norm='l1', axis=0)
        m2 = H.mkl_matrix_power(norm, 2)
        # record the matrix at the end of every iteration = every return value of prune()
        per_iter = []
        orig_prune = H.prune

        def rec_prune(matrix, pruning, dense_matrix, _o=orig_prune, _l=per_iter):
            r = _o(matrix, pruning, dense_matrix)
            _l.append(canon(r))
            return r
        H.prune = rec_prune
        try:
            res = H.mcl(m2, 2, infl, 200, 1e-4, False)
        finally:
            H.prune = orig_prune
        clusters = H.interpret_result(res, False)
        for k, v in zip(('p', 'j', 'x'), canon(link)):
            out['%s_link_%s' % (tag, k)] = v
        for k, v in zip(('p', 'j', 'x'), canon(norm)):
            out['%s_norm_%s' % (tag, k)] = v
        for k, v in zip(('p', 'j', 'x'), canon(m2)):
            out['%s_m2_%s' % (tag, k)] = v
        out[tag + '_inflation'] = np.float64(infl)
        out[tag + '_niter'] = np.int32(len(per_iter))
        for it, (p, j, x) in enumerate(per_iter):
            out['%s_it%d_p' % (tag, it)] = p
            out['%s_it%d_j' % (tag, it)] = j
            out['%s_it%d_x' % (tag, it)] = x
        assert clusters is not None
        out[tag + '_clusters_ptr'] = np.cumsum([0] + [len(c) for c in clusters]).astype(np.int32)
        out[tag + '_clusters'] = np.concatenate([np.asarray(c, np.int32) for c in clusters])
        print('mcl case', tag, 'n', n, 'iters', len(per_iter), 'clusters', len(clusters))
    np.savez_compressed(path, **out)


class Args:
    pass


def ingest_case(nchrs, chr_len, mean_len, npairs, flank_kb, Nx, bin_size_kb, seed, max_read_pairs):
    g = synth.make_genome(nchrs, chr_len, mean_len, cv=0.5, min_len=2000, seed=seed)
    rng = np.random.default_rng(seed + 1)
    fa_dict = {}
    for nm, ln in zip(g.names, g.length):
        seq = ''.join(rng.choice(list('ACGT'), int(ln)))
        fa_dict[nm] = [seq, int(ln), H.count_RE_sites(seq, 'GATC') + 1]
    _, bin_set, bin_size, frag_len_dict, Nx_frag_set, RE_site_dict, split_ctg_set = H.stat_fragments(
        fa_dict, 'GATC', {}, set(), nchrs=nchrs, flank=flank_kb, Nx=Nx, bin_size=bin_size_kb)
    id1, p1, id2, p2 = [t.numpy() for t in synth.sample_pairs(g, npairs, seed=seed + 2, cis=0.8)]
    # sprinkle alignments to names that are not in the FASTA, and exact duplicates
    id1 = id1.copy(); id2 = id2.copy()
    bad = rng.random(npairs) < 0.01
    id1[bad & (rng.random(npairs) < 0.5)] = -1
    id2[bad & (id1 >= 0)] = -1
    names = list(g.names)

    def nm(i):
        return names[i] if i >= 0 else 'unplaced_scaffold'

    if not split_ctg_set:   # run() feeds the inter-contig generator in that case (:2865)
        keep = id1 != id2
        id1, p1, id2, p2 = id1[keep], p1[keep], id2[keep], p2[keep]
    aln = [(nm(a), nm(b), int(x), int(y)) for a, x, b, y in zip(id1, p1, id2, p2)]
    args = Args()
    args.flank = flank_kb
    args.remove_allelic_links = 4 if max_read_pairs else 0
    args.remove_concentrated_links = False
    args.max_read_pairs = max_read_pairs
    args.nwindows = 50
    captured = {}
    orig_ccr = H.cal_concordance_ratio
    H.cal_concordance_ratio = lambda coord_list, shorter_len, nwindows: tuple(coord_list)
    try:
        if split_ctg_set:
            full, flank, HT, clm, frag_link, coord, c2f = H.parse_alignments(
                iter(aln), fa_dict, args, bin_size, frag_len_dict, Nx_frag_set, split_ctg_set, 'int32', 'int32')
        else:
            c2f = None
            full, flank, HT, clm, frag_link, coord = H.parse_alignments_for_ctgs(
                iter(aln), fa_dict, args, frag_len_dict, Nx_frag_set, 'int32', 'int32')
    finally:
        H.cal_concordance_ratio = orig_ccr
    # ---- integer views
    cid = {n_: i for i, n_ in enumerate(names)}
    frag_names, ctg_frag0, ctg_split = [], [], []
    for n_ in names:
        ctg_frag0.append(len(frag_names))
        if n_ in split_ctg_set:
            ctg_split.append(1)
            nb = int(np.ceil(fa_dict[n_][1] / bin_size))
            frag_names += ['{}_bin{}'.format(n_, k + 1) for k in range(nb)]
        else:
            ctg_split.append(0)
            frag_names.append(n_)
    fid = {n_: i for i, n_ in enumerate(frag_names)}

    def rank(lst):
        order = sorted(range(len(lst)), key=lambda i: lst[i])
        r = np.empty(len(lst), np.int32)
        r[order] = np.arange(len(lst), dtype=np.int32)
        return r
    out = dict(
        names=np.array(names), frag_names=np.array(frag_names),
        ctg_rank=rank(names), ctg_len=np.array([fa_dict[n_][1] for n_ in names], np.int64),
        ctg_frag0=np.array(ctg_frag0, np.int32), ctg_split=np.array(ctg_split, np.uint8),
        bin_size=np.int64(bin_size if split_ctg_set else 0),
        frag_rank=rank(frag_names), frag_len=np.array([frag_len_dict[f] for f in frag_names], np.int64),
        frag_nx=np.array([f in Nx_frag_set for f in frag_names], np.uint8),
        flank=np.int64(flank_kb * 1000), bins=np.int32(bool(split_ctg_set)),
        max_read_pairs=np.int32(max_read_pairs),
        id1=id1.astype(np.int32), pos1=p1.astype(np.int64), id2=id2.astype(np.int32), pos2=p2.astype(np.int64),
        full_i=np.array([cid[k[0]] for k in full], np.int32), full_j=np.array([cid[k[1]] for k in full], np.int32),
        full_cnt=np.array(list(full.values()), np.int64),
        flank_i=np.array([fid[k[0]] for k in flank], np.int32), flank_j=np.array([fid[k[1]] for k in flank], np.int32),
        flank_cnt=np.array(list(flank.values()), np.int64),
        frag_links=np.array([frag_link.get(f, 0) for f in frag_names], np.int64),
    )
    ht = np.zeros((len(full), 4), np.int64)
    kidx = {k: i for i, k in enumerate(full)}
    for (a, b), c in HT.items():
        k = kidx[(a[:-2], b[:-2])]
        ht[k, (a[-1] == 'T') * 2 + (b[-1] == 'T')] = c
    out['ht_cnt'] = ht
    # insertion order of HT_link_dict: (row of full_link_dict, quadrant) per key
    out['ht_order'] = np.array([(kidx[(a[:-2], b[:-2])], (a[-1] == 'T') * 2 + (b[-1] == 'T')) for a, b in HT], np.int32).reshape(-1, 2)
    clm_ptr, clm_all = [0], []
    for k in full:
        clm_all += list(clm[k])
        clm_ptr.append(len(clm_all))
    out['clm_ptr'] = np.array(clm_ptr, np.int64)
    out['clm'] = np.array(clm_all, np.int64)
    if max_read_pairs:
        crd_ptr, crd_all = [0], []
        for k in full:
            v = coord[k]
            v = list(v[0]) if isinstance(v, list) else list(v)
            crd_all += v
            crd_ptr.append(len(crd_all))
        out['crd_ptr'] = np.array(crd_ptr, np.int64)
        out['crd'] = np.array(crd_all, np.int64)
    if c2f is not None:      # ctg_pair_to_frag :1731-1733 -> (ctg_i, ctg_j, frag_i, frag_j) rows, sorted
        rows = sorted((cid[ck[0]], cid[ck[1]], fid[fk[0]], fid[fk[1]]) for ck, fs in c2f.items() for fk in fs)
        out['c2f'] = np.array(rows, np.int32).reshape(-1, 4)
    # ---- dict_to_matrix on a filtered fragment set (drop ~15 % of the Nx set, keep some link-less)
    fset = set(f for f in Nx_frag_set if rng.random() > 0.15)
    mat, fidx = H.dict_to_matrix(flank, fset, dense_matrix=False, add_self_loops=True)
    p, j, x = canon(mat)
    out['d2m_in_set'] = np.array([f in fset for f in frag_names], np.uint8)
    out['d2m_p'], out['d2m_j'], out['d2m_x'] = p, j, x
    out['d2m_frag_index'] = np.array([fidx.get(f, -1) for f in frag_names], np.int32)
    # normalize_by_nlinks variant (:718-724): values become python floats before the float32 cast
    flank2 = dict(flank)
    H.normalize_by_nlinks(flank2, frag_link)
    mat2, _ = H.dict_to_matrix(flank2, fset, dense_matrix=False, add_self_loops=True)
    out['d2m_nlinks_x'] = canon(mat2)[2]
    print('ingest case: ctgs', len(names), 'frags', len(frag_names), 'pairs', len(aln), 'full keys', len(full),
          'flank keys', len(flank), 'bins' if split_ctg_set else 'ctgs', 'matrix', mat.shape, mat.nnz)
    return out


def gen_ingest(path_ctgs, path_bins):
    np.savez_compressed(path_ctgs, **ingest_case(3, 600_000, 12_000, 30_000, 3, 80, 0, 11, 5))
    np.savez_compressed(path_bins, **ingest_case(2, 500_000, 25_000, 30_000, 2, 90, 16, 21, 4))


def gen_pipeline(path):
    """ingest -> dict_to_matrix -> run_mcl_clustering on a ~C1-shaped toy: expected cluster files."""
    import tempfile
    g = synth.make_genome(4, 1_500_000, 30_000, cv=0.3, min_len=5000, seed=5)
    names = list(g.names)
    fa_dict = {n_: [None, int(l), int(r)] for n_, l, r in zip(names, g.length, g.re_sites)}
    id1, p1, id2, p2 = [t.numpy() for t in synth.sample_pairs(g, 120_000, seed=6, cis=0.9)]
    keep = id1 != id2
    id1, p1, id2, p2 = id1[keep], p1[keep], id2[keep], p2[keep]
    aln = ((names[a], names[b], int(x), int(y)) for a, x, b, y in zip(id1, p1, id2, p2))
    args = Args()
    args.flank = 500
    args.remove_allelic_links = 0
    args.remove_concentrated_links = False
    args.max_read_pairs = 200
    args.nwindows = 50
    frag_len_dict = {n_: fa_dict[n_][1] for n_ in names}
    Nx_set = set(names)
    full, flank, HT, clm, frag_link, coord = H.parse_alignments_for_ctgs(
        aln, fa_dict, args, frag_len_dict, Nx_set, 'int32', 'int32')
    # filtered_frags: python set built the way filter_fragments() leaves it is order-irrelevant for
    # linked fragments; we keep every contig (neutral filters) and store the link-less order used.
    mat, fidx = H.dict_to_matrix(flank, Nx_set, dense_matrix=False, add_self_loops=True)
    cwd = os.getcwd()
    out = dict(names=np.array(names), length=g.length, re_sites=g.re_sites,
               id1=id1.astype(np.int16) if len(names) < 32768 else id1, pos1=p1, id2=id2.astype(np.int16) if len(names) < 32768 else id2, pos2=p2,
               frag_index=np.array([fidx[n_] for n_ in names], np.int32), nchrs=np.int32(4))
    with tempfile.TemporaryDirectory() as td:
        os.chdir(td)
        try:
            H.logger.setLevel('INFO')
            import logging
            fh = logging.FileHandler('log.txt', 'w')
            H.logger.addHandler(fh)
            res, nrounds = H.run_mcl_clustering(mat, set(), frag_len_dict, fidx, 2, 1.2, 2.0, 0.4, 200, 1e-4,
                                                fa_dict, 4, False)
            H.logger.removeHandler(fh)
            fh.close()
            H.logger.setLevel('WARNING')
            infl = []
            for d in sorted(os.listdir('.')):
                if d.startswith('inflation_'):
                    infl.append(d.split('_', 1)[1])
                    out['clusters_txt_' + infl[-1]] = np.array(open('{0}/mcl_{0}.clusters.txt'.format(d)).read())
                    groups = sorted(f for f in os.listdir(d) if f.startswith('group'))
                    out['group_files_' + infl[-1]] = np.array(groups)
                    out['group0_txt_' + infl[-1]] = np.array(open(os.path.join(d, groups[0])).read())
            out['inflations'] = np.array(infl)
            out['log_recommend'] = np.array([l.strip() for l in open('log.txt') if 'You could try' in l] or [''])
        finally:
            os.chdir(cwd)
    print('pipeline case: ctgs', len(names), 'pairs', len(id1), 'inflations', infl, 'matrix', mat.shape, mat.nnz,
          'recommend:', out['log_recommend'])
    np.savez_compressed(path, **out)


def _run_mcl_files(mat, bin_set, frag_len_dict, fidx, fa_dict, nchrs, infl_range, out):
    """run the reference's run_mcl_clustering in a temp dir and freeze every file it writes"""
    import logging
    import tempfile
    cwd = os.getcwd()
    with tempfile.TemporaryDirectory() as td:
        os.chdir(td)
        try:
            H.logger.setLevel('INFO')
            fh = logging.FileHandler('log.txt', 'w')
            H.logger.addHandler(fh)
            H.run_mcl_clustering(mat, bin_set, frag_len_dict, fidx, 2, infl_range[0], infl_range[1], infl_range[2], 200, 1e-4,
                                 fa_dict, nchrs, False)
            H.logger.removeHandler(fh)
            fh.close()
            H.logger.setLevel('WARNING')
            infl = []
            for d in sorted(os.listdir('.')):
                if d.startswith('inflation_'):
                    infl.append(d.split('_', 1)[1])
                    out['clusters_txt_' + infl[-1]] = np.array(open('{0}/mcl_{0}.clusters.txt'.format(d)).read())
                    groups = sorted(f for f in os.listdir(d) if f.startswith('group'))
                    out['group_files_' + infl[-1]] = np.array(groups)
                    out['group_txt_' + infl[-1]] = np.array([open(os.path.join(d, g)).read() for g in groups])
            out['inflations'] = np.array(infl)
            out['log_recommend'] = np.array([l.strip() for l in open('log.txt') if 'You could try' in l] or [''])
        finally:
            os.chdir(cwd)
    return infl


def gen_pipeline_bins(path):
    """contigs longer than bin_size are split (parse_alignments :1658-1752, bin -> contig vote :2172-2194)"""
    g = synth.make_genome(3, 2_000_000, 250_000, cv=0.5, min_len=20_000, seed=31)
    rng = np.random.default_rng(32)
    names = list(g.names)
    fa_dict = {}
    for nm, ln in zip(names, g.length):
        seq = ''.join(rng.choice(list('ACGT'), int(ln)))
        fa_dict[nm] = [seq, int(ln), H.count_RE_sites(seq, 'GATC') + 1]
    _, bin_set, bin_size, frag_len_dict, Nx_frag_set, RE_site_dict, split_ctg_set = H.stat_fragments(
        fa_dict, 'GATC', {}, set(), nchrs=3, flank=50, Nx=100, bin_size=100)
    id1, p1, id2, p2 = [t.numpy() for t in synth.sample_pairs(g, 150_000, seed=33, cis=0.9)]
    aln = ((names[a], names[b], int(x), int(y)) for a, x, b, y in zip(id1, p1, id2, p2))
    args = Args()
    args.flank = 50
    args.remove_allelic_links = 0
    args.remove_concentrated_links = False
    args.max_read_pairs = 200
    args.nwindows = 50
    full, flank, HT, clm, frag_link, coord, _ = H.parse_alignments(
        aln, fa_dict, args, bin_size, frag_len_dict, Nx_frag_set, split_ctg_set, 'int32', 'int32')
    mat, fidx = H.dict_to_matrix(flank, Nx_frag_set, dense_matrix=False, add_self_loops=True)
    frag_names = list(frag_len_dict)
    out = dict(names=np.array(names), length=g.length, re_sites=np.array([fa_dict[n_][2] for n_ in names], np.int64),
               bin_size=np.int64(bin_size), split=np.array([n_ in split_ctg_set for n_ in names], np.uint8),
               frag_names=np.array(frag_names), frag_len=np.array([frag_len_dict[f] for f in frag_names], np.int64),
               frag_nx=np.array([f in Nx_frag_set for f in frag_names], np.uint8),
               frag_is_bin=np.array([f in bin_set for f in frag_names], np.uint8),
               id1=id1.astype(np.int16), pos1=p1, id2=id2.astype(np.int16), pos2=p2,
               frag_index=np.array([fidx.get(f, -1) for f in frag_names], np.int32), nchrs=np.int32(3),
               n_full=np.int64(len(full)), n_flank=np.int64(len(flank)), full_total=np.int64(sum(full.values())))
    for n_ in names:
        fa_dict[n_][0] = None
    infl = _run_mcl_files(mat, bin_set, frag_len_dict, fidx, fa_dict, 3, (1.2, 2.4, 0.4), out)
    print('pipeline bins case: ctgs', len(names), 'frags', len(frag_names), 'split', len(split_ctg_set), 'pairs', len(id1),
          'inflations', infl, 'matrix', mat.shape, mat.nnz, 'recommend:', out['log_recommend'])
    np.savez_compressed(path, **out)


def gen_pipeline_c1(path):
    """BASELINE.json configs[0]: ~1k contigs (4 chr x 25 Mb, mean 100 kb), 1 M pairs, nchrs = 4.  The pairs are
    NOT stored: the test regenerates them with synth.sample_pairs(seed) (torch CPU generator, same image)."""
    g = synth.make_genome(4, 25_000_000, 100_000, cv=0.3, min_len=5000, seed=12345)
    names = list(g.names)
    fa_dict = {n_: [None, int(l), int(r)] for n_, l, r in zip(names, g.length, g.re_sites)}
    id1, p1, id2, p2 = [t.numpy() for t in synth.sample_pairs(g, 1_000_000, seed=12345)]
    keep = id1 != id2
    aln = ((names[a], names[b], int(x), int(y)) for a, x, b, y in zip(id1[keep], p1[keep], id2[keep], p2[keep]))
    args = Args()
    args.flank = 500
    args.remove_allelic_links = 0
    args.remove_concentrated_links = False
    args.max_read_pairs = 200
    args.nwindows = 50
    frag_len_dict = {n_: fa_dict[n_][1] for n_ in names}
    Nx_set = set(names)
    full, flank, HT, clm, frag_link, coord = H.parse_alignments_for_ctgs(aln, fa_dict, args, frag_len_dict, Nx_set, 'int32', 'int32')
    mat, fidx = H.dict_to_matrix(flank, Nx_set, dense_matrix=False, add_self_loops=True)
    out = dict(n_contigs=np.int64(len(names)), pairs_checksum=np.int64(int(id1.sum() + p1.sum() + id2.sum() + p2.sum())),
               n_full=np.int64(len(full)), n_flank=np.int64(len(flank)), full_total=np.int64(sum(full.values())),
               frag_index=np.array([fidx[n_] for n_ in names], np.int32), matrix_nnz=np.int64(mat.nnz), nchrs=np.int32(4))
    infl = _run_mcl_files(mat, set(), frag_len_dict, fidx, fa_dict, 4, (1.4, 2.2, 0.4), out)
    print('pipeline C1 case: ctgs', len(names), 'pairs', int(keep.sum()), 'keys', len(full), 'inflations', infl,
          'matrix', mat.shape, mat.nnz, 'recommend:', out['log_recommend'])
    np.savez_compressed(path, **out)


C4 = dict(nchrs=3, chr_len=3_000_000, mean_len=60_000, ploidy=4, npairs=400_000, allelic=0.08, seed=4040)


def c4_inputs():
    """BASELINE.json configs[3] at test scale: autotetraploid (4 collinear haplotypes), cis power-law pairs inside a
    haplotype + 8 % allelic contacts between homologous positions; shared by the generator and the GPU test"""
    base = synth.make_genome(C4['nchrs'], C4['chr_len'], C4['mean_len'], cv=0.3, min_len=8000, seed=C4['seed'])
    g = synth.make_polyploid(base, C4['ploidy'])
    id1, p1, id2, p2 = [t.numpy() for t in synth.sample_pairs(g, C4['npairs'], seed=C4['seed'] + 1, cis=0.9)]
    id1, p1, id2, p2 = synth.add_allelic_pairs(g, base.n, C4['ploidy'], id1, p1, id2, p2, C4['allelic'], C4['seed'] + 2)
    keep = id1 != id2                                   # run() feeds the inter-contig generator (:2865)
    return g, id1[keep], p1[keep], id2[keep], p2[keep]


def gen_pipeline_c4(path):
    """--remove_allelic_links 4 on an autotetraploid: parse_alignments_for_ctgs -> remove_allelic_HiC_links (:474-689,
    networkx cliques + Hungarian matching, stays the reference's Python) -> dict_to_matrix -> run_mcl_clustering.
    Frozen: ctg_coord_dict as remove_allelic_HiC_links receives it (key order, collapsed [ratio, 1] entries,
    raw coordinate lists), which keys and fragments it removed, and the cluster files."""
    g, id1, p1, id2, p2 = c4_inputs()
    names = list(g.names)
    fa_dict = {n_: [None, int(l), int(r)] for n_, l, r in zip(names, g.length, g.re_sites)}
    aln = ((names[a], names[b], int(x), int(y)) for a, x, b, y in zip(id1, p1, id2, p2))
    args = Args()
    args.flank = 500
    args.remove_allelic_links = 4
    args.remove_concentrated_links = False
    args.max_read_pairs = 40
    args.min_read_pairs = 20
    args.concordance_ratio_cutoff = 0.2
    args.nwindows = 50
    frag_len_dict = {n_: fa_dict[n_][1] for n_ in names}
    Nx_set = set(names)
    full, flank, HT, clm, frag_link, coord = H.parse_alignments_for_ctgs(aln, fa_dict, args, frag_len_dict, Nx_set, 'int32', 'int32')
    cid = {n_: i for i, n_ in enumerate(names)}
    ckeys = list(coord)
    collapsed = np.array([isinstance(coord[k], list) for k in ckeys], bool)
    ratio = np.array([coord[k][0] if isinstance(coord[k], list) else -1.0 for k in ckeys], np.float64)
    raw_ptr, raw = [0], []
    for k in ckeys:
        if not isinstance(coord[k], list):
            raw += list(coord[k])
        raw_ptr.append(len(raw))
    pre_full, pre_flank = list(full), list(flank)
    remaining = H.remove_allelic_HiC_links(fa_dict, coord, full, args, flank, set(names))
    out = dict(pairs_checksum=np.int64(int(id1.sum() + p1.sum() + id2.sum() + p2.sum())), n_contigs=np.int64(len(names)),
               coord_i=np.array([cid[k[0]] for k in ckeys], np.int32), coord_j=np.array([cid[k[1]] for k in ckeys], np.int32),
               coord_collapsed=collapsed, coord_ratio=ratio, coord_raw_ptr=np.array(raw_ptr, np.int64), coord_raw=np.array(raw, np.int64),
               n_full=np.int64(len(pre_full)), n_flank=np.int64(len(pre_flank)),
               full_removed=np.array([k not in full for k in pre_full], bool),
               flank_removed=np.array([k not in flank for k in pre_flank], bool),
               remaining=np.array([n_ in remaining for n_ in names], bool), nchrs=np.int32(C4['nchrs'] * C4['ploidy']))
    mat, fidx = H.dict_to_matrix(flank, remaining, dense_matrix=False, add_self_loops=True)
    out['frag_index'] = np.array([fidx.get(n_, -1) for n_ in names], np.int32)
    infl = _run_mcl_files(mat, set(), frag_len_dict, fidx, fa_dict, int(out['nchrs']), (1.4, 2.6, 0.4), out)
    print('pipeline c4 case: ctgs', len(names), 'pairs', len(id1), 'coord keys', len(ckeys), 'collapsed', int(collapsed.sum()),
          'full removed', int(out['full_removed'].sum()), 'flank removed', int(out['flank_removed'].sum()),
          'fragments kept', int(out['remaining'].sum()), 'matrix', mat.shape, mat.nnz, 'recommend:', out['log_recommend'])
    np.savez_compressed(path, **out)


def gen_pipeline_c4_40k(path, stage_dir='/tmp/c4_40k'):
    """BASELINE.json configs[3] AT ITS STATED SIZE (VERDICT r03 #1a): the reference on a 40k-contig autotetraploid —
    parse_alignments_for_ctgs :1596 -> remove_allelic_HiC_links :474-689 -> dict_to_matrix :310 -> run_mcl_clustering :2132 at four
    inflations.  Frozen: digests of ctg_coord_dict as the filter receives it, ITS VERDICT (which keys of full / flank_link_dict
    and which fragments go: bit masks in dict order), the index map, SHA-256 of every file run_mcl_clustering writes and the
    integer contig -> group map of every inflation.  ~1-2 h of one core, ~40 GB; stages are check-pointed under stage_dir."""
    import gc
    import hashlib
    import time
    from tests import c4_40k
    cfg = c4_40k.CFG
    os.makedirs(stage_dir, exist_ok=True)
    t0 = time.time()
    gen, base, id1, p1, id2, p2 = c4_40k.inputs()
    names = list(gen.names)
    cid = {n_: i for i, n_ in enumerate(names)}
    fa_dict = {n_: [None, int(l), int(r)] for n_, l, r in zip(names, gen.length, gen.re_sites)}
    frag_len_dict = {n_: fa_dict[n_][1] for n_ in names}
    out = dict(pairs_checksum=np.int64(c4_40k.checksum(id1, p1, id2, p2)), n_contigs=np.int64(len(names)), n_pairs=np.int64(len(id1)),
               nchrs=np.int32(cfg['nchrs'] * cfg['ploidy']))
    print('c4_40k: contigs', len(names), 'inter-contig pairs', len(id1), flush=True)
    args = Args()
    args.flank = cfg['flank']
    args.remove_allelic_links = cfg['ploidy']
    args.remove_concentrated_links = False
    args.max_read_pairs = cfg['max_read_pairs']
    args.min_read_pairs = cfg['min_read_pairs']
    args.concordance_ratio_cutoff = cfg['concordance_ratio_cutoff']
    args.nwindows = cfg['nwindows']
    aln = ((names[a], names[b], x, y) for a, x, b, y in zip(id1.tolist(), p1.tolist(), id2.tolist(), p2.tolist()))
    full, flank, HT, clm, frag_link, coord = H.parse_alignments_for_ctgs(aln, fa_dict, args, frag_len_dict, set(names), 'int32', 'int32')
    del id1, p1, id2, p2, aln, clm, HT
    gc.collect()
    print('c4_40k: parsed, keys', len(full), len(flank), 'coord', len(coord), '%.0f s' % (time.time() - t0), flush=True)
    out.update({k: np.array(v) for k, v in c4_40k.coord_digest(coord, cid).items()})
    out['n_full'], out['n_flank'] = np.int64(len(full)), np.int64(len(flank))
    out['full_total'] = np.int64(sum(full.values()))
    pre_full, pre_flank = list(full), list(flank)
    remaining = H.remove_allelic_HiC_links(fa_dict, coord, full, args, flank, set(names))
    del coord
    out['full_removed'] = np.packbits(np.fromiter((k not in full for k in pre_full), bool, len(pre_full)))
    out['flank_removed'] = np.packbits(np.fromiter((k not in flank for k in pre_flank), bool, len(pre_flank)))
    out['remaining'] = np.array([n_ in remaining for n_ in names], bool)
    print('c4_40k: allelic filter done: full', len(pre_full), '->', len(full), 'flank', len(pre_flank), '->', len(flank),
          'fragments kept', len(remaining), '%.0f s' % (time.time() - t0), flush=True)
    del pre_full, pre_flank, full
    gc.collect()
    mat, fidx = H.dict_to_matrix(flank, remaining, dense_matrix=False, add_self_loops=True)
    del flank
    gc.collect()
    out['frag_index'] = np.array([fidx.get(n_, -1) for n_ in names], np.int32)
    out['matrix_nnz'] = np.int64(mat.nnz)
    out['matrix_sha'] = np.array(hashlib.sha256(b''.join(np.ascontiguousarray(a).tobytes() for a in canon(mat))).hexdigest())
    np.savez_compressed(os.path.join(stage_dir, 'stage1.npz'), **out)
    print('c4_40k: matrix', mat.shape, mat.nnz, '%.0f s' % (time.time() - t0), flush=True)
    import logging
    cwd = os.getcwd()
    work = os.path.join(stage_dir, 'run')
    os.makedirs(work, exist_ok=True)
    os.chdir(work)
    try:
        H.logger.setLevel('INFO')
        fh = logging.FileHandler('log.txt', 'w')
        H.logger.addHandler(fh)
        lo, hi, step = cfg['inflations']
        H.run_mcl_clustering(mat, set(), frag_len_dict, fidx, 2, lo, hi, step, 200, 1e-4, fa_dict, int(out['nchrs']), False)
        H.logger.removeHandler(fh)
        fh.close()
        H.logger.setLevel('WARNING')
        infl = []
        for d in sorted(os.listdir('.')):
            if d.startswith('inflation_'):
                tag = d.split('_', 1)[1]
                infl.append(tag)
                txt = open('{0}/mcl_{0}.clusters.txt'.format(d)).read()
                out['clusters_sha_' + tag] = np.array(hashlib.sha256(txt.encode()).hexdigest())
                out['group_map_' + tag] = c4_40k.group_map(txt, cid)
                groups = sorted(f for f in os.listdir(d) if f.startswith('group'))
                out['group_files_' + tag] = np.array(groups)
                out['group_sha_' + tag] = np.array([c4_40k.file_digest(os.path.join(d, g_)) for g_ in groups])
        out['inflations'] = np.array(infl)
        log = [l.strip() for l in open('log.txt')]
        out['log_recommend'] = np.array([l for l in log if 'You could try' in l] or [''])
        out['log_mcl'] = np.array([l for l in log if 'rounds of iterations' in l])
    finally:
        os.chdir(cwd)
    print('c4_40k: inflations', infl, 'groups', [int(out['group_map_' + t].max()) + 1 for t in infl], 'recommend:', out['log_recommend'],
          '%.0f s' % (time.time() - t0), flush=True)
    np.savez_compressed(path, **out)


def gen_resites(path):
    """count_RE_sites (:75-84) on slices, and stat_fragments (:188-296) on a small assembly"""
    rng = np.random.default_rng(77)
    parts = [''.join(rng.choice(list('ACGT'), 30_000)), 'A' * 1003, 'GC' * 777, ''.join(rng.choice(list('ACGTN'), 5000)),
             'GATC' * 500, ''.join(rng.choice(list('AG'), 4000)), 'GANTCGAATC' * 50, ''.join(rng.choice(list('ACGT'), 9000))]
    seq = ''.join(parts)
    n = len(seq)
    off = rng.integers(0, n, 400)
    ln = np.minimum(rng.integers(0, 9000, 400), n - off)
    off = np.concatenate([off, [0, 0, n - 3, n, 5]]).astype(np.int64)
    ln = np.concatenate([ln, [n, 0, 3, 0, 4097]]).astype(np.int64)
    out = dict(seq=np.frombuffer(seq.encode(), np.uint8), seg_off=off, seg_len=ln)
    res = []
    for RE in ('GATC', 'GATC,GANTC', 'AAGCTT', 'GCGC', 'AAAA,GATC', 'A', 'GANNTC', 'ACGTACGTACGTACGTACGTACGTACGTACGT'):
        res.append(RE)
        out['counts_' + RE] = np.array([H.count_RE_sites(seq[a:a + l], RE) for a, l in zip(off, ln)], np.int64)
        out['sites_' + RE] = np.array(H.parse_RE_sites([x.strip().upper() for x in RE.split(',') if x.strip()]))
    out['REs'] = np.array(res)
    # stat_fragments: 14 contigs, flank-only counting, bins, Nx < 100, whitelist
    fa = {}
    names = []
    for i in range(14):
        L = int(rng.integers(3000, 90_000))
        sq = ''.join(rng.choice(list('ACGT'), L))
        nm = 'ptg%03dl' % i
        names.append(nm)
        fa[nm] = [sq, L, H.count_RE_sites(sq, 'GATC,GANTC') + 1]
    out['sf_names'] = np.array(names)
    out['sf_seq'] = np.frombuffer(''.join(fa[n_][0] for n_ in names).encode(), np.uint8)
    out['sf_len'] = np.array([fa[n_][1] for n_ in names], np.int64)
    out['sf_re'] = np.array([fa[n_][2] for n_ in names], np.int64)
    r = H.stat_fragments(fa, 'GATC,GANTC', {}, {'ptg003l'}, nchrs=2, flank=4, Nx=70, bin_size=20)
    out['sf_sorted'] = np.array([f for f, _ in r[0]])
    out['sf_bin_size'] = np.int64(r[2])
    out['sf_frags'] = np.array(list(r[3]))
    out['sf_frag_len'] = np.array(list(r[3].values()), np.int64)
    out['sf_nx'] = np.array(sorted(r[4]))
    out['sf_re_frags'] = np.array(list(r[5]))
    out['sf_re_counts'] = np.array(list(r[5].values()), np.int64)
    out['sf_split'] = np.array(sorted(r[6]))
    out['sf_bins'] = np.array(sorted(r[1]))
    print('resites case: seq', n, 'segments', len(off), 'REs', res, 'stat_fragments frags', len(r[3]), 'bins', len(r[1]))
    np.savez_compressed(path, **out)


def gen_filter(path):
    """filter_fragments (:741-940) on a 230-contig assembly with some link-poor / promiscuous contigs"""
    g = synth.make_genome(4, 1_600_000, 28_000, cv=0.4, min_len=4000, seed=41)
    names = list(g.names)
    rng = np.random.default_rng(42)
    fa_dict = {n_: [None, int(l), int(r)] for n_, l, r in zip(names, g.length, g.re_sites)}
    id1, p1, id2, p2 = [t.numpy() for t in synth.sample_pairs(g, 90_000, seed=43, cis=0.75)]
    keep = id1 != id2
    id1, p1, id2, p2 = id1[keep], p1[keep], id2[keep], p2[keep]
    aln = ((names[a], names[b], int(x), int(y)) for a, x, b, y in zip(id1, p1, id2, p2))
    args = Args()
    args.flank = 500
    args.remove_allelic_links = 0
    args.remove_concentrated_links = False
    args.max_read_pairs = 200
    args.nwindows = 50
    frag_len_dict = {n_: fa_dict[n_][1] for n_ in names}
    Nx_set = set(names[:-7])                                   # a few contigs outside the Nx set
    full, flank, HT, clm, frag_link, coord = H.parse_alignments_for_ctgs(aln, fa_dict, args, frag_len_dict, Nx_set, 'int32', 'int32')
    RE_site_dict = {n_: fa_dict[n_][2] for n_ in names}
    cid = {n_: i for i, n_ in enumerate(names)}
    out = dict(names=np.array(names), nx=np.array([n_ in Nx_set for n_ in names], np.uint8),
               re_sites=np.array([RE_site_dict[n_] for n_ in names], np.int64),
               frag_links=np.array([frag_link.get(n_, -1) for n_ in names], np.int64),
               flank_i=np.array([cid[k[0]] for k in flank], np.int32), flank_j=np.array([cid[k[1]] for k in flank], np.int32),
               flank_cnt=np.array(list(flank.values()), np.int64))
    cases = [(5, '0.2X', '1.9X', 10, '1.5X', 0, None), (3, '0.1X', '3X', 5, '0.5X', 0, {names[3], names[40]}),
             (5, '0.2X', '1.9X', 10, '3X', 600, None), (0, '0X', '100X', 10, '100X', 0, None)]
    H.logger.setLevel('WARNING')
    for k, (cut, lo, up, topn, rsu, hard, wl) in enumerate(cases):
        res = H.filter_fragments(set(Nx_set), RE_site_dict, cut, frag_link, lo, up, topn, rsu, hard, flank, {}, '1.5X', wl)
        out['case%d_params' % k] = np.array([str(cut), lo, up, str(topn), rsu, str(hard)])
        out['case%d_whitelist' % k] = np.array(sorted(wl) if wl else [], dtype=str)
        out['case%d_kept' % k] = np.array(sorted(res))
        print('filter case', k, 'kept', len(res), 'of', len(Nx_set))
    out['n_cases'] = np.int32(len(cases))
    np.savez_compressed(path, **out)


def gen_coord_stats(path):
    """cal_concordance_ratio :419-428 and cal_concentration_adj_ratio :431-451 on random coordinate lists (the host
    statistics behind --remove_allelic_links / --remove_concentrated_links; host mirrors in cluster.py)"""
    from array import array
    rng = np.random.default_rng(91)
    lists, lens, conc, adj = [], [], [], []
    for k in range(60):
        npairs = int(rng.integers(3, 400))
        shorter = int(rng.integers(20_000, 3_000_000))
        if k % 3 == 0:      # collinear (allelic-looking) pairs with noise
            x = rng.integers(0, shorter, npairs)
            y = np.clip(x + rng.integers(-shorter // 40, shorter // 40 + 1, npairs), 0, None)
        elif k % 3 == 1:    # concentrated in one bin
            x = rng.integers(0, shorter, npairs); y = rng.integers(0, shorter, npairs)
            x[: npairs // 2] = x[0] // 10000 * 10000 + rng.integers(0, 10000, npairs // 2)
        else:
            x = rng.integers(0, shorter, npairs); y = rng.integers(0, shorter, npairs)
        c = np.stack([x, y], 1).reshape(-1).astype(np.int32)
        lists.append(c); lens.append(shorter)
        conc.append(float(H.cal_concordance_ratio(array('i', c.tolist()), shorter, 50)))
        adj.append(float(H.cal_concentration_adj_ratio(array('i', c.tolist()))))
    ptr = np.cumsum([0] + [len(c) for c in lists])
    np.savez_compressed(path, ptr=ptr, coords=np.concatenate(lists), shorter=np.array(lens, np.int64),
                        concordance=np.array(conc), adj=np.array(adj))
    print('coord stats case:', len(lists), 'lists')


def gen_pairs_text(path):
    """a1: pairs_generator / pairs_generator_inter_ctgs (:1539-1583) on a .pairs text with every line shape the
    tokeniser has to get right: header and blank lines, CRLF and lone-CR line ends, space / tab / \\x0b / \\x1c
    separators, leading whitespace, extra columns, names that are not in the FASTA, signs, zero padding and
    underscores in the integers, no newline at the end.  Frozen: the text, the names, the yielded tuples (as ids)
    and the bytes of alignments.bed."""
    import tempfile
    rng = np.random.default_rng(77)
    names = ['ctg%d' % k for k in range(40)] + ['Chr1_7_100_200_+_100', 'a', 'ab', 'ctg1x', 'utg000001l|arrow']
    pool = names + ['unplaced_scaffold', 'ctg', 'ctg400', 'A']
    lines = ['## pairs format v1.0', '#columns: readID chr1 pos1 chr2 pos2 strand1 strand2', '']
    for k in range(3000):
        a, b = pool[rng.integers(len(pool))], pool[rng.integers(len(pool))]
        if rng.random() < 0.2:
            b = a
        x, y = int(rng.integers(1, 2_000_000_000)), int(rng.integers(1, 5_000_000))
        sx, sy = str(x), str(y)
        r = rng.random()
        if r < 0.03:
            sx = '+' + sx
        elif r < 0.06:
            sx = '00' + sx
        elif r < 0.09 and len(sy) > 3:
            sy = sy[:2] + '_' + sy[2:]
        elif r < 0.10:
            sx = '0'                              # -> pos -1
        elif r < 0.11:
            sx = '-12'
        sep = '\t' if rng.random() < 0.9 else [' ', '  ', '\t ', '\x0b', '\x1c', '\x0c'][rng.integers(6)]
        cols = ['read%d' % k, a, sx, b, sy]
        if rng.random() < 0.7:
            cols += ['+', '-']
        if rng.random() < 0.05:
            cols += ['extra col']
        line = sep.join(cols)
        if rng.random() < 0.03:
            line = '  ' + line
        if rng.random() < 0.03:
            line += ' \t'
        lines.append(line)
        if rng.random() < 0.02:
            lines.append(['', '   ', '#comment in the middle', '\t'][rng.integers(4)])
    text = ''
    for ln in lines:
        r = rng.random()
        text += ln + ('\r\n' if r < 0.1 else '\r' if r < 0.13 else '\n')
    text += 'lastread\tctg3\t77\tctg9\t99'     # no line end at EOF
    raw = text.encode()
    out = dict(text=np.frombuffer(raw, np.uint8), names=np.array(names))
    cid = {n_: i for i, n_ in enumerate(names)}
    cwd = os.getcwd()
    with tempfile.TemporaryDirectory() as d:
        os.chdir(d)
        try:
            with open('in.pairs', 'wb') as f:
                f.write(raw)
            for tag, gen in (('all', H.pairs_generator), ('inter', H.pairs_generator_inter_ctgs)):
                tuples = list(gen('in.pairs', 'pairs'))
                out[tag] = np.array([(cid.get(a, -1), x, cid.get(b, -1), y) for a, b, x, y in tuples], np.int64)
                out[tag + '_known'] = np.array([(a in cid) * 1 + (b in cid) * 2 for a, b, x, y in tuples], np.uint8)
                with open('alignments.bed', 'rb') as f:
                    out['bed_' + tag] = np.frombuffer(f.read(), np.uint8)
        finally:
            os.chdir(cwd)
    assert np.array_equal(out['bed_all'], out['bed_inter'])
    del out['bed_inter']
    print('pairs text case: bytes', len(raw), 'tuples', len(out['all']), 'inter', len(out['inter']), 'bed bytes', len(out['bed_all']))
    np.savez_compressed(path, **out)


def gen_ingest_wide(path):
    """Contigs of 2^31 bp and more: determine_int_type :116-147 switches the reference's coordinate containers to
    int64 there.  An assembly with one 3.0 Gb and one 2.3 Gb contig among ordinary ones (fa_dict entries without sequence: only the
    lengths matter to the parser), a .pairs TEXT whose positions go up to 3e9 read by the reference's own
    pairs_generator_inter_ctgs (alignments.bed included), parse_alignments_for_ctgs with the int types the reference itself
    picks.  Frozen: text, names, lengths, the yielded tuples, alignments.bed, every container of the parser."""
    import tempfile
    rng = np.random.default_rng(4242)
    names = ['chrBig_3Gb', 'ctg_a', 'chrBig_2Gb', 'ctg_b', 'ctg_c', 'ctg_d', 'ctg_e', 'ctg_f']
    lens = [3_000_000_123, 400_000, 2_300_000_000, 90_000, 1_500_000, 35_000, 2_147_483_647, 700_000]
    fa_dict = {n_: [None, l_, max(2, l_ // 256)] for n_, l_ in zip(names, lens)}
    pos_t, dist_t = H.determine_int_type(fa_dict)
    assert (pos_t, dist_t) == ('int64', 'int64')
    lines = ['## pairs format v1.0']
    w = np.array(lens, np.float64) ** 0.5
    w /= w.sum()
    for k in range(6000):
        a, b = rng.choice(len(names), 2, p=w)
        if rng.random() < 0.1:
            b = a
        x = int(rng.integers(1, lens[a] + 1))
        y = int(rng.integers(1, lens[b] + 1))
        if rng.random() < 0.3:                                   # near the ends: flank logic on both sides of 2^31 / 2^32 - flank
            x = int(rng.choice([rng.integers(1, 600_000), lens[a] - rng.integers(0, 600_000)]))
            x = min(max(x, 1), lens[a])
        nm_a = names[a] if rng.random() > 0.02 else 'unplaced'
        lines.append('r%d\t%s\t%d\t%s\t%d\t+\t-' % (k, nm_a, x, names[b], y))
    raw = ('\n'.join(lines) + '\n').encode()
    cid = {n_: i for i, n_ in enumerate(names)}
    args = Args()
    args.flank = 500
    args.remove_allelic_links = 4
    args.remove_concentrated_links = False
    args.max_read_pairs = 6
    args.nwindows = 50
    frag_len_dict = {n_: l_ for n_, l_ in zip(names, lens)}
    out = dict(text=np.frombuffer(raw, np.uint8), names=np.array(names), ctg_len=np.array(lens, np.int64), flank=np.int64(500_000),
               max_read_pairs=np.int32(6), pos_int_type=np.array(pos_t), dist_int_type=np.array(dist_t))
    cwd = os.getcwd()
    orig_ccr = H.cal_concordance_ratio
    H.cal_concordance_ratio = lambda coord_list, shorter_len, nwindows: tuple(coord_list)
    with tempfile.TemporaryDirectory() as d:
        os.chdir(d)
        try:
            with open('in.pairs', 'wb') as f:
                f.write(raw)
            tuples = list(H.pairs_generator_inter_ctgs('in.pairs', 'pairs'))
            with open('alignments.bed', 'rb') as f:
                out['bed'] = np.frombuffer(f.read(), np.uint8)
            full, flank, HT, clm, frag_link, coord = H.parse_alignments_for_ctgs(iter(tuples), fa_dict, args, frag_len_dict, set(names), pos_t, dist_t)
        finally:
            os.chdir(cwd)
            H.cal_concordance_ratio = orig_ccr
    out['tuples'] = np.array([(cid.get(a, -1), x, cid.get(b, -1), y) for a, b, x, y in tuples], np.int64)
    assert out['tuples'][:, [1, 3]].max() > 2 ** 31 and all(v.typecode == 'l' for v in clm.values())
    out.update(full_i=np.array([cid[k[0]] for k in full], np.int32), full_j=np.array([cid[k[1]] for k in full], np.int32),
               full_cnt=np.array(list(full.values()), np.int64),
               flank_i=np.array([cid[k[0]] for k in flank], np.int32), flank_j=np.array([cid[k[1]] for k in flank], np.int32),
               flank_cnt=np.array(list(flank.values()), np.int64), frag_links=np.array([frag_link.get(n_, 0) for n_ in names], np.int64))
    ht = np.zeros((len(full), 4), np.int64)
    kidx = {k: i for i, k in enumerate(full)}
    for (a, b), c in HT.items():
        ht[kidx[(a[:-2], b[:-2])], (a[-1] == 'T') * 2 + (b[-1] == 'T')] = c
    out['ht_cnt'] = ht
    clm_ptr, clm_all, crd_ptr, crd_all = [0], [], [0], []
    for k in full:
        clm_all += list(clm[k])
        clm_ptr.append(len(clm_all))
        v = coord[k]
        crd_all += list(v[0]) if isinstance(v, list) else list(v)
        crd_ptr.append(len(crd_all))
    out.update(clm_ptr=np.array(clm_ptr, np.int64), clm=np.array(clm_all, np.int64), crd_ptr=np.array(crd_ptr, np.int64), crd=np.array(crd_all, np.int64))
    print('wide ingest case: tuples', len(tuples), 'full keys', len(full), 'flank keys', len(flank), 'max position', int(out['tuples'][:, [1, 3]].max()),
          'max CLM distance', int(out['clm'].max()))
    np.savez_compressed(path, **out)


def gen_weights(path):
    """a6: the reference's in-place dict rewrites normalize_by_nlinks :718-724, normalize_by_length :727-738 (dead code),
    reduce_inter_hap_HiC_links :695-707 on a synthetic flank_link_dict; inputs as arrays in dict order, outputs as
    float64 values in the order of the surviving keys (+ which keys survived)."""
    from collections import defaultdict
    rng = np.random.default_rng(61)
    n_frag, n_keys = 400, 6000
    names = ['ctg%04d' % k for k in range(n_frag)]
    a = rng.integers(0, n_frag, 3 * n_keys)
    b = rng.integers(0, n_frag, 3 * n_keys)
    ok = a < b
    key = np.unique(a[ok].astype(np.int64) * n_frag + b[ok])
    key = key[rng.permutation(len(key))][:n_keys]
    fi, fj = (key // n_frag).astype(np.int32), (key % n_frag).astype(np.int32)
    cnt = np.where(rng.random(len(key)) < 0.7, rng.integers(1, 4, len(key)), rng.integers(4, 3000, len(key))).astype(np.int64)
    links = np.zeros(n_frag, np.int64)
    np.add.at(links, fi, cnt)
    np.add.at(links, fj, cnt)
    links += rng.integers(0, 50, n_frag)                       # totals also count links to fragments outside the dict
    links = np.maximum(links, 1)
    length = rng.integers(5_000, 3_000_000, n_frag).astype(np.int64)
    hap = rng.integers(0, 3, n_frag).astype(np.int32)          # 'h0', 'h1', 'h2'
    out = dict(fi=fi, fj=fj, cnt=cnt, links=links, length=length, hap=hap, flank_kb=np.int64(500))

    def as_dict():
        d = defaultdict(int)
        for i, j, c in zip(fi.tolist(), fj.tolist(), cnt.tolist()):
            d[(names[i], names[j])] = c
        return d
    d = as_dict()
    H.normalize_by_nlinks(d, {names[k]: int(links[k]) for k in range(n_frag)})
    out['nlinks'] = np.array(list(d.values()), np.float64)
    d = as_dict()
    H.normalize_by_length(d, {names[k]: int(length[k]) for k in range(n_frag)}, 500)
    out['by_length'] = np.array(list(d.values()), np.float64)
    rdd = {names[k]: ('h%d' % hap[k], 30.0) for k in range(n_frag)}
    order = {(names[i], names[j]): k for k, (i, j) in enumerate(zip(fi.tolist(), fj.tolist()))}
    for tag_, w in (('w1', 1.0), ('w05', 0.5), ('w03', 0.3)):
        d = as_dict()
        H.reduce_inter_hap_HiC_links(d, rdd, w, target='flank_link_dict')
        out['hap_%s_kept' % tag_] = np.array([order[k] for k in d], np.int64)
        out['hap_%s_values' % tag_] = np.array([float(v) for v in d.values()], np.float64)
        out['hap_%s_is_int' % tag_] = np.array([isinstance(v, int) for v in d.values()], bool)
    # after the nlinks normalisation the matrix build casts to float32 (:368): freeze that matrix too
    d = as_dict()
    H.normalize_by_nlinks(d, {names[k]: int(links[k]) for k in range(n_frag)})
    m, fidx = H.dict_to_matrix(d, set(names), dense_matrix=False, add_self_loops=True)
    p_, j_, x_ = canon(m)
    out.update(nl_m_p=p_, nl_m_j=j_, nl_m_x=x_, nl_fidx=np.array([fidx[nm] for nm in names], np.int32))
    print('weights case: keys', len(key), 'deleted at w=1:', len(key) - len(out['hap_w1_kept']))
    np.savez_compressed(path, **out)


def gen_reassign(path):
    """f3: HapHiC_reassign.parse_link_dict :217-263 (the reference's own function) on a synthetic full_link_dict:
    the nested per-(contig, group) sums with BOTH dict orders, and linked_ctg_dict."""
    import HapHiC_reassign as R
    rng = np.random.default_rng(71)
    n_ctg, n_groups, n_keys = 300, 7, 5000
    names = ['c%03d' % k for k in range(n_ctg)]
    a = rng.integers(0, n_ctg, 3 * n_keys)
    b = rng.integers(0, n_ctg, 3 * n_keys)
    ok = a < b
    key = np.unique(a[ok].astype(np.int64) * n_ctg + b[ok])
    key = key[rng.permutation(len(key))][:n_keys]
    fi, fj = (key // n_ctg).astype(np.int32), (key % n_ctg).astype(np.int32)
    links = rng.integers(1, 500, len(key)).astype(np.int64)
    group = rng.integers(-1, n_groups, n_ctg).astype(np.int32)            # -1: 'ungrouped'
    link_dict = {(names[i], names[j]): int(v) for i, j, v in zip(fi.tolist(), fj.tolist(), links.tolist())}
    ctg_group_dict = {names[k]: ('ungrouped' if group[k] < 0 else 'group%d' % group[k]) for k in range(n_ctg)}
    cgl, linked = R.parse_link_dict(link_dict, ctg_group_dict, normalize_by_nlinks=False)
    cid = {n: k for k, n in enumerate(names)}
    rows = [(cid[c], int(g[5:]), int(v)) for c, inner in cgl.items() for g, v in inner.items()]
    out = dict(fi=fi, fj=fj, links=links, group=group, n_groups=np.int64(n_groups), outer=np.array([cid[c] for c in cgl], np.int32),
               cells=np.array(rows, np.int64), linked_ptr=np.cumsum([0] + [len(linked[n]) for n in names]).astype(np.int64),
               linked=np.array([cid[x] for n in names for x in sorted(linked[n])], np.int32))
    print('reassign case: keys', len(key), 'cells', len(rows))
    np.savez_compressed(path, **out)


def gen_plot(path):
    """f4: HapHiC_plot.py parse_agp :41-103, generate_contact_matrix :106-150 and parse_pairs :153-202 (the reference's own
    functions) on the synthetic AGP / .pairs texts of tests/plot_fixture.py.  `portion` is absent here: the interval stand-in
    is oracle/plot_oracle.Closed (closed interval: intersection, containment, .lower / .upper, hashable)."""
    import tempfile
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    import plot_fixture
    from oracle.plot_oracle import Closed
    sys.modules['portion'].closed = lambda a, b: Closed(a, b)
    import HapHiC_plot as P
    P.logger.setLevel('CRITICAL')
    out = {}
    with tempfile.TemporaryDirectory() as tmp:
        for name, (kw, min_len, specified) in plot_fixture.CASES.items():
            case = plot_fixture.make_case(**kw)
            agp, pairs = os.path.join(tmp, name + '.agp'), os.path.join(tmp, name + '.pairs')
            open(agp, 'w').write(case['agp'])
            open(pairs, 'w').write(case['pairs'])
            bin_size = case['bin_size']
            ctg_dict, ctg_aln_dict, group_size_dict, frag_set, group_frag_dict = P.parse_agp(agp, bin_size)
            mat, to_total, group_list, ctg_set = P.generate_contact_matrix(group_size_dict, frag_set, group_frag_dict, bin_size, min_len, specified)
            try:
                mat = P.parse_pairs(pairs, ctg_dict, ctg_aln_dict, bin_size, mat, to_total, group_list, ctg_set)
                error = ''
            except Exception as e:
                error = str(e)
            out[name + '__agp'] = np.frombuffer(case['agp'].encode(), np.uint8)
            out[name + '__pairs'] = np.frombuffer(case['pairs'].encode(), np.uint8)
            out[name + '__matrix'] = np.asarray(mat, np.int64)
            out[name + '__error'] = np.frombuffer(error.encode(), np.uint8)
            out[name + '__groups'] = np.frombuffer(','.join(group_list).encode(), np.uint8)
            out[name + '__ctg_set'] = np.frombuffer(','.join(sorted(ctg_set)).encode(), np.uint8)
            out[name + '__params'] = np.array([bin_size, int(round(min_len * 1000000))], np.int64)
            out[name + '__specified'] = np.frombuffer((specified or '').encode(), np.uint8)
            print('plot case', name, 'bins', mat.shape[0], 'contacts', int(mat.sum()), 'error' if error else '')
    np.savez_compressed(path, **out)


if __name__ == '__main__':
    assert os.environ.get('PYTHONHASHSEED') == '0', 'run with PYTHONHASHSEED=0'
    if len(sys.argv) > 1 and sys.argv[1] == 'c4_40k':         # hours of one core: never part of the default run
        gen_pipeline_c4_40k(os.path.join(HERE, 'pipeline_c4_40k.npz'))
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == 'plot':
        gen_plot(os.path.join(HERE, 'plot.npz'))
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == 'reassign':
        gen_reassign(os.path.join(HERE, 'reassign.npz'))
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == 'wide':
        gen_ingest_wide(os.path.join(HERE, 'ingest_wide.npz'))
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == 'weights':
        gen_weights(os.path.join(HERE, 'weights.npz'))
        sys.exit(0)
    gen_mcl(os.path.join(HERE, 'mcl_cases.npz'))
    gen_ingest(os.path.join(HERE, 'ingest_ctgs.npz'), os.path.join(HERE, 'ingest_bins.npz'))
    gen_pipeline(os.path.join(HERE, 'pipeline_toy.npz'))
    gen_pipeline_bins(os.path.join(HERE, 'pipeline_bins.npz'))
    gen_pipeline_c1(os.path.join(HERE, 'pipeline_c1.npz'))
    gen_resites(os.path.join(HERE, 'resites.npz'))
    gen_filter(os.path.join(HERE, 'filter.npz'))
    gen_pairs_text(os.path.join(HERE, 'pairs_text.npz'))
    gen_coord_stats(os.path.join(HERE, 'coord_stats.npz'))
    gen_pipeline_c4(os.path.join(HERE, 'pipeline_c4.npz'))
    gen_weights(os.path.join(HERE, 'weights.npz'))
    gen_reassign(os.path.join(HERE, 'reassign.npz'))
    gen_plot(os.path.join(HERE, 'plot.npz'))
    gen_ingest_wide(os.path.join(HERE, 'ingest_wide.npz'))
