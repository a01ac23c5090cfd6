"""ctypes front-end of oracle/hhx_oracle.c — CPU restatement of the reference hot path.

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
leg, never by haphic_amd/.  See the header of hhx_oracle.c for the reference citations and how the
oracle is pinned (tests/golden/, generated from the reference's own Python functions).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, '_build', 'libhhx_oracle.so')

_i32p = np.ctypeslib.ndpointer(np.int32, flags='C_CONTIGUOUS')
_i64p = np.ctypeslib.ndpointer(np.int64, flags='C_CONTIGUOUS')
_f32p = np.ctypeslib.ndpointer(np.float32, flags='C_CONTIGUOUS')
_f64p = np.ctypeslib.ndpointer(np.float64, flags='C_CONTIGUOUS')
_u8p = np.ctypeslib.ndpointer(np.uint8, flags='C_CONTIGUOUS')


def build(force=False):
    src = os.path.join(_HERE, 'hhx_oracle.c')
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(['make', '-C', _HERE, '-s'] + (['-B'] if force else []))
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_SO)
        L.orc_normalize_l1.argtypes = [C.c_int32, _i32p, _f32p]
        L.orc_normalize_l1.restype = None
        L.orc_power.argtypes = [C.c_int64, _f32p, C.c_double]
        L.orc_power.restype = None
        L.orc_spgemm.argtypes = [C.c_int32, C.c_int32, _i32p, _i32p, _f32p, _i32p, _i32p, _f32p, _i32p,
                                 C.c_void_p, C.c_void_p, C.c_int, C.c_int]
        L.orc_spgemm.restype = C.c_int64
        L.orc_expand_links.argtypes = [C.c_int32, _i32p, _i32p, _f32p, C.c_int32, C.c_void_p, _i32p, C.c_void_p, C.c_void_p]
        L.orc_expand_links.restype = C.c_int64
        L.orc_expand_links_ex.argtypes = L.orc_expand_links.argtypes + [C.c_int]
        L.orc_expand_links_ex.restype = C.c_int64
        L.orc_links_iteration0.argtypes = [C.c_int32, _i32p, _i32p, _f32p, C.c_int32, C.c_void_p, C.c_double, C.c_double, C.c_int32,
                                           _i32p, _i32p, _f32p, C.POINTER(C.c_int64)]
        L.orc_links_iteration0.restype = C.c_int64
        L.orc_links_shift.argtypes = [C.c_int32, _i32p, _f32p, C.c_void_p]
        L.orc_links_shift.restype = C.c_int
        L.orc_set_threads.argtypes = [C.c_int]
        L.orc_set_threads.restype = None
        L.orc_get_threads.argtypes = []
        L.orc_get_threads.restype = C.c_int
        L.orc_prune.argtypes = [C.c_int32, _i32p, _i32p, _f32p, C.c_double, _i32p, _i32p, _f32p]
        L.orc_prune.restype = C.c_int64
        L.orc_convergence_stat.argtypes = [C.c_int32, _i32p, _i32p, _f32p, _i32p, _i32p, _f32p]
        L.orc_convergence_stat.restype = C.c_float
        L.orc_mcl.argtypes = [C.c_int32, _i32p, _i32p, _f32p, C.c_int, C.c_double, C.c_int, C.c_double,
                              C.c_int, C.c_int, C.c_int64, _i32p, _i32p, _f32p, C.POINTER(C.c_int),
                              C.POINTER(C.c_int), C.c_void_p]
        L.orc_mcl.restype = C.c_int64
        L.orc_mcl_from.argtypes = L.orc_mcl.argtypes + [C.c_int]
        L.orc_mcl_from.restype = C.c_int64
        L.orc_interpret.argtypes = [C.c_int32, _i32p, _i32p, _f32p, _i32p, _i32p, _i32p]
        L.orc_interpret.restype = C.c_int32
        L.orc_ingest_new.argtypes = [C.c_int32, C.c_int, C.c_int]
        L.orc_ingest_new.restype = C.c_void_p
        L.orc_ingest_push.argtypes = [C.c_void_p, C.c_int64, _i32p, _i64p, _i32p, _i64p, C.c_int, C.c_int32,
                                      _i32p, _i64p, _i32p, _u8p, C.c_int64, _i32p, _i64p, _u8p, C.c_int64]
        L.orc_ingest_push.restype = None
        L.orc_ingest_sizes.argtypes = [C.c_void_p] + [C.POINTER(C.c_int64)] * 4
        L.orc_ingest_sizes.restype = None
        L.orc_ingest_fetch.argtypes = [C.c_void_p, _i32p, _i32p, _i64p, _i64p, _i32p, _i32p, _i64p, _i64p,
                                       C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_ingest_fetch.restype = None
        L.orc_ingest_free.argtypes = [C.c_void_p]
        L.orc_ingest_free.restype = None
        L.orc_dict_to_matrix.argtypes = [C.c_int64, _i32p, _i32p, _f64p, C.c_int32, _u8p, C.c_int32, C.c_int,
                                         _i32p, C.POINTER(C.c_int32), _i32p, C.c_void_p, C.c_void_p]
        L.orc_dict_to_matrix.restype = C.c_int64
        L.orc_rank_sums.argtypes = [C.c_int32, _i32p, _i32p, _f32p, C.c_int, _i64p]
        L.orc_rank_sums.restype = None
        L.orc_count_re_sites.argtypes = [_u8p, C.c_int64, _i64p, _i64p, C.c_int32, _u8p, _i32p, _i64p]
        L.orc_count_re_sites.restype = None
        _lib = L
        L.orc_set_threads(effective_cpus())           # the default: not OpenMP's (every hardware thread of the host, quota or not)
    return _lib


def effective_cpus():
    """CPUs this process may actually use: the affinity mask capped by the cgroup CPU quota (a container on a 256-thread host with
    cpu.max = "1600000 100000" gets 16 CPUs' worth of time: 128 OpenMP threads there run at HALF the rate of 16 — measured,
    tools/oracle_scaling.py)"""
    n = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1)
    try:
        quota, period = open('/sys/fs/cgroup/cpu.max').read().split()[:2]
        if quota != 'max':
            n = min(n, max(1, -(-int(quota) // int(period))))
    except (OSError, ValueError):
        try:
            q = int(open('/sys/fs/cgroup/cpu/cpu.cfs_quota_us').read())
            p_ = int(open('/sys/fs/cgroup/cpu/cpu.cfs_period_us').read())
            if q > 0 and p_ > 0:
                n = min(n, max(1, -(-q // p_)))
        except (OSError, ValueError):
            pass
    return max(1, n)


def set_threads(n):
    """host threads of the row-parallel loops (0 = every CPU this process may use, effective_cpus()); results do not depend on it"""
    n = int(n)
    lib().orc_set_threads(n if n > 0 else effective_cpus())


def get_threads():
    return int(lib().orc_get_threads())


def _csr(indptr, indices, data):
    return (np.ascontiguousarray(indptr, np.int32), np.ascontiguousarray(indices, np.int32),
            np.ascontiguousarray(data, np.float32))


# ---------------------------------------------------------------- MCL pieces (CSR(T) == CSC(M) triples)
def normalize_l1(indptr, data):
    indptr = np.ascontiguousarray(indptr, np.int32)
    out = np.array(data, np.float32, copy=True)
    lib().orc_normalize_l1(len(indptr) - 1, indptr, out)
    return out


def power(data, r):
    out = np.array(data, np.float32, copy=True)
    lib().orc_power(out.size, out, float(r))
    return out


def spgemm(A, B, n_cols=None, mode=0, fx_shift=62):
    ap, aj, ax = _csr(*A)
    bp, bj, bx = _csr(*B)
    n_rows = len(ap) - 1
    n_cols = n_cols if n_cols is not None else len(bp) - 1
    cp = np.zeros(n_rows + 1, np.int32)
    nnz = lib().orc_spgemm(n_rows, n_cols, ap, aj, ax, bp, bj, bx, cp, None, None, mode, fx_shift)
    cj = np.zeros(max(nnz, 1), np.int32)
    cx = np.zeros(max(nnz, 1), np.float32)
    lib().orc_spgemm(n_rows, n_cols, ap, aj, ax, bp, bj, bx, cp, cj.ctypes.data, cx.ctypes.data, mode, fx_shift)
    return cp, cj[:nnz], cx[:nnz]


def links_shift(L):
    """the fixed-point shift s of the integer specification for the raw link matrix L, or -1 if it is not applicable"""
    lp, lj, lx = _csr(*L)
    return int(lib().orc_links_shift(len(lp) - 1, lp, lx, None))


def expand_links(L, rows=None, divide=True):
    """run_mcl_clustering :2144-2147 on the RAW link matrix L (integer counts, symmetric): the rows `rows` (all if None) of
    the pre-expanded matrix M^2, M = L1-normalised L, in the kernels' integer specification (hhx_oracle.c: orc_expand_links).
    divide=False: the values are y = float(S_ij) of the symmetric matrix S = L D^-1 L, without the final division by d_i.
    Returns the CSR triple (indptr over the selected rows, columns, values)."""
    lp, lj, lx = _csr(*L)
    n = len(lp) - 1
    if rows is None:
        n_rows, rp = n, None
    else:
        rows = np.ascontiguousarray(rows, np.int32)
        n_rows, rp = len(rows), rows.ctypes.data
    cp = np.zeros(n_rows + 1, np.int32)
    nnz = lib().orc_expand_links_ex(n, lp, lj, lx, n_rows, rp, cp, None, None, int(bool(divide)))
    if nnz == -2:
        raise ValueError('the integer specification does not apply to this matrix (row sums beyond 2^18, zero rows or non-integer values)')
    cj = np.zeros(max(nnz, 1), np.int32)
    cx = np.zeros(max(nnz, 1), np.float32)
    lib().orc_expand_links_ex(n, lp, lj, lx, n_rows, rp, cp, cj.ctypes.data, cx.ctypes.data, int(bool(divide)))
    return cp, cj[:nnz], cx[:nnz]


def links_iteration0(L, rows, inflation, pruning):
    """iteration 0 of mcl() (:2037-2042: power, normalize, prune) on the rows `rows` of the pre-expanded raw link matrix L, one
    pass over the products (hhx_oracle.c: orc_links_iteration0) — bit for bit prune(normalize_l1(power(expand_links(L, rows)))).
    Returns (indptr over the selected rows, columns, values, entries of M^2 in those rows)."""
    lp, lj, lx = _csr(*L)
    n = len(lp) - 1
    rows = np.ascontiguousarray(rows, np.int32)
    cap_row = n if pruning <= 0 else min(n, int(1.0 / pruning) + 2)
    cnt = np.zeros(len(rows), np.int32)
    oj = np.zeros(max(len(rows) * cap_row, 1), np.int32)           # lazily paged: only what the rows keep is touched
    ox = np.zeros(max(len(rows) * cap_row, 1), np.float32)
    nc = C.c_int64(0)
    total = lib().orc_links_iteration0(n, lp, lj, lx, len(rows), rows.ctypes.data, float(inflation), float(pruning), cap_row, cnt, oj, ox,
                                       C.byref(nc))
    if total == -2:
        raise ValueError('the integer specification does not apply to this matrix')
    if total < 0:
        raise RuntimeError('oracle: a pruned row holds more than 1 / pruning + 1 entries')
    ptr = np.zeros(len(rows) + 1, np.int32)
    ptr[1:] = np.cumsum(cnt)
    take = (np.arange(total, dtype=np.int64) - np.repeat(ptr[:-1].astype(np.int64), cnt)
            + np.repeat(np.arange(len(rows), dtype=np.int64) * cap_row, cnt))
    return ptr, oj[take], ox[take], nc.value


def prune(A, pruning):
    ap, aj, ax = _csr(*A)
    n = len(ap) - 1
    op = np.zeros(n + 1, np.int32)
    oj = np.zeros(max(ax.size, 1), np.int32)
    ox = np.zeros(max(ax.size, 1), np.float32)
    nnz = lib().orc_prune(n, ap, aj, ax, float(pruning), op, oj, ox)
    return op, oj[:nnz], ox[:nnz]


def convergence_stat(A, B):
    ap, aj, ax = _csr(*A)
    bp, bj, bx = _csr(*B)
    return float(lib().orc_convergence_stat(len(ap) - 1, ap, aj, ax, bp, bj, bx))


def mcl(A, expansion, inflation, iters, pruning, spgemm_mode=0, fx_shift=62, want_stats=False, first_it=0):
    """mcl() :2026-2062 on the pre-expanded matrix.  Returns (indptr, indices, data, n_iter, converged[, stats]).
    first_it >= 1: A is what the first `first_it` iterations of the loop left; the loop is picked up there (iteration
    `first_it` expands; the convergence test of :2044 runs from iteration max(first_it, 2) on, at first_it against A)."""
    ap, aj, ax = _csr(*A)
    n = len(ap) - 1
    cap = max(int(ax.size), 1)
    if n:               # a pruned row holds at most 1 / pruning entries (+ the restored maximum); np.zeros pages are lazy
        per_row = n if pruning <= 0 else min(n, int(1.0 / pruning) + 1)
        cap = max(cap, n * per_row)
    op = np.zeros(n + 1, np.int32)
    oj = np.zeros(cap, np.int32)
    ox = np.zeros(cap, np.float32)
    n_iter, conv = C.c_int(0), C.c_int(0)
    stats = np.zeros((max(iters, 1), 4), np.int64)
    nnz = lib().orc_mcl_from(n, ap, aj, ax, int(expansion), float(inflation), int(iters), float(pruning),
                             int(spgemm_mode), int(fx_shift), cap, op, oj, ox, C.byref(n_iter), C.byref(conv),
                             stats.ctypes.data, int(first_it))
    if nnz < 0:
        raise RuntimeError('oracle mcl: output larger than input (unexpected)')
    res = (op, oj[:nnz], ox[:nnz], n_iter.value, bool(conv.value))
    return res + (stats[int(first_it):n_iter.value],) if want_stats else res


def interpret(A):
    """Array half of interpret_result(): (attractors, att_ptr, members), all ascending."""
    ap, aj, ax = _csr(*A)
    n = len(ap) - 1
    att = np.zeros(max(n, 1), np.int32)
    ptr = np.zeros(n + 1, np.int32)
    mem = np.zeros(max(ax.size, 1), np.int32)
    na = lib().orc_interpret(n, ap, aj, ax, att, ptr, mem)
    return att[:na], ptr[:na + 1], mem[:ptr[na]]


# ---------------------------------------------------------------- ingest
class FragTable:
    """Integer view of fa_dict / stat_fragments() outputs that the id-based ingest needs.

    ctg_rank / frag_rank are ranks under Python string ordering of the names."""

    def __init__(self, ctg_rank, ctg_len, ctg_frag0, ctg_split, bin_size, frag_rank, frag_len, frag_nx):
        self.ctg_rank = np.ascontiguousarray(ctg_rank, np.int32)
        self.ctg_len = np.ascontiguousarray(ctg_len, np.int64)
        self.ctg_frag0 = np.ascontiguousarray(ctg_frag0, np.int32)
        self.ctg_split = np.ascontiguousarray(ctg_split, np.uint8)
        self.bin_size = int(bin_size)
        self.frag_rank = np.ascontiguousarray(frag_rank, np.int32)
        self.frag_len = np.ascontiguousarray(frag_len, np.int64)
        self.frag_nx = np.ascontiguousarray(frag_nx, np.uint8)
        self.n_ctg = len(self.ctg_rank)
        self.n_frag = len(self.frag_rank)


def ingest(table, id1, pos1, id2, pos2, flank, bins=False, want_clm=False, max_read_pairs=0, chunk=None):
    """parse_alignments_for_ctgs (bins=False) / parse_alignments (bins=True) on integer ids."""
    L = lib()
    h = L.orc_ingest_new(table.n_frag, int(want_clm), int(max_read_pairs))
    id1 = np.ascontiguousarray(id1, np.int32)
    id2 = np.ascontiguousarray(id2, np.int32)
    pos1 = np.ascontiguousarray(pos1, np.int64)
    pos2 = np.ascontiguousarray(pos2, np.int64)
    n = id1.size
    step = chunk or max(n, 1)
    for s in range(0, n, step):
        e = min(n, s + step)
        L.orc_ingest_push(h, e - s, id1[s:e], pos1[s:e], id2[s:e], pos2[s:e], int(bins), table.n_ctg,
                          table.ctg_rank, table.ctg_len, table.ctg_frag0, table.ctg_split, table.bin_size,
                          table.frag_rank, table.frag_len, table.frag_nx, int(flank))
    sz = [C.c_int64(0) for _ in range(4)]
    L.orc_ingest_sizes(h, *[C.byref(s) for s in sz])
    nf, nk, nclm, ncrd = [s.value for s in sz]
    out = dict(full_i=np.zeros(nf, np.int32), full_j=np.zeros(nf, np.int32), full_cnt=np.zeros(nf, np.int64),
               ht_cnt=np.zeros((nf, 4), np.int64), flank_i=np.zeros(nk, np.int32), flank_j=np.zeros(nk, np.int32),
               flank_cnt=np.zeros(nk, np.int64), frag_links=np.zeros(max(table.n_frag, 1), np.int64))
    extra = want_clm or max_read_pairs
    if extra:
        out.update(clm_ptr=np.zeros(nf + 1, np.int64), clm=np.zeros(max(nclm, 1), np.int64),
                   crd_ptr=np.zeros(nf + 1, np.int64), crd=np.zeros(max(ncrd, 1), np.int64))
    L.orc_ingest_fetch(h, out['full_i'], out['full_j'], out['full_cnt'], out['ht_cnt'].reshape(-1),
                       out['flank_i'], out['flank_j'], out['flank_cnt'], out['frag_links'],
                       out['clm_ptr'].ctypes.data if extra else None, out['clm'].ctypes.data if extra else None,
                       out['crd_ptr'].ctypes.data if extra else None, out['crd'].ctypes.data if extra else None)
    L.orc_ingest_free(h)
    out['frag_links'] = out['frag_links'][:table.n_frag]
    if extra:
        out['clm'] = out['clm'][:nclm]
        out['crd'] = out['crd'][:ncrd]
    return out


def parse_pairs_text(text, names, wide=False):
    """pairs_generator (scripts/HapHiC_cluster.py:1539-1559) on a bytes object: one (id1, pos1, id2, pos2) row per
    LINE (skipped lines and unknown names -> id -1, so that rows stay aligned with lines) and the alignments.bed
    bytes (:1557).  wide: positions as int64 (contigs beyond 2^31 bp, :116-147).  Pure-Python restatement, small inputs only."""
    import io
    cid = {n: i for i, n in enumerate(names)}
    rows, bed = [], []
    for line in io.TextIOWrapper(io.BytesIO(bytes(text)), encoding='utf-8', newline=None):   # 'rt': universal newlines
        if not line.strip() or line.startswith('#'):                                           # :1552
            rows.append((-1, 0, -1, 0))
            continue
        cols = line.split()                                                                    # :1554
        ref, pos, mref, mpos = cols[1], int(cols[2]) - 1, cols[3], int(cols[4]) - 1            # :1556
        bed.append('{0}\t{1}\t{2}\t{3}/1\t255\t.\n{4}\t{5}\t{6}\t{3}/2\t255\t.\n'.format(ref, pos, pos, cols[0], mref, mpos, mpos))
        rows.append((cid.get(ref, -1), pos, cid.get(mref, -1), mpos))
    a = np.array(rows, np.int64).reshape(-1, 4)
    pos_t = np.int64 if wide else np.int32
    return a[:, 0].astype(np.int32), a[:, 1].astype(pos_t), a[:, 2].astype(np.int32), a[:, 3].astype(pos_t), ''.join(bed).encode()


def ht_first(table, id1, pos1, id2, pos2, full_i, full_j):
    """HT_link_dict's insertion order (update_HT_link_dict scripts/HapHiC_cluster.py:404-416, called at :1646 / :1746 for
    every pair that enters full_link_dict): [K, 4] position, among those pairs in stream order, of the first pair of
    contig pair k = (full_i[k], full_j[k]) in quadrant [HH, HT, TH, TT]; INT64_MAX where the quadrant is empty."""
    id1, pos1, id2, pos2 = (np.asarray(a, np.int64) for a in (id1, pos1, id2, pos2))
    ok = (id1 >= 0) & (id2 >= 0) & (id1 != id2)
    id1, pos1, id2, pos2 = id1[ok], pos1[ok], id2[ok], pos2[ok]
    sw = table.ctg_rank[id1] > table.ctg_rank[id2]                                 # :1629 / :1706 sorted by contig name
    ci, cj = np.where(sw, id2, id1), np.where(sw, id1, id2)
    xi, xj = np.where(sw, pos2, pos1) + 1, np.where(sw, pos1, pos2) + 1
    q = (xi * 2 > table.ctg_len[ci]).astype(np.int64) * 2 + (xj * 2 > table.ctg_len[cj])    # :408
    row = {(int(a), int(b)): k for k, (a, b) in enumerate(zip(full_i, full_j))}
    first = np.full((len(row), 4), np.iinfo(np.int64).max, np.int64)
    for n, (a, b, qq) in enumerate(zip(ci.tolist(), cj.tolist(), q.tolist())):
        k = row[(a, b)]
        if first[k, qq] > n:
            first[k, qq] = n
    return first


def frag_pairs(table, id1, pos1, id2, pos2):
    """ctg_pair_to_frag (scripts/HapHiC_cluster.py:1696-1733): the distinct oriented fragment pairs of a stream on
    split contigs, whatever their flank / Nx status.  numpy restatement; returns sorted (frag_i, frag_j) rows."""
    id1, pos1, id2, pos2 = (np.asarray(a, np.int64) for a in (id1, pos1, id2, pos2))
    ok = (id1 >= 0) & (id2 >= 0)                                                   # :1702 not in the fasta
    id1, pos1, id2, pos2 = id1[ok], pos1[ok], id2[ok], pos2[ok]
    split = table.ctg_split.astype(bool)
    keep = (id1 != id2) | split[id1]                                               # :1698
    id1, pos1, id2, pos2 = id1[keep], pos1[keep], id2[keep], pos2[keep]
    c1, c2 = pos1 + 1, pos2 + 1
    r1, r2 = table.ctg_rank[id1].astype(np.int64), table.ctg_rank[id2].astype(np.int64)
    sw = (r1 > r2) | ((r1 == r2) & (c1 > c2))                                      # :1706 sorted((name, coord))
    ci, cj = np.where(sw, id2, id1), np.where(sw, id1, id2)
    xi, xj = np.where(sw, c2, c1), np.where(sw, c1, c2)

    def conv(c, x):                                                                # convert_frags :1665-1673
        b = split[c]
        return table.ctg_frag0[c].astype(np.int64) + np.where(b, (x - 1) // max(table.bin_size, 1), 0), b
    fi, bi = conv(ci, xi)
    fj, bj = conv(cj, xj)
    ne = fi != fj                                                                  # :1714
    fi, fj, anyb = fi[ne], fj[ne], (bi | bj)[ne]
    sw = anyb & (table.frag_rank[fi] > table.frag_rank[fj])                        # :1718-1719
    fi, fj = np.where(sw, fj, fi), np.where(sw, fi, fj)
    u = np.unique(np.stack([fi, fj], 1), axis=0) if len(fi) else np.zeros((0, 2), np.int64)
    return u[:, 0].astype(np.int32), u[:, 1].astype(np.int32)


def dict_to_matrix(fi, fj, val, n_frag, in_set, n_rest, add_self_loops=True):
    """Array half of dict_to_matrix() :310-373.  Returns (indptr, indices, data, frag_index, n_linked)."""
    L = lib()
    fi = np.ascontiguousarray(fi, np.int32)
    fj = np.ascontiguousarray(fj, np.int32)
    val = np.ascontiguousarray(val, np.float64)
    in_set = np.ascontiguousarray(in_set, np.uint8)
    frag_index = np.zeros(max(n_frag, 1), np.int32)
    n_linked = C.c_int32(0)
    # counting call needs indptr sized by the (yet unknown) shape: upper bound n_frag + n_rest
    indptr = np.zeros(n_frag + n_rest + 2, np.int32)
    nnz = L.orc_dict_to_matrix(fi.size, fi, fj, val, n_frag, in_set, n_rest, int(add_self_loops), frag_index,
                               C.byref(n_linked), indptr, None, None)
    shape = n_linked.value + n_rest
    indices = np.zeros(max(nnz, 1), np.int32)
    data = np.zeros(max(nnz, 1), np.float32)
    L.orc_dict_to_matrix(fi.size, fi, fj, val, n_frag, in_set, n_rest, int(add_self_loops), frag_index,
                         C.byref(n_linked), indptr, indices.ctypes.data, data.ctypes.data)
    return indptr[:shape + 1].copy(), indices[:nnz], data[:nnz], frag_index[:n_frag], n_linked.value


def count_re_sites(seq, seg_off, seg_len, sites):
    """count_RE_sites :75-84 over segments of a byte buffer; sites = list of bytes (N-expanded)"""
    buf = np.ascontiguousarray(np.frombuffer(seq, np.uint8) if not isinstance(seq, np.ndarray) else seq, np.uint8)
    if buf.size == 0:
        buf = np.zeros(1, np.uint8)
    off = np.ascontiguousarray(seg_off, np.int64)
    ln = np.ascontiguousarray(seg_len, np.int64)
    pats = np.ascontiguousarray(np.frombuffer(b''.join(sites) or b'\0', np.uint8))
    plen = np.array([len(x) for x in sites] or [0], np.int32)
    out = np.zeros(max(off.size, 1), np.int64)
    lib().orc_count_re_sites(buf, off.size, off if off.size else np.zeros(1, np.int64), ln if ln.size else np.zeros(1, np.int64),
                             len(sites), pats, plen, out)
    return out[:off.size]


def rank_sums(A, topN):
    """filter_fragments rank-sum statistic :866-892 for every row of the (self-loop free) link matrix"""
    ap, aj, ax = _csr(*A)
    n = len(ap) - 1
    out = np.zeros(max(n, 1), np.int64)
    lib().orc_rank_sums(n, ap, aj if aj.size else np.zeros(1, np.int32), ax if ax.size else np.zeros(1, np.float32), int(topN), out)
    return out[:n]


# ---------------------------------------------------------------- a6: link weights (numpy restatement)
def link_weights(fi, fj, value, mode, per_frag=None, tag=None, param=0.0):
    """normalize_by_nlinks :718-724 (mode 0), normalize_by_length :727-738 (mode 1), reduce_inter_hap_HiC_links
    :695-707 (mode 2) on arrays in dict order.  Python float arithmetic == IEEE double; `** 0.5` is C pow().
    Returns the new float64 values (mode 2: zeros mark the entries the reference deletes)."""
    v = np.array(value, np.float64, copy=True)
    fi, fj = np.asarray(fi), np.asarray(fj)
    if mode == 0:
        prod = (np.asarray(per_frag, np.int64)[fi] * np.asarray(per_frag, np.int64)[fj]).astype(np.float64)
        return v / np.array([float(x) ** 0.5 for x in prod.tolist()])
    if mode == 1:
        L = np.asarray(per_frag, np.int64)
        fa = np.minimum(L[fi], param).astype(np.float64)
        fb = np.minimum(L[fj], param).astype(np.float64)
        return v / ((fa / 1000000) * (fb / 1000000))
    t = np.asarray(tag)
    diff = t[fi] != t[fj]
    v[diff] = v[diff] - v[diff] * param
    return v


# ---------------------------------------------------------------- f4: BAM (pure-Python restatement, small inputs only)
def parse_bam(data, need_flags=0x40, drop_same_ref=False):
    """bam_generator :1586-1593 on the bytes of a BAM file, following the SAM specification §4.1-4.2: gunzip the BGZF
    members, read the header, then one (reference_name, next_reference_name, reference_start, next_reference_start)
    tuple per record that passes the htslib filter the reference sets (:2837 :2855 'flag.read1', :2862 additionally
    'refid != mrefid').  A reference id of -1 gives the name None, as pysam does.
    PARITY UNPINNED against htslib itself (pysam is not installed here): pinned by the file-format specification and by a
    BAM written byte by byte in tests/bam_fixture.py.  Returns (header_text, reference names, tuples)."""
    import struct
    import zlib
    raw, rest = bytearray(), bytes(data)
    while rest:                                           # concatenated gzip members
        d = zlib.decompressobj(31)
        raw += d.decompress(rest)
        rest = d.unused_data
    assert raw[:4] == b'BAM\x01', 'not a BAM stream'
    l_text = struct.unpack_from('<i', raw, 4)[0]
    text = raw[8:8 + l_text].rstrip(b'\x00').decode()
    at = 8 + l_text
    n_ref = struct.unpack_from('<i', raw, at)[0]
    at += 4
    names = []
    for _ in range(n_ref):
        l_name = struct.unpack_from('<i', raw, at)[0]
        names.append(raw[at + 4:at + 4 + l_name - 1].decode())
        at += 4 + l_name + 4
    out = []
    while at < len(raw):
        block_size, ref, pos = struct.unpack_from('<iii', raw, at)
        flag = struct.unpack_from('<H', raw, at + 18)[0]
        mref, mpos = struct.unpack_from('<ii', raw, at + 24)
        at += 4 + block_size
        if (flag & need_flags) != need_flags or (drop_same_ref and ref == mref):
            continue
        out.append((names[ref] if ref >= 0 else None, names[mref] if mref >= 0 else None, pos, mpos))
    return text, names, out


# ---------------------------------------------------------------- f3: reassign's per-group link sums (numpy restatement)
def group_link_sums(fi, fj, links, group, n_groups):
    """HapHiC_reassign.py parse_link_dict :217-263 (no normalisation) on arrays in dict order: sums[i][group[j]] += links,
    sums[j][group[i]] += links, and the dict position (2 k + side) of each cell's first contribution (-1: none)."""
    fi, fj, links, group = (np.asarray(a) for a in (fi, fj, links, group))
    sums = np.zeros((len(group), n_groups), np.int64)
    first = np.full((len(group), n_groups), -1, np.int64)
    for k in range(len(fi)):
        a, b = int(fi[k]), int(fj[k])
        for side, (row, g) in enumerate(((a, int(group[b])), (b, int(group[a])))):
            if g >= 0:
                sums[row, g] += int(links[k])
                if first[row, g] < 0:
                    first[row, g] = 2 * k + side
    return sums, first


def clm_text(names, full_i, full_j, clm_ptr, clm):
    """output_clm (scripts/HapHiC_cluster.py:376-392) on the arrays of ingest(want_clm=True): the bytes of paired_links.clm.
    clm_ptr[k] .. clm_ptr[k + 1] delimit the VALUES (four per read pair, update_clm_dict :395-401) of contig pair k in dict order.
    Pure-Python restatement, small inputs only."""
    ori = (('+', '+'), ('+', '-'), ('-', '+'), ('-', '-'))                                     # :380
    out = []
    for k, (a, b) in enumerate(zip(np.asarray(full_i).tolist(), np.asarray(full_j).tolist())):
        list_ = np.asarray(clm[clm_ptr[k]:clm_ptr[k + 1]]).tolist()
        if len(list_) < 8:                                                                     # :385
            continue
        for n in range(4):
            new_list = ['{0} {0}'.format(v) for v in sorted(list_[n::4])]                      # :388
            out.append('{}{} {}{}\t{}\t{}\n'.format(names[a], ori[n][0], names[b], ori[n][1], len(new_list) * 2, ' '.join(new_list)))
    return ''.join(out).encode()


def link_pickle(names, fi, fj, count):
    """output_pickle (scripts/HapHiC_cluster.py:710-715) of a link dict given as arrays: pickle.dumps of the defaultdict(int) the
    reference's loops build (:1605 :1615), keys in array order."""
    import pickle
    from collections import defaultdict
    d = defaultdict(int)
    for a, b, c in zip(np.asarray(fi).tolist(), np.asarray(fj).tolist(), np.asarray(count).tolist()):
        d[(names[a], names[b])] = c
    return pickle.dumps(d)
