"""An oracle-backed stand-in for haphic_amd._lib, for CPU tests of the HOST logic in haphic_amd/cluster.py (dict
building, file writing, the inflation sweep, filters): the same call surface, numpy triples instead of device handles,
the C oracle instead of the HIP library.  Test infrastructure only — the product has no such path."""
import numpy as np
import scipy.sparse as sp

from oracle import oracle as orc

FX = dict(spgemm_mode=1, fx_shift=52)


class DeviceCSR:
    def __init__(self, indptr, indices, data, n_cols=None):
        self.a = (np.ascontiguousarray(indptr, np.int32), np.ascontiguousarray(indices, np.int32), np.ascontiguousarray(data, np.float32))
        self.n_cols = len(self.a[0]) - 1 if n_cols is None else n_cols

    @classmethod
    def from_arrays(cls, indptr, indices, data, n_cols=None):
        return cls(indptr, indices, data, n_cols)

    @classmethod
    def from_scipy_csc(cls, m):
        m = m.tocsc()
        if not m.has_canonical_format:
            m = m.copy()
            m.sum_duplicates()
        return cls(m.indptr, m.indices, m.data.astype(np.float32), m.shape[0])

    @property
    def shape3(self):
        return len(self.a[0]) - 1, self.n_cols, int(self.a[0][-1])

    @property
    def nnz(self):
        return self.shape3[2]

    def to_arrays(self):
        return self.a

    def to_scipy_csc(self):
        r, c, _ = self.shape3
        return sp.csc_matrix((self.a[2], self.a[1], self.a[0]), shape=(c, r))

    def copy(self):
        return DeviceCSR(*(x.copy() for x in self.a), n_cols=self.n_cols)

    def row_block(self, r0, r1):
        p, j, x = self.a
        return DeviceCSR(p[r0:r1 + 1] - p[r0], j[p[r0]:p[r1]], x[p[r0]:p[r1]], self.n_cols)

    def free(self):
        pass


def check(rc):
    assert rc in (0, None)


class _Raw:
    def hhx_pool_trim(self):
        return 0

    def hhx_pool_trim_keep(self, keep_bytes):
        return 0

    def hhx_synchronize(self):
        return 0


def load():
    return _Raw()


def ptr(a):
    return a


def files_async():
    return False                      # the CPU stand-in writes every file on the caller's thread


def files_join():
    pass


def links_plan(links):
    return False, 0


def mem_info():
    return 1 << 40, 1 << 40


def pool_prewarm(sizes):
    pass


def pool_cached_bytes():
    return 0


def normalize_l1(m):
    m.a = (m.a[0], m.a[1], orc.normalize_l1(m.a[0], m.a[2]))
    return m


def prune(m, pruning):
    return DeviceCSR(*orc.prune(m.a, pruning), n_cols=m.n_cols)


def spgemm(a, b, fx_shift=52, want_products=False):
    c = orc.spgemm(a.a, b.a, n_cols=b.n_cols, mode=1, fx_shift=52)
    return DeviceCSR(*c, n_cols=b.n_cols)


def inflate_prune_keep(c, inflation, pruning):
    x = orc.normalize_l1(c.a[0], orc.power(c.a[2], inflation))
    return DeviceCSR(*orc.prune((c.a[0], c.a[1], x), pruning), n_cols=c.n_cols)


class DenseRows:
    """rows [r0, r1) of M^2 (M = the L1-normalised link matrix), the stand-in of hhx_dense"""

    def __init__(self, links, r0, r1, fx_shift=52):
        p, j, x = links.a
        norm = (p, j, orc.normalize_l1(p, x))
        blk = (p[r0:r1 + 1] - p[r0], j[p[r0]:p[r1]], norm[2][p[r0]:p[r1]])
        self.c = DeviceCSR(*orc.spgemm(blk, norm, n_cols=links.n_cols, mode=1, fx_shift=52), n_cols=links.n_cols)
        self.n_products = int(np.diff(p)[blk[1]].sum())
        self.nnz_expanded = self.c.nnz

    def inflate_prune(self, inflation, pruning):
        return inflate_prune_keep(self.c, inflation, pruning)

    def inflate_prune_multi(self, inflations, pruning):
        return [self.inflate_prune(r, pruning) for r in inflations]

    def free(self):
        pass


def vstack(blocks):
    ps = [b.a[0] for b in blocks]
    off = np.cumsum([0] + [p[-1] for p in ps])
    indptr = np.concatenate([ps[0][:1]] + [p[1:] + o for p, o in zip(ps, off)])
    return DeviceCSR(indptr, np.concatenate([b.a[1] for b in blocks]), np.concatenate([b.a[2] for b in blocks]), blocks[0].n_cols)


def mcl(m, expansion, inflation, max_iter, pruning, want_stats=False, normalized=False, links=False):
    a = m.a
    if links:
        a = (a[0], a[1], orc.normalize_l1(a[0], a[2]))
    if links or normalized:
        for _ in range(2, expansion + 1):
            a = orc.spgemm(a, (m.a[0], m.a[1], orc.normalize_l1(m.a[0], m.a[2])) if links else m.a, mode=1, fx_shift=52)
    o = orc.mcl(a, expansion, inflation, max_iter, pruning, **FX)
    return DeviceCSR(*o[:3], n_cols=m.n_cols), o[3], bool(o[4])


def mcl_resume(first, done, expansion, inflation, max_iter, pruning):
    """the loop of mcl() :2026-2062 from iteration `done` on, written out with the oracle's pieces"""
    cur, last = first.a, None
    n_iter, conv = done, False
    for it in range(done, max_iter):
        c = cur
        for _ in range(1, expansion):
            c = orc.spgemm(c, cur, mode=1, fx_shift=52)
        x = orc.normalize_l1(c[0], orc.power(c[2], inflation))
        p = orc.prune((c[0], c[1], x), pruning)
        n_iter = it + 1
        if it > 1 and np.float32(orc.convergence_stat(p, cur)) <= np.float32(1e-8):
            cur, conv = p, True
            break
        cur = p
    return DeviceCSR(*cur, n_cols=first.n_cols), n_iter, conv


def interpret(m):
    return orc.interpret(m.a)


def dict_to_matrix(frag_i, frag_j, value, n_frag, in_set, n_rest, add_self_loops=True, on_device=False, n_keys=None):
    p, j, x, fidx, nl = orc.dict_to_matrix(np.asarray(frag_i, np.int32), np.asarray(frag_j, np.int32), np.asarray(value, np.float64),
                                           n_frag, np.ascontiguousarray(in_set, np.uint8), n_rest, add_self_loops)
    return DeviceCSR(p, j, x), fidx, nl


def link_weights(frag_i, frag_j, value, mode, n_frag, per_frag=None, tag=None, param=0.0, device_ptrs=None):
    new = orc.link_weights(frag_i, frag_j, value, mode, per_frag=per_frag, tag=tag, param=param)
    zeros = int(((new == 0) & (np.asarray(tag)[frag_i] != np.asarray(tag)[frag_j])).sum()) if mode == 2 else 0
    value[:] = new
    return zeros


def write_link_pickle(path, i, j, count, names):
    data = orc.link_pickle(names, i, j, count)
    with open(path, 'wb') as f:
        f.write(data)
    return len(data)


def group_link_sums(frag_i, frag_j, links, group, n_groups):
    return orc.group_link_sums(frag_i, frag_j, links, group, n_groups)


def rank_sums(m, topN):
    return orc.rank_sums(m.a, topN)


def count_re_sites(seq, seg_off, seg_len, sites):
    return orc.count_re_sites(seq, seg_off, seg_len, sites)


class PairsParser:
    def __init__(self, names):
        self.names = list(names)
        self.n_lines = self.bed_bytes = 0
        self.wide = False

    def set_wide(self, on=True):
        self.wide = bool(on)

    def parse(self, text, want_bed=False, device_ptr=None, n_bytes=None):
        *self.arr, self.bed = orc.parse_pairs_text(bytes(text), self.names, wide=self.wide)
        self.n_lines, self.bed_bytes = len(self.arr[0]), len(self.bed) if want_bed else 0
        return self.n_lines

    def device_arrays(self):
        return list(self.arr) + [None]

    def fetch(self, want_bed=False):
        return list(self.arr) + [self.bed if want_bed else b'']

    def fetch_bed(self):
        return np.frombuffer(self.bed, np.uint8)

    def bed_host(self):
        return np.frombuffer(self.bed, np.uint8)

    def destroy(self):
        pass


class Ingest:
    def __init__(self, table, flank, bins=False, skip_intra=False, expected_keys=0):
        self.t = orc.FragTable(table.ctg_rank, table.ctg_len, table.ctg_frag0, table.ctg_split, int(table.bin_size), table.frag_rank,
                               table.frag_len, table.frag_nx)
        self.flank, self.bins, self.skip_intra = int(flank), bool(bins), bool(skip_intra)
        self.parts = []
        self.pairs = self.frag = False
        self.n_frag = table.n_frag
        self.out = None
        self.weights = None

    def keep_pairs(self, on=True):
        self.pairs = bool(on)

    def keep_frag_pairs(self, on=True):
        self.frag = bool(on)

    def push(self, id1, pos1, id2, pos2, wide=None):
        self.parts.append([np.array(a, np.int64) for a in (id1, pos1, id2, pos2)])

    def push_device(self, n, id1, pos1, id2, pos2, wide=False):
        self.push(id1[:n], pos1[:n], id2[:n], pos2[:n])

    def _stream(self):
        if not self.parts:
            return [np.zeros(0, np.int64)] * 4
        a = [np.concatenate([p[c] for p in self.parts]) for c in range(4)]
        if self.skip_intra:
            keep = a[0] != a[2]
            a = [x[keep] for x in a]
        return a

    def finalize(self):
        self.fetch()
        return self.n_full, self.n_flank

    def fetch(self, max_read_pairs=0, want=None):
        a = self._stream()
        self.out = orc.ingest(self.t, a[0].astype(np.int32), a[1], a[2].astype(np.int32), a[3], self.flank, bins=self.bins,
                              want_clm=self.pairs, max_read_pairs=max_read_pairs)
        self.n_full, self.n_flank = len(self.out['full_i']), len(self.out['flank_i'])
        return self.out if want is None else {k: self.out[k] for k in want}

    def fetch_flank_values(self):
        o = self.out if self.out is not None else self.fetch()
        return self.weights.copy() if self.weights is not None else o['flank_cnt'].astype(np.float64)

    def weigh_flank(self, mode, per_frag=None, tag=None, param=0.0):
        o = self.out if self.out is not None else self.fetch()
        self.weights = self.fetch_flank_values()
        return link_weights(o['flank_i'], o['flank_j'], self.weights, mode, self.n_frag, per_frag=per_frag, tag=tag, param=param)

    def link_matrix(self, in_set, n_rest=-1, add_self_loops=True, weighted=False):
        o = self.out if self.out is not None else self.fetch()
        assert weighted == (self.weights is not None)
        in_set = np.ascontiguousarray(in_set, np.uint8)
        if n_rest < 0:
            ok = in_set[o['flank_i']].astype(bool) & in_set[o['flank_j']].astype(bool)
            linked = np.zeros(len(in_set), bool)
            linked[o['flank_i'][ok]] = True
            linked[o['flank_j'][ok]] = True
            n_rest = int(in_set.sum() - linked.sum())
        return dict_to_matrix(o['flank_i'], o['flank_j'], self.fetch_flank_values(), self.n_frag, in_set, n_rest, add_self_loops)

    def write_clm(self, path, ctg_names):
        o = self.fetch()
        text = orc.clm_text(ctg_names, o['full_i'], o['full_j'], o['clm_ptr'], o['clm'])
        with open(path, 'wb') as f:
            f.write(text)
        return text.count(b'\n'), len(text)

    def fetch_pairs(self, max_read_pairs, full_cnt):
        o = self.fetch(max_read_pairs)
        crd_ptr = o['crd_ptr'] // 2 if 'crd_ptr' in o else np.zeros(len(full_cnt) + 1, np.int64)
        return o['clm_ptr'] // 4, o['clm'], crd_ptr, o.get('crd', np.zeros(0, np.int64))

    def fetch_ht_order(self):
        a = self._stream()
        o = self.out if self.out is not None else self.fetch()
        return orc.ht_first(self.t, a[0], a[1], a[2], a[3], o['full_i'], o['full_j'])

    def fetch_ht_items(self):
        o = self.out if self.out is not None else self.fetch()
        first = self.fetch_ht_order()
        k, q = np.nonzero(o['ht_cnt'])
        order = np.argsort(first[k, q], kind='stable')
        k, q = k[order], q[order]
        return (2 * o['full_i'][k] + (q >> 1)).astype(np.int32), (2 * o['full_j'][k] + (q & 1)).astype(np.int32), o['ht_cnt'][k, q]

    def n_ht_items(self):
        o = self.out if self.out is not None else self.fetch()
        return int(np.count_nonzero(o['ht_cnt']))

    def fetch_frag_pairs(self):
        a = self._stream()
        return orc.frag_pairs(self.t, a[0], a[1], a[2], a[3])

    def destroy(self):
        pass
