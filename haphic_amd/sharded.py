"""Multi-GPU driver of the hot path: one process per GPU, torch.distributed ("nccl" == RCCL over xGMI).

SURVEY §8e:
  * ingest shards by contiguous chunks of the pair stream (pairs are independent, counts additive);
    rank r numbers its pairs from the global ordinal of its chunk (hhx_ingest_set_ordinal_base), so
    the first-seen ordinals of the per-rank tables are comparable.  ONE exchange merges them: an
    all-gather(v) of the aggregated rows (key, first-seen ordinals, counts) followed by the same
    partition + LDS-aggregate pipeline the ingest itself uses (hhx_ingest_push_table + finalize):
    counts add, ordinals take the minimum == the reference loop run over the whole stream.
  * MCL shards T = M^T by row block (== column block of the reference's M).  Everything except the
    right operand of the expansion is row-local, so per iteration there is ONE all-gather(v) of the
    pruned row blocks (indices + values + row lengths) and ONE all-reduce(max) of the convergence
    statistic.  Fixed-point accumulation makes the result bit-identical for any GPU count.

The collective logic is written against a small `engine` interface so that tests/ can drive it on
CPU (gloo, world_size 2) with an oracle-backed engine; the product engine is HipEngine (no fallback).
"""
import numpy as np

from . import _lib


# ------------------------------------------------------------------ engine: device matrices <-> torch
class _DevArray:
    def __init__(self, ptr, n, typestr):
        self.__cuda_array_interface__ = {'shape': (int(n),), 'typestr': typestr, 'data': (int(ptr), False), 'version': 2}


class HipEngine:
    """Matrices are _lib.DeviceCSR handles; tensors are torch views of their device buffers."""

    def __init__(self, device):
        import torch
        self.torch = torch
        self.device = torch.device(device)

    def view(self, ptr, n, typestr, dtype):
        if n == 0:
            return self.torch.empty(0, dtype=dtype, device=self.device)
        return self.torch.as_tensor(_DevArray(ptr, n, typestr), device=self.device)

    def tensors(self, m):
        r, c, z = m.shape3
        a, b, d = m.device_ptrs()
        t = self.torch
        return self.view(a, r + 1, '<i4', t.int32), self.view(b, z, '<i4', t.int32), self.view(d, z, '<f4', t.float32)

    def from_tensors(self, n_rows, n_cols, indptr, indices, data):
        self.torch.cuda.current_stream(self.device).synchronize()
        return _lib.DeviceCSR.from_device(n_rows, n_cols, int(indices.numel()), indptr.data_ptr(), indices.data_ptr(),
                                          data.data_ptr())

    def shape(self, m):
        return m.shape3

    def row_block(self, m, r0, r1):
        return m.row_block(r0, r1)

    def spgemm(self, a, b):
        return _lib.spgemm(a, b, fx_shift=52, want_products=True)

    def inflate_prune(self, c, inflation, pruning):
        return _lib.inflate_prune(c, inflation, pruning)

    def expand_inflate_prune(self, a, b, inflation, pruning):
        return _lib.expand_inflate_prune(a, b, inflation, pruning, fx_shift=52)

    def convergence_stat(self, m, last):
        return _lib.convergence_stat(m, last)

    def copy(self, m):
        return m.copy()

    def free(self, m):
        m.free()

    def sync(self):
        _lib.check(_lib.load().hhx_synchronize())

    def table_tensors(self, n, ptrs):
        """torch views (int64 / int32 bit patterns) of an aggregated ingest table: key, ord_full, ord_flank, ht, fl"""
        t = self.torch
        key, of, ok, ht, fl = ptrs
        return [self.view(key, n, '<i8', t.int64), self.view(of, n, '<i8', t.int64), self.view(ok, n, '<i8', t.int64),
                self.view(ht, 4 * n, '<i4', t.int32), self.view(fl, n, '<i4', t.int32)]


# ------------------------------------------------------------------ collectives on variable-size blocks
def _all_gather_var(t, dist, torch):
    """all-gather of 1-D tensors of different lengths (RCCL has no all-gather-v): gather the lengths,
    pad to the maximum, one all_gather_into_tensor, then slice.  Returns the list of per-rank tensors."""
    world = dist.get_world_size()
    n = torch.tensor([t.numel()], dtype=torch.int64, device=t.device)
    sizes = torch.empty(world, dtype=torch.int64, device=t.device)
    dist.all_gather_into_tensor(sizes, n)
    sizes = sizes.tolist()
    mx = max(max(sizes), 1)
    buf = torch.zeros(mx, dtype=t.dtype, device=t.device)
    buf[:t.numel()] = t
    out = torch.empty(world * mx, dtype=t.dtype, device=t.device)
    dist.all_gather_into_tensor(out, buf)
    return [out[r * mx:r * mx + sizes[r]] for r in range(world)]


def row_ranges(n, world):
    """contiguous, near-equal row blocks"""
    base, rem = divmod(n, world)
    bounds = [0]
    for r in range(world):
        bounds.append(bounds[-1] + base + (1 if r < rem else 0))
    return bounds


def allgather_rows(engine, local, n_cols, dist):
    """all-gather(v) of row blocks -> the full matrix on every rank (rows in rank order)"""
    torch = engine.torch
    ip, ix, dx = engine.tensors(local)
    lens = (ip[1:] - ip[:-1]).contiguous()
    lens_all = _all_gather_var(lens, dist, torch)
    ix_all = _all_gather_var(ix.contiguous(), dist, torch)
    dx_all = _all_gather_var(dx.contiguous(), dist, torch)
    lens_cat = torch.cat(lens_all)
    n_rows = int(lens_cat.numel())
    indptr = torch.zeros(n_rows + 1, dtype=torch.int32, device=lens_cat.device)
    indptr[1:] = torch.cumsum(lens_cat, 0).to(torch.int32)
    return engine.from_tensors(n_rows, n_cols, indptr, torch.cat(ix_all).contiguous(), torch.cat(dx_all).contiguous())


def mcl_sharded_engine(engine, full_norm, expansion, inflation, iters, pruning, dist):
    """run_mcl_clustering's pre-expansion (:2146-2147) + mcl() (:2026-2062) with T sharded by row block.
    full_norm: the L1-normalised link matrix, replicated.  The pre-expansion is fused into iteration 0
    (the expanded rows are consumed in LDS, never materialised).  Returns (full result, n_iter,
    converged, stats)."""
    torch = engine.torch
    n = engine.shape(full_norm)[0]
    world, rank = dist.get_world_size(), dist.get_rank()
    b = row_ranges(n, world)
    r0, r1 = b[rank], b[rank + 1]
    stats = []
    cur_local = engine.row_block(full_norm, r0, r1)   # this rank's rows of the current matrix
    cur_full = full_norm                              # all rows: right operand of the expansion
    own_full = False
    converged = False
    n_iter = 0
    for it in range(iters):
        st_a = engine.shape(cur_full)[2]
        st_f = 0
        run = cur_local
        for _ in range(2, expansion):                 # T^(e-1) rows, :2017-2023
            nxt, f = engine.spgemm(run, cur_full)
            st_f += f
            if run is not cur_local:
                engine.free(run)
            run = nxt
        if expansion > 1:
            p, f, st_c = engine.expand_inflate_prune(run, cur_full, inflation, pruning)   # :2030-2042 fused
            st_f += f
        else:
            c = engine.copy(run)
            st_c = engine.shape(c)[2]
            p = engine.inflate_prune(c, inflation, pruning)
            engine.free(c)
        if run is not cur_local:
            engine.free(run)
        n_iter = it + 1
        red = torch.tensor([0.0, float(st_c), float(engine.shape(p)[2]), float(st_f)], dtype=torch.float64,
                           device=engine.device)
        if it > 1:
            red[0] = engine.convergence_stat(p, cur_local)
        mx = red[:1].clone()
        dist.all_reduce(mx, op=dist.ReduceOp.MAX)            # convergence: one float, max over ranks
        dist.all_reduce(red[1:], op=dist.ReduceOp.SUM)       # bookkeeping only (nnz / product counts)
        full = allgather_rows(engine, p, n, dist)            # the per-iteration all-gather(v)
        stats.append([st_a, int(red[1].item()), int(red[2].item()), int(red[3].item())])
        engine.free(cur_local)
        if own_full:
            engine.free(cur_full)
        cur_local, cur_full, own_full = p, full, True
        if it > 1 and np.float32(mx.item()) <= np.float32(1e-8):
            converged = True
            break
    engine.free(cur_local)
    if not own_full:
        cur_full = engine.copy(cur_full)
    return cur_full, n_iter, converged, np.asarray(stats, np.int64)


# ------------------------------------------------------------------ product entry points (HIP engine)
def mcl_sharded(full_norm, expansion, inflation, iters, pruning, dist, device):
    eng = HipEngine(device)
    return mcl_sharded_engine(eng, full_norm, expansion, inflation, iters, pruning, dist)


def gather_tables(engine, tensors, dist):
    """the ingest exchange: all-gather(v) of every column of the per-rank aggregated tables, rank order"""
    torch = engine.torch
    return [torch.cat(_all_gather_var(t.contiguous(), dist, torch)).contiguous() for t in tensors]


def merge_flank_and_build(ing, table, flank, bins, in_set, dist, device, full_tables=False):
    """exchange step of the sharded ingest + dict_to_matrix on the merged table (replicated on every rank).
    `ing`: this rank's finalized Ingest (ordinal base = number of pairs held by lower ranks).
    full_tables=False exchanges only what the link matrix needs — key, first flank ordinal, flank count
    (20 B per row instead of 44): the contig-pair table (full_link_dict / HT counts) stays per rank and is
    merged only when a caller wants the host dicts (full_tables=True)."""
    eng = HipEngine(device)
    t = eng.torch
    merged = _lib.Ingest(table, flank, bins=bins)
    for which in ((1,) if not full_tables else ((0,) if not bins else (0, 1))):
        n, *ptrs = ing.table_device(which)
        cols = eng.table_tensors(n, ptrs)
        if full_tables:
            g = gather_tables(eng, cols, dist)
        else:
            key, _of, ok, _ht, fl = cols
            if n:                                            # rows that never entered flank_link_dict are not needed
                keep = ok != -1                              # NO_ORD == 2^64 - 1 == -1 as int64
                key, ok, fl = key[keep], ok[keep], fl[keep]
            gk, gok, gfl = gather_tables(eng, [key, ok, fl], dist)
            g = [gk, t.full_like(gk, -1), gok, t.zeros(4 * gk.numel(), dtype=t.int32, device=gk.device), gfl]
        t.cuda.current_stream(eng.device).synchronize()
        merged.push_table(which, g[0].numel(), *[x.data_ptr() for x in g])
    merged.finalize()
    m, fidx, n_linked = merged.link_matrix(in_set)
    return m, n_linked, merged


# ------------------------------------------------------------------ inflation sweep: replicas, no data-path collective
def inflation_sweep(run_one, inflations, dist):
    """run_mcl_clustering :2155-2165 restarts every inflation from the same matrix, so the sweep is embarrassingly
    parallel: rank r runs inflations[r::world] with `run_one(inflation) -> picklable result` (e.g. the
    attractor arrays of hhx_interpret) on its own replica of the matrix; ONE all_gather_object of the small results
    at the end.  Returns the results in the order of `inflations` on every rank."""
    world, rank = dist.get_world_size(), dist.get_rank()
    mine = [(k, run_one(inflations[k])) for k in range(rank, len(inflations), world)]
    gathered = [None] * world
    dist.all_gather_object(gathered, mine)
    out = [None] * len(inflations)
    for part in gathered:
        for k, res in part:
            out[k] = res
    return out
