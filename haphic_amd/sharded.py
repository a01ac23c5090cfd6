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
    pruned row blocks — one packed message per rank: row lengths | indices | values — preceded by ONE
    48-byte header all-gather that carries the sizes, the convergence statistic (its max over ranks is
    the all-reduce(max) of :2044-2050) and the bookkeeping counts (exchange_rows).  Fixed-point
    accumulation makes the result bit-identical for any GPU count.

The collective logic is written against a small `engine` interface so that tests/ can drive it on
CPU (gloo, world_size 2) with an oracle-backed engine; the product engine is HipEngine (no fallback).
"""
import os
import time

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

    def scratch(self, name, n, dtype):
        """a view of `n` elements of a persistent device buffer (grown geometrically, never shrunk): the per-iteration exchange of the
        row-block MCL re-uses its message and gather buffers instead of asking the allocator every iteration"""
        pool = self.__dict__.setdefault('_scratch', {})
        buf = pool.get(name)
        if buf is None or buf.dtype != dtype or buf.numel() < n:
            if buf is not None:
                del pool[name]
            buf = pool[name] = self.torch.empty(max(int(n * 1.25), 1024), dtype=dtype, device=self.device)
        return buf[:int(n)]

    def pack_block(self, m, words):
        """m's exchange message [row lengths | column indices | value bits] in an int32 tensor of `words` elements"""
        buf = self.scratch('msg', int(words), self.torch.int32)
        _lib.check(_lib.load().hhx_csr_pack_block(m.h, _lib.C.c_void_p(buf.data_ptr()), int(words)))
        return buf

    def unpack_blocks(self, rows, nnzs, packed, stride, n_cols):
        """the gathered messages (message b at packed[b * stride:]) stacked into one matrix"""
        rows = np.ascontiguousarray(rows, np.int64)
        nnzs = np.ascontiguousarray(nnzs, np.int64)
        self.torch.cuda.current_stream(self.device).synchronize()
        out = _lib.C.c_void_p()
        _lib.check(_lib.load().hhx_csr_unpack_blocks(len(rows), rows.ctypes.data_as(_lib.c_i64p), nnzs.ctypes.data_as(_lib.c_i64p),
                                                     _lib.C.c_void_p(packed.data_ptr()), int(stride), int(n_cols), _lib.C.byref(out)))
        return _lib.DeviceCSR(out)

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

    def expand_links(self, links, r0, r1, inflation, pruning):
        """iteration 0 for this rank's rows [r0, r1) of the whole RAW link matrix `links`"""
        return _lib.expand_links(links, r0, r1, inflation, pruning, fx_shift=52)

    def row_products(self, a, b):
        return _lib.row_products(a, b)

    # ---- iteration 0 on the symmetric half across ranks (expand_links_symmetric below)
    def links_integer_ok(self, links):
        return _lib.links_integer_ok(links)

    def dense_upper(self, links, r0, r1):
        """rows [r0, r1) of Y = float(S), blocks (I, J >= I) only: (handle, [n_rows, n] float32 torch view, columns per window)"""
        d = _lib.DenseRows(links, r0, r1, upper_only=True)
        ptr, ld, cap, _nw = d.device()
        y = self.view(ptr, d.n_rows * ld, '<f4', self.torch.float32).view(d.n_rows, ld)[:, :d.n_cols] if d.n_rows else \
            self.torch.empty((0, d.n_cols), dtype=self.torch.float32, device=self.device)         # rows `ld` floats apart
        return d, y, cap

    def dense_drop(self, d):
        d.free()

    # rectangles of the dense block between the ranks: hand-written tile transposes / one strided copy on the library's stream
    # (y: the [rows, n] view of dense_upper, rows y.stride(0) floats apart)
    def _at(self, y, r, c):
        return _lib.C.c_void_p(y.data_ptr() + 4 * (int(r) * int(y.stride(0)) + int(c)))

    def mirror_block(self, y, r_src, c_src, rows, cols, r_dst, c_dst):
        """y[r_dst : r_dst + cols, c_dst : c_dst + rows] = y[r_src : r_src + rows, c_src : c_src + cols] transposed (disjoint regions)"""
        _lib.check(_lib.load().hhx_transpose_f32(self._at(y, r_src, c_src), int(y.stride(0)), self._at(y, r_dst, c_dst), int(y.stride(0)), int(rows), int(cols)))

    def pack_columns(self, y, c0, c1, out):
        """out (1-D, y.shape[0] * (c1 - c0) floats) = the rectangle y[:, c0:c1], row-major"""
        _lib.check(_lib.load().hhx_copy_rect_f32(self._at(y, 0, c0), int(y.stride(0)), _lib.C.c_void_p(out.data_ptr()), int(c1 - c0), int(y.shape[0]), int(c1 - c0)))

    def unpack_transposed(self, y, c0, buf, rows_s):
        """y[:, c0 : c0 + rows_s] = buf viewed as [rows_s, y.shape[0]] (another rank's rows), transposed"""
        self.torch.cuda.current_stream(self.device).synchronize()           # the collective that filled buf ran on torch's stream
        _lib.check(_lib.load().hhx_transpose_f32(_lib.C.c_void_p(buf.data_ptr()), int(y.shape[0]), self._at(y, 0, c0), int(y.stride(0)), int(rows_s), int(y.shape[0])))

    def dense_finish(self, d, inflation, pruning):
        """the rows finished from the completed block: (pruned rows, products, nnz of the expanded rows)"""
        self.torch.cuda.current_stream(self.device).synchronize()
        p = d.inflate_prune(inflation, pruning)
        out = (p, d.n_products, d.nnz_expanded)
        d.free()
        return out

    # ---- the inflation sweep shared out over the ranks (sweep_sharded below): this rank's rows of M^2, kept for every inflation
    def dense_rows(self, links, r0, r1):
        """rows [r0, r1) of the pre-expanded matrix as a float32 block (hhx_expand_links_dense): (handle, products, nnz of the rows)"""
        d = _lib.DenseRows(links, r0, r1)
        return d, d.n_products, d.nnz_expanded

    def dense_first(self, d, inflation, pruning):
        """iteration 0 of mcl() of the block's rows at `inflation` (hhx_dense_inflate_prune); the block stays"""
        return d.inflate_prune(inflation, pruning)

    def dense_free(self, d):
        d.free()

    def interpret(self, m):
        return _lib.interpret(m)

    def convergence_stat(self, m, last):
        return _lib.convergence_stat(m, last)

    def mcl_resume(self, m, done, expansion, inflation, iters, pruning):
        """the iterations after `done` on the whole matrix `m` (replicated tail): (matrix, n_iter, converged, stats rows)"""
        return _lib.mcl_resume(m, done, expansion, inflation, iters, pruning, want_stats=True)

    def copy(self, m):
        return m.copy()

    def free(self, m):
        m.free()

    def sync(self):
        _lib.check(_lib.load().hhx_synchronize())
        if self.device.type == 'cuda':
            self.torch.cuda.current_stream(self.device).synchronize()       # the collectives run on torch's stream

    # ---- sharded link-matrix build (hhx_shard_*): `src` is this rank's finalized _lib.Ingest
    def shard_open(self, src, in_set):
        in_set = np.ascontiguousarray(in_set, np.uint8)
        h = _lib.C.c_void_p()
        _lib.check(_lib.load().hhx_shard_create(src.h, _lib.ptr(in_set), _lib.C.byref(h)))
        return {'h': h, 'n_frag': src.n_frag}

    def shard_first(self, st):
        p = _lib.C.c_void_p()
        _lib.check(_lib.load().hhx_shard_first(st['h'], _lib.C.byref(p)))
        return self.view(p.value, st['n_frag'], '<i8', self.torch.int64)

    def rank_first(self, first):
        fidx = self.torch.empty(first.numel(), dtype=self.torch.int32, device=self.device)
        nl = _lib.C.c_int32(0)
        self.torch.cuda.current_stream(self.device).synchronize()
        _lib.check(_lib.load().hhx_rank_first(first.numel(), _lib.C.c_void_p(first.data_ptr()), _lib.C.c_void_p(fidx.data_ptr()), _lib.C.byref(nl)))
        return fidx, nl.value

    def shard_emit(self, st, fidx, bounds):
        b = np.ascontiguousarray(bounds, np.int32)
        counts = np.zeros(len(b) - 1, np.int64)
        w0, w1 = _lib.C.c_void_p(), _lib.C.c_void_p()
        _lib.check(_lib.load().hhx_shard_emit(st['h'], _lib.C.c_void_p(fidx.data_ptr()), len(b), b.ctypes.data_as(_lib.c_i32p), _lib.C.byref(w0),
                                              _lib.C.byref(w1), counts.ctypes.data_as(_lib.c_i64p)))
        n = int(counts.sum())
        return self.view(w0.value, n, '<i8', self.torch.int64), self.view(w1.value, n, '<i8', self.torch.int64), counts.tolist()

    def shard_close(self, st):
        _lib.load().hhx_shard_destroy(st['h'])

    def rows_from_entries(self, w0, w1, r0, r1, shape, recv_counts=None):
        """the owner's CSR row block from the entries it received; recv_counts (entries per source rank, in the order they lie in
        w0 / w1): every source's run is already in row order, so the rows are merged straight from the runs (hhx_rows_from_runs)"""
        self.torch.cuda.current_stream(self.device).synchronize()
        out = _lib.C.c_void_p()
        if recv_counts is not None and 1 <= len(recv_counts) <= 64:
            off = np.zeros(len(recv_counts) + 1, np.int64)
            off[1:] = np.cumsum(recv_counts)
            _lib.check(_lib.load().hhx_rows_from_runs(len(recv_counts), off.ctypes.data_as(_lib.c_i64p), _lib.C.c_void_p(w0.data_ptr()),
                                                      _lib.C.c_void_p(w1.data_ptr()), int(r0), int(r1), int(shape), 1, _lib.C.byref(out)))
            return _lib.DeviceCSR(out)
        _lib.check(_lib.load().hhx_rows_from_entries(int(w0.numel()), _lib.C.c_void_p(w0.data_ptr()), _lib.C.c_void_p(w1.data_ptr()), int(r0), int(r1),
                                                     int(shape), 1, _lib.C.byref(out)))
        return _lib.DeviceCSR(out)

    def normalize_l1(self, m):
        _lib.normalize_l1(m)
        return m

    def table_tensors(self, n, ptrs):
        """torch views (int64 / int32 bit patterns) of an aggregated ingest table: key, ord_full, ord_flank, ht, fl"""
        t = self.torch
        key, of, ok, ht, fl = ptrs
        return [self.view(key, n, '<i8', t.int64), self.view(of, n, '<i8', t.int64), self.view(ok, n, '<i8', t.int64),
                self.view(ht, 4 * n, '<i4', t.int32), self.view(fl, n, '<i4', t.int32)]


# ------------------------------------------------------------------ collectives on variable-size blocks
CHECK_EXCHANGES = os.environ.get('HAPHIC_CHECK_EXCHANGES', '1') != '0'      # checksum every all-to-all(v) round (first runs on real xGMI: keep it on)
MAX_MESSAGE_BYTES = 1 << 30     # one collective call never moves more than this per peer: all_to_all_single of this stack (RCCL 2.26.6 /
                                # torch 2.10) silently delivers only the first half of any per-peer message above 2^30 bytes
                                # (tools/rccl_probe.py -> profiles/r03_rccl_probe.jsonl; all-gather and all-reduce are intact to 8.7 GB)


# Per-stage wall clocks of the multi-rank paths, for bench.py --gpus N: STAGES = {} switches the recording on (None: off, no
# extra synchronisation).  Every entry: [milliseconds, bytes received by this rank, calls].  The clock of a stage is read after a device
# synchronisation, so a recorded run has a handful of stream syncs more than an unrecorded one.
STAGES = None


class _stage:
    def __init__(self, engine, name, nbytes=0):
        self.engine, self.name, self.nbytes = engine, name, nbytes

    def __enter__(self):
        if STAGES is not None:
            self.engine.sync()
            self.t0 = time.perf_counter()
        return self

    def __exit__(self, *exc):
        if STAGES is not None and exc[0] is None:
            self.engine.sync()
            rec = STAGES.setdefault(self.name, [0.0, 0, 0])
            rec[0] += (time.perf_counter() - self.t0) * 1e3
            rec[1] += int(self.nbytes)
            rec[2] += 1
        return False


def _bits_sum(x, torch):
    """a wrapping int64 checksum of a tensor's bit patterns (the elements re-read as integers of their own size: legal at any storage offset)"""
    if x.numel() == 0:
        return 0
    as_int = {1: torch.int8, 2: torch.int16, 4: torch.int32, 8: torch.int64}[x.element_size()]
    return int(x.contiguous().view(as_int).sum(dtype=torch.int64).item())


def _all_gather_var(t, dist, torch):
    """all-gather of 1-D tensors of different lengths (RCCL has no all-gather-v): gather the lengths, then rounds of
    all_gather_into_tensor on slices padded to the round's longest piece.  Returns the list of per-rank tensors."""
    world = dist.get_world_size()
    n = torch.tensor([t.numel()], dtype=torch.int64, device=t.device)
    sizes = torch.empty(world, dtype=torch.int64, device=t.device)
    dist.all_gather_into_tensor(sizes, n)
    sizes = sizes.tolist()
    step = max(1, MAX_MESSAGE_BYTES // t.element_size())
    outs = [torch.empty(sizes[r], dtype=t.dtype, device=t.device) for r in range(world)]
    for lo in range(0, max(max(sizes), 1), step):
        mx = max(1, min(step, max(sizes) - lo))
        buf = torch.zeros(mx, dtype=t.dtype, device=t.device)
        mine = t[lo:lo + mx]
        buf[:mine.numel()] = mine
        out = torch.empty(world * mx, dtype=t.dtype, device=t.device)
        dist.all_gather_into_tensor(out, buf)
        for r in range(world):
            k = max(0, min(mx, sizes[r] - lo))
            if k:
                outs[r][lo:lo + k] = out[r * mx:r * mx + k]
    return outs


def _all_to_all_var(t, send_counts, dist, torch):
    """all-to-all(v) of a 1-D tensor laid out as [to rank 0 | to rank 1 | ...]: exchange the counts, then rounds of
    all_to_all_single in which every (sender, receiver) pair moves at most MAX_MESSAGE_BYTES.  Returns the received
    tensor laid out as [from rank 0 | from rank 1 | ...] and the per-source counts."""
    world = dist.get_world_size()
    send_n = torch.tensor(send_counts, dtype=torch.int64, device=t.device)
    recv_n = torch.empty(world, dtype=torch.int64, device=t.device)
    dist.all_to_all_single(recv_n, send_n)
    recv_counts = recv_n.tolist()
    longest = torch.tensor([max(max(send_counts), max(recv_counts), 0)], dtype=torch.int64, device=t.device)
    dist.all_reduce(longest, op=dist.ReduceOp.MAX)                    # every rank runs the same number of rounds
    step = max(1, MAX_MESSAGE_BYTES // t.element_size())
    out = torch.empty(int(sum(recv_counts)), dtype=t.dtype, device=t.device)
    s_off = np.concatenate([[0], np.cumsum(send_counts)]).tolist()
    r_off = np.concatenate([[0], np.cumsum(recv_counts)]).tolist()
    broken = None
    for lo in range(0, max(int(longest.item()), 1), step):
        s_k = [max(0, min(step, c - lo)) for c in send_counts]
        r_k = [max(0, min(step, c - lo)) for c in recv_counts]
        pieces = [t[s_off[p] + lo:s_off[p] + lo + s_k[p]] for p in range(world)]
        piece = torch.cat(pieces) if sum(s_k) else t[:0]
        got = torch.empty(int(sum(r_k)), dtype=t.dtype, device=t.device)
        dist.all_to_all_single(got, piece.contiguous(), output_split_sizes=r_k, input_split_sizes=s_k)
        if CHECK_EXCHANGES:
            # all_to_all_single of this stack has been seen to deliver a message SHORT without an error (profiles/r03_rccl_probe.jsonl: beyond
            # 2^30 B per peer); the slices above stay below that, and this makes sure: what arrived from every peer must have the checksum the
            # peer computed over what it sent (one more exchange of `world` integers per round)
            sent = torch.tensor([_bits_sum(x, torch) for x in pieces], dtype=torch.int64, device=t.device)
            want = torch.empty(world, dtype=torch.int64, device=t.device)
            dist.all_to_all_single(want, sent)
            want = want.tolist()
        at = 0
        for p in range(world):
            if r_k[p]:
                part = got[at:at + r_k[p]]
                if CHECK_EXCHANGES and broken is None and _bits_sum(part, torch) != want[p]:
                    broken = 'all-to-all: the %d elements rank %d sent to rank %d in one round did not arrive intact (%d expected in all)' % (
                        r_k[p], p, dist.get_rank(), recv_counts[p])
                out[r_off[p] + lo:r_off[p] + lo + r_k[p]] = part
                at += r_k[p]
    if broken is not None:                    # raised after the last round: every rank has run the same sequence of collectives
        raise RuntimeError(broken)
    return out, recv_counts


def row_ranges(n, world):
    """contiguous, near-equal row blocks"""
    base, rem = divmod(n, world)
    bounds = [0]
    for r in range(world):
        bounds.append(bounds[-1] + base + (1 if r < rem else 0))
    return bounds


def balanced_ranges(cost, world):
    """contiguous row blocks of (nearly) equal total cost: block k ends at the first row where the running cost reaches
    (k + 1) / world of the total.  cost: int64 per row (products of the expansion, hhx_row_products)."""
    c = np.cumsum(np.asarray(cost, np.int64))
    total = int(c[-1]) if len(c) else 0
    n = len(c)
    bounds = [0]
    for k in range(1, world):
        cut = int(np.searchsorted(c, total * k / world, side='left')) + 1 if total else n * k // world
        # a few very heavy rows must not leave a rank without rows (every rank runs the same kernels and collectives; an
        # empty block is legal but pointless): at least one row per rank while there are rows to give
        cut = max(cut, bounds[-1] + 1)
        bounds.append(max(bounds[-1], min(n - min(n, world - k), cut) if n >= world else min(n, cut)))
    bounds.append(n)
    return bounds


HEADER = 6     # doubles per rank: rows, entries, convergence statistic, nnz of the expanded rows, survivors, products


def exchange_rows(engine, local, n_cols, dist, stat=0.0, counts=(0, 0, 0), stage='exchange_rows'):
    """The exchange of the row-block MCL — all-gather(v) of the row blocks, all-reduce(max) of the convergence statistic
    (:2044-2050) and the sums of the per-iteration counts — in TWO collectives:
      1. all-gather of a 6-double header per rank: the sizes of the variable-length blocks, the convergence statistic (a
         float32, exact in a double) and the bookkeeping counts (< 2^53, exact) ride in the same 48 bytes, so the max and
         the sums are formed locally from the gathered headers — no separate all-reduce;
      2. all-gather of ONE packed int32 message per rank, [row lengths | column indices | value bits] (hhx_csr_pack_block),
         padded to the longest message; hhx_csr_unpack_blocks stacks the world's messages into the CSR triple with the row
         pointer formed by one device scan.
    A message longer than MAX_MESSAGE_BYTES goes in slices (the raw link matrix at 2 ranks: 1.3 GB per rank).
    Returns (the full matrix, rows in rank order; the headers as a [world, 6] float64 numpy array)."""
    torch = engine.torch
    world = dist.get_world_size()
    r, _c, z = engine.shape(local)
    head = torch.tensor([float(r), float(z), float(stat)] + [float(c) for c in counts], dtype=torch.float64, device=engine.device)
    scratch = getattr(engine, 'scratch', None) or (lambda _name, n_, dt: torch.empty(int(n_), dtype=dt, device=engine.device))
    heads = scratch('heads', world * HEADER, torch.float64)
    dist.all_gather_into_tensor(heads, head)
    heads = heads.cpu().numpy().reshape(world, HEADER)               # the one host sync of the exchange
    rows, nnzs = heads[:, 0].astype(np.int64), heads[:, 1].astype(np.int64)
    stride = int(max(1, (rows + 2 * nnzs).max()))
    msg = engine.pack_block(local, stride)
    out = scratch('gathered', world * stride, torch.int32)       # persistent: consumed by unpack_blocks before the next exchange
    step = max(1, MAX_MESSAGE_BYTES // 4)
    with _stage(engine, stage, world * stride * 4):
        if stride <= step:
            dist.all_gather_into_tensor(out, msg)
        else:
            out2 = out.view(world, stride)
            for lo in range(0, stride, step):
                hi = min(stride, lo + step)
                piece = scratch('piece', world * (hi - lo), torch.int32)
                dist.all_gather_into_tensor(piece, msg[lo:hi].contiguous())
                out2[:, lo:hi] = piece.view(world, hi - lo)
        full = engine.unpack_blocks(rows, nnzs, out, stride, n_cols)     # fails if the row lengths of a message do not add up to its header
    return full, heads


def allgather_rows(engine, local, n_cols, dist, stage='allgather_rows'):
    """all-gather(v) of row blocks -> the full matrix on every rank (rows in rank order)"""
    return exchange_rows(engine, local, n_cols, dist, stage=stage)[0]


# Iteration 0 on the symmetric half across ranks trades 0.4 F / N products per rank (187 / N ms at C3) for an exchange of
# n^2 * 4 B / N^2 per peer pair (10 GB at N = 2, 2.5 GB at 4, 0.6 GB at 8: ~156 / 39 / 10 ms at ~64 GB/s per xGMI link and
# direction, the links of a rank running side by side): by that model it pays from 8 ranks on, is a wash at 4 and loses at 2 —
# so it would be on from 8 ranks.  No run on more than one physical GPU exists yet, so it is OPT-IN until one does:
# SYMMETRIC_MIN_WORLD = None (off) unless HAPHIC_SYMMETRIC_MIN_WORLD names a world size; tests set it to 2.
SYMMETRIC_MIN_WORLD = int(os.environ['HAPHIC_SYMMETRIC_MIN_WORLD']) if os.environ.get('HAPHIC_SYMMETRIC_MIN_WORLD') else None
SYMMETRIC_HALF = True          # tests switch it off to compare the two multi-rank paths


def symmetric_window(engine, n):
    """columns per window of the expansion's plan for an order-n matrix (the library's own rule: the fewest windows whose 8-byte
    accumulators fit the 160 KB of LDS; hhx_expand.hip) — only the balance of the row blocks depends on it"""
    cap_max = ((160 * 1024 - 784) // 8) & ~63                    # 784: the window kernel's scratch slots and reduction arrays
    n_win = -(-n // cap_max)
    return (-(-n // n_win) + 63) & ~63


def upper_cost(products, cap):
    """cost of a row in the symmetric iteration 0: its products in the column windows J >= its own block — estimated as the
    share of the columns right of the block start (the link matrix has no column structure a row block could exploit)"""
    products = np.asarray(products, np.float64)
    n = len(products)
    start = (np.arange(n, dtype=np.int64) // cap) * cap
    return np.maximum(1, products * (n - start) / max(n, 1)).astype(np.int64)


def expand_links_symmetric(engine, links_full, bounds, inflation, pruning, dist):
    """Iteration 0 of mcl() on the raw link matrix with the SYMMETRIC HALF shared out over the ranks: S = L D^-1 L
    is an exactly symmetric integer matrix, so rank r fills only the blocks (I, J >= I) of ITS rows [bounds[r], bounds[r + 1]) of
    Y = float(S) — 60 % of their products at five column windows — and the ranks hand each other the mirror image:
      * rank s sends the rectangle Y[rows of s][columns = rows of r] to every rank r > s (ONE all-to-all(v) of float32; every
        entry of such a rectangle lies right of its row's own block, so s has computed it); r stores its transpose;
      * what mirrors inside a rank's own rows (block pairs I < J both cut by the row range) is transposed locally.
    Entries that both sides computed (the diagonal blocks) are bit-identical, so overwriting them is harmless.  Then the dense
    epilogue finishes the rows — the same bits as the one-GPU call.  Returns (pruned rows of this rank, products, nnz of M^2)."""
    torch = engine.torch
    world, rank = dist.get_world_size(), dist.get_rank()
    r0, r1 = bounds[rank], bounds[rank + 1]
    try:
        d, y, cap = engine.dense_upper(links_full, r0, r1)
    except RuntimeError:                          # e.g. the block does not fit this rank's memory
        d = None
    if not _agree(d is not None, engine, dist):   # all or none: a rank alone in the all-to-all below would hang the others
        if d is not None:
            engine.dense_drop(d)
        return None, 0, 0
    n_loc = r1 - r0
    # mirror inside the own rows: for the blocks I < J that both meet [r0, r1) (engine.mirror_block: tile transposes)
    if n_loc:
        cuts = sorted({r0, r1} | {c for c in range((r0 // cap + 1) * cap, r1, cap)})
        segs = list(zip(cuts[:-1], cuts[1:]))                    # the row range cut at the block boundaries
        for a in range(len(segs)):
            for b_ in range(a + 1, len(segs)):
                (ia, ib), (ja, jb) = segs[a], segs[b_]
                engine.mirror_block(y, ia - r0, ja, ib - ia, jb - ja, ja - r0, ia)      # y[rows of J, columns of I] = y[rows of I, columns of J]^T
    # the rectangles for the ranks above, packed side by side in ONE send buffer laid out by destination (engine.pack_columns: one
    # strided device copy each, straight into its place — no intermediate tensors, no concatenation)
    send_counts = [0] * world
    for r in range(rank + 1, world):
        if n_loc and bounds[r + 1] > bounds[r]:
            send_counts[r] = n_loc * (bounds[r + 1] - bounds[r])
    send = torch.empty(int(sum(send_counts)), dtype=torch.float32, device=y.device)
    at = 0
    for r in range(rank + 1, world):
        if send_counts[r]:
            engine.pack_columns(y, bounds[r], bounds[r + 1], send[at:at + send_counts[r]])
            at += send_counts[r]
    engine.sync()                                               # the packing ran on the engine's stream; the collective runs on torch's
    got, recv_counts = _all_to_all_var(send, send_counts, dist, torch)
    at = 0
    for s_ in range(world):
        k = recv_counts[s_]
        if k:
            rows_s = bounds[s_ + 1] - bounds[s_]
            assert s_ < rank and k == rows_s * n_loc
            engine.unpack_transposed(y, bounds[s_], got[at:at + k], rows_s)      # rank s_'s rows, stored as this rank's columns
            at += k
    engine.sync()
    del got, send
    return engine.dense_finish(d, inflation, pruning)


REPLICATE_NNZ = 4_000_000      # below this many entries the iterations are cheaper than their collectives: every rank runs them whole


def _agree(ok, engine, dist):
    """True only if every rank says so: one all-reduce(min) — a rank that cannot take a path (e.g. its dense block does not fit)
    must not leave the others waiting in that path's collective"""
    t = engine.torch.tensor([1 if ok else 0], dtype=engine.torch.int32, device=engine.device)
    dist.all_reduce(t, op=dist.ReduceOp.MIN)
    return bool(int(t.item()))


def sharded_iteration(engine, cur_full, it, expansion, inflation, pruning, dist, stage='mcl_sharded_iteration'):
    """Iteration `it` >= 1 of mcl() (:2030-2050) with T sharded by row block: the blocks are cut anew at equal PRODUCT counts of
    THIS iteration (every rank holds cur_full, so re-cutting costs no communication — a row's weight changes from one iteration to
    the next as clusters form), every rank expands / inflates / prunes its rows against the whole matrix, then the two-collective
    exchange.  Returns (the new full matrix, largest convergence statistic over the ranks, [nnz_A, nnz_C, survivors, products])."""
    world, rank = dist.get_world_size(), dist.get_rank()
    n = engine.shape(cur_full)[0]
    compute = _stage(engine, stage + '_compute')
    compute.__enter__()
    b = balanced_ranges(engine.row_products(cur_full, cur_full), world)
    a = engine.row_block(cur_full, b[rank], b[rank + 1])
    st_f = 0
    run = a
    for _ in range(2, expansion):                 # T^(e-1) rows, :2017-2023
        nxt, f = engine.spgemm(run, cur_full)
        st_f += f
        if run is not a:
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
    if run is not a:
        engine.free(run)
    stat = engine.convergence_stat(p, a) if it > 1 else 0.0
    engine.free(a)
    compute.__exit__(None, None, None)
    full, heads = exchange_rows(engine, p, n, dist, stat=stat, counts=(st_c, engine.shape(p)[2], st_f), stage=stage + '_exchange')
    engine.free(p)
    return full, np.float32(heads[:, 2].max()), [engine.shape(cur_full)[2]] + [int(v) for v in heads[:, 3:].sum(axis=0)]


def mcl_sharded_engine(engine, full_norm, expansion, inflation, iters, pruning, dist, local_block=None, n=None, local_links=None,
                       replicate_nnz=None):
    """run_mcl_clustering's pre-expansion (:2146-2147) + mcl() (:2026-2062) with T sharded by row block.
    Three ways in:  full_norm = the L1-normalised link matrix, replicated;  local_block = this rank's rows
    [row_ranges(n)[rank], ...) of it (the right operand of iteration 0 is then all-gathered once);  local_links = this
    rank's rows of the RAW link matrix as build_link_matrix_sharded leaves them: the raw blocks are all-gathered and
    iteration 0 goes through the class stream (hhx_expand_links), the normalisation (:2144) being row-local.
    The pre-expansion is fused into iteration 0 (the expanded rows are consumed in LDS, never materialised).
    Once the all-gathered matrix has at most replicate_nnz entries (after the first few iterations T has a handful of
    entries per row: an iteration is a fraction of a millisecond, an all-gather(v) + all-reduce is not) every rank runs
    the remaining iterations on the whole matrix — same kernels on the same values, so the same bits on every rank and
    for every GPU count, and no collective at all.
    Returns (full result, n_iter, converged, stats)."""
    if replicate_nnz is None:
        replicate_nnz = REPLICATE_NNZ
    world, rank = dist.get_world_size(), dist.get_rank()
    links_full = None
    own_full = False
    if local_links is not None:
        links_full = allgather_rows(engine, local_links, n, dist, stage='mcl_raw_allgather')       # exchange: the raw link matrix, once
        if expansion != 2:
            # mkl_matrix_power :2017-2023 recurses for any e: T^(e-1) rows need the normalised matrix as a plain operand, so the
            # raw blocks are normalised (:2144, row-local) and the general path below runs
            full_norm = engine.normalize_l1(links_full)
            links_full = None
            own_full = True
    elif full_norm is None:
        full_norm = allgather_rows(engine, local_block, n, dist)         # n: order of the matrix
        own_full = True
    stats = []
    converged = False
    n_iter = 0
    if links_full is not None:
        n = engine.shape(links_full)[0]
        if iters < 1:
            return engine.normalize_l1(links_full), 0, False, np.zeros((0, 4), np.int64)
        # Every rank holds the whole raw matrix now, so the MCL row blocks need not be the build's: they are cut at
        # equal PRODUCT counts of iteration 0.  Equal row counts are not balanced: a contig's matrix
        # index is its first-seen rank in the pair stream, and heavily linked contigs are seen first.
        products = engine.row_products(links_full, links_full)
        # the symmetric half across the ranks: opt-in (SYMMETRIC_MIN_WORLD), when the engine has it and the integer arithmetic
        # applies; the rows are then balanced by their products right of their own block.  Every rank must agree before any
        # of them enters its all-to-all
        symmetric = (SYMMETRIC_MIN_WORLD is not None and world >= SYMMETRIC_MIN_WORLD and SYMMETRIC_HALF and hasattr(engine, 'dense_upper')
                     and engine.links_integer_ok(links_full))
        p = None
        with _stage(engine, 'mcl_iteration0_compute'):
            if symmetric:
                b = balanced_ranges(upper_cost(products, symmetric_window(engine, n)), world)
                p, f, st_c = expand_links_symmetric(engine, links_full, b, inflation, pruning, dist)   # None: some rank could not hold its block
            if p is None:
                b = balanced_ranges(products, world)
                p, f, st_c = engine.expand_links(links_full, b[rank], b[rank + 1], inflation, pruning)       # iteration 0, class stream
        st_a = engine.shape(links_full)[2]
        engine.free(links_full)
        cur_full, heads = exchange_rows(engine, p, n, dist, counts=(st_c, engine.shape(p)[2], f), stage='mcl_iteration0_exchange')
        engine.free(p)
        stats.append([st_a] + [int(v) for v in heads[:, 3:].sum(axis=0)])
        own_full = True
        n_iter = 1
    else:
        cur_full = full_norm                          # all rows: right operand of the expansion
        n = engine.shape(cur_full)[0]
    for it in range(n_iter, iters):
        if world > 1 and it >= 1 and engine.shape(cur_full)[2] <= replicate_nnz and hasattr(engine, 'mcl_resume'):
            with _stage(engine, 'mcl_replicated_tail'):
                res, n_iter, converged, tail = engine.mcl_resume(cur_full, it, expansion, inflation, iters, pruning)
            stats.extend(np.asarray(tail, np.int64).tolist())
            if own_full:
                engine.free(cur_full)
            cur_full, own_full = res, True
            break
        full, mx, st = sharded_iteration(engine, cur_full, it, expansion, inflation, pruning, dist)
        stats.append(st)
        if own_full:
            engine.free(cur_full)
        cur_full, own_full = full, True
        n_iter = it + 1
        if it > 1 and mx <= np.float32(1e-8):
            converged = True
            break
    if not own_full:
        cur_full = engine.copy(cur_full)
    return cur_full, n_iter, converged, np.asarray(stats, np.int64).reshape(-1, 4)


# ------------------------------------------------------------------ the inflation sweep of run_mcl_clustering across the ranks
SWEEP_SHARD_PRODUCTS = 4e9     # an iteration of at least this many products (~2 ms of kernel on one GPU) is shared out over the ranks


def sweep_sharded(engine, links_full, inflations, iters, pruning, dist, shard_products=None):
    """run_mcl_clustering :2144-2165 (expansion 2) over N ranks that all hold the raw link matrix.  Every inflation restarts mcl()
    from the same pre-expanded matrix; on one GPU that is one expansion + an epilogue and a tail per inflation, and the tails of
    the LOW inflations are nearly all of the time (14 of 18.5 s at 100k contigs: inflations 1.1-1.3 run ~100 iterations on matrices
    of 10^8 entries) — dealing whole inflations to the ranks would end when the rank holding 1.1 ends.  Here:
      1. ONE expansion, shared: rank r holds only ITS rows of M^2 (products-balanced row block) as the dense float32 block;
      2. iteration 0 of every inflation = the dense epilogue over the rank's rows + the two-collective exchange: every rank
         gets T1(inflation) whole;
      3. while the next iteration of an inflation is HEAVY (nnz^2 / n >= shard_products) it is run row-sharded by all ranks
         together (sharded_iteration: blocks re-cut at equal products every iteration);
      4. what is left of every inflation — a matrix + the number of iterations done — is a LIGHT task: given to the least-loaded
         rank by predicted cost (nnz^2 / n: the products of its next iteration), run there without collectives (hhx_mcl_resume);
      5. ONE all_gather_object of the attractor arrays.
    Same kernels on the same values: the results are bit-identical to the one-GPU sweep for every rank count.
    Returns, on every rank, a list over `inflations` of (att, att_ptr, members, shape, n_iter, converged)."""
    if shard_products is None:
        shard_products = SWEEP_SHARD_PRODUCTS
    world, rank = dist.get_world_size(), dist.get_rank()
    n = engine.shape(links_full)[0]
    b = balanced_ranges(engine.row_products(links_full, links_full), world)
    # This rank's rows of M^2 as a dense float32 block — if EVERY rank can hold its block (the blocks are cut by products, not by
    # rows, so their sizes differ).  The ranks agree before anyone relies on it: a rank that cannot allocate must not leave the
    # others in the exchange below.  Without the blocks every rank still expands only its own rows, once per inflation, through
    # the fused iteration 0 (hhx_expand_links: no dense block) — same accumulators, same bits, one expansion per inflation.
    d, f_local, c_local = None, 0, 0
    try:
        with _stage(engine, 'sweep_expansion'):
            d, f_local, c_local = engine.dense_rows(links_full, b[rank], b[rank + 1])
    except (RuntimeError, MemoryError):
        d = None
    if not _agree(d is not None, engine, dist):
        if d is not None:
            engine.dense_free(d)
        d = None
    load = [0.0] * world
    mine = []                                             # (index of the inflation, matrix, iterations done)
    done = {}                                             # inflations that converged / ran out of iterations inside the sharded phase
    try:
        for k, infl in enumerate(inflations):
            infl = float(infl)
            with _stage(engine, 'sweep_iteration0_compute'):
                if d is not None:
                    p = engine.dense_first(d, infl, pruning)
                else:
                    p, f_local, c_local = engine.expand_links(links_full, b[rank], b[rank + 1], infl, pruning)
            cur, _heads = exchange_rows(engine, p, n, dist, counts=(c_local, engine.shape(p)[2], f_local), stage='sweep_iteration0_exchange')
            engine.free(p)
            it, converged, finished = 1, False, iters <= 1
            while not finished and world > 1 and float(engine.shape(cur)[2]) ** 2 / max(n, 1) >= shard_products:
                full, mx, _st = sharded_iteration(engine, cur, it, 2, infl, pruning, dist, stage='sweep_sharded_iteration')
                engine.free(cur)
                cur = full
                it += 1
                if it > 2 and mx <= np.float32(1e-8):
                    converged = finished = True
                elif it >= iters:
                    finished = True
            if finished:
                done[k] = (cur, it, converged)
                continue
            owner = min(range(world), key=lambda r: (load[r], r))          # the same on every rank
            load[owner] += float(engine.shape(cur)[2]) ** 2 / max(n, 1) + float(engine.shape(cur)[2])
            if owner == rank:
                mine.append((k, cur, it))
            else:
                engine.free(cur)
    except BaseException:
        for _k, cur, _it in mine:                        # nobody else holds these
            engine.free(cur)
        for cur, _it, _c in done.values():
            engine.free(cur)
        raise
    finally:
        if d is not None:
            engine.dense_free(d)
    results = []
    for k, (cur, it, converged) in done.items():          # every rank holds these: rank k % world reads them out
        if k % world == rank:
            results.append((k,) + tuple(engine.interpret(cur)) + (n, it, converged))
        engine.free(cur)
    for k, cur, it in mine:
        with _stage(engine, 'sweep_light_tails_on_this_rank'):
            res, n_iter, converged, _tail = engine.mcl_resume(cur, it, 2, float(inflations[k]), iters, pruning)
        engine.free(cur)
        results.append((k,) + tuple(engine.interpret(res)) + (n, n_iter, converged))
        engine.free(res)
    gathered = [None] * world
    with _stage(engine, 'sweep_gather_results'):
        dist.all_gather_object(gathered, results)
    out = [None] * len(inflations)
    for part in gathered:
        for r in part:
            out[r[0]] = r[1:]
    return out


# ------------------------------------------------------------------ product entry points (HIP engine)
def mcl_sharded(full_norm, expansion, inflation, iters, pruning, dist, device, local_block=None, n=None, local_links=None, replicate_nnz=None):
    eng = HipEngine(device)
    return mcl_sharded_engine(eng, full_norm, expansion, inflation, iters, pruning, dist, local_block=local_block, n=n, local_links=local_links,
                              replicate_nnz=replicate_nnz)


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


def build_link_matrix_sharded(engine, src, in_set, dist, timings=None, rest_order=None):
    """dict_to_matrix (:310-373, add_self_loops) over a pair stream that is split across the ranks: `src` is this rank's
    finalized ingest of its chunk (global ordinals, hhx_ingest_set_ordinal_base).  One all-reduce(min) of the first
    positions (8 B per fragment) fixes every fragment's matrix index on all ranks; one all-to-all(v) moves each matrix
    entry to the owner of its row; the owner adds the counts of the chunks.  Returns (this rank's CSR row block of the
    link matrix [row_ranges(shape)[rank], ...), fragment -> matrix index array (int32 numpy, -1 = not in the matrix),
    n_linked, shape).  Link-less members of in_set take the trailing indices; the reference numbers them in CPython's
    iteration order of the set `frag_set - frags_in_dict` (:357-359), which only the caller that holds the names can
    know: pass it as `rest_order` (fragment ids, every rank the same) to reproduce it — haphic_amd.cluster.dict_to_matrix
    does the same on one GPU.  Without it they are numbered by fragment id, which can number singleton groups differently
    from the reference on length ties."""
    torch = engine.torch
    world, rank = dist.get_world_size(), dist.get_rank()
    import time
    in_set = np.ascontiguousarray(in_set, np.uint8)
    marks = [('start', time.perf_counter())]

    def mark(name):                      # stage wall times for bench.py (a sync per stage: only when asked for)
        if timings is not None:
            engine.sync()
            marks.append((name, time.perf_counter()))
    st = engine.shard_open(src, in_set)
    try:
        first = engine.shard_first(st).clone()
        mark('partition_by_row')
        dist.all_reduce(first, op=dist.ReduceOp.MIN)                       # exchange 1: first positions
        fidx, n_linked = engine.rank_first(first)
        mark('allreduce_rank')
        shape = int(in_set.sum())                                          # every member of frag_set gets a row (linked ones first)
        bounds = row_ranges(shape, world)
        w0, w1, counts = engine.shard_emit(st, fidx, bounds)
        mark('emit')
        e0, recv = _all_to_all_var(w0, counts, dist, torch)                 # exchange 2: the entries, to their row owners
        e1, _recv = _all_to_all_var(w1, counts, dist, torch)
        mark('all_to_all')
        block = engine.rows_from_entries(e0, e1, bounds[rank], bounds[rank + 1], shape, recv_counts=recv)
        mark('rows')
        if timings is not None:
            for (_, t0), (name, t1) in zip(marks[:-1], marks[1:]):
                timings[name + '_ms'] = timings.get(name + '_ms', 0.0) + (t1 - t0) * 1e3
            timings['entries_sent'] = int(sum(counts))
    finally:
        engine.shard_close(st)
    fi = fidx.cpu().numpy().astype(np.int32)
    rest = np.flatnonzero((in_set != 0) & (fi < 0))                        # link-less members: trailing indices
    if rest_order is not None:
        rest_order = np.asarray(rest_order, np.int64)
        if sorted(rest_order.tolist()) != rest.tolist():
            raise ValueError('rest_order must list exactly the link-less members of in_set')
        rest = rest_order
    fi[rest] = n_linked + np.arange(len(rest), dtype=np.int32)
    return block, fi, n_linked, shape


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
