"""world_size-2 gloo tests of the multi-GPU driver (haphic_amd/sharded.py) on CPU.  The collective
logic (row-block shard, all-gather(v), max all-reduce, chunk-ordered table merge) is the product code;
the arithmetic is supplied by an oracle-backed engine that exists only in this test."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from haphic_amd import sharded, synth
from oracle import oracle as orc


class OracleEngine:
    """matrices are (indptr, indices, data) numpy triples; tensors are CPU torch views"""
    torch = torch
    device = torch.device('cpu')

    def tensors(self, m):
        return tuple(torch.from_numpy(np.ascontiguousarray(a)) for a in m)

    def from_tensors(self, n_rows, n_cols, indptr, indices, data):
        return (indptr.numpy().copy(), indices.numpy().copy(), data.numpy().copy())

    def pack_block(self, m, words):
        buf = torch.zeros(int(words), dtype=torch.int32)
        r, z = len(m[0]) - 1, int(m[0][-1])
        buf[:r] = torch.from_numpy(np.diff(m[0]).astype(np.int32))
        buf[r:r + z] = torch.from_numpy(np.ascontiguousarray(m[1], np.int32))
        buf[r + z:r + 2 * z] = torch.from_numpy(np.ascontiguousarray(m[2], np.float32).view(np.int32))
        return buf

    def unpack_blocks(self, rows, nnzs, packed, stride, n_cols):
        p = packed.numpy()
        lens, ix, dx = [], [], []
        for b, (r, z) in enumerate(zip(rows.tolist(), nnzs.tolist())):
            msg = p[b * stride:(b + 1) * stride]
            lens.append(msg[:r]); ix.append(msg[r:r + z]); dx.append(msg[r + z:r + 2 * z].view(np.float32))
        indptr = np.zeros(int(sum(rows)) + 1, np.int32)
        indptr[1:] = np.cumsum(np.concatenate(lens))
        return indptr, np.concatenate(ix).astype(np.int32), np.concatenate(dx).astype(np.float32)

    def shape(self, m):
        m = getattr(m, 'a', m)                   # (the link matrix of cluster.run_mcl_clustering under tests.oracle_lib)
        return len(m[0]) - 1, None, int(m[0][-1])

    def row_block(self, m, r0, r1):
        lo, hi = m[0][r0], m[0][r1]
        return (m[0][r0:r1 + 1] - lo).astype(np.int32), m[1][lo:hi].copy(), m[2][lo:hi].copy()

    def spgemm(self, a, b):
        lens = np.diff(b[0])
        return orc.spgemm(a, b, n_cols=len(b[0]) - 1, mode=1, fx_shift=52), int(lens[a[1]].sum())

    def inflate_prune(self, c, inflation, pruning):
        x = orc.normalize_l1(c[0], orc.power(c[2], inflation))
        return orc.prune((c[0], c[1], x), pruning)

    def expand_inflate_prune(self, a, b, inflation, pruning):
        c, f = self.spgemm(a, b)
        return self.inflate_prune(c, inflation, pruning), f, int(c[0][-1])

    def row_products(self, a, b):
        a, b = getattr(a, 'a', a), getattr(b, 'a', b)
        return np.add.reduceat(np.concatenate([np.diff(b[0])[a[1]].astype(np.int64), [0]]), np.minimum(a[0][:-1], len(a[1]))) * (np.diff(a[0]) > 0)

    def expand_links(self, links, r0, r1, inflation, pruning):
        norm = self.normalize_l1(links)
        return self.expand_inflate_prune(self.row_block(norm, r0, r1), norm, inflation, pruning)

    def convergence_stat(self, m, last):
        return orc.convergence_stat(m, last)

    # ---- the symmetric half across ranks (sharded.expand_links_symmetric): numpy stand-ins of hhx_expand_links_dense(upper_only) and
    # hhx_dense_inflate_prune.  CAP columns per window (a small number, so that the row blocks of the test cut several windows)
    CAP = 64

    def links_integer_ok(self, links):
        return orc.links_shift(links) > 0

    def dense_upper(self, links, r0, r1):
        self.upper_calls = getattr(self, 'upper_calls', 0) + 1
        n = len(links[0]) - 1
        y = orc.expand_links(links, rows=np.arange(r0, r1, dtype=np.int32), divide=False)
        dense = np.zeros((r1 - r0, n), np.float32)
        rows = np.repeat(np.arange(r1 - r0), np.diff(y[0]))
        dense[rows, y[1]] = y[2]
        for i in range(r0, r1):                                  # what upper_only leaves unwritten: the columns left of the row's own block
            dense[i - r0, :(i // self.CAP) * self.CAP] = np.nan
        sums = np.add.reduceat(links[2].astype(np.float64), links[0][:-1])
        yt = torch.from_numpy(dense)
        return {'y': yt, 'd': sums[r0:r1], 'f': int(np.diff(links[0])[links[1][links[0][r0]:links[0][r1]]].sum())}, yt, self.CAP

    def dense_finish(self, h, inflation, pruning):
        y = h['y'].numpy()
        assert not np.isnan(y).any(), 'a part of the block was never mirrored'
        x = (y.astype(np.float64) / h['d'][:, None]).astype(np.float32)
        rows, cols = np.nonzero(y)
        p = np.zeros(y.shape[0] + 1, np.int32)
        p[1:] = np.cumsum(np.bincount(rows, minlength=y.shape[0]))
        c = (p, cols.astype(np.int32), x[rows, cols])
        return self.inflate_prune(c, inflation, pruning), h['f'], int(p[-1])

    def dense_drop(self, h):
        pass

    def mirror_block(self, y, r_src, c_src, rows, cols, r_dst, c_dst):
        y[r_dst:r_dst + cols, c_dst:c_dst + rows] = y[r_src:r_src + rows, c_src:c_src + cols].t()

    def pack_columns(self, y, c0, c1, out):
        out.copy_(y[:, c0:c1].contiguous().view(-1))

    def unpack_transposed(self, y, c0, buf, rows_s):
        y[:, c0:c0 + rows_s] = buf.view(rows_s, y.shape[0]).t()

    # ---- the sweep shared out over the ranks (sharded.sweep_sharded): stand-ins of hhx_expand_links_dense on a row block,
    # hhx_dense_inflate_prune, hhx_interpret
    def dense_rows(self, links, r0, r1):
        links = getattr(links, 'a', links)
        self.dense_calls = getattr(self, 'dense_calls', 0) + 1
        norm = self.normalize_l1(links)
        c, f = self.spgemm(self.row_block(norm, r0, r1), norm)
        return c, f, int(c[0][-1])

    def dense_first(self, c, inflation, pruning):
        return self.inflate_prune(c, inflation, pruning)

    def dense_free(self, c):
        pass

    def interpret(self, m):
        return orc.interpret(m)

    def mcl_resume(self, m, done, expansion, inflation, iters, pruning):
        """mcl() :2026-2062 from iteration `done` on the whole matrix (hhx_mcl_resume)"""
        cur, n_iter, conv, stats = m, done, False, []
        for it in range(done, iters):
            run = cur
            for _ in range(2, expansion):
                run, _f = self.spgemm(run, cur)
            p, f, st_c = self.expand_inflate_prune(run, cur, inflation, pruning)
            stats.append([self.shape(cur)[2], st_c, self.shape(p)[2], f])
            n_iter = it + 1
            stop = it > 1 and np.float32(self.convergence_stat(p, cur)) <= np.float32(1e-8)
            cur = p
            if stop:
                conv = True
                break
        return cur, n_iter, conv, stats

    def copy(self, m):
        return tuple(a.copy() for a in m)

    def free(self, m):
        pass

    def sync(self):
        pass

    def normalize_l1(self, m):
        return (m[0], m[1], orc.normalize_l1(m[0], m[2]))

    # ---- sharded link-matrix build: numpy stand-ins of hhx_shard_* (src: this chunk's flank table with ordinals)
    def shard_open(self, src, in_set):
        ok = in_set[src['i']].astype(bool) & in_set[src['j']].astype(bool)
        return {'i': src['i'][ok], 'j': src['j'][ok], 'cnt': src['cnt'][ok], 'ord': src['ord'][ok], 'n_frag': len(in_set)}

    def shard_first(self, st):
        first = np.full(st['n_frag'], np.iinfo(np.int64).max, np.int64)
        np.minimum.at(first, st['i'], 2 * st['ord'])
        np.minimum.at(first, st['j'], 2 * st['ord'] + 1)
        return torch.from_numpy(first)

    def rank_first(self, first):
        f = first.numpy()
        linked = f != np.iinfo(np.int64).max
        fidx = np.full(len(f), -1, np.int32)
        order = np.flatnonzero(linked)[np.argsort(f[linked], kind='stable')]
        fidx[order] = np.arange(len(order), dtype=np.int32)
        return torch.from_numpy(fidx), int(linked.sum())

    def shard_emit(self, st, fidx, bounds):
        fx = fidx.numpy().astype(np.int64)
        row = np.concatenate([fx[st['i']], fx[st['j']]])
        col = np.concatenate([fx[st['j']], fx[st['i']]])
        cnt = np.concatenate([st['cnt'], st['cnt']]).astype(np.int64)
        o = np.argsort(row, kind='stable')
        row, col, cnt = row[o], col[o], cnt[o]
        counts = [int(((row >= bounds[k]) & (row < bounds[k + 1])).sum()) for k in range(len(bounds) - 1)]
        return torch.from_numpy((row << 29) | col), torch.from_numpy(cnt), counts

    def shard_close(self, st):
        pass

    def rows_from_entries(self, w0, w1, r0, r1, shape, recv_counts=None):
        import scipy.sparse as sp
        w0, w1 = w0.numpy(), w1.numpy()
        row, col = (w0 >> 29) - r0, w0 & ((1 << 29) - 1)
        m = sp.coo_matrix((w1.astype(np.float64), (row, col)), shape=(r1 - r0, shape)).tocsr()      # sums duplicates
        m = (m + sp.coo_matrix((np.ones(r1 - r0), (np.arange(r1 - r0), np.arange(r0, r1))), shape=(r1 - r0, shape))).tocsr()
        m.sort_indices()
        return m.indptr.astype(np.int32), m.indices.astype(np.int32), m.data.astype(np.float32)


def stochastic(n, deg, seed):
    import scipy.sparse as sp
    m = sp.random(n, n, density=deg / n, random_state=seed, dtype=np.float32, format='csr')
    blocks = sp.block_diag([sp.random(n // 4, n // 4, density=0.3, random_state=seed + k, dtype=np.float32) for k in range(4)])
    m = (m * 0.05 + blocks + blocks.T + sp.identity(n, dtype=np.float32)).tocsr()
    m.sort_indices()
    return m.indptr.astype(np.int32), m.indices.astype(np.int32), orc.normalize_l1(m.indptr, m.data)


def _worker(rank, world, port, q):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        sharded.MAX_MESSAGE_BYTES = 4096            # every collective of this test takes several rounds
        eng = OracleEngine()
        # ragged all-to-all(v) / all-gather(v) against plain arithmetic, including empty pieces
        def piece(src, dst):                          # what rank src sends to rank dst: empty on the diagonal + 1
            return torch.arange(0 if dst == (src + 1) % world else 700 * src + 300 * dst + 5, dtype=torch.int64) + 100000 * src + 1000 * dst
        send = torch.cat([piece(rank, d) for d in range(world)])
        got, rc = sharded._all_to_all_var(send, [piece(rank, d).numel() for d in range(world)], dist, torch)
        assert torch.equal(got, torch.cat([piece(s_, rank) for s_ in range(world)])) and rc == [piece(s_, rank).numel() for s_ in range(world)]
        parts = sharded._all_gather_var(send, dist, torch)
        assert all(torch.equal(parts[r], torch.cat([piece(r, d) for d in range(world)])) for r in range(world))
        # a transport that delivers a message SHORT without an error (what all_to_all_single of the RCCL stack did beyond 2^30 B per peer,
        # profiles/r03_rccl_probe.jsonl): every round is checksummed, the receiver raises
        real = dist.all_to_all_single

        def lossy(out, inp, output_split_sizes=None, input_split_sizes=None):
            real(out, inp, output_split_sizes=output_split_sizes, input_split_sizes=input_split_sizes)
            if output_split_sizes is not None and out.numel() > 8 and out.dtype == torch.int64 and out.numel() != world:
                out[-3:] = 0                        # the tail of the last piece never arrived
        dist.all_to_all_single = lossy
        try:
            failed = False
            try:
                sharded._all_to_all_var(send, [piece(rank, d).numel() for d in range(world)], dist, torch)
            except RuntimeError as e:
                failed = 'did not arrive intact' in str(e)
            # (a rank whose received data fit in 8 elements is not touched by `lossy`: it must come through clean; the others must raise,
            # and only after the last round — every rank runs the same sequence of collectives)
            assert failed == (sum(piece(s_, rank).numel() for s_ in range(world)) > 8), 'a truncated all-to-all went unnoticed'
        finally:
            dist.all_to_all_single = real
        dist.barrier()
        # ---- MCL row-block shard
        T = stochastic(400, 6, 5)
        sharded.STAGES = {}                                   # per-stage clocks for bench.py --gpus N: every stage of the path must report
        res, n_iter, conv, stats = sharded.mcl_sharded_engine(eng, T, 2, 2.0, 100, 1e-4, dist, replicate_nnz=0)     # every iteration sharded
        st_rec, sharded.STAGES = sharded.STAGES, None
        assert set(st_rec) == {'mcl_sharded_iteration_compute', 'mcl_sharded_iteration_exchange'}, st_rec
        assert st_rec['mcl_sharded_iteration_exchange'][2] == n_iter and st_rec['mcl_sharded_iteration_exchange'][1] > 0 and st_rec['mcl_sharded_iteration_compute'][0] > 0
        # replicated tail: the row-block iterations until the matrix is small, then every rank finishes alone — same everything
        thr = int(stats[2][0])                                # entries of the matrix that enters iteration 2
        res_r, n_iter_r, conv_r, stats_r = sharded.mcl_sharded_engine(eng, T, 2, 2.0, 100, 1e-4, dist, replicate_nnz=thr)
        assert (n_iter_r, conv_r) == (n_iter, conv) and all(np.array_equal(x, y) for x, y in zip(res_r, res))
        assert np.array_equal(np.asarray(stats_r), np.asarray(stats))
        # ---- sharded ingest: each rank owns one contiguous chunk of the stream
        gen = synth.make_genome(3, 400_000, 10_000, seed=2)
        n = gen.n
        lex = gen.lexical_rank()
        t = orc.FragTable(lex, gen.length, np.arange(n, dtype=np.int32), np.zeros(n, np.uint8), 0, lex, gen.length,
                          np.ones(n, np.uint8))
        id1, p1, id2, p2 = [a.numpy() for a in synth.sample_pairs(gen, 40_000, seed=3)]
        keep = id1 != id2
        a = [x[keep] for x in (id1, p1, id2, p2)]
        cuts = _chunk_cuts(len(a[0]), world)
        sl = slice(cuts[rank], cuts[rank + 1])
        loc = orc.ingest(t, a[0][sl], a[1][sl].astype(np.int64), a[2][sl], a[3][sl].astype(np.int64), 3000)
        # this rank's aggregated table: key, first-seen ordinal (chunk base + position in the chunk's
        # insertion order: order-preserving, and earlier chunks win), count
        base = cuts[rank]
        key = (loc['flank_i'].astype(np.int64) << 29) | loc['flank_j'].astype(np.int64)
        ordk = base + np.arange(len(key), dtype=np.int64)
        g = sharded.gather_tables(eng, [torch.from_numpy(key), torch.from_numpy(ordk), torch.from_numpy(loc['flank_cnt'])], dist)
        gk, go, gc = [x.numpy() for x in g]
        uniq, inv = np.unique(gk, return_inverse=True)
        cnt = np.bincount(inv, weights=gc.astype(np.float64)).astype(np.int64)
        first = np.full(len(uniq), np.iinfo(np.int64).max)
        np.minimum.at(first, inv, go)
        order = np.argsort(first, kind='stable')
        merged = (len(uniq), (uniq[order] >> 29).astype(np.int32), (uniq[order] & ((1 << 29) - 1)).astype(np.int32), cnt[order])
        # ---- the row-owner build of the link matrix from the same two chunks (all-reduce(min) + all-to-all(v))
        in_set = np.ones(n, np.uint8)
        in_set[::11] = 0                                               # a filtered fragment set
        src = {'i': loc['flank_i'], 'j': loc['flank_j'], 'cnt': loc['flank_cnt'], 'ord': ordk}
        block, fi, n_linked, shape = sharded.build_link_matrix_sharded(eng, src, in_set, dist)
        norm = eng.normalize_l1(block)
        res2, n_iter2, conv2, _st2 = sharded.mcl_sharded_engine(eng, None, 2, 2.0, 100, 1e-4, dist, local_block=norm, n=shape)
        # the same from the RAW row blocks (all-gathered once, iteration 0 through expand_links): identical results
        res3, n_iter3, conv3, _st3 = sharded.mcl_sharded_engine(eng, None, 2, 2.0, 100, 1e-4, dist, local_links=block, n=shape)
        assert (n_iter3, conv3) == (n_iter2, conv2) and all(np.array_equal(x, y) for x, y in zip(res3, res2))
        # --expansion 3 from the raw blocks (mkl_matrix_power :2017-2023 recurses for any e): the blocks are normalised and the
        # general path runs — equal to the single-process oracle on the stacked matrix
        res5, n_iter5, conv5, _st5 = sharded.mcl_sharded_engine(eng, None, 3, 2.0, 100, 1e-4, dist, local_links=block, n=shape)
        fl = sharded.allgather_rows(eng, block, shape, dist)
        T5 = eng.normalize_l1(fl)
        pre5 = orc.spgemm(orc.spgemm(T5, T5, mode=1, fx_shift=52), T5, mode=1, fx_shift=52)
        o5 = orc.mcl(pre5, 3, 2.0, 100, 1e-4, spgemm_mode=1, fx_shift=52)
        assert (n_iter5, conv5) == (o5[3], o5[4]) and all(np.array_equal(x, y) for x, y in zip(res5, o5[:3])), 'expansion 3 from raw row blocks'
        # ... and with iteration 0 on the SYMMETRIC HALF shared out over the ranks (each rank fills the upper blocks of its rows, one
        # all-to-all(v) mirrors them): the integer specification of the pre-expansion, so the reference run is the oracle's own
        sharded.SYMMETRIC_MIN_WORLD = 2
        old_window = sharded.symmetric_window
        sharded.symmetric_window = lambda engine, n_: OracleEngine.CAP
        try:
            eng.upper_calls = 0
            res4, n_iter4, conv4, st4 = sharded.mcl_sharded_engine(eng, None, 2, 2.0, 100, 1e-4, dist, local_links=block, n=shape)
            assert eng.upper_calls == 1, 'the symmetric path was not taken'
        finally:
            sharded.symmetric_window = old_window
            sharded.SYMMETRIC_MIN_WORLD = None
        full_links = sharded.allgather_rows(eng, block, shape, dist)
        o4 = orc.mcl(orc.expand_links(full_links), 2, 2.0, 100, 1e-4, spgemm_mode=1, fx_shift=52, want_stats=True)
        assert (n_iter4, conv4) == (o4[3], o4[4]) and all(np.array_equal(x, y) for x, y in zip(res4, o4[:3])), 'symmetric half across ranks'
        assert np.array_equal(np.asarray(st4)[1:, 1:], o4[5][1:, 1:]) and np.array_equal(np.asarray(st4)[0, 1:3], o4[5][0, 1:3])
        built = (block, fi, n_linked, shape, res2, n_iter2, conv2)
        sweep = sharded.inflation_sweep(lambda infl: (round(infl * 10), rank), [1.2, 1.4, 1.6, 1.8, 2.0], dist)
        q.put((rank, res, n_iter, conv, stats, merged, sweep, built))
    finally:
        dist.destroy_process_group()


def _sweep_worker(rank, world, port, outdir, q, shard_products=None, expansion=2):
    """run_mcl_clustering(dist=...): the sweep shared out over the ranks (sharded.sweep_sharded; expansion 3: whole inflations dealt
    round-robin), rank 0 writes the files"""
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from haphic_amd import cluster
        from tests import oracle_lib
        from tests.conftest import load_golden
        cluster._lib = oracle_lib
        g = load_golden('pipeline_toy.npz')
        names = [str(x) for x in g['names']]
        fa_dict = {n_: [None, int(l), int(r)] for n_, l, r in zip(names, g['length'], g['re_sites'])}
        aln = ((names[a], names[b], int(x), int(y)) for a, x, b, y in zip(g['id1'], g['pos1'], g['id2'], g['pos2']))

        class A:
            flank = 500
            remove_allelic_links = 0
            remove_concentrated_links = False
            max_read_pairs = 200
            nwindows = 50
        frag_len_dict = {n_: fa_dict[n_][1] for n_ in names}
        full, flank, HT, clm, frag_link, coord = cluster.parse_alignments_for_ctgs(aln, fa_dict, A(), frag_len_dict, set(names), 'int32', 'int32')
        mat, fidx = cluster.dict_to_matrix(flank, set(names), dense_matrix=False, add_self_loops=True, _device=True)
        eng = OracleEngine()
        if shard_products is not None:
            sharded.SWEEP_SHARD_PRODUCTS = shard_products
        records = []
        import logging
        handler = logging.Handler()
        handler.emit = lambda rec: records.append(rec.getMessage())
        cluster.logger.addHandler(handler)
        cluster.logger.setLevel('INFO')
        shared = []
        orig = sharded.sharded_iteration
        sharded.sharded_iteration = lambda *a, **k: (shared.append(1), orig(*a, **k))[1]
        res, nrounds = cluster.run_mcl_clustering(mat, set(), frag_len_dict, fidx, expansion, 1.2, 2.0, 0.4, 200, 1e-4, fa_dict, int(g['nchrs']), False,
                                                  outdir_root=outdir, dist=dist, _engine=eng)
        q.put((rank, nrounds, [(str(i), [(list(c), l) for c, l in r]) for i, r in res], [m_ for m_ in records if 'rounds of iterations' in m_],
               len(shared), getattr(eng, 'dense_calls', 0)))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
@pytest.mark.parametrize('world,shard_products', [(2, None), (3, 2000.0), (2, 0.0)])
def test_inflation_sweep_across_ranks_writes_the_reference_files(tmp_path, world, shard_products):
    """sharded.sweep_sharded through run_mcl_clustering: ONE expansion shared by the ranks (every rank expands only its rows), the
    heavy iterations row-sharded, the light remainders dealt to the ranks; shard_products = None: the product threshold (every
    tail of this toy is light: replicas from iteration 1 on), 2000: the first iterations of every inflation are shared, 0: every
    iteration of every inflation is (the tasks finish inside the sharded phase).  The files must be the reference's in every mode."""
    from tests.conftest import load_golden
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_sweep_worker, args=(r, world, port, str(tmp_path), q, shard_products)) for r in range(world)]
    for p in procs:
        p.start()
    outs = sorted([q.get(timeout=240) for _ in range(world)], key=lambda o: o[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert all(o[1:4] == outs[0][1:4] for o in outs) and outs[0][1] == 3            # every rank returns the whole sweep, logs the same lines
    assert len(outs[0][3]) == 3 and all('converged after' in l for l in outs[0][3])
    assert all(o[5] == 1 for o in outs), 'one expansion per rank, of its own rows'
    if shard_products is None:
        assert all(o[4] == 0 for o in outs)
    else:
        assert all(o[4] == outs[0][4] and o[4] >= 3 for o in outs), 'every rank takes part in every shared iteration'
    g = load_golden('pipeline_toy.npz')
    for infl in g['inflations']:
        infl = str(infl)
        d = tmp_path / ('inflation_' + infl)
        assert (d / 'mcl_inflation_{}.clusters.txt'.format(infl)).read_text() == str(g['clusters_txt_' + infl])
        groups = sorted(f for f in os.listdir(d) if f.startswith('group'))
        assert groups == [str(x) for x in g['group_files_' + infl]]
        assert (d / groups[0]).read_text() == str(g['group0_txt_' + infl])


def _chunk_cuts(n, world):
    """uneven contiguous chunks of the pair stream; the last rank gets only a handful of pairs"""
    cuts = [0] + [n * (k + 1) // world + 17 for k in range(world - 1)] + [n]
    if world > 2:
        cuts[-2] = n - 5
    return cuts


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.timeout(300)
@pytest.mark.parametrize('world', [2, 3])
def test_sharded_mcl_and_merge(world):
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    outs = sorted([q.get(timeout=240) for _ in range(world)], key=lambda o: o[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    # single-process truth
    T = stochastic(400, 6, 5)
    pre = orc.spgemm(T, T, mode=1, fx_shift=52)
    o = orc.mcl(pre, 2, 2.0, 100, 1e-4, spgemm_mode=1, fx_shift=52, want_stats=True)
    for rank, res, n_iter, conv, stats, merged, sweep, _built in outs:
        assert sweep == [(12 + 2 * k, k % world) for k in range(5)]        # inflations dealt round-robin, results in order
        assert (n_iter, conv) == (o[3], o[4])
        assert all(np.array_equal(x, y) for x, y in zip(res, o[:3])), 'sharded MCL is not bit-identical to 1 process'
        # iteration 0 of the sharded driver includes the fused pre-expansion: nnz entering = nnz(T), F > 0
        assert np.array_equal(stats[1:], o[5][1:]) and np.array_equal(stats[0, 1:3], o[5][0, 1:3])
        assert stats[0, 0] == T[0][-1] and stats[0, 3] == int(np.diff(T[0])[T[1]].sum())
    gen = synth.make_genome(3, 400_000, 10_000, seed=2)
    n = gen.n
    lex = gen.lexical_rank()
    t = orc.FragTable(lex, gen.length, np.arange(n, dtype=np.int32), np.zeros(n, np.uint8), 0, lex, gen.length, np.ones(n, np.uint8))
    id1, p1, id2, p2 = [a.numpy() for a in synth.sample_pairs(gen, 40_000, seed=3)]
    keep = id1 != id2
    whole = orc.ingest(t, id1[keep], p1[keep].astype(np.int64), id2[keep], p2[keep].astype(np.int64), 3000)
    for rank, _res, _n, _c, _st, (k, mi, mj, mc), _sw, _built in outs:
        assert k == len(whole['flank_i'])
        assert np.array_equal(mi, whole['flank_i']) and np.array_equal(mj, whole['flank_j']) and np.array_equal(mc, whole['flank_cnt'])


    # row-owner build: the two row blocks stacked == dict_to_matrix of the whole stream; MCL from the blocks == 1 process
    in_set = np.ones(n, np.uint8)
    in_set[::11] = 0
    ok = in_set[whole['flank_i']].astype(bool) & in_set[whole['flank_j']].astype(bool)
    linked = np.zeros(n, bool)
    linked[whole['flank_i'][ok]] = True
    linked[whole['flank_j'][ok]] = True
    n_rest = int(in_set.sum() - linked.sum())
    rp, rj, rx, ridx, rl = orc.dict_to_matrix(whole['flank_i'], whole['flank_j'], whole['flank_cnt'].astype(np.float64), n, in_set, n_rest)
    blocks = [o[7][0] for o in outs]
    offs = np.cumsum([0] + [b[0][-1] for b in blocks])
    assert np.array_equal(np.concatenate([b[0][:-1] + o_ for b, o_ in zip(blocks, offs)] + [offs[-1:]]), rp)
    assert np.array_equal(np.concatenate([b[1] for b in blocks]), rj) and np.array_equal(np.concatenate([b[2] for b in blocks]), rx)
    T2 = (rp, rj, orc.normalize_l1(rp, rx))
    o2 = orc.mcl(orc.spgemm(T2, T2, mode=1, fx_shift=52), 2, 2.0, 100, 1e-4, spgemm_mode=1, fx_shift=52)
    for o in outs:
        _b, fi, n_linked, shape, res2, n_iter2, conv2 = o[7]
        assert (n_linked, shape) == (rl, len(rp) - 1) and np.array_equal(fi[ridx >= 0][fi[ridx >= 0] < rl], ridx[ridx >= 0][ridx[ridx >= 0] < rl])
        assert np.array_equal(fi >= 0, in_set.astype(bool))
        assert (n_iter2, conv2) == (o2[3], o2[4]) and all(np.array_equal(x, y) for x, y in zip(res2, o2[:3]))


def test_row_ranges():
    assert sharded.row_ranges(10, 3) == [0, 4, 7, 10]
    assert sharded.row_ranges(2, 4) == [0, 1, 2, 2, 2]
    assert sharded.balanced_ranges([5, 1, 1, 1, 1, 1], 2) == [0, 1, 6]
    assert sharded.balanced_ranges([1, 1, 1, 1, 1, 5], 2) == [0, 5, 6]
    assert sharded.balanced_ranges([0, 0, 0], 2) == [0, 1, 3] and sharded.balanced_ranges([], 3) == [0, 0, 0, 0]
    # a few very heavy rows: no rank is left without rows while there are rows to give
    assert sharded.balanced_ranges([100, 1], 2) == [0, 1, 2]
    assert sharded.balanced_ranges([100, 1, 1, 1], 4) == [0, 1, 2, 3, 4]
    assert sharded.balanced_ranges([1, 1, 1, 100], 4) == [0, 1, 2, 3, 4]
    assert sharded.balanced_ranges([100, 1], 4) == [0, 1, 2, 2, 2]
    for w in (2, 3, 8):
        bb = sharded.balanced_ranges([10 ** 9, 10 ** 9] + [1] * 20, w)
        assert bb[0] == 0 and bb[-1] == 22 and all(y > x for x, y in zip(bb, bb[1:])), bb
    b = sharded.balanced_ranges(np.arange(1000), 4)
    c = np.cumsum(np.arange(1000))
    assert b[0] == 0 and b[-1] == 1000 and all(x <= y for x, y in zip(b, b[1:]))
    per = [int(np.arange(1000)[b[k]:b[k + 1]].sum()) for k in range(4)]
    assert max(per) - min(per) <= 2 * 999


def _sweep_edge_worker(rank, world, port, q):
    """sharded.sweep_sharded against the same engine on one rank: iteration limits 1 / 2 / 3 / 200 (a limit that ends an inflation
    inside the shared phase, at its hand-over, and in the light remainder), more ranks than inflations, a threshold of zero"""
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        eng = OracleEngine()
        # a symmetric integer link matrix with unit self loops (what dict_to_matrix builds), three planted groups
        rng = np.random.default_rng(3)
        n = 150
        grp = rng.integers(0, 3, n)
        iu, ju = np.triu_indices(n, 1)
        keep = rng.random(iu.size) < np.where(grp[iu] == grp[ju], 0.5, 0.03)
        iu, ju = iu[keep], ju[keep]
        cnt = rng.integers(1, 9, iu.size).astype(np.float32)
        import scipy.sparse as sp
        m = sp.coo_matrix((np.concatenate([cnt, cnt, np.ones(n, np.float32)]), (np.concatenate([iu, ju, np.arange(n)]), np.concatenate([ju, iu, np.arange(n)]))),
                          shape=(n, n)).tocsr()
        m.sort_indices()
        links = (m.indptr.astype(np.int32), m.indices.astype(np.int32), m.data.astype(np.float32))
        out = []

        class NoRoom(OracleEngine):
            """the last rank cannot hold its rows of M^2: every rank must fall back to the fused iteration 0 together"""
            def dense_rows(self, links_, r0, r1):
                if rank == world - 1:
                    raise RuntimeError('hipMalloc: out of memory (simulated)')
                return OracleEngine.dense_rows(self, links_, r0, r1)
        short = NoRoom()
        got = sharded.sweep_sharded(short, links, [1.4, 2.0], 200, 1e-4, dist, shard_products=200.0)
        ref = sharded.sweep_sharded(eng, links, [1.4, 2.0], 200, 1e-4, dist, shard_products=200.0)
        for a_, b_ in zip(got, ref):
            assert all(np.array_equal(x, y) for x, y in zip(a_[:3], b_[:3])) and tuple(a_[3:]) == tuple(b_[3:])
        for inflations, iters, thr in (((1.4, 2.0), 200, None), ((1.4, 2.0), 1, 0.0), ((1.4, 2.0), 2, 0.0), ((1.4, 2.0), 3, 0.0), ((1.4, 2.0, 3.0), 3, 1e9),
                                       ((2.0,), 200, 0.0), ((1.2, 1.6, 2.0, 2.4, 2.8), 200, 200.0)):
            got = sharded.sweep_sharded(eng, links, list(inflations), iters, 1e-4, dist, shard_products=thr)
            d, _f, _c = eng.dense_rows(links, 0, n)
            for infl, g_ in zip(inflations, got):
                first = eng.dense_first(d, infl, 1e-4)
                if iters <= 1:
                    want_m, want_it, want_conv = first, 1, False
                else:
                    want_m, want_it, want_conv, _st = eng.mcl_resume(first, 1, 2, infl, iters, 1e-4)
                wa, wp, wm = eng.interpret(want_m)
                assert (g_[4], bool(g_[5])) == (want_it, bool(want_conv)), (inflations, iters, thr, infl, g_[4], g_[5], want_it, want_conv)
                assert g_[3] == n and np.array_equal(g_[0], wa) and np.array_equal(g_[1], wp) and np.array_equal(g_[2], wm), (inflations, iters, thr, infl)
            out.append(len(got))
        q.put((rank, out))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
@pytest.mark.parametrize('world', [2, 4])
def test_sweep_sharded_edge_cases(world):
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_sweep_edge_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    outs = sorted([q.get(timeout=240) for _ in range(world)], key=lambda o: o[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert all(o[1] == [2, 2, 2, 2, 3, 1, 5] for o in outs)
