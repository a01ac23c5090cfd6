"""The HIP engine under SEVERAL ranks (VERDICT r01 "Next round" #3): 2 and 3 processes share the one GPU of the test
box and run the multi-GPU driver unchanged (haphic_amd/sharded.py: chunked ingest with global ordinals, all-reduce(min)
+ all-to-all(v) row-owner build of the link matrix, all-gather(v) of the raw row blocks, iteration 0 on the SYMMETRIC HALF shared
out over the ranks (each rank fills the upper blocks of its rows, one all-to-all(v) mirrors them) and, for comparison, per row
block over all products, the two-collective exchange per iteration); the collectives travel through host memory over gloo
(haphic_amd.host_transport.HostStagedCollectives) because RCCL refuses two ranks on one device.  BASELINE configs[1] size; everything
must be bit-identical to the one-rank result."""
import os
import socket

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

C2 = (624, 16, 50_000, 50_000_000)                 # BASELINE configs[1]: one column window
WIDE = (3000, 8, 20_000, 20_000_000)               # 24k contigs: TWO column windows, the row blocks of the ranks cut them


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _cuts(n, world):
    """uneven contiguous chunks of the pair stream"""
    c = [0] + [n * (k + 1) // world + 1017 * (k + 1) for k in range(world - 1)] + [n]
    return c


def _setup(cfg):
    import torch
    from haphic_amd import _lib, synth
    from haphic_amd.cluster import FragTable
    per_chr, nchrs, mean, n_pairs = cfg
    torch.cuda.set_device(0)
    _lib.check(_lib.load().hhx_set_device(0))
    gen = synth.make_genome(nchrs, per_chr * mean, mean, seed=12345)
    table = FragTable.for_contigs(gen.lexical_rank(), gen.length, np.ones(gen.n, np.uint8))
    pairs = synth.sample_pairs(gen, n_pairs, seed=12345, device='cuda:0')       # the SAME stream in every process
    torch.cuda.synchronize()
    return gen, table, pairs


def _worker(rank, world, port, q, cfg):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    import torch
    import torch.distributed as dist
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from haphic_amd import _lib, host_transport, sharded
        gen, table, pairs = _setup(cfg)
        PAIRS = cfg[3]
        hd = host_transport.HostStagedCollectives(dist)
        sharded.SYMMETRIC_MIN_WORLD = 2                  # (on from 8 ranks by default: the bandwidth model of sharded.py)
        cuts = _cuts(PAIRS, world)
        lo, hi = cuts[rank], cuts[rank + 1]
        ing = _lib.Ingest(table, 500_000, bins=False, skip_intra=True)
        ing.set_ordinal_base(lo)
        ing.push_device(hi - lo, *[x[lo:hi].data_ptr() for x in pairs])
        torch.cuda.synchronize()
        ing.finalize()
        del pairs
        eng = sharded.HipEngine('cuda:0')
        in_set = np.ones(gen.n, np.uint8)
        block, fi, n_linked, shape = sharded.build_link_matrix_sharded(eng, ing, in_set, hd)
        ing.destroy()
        blk = block.to_arrays()
        # world 2: every iteration sharded; world 3: the row-block iterations until the matrix has < 300k entries, the rest replicated
        res, n_iter, conv, stats = sharded.mcl_sharded_engine(eng, None, 2, 2.0, 200, 1e-4, hd, local_links=block, n=shape,
                                                              replicate_nnz=0 if world == 2 else 300_000)
        # iteration 0 went over the symmetric half shared out over the ranks (expand_links_symmetric); the plain row-block path
        # (every rank walks all products of its rows) must give the same bits
        sharded.SYMMETRIC_HALF = False
        res_b, n_iter_b, conv_b, stats_b = sharded.mcl_sharded_engine(eng, None, 2, 2.0, 200, 1e-4, hd, local_links=block, n=shape,
                                                                      replicate_nnz=0 if world == 2 else 300_000)
        sharded.SYMMETRIC_HALF = True
        assert (n_iter_b, conv_b) == (n_iter, conv) and all(np.array_equal(x, y) for x, y in zip(res_b.to_arrays(), res.to_arrays()))
        assert np.array_equal(np.asarray(stats_b), np.asarray(stats))
        q.put((rank, blk, fi, n_linked, shape, res.to_arrays(), n_iter, conv, stats))
        hd.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(400)
@pytest.mark.parametrize('world,cfg', [(2, C2), (3, C2), (3, WIDE), (2, WIDE), (8, WIDE)])      # 8: the rank count of one node
def test_hip_engine_under_several_ranks_on_one_gpu(world, cfg):
    import torch
    import torch.multiprocessing as mp
    from haphic_amd import _lib
    gen, table, pairs = _setup(cfg)
    PAIRS = cfg[3]
    one = _lib.Ingest(table, 500_000, bins=False, skip_intra=True)
    one.push_device(PAIRS, *[x.data_ptr() for x in pairs])
    torch.cuda.synchronize()
    one.finalize()
    del pairs
    m, fidx, n_linked = one.link_matrix(np.ones(gen.n, np.uint8))
    one.destroy()
    mp_, mj, mx = m.to_arrays()
    want, n_iter, conv, stats = _lib.mcl(m, 2, 2.0, 200, 1e-4, want_stats=True, links=True)
    wp, wj, wx = want.to_arrays()
    want.free()
    m.free()
    torch.cuda.empty_cache()
    _lib.check(_lib.load().hhx_pool_trim())
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q, cfg)) for r in range(world)]
    for p in procs:
        p.start()
    import queue
    import time
    outs, t0 = [], time.time()
    while len(outs) < world:                                  # a dead worker must fail the test at once, not at a timeout
        try:
            outs.append(q.get(timeout=2))
        except queue.Empty:
            dead = [p.exitcode for p in procs if p.exitcode not in (None, 0)]
            if dead or time.time() - t0 > 360:
                for p in procs:
                    p.kill()
                raise AssertionError('worker exit codes %r after %.0f s' % ([p.exitcode for p in procs], time.time() - t0))
    outs.sort(key=lambda o: o[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    # the row blocks stacked are the one-rank link matrix, bit for bit; the index map too
    blocks = [o[1] for o in outs]
    offs = np.cumsum([0] + [int(b[0][-1]) for b in blocks])
    assert np.array_equal(np.concatenate([b[0][:-1] + o_ for b, o_ in zip(blocks, offs)] + [offs[-1:]]).astype(np.int64), mp_.astype(np.int64))
    assert np.array_equal(np.concatenate([b[1] for b in blocks]), mj) and np.array_equal(np.concatenate([b[2] for b in blocks]), mx)
    for o in outs:
        assert o[3] == n_linked and o[4] == len(mp_) - 1 and np.array_equal(o[2], fidx)
        rp, rj, rx = o[5]
        assert (o[6], o[7]) == (n_iter, conv), 'iteration count / convergence flag under %d ranks' % world
        assert np.array_equal(rp, wp) and np.array_equal(rj, wj) and np.array_equal(rx, wx), 'MCL result under %d ranks differs' % world
        assert np.array_equal(o[8][:, 1:], stats[:, 1:])                       # nnz(C), nnz(P), products of every iteration
    # the MCL row blocks are cut at equal PRODUCT counts (hhx_row_products + balanced_ranges); equal row counts would not
    # be balanced — a contig's matrix index is its first-seen rank in the pair stream, heavily linked contigs come first
    lens = np.diff(mp_).astype(np.int64)
    F = np.add.reduceat(lens[mj], mp_[:-1])
    from haphic_amd.sharded import balanced_ranges, row_ranges
    b = balanced_ranges(F, world)
    per = np.array([F[b[k]:b[k + 1]].sum() for k in range(world)], np.float64)
    assert per.max() / per.mean() < 1.005, per
    e = row_ranges(len(lens), world)
    per_rows = np.array([F[e[k]:e[k + 1]].sum() for k in range(world)], np.float64)
    assert per_rows.max() / per_rows.mean() > per.max() / per.mean()


def _sweep_setup(cfg):
    """the link matrix of `cfg` on this process's device + the host containers run_mcl_clustering needs"""
    import torch
    from haphic_amd import _lib
    gen, table, pairs = _setup(cfg)
    ing = _lib.Ingest(table, 500_000, bins=False, skip_intra=True)
    ing.push_device(cfg[3], *[x.data_ptr() for x in pairs])
    torch.cuda.synchronize()
    ing.finalize()
    del pairs
    m, fidx, n_linked = ing.link_matrix(np.ones(gen.n, np.uint8))
    ing.destroy()
    torch.cuda.empty_cache()
    names = list(gen.names)
    fa_dict = {n_: [None, int(l), int(r)] for n_, l, r in zip(names, gen.length, gen.re_sites)}
    frag_len_dict = {n_: fa_dict[n_][1] for n_ in names}
    frag_index = {n_: int(i) for n_, i in zip(names, fidx)}
    return gen, m, fa_dict, frag_len_dict, frag_index


SWEEP = (1.4, 3.0, 0.8)                            # inflations 1.4, 2.2, 3.0
SWEEP_CFG = (3000, 8, 20_000, 8_000_000)           # 24k contigs (two column windows), fewer pairs than WIDE: eight processes ingest it side by side


def _sweep_worker(rank, world, port, q, cfg, outdir):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    import torch.distributed as dist
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from haphic_amd import cluster, host_transport, sharded
        gen, m, fa_dict, frag_len_dict, frag_index = _sweep_setup(cfg)
        hd = host_transport.HostStagedCollectives(dist)
        sharded.SWEEP_SHARD_PRODUCTS = 3e8             # at 24k contigs / 8 M pairs: the first iterations of the lower inflations are shared
        shared = []
        orig = sharded.sharded_iteration
        sharded.sharded_iteration = lambda *a, **k: (shared.append(1), orig(*a, **k))[1]
        res, nrounds = cluster.run_mcl_clustering(m, set(), frag_len_dict, frag_index, 2, SWEEP[0], SWEEP[1], SWEEP[2], 200, 1e-4, fa_dict, cfg[1],
                                                  False, outdir_root=outdir, dist=hd, _engine=sharded.HipEngine('cuda:0'))
        q.put((rank, nrounds, len(shared), [(str(i), [(list(c), l) for c, l in r]) for i, r in res]))
        hd.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(600)
@pytest.mark.parametrize('world', [2, 8])
def test_inflation_sweep_shared_out_over_the_ranks(world, tmp_path):
    """run_mcl_clustering(dist=...) -> sharded.sweep_sharded with the HIP engine under 2 and 8 ranks on the one GPU (host-staged
    collectives): ONE expansion shared by the ranks (every rank holds only its rows of M^2), the heavy iterations row-sharded, the
    light remainders dealt by predicted cost — the files rank 0 writes must be byte-identical to the one-GPU sweep's."""
    import queue
    import time
    import torch
    import torch.multiprocessing as mp
    from haphic_amd import _lib, cluster
    cfg = SWEEP_CFG
    gen, m, fa_dict, frag_len_dict, frag_index = _sweep_setup(cfg)
    one_dir, many_dir = tmp_path / 'one', tmp_path / 'many'
    # the one-GPU sweep through the same arithmetic (DenseSweep: the integer pre-expansion as float32 rows; _block_rows forces it at this order)
    want, nrounds = cluster.run_mcl_clustering(m, set(), frag_len_dict, frag_index, 2, SWEEP[0], SWEEP[1], SWEEP[2], 200, 1e-4, fa_dict, cfg[1], False,
                                               outdir_root=str(one_dir), _block_rows=m.shape3[0])
    m.free()
    torch.cuda.empty_cache()
    _lib.check(_lib.load().hhx_pool_trim())
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_sweep_worker, args=(r, world, port, q, cfg, str(many_dir))) for r in range(world)]
    for p in procs:
        p.start()
    outs, t0 = [], time.time()
    while len(outs) < world:
        try:
            outs.append(q.get(timeout=2))
        except queue.Empty:
            dead = [p.exitcode for p in procs if p.exitcode not in (None, 0)]
            if dead or time.time() - t0 > 540:
                for p in procs:
                    p.kill()
                raise AssertionError('worker exit codes %r after %.0f s' % ([p.exitcode for p in procs], time.time() - t0))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    outs.sort(key=lambda o: o[0])
    want_l = [(str(i), [(list(c), l) for c, l in r]) for i, r in want]
    assert all(o[1] == nrounds and o[3] == want_l for o in outs), 'result_clusters_list under %d ranks' % world
    assert all(o[2] == outs[0][2] for o in outs) and outs[0][2] >= 2, 'shared iterations: %r' % [o[2] for o in outs]
    dirs = sorted(d for d in os.listdir(one_dir) if d.startswith('inflation_'))
    assert len(dirs) == 3 and dirs == sorted(d for d in os.listdir(many_dir) if d.startswith('inflation_'))
    for d in dirs:
        files = sorted(os.listdir(one_dir / d))
        assert files == sorted(os.listdir(many_dir / d))
        for f in files:
            assert (one_dir / d / f).read_bytes() == (many_dir / d / f).read_bytes(), '%s/%s differs under %d ranks' % (d, f, world)
