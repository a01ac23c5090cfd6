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


def test_cluster_files_byte_identical(golden_pipeline, tmp_path):
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
                                                  int(g['nchrs']), False, outdir_root=str(tmp_path))
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
        m, n_linked, merged = sharded.merge_flank_and_build(ing, t, 3000, False, in_set, dist, 'cuda:0')
        m1, fidx, nl1 = ing.link_matrix(in_set)
        assert merged.n_flank == ing.n_flank and nl1 == n_linked
        assert all(np.array_equal(x, y) for x, y in zip(m.to_arrays(), m1.to_arrays()))
        a, b = merged.fetch(), ing.fetch()
        for k in a:
            assert np.array_equal(a[k], b[k]), k
    finally:
        dist.destroy_process_group()
