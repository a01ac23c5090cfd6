"""stage-by-stage run of the row-owner link-matrix build on one GPU (world-1 nccl), with a sync after every stage"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import torch.distributed as dist
from haphic_amd import _lib, sharded, synth
from haphic_amd.cluster import FragTable
n_ctg, pairs = int(sys.argv[1]), int(sys.argv[2])
os.environ.setdefault('MASTER_ADDR', '127.0.0.1'); os.environ.setdefault('MASTER_PORT', '29591')
dist.init_process_group('nccl', rank=0, world_size=1, device_id=torch.device('cuda:0'))
gen = synth.make_genome(24, max(1, n_ctg // 24) * 30_000, 30_000, seed=12345)
table = FragTable.for_contigs(gen.lexical_rank(), gen.length, np.ones(gen.n, np.uint8))
a = synth.sample_pairs(gen, pairs, seed=1, device='cuda:0')
ing = _lib.Ingest(table, 500_000, bins=False, skip_intra=True)
ing.push_device(pairs, *[x.data_ptr() for x in a]); ing.finalize()
in_set = np.ones(gen.n, np.uint8)
sync = lambda tag: (_lib.check(_lib.load().hhx_synchronize()), torch.cuda.synchronize(), print('ok', tag, flush=True))
eng = sharded.HipEngine('cuda:0')
st = eng.shard_open(ing, in_set); sync('open')
first = eng.shard_first(st).clone(); sync('first')
dist.all_reduce(first, op=dist.ReduceOp.MIN); sync('allreduce')
fidx, nl = eng.rank_first(first); sync('rank %d' % nl)
shape = int(in_set.sum()); bounds = sharded.row_ranges(shape, 1)
w0, w1, counts = eng.shard_emit(st, fidx, bounds); sync('emit %s' % counts)
r0, _ = sharded._all_to_all_var(w0, counts, dist, torch)
r1, _ = sharded._all_to_all_var(w1, counts, dist, torch); sync('a2a')
print('a2a exact:', bool(torch.equal(r0, w0)), bool(torch.equal(r1, w1)), flush=True)
blk = eng.rows_from_entries(r0, r1, 0, shape, shape); sync('rows nnz %d' % blk.nnz)
m, fi, nl2 = ing.link_matrix(in_set); sync('ref')
print('equal:', all(np.array_equal(x, y) for x, y in zip(blk.to_arrays(), m.to_arrays())), flush=True)
eng.shard_close(st)
_lib.normalize_l1(blk); sync('norm')
t0 = time.perf_counter()
res, it, cv, stats = sharded.mcl_sharded(None, 2, 2.0, 200, 1e-4, dist, 'cuda:0', local_block=blk, n=shape); sync('mcl %d it %.3f s' % (it, time.perf_counter() - t0))
dist.destroy_process_group()
