"""How the oracle's row-parallel product walk scales over the host threads of the GPU box (it decides how much of the 1200 s of the
driver's gpu test run the whole-matrix oracle checks may take): rows of iteration 0 of the C3 link matrix at 1 ... 256 threads."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import torch
    from haphic_amd import _lib, synth
    from haphic_amd.cluster import FragTable
    from oracle import oracle as orc
    info = {'cpu_count': os.cpu_count(), 'affinity': len(os.sched_getaffinity(0))}
    for f in ('/sys/fs/cgroup/cpu.max', '/sys/fs/cgroup/cpu/cpu.cfs_quota_us', '/sys/fs/cgroup/cpu/cpu.cfs_period_us'):
        try:
            info[f] = open(f).read().strip()
        except OSError:
            pass
    print(json.dumps(info), flush=True)
    gen = synth.make_genome(24, (100_000 // 24) * 30_000, 30_000, seed=12345)
    n = gen.n
    table = FragTable.for_contigs(gen.lexical_rank(), gen.length, np.ones(n, np.uint8))
    id1, p1, id2, p2 = synth.sample_pairs(gen, 500_000_000, seed=12345, device='cuda:0')
    torch.cuda.synchronize()
    ing = _lib.Ingest(table, 500_000, bins=False, skip_intra=True)
    ing.push_device(id1.numel(), id1.data_ptr(), p1.data_ptr(), id2.data_ptr(), p2.data_ptr())
    ing.finalize()
    m, _f, _n = ing.link_matrix(np.ones(n, np.uint8))
    ing.destroy()
    L = m.to_arrays()
    lens = np.diff(L[0]).astype(np.int64)
    for threads in (1, 8, 16, 32, 64, 128, 256):
        rows = np.arange(0, n, max(1, n // (64 * min(threads, 32))), dtype=np.int32)[:64 * min(threads, 32)]
        prod = int(sum(int(lens[L[1][L[0][i]:L[0][i + 1]]].sum()) for i in rows))
        orc.set_threads(threads)
        t0 = time.perf_counter()
        orc.links_iteration0(L, rows, 2.0, 1e-4)
        dt = time.perf_counter() - t0
        print(json.dumps({'threads': threads, 'rows': int(len(rows)), 'products': prod, 'seconds': round(dt, 2), 'products_per_s': prod / dt,
                          'per_thread': prod / dt / threads}), flush=True)


if __name__ == '__main__':
    main()
