"""Knife-edge audit (VERDICT r01 #1d): how often does the exact fixed-point accumulation of the HIP kernels (oracle
mode 1) decide `>= pruning` differently from the reference's float32 accumulation (oracle mode 0 = scipy's
sequential float32 sums, the stand-in for sparse_dot_mkl)?  Runs both mcl() trajectories on the BASELINE configs[1]
link matrix (10k contigs / 50 M pairs) iteration by iteration on the host cores and counts, per iteration, the
entries kept by one and pruned by the other, the largest relative value difference on the common entries, and
whether the final clusters agree.  CPU only (the oracle); writes one JSON object."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    from haphic_amd import synth
    from oracle import oracle as orc
    contigs, pairs = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (10000, 50_000_000)
    gen = synth.make_genome(16, max(1, contigs // 16) * 50_000, 50_000, seed=12345)
    n = gen.n
    lex = gen.lexical_rank()
    t = orc.FragTable(lex, gen.length, np.arange(n, dtype=np.int32), np.zeros(n, np.uint8), 0, lex, gen.length, np.ones(n, np.uint8))
    dev = 'cuda:0' if __import__('torch').cuda.is_available() else 'cpu'
    h = [a.cpu().numpy() for a in synth.sample_pairs(gen, pairs, seed=12345, device=dev)]
    keep = h[0] != h[2]
    t0 = time.perf_counter()
    r = orc.ingest(t, h[0][keep], h[1][keep].astype(np.int64), h[2][keep], h[3][keep].astype(np.int64), 500_000)
    in_set = np.ones(n, np.uint8)
    linked = np.zeros(n, bool)
    linked[r['flank_i']] = True
    linked[r['flank_j']] = True
    p, j, x, fidx, nl = orc.dict_to_matrix(r['flank_i'], r['flank_j'], r['flank_cnt'].astype(np.float64), n, in_set, int(n - linked.sum()))
    xn = orc.normalize_l1(p, x)
    out = {'contigs': int(n), 'pairs': int(pairs), 'link_matrix_nnz': int(len(j)), 'threads': orc.get_threads(), 'iterations': []}
    cur = {0: (p, j, xn), 1: (p, j, xn)}
    last = {0: None, 1: None}
    done = {0: None, 1: None}
    for it in range(200):
        nxt = {}
        for mode in (0, 1):
            if done[mode] is not None:
                nxt[mode] = cur[mode]
                continue
            A = cur[mode]
            c = orc.spgemm(A, A, mode=mode, fx_shift=52)            # iteration 0: the pre-expansion :2147; later: :2030-2033
            xs = orc.normalize_l1(c[0], orc.power(c[2], 2.0))
            P = orc.prune((c[0], c[1], xs), 1e-4)
            if it > 1 and orc.convergence_stat(P, last[mode]) <= np.float32(1e-8):
                done[mode] = it + 1
            last[mode] = P
            nxt[mode] = P
        a, b = nxt[0], nxt[1]
        # entries of one pattern missing from the other
        ka = np.repeat(np.arange(n, dtype=np.int64), np.diff(a[0])) * n + a[1]
        kb = np.repeat(np.arange(n, dtype=np.int64), np.diff(b[0])) * n + b[1]
        only_a = np.setdiff1d(ka, kb, assume_unique=True)
        only_b = np.setdiff1d(kb, ka, assume_unique=True)
        common_a = np.isin(ka, kb, assume_unique=True)
        common_b = np.isin(kb, ka, assume_unique=True)
        va, vb = a[2][common_a].astype(np.float64), b[2][common_b].astype(np.float64)
        rel = float(np.max(np.abs(va - vb) / np.maximum(np.abs(vb), 1e-300))) if len(va) else 0.0
        out['iterations'].append({'it': it, 'nnz_f32': int(len(ka)), 'nnz_exact': int(len(kb)), 'kept_only_by_f32': int(len(only_a)),
                                  'kept_only_by_exact': int(len(only_b)), 'max_rel_diff_common': rel})
        cur = nxt
        if done[0] is not None and done[1] is not None:
            break
    ca = {tuple(m[q[a]:q[a + 1]].tolist()) for at, q, m in [orc.interpret(cur[0])] for a in range(len(at))}
    cb = {tuple(m[q[a]:q[a + 1]].tolist()) for at, q, m in [orc.interpret(cur[1])] for a in range(len(at))}
    out.update(iterations_f32=done[0], iterations_exact=done[1], clusters_f32=len(ca), clusters_exact=len(cb), clusters_identical=ca == cb,
               total_flips=int(sum(i['kept_only_by_f32'] + i['kept_only_by_exact'] for i in out['iterations'])), seconds=time.perf_counter() - t0)
    print(json.dumps(out))


if __name__ == '__main__':
    main()
