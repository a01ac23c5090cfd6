#!/usr/bin/env python3
"""Rate of the `haphic plot` binning kernel (hhx_contacts.hip) on synthetic read pairs resident in HBM.
usage: contact_probe.py [n_pairs] [bin_size]      -> one JSON line"""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from haphic_amd import _lib  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 200_000_000
    bin_size = int(sys.argv[2]) if len(sys.argv) > 2 else 500_000
    dev = torch.device('cuda:0')
    rng = np.random.default_rng(0)
    n_ctg = 5000
    lens = rng.integers(50_000, 1_200_000, n_ctg).astype(np.int64)          # ~3 Gb assembly, every contig one '+' piece of a scaffold
    per = n_ctg // 24
    in_set, aln_ptr, list_ptr, lo, hi, cell = np.ones(n_ctg, np.uint8), [0], [0], [], [], []
    total_bins, at = 0, 1
    for c in range(n_ctg):
        if c % per == 0 and c // per < 24:
            total_bins += (at // bin_size + 1) if c else 0
            at = 1
        g0 = total_bins
        # pieces of scaffold bins this contig covers
        first_bin, last_bin = (at - 1) // bin_size, (at + lens[c] - 2) // bin_size
        segs = []
        for gb in range(first_bin, last_bin + 1):
            s_lo = max(gb * bin_size + 1, at) - at + 1
            s_hi = min((gb + 1) * bin_size, at + lens[c] - 1) - at + 1
            segs.append((s_lo, s_hi, g0 + gb))
        for ab in range((lens[c] - 1) // bin_size + 1):
            for s_lo, s_hi, b in segs:
                if (s_lo - 1) // bin_size <= ab <= (s_hi - 1) // bin_size:
                    lo.append(s_lo); hi.append(s_hi); cell.append(b)
            list_ptr.append(len(lo))
        aln_ptr.append(len(list_ptr) - 1)
        at += lens[c] + 100
    total_bins += at // bin_size + 1
    cm = _lib.ContactMap(in_set, aln_ptr, list_ptr, lo, hi, cell, bin_size, total_bins)
    g = torch.Generator(device=dev).manual_seed(1)
    tl = torch.from_numpy(lens).to(dev)
    id1 = torch.randint(0, n_ctg, (n,), device=dev, generator=g, dtype=torch.int32)
    same = torch.rand(n, device=dev, generator=g) < 0.7                      # Hi-C: most pairs are intra-contig and close
    id2 = torch.where(same, id1, torch.randint(0, n_ctg, (n,), device=dev, generator=g, dtype=torch.int32))
    pos1 = (torch.rand(n, device=dev, generator=g, dtype=torch.float64) * tl[id1.long()]).to(torch.int32) + 1
    far = (torch.rand(n, device=dev, generator=g, dtype=torch.float64) * tl[id2.long()]).to(torch.int32) + 1
    near = torch.minimum(torch.clamp(pos1 + torch.randint(-20000, 20000, (n,), device=dev, generator=g, dtype=torch.int32), min=1), tl[id1.long()].to(torch.int32))
    pos2 = torch.where(same, near, far)
    torch.cuda.synchronize()
    ptrs = [t.data_ptr() for t in (id1, pos1, id2, pos2)]
    assert cm.push_device(n, *ptrs) == -1                                    # warm-up (counts twice; rates only)
    reps = 5
    t0 = time.perf_counter()
    for _ in range(reps):
        cm.push_device(n, *ptrs)
    dt = (time.perf_counter() - t0) / reps
    total = int(cm.fetch().sum())
    print(json.dumps({'probe': 'contact_map', 'pairs': n, 'bins': int(total_bins), 'matrix_MB': int(total_bins) ** 2 * 8 / 1e6, 'ms': dt * 1e3,
                      'pairs_per_s': n / dt, 'alg_GBs': 16 * n / dt / 1e9, 'counted_per_push': total // (reps + 1)}))
    cm.destroy()


if __name__ == '__main__':
    main()
