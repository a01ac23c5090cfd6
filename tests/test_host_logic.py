"""host-side logic that needs no GPU: chunking of a .pairs file into whole-line pieces"""
import gzip

import numpy as np
import pytest


@pytest.mark.parametrize('fmt', ['pairs', 'bgzipped_pairs'])
@pytest.mark.parametrize('chunk', [7, 64, 1000, 1 << 20])
def test_pairs_text_chunks_are_whole_lines(tmp_path, fmt, chunk):
    from haphic_amd import cluster
    rng = np.random.default_rng(3)
    lines = [('r%d\tctg%d\t%d\tctg%d\t%d' % (k, rng.integers(50), rng.integers(1, 10 ** 6), rng.integers(50), rng.integers(1, 10 ** 6))).encode()
             for k in range(400)]
    lines[100] = b'x' * 300                                   # a line longer than the small chunk sizes
    raw = b'\n'.join(lines)                                    # no newline at the end of the file
    path = tmp_path / ('in.pairs' + ('.gz' if fmt != 'pairs' else ''))
    if fmt == 'pairs':
        path.write_bytes(raw)
    else:
        with gzip.open(path, 'wb') as f:
            f.write(raw)
    aln = cluster.PairsText(str(path), fmt, inter_only=False, chunk_bytes=chunk)
    pieces = [bytes(c) for c in aln._chunks()]
    assert b''.join(pieces) == raw
    assert all(p.endswith(b'\n') for p in pieces[:-1]) and all(pieces)


def test_pairs_text_empty_file(tmp_path):
    from haphic_amd import cluster
    p = tmp_path / 'empty.pairs'
    p.write_bytes(b'')
    assert list(cluster.PairsText(str(p), 'pairs', inter_only=True)._chunks()) == []


def test_inflation_values_match_numpy_decimal_arange():
    """run_mcl_clustering :2138-2155 iterates numpy.arange over Decimal objects"""
    from decimal import Decimal
    from haphic_amd import cluster
    for lo, hi, st in ((1.0, 3.0, 0.1), (1.1, 1.7, 0.2), (2.0, 2.0, 0.5), (1.2, 2.4, 0.4)):
        want = list(np.arange(Decimal(str(lo)), Decimal(str(hi)) + Decimal(str(st)), Decimal(str(st))))
        assert cluster._inflation_values(lo, hi, st) == want


def test_bench_self_launch_command(monkeypatch):
    """`python bench.py --gpus N` without a rendezvous in the environment re-executes itself under torch.distributed.run with
    the driver's own flags; under torchrun (WORLD_SIZE set) or at N = 1 it does nothing"""
    import argparse
    import os
    import sys
    import bench
    calls = []
    monkeypatch.setattr(os, 'execv', lambda exe, argv: calls.append((exe, argv)))
    monkeypatch.setattr(sys, 'argv', ['bench.py', '--gpus', '4', '--steps', '3', '--warmup', '1'])
    monkeypatch.delenv('WORLD_SIZE', raising=False)
    bench.self_launch(argparse.Namespace(gpus=4, master_port=0))
    (exe, argv), = calls
    assert exe == sys.executable and argv[:3] == [sys.executable, '-m', 'torch.distributed.run']
    assert '--nnodes=1' in argv and argv[argv.index('--nproc-per-node') + 1] == '4' and argv[argv.index('--master-addr') + 1] == '127.0.0.1'
    assert int(argv[argv.index('--master-port') + 1]) > 0
    assert argv[-7] == os.path.abspath(bench.__file__) and argv[-6:] == ['--gpus', '4', '--steps', '3', '--warmup', '1']
    monkeypatch.setenv('WORLD_SIZE', '4')
    bench.self_launch(argparse.Namespace(gpus=4, master_port=0))
    monkeypatch.delenv('WORLD_SIZE')
    bench.self_launch(argparse.Namespace(gpus=1, master_port=0))
    assert len(calls) == 1


# ---- the two arithmetic claims of k_dense_epilogue_sw (csrc/hhx_expand.hip), restated in numpy --------------------------------
def test_dense_epilogue_slot_swizzle_is_bank_conflict_free():
    """A thread owns `per` consecutive 4-byte slots; slot t is stored at t ^ ((t >> 5) & gm), gm = (power of two in per, capped
    at 32) - 1.  ds_read_b32 serves a wave in two groups of 32 lanes over 32 banks: for every per <= DE_PER, every k and every
    group, the 32 lanes must fall on 32 different banks; and the fill (lane t of a group writes slot base + t) must too."""
    for per in range(1, 21):
        gm = min(32, per & -per) - 1
        tid = np.arange(1024)
        for k in range(per):
            t = tid * per + k
            bank = (t ^ ((t >> 5) & gm)) & 31
            groups = bank.reshape(-1, 32)
            assert all(len(set(g.tolist())) == 32 for g in groups), (per, k)
        for i in range(per):                                          # the fill: t = tid + i * 1024
            t = tid + i * 1024
            phys = t ^ ((t >> 5) & gm)
            assert np.array_equal(phys, (tid ^ ((tid >> 5) & gm)) + i * 1024)          # one swizzled base, constant offsets
            assert all(len(set((g & 31).tolist())) == 32 for g in phys.reshape(-1, 32)), (per, i)
        # and the map is a permutation of the slots of a window (rounded up to a 32-slot row)
        n = per * 1024
        t = np.arange(n)
        assert np.array_equal(np.sort(t ^ ((t >> 5) & gm)), t)


def test_quotient_by_reciprocal_matches_true_quotient_outside_the_midpoint_window():
    """float32(a / d) == float32(a * (1 / d)) whenever the double product a * RN(1 / d) is more than 8 ulps away from the midpoint
    of two float32 values and not below 2^-120 (quot_f32 takes the true quotient otherwise).  Random operands shaped like the
    kernel's: a = a float32 (y of the integer arithmetic, or p = x^2), d = an integer row sum or a running sum of float32s."""
    rng = np.random.default_rng(20260926)
    checked = 0
    for scale in (1.0, 1e-6, 1e-12, 1e6):
        a = (rng.random(2_000_000, dtype=np.float32) * np.float32(scale)).astype(np.float32).astype(np.float64)
        for d in (rng.integers(1, 1 << 34, 2_000_000).astype(np.float64), rng.random(2_000_000) * 1e3 + 1e-3):
            q = a * (1.0 / d)
            lo = q.view(np.uint64) & np.uint64(0x1fffffff)
            exact = ((lo - np.uint64(0x0ffffff8)) & np.uint64(0xffffffff)) > np.uint64(16)
            exact &= q >= 2.0 ** -120
            true = (a / d).astype(np.float32)
            fast = q.astype(np.float32)
            assert np.array_equal(true[exact], fast[exact])
            checked += int(exact.sum())
    assert checked > 15_000_000
    # and the window does catch what it is for: products placed exactly on a midpoint are sent to the true quotient
    mid = (np.float32(1.5).astype(np.float64) + np.float32(1.5000001).astype(np.float64)) / 2
    lo = np.array([mid]).view(np.uint64) & np.uint64(0x1fffffff)
    assert int(lo[0]) == 0x10000000


def test_early_pruning_bound_of_the_dense_epilogue():
    """k_dense_epilogue_sw decides "pruned" before the running row sum S of a window is known: S is at least the sum S' of the
    windows before, and p < float32(thr * S' * (1 - 2^-18)) must imply float32(p / S) < thr for every S >= S' (the exact test
    visits only the other slots)."""
    rng = np.random.default_rng(7)
    thr = np.float32(1e-4)
    n = 4_000_000
    s_prev = rng.random(n) * 10.0 ** rng.integers(-6, 3, n) + 1e-9
    s = s_prev * (1.0 + rng.random(n) * rng.integers(0, 2, n))                      # S == S' half of the time: the tight case
    lo = (np.float64(thr) * s_prev * (1.0 - 2.0 ** -18)).astype(np.float32)
    # p just below the bound (the hardest case), and anywhere below it
    for p in (np.nextafter(lo, np.float32(0)), (lo * rng.random(n, dtype=np.float32)).astype(np.float32)):
        below = p < lo
        q = (p.astype(np.float64) / s).astype(np.float32)
        assert not (q[below] >= thr).any()


def test_triangle_storage_addresses():
    """hhx_expand.hip: the upper block triangle of the dense block — block row I starts at cap (I ldn - cap I (I - 1) / 2) floats and
    keeps its windows J >= I with a pitch of ldn - I cap.  Every (row, window J >= block of the row, slot) must get its own float, the
    whole must be exactly cap^2 n_win (n_win + 1) / 2 floats, and every row segment must start on a 128-byte line (cap % 64 == 0)."""
    for cap, n_win in ((64, 1), (64, 3), (128, 5), (192, 10)):
        ldn = n_win * cap
        off = lambda I: cap * (I * ldn - cap * I * (I - 1) // 2)
        seen = np.zeros(cap * cap * n_win * (n_win + 1) // 2, np.int8)
        for I in range(n_win):
            pitch = ldn - I * cap
            for r in range(cap):
                for J in range(I, n_win):
                    a = off(I) + r * pitch + (J - I) * cap
                    assert a % 32 == 0
                    assert not seen[a:a + cap].any()
                    seen[a:a + cap] = 1
        assert seen.all()


def test_id_arrays_alignments():
    """cluster.IdArrays: alignments handed over as id arrays iterate to the reference generators' tuples (:1539-1593)"""
    from haphic_amd import cluster
    names = ['ctgB', 'ctgA', 'ctgC']
    a = cluster.IdArrays(names, [0, 2, -1], [5, 6, 7], [1, 1, 2], [50, 60, 70])
    assert len(a) == 3
    assert list(a) == [('ctgB', 'ctgA', 5, 50), ('ctgC', 'ctgA', 6, 60), (None, 'ctgC', 7, 70)]
    with pytest.raises(ValueError):
        cluster.IdArrays(names, [0], [1, 2], [0], [1])


def test_oracle_threads_respect_the_cpu_quota():
    """the oracle's default thread count is the CPUs this process may use (affinity capped by the cgroup quota), not every
    hardware thread of the host: on the GPU box (256 threads, quota 16) 128 threads walked products at half the rate of 16"""
    import os
    from oracle import oracle as orc
    e = orc.effective_cpus()
    assert 1 <= e <= (os.cpu_count() or 1)
    orc.set_threads(0)
    assert orc.get_threads() == e
    orc.set_threads(1)
    assert orc.get_threads() == 1
    orc.set_threads(0)


def test_cluster_files_array_path_equals_the_loops():
    """run_mcl_clustering's per-contig half (:2073-2095, :2172-2218) on arrays (cluster._clusters_from_arrays / _groups_from_arrays) against the
    reference's loops restated here: validity of the partition, the set of tuples, group order (total length descending, stable over the set's
    iteration order), contig order within a group (length descending, stable over ascending matrix index), the bytes of every group file"""
    from collections import defaultdict
    from haphic_amd import cluster
    rng = np.random.default_rng(7)

    def loop_clusters(att, att_ptr, members, shape):
        clusters = set()
        for a in range(len(att)):
            clusters.add(tuple(members[att_ptr[a]:att_ptr[a + 1]].tolist()))
        nodes = set()
        for c in clusters:
            for node in c:
                if node in nodes:
                    return None
                nodes.add(node)
        return list(clusters) if len(nodes) == shape else None

    for trial in range(400):
        shape = int(rng.integers(0, 40))
        k = int(rng.integers(0, 8))
        if rng.random() < 0.7 and shape:
            parts = np.split(rng.permutation(shape), np.sort(rng.integers(0, shape + 1, max(k - 1, 0)))) if k else []
            if rng.random() < 0.15 and len(parts) > 1:
                parts.append(parts[0])                              # two attractors, one cluster
        else:
            parts = [rng.integers(0, max(shape, 1), int(rng.integers(0, 6))) for _ in range(k)]
        members = np.concatenate(parts).astype(np.int32) if parts else np.zeros(0, np.int32)
        ptr = np.concatenate([[0], np.cumsum([len(p) for p in parts])]).astype(np.int32)
        got = cluster._clusters_from_arrays(np.arange(len(parts)), ptr, members, shape)
        want = loop_clusters(np.arange(len(parts)), ptr, members, shape)
        assert (got is None) == (want is None) and (got is None or got == want)
        if not got:
            continue
        names = np.empty(shape, object)
        names[:] = ['ctg%d' % i for i in range(shape)]
        lens = rng.choice([5000, 7000, 7000, 9000, 12000], shape)            # many ties, in contig lengths and in group totals
        lines = np.array(['%s\t%d\t%d\n' % (names[i], lens[i] // 256 + 1, lens[i]) for i in range(shape)], object)
        result, bodies = cluster._groups_from_arrays(got, names, lens, lines)
        groups = defaultdict(lambda: [[], 0])
        for kk, indexes in enumerate(got):
            for i in indexes:
                groups[kk][0].append(names[i])
                groups[kk][1] += int(lens[i])
        ref = sorted(tuple(groups.values()), key=lambda g: g[1], reverse=True)
        length = dict(zip(names.tolist(), lens.tolist()))
        for ctgs, _t in ref:
            ctgs.sort(key=lambda c: length[c], reverse=True)
        assert result == [list(g) for g in ref] and all(type(t) is int for _c, t in result)
        assert bodies == [''.join('%s\t%d\t%d\n' % (c, length[c] // 256 + 1, length[c]) for c in ctgs) for ctgs, _t in ref]


def test_dense_block_is_taken_ahead_only_when_the_sweep_will_use_it(monkeypatch):
    """cluster._prewarm_dense_block: the sweep's n x n float32 block is taken from the driver while the alignment file is read — only for orders whose block the library
    keeps whole (at most a quarter of the device), on a device that is otherwise empty, in a single-process run; the size is that of the n x roundup32(n) block"""
    import threading
    import types
    from haphic_amd import cluster
    GiB = 1 << 30
    asked = []
    state = {'free': 280 * GiB, 'total': 288 * GiB, 'cached': 0}
    fake = types.SimpleNamespace(mem_info=lambda: (state['free'], state['total']), pool_cached_bytes=lambda: state['cached'],
                                 pool_prewarm=lambda sizes: asked.append(list(sizes)))
    monkeypatch.setattr(cluster, '_lib', fake)
    monkeypatch.setattr(cluster, '_DENSE_WARM', None)

    def run(n):
        monkeypatch.setattr(cluster, '_DENSE_WARM', None)
        del asked[:]
        cluster._prewarm_dense_block(n)
        t = cluster._DENSE_WARM
        if t is not None:
            assert isinstance(t, threading.Thread)
            t.join()
        return list(asked)

    assert run(99878) == [[4 * 99878 * 99904]]                       # BASELINE configs[2]: 39.9 GB
    assert run(19999) == []                                          # small: a fresh block of that size costs nothing
    assert run(140000) == []                                         # 78 GB > a quarter of 288 GB: the library stores the upper block triangle, row block by row block
    state['free'] = 150 * GiB
    assert run(99878) == []                                          # somebody else is using the device
    state['cached'] = 100 * GiB
    assert run(99878) == [[4 * 99878 * 99904]]                       # ... unless it is this library's own cache
    # several ranks: the sweep is shared out, no rank keeps the whole block
    dist = types.SimpleNamespace(is_available=lambda: True, is_initialized=lambda: True, get_world_size=lambda: 8)
    monkeypatch.setitem(__import__('sys').modules, 'torch', types.SimpleNamespace(distributed=dist))
    assert run(99878) == []
