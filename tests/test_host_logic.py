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
