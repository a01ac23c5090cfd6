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
