"""f4: the BAM front end (BGZF inflate on host threads + record decode on the device, haphic_amd/csrc/hhx_bam.hip)
against the pure-Python restatement of bam_generator :1586-1593 (oracle.parse_bam) on BAM files written byte by byte
from the SAM specification (tests/bam_fixture.py).  No htslib here: PARITY UNPINNED against pysam itself."""
import gzip
import logging

import numpy as np
import pytest

from oracle import oracle as orc
from tests import bam_fixture as bf


def test_oracle_bam_restatement_and_fixture():
    refs, recs, data = bf.random_case()
    assert data.endswith(bf.EOF_BLOCK) and gzip.decompress(data)[:4] == b'BAM\x01'       # a gzip reader accepts the BGZF members
    text, names, tup = orc.parse_bam(data, 0x40, False)
    read1 = [x for x in recs if x[4] & 0x40]
    want = [(refs[r][0] if r >= 0 else None, refs[m][0] if m >= 0 else None, p, q) for r, p, m, q, f in read1]
    assert text.startswith('@HD\tVN:1.6\tSO:unsorted') and names == [r[0] for r in refs] and tup == want
    assert orc.parse_bam(data, 0x40, True)[2] == [t for t, x in zip(want, read1) if x[0] != x[2]]
    assert len(orc.parse_bam(data, 0, False)[2]) == len(recs)


def test_oracle_bam_rich_fixture():
    """optional fields of every type, multi-operation and > 65535-operation CIGARs (CG tag), records larger than a BGZF
    block, more than 65535 references: the record walk must rely on block_size alone"""
    refs, recs, data = bf.random_case(n_ref=70_000, n_rec=4000, seed=11, block_payload=0xff00, rich=True)
    raw = gzip.decompress(data)
    assert raw[:4] == b'BAM\x01' and b'CGBI' in raw
    text, names, tup = orc.parse_bam(data, 0x40, False)
    read1 = [x for x in recs if x[4] & 0x40]
    assert len(names) == 70_000 and max(x[0] for x in read1) > 65_535
    assert tup == [(refs[r][0] if r >= 0 else None, refs[m][0] if m >= 0 else None, p, q) for r, p, m, q, f in read1]


def test_check_sorting_order_mirror(caplog):
    from haphic_amd import cluster
    with caplog.at_level(logging.INFO, logger='HapHiC_cluster'):
        cluster.check_sorting_order('@HD\tVN:1.6\tSO:queryname\n@SQ\tSN:a\tLN:5\n')
        cluster.check_sorting_order('@SQ\tSN:a\tLN:5\n')
    assert 'The sorting order of the BAM file is queryname' in caplog.text and 'unknown, but the program will continue' in caplog.text
    with pytest.raises(RuntimeError, match='coordinate. It should be unsorted or name-sorted'):
        cluster.check_sorting_order('@HD\tVN:1.6\tSO:coordinate\n')


@pytest.mark.gpu
def test_bam_front_end_against_oracle(tmp_path):
    from haphic_amd import _lib, cluster
    refs, recs, data = bf.random_case(n_ref=60, n_rec=30000, seed=5, block_payload=3000)
    path = tmp_path / 'hic.bam'
    path.write_bytes(data)
    for opts, drop in (([b'filter=flag.read1'], False), ([b'filter=flag.read1 && refid != mrefid'], True)):
        want = orc.parse_bam(data, 0x40, drop)[2]
        for batch in (256 << 20, 1 << 16, 200_000):                       # one batch; one BGZF block per batch; a few blocks
            gen = cluster.bam_generator(str(path), 3, opts)
            gen.batch_bytes = batch
            assert list(gen) == want, (opts, batch)
    # what real BAMs carry beyond the fixed fields: optional fields ('B' arrays too), long CIGARs (CG tag), records larger
    # than a BGZF block and than the batch limit, 70k references
    refs2, recs2, data2 = bf.random_case(n_ref=70_000, n_rec=20000, seed=12, block_payload=0xff00, rich=True)
    rich = tmp_path / 'rich.bam'
    rich.write_bytes(data2)
    for drop, opts in ((False, [b'filter=flag.read1']), (True, [b'filter=flag.read1 && refid != mrefid'])):
        want2 = orc.parse_bam(data2, 0x40, drop)[2]
        for batch in (256 << 20, 1 << 16, 1_000_000):
            gen = cluster.bam_generator(str(rich), 4, opts)
            gen.batch_bytes = batch
            assert list(gen) == want2, ('rich', opts, batch)
    # through the ingest: some BAM references are not in the FASTA (dropped like `ref not in fa_dict`), FASTA order differs
    rng = np.random.default_rng(9)
    keep = [k for k in range(len(refs)) if k % 5 != 3]
    rng.shuffle(keep)
    names = [refs[k][0] for k in keep]
    length = np.array([refs[k][1] for k in keep], np.int64)
    order = sorted(range(len(names)), key=names.__getitem__)
    rank = np.empty(len(names), np.int32)
    rank[order] = np.arange(len(names), dtype=np.int32)
    table = cluster.FragTable.for_contigs(rank, length, np.ones(len(names), np.uint8), names=names)
    cid = {n: i for i, n in enumerate(names)}
    tup = orc.parse_bam(data, 0x40, True)[2]
    h = [np.array([cid.get(t[c], -1) if t[c] is not None else -1 for t in tup], np.int32) if c in (0, 1) else np.array([t[c] for t in tup], np.int64)
         for c in (0, 2, 1, 3)]
    ot = orc.FragTable(rank, length, np.arange(len(names), dtype=np.int32), np.zeros(len(names), np.uint8), 0, rank, length, np.ones(len(names), np.uint8))
    ref = orc.ingest(ot, h[0], h[1], h[2], h[3], 500_000)
    gen = cluster.bam_generator(str(path), 2, [b'filter=flag.read1 && refid != mrefid'])
    gen.batch_bytes = 300_000
    got = cluster.ingest_links(gen, table, 500_000, bins=False)
    for k in ('full_i', 'full_j', 'full_cnt', 'ht_cnt', 'flank_i', 'flank_j', 'flank_cnt', 'frag_links'):
        assert np.array_equal(got[k], ref[k]), k
    # error behaviour: coordinate-sorted input (:1347-1359), a truncated container, not a BAM at all
    bad = tmp_path / 'sorted.bam'
    bad.write_bytes(bf.bam_bytes(refs[:3], recs[:0], header_text='@HD\tVN:1.6\tSO:coordinate\n'))
    with pytest.raises(RuntimeError, match='coordinate'):
        list(cluster.bam_generator(str(bad), 1, [b'filter=flag.read1']))
    cut = tmp_path / 'cut.bam'
    cut.write_bytes(data[:len(data) // 2])
    with pytest.raises(RuntimeError, match='truncated'):
        list(cluster.bam_generator(str(cut), 1, [b'filter=flag.read1']))
    txt = tmp_path / 'not.bam'
    txt.write_bytes(b'this is not a BAM file, not even gzip' * 10)
    with pytest.raises(RuntimeError, match='BGZF|BAM'):
        list(cluster.bam_generator(str(txt), 1, [b'filter=flag.read1']))
    with pytest.raises(NotImplementedError):
        cluster.bam_generator(str(path), 1, [b'filter=mapq >= 30'])
