"""A BAM file written byte by byte from the SAM specification (§4.1 BGZF, §4.2 BAM) — the fixture of the f4 tests.
No htslib / pysam here, so this writer plus the specification are what pins the BAM front end."""
import struct
import zlib

import numpy as np

EOF_BLOCK = bytes.fromhex('1f8b08040000000000ff0600424302001b0003000000000000000000')      # the spec's empty BGZF block


def bgzf_block(payload, level=6):
    assert len(payload) <= 0xff00
    c = zlib.compressobj(level, zlib.DEFLATED, -15)
    cdata = c.compress(payload) + c.flush()
    bsize = 12 + 6 + len(cdata) + 8                                  # header + BC subfield + data + CRC32 + ISIZE
    head = struct.pack('<4BI2BH', 0x1f, 0x8b, 8, 4, 0, 0, 0xff, 6) + b'BC' + struct.pack('<HH', 2, bsize - 1)
    return head + cdata + struct.pack('<II', zlib.crc32(payload) & 0xffffffff, len(payload))


def record(ref, pos, mref, mpos, flag, name=b'r', seq_len=10, n_cigar=1):
    name = name + b'\x00'
    body = struct.pack('<iiBBHHHiiii', ref, pos, len(name), 30, 4680, n_cigar, flag, seq_len, mref, mpos, 0)
    body += name + struct.pack('<I', (seq_len << 4) | 0) * n_cigar + bytes((seq_len + 1) // 2) + bytes([30] * seq_len)
    return struct.pack('<i', len(body)) + body


def bam_bytes(refs, records, header_text='@HD\tVN:1.6\tSO:unsorted\n', block_payload=700, seed=0):
    """refs: [(name, length)]; records: [(ref, pos, mref, mpos, flag)].  block_payload: inflated bytes per BGZF block —
    small, so that the header, the reference list and most records straddle block boundaries."""
    rng = np.random.default_rng(seed)
    text = header_text + ''.join('@SQ\tSN:%s\tLN:%d\n' % r for r in refs)
    raw = b'BAM\x01' + struct.pack('<i', len(text)) + text.encode() + struct.pack('<i', len(refs))
    for name, length in refs:
        raw += struct.pack('<i', len(name) + 1) + name.encode() + b'\x00' + struct.pack('<i', length)
    for k, (ref, pos, mref, mpos, flag) in enumerate(records):
        raw += record(ref, pos, mref, mpos, flag, name=b'read%d' % k, seq_len=int(rng.integers(1, 150)), n_cigar=int(rng.integers(1, 4)))
    out, at = b'', 0
    while at < len(raw):
        n = int(rng.integers(max(1, block_payload // 2), block_payload + 1))
        out += bgzf_block(raw[at:at + n])
        at += n
    return out + EOF_BLOCK


def random_case(n_ref=40, n_rec=5000, seed=1, **kw):
    rng = np.random.default_rng(seed)
    refs = [('ctg%03d' % k if k % 7 else 'unplaced_%d' % k, int(rng.integers(20_000, 900_000))) for k in range(n_ref)]
    recs = []
    for _ in range(n_rec):
        ref = int(rng.integers(-1, n_ref))                            # -1: unmapped
        mref = ref if rng.random() < 0.3 else int(rng.integers(-1, n_ref))
        flag = int(rng.choice([0x41, 0x81, 0x61, 0x91, 0x1, 0x4d, 0x40 | 0x800]))
        pos = int(rng.integers(0, refs[ref][1])) if ref >= 0 else -1
        mpos = int(rng.integers(0, refs[mref][1])) if mref >= 0 else -1
        recs.append((ref, pos, mref, mpos, flag))
    return refs, recs, bam_bytes(refs, recs, seed=seed, **kw)
