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


def aux_tags(rng):
    """a random run of optional fields (SAM spec §4.2.4): every value type, 'B' arrays of every element type included"""
    out = b''
    for _ in range(int(rng.integers(0, 7))):
        tag = bytes(rng.choice(list(b'ABCDEFGHIJKLMNOPQRSTUVWXYZ'), 2).tolist())
        kind = int(rng.integers(0, 11))
        if kind == 0:
            out += tag + b'A' + bytes([int(rng.integers(33, 127))])
        elif kind in (1, 2, 3, 4, 5, 6):
            code, fmt, lo, hi = [(b'c', '<b', -128, 128), (b'C', '<B', 0, 256), (b's', '<h', -32768, 32768), (b'S', '<H', 0, 65536),
                                 (b'i', '<i', -2 ** 31, 2 ** 31), (b'I', '<I', 0, 2 ** 32)][kind - 1]
            out += tag + code + struct.pack(fmt, int(rng.integers(lo, hi)))
        elif kind == 7:
            out += tag + b'f' + struct.pack('<f', float(rng.normal()))
        elif kind == 8:
            out += tag + b'Z' + bytes(rng.integers(33, 127, int(rng.integers(0, 40))).tolist()) + b'\x00'
        elif kind == 9:
            out += tag + b'H' + b''.join(b'%02X' % int(v) for v in rng.integers(0, 256, int(rng.integers(0, 12)))) + b'\x00'
        else:
            code, fmt, lo, hi = [(b'c', 'b', -128, 128), (b'C', 'B', 0, 256), (b's', 'h', -32768, 32768), (b'S', 'H', 0, 65536),
                                 (b'i', 'i', -2 ** 31, 2 ** 31), (b'I', 'I', 0, 2 ** 32), (b'f', 'f', 0, 0)][int(rng.integers(0, 7))]
            cnt = int(rng.integers(0, 30))
            vals = [float(v) for v in rng.normal(size=cnt)] if code == b'f' else [int(v) for v in rng.integers(lo, hi, cnt)]
            out += tag + b'B' + code + struct.pack('<i', cnt) + struct.pack('<%d%s' % (cnt, fmt), *vals)
    return out


def record(ref, pos, mref, mpos, flag, name=b'r', seq_len=10, n_cigar=1, aux=b'', real_cigar_ops=0):
    """real_cigar_ops > 65535: the record as htslib writes a long CIGAR (SAM spec §4.2.2) — n_cigar_op = 2, the placeholder
    CIGAR <seq_len>S<ref_len>N, the real operations in a CG:B,I tag"""
    name = name + b'\x00'
    cigar = struct.pack('<I', (seq_len << 4) | 0) * n_cigar
    if real_cigar_ops:
        n_cigar = 2
        cigar = struct.pack('<II', (seq_len << 4) | 4, (seq_len << 4) | 3)
        ops = [(1 << 4) | (k & 1) for k in range(real_cigar_ops)]                 # 1M1I1M1I...
        aux = aux + b'CGBI' + struct.pack('<i', real_cigar_ops) + struct.pack('<%dI' % real_cigar_ops, *ops)
    body = struct.pack('<iiBBHHHiiii', ref, pos, len(name), 30, 4680, n_cigar, flag, seq_len, mref, mpos, 0)
    body += name + cigar + bytes((seq_len + 1) // 2) + bytes([30] * seq_len) + aux
    return struct.pack('<i', len(body)) + body


def bam_bytes(refs, records, header_text='@HD\tVN:1.6\tSO:unsorted\n', block_payload=700, seed=0, rich=False):
    """refs: [(name, length)]; records: [(ref, pos, mref, mpos, flag)].  block_payload: inflated bytes per BGZF block —
    small, so that the header, the reference list and most records straddle block boundaries.
    rich: what real Hi-C BAMs carry beyond the fixed fields — optional fields of every type ('B' arrays included) on every
    record, now and then a read of ~100 kb (a record larger than any BGZF block: it spans several) and a CIGAR of more than
    65535 operations (placeholder CIGAR + CG tag)."""
    rng = np.random.default_rng(seed)
    text = header_text + ''.join('@SQ\tSN:%s\tLN:%d\n' % r for r in refs)
    parts = [b'BAM\x01' + struct.pack('<i', len(text)) + text.encode() + struct.pack('<i', len(refs))]
    for name, length in refs:
        parts.append(struct.pack('<i', len(name) + 1) + name.encode() + b'\x00' + struct.pack('<i', length))
    for k, (ref, pos, mref, mpos, flag) in enumerate(records):
        kw = {}
        seq_len, n_cigar = int(rng.integers(1, 150)), int(rng.integers(1, 4))
        if rich:
            kw['aux'] = aux_tags(rng)
            u = rng.random()
            if u < 0.002:
                seq_len = int(rng.integers(90_000, 140_000))                  # a long read: the record alone is > 64 KiB
            elif u < 0.003:
                seq_len, kw['real_cigar_ops'] = 70_000, int(rng.integers(65_536, 70_000))
            elif u < 0.1:
                n_cigar = int(rng.integers(4, 400))
        parts.append(record(ref, pos, mref, mpos, flag, name=b'read%d' % k, seq_len=seq_len, n_cigar=n_cigar, **kw))
    raw = b''.join(parts)
    out, at = [], 0
    while at < len(raw):
        n = int(rng.integers(max(1, block_payload // 2), block_payload + 1))
        out.append(bgzf_block(raw[at:at + n]))
        at += n
    return b''.join(out) + EOF_BLOCK


def random_case(n_ref=40, n_rec=5000, seed=1, **kw):
    rng = np.random.default_rng(seed)
    refs = [('ctg%03d' % k if k % 7 else 'unplaced_%d' % k, int(rng.integers(20_000, 900_000))) for k in range(n_ref)]      # > 65535 references: ids need 32 bits
    recs = []
    for _ in range(n_rec):
        ref = int(rng.integers(-1, n_ref))                            # -1: unmapped
        mref = ref if rng.random() < 0.3 else int(rng.integers(-1, n_ref))
        flag = int(rng.choice([0x41, 0x81, 0x61, 0x91, 0x1, 0x4d, 0x40 | 0x800]))
        pos = int(rng.integers(0, refs[ref][1])) if ref >= 0 else -1
        mpos = int(rng.integers(0, refs[mref][1])) if mref >= 0 else -1
        recs.append((ref, pos, mref, mpos, flag))
    return refs, recs, bam_bytes(refs, recs, seed=seed, **kw)
