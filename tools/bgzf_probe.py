"""a1 from a bgzipped .pairs file (`bgzipped_pairs`): the library's BGZF reader (hhx_text_reader_open_bgzf: blocks inflated by 8 threads into pinned memory) against Python's gzip
module (what the mirrors used until round 6, and still use for plain gzip streams).  python tools/bgzf_probe.py [pairs]"""
import json
import os
import struct
import sys
import time
import zlib
from concurrent.futures import ProcessPoolExecutor

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def bgzf_block(payload):
    c = zlib.compressobj(1, zlib.DEFLATED, -15)
    cdata = c.compress(payload) + c.flush()
    bsize = 12 + 6 + len(cdata) + 8
    return struct.pack('<4BI2BH', 0x1f, 0x8b, 8, 4, 0, 0, 0xff, 6) + b'BC' + struct.pack('<HH', 2, bsize - 1) + cdata + struct.pack('<II', zlib.crc32(payload) & 0xffffffff, len(payload))


def compress_range(args):
    path, lo, hi = args
    with open(path, 'rb') as f:
        f.seek(lo)
        data = f.read(hi - lo)
    return b''.join(bgzf_block(data[a:a + 0xff00]) for a in range(0, len(data), 0xff00))


def main():
    from haphic_amd import _lib, cluster, synth
    import c3_run
    pairs = int(sys.argv[1]) if len(sys.argv) > 1 else 20_000_000
    _lib.check(_lib.load().hhx_set_device(0))
    gen = synth.make_genome(24, 100_000 // 24 * 30_000, 30_000, seed=12345)
    names = list(gen.names)
    arrays = list(synth.sample_pairs(gen, pairs, seed=12346, device='cuda:0'))
    d = '/dev/shm/hhx_bgzf_probe'
    os.makedirs(d, exist_ok=True)
    plain = os.path.join(d, 'hic.pairs')
    size, _ = c3_run.write_pairs_file(plain, gen, *arrays)
    t = time.perf_counter()
    step = 64 * 0xff00
    with ProcessPoolExecutor(12) as pool, open(plain + '.gz', 'wb') as out:
        for piece in pool.map(compress_range, [(plain, lo, min(size, lo + step)) for lo in range(0, size, step)], chunksize=4):
            out.write(piece)
        out.write(bytes.fromhex('1f8b08040000000000ff0600424302001b0003000000000000000000'))
    res = {'pairs': pairs, 'text_bytes': size, 'bgzf_bytes': os.path.getsize(plain + '.gz'), 'bgzip_s_untimed': time.perf_counter() - t}
    table = cluster.FragTable.for_contigs(gen.lexical_rank(), gen.length, np.ones(gen.n, np.uint8), names=names)
    os.chdir(d)
    for mode, sync in (('library_bgzf_reader', '0'), ('python_gzip', '1'), ('plain_text_file', '0')):
        os.environ['HAPHIC_SYNC_FILES'] = sync
        path, fmt = (plain, 'pairs') if mode == 'plain_text_file' else (plain + '.gz', 'bgzipped_pairs')
        aln = cluster.pairs_generator_inter_ctgs(path, fmt)
        t = time.perf_counter()
        ing = _lib.Ingest(table, 500_000, bins=False, skip_intra=True)
        ing.keep_pairs()
        for parser, k in aln.batches(names):
            ing.push_device(k, *parser.device_arrays()[:4])
        ing.finalize()
        _lib.check(_lib.load().hhx_synchronize())
        dt = time.perf_counter() - t
        n_full = ing.n_full
        ing.destroy()
        _lib.files_join()
        res[mode] = {'seconds': dt, 'pairs_per_s': pairs / dt, 'text_GBs': size / dt / 1e9, 'full_keys': n_full, 'stats': aln.stats, 'bed_bytes': os.path.getsize('alignments.bed')}
        os.remove('alignments.bed')
    os.environ.pop('HAPHIC_SYNC_FILES', None)
    assert res['library_bgzf_reader']['full_keys'] == res['python_gzip']['full_keys'] == res['plain_text_file']['full_keys']
    assert res['library_bgzf_reader']['bed_bytes'] == res['python_gzip']['bed_bytes'] == res['plain_text_file']['bed_bytes']
    for f in (plain, plain + '.gz'):
        os.remove(f)
    print(json.dumps(res))


if __name__ == '__main__':
    main()
