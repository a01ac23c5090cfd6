"""Where the time of a1 (`.pairs` text -> id arrays + alignments.bed) goes on this box: the same file through cluster.PairsText with the BED written, fetched
but not written, and off; the file in a RAM disk and in the temporary directory; plus the raw host rates the stage depends on (reading the file, writing
as many bytes as the BED has).   python tools/text_probe.py [pairs] [contigs]"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def main():
    import torch
    from haphic_amd import _lib, cluster, synth
    import c3_run
    pairs = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000_000
    contigs = int(sys.argv[2]) if len(sys.argv) > 2 else 100_000
    _lib.check(_lib.load().hhx_set_device(0))
    gen = synth.make_genome(24, contigs // 24 * 30_000, 30_000, seed=12345)
    names = list(gen.names)
    arrays = list(synth.sample_pairs(gen, pairs, seed=12346, device='cuda:0'))
    out = {'pairs': pairs}
    for where in ('/dev/shm', '/tmp'):
        d = os.path.join(where, 'hhx_text_probe')
        os.makedirs(d, exist_ok=True)
        path = os.path.join(d, 'hic.pairs')
        t = time.perf_counter()
        size, _ = c3_run.write_pairs_file(path, gen, *arrays)
        res = {'file_bytes': size, 'write_pairs_file_s': time.perf_counter() - t}
        # raw: read the whole file once through the page cache (read() into a reused buffer), 1 and 8 threads
        from concurrent.futures import ThreadPoolExecutor

        def read_range(lo, hi):
            buf = bytearray(64 << 20)
            fd = os.open(path, os.O_RDONLY)
            at = lo
            while at < hi:
                k = os.preadv(fd, [memoryview(buf)[:min(len(buf), hi - at)]], at)
                if k <= 0:
                    break
                at += k
            os.close(fd)
        for thr in (1, 8):
            t = time.perf_counter()
            with ThreadPoolExecutor(thr) as pool:
                list(pool.map(lambda k: read_range(size * k // thr, size * (k + 1) // thr), range(thr)))
            res['read_%d_threads_GBs' % thr] = size / (time.perf_counter() - t) / 1e9
        table = cluster.FragTable.for_contigs(gen.lexical_rank(), gen.length, np.ones(gen.n, np.uint8), names=names)
        os.chdir(d)
        for mode in ('bed_written', 'bed_fetched_not_written', 'no_bed'):
            aln = cluster.pairs_generator_inter_ctgs(path, 'pairs')
            if mode == 'no_bed':
                aln.bed_path = None
            if mode == 'bed_fetched_not_written':
                cluster._pwrite_all_saved = cluster._pwrite_all
                cluster._pwrite_all = lambda fd, view, offset: None
            t = time.perf_counter()
            ing = _lib.Ingest(table, 500_000, bins=False, skip_intra=True)
            ing.keep_pairs()
            for parser, k in aln.batches(names):
                ing.push_device(k, *parser.device_arrays()[:4])
            ing.finalize()
            _lib.check(_lib.load().hhx_synchronize())
            dt = time.perf_counter() - t
            ing.destroy()
            if mode == 'bed_fetched_not_written':
                cluster._pwrite_all = cluster._pwrite_all_saved
            res[mode] = {'seconds': dt, 'pairs_per_s': pairs / dt, 'text_GBs': size / dt / 1e9, 'stats': aln.stats}
            if os.path.exists('alignments.bed'):
                res[mode]['bed_file_bytes'] = os.path.getsize('alignments.bed')
                os.remove('alignments.bed')
        os.chdir('/')
        os.remove(path)
        out[where] = res
    print(json.dumps(out))


if __name__ == '__main__':
    main()
