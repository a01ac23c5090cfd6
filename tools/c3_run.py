"""VERDICT r05 #1 — ONE wall clock over the whole seam sequence of the reference's run() :2787-2956 at C3 (BASELINE.json configs[2]: 100 k contigs /
500 M read pairs), starting from the `.pairs` FILE, through the mirrors of haphic_amd/cluster.py in run()'s own order:

    stat_fragments :2829 -> pairs_generator_inter_ctgs :2866 (the device tokeniser, alignments.bed written on the way as the reference's generator
    does :1549-1557) -> parse_alignments_for_ctgs :2872 -> output_pickle(HT_links.pkl) :2879 -> output_clm :2888 -> filter_fragments :2905 ->
    output_pickle(full_links.pkl) :2929 -> dict_to_matrix :2934  ["Hi-C linking matrix was constructed in ..." :2941] -> run_mcl_clustering
    :2945 (20 inflations 1.1 ... 3.0, every cluster / group file written) -> the file-writer thread joined (patch_reference's run() does that).

The reference module does not exist on the GPU box, so run()'s own glue (argparse, logging set-up), parse_fasta (the assembly is synthetic: fa_dict is
synthesised in the shape parse_fasta returns) and output_statistics :2279 are not part of the figure (tools/c1_run.py says the same for C1).
The `.pairs` text itself is produced beforehand, untimed, from the synthetic read pairs in HBM (hhx_pairs_format) — the same model and seeds bench.py samples.

    python tools/c3_run.py [--contigs 100000] [--pairs 500000000] [--dir /dev/shm] [--keep]          one JSON object on stdout
"""
import argparse
import json
import os
import shutil
import sys
import tempfile
import time
import types

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def cgroup_memory():
    """(limit, current) bytes of the memory cgroup this process lives in, (None, None) if there is none: a RAM disk's pages count against it"""
    out = []
    for name in ('memory.max', 'memory.current'):
        try:
            with open('/sys/fs/cgroup/' + name) as f:
                v = f.read().strip()
            out.append(None if v == 'max' else int(v))
        except (OSError, ValueError):
            out.append(None)
    return tuple(out)


def room_for(need_bytes, where):
    """None if `where` can take need_bytes more (file system AND, for a RAM disk, the memory cgroup); else the reason"""
    try:
        free = shutil.disk_usage(where).free
    except OSError as e:
        return str(e)
    if free < need_bytes + (4 << 30):
        return '%s has %.0f GB free, %.0f GB needed' % (where, free / 1e9, need_bytes / 1e9)
    if where.startswith('/dev/shm'):
        limit, current = cgroup_memory()
        try:
            import psutil
            avail = psutil.virtual_memory().available
        except ImportError:
            avail = None
        if limit is not None and current is not None:
            avail = limit - current if avail is None else min(avail, limit - current)
        if avail is not None and avail < need_bytes + (32 << 30):
            return 'memory: %.0f GB available to this cgroup, %.0f GB of RAM-disk files + 32 GB of head room needed' % (avail / 1e9, need_bytes / 1e9)
    return None


def write_pairs_file(path, gen, id1, p1, id2, p2, slice_pairs=16_000_000, writers=8):
    """the read pairs (torch int32 tensors on the device) as .pairs text: formatted on the device slice by slice, copied through two pinned buffers,
    written by a pool of pwrite() threads.  Returns (bytes, lines)."""
    import torch
    from concurrent.futures import ThreadPoolExecutor, wait
    from haphic_amd import _lib
    names = list(gen.names)
    ps = _lib.PairsParser(names)
    head = b'## pairs format v1.0\n#columns: readID chr1 pos1 chr2 pos2 strand1 strand2\n'
    n = id1.numel()
    fd = os.open(path, os.O_WRONLY | os.O_CREAT | os.O_TRUNC, 0o644)
    at = os.pwrite(fd, head, 0)
    pool = ThreadPoolExecutor(writers)
    piece = 16 << 20
    pending = [[], []]
    host = [None, None]
    dev_text = None

    def pwrite_all(view, offset):
        while len(view):
            k = os.pwrite(fd, view, offset)
            view, offset = view[k:], offset + k
    try:
        for s, lo in enumerate(range(0, n, slice_pairs)):
            hi = min(n, lo + slice_pairs)
            ptrs = [t[lo:hi].data_ptr() for t in (id1, p1, id2, p2)]
            nb = ps.format_pairs(hi - lo, *ptrs, first_read=lo)
            if dev_text is None or dev_text.numel() < nb:
                dev_text = None
                dev_text = torch.empty(int(nb * 1.05) + 4096, dtype=torch.uint8, device=id1.device)
            ps.format_pairs(hi - lo, *ptrs, first_read=lo, text_ptr=dev_text.data_ptr(), capacity=dev_text.numel())
            b = s & 1
            wait(pending[b])
            for f in pending[b]:
                f.result()
            if host[b] is None or host[b].numel() < nb:
                host[b] = torch.empty(dev_text.numel(), dtype=torch.uint8, pin_memory=True)
            host[b][:nb].copy_(dev_text[:nb])
            torch.cuda.synchronize()
            view = memoryview(host[b].numpy())[:nb]
            pending[b] = [pool.submit(pwrite_all, view[a:a + piece], at + a) for a in range(0, nb, piece)]
            at += nb
        for b in (0, 1):
            wait(pending[b])
            for f in pending[b]:
                f.result()
    finally:
        pool.shutdown(wait=True)
        os.close(fd)
        ps.destroy()
    return at, n


def run_sequence(pairs_path, gen, nchrs, workdir, sweep=True, log=None):
    """the seam sequence of run() :2829-2945 in `workdir`; returns the per-stage wall clocks and what was written"""
    import logging
    from haphic_amd import _lib, cluster
    lg = logging.getLogger('HapHiC_cluster')
    level = lg.level
    lg.setLevel(logging.WARNING)
    sync = lambda: _lib.check(_lib.load().hhx_synchronize())            # noqa: E731
    names = list(gen.names)
    fa_dict = {nm: [None, int(ln), int(ln) // 256 + 1] for nm, ln in zip(names, gen.length.tolist())}      # parse_fasta's shape (:111); GATC density of uniform ACGT
    a = types.SimpleNamespace(flank=500, remove_allelic_links=0, remove_concentrated_links=False, max_read_pairs=200, nwindows=50, skip_clustering=not sweep)
    cwd = os.getcwd()
    os.chdir(workdir)
    t = {}
    try:
        t0 = time.perf_counter()
        # --Nx 100 and the reference's default filter thresholds (:2599-2632): the matrix order stays ~ the number of contigs (SURVEY 8, config note)
        _, bin_set, bin_size, frag_len_dict, nx, re_dict, split_set = cluster.stat_fragments(fa_dict, 'GATC', {}, set(), nchrs=nchrs, flank=500, Nx=100, bin_size=-1)
        assert not split_set, 'the synthetic contigs are shorter than the bin size: parse_alignments_for_ctgs is the path (SURVEY 8)'
        t['stat_fragments_s'] = time.perf_counter() - t0
        t1 = time.perf_counter()
        aln = cluster.pairs_generator_inter_ctgs(pairs_path, 'pairs')
        full, flank, HT, clm, frag_link, coord = cluster.parse_alignments_for_ctgs(aln, fa_dict, a, frag_len_dict, nx, 'int32', 'int32')
        sync()
        t['pairs_text_to_containers_s'] = time.perf_counter() - t1            # a1 (tokeniser + alignments.bed) + S5
        text_stats = dict(getattr(aln, 'stats', {}))
        t2 = time.perf_counter()
        cluster.output_pickle(HT, 'HT_link_dict', 'HT_links.pkl')
        del HT
        t['output_pickle_HT_call_s'] = time.perf_counter() - t2
        t3 = time.perf_counter()
        cluster.output_clm(clm)
        del clm
        t['output_clm_call_s'] = time.perf_counter() - t3
        t4 = time.perf_counter()
        kept = cluster.filter_fragments(nx, re_dict, 5, frag_link, '0.2X', '1.9X', 10, '1.5X', 0, flank, {}, '1.5X', set())
        sync()
        t['filter_fragments_s'] = time.perf_counter() - t4
        t5 = time.perf_counter()
        cluster.output_pickle(full, 'full_link_dict', 'full_links.pkl')
        t['output_pickle_full_call_s'] = time.perf_counter() - t5
        t6 = time.perf_counter()
        m, fidx = cluster.dict_to_matrix(flank, kept, dense_matrix=False, add_self_loops=True)
        sync()
        t['dict_to_matrix_s'] = time.perf_counter() - t6
        n_full, n_flank = len(full), len(flank)
        frozen = bool(full.frozen and flank.frozen)
        del flank
        t_matrix = time.perf_counter()
        t['link_matrix_ready_s'] = t_matrix - t0              # "Hi-C linking matrix was constructed in {}s" :2941
        pend_at_matrix = _lib.files_pending()[0]
        shape3 = m._dev.shape3                                  # (m.shape would download the scipy matrix the reference would have got)
        out = {'matrix_order': int(shape3[0]), 'matrix_nnz': int(shape3[2]), 'fragments_kept': len(kept), 'full_keys': n_full,
               'flank_keys': n_flank, 'files_still_queued_when_the_matrix_is_ready': pend_at_matrix,
               'containers_frozen_until_the_clustering': frozen}
        rounds = None
        # a 2 Hz timeline of the writer thread's queue and of the device memory, beside the sweep (who finishes when)
        import threading
        timeline, stop = [], threading.Event()

        def sample():
            while not stop.wait(0.5):
                try:
                    timeline.append([round(time.perf_counter() - t0, 2), _lib.files_pending()[0], round(_lib.mem_info()[0] / 1e9, 1)])
                except Exception:                # noqa: BLE001
                    break
        sampler = threading.Thread(target=sample, daemon=True)
        sampler.start()
        profile = bool(os.environ.get('C3_RUN_PROFILE'))     # measurement: the library's kernel clocks and pool counters over the sweep
        if sweep and profile:
            _lib.profile_reset()
            _lib.profile_enable(True)
        if sweep:
            res, rounds = cluster.run_mcl_clustering(m, bin_set, frag_len_dict, fidx, 2, 1.1, 3.0, 0.1, 200, 1e-4, fa_dict, nchrs, False)
            sync()
            t['run_mcl_clustering_s'] = time.perf_counter() - t_matrix
            if profile:
                _lib.profile_enable(False)
                out['sweep_kernel_ms_launches'] = {k: [round(_lib.profile_get(k)[0], 1), _lib.profile_get(k)[1]] for k in (
                    'expand_window', 'expand_window_short', 'expand_hash', 'expand_finalize', 'expand_compact', 'expand_tiny', 'class_layout', 'row_order', 'convergence',
                    'inflate_stats', 'prune_write', 'dense_epilogue', 'dense_transpose') if _lib.profile_get(k)[1]}
                out['sweep_pool'] = {k: _lib.profile_counter(k) for k in ('pool_fresh_bytes', 'pool_fresh_calls', 'pool_fresh_us', 'pool_trims_on_failure', 'expand_pool_retries')}
            out['sweep_stages_s'] = {k: ([round(x, 3) for x in v] if isinstance(v, list) else round(v, 3)) for k, v in cluster.SWEEP_STAGES.items()}
            out['mcl_rounds'] = rounds
            out['inflations_with_a_valid_partition'] = len(res)
            out['per_inflation_mcl_s_files_s'] = [[r[0]] + [round(x, 3) for x in r[1:]] for r in cluster.SWEEP_TIMING]     # inflation, mcl() + interpret, cluster lists on the caller's thread, the directory on the helper thread
        pend_at_end = _lib.files_pending()[0]
        tj = time.perf_counter()
        _lib.files_join()
        t['files_join_wait_s'] = time.perf_counter() - tj
        stop.set()
        out['timeline_s_filesqueued_freeGB'] = timeline[::4] if len(timeline) > 40 else timeline[::2]
        t['whole_job_s'] = time.perf_counter() - t0
        out['files_still_queued_when_the_clustering_ended'] = pend_at_end
        del full
        out['files'] = {f: os.path.getsize(f) for f in ('HT_links.pkl', 'paired_links.clm', 'full_links.pkl', 'alignments.bed') if os.path.exists(f)}
        out['inflation_dirs'] = len([d for d in os.listdir('.') if d.startswith('inflation_')])
        out['seconds'] = t
        out['text_stage'] = text_stats
        return out
    finally:
        os.chdir(cwd)
        lg.setLevel(level)


def run_job(contigs=100000, pairs=500_000_000, nchrs=24, mean_len=30_000, where=None, keep=False, sweep=True, device='cuda:0', arrays=None, gen=None):
    """generate the .pairs file (untimed), run the sequence, clean up.  arrays: (id1, p1, id2, p2) torch tensors already on the device (bench.py)"""
    import torch
    from haphic_amd import _lib, synth
    _lib.load()
    if gen is None:
        per_chr = max(1, contigs // nchrs)
        gen = synth.make_genome(nchrs, per_chr * mean_len, mean_len, seed=12345)
    name_bytes = float(np.mean([len(nm) for nm in gen.names]))
    line = 2 * name_bytes + 2 * 6 + 10 + 8                     # two names, two positions, the read id, tabs and strands
    need = pairs * (line + (2 * name_bytes + 2 * 12 + 36) + 56 + 12)      # .pairs + alignments.bed + paired_links.clm + pickles
    if where is None:
        for cand in ('/dev/shm', tempfile.gettempdir()):
            why = room_for(need, cand)
            if why is None:
                where = cand
                break
        if where is None:
            return {'skipped': why, 'bytes_needed_estimate': need}
    else:
        why = room_for(need, where)
        if why is not None:
            return {'skipped': why, 'bytes_needed_estimate': need}
    d = tempfile.mkdtemp(prefix='hhx_c3run_', dir=where)
    try:
        tg = time.perf_counter()
        if arrays is None:
            parts = [synth.sample_pairs(gen, min(250_000_000, pairs - lo), seed=12345 + 1 + 1000 * k, device=device) for k, lo in enumerate(range(0, pairs, 250_000_000))]
            arrays = [torch.cat([q[c] for q in parts]) if len(parts) > 1 else parts[0][c] for c in range(4)]
            del parts
        path = os.path.join(d, 'hic.pairs')
        size, lines = write_pairs_file(path, gen, *arrays)
        del arrays
        torch.cuda.synchronize()
        torch.cuda.empty_cache()
        gen_s = time.perf_counter() - tg
        os.makedirs(os.path.join(d, 'run'))
        out = run_sequence(path, gen, nchrs, os.path.join(d, 'run'), sweep=sweep)
        s = out['seconds']
        out.update(contigs=int(gen.n), pairs=int(lines), pairs_file_bytes=int(size), directory=where, pairs_file_written_in_s_untimed=gen_s,
                   pairs_per_s_link_matrix_ready=lines / s['link_matrix_ready_s'], pairs_text_GBs=size / s['pairs_text_to_containers_s'] / 1e9,
                   cgroup_memory_limit_and_current=cgroup_memory(),
                   what='run() :2829-2945 through haphic_amd.cluster from the .pairs FILE on one MI355X: stat_fragments -> pairs_generator_inter_ctgs (+ alignments.bed) -> '
                        'parse_alignments_for_ctgs -> output_pickle(HT) -> output_clm -> filter_fragments -> output_pickle(full) -> dict_to_matrix [link_matrix_ready_s] -> '
                        'run_mcl_clustering (20 inflations, every file written) -> files joined [whole_job_s]; the three writers only queue their file '
                        '(library thread), *_call_s is what the caller waits; fa_dict synthesised (no FASTA), output_statistics :2279 not included')
        try:
            with open('/sys/fs/cgroup/memory.peak') as f:
                out['cgroup_memory_peak'] = int(f.read())
        except (OSError, ValueError):
            pass
        return out
    finally:
        if not keep:
            shutil.rmtree(d, ignore_errors=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--contigs', type=int, default=100000)
    ap.add_argument('--pairs', type=int, default=500_000_000)
    ap.add_argument('--nchrs', type=int, default=24)
    ap.add_argument('--mean-len', type=int, default=30_000)
    ap.add_argument('--dir', default=None, help='where the .pairs file and the run directory go (default: /dev/shm if the memory cgroup has the room, else the temporary directory)')
    ap.add_argument('--no-sweep', action='store_true', help='stop when the link matrix is ready (--skip_clustering :2943)')
    ap.add_argument('--keep', action='store_true')
    ap.add_argument('--sync-files', action='store_true', help='HAPHIC_SYNC_FILES=1: the three writers on the caller\'s thread, as before round 6')
    args = ap.parse_args()
    if args.sync_files:
        os.environ['HAPHIC_SYNC_FILES'] = '1'
    from haphic_amd import _lib
    _lib.check(_lib.load().hhx_set_device(0))
    out = run_job(args.contigs, args.pairs, args.nchrs, args.mean_len, where=args.dir, keep=args.keep, sweep=not args.no_sweep)
    print(json.dumps(out))


if __name__ == '__main__':
    main()
