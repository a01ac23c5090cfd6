"""BASELINE.md §4, row C1, the MI355X side: `haphic cluster asm.fa hic.pairs 4` with every option at its default, run through the
mirrors of haphic_amd/cluster.py in the order of the reference's run() :2738-2959 (parse_fasta -> stat_fragments ->
pairs_generator_inter_ctgs -> parse_alignments_for_ctgs -> HT_links.pkl -> paired_links.clm -> filter_fragments -> full_links.pkl ->
dict_to_matrix -> run_mcl_clustering with its 20 inflation directories), from the same FASTA + .pairs FILES that
tools/reference_c1_baseline.py hands to the unmodified reference (same seeds).  The reference module itself does not exist on the GPU
box, so run()'s own glue (argparse, logging set-up) and output_statistics :2279 (the statistics for `haphic reassign`, after the
cluster files are written) are not part of this figure; the reference-side figure that corresponds is
`haphic_cluster_wall_s - output_statistics` of profiles/r05_reference_c1_baseline.json.  One JSON object."""
import json
import os
import sys
import tempfile
import time
import types

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import logging
    from haphic_amd import _lib, cluster, synth
    logging.getLogger('HapHiC_cluster').setLevel(logging.WARNING)
    gen = synth.make_genome(4, 25_000_000, 100_000, cv=0.3, min_len=5000, seed=12345)
    id1, p1, id2, p2 = [t.numpy() for t in synth.sample_pairs(gen, 1_000_000, seed=12345)]
    rng = np.random.default_rng(1)
    _lib.check(_lib.load().hhx_set_device(0))
    out = {'contigs': int(gen.n), 'pairs': 1_000_000, 'nchrs': 4}
    with tempfile.TemporaryDirectory() as td:
        fa = os.path.join(td, 'asm.fa')
        with open(fa, 'w') as f:
            for nm, ln in zip(gen.names, gen.length.tolist()):
                seq = rng.choice(np.frombuffer(b'ACGT', np.uint8), ln).tobytes().decode()
                f.write('>%s\n' % nm)
                f.write('\n'.join(seq[k:k + 80] for k in range(0, ln, 80)) + '\n')
        pairs = os.path.join(td, 'hic.pairs')
        with open(pairs, 'w') as f:
            f.write('## pairs format v1.0\n#columns: readID chr1 pos1 chr2 pos2 strand1 strand2\n')
            for k, (a, x, b, y) in enumerate(zip(id1.tolist(), p1.tolist(), id2.tolist(), p2.tolist())):
                f.write('r%d\t%s\t%d\t%s\t%d\t+\t-\n' % (k, gen.names[a], x + 1, gen.names[b], y + 1))
        os.makedirs(os.path.join(td, 'run'))
        os.chdir(os.path.join(td, 'run'))
        # the defaults of parse_arguments :2560-2730
        a = types.SimpleNamespace(flank=500, remove_allelic_links=0, remove_concentrated_links=False, max_read_pairs=200, nwindows=50)
        for attempt in ('warm-up (library load, first kernel launches)', 'timed'):
            for fn in os.listdir('.'):
                if os.path.isfile(fn):
                    os.remove(fn)
            t = [time.perf_counter()]
            fa_dict = cluster.parse_fasta(fa, RE='GATC')
            t.append(time.perf_counter())
            _, bin_set, bin_size, frag_len_dict, nx, re_dict, split_set = cluster.stat_fragments(fa_dict, 'GATC', {}, set(), nchrs=4, flank=500, Nx=80, bin_size=-1)
            t.append(time.perf_counter())
            assert not split_set
            aln = cluster.pairs_generator_inter_ctgs(pairs, 'pairs')
            full, flank, HT, clm, frag_link, coord = cluster.parse_alignments_for_ctgs(aln, fa_dict, a, frag_len_dict, nx, 'int32', 'int32')
            t.append(time.perf_counter())
            cluster.output_pickle(HT, 'HT_link_dict', 'HT_links.pkl')
            cluster.output_clm(clm)
            t.append(time.perf_counter())
            kept = cluster.filter_fragments(nx, re_dict, 5, frag_link, '0.2X', '1.9X', 10, '1.5X', 0, flank, {}, '1.5X', set())
            cluster.output_pickle(full, 'full_link_dict', 'full_links.pkl')
            m, fidx = cluster.dict_to_matrix(flank, kept, dense_matrix=False, add_self_loops=True)
            t.append(time.perf_counter())
            res, rounds = cluster.run_mcl_clustering(m, bin_set, frag_len_dict, fidx, 2, 1.1, 3.0, 0.1, 200, 1e-4, fa_dict, 4, False)
            _lib.files_join()                 # alignments.bed, HT_links.pkl, paired_links.clm, full_links.pkl: queued on the library's writer thread, complete here
            t.append(time.perf_counter())
            frozen = all(c.frozen for c in (full, flank, HT, clm))
        out.update(wall_s=t[-1] - t[0], parse_fasta_s=t[1] - t[0], stat_fragments_s=t[2] - t[1], parse_alignments_for_ctgs_s=t[3] - t[2],
                   ht_pickle_and_clm_s=t[4] - t[3], filter_pickle_dict_to_matrix_s=t[5] - t[4], run_mcl_clustering_s=t[6] - t[5],
                   matrix_constructed_s=t[5] - t[0], mcl_rounds=rounds, fragments_kept=len(kept), inflation_dirs=len([d for d in os.listdir('.') if d.startswith('inflation_')]),
                   files={f: os.path.getsize(f) for f in ('HT_links.pkl', 'paired_links.clm', 'full_links.pkl', 'alignments.bed') if os.path.exists(f)},
                   containers_still_frozen=frozen,
                   what='the seam sequence of run() through haphic_amd.cluster on one MI355X, second of two runs in one process; FASTA and .pairs read from files, '
                        'every file of `haphic cluster` written except the statistics of output_statistics :2279')
        os.chdir('/')
    print(json.dumps(out))


if __name__ == '__main__':
    main()
