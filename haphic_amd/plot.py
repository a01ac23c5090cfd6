"""`haphic plot` read-pair binning on the MI355X (second half): mirrors of HapHiC_plot.py
parse_pairs :153-202 and parse_bam :205-245 with the reference's signatures.  parse_agp / generate_contact_matrix /
normalisation / drawing stay the reference's code; the per-read-pair loop runs on the device (hhx_contact_map_*):

    import HapHiC_plot as P
    import haphic_amd.plot
    haphic_amd.plot.patch_plot(P)        # then P.main() as usual

The reference's range dicts are flattened once per call (O(scaffold bins) Python work), the alignment file goes through
the same device front ends as the cluster step (.pairs text: hhx_pairs_parser_*; BAM: hhx_bam_*)."""
import numpy as np

from . import _lib, cluster
from .cluster import logger


class ContactTable:
    """ctg_dict / ctg_aln_dict / group_to_total_bin_dict of HapHiC_plot.py :41-103 :106-150 as the arrays of
    hhx_contact_map_create.  A range key is anything with .lower / .upper (portion's closed intervals in the reference)."""

    def __init__(self, ctg_dict, ctg_aln_dict, bin_size, group_to_total_bin_dict, group_list, ctg_set, n_total_bins):
        self.names = list(ctg_aln_dict.keys())
        self.bin_size, self.n_total_bins = int(bin_size), int(n_total_bins)
        drawn = set(group_list)
        in_set, aln_ptr, list_ptr, lo, hi, cell = [], [0], [0], [], [], []
        for ctg in self.names:
            in_set.append(1 if ctg in ctg_set else 0)
            bins = ctg_aln_dict[ctg]
            ranges = ctg_dict[ctg]
            for aln_bin in range((max(bins) + 1) if bins else 0):
                for r in bins.get(aln_bin, ()):              # list order = the order convert_group_bin_id tries them (:158)
                    group_and_bin = ranges[r]                # :159 — the LAST AGP line that named this exact range
                    lo.append(r.lower)
                    hi.append(r.upper)
                    cell.append(group_to_total_bin_dict[group_and_bin] if group_and_bin[0] in drawn else -1)      # :161-163
                list_ptr.append(len(lo))
            aln_ptr.append(len(list_ptr) - 1)
        limit = np.iinfo(np.int32).max
        if hi and max(hi) >= limit:
            raise RuntimeError('contig coordinates beyond int32 are not supported by the MI355X contact map')
        self.arrays = (np.array(in_set, np.uint8), np.array(aln_ptr, np.int64), np.array(list_ptr, np.int32), np.array(lo, np.int32),
                       np.array(hi, np.int32), np.array(cell, np.int32))

    def device(self):
        return _lib.ContactMap(*self.arrays, self.bin_size, self.n_total_bins)


def _raise_missing(ctg, pos, what):
    """convert_group_bin_id's KeyError branch :164-168 / :216-220"""
    error_message = ('Cannot find alignment position: {}:{} in the input AGP file. Please check whether the input AGP and {} files '
                     'match'.format(ctg, pos, what))
    logger.error(error_message)
    raise Exception(error_message)


def _bin_batches(batches, table, contact_matrix, pos_offset, what, host_arrays):
    """batches: iterable of (n, [id1, pos1, id2, pos2] device pointers); host_arrays(): the last batch on the host"""
    cm = table.device()
    try:
        for n, ptrs in batches:
            if not n:
                continue
            bad = cm.push_device(n, *ptrs, pos_offset=pos_offset)
            if bad >= 0:
                id1, p1, id2, p2 = host_arrays()
                k, side = bad >> 1, bad & 1
                _raise_missing(table.names[(id2 if side else id1)[k]], int((p2 if side else p1)[k]) + pos_offset, what)
        contact_matrix += cm.fetch().astype(contact_matrix.dtype, copy=False)
    finally:
        cm.destroy()
    return contact_matrix


def parse_pairs(pairs, ctg_dict, ctg_aln_dict, bin_size, contact_matrix, group_to_total_bin_dict, group_list, ctg_set):
    """parse_pairs() :153-202"""
    logger.info('Parsing input pairs file...')
    if pairs.endswith('.pairs'):
        fmt = 'pairs'
    else:
        assert pairs.endswith('.pairs.gz')
        fmt = 'bgzipped_pairs'
    table = ContactTable(ctg_dict, ctg_aln_dict, bin_size, group_to_total_bin_dict, group_list, ctg_set, contact_matrix.shape[0])
    text = cluster.PairsText(pairs, fmt, inter_only=False, bed_path=None)
    state = {}

    def batches():
        for parser, n in text.batches(table.names):
            state['parser'] = parser
            yield n, parser.device_arrays()[:4]

    # the device tokeniser yields the cluster step's 0-based positions (int(cols[2]) - 1, HapHiC_cluster.py :1556): + 1 back
    return _bin_batches(batches(), table, contact_matrix, 1, '.pairs', lambda: state['parser'].fetch()[:4])


def parse_bam(bam, ctg_dict, ctg_aln_dict, bin_size, contact_matrix, group_to_total_bin_dict, group_list, ctg_set, threads):
    """parse_bam() :205-245 (format_options = [b'filter=flag.read1'] :222; positions are reference_start + 1 :232 :236)"""
    logger.info('Parsing input BAM file...')
    table = ContactTable(ctg_dict, ctg_aln_dict, bin_size, group_to_total_bin_dict, group_list, ctg_set, contact_matrix.shape[0])
    state = {}

    def batches():
        reader = _lib.BamReader(bam, threads)
        state['reader'] = reader
        try:
            reader.set_contigs({n: i for i, n in enumerate(table.names)})
            while True:
                n, ptrs = reader.next_batch(0x40, False)
                if n == 0:
                    break
                yield n, ptrs
        finally:
            reader.close()

    return _bin_batches(batches(), table, contact_matrix, 1, 'BAM', lambda: state['reader'].fetch())


def patch_plot(P):
    """P: the imported HapHiC_plot module.  Returns {name: original}."""
    _lib.load()
    saved = {'parse_pairs': P.parse_pairs, 'parse_bam': P.parse_bam}
    P.parse_pairs, P.parse_bam = parse_pairs, parse_bam
    return saved
