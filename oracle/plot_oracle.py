"""CPU restatement of the `haphic plot` binning path: HapHiC_plot.py parse_agp :41-103,
generate_contact_matrix :106-150, parse_pairs :153-202 / parse_bam :205-245 — plain Python loops over dicts, the way the
reference does it, for small cases.

TEST INFRASTRUCTURE ONLY (imported by tests/ alone, never by haphic_amd/).  Pinned by tests/golden/plot.npz: contact
matrices produced HERE by the reference's own functions (tests/golden/make_golden.py gen_plot, with a stand-in for the
absent `portion` package), compared in tests/test_plot.py.

`Closed` stands for portion's closed interval: the reference only builds closed(a, b), intersects two of them, reads
.lower / .upper, tests `pos in interval` and uses intervals as dict keys."""
import collections

import numpy as np


class Closed(collections.namedtuple('Closed', 'lower upper')):
    __slots__ = ()

    def __and__(self, other):
        return Closed(max(self.lower, other.lower), min(self.upper, other.upper))

    def __contains__(self, pos):
        return self.lower <= pos <= self.upper


def parse_agp(agp_text, bin_size):
    """:41-103.  Returns the reference's five containers (same shapes, Closed for the interval keys)."""
    ctg_dict = collections.defaultdict(dict)            # ctg -> {range on the raw contig: (scaffold, scaffold bin)}
    ctg_aln_dict = collections.defaultdict(dict)        # ctg -> {alignment bin: [ranges touching it]}
    group_size_dict = collections.OrderedDict()
    frag_set = set()
    group_frag_dict = collections.defaultdict(set)
    for line in agp_text.splitlines():
        if line.startswith('#') or not line.strip():
            continue
        cols = line.split()
        if cols[4] != 'W':                               # gap lines :60
            continue
        group, g_lo, g_hi = cols[0], int(cols[1]), int(cols[2])
        ctg, c_lo, c_hi, strand = cols[5], int(cols[6]), int(cols[7]), cols[8]
        group_size_dict[group] = g_hi                    # the last component line of a scaffold fixes its size :69
        frag = (ctg, c_lo, c_hi)
        frag_set.add(frag)
        group_frag_dict[group].add(frag)
        for gbin in range((g_lo - 1) // bin_size, (g_hi - 1) // bin_size + 1):        # :63-64 :75
            part = Closed(gbin * bin_size + 1, (gbin + 1) * bin_size) & Closed(g_lo, g_hi)     # :77-79
            if strand == '+':                            # :82-84
                raw = Closed(part.lower - g_lo + c_lo, part.upper - g_lo + c_lo)
            else:                                        # :86-88
                assert strand == '-'
                raw = Closed(c_hi - (part.upper - g_lo), c_hi - (part.lower - g_lo))
            ctg_dict[ctg][raw] = (group, gbin)           # :91
            for aln_bin in range((raw.lower - 1) // bin_size, (raw.upper - 1) // bin_size + 1):   # :93-100
                ctg_aln_dict[ctg].setdefault(aln_bin, []).append(raw)
    return ctg_dict, ctg_aln_dict, group_size_dict, frag_set, group_frag_dict


def generate_contact_matrix(group_size_dict, frag_set, group_frag_dict, bin_size, min_len, specified_scaffolds):
    """:106-150.  min_len in Mb as on the command line; frag_set is reduced in place like the reference does (:141)."""
    min_len = min_len * 1000000
    total = 0
    to_total = {}
    group_list = []
    if specified_scaffolds:                              # :125-135: the user's order, no length filter
        for group in specified_scaffolds.split(','):
            if group not in group_size_dict:
                raise RuntimeError('Cannot find {} in the input AGP file'.format(group))
            nb = group_size_dict[group] // bin_size + 1
            for k in range(nb):
                to_total[(group, k)] = total + k
            total += nb
            group_list.append(group)
    else:                                                # :137-146: AGP order, scaffolds >= min_len
        for group, size in group_size_dict.items():
            if size >= min_len:
                nb = size // bin_size + 1
                for k in range(nb):
                    to_total[(group, k)] = total + k
                total += nb
                group_list.append(group)
            else:
                frag_set -= group_frag_dict[group]
    ctg_set = {frag[0] for frag in frag_set}             # :146-148
    return np.zeros((total, total), dtype=int), to_total, group_list, ctg_set


def convert_group_bin_id(ctg, pos, ctg_dict, ctg_aln_dict, bin_size, to_total, group_list):
    """:155-168.  KeyError (alignment bin not in the AGP) -> Exception with the reference's message; None as there."""
    try:
        for raw in ctg_aln_dict[ctg][(pos - 1) // bin_size]:
            group_and_bin = ctg_dict[ctg][raw]
            if pos in raw:
                if group_and_bin[0] not in group_list:
                    return None
                return to_total[group_and_bin]
    except KeyError as e:
        raise Exception('Cannot find alignment position: {}:{} in the input AGP file'.format(ctg, pos)) from e
    return None


def bin_pairs(records, ctg_dict, ctg_aln_dict, bin_size, contact_matrix, to_total, group_list, ctg_set):
    """the loop body shared by parse_pairs :184-200 and parse_bam :228-243; records = (ref, pos, mref, mpos), 1-based"""
    for ref, pos, mref, mpos in records:
        if ref not in ctg_set or mref not in ctg_set:
            continue
        a = convert_group_bin_id(ref, pos, ctg_dict, ctg_aln_dict, bin_size, to_total, group_list)
        if a is None:
            continue
        b = convert_group_bin_id(mref, mpos, ctg_dict, ctg_aln_dict, bin_size, to_total, group_list)
        if b is None:
            continue
        contact_matrix[a, b] += 1
    return contact_matrix


def pairs_records(text):
    """parse_pairs :173-182: the (ref, pos, mref, mpos) of every body line of a .pairs text"""
    for line in text.splitlines():
        if not line.strip() or line.startswith('#'):
            continue
        cols = line.split()
        yield cols[1], int(cols[2]), cols[3], int(cols[4])


def bin_flat(in_set, aln_ptr, list_ptr, seg_lo, seg_hi, seg_bin, bin_size, n_bins, id1, pos1, id2, pos2, pos_offset=0):
    """the same loop over the FLATTENED tables of include/haphic_hip.h (hhx_contact_map_create), vectorised with numpy for
    million-pair cases: -> (matrix int64 [n_bins, n_bins], bad) with bad = -1 or 2 * k + side of the first KeyError."""
    id1, id2 = np.asarray(id1, np.int64), np.asarray(id2, np.int64)
    n_ctg = len(in_set)

    def convert(ids, pos):
        out = np.full(len(ids), -1, np.int64)            # -1 None, -2 KeyError
        q = pos - 1
        slot = aln_ptr[ids] + np.where(q >= 0, q // bin_size, 0)
        key_error = (q < 0) | (slot >= aln_ptr[ids + 1])
        slot = np.where(key_error, 0, slot)
        b, e = list_ptr[slot].astype(np.int64), list_ptr[slot + 1].astype(np.int64)
        key_error |= b == e
        out[key_error] = -2
        open_ = ~key_error
        depth = int((e - b)[open_].max()) if open_.any() else 0
        for d in range(depth):                           # first range of the list that holds the position
            s = np.minimum(b + d, max(len(seg_lo) - 1, 0))
            hit = open_ & (b + d < e) & (pos >= seg_lo[s]) & (pos <= seg_hi[s])
            out[hit] = seg_bin[s][hit]
            open_ &= ~hit
        return out

    ok = (id1 >= 0) & (id2 >= 0) & (id1 < n_ctg) & (id2 < n_ctg)
    ok[ok] &= (in_set[id1[ok]] != 0) & (in_set[id2[ok]] != 0)
    idx = np.flatnonzero(ok)
    a = convert(id1[idx], np.asarray(pos1, np.int64)[idx] + pos_offset)
    look_b = a >= 0                                      # the mate is only converted when the first end gave a bin
    b = np.full(len(idx), -1, np.int64)
    b[look_b] = convert(id2[idx][look_b], np.asarray(pos2, np.int64)[idx][look_b] + pos_offset)
    bad_at = np.concatenate([2 * idx[a == -2], 2 * idx[look_b & (b == -2)] + 1])
    bad = int(bad_at.min()) if len(bad_at) else -1
    good = (a >= 0) & (b >= 0)
    mat = np.zeros((n_bins, n_bins), np.int64)
    np.add.at(mat, (a[good], b[good]), 1)
    return mat, bad
