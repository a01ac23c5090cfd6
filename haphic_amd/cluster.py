"""Host-side mirror of the reference's interface for the hot path (scripts/HapHiC_cluster.py).

Same names, argument meaning and error behaviour as the reference functions, so that they can be
re-bound over the reference module (haphic_amd.patch_reference / INTEGRATION.md):

    S1  dot_product_mkl(A, B)                       :39-43, :2017-2023
    S3  normalize(M, norm='l1', axis=0), prune()    :1987-2014
    S2  mcl(matrix, expansion, inflation, iters, pruning, dense_matrix)   :2026-2062
    a12 interpret_result(result_matrix, dense_matrix)                     :2065-2095
    S4  dict_to_matrix(link_dict, frag_set, dense_matrix, add_self_loops) :310-373
    S5  parse_alignments_for_ctgs(...) / parse_alignments(...)            :1596-1752
    S6  run_mcl_clustering(...)                                           :2132-2242

All arithmetic runs in libhaphic_hip.so; there is no CPU fallback.  What stays in Python is what the
reference's observable ordering depends on CPython for: the set-of-tuples in interpret_result
(:2073-2095), the set difference that numbers link-less fragments (:357-359), stable sorts (:2197).
"""
import logging
import os
import sys
import time
from array import array
from collections import defaultdict
from decimal import Decimal
from math import ceil

import numpy as np

from . import _lib

logger = logging.getLogger('HapHiC_cluster')   # same logger name family; handlers come from the caller


# ------------------------------------------------------------------ integer view of the assembly
class FragTable:
    """Contig / fragment tables the id-based kernels need (see include/haphic_hip.h, ingest)."""

    def __init__(self, ctg_names, ctg_rank, ctg_len, ctg_frag0, ctg_split, bin_size, frag_names, frag_rank,
                 frag_len, frag_nx):
        self.ctg_names = ctg_names
        self.frag_names = frag_names
        self.ctg_rank = np.ascontiguousarray(ctg_rank, np.int32)
        self.ctg_len = np.ascontiguousarray(ctg_len, np.int64)
        self.ctg_frag0 = np.ascontiguousarray(ctg_frag0, np.int32)
        self.ctg_split = np.ascontiguousarray(ctg_split, np.uint8)
        self.bin_size = int(bin_size)
        self.frag_rank = np.ascontiguousarray(frag_rank, np.int32)
        self.frag_len = np.ascontiguousarray(frag_len, np.int64)
        self.frag_nx = np.ascontiguousarray(frag_nx, np.uint8)
        self.n_ctg = len(self.ctg_rank)
        self.n_frag = len(self.frag_rank)
        # determine_int_type :116-147: a contig longer than 2^31 - 1 bp switches the reference's coordinate arrays to int64; here
        # it switches the position arrays of the device path (tokeniser, ingest) to 64 bits
        self.wide = bool(self.n_ctg and int(self.ctg_len.max()) > 2 ** 31 - 1)

    @staticmethod
    def _rank(names):
        order = sorted(range(len(names)), key=names.__getitem__)     # Python str order == the reference's sorted()
        r = np.empty(len(names), np.int32)
        r[order] = np.arange(len(names), dtype=np.int32)
        return r

    @classmethod
    def for_contigs(cls, ctg_rank, ctg_len, nx, names=None):
        """no contig is split (parse_alignments_for_ctgs): fragment == contig"""
        n = len(ctg_rank)
        return cls(names, ctg_rank, ctg_len, np.arange(n, dtype=np.int32), np.zeros(n, np.uint8), 0, names, ctg_rank,
                   ctg_len, nx)

    @classmethod
    def from_reference(cls, fa_dict, frag_len_dict, Nx_frag_set, split_ctg_set=(), bin_size=0):
        """Build from the reference's own containers: fa_dict (parse_fasta :87-113) and the outputs of
        stat_fragments (:188-296).  Fragment ids follow fa_dict order; a split contig owns
        ceil(len / bin_size) consecutive ids named '{ctg}_bin{k}' (:233)."""
        names = list(fa_dict)
        frag_names, frag0, split = [], [], []
        for c in names:
            frag0.append(len(frag_names))
            if c in split_ctg_set:
                split.append(1)
                nb = int(ceil(fa_dict[c][1] / bin_size))
                frag_names.extend('{}_bin{}'.format(c, k + 1) for k in range(nb))
            else:
                split.append(0)
                frag_names.append(c)
        return cls(names, cls._rank(names), [fa_dict[c][1] for c in names], frag0, split,
                   bin_size if split_ctg_set else 0, frag_names, cls._rank(frag_names),
                   [frag_len_dict[f] for f in frag_names], [f in Nx_frag_set for f in frag_names])


# ------------------------------------------------------------------ a5: restriction sites, fragment statistics
def parse_RE_sites(sites):
    """parse_RE_sites() :56-72 — every N of a site stands for A / T / C / G.  The reference expands the first N of each
    site, then re-scans the whole list; the resulting ORDER (it decides nothing downstream: the counts are summed) is
    reproduced by expanding the wildcards left to right with the leftmost one varying slowest, in A, T, C, G order."""
    from itertools import product
    expanded = []
    for site in sites:
        parts = site.split('N')
        for letters in product('ATCG', repeat=len(parts) - 1):
            expanded.append(''.join(p + l for p, l in zip(parts, letters + ('',))))
    return expanded


def _sites_of(RE):
    sites = [site.strip().upper() for site in RE.split(',') if site.strip()]
    return [x.encode() for x in parse_RE_sites(sites)]


def count_RE_sites(seq, RE):
    """count_RE_sites() :75-84 on one sequence (str or bytes)"""
    buf = seq.encode() if isinstance(seq, str) else bytes(seq)
    return int(_lib.count_re_sites(buf, [0], [len(buf)], _sites_of(RE))[0])


def parse_fasta(fasta, RE='GATC', keep_letter_case=False, logger=logger):
    """parse_fasta() :87-113 — fa_dict[ctg] = [seq, len, RE sites + 1]; the RE sites of all contigs are
    counted in one device pass over the genome bytes"""
    logger.info('Parsing input FASTA file...')
    fa_dict = dict()
    with open(fasta) as f:
        for line in f:
            if not line.strip():
                continue
            if line.startswith('>'):
                ctg = line.split()[0][1:]
                fa_dict[ctg] = list()
            else:
                fa_dict[ctg].append(line.strip() if keep_letter_case else line.strip().upper())
    seqs = {ctg: ''.join(seq_list) for ctg, seq_list in fa_dict.items()}
    lens = np.fromiter((len(x) for x in seqs.values()), np.int64, len(seqs))
    offs = np.concatenate([[0], np.cumsum(lens)[:-1]]).astype(np.int64) if len(lens) else np.zeros(0, np.int64)
    counts = _lib.count_re_sites(''.join(seqs.values()).encode(), offs, lens, _sites_of(RE)) if len(lens) else []
    for k, (ctg, seq) in enumerate(seqs.items()):
        fa_dict[ctg] = [seq, len(seq), int(counts[k]) + 1]       # pseudo-count of 1 (:110)
    return fa_dict


def _resolve_bin_size(bin_size, total_len, nchrs, logger):
    """--bin_size semantics (:208-220): 0 = never split, negative = derive from the genome, positive = kbp"""
    from math import inf
    if not bin_size:
        logger.info('bin_size is set to {}, no fragments will be split'.format(bin_size))
        return inf
    if bin_size < 0:
        derived = max(min(int(total_len / nchrs / 30), 2000000), 100000)
        logger.info('bin_size is calculated to be {} bp'.format(derived))
        return derived
    logger.info('bin_size is manually designated to {} bp'.format(bin_size * 1000))
    return bin_size * 1000


class _ReCountBatch:
    """RE-site counts of many slices of the genome in ONE device pass (hhx_count_re_sites): callers register slices of
    the contig whose sequence was appended last, `resolve` returns count + 1 (the reference's pseudo-count) per key."""

    def __init__(self, flank):
        self.flank, self.pieces, self.cursor = flank, [], 0
        self.keys, self.off, self.len = [], [], []

    def add_sequence(self, seq):
        self.base = self.cursor
        self.pieces.append(seq)
        self.cursor += len(seq)

    def want(self, key, start, length):
        """whole slice, or only its two flanks when it is longer than both together (:192-199)"""
        f = self.flank
        spans = [(start, length)] if (not f or length <= 2 * f) else [(start, f), (start + length - f, f)]
        for o, n in spans:
            self.keys.append(key); self.off.append(self.base + o); self.len.append(n)

    def resolve(self, RE):
        totals = {}
        if self.keys:
            counts = _lib.count_re_sites(''.join(self.pieces).encode(), self.off, self.len, _sites_of(RE))
            for key, c in zip(self.keys, counts.tolist()):
                totals[key] = totals.get(key, 1) + c
        return totals


def stat_fragments(fa_dict, RE, read_depth_dict, whitelist, nchrs=0, flank=0, Nx=100, bin_size=0, logger=logger):
    """stat_fragments() :188-296 with every RE-site count of the bins / flanks coming from one device call.  Same seven
    return values, same side effects on fa_dict (sequences dropped) and read_depth_dict (bins inherit their contig's
    entry), same seeded shuffle before the length sort that defines the Nx set."""
    import random
    logger.info('Making some statistics of fragments (contigs / bins)')
    total_len = sum(info[1] for info in fa_dict.values())
    bin_size = _resolve_bin_size(bin_size, total_len, nchrs, logger)
    batch = _ReCountBatch(flank * 1000)
    frag_len_dict, known_sites = {}, {}
    bin_set, split_ctg_set = set(), set()
    for ctg, info in fa_dict.items():
        seq, ctg_len, whole_contig_sites = info
        if ctg_len > bin_size:                                       # :230 — the contig becomes ceil(len / bin_size) bins
            split_ctg_set.add(ctg)
            batch.add_sequence(seq)
            starts = range(0, ctg_len, bin_size)
            for m, start in enumerate(starts, 1):
                name = '{}_bin{}'.format(ctg, m)
                assert name not in fa_dict
                bin_set.add(name)
                frag_len_dict[name] = min(bin_size, ctg_len - start)
                batch.want(name, start, frag_len_dict[name])
                if read_depth_dict:
                    read_depth_dict[name] = read_depth_dict[ctg]
            if read_depth_dict:
                del read_depth_dict[ctg]
        else:
            frag_len_dict[ctg] = ctg_len
            if not batch.flank or ctg_len <= 2 * batch.flank:
                known_sites[ctg] = whole_contig_sites                # parse_fasta already counted the whole contig
            else:
                batch.add_sequence(seq)
                batch.want(ctg, 0, ctg_len)
        info[0] = None                                               # :266
    known_sites.update(batch.resolve(RE))
    RE_site_dict = {frag: known_sites[frag] for frag in frag_len_dict}
    # Nx set (:268-288): seeded shuffle, stable sort by length, keep while the running share is below Nx, plus one
    order = list(frag_len_dict)
    random.seed(12345)
    random.shuffle(order)
    sorted_frag_list = sorted(((frag, frag_len_dict[frag]) for frag in order), key=lambda item: item[1], reverse=True)
    Nx_frag_set, running = set(), 0
    for frag, length in sorted_frag_list:
        running += length
        if Nx == 100 or running / total_len * 100 < Nx:
            Nx_frag_set.add(frag)
    if Nx != 100:
        Nx_frag_set.add(sorted_frag_list[len(Nx_frag_set)][0])
    if whitelist:
        Nx_frag_set.update(frag for frag, _ in sorted_frag_list if frag.rsplit('_bin', 1)[0] in whitelist)
    return sorted_frag_list, bin_set, bin_size, frag_len_dict, Nx_frag_set, RE_site_dict, split_ctg_set


# ------------------------------------------------------------------ f1: filter_fragments
def check_param(param, string, suffix, true_suffix=''):
    """check_param() :2481-2507 — "0.2" (a fraction, must lie in [0, 1]) or "1.5X" (a multiple: any number followed by
    one of the `suffix` characters).  Returns (number, suffix character or ''); logs and raises like the reference."""
    if not string:
        logger.error('Parameter {} is empty'.format(param))
        raise RuntimeError('Parameter check failed')
    text, mode = string, true_suffix
    if suffix and len(text) > 1 and text[-1] in suffix:
        text, mode = text[:-1], text[-1]
    try:
        number = float(text)
    except ValueError:
        number = None
    if number is not None and (mode or 0 <= number <= 1):
        return number, mode
    logger.error('Parameter {} {} is illegal'.format(param, string + true_suffix))
    raise RuntimeError('Parameter check failed')


def _cut(stage, sorted_values, string, flag, scale, offset=0.0, strict=True):
    """Index at which an ascending list is cut by a "0.2"-style (fraction of the list) or "0.2X"-style (multiple of
    `scale`, plus `offset`) parameter — the two modes of the reference's thresholds (:779-810, :841-854, :911-923) —
    with the reference's log line that restates the parameter in the other mode.
    strict: first value > limit (upper cuts); otherwise first value >= limit (the lower density cut)."""
    from bisect import bisect_left, bisect_right
    num, mode = check_param(flag, string, {'X', 'x'})
    n = len(sorted_values)
    if mode:
        limit = offset + num * scale
        idx = (bisect_right if strict else bisect_left)(sorted_values, limit)
        logger.info('[{}] Parameter {} {} is set to "multiple" mode and equivalent to {} in "fraction" mode'.format(
            stage, flag, string, idx / n))
    else:
        idx = int(n * float(string))
        logger.info('[{}] Parameter {} {} is set to "fraction" mode and equivalent to {}X in "multiple" mode'.format(
            stage, flag, string, (sorted_values[max(0, idx - 1)] - offset) / scale))
    return idx


def filter_fragments(Nx_frag_set, RE_site_dict, RE_site_cutoff, frag_link_dict, density_lower, density_upper,
                     topN, rank_sum_upper, rank_sum_hard_cutoff, flank_link_dict, read_depth_dict, read_depth_upper, whitelist):
    """filter_fragments() :741-940.  The Nx / RE-site / link-density / read-depth steps are O(n) list work on the
    host; the rank-sum statistic (:866-892) — a dense n x n matrix and O(n^2 log n) Python sorting in the
    reference — is computed on the device from the sparse link matrix (hhx_rank_sums).  Every info / debug line of the
    reference is emitted (users and HapHiC_pipeline read that log)."""
    from numpy import quantile
    logger.info('Filtering fragments...')
    # (1) Nx set, (2) RE sites: link density = flank links / RE sites (:753-765); set iteration order, then a stable sort
    density, total_links, total_RE_sites = [], 0, 1
    for frag in Nx_frag_set:
        sites = RE_site_dict[frag]
        if sites > RE_site_cutoff:
            links = frag_link_dict.get(frag) if frag in frag_link_dict else None
            if links is None:
                density.append((frag, 0))
            else:
                total_links += links
                total_RE_sites += sites - 1
                density.append((frag, links / sites))
    whitelisted = {f for f in Nx_frag_set if whitelist and f.rsplit('_bin', 1)[0] in whitelist}
    logger.info('[Nx filtering] {} fragments kept'.format(len(Nx_frag_set)))
    logger.info('[RE sites filtering] {} fragments removed, {} fragments kept'.format(len(Nx_frag_set) - len(density), len(density)))
    # (3) link density between density_lower and density_upper (:767-823)
    density.sort(key=lambda x: x[1])
    values = [d for _, d in density]
    average = total_links / total_RE_sites
    lower = _cut('link density filtering', values, density_lower, '--density_lower', average, strict=False)
    upper = _cut('link density filtering', values, density_upper, '--density_upper', average)
    unfiltered = density
    density = density[lower:upper]
    kept = {frag for frag, _ in density}
    logger.info('[link density filtering] {} fragments removed, {} fragments kept'.format(len(unfiltered) - len(kept), len(kept)))
    for frag, d in unfiltered[:lower] + unfiltered[upper:]:
        logger.debug('[link density filtering] Fragment {} is removed, density={}'.format(frag, d))
    # (4) read depth: Q3 + k * IQR over ALL density-ranked fragments (:825-863)
    if read_depth_dict:
        depth = sorted(((frag, read_depth_dict[frag][1]) for frag, _ in unfiltered), key=lambda x: x[1])
        q1, m, q3 = quantile([d for _, d in depth], (0.25, 0.5, 0.75))
        logger.info('[read depth filtering] Q1={}, median={}, Q3={}, IQR=Q3-Q1={}'.format(q1, m, q3, q3 - q1))
        cut = _cut('read depth filtering', [d for _, d in depth], read_depth_upper, '--read_depth_upper', q3 - q1, offset=q3)
        kept &= {frag for frag, _ in depth[:cut]}
        by_density = {frag for frag, _ in unfiltered[:lower] + unfiltered[upper:]}
        only_by_depth = {frag for frag, _ in depth[cut:]} - by_density
        logger.info('[read depth filtering] {} fragments removed, {} fragments kept'.format(len(only_by_depth), len(kept)))
        for frag, d in depth[cut:]:
            if frag in only_by_depth:
                logger.debug('[read depth filtering] Fragment {} is removed, read depth={}'.format(frag, d))
        density = [(frag, d) for frag, d in density if frag in kept]
    # (5) rank sums between the topN strongest neighbours (:866-892), on the device
    m_dev, frag_index_dict = dict_to_matrix(flank_link_dict, kept, dense_matrix=False, add_self_loops=False, _device=True)
    try:
        rs = _lib.rank_sums(m_dev, topN)
    finally:
        m_dev.free()
    ranked = [(frag, int(rs[frag_index_dict[frag]])) for frag, _ in density]
    if rank_sum_hard_cutoff:
        for frag, r in ranked:
            if r > rank_sum_hard_cutoff:
                logger.debug('[rank sum filtering] Fragment {} is removed by hard filtering, rank sum={}'.format(frag, r))
        n_before = len(ranked)
        ranked = [x for x in ranked if x[1] <= rank_sum_hard_cutoff]
        logger.info('[rank sum filtering] {} fragments removed by hard filtering, {} fragments kept'.format(n_before - len(ranked), len(ranked)))
    ranked.sort(key=lambda x: x[1])
    sums = [r for _, r in ranked]
    q1, m, q3 = quantile(sums, (0.25, 0.5, 0.75))
    logger.info('[rank sum filtering] Q1={}, median={}, Q3={}, IQR=Q3-Q1={}'.format(q1, m, q3, q3 - q1))
    cut = _cut('rank sum filtering', sums, rank_sum_upper, '--rank_sum_upper', q3 - q1, offset=q3)
    filtered_frags = {frag for frag, _ in ranked[:cut]}
    logger.info('[rank sum filtering] {} fragments removed, {} fragments kept'.format(len(ranked) - len(filtered_frags), len(filtered_frags)))
    for frag, r in ranked[cut:]:
        logger.debug('[rank sum filtering] Fragment {} is removed, rank sum={}'.format(frag, r))
    if whitelisted:
        added = whitelisted - filtered_frags
        for frag in added:
            logger.debug('[rank sum filtering] Fragment {} is added since it is on the whitelist'.format(frag))
        filtered_frags |= whitelisted
        logger.info('[rank sum filtering] {} fragments added, {} fragments are used to perform Markov clustering'.format(
            len(added), len(filtered_frags)))
    return filtered_frags


# ------------------------------------------------------------------ a6: link weights (in-place dict rewrites)
def _frozen(container, kind=None):
    """the container is an array-backed table of containers.py that nobody has touched yet (and of that kind)"""
    return getattr(container, 'frozen', False) is True and (kind is None or container._kind == kind)


def _dict_arrays(link_dict, names):
    """(frag_i, frag_j, value) arrays of a link dict in dict order; names: fragment -> id, extended on the fly in first-seen order
    (i before j, key by key).  No Python statement per key: the keys are flattened, numbered and gathered by C-level iterators."""
    from itertools import chain
    if _frozen(link_dict):                                 # array-backed (containers.LinkTable): ids of the table -> ids of `names`
        i, j, v, id_names = link_dict.arrays()
        both = np.stack([i, j], axis=1).ravel()
        seen, first = np.unique(both, return_index=True)
        remap = np.zeros(len(id_names), np.int32)
        for t in seen[np.argsort(first, kind='stable')].tolist():
            remap[t] = names.setdefault(id_names[t], len(names))
        return remap[i], remap[j], np.asarray(v, np.float64)
    n = len(link_dict)
    flat = list(chain.from_iterable(link_dict))            # a0, b0, a1, b1, ...
    if len(flat) != 2 * n:
        raise ValueError('link dict keys must be pairs of fragment names')
    for name in dict.fromkeys(flat):                       # distinct names in first-seen order
        names.setdefault(name, len(names))
    ids = np.fromiter(map(names.__getitem__, flat), np.int32, 2 * n).reshape(n, 2)
    val = np.fromiter(link_dict.values(), np.float64, n)
    return np.ascontiguousarray(ids[:, 0]), np.ascontiguousarray(ids[:, 1]), val


def _store_values(link_dict, val):
    """dict[key] = float for every key, in order, without a Python statement per key"""
    dict.update(link_dict, zip(link_dict, val.tolist()))


def normalize_by_nlinks(flank_link_dict, frag_link_dict):
    """normalize_by_nlinks() :718-724 — every value divided by the geometric mean of its fragments' link totals"""
    logger.info('Normalizing flank_link_dict by the number of links to other contigs...')
    if _frozen(flank_link_dict, 'flank'):                  # the table is still in HBM: weigh it there
        session = flank_link_dict._session
        get = frag_link_dict.get
        totals = np.fromiter((get(f, 0) for f in session.table.frag_names), np.int64, session.table.n_frag)
        session.weigh_flank(0, per_frag=totals)
        return
    names = {}
    fi, fj, val = _dict_arrays(flank_link_dict, names)
    totals = np.fromiter((frag_link_dict[f] for f in names), np.int64, len(names))
    _lib.link_weights(fi, fj, val, 0, len(names), per_frag=totals)
    _store_values(flank_link_dict, val)


def normalize_by_length(flank_link_dict, frag_len_dict, flank):
    """normalize_by_length() :727-738 (the reference never calls it; here for completeness)"""
    logger.info('Normalizing flank_link_dict by length...')
    names = {}
    fi, fj, val = _dict_arrays(flank_link_dict, names)
    lengths = np.fromiter((frag_len_dict[f] for f in names), np.int64, len(names))
    _lib.link_weights(fi, fj, val, 1, len(names), per_frag=lengths, param=flank * 2000)
    _store_values(flank_link_dict, val)


def reduce_inter_hap_HiC_links(link_dict, read_depth_dict, phasing_weight, target='flank_link_dict'):
    """reduce_inter_hap_HiC_links() :695-707 — links between fragments of different haplotype tags lose phasing_weight
    of their value; entries that reach zero leave the dict"""
    logger.info('Reducing inter-haplotype Hi-C links in {}...'.format(target))
    names = {}
    fi, fj, val = _dict_arrays(link_dict, names)
    tags = {}
    tag = np.fromiter((tags.setdefault(read_depth_dict[f][0], len(tags)) for f in names), np.int32, len(names))
    n_zero = _lib.link_weights(fi, fj, val, 2, len(names), tag=tag, param=phasing_weight)
    changed = np.flatnonzero(tag[fi] != tag[fj])
    keys = list(link_dict)
    for k in changed.tolist():                            # untouched entries keep their Python type (int counts stay int)
        link_dict[keys[k]] = float(val[k])
    if n_zero:
        for k in np.flatnonzero((val == 0) & (tag[fi] != tag[fj])).tolist():
            del link_dict[keys[k]]


# ------------------------------------------------------------------ f3: reassign's per-group link sums
def parse_link_dict(link_dict, ctg_group_dict, normalize_by_nlinks=False, _original=None):
    """HapHiC_reassign.py parse_link_dict() :217-263.  Integer link counts without normalisation (the default of
    `haphic reassign`): the per-(contig, group) sums come from the device (hhx_group_link_sums), the inner dicts are
    rebuilt in the order their keys first received a contribution, linked_ctg_dict (sets) is built on the host.
    Anything else — normalize_by_nlinks, float links — is float arithmetic whose summation order is the dict order:
    handed back to the reference's own function (`_original`, bound by patch_reassign)."""
    integral = all(isinstance(v, (int, np.integer)) for v in link_dict.values())
    if normalize_by_nlinks or not integral:
        if _original is None:
            raise ValueError('parse_link_dict on float links / with normalize_by_nlinks stays the reference function')
        return _original(link_dict, ctg_group_dict, normalize_by_nlinks)
    names = {}
    fi, fj, val = _dict_arrays(link_dict, names)
    id_names = list(names)
    groups = {}
    grp = np.fromiter((-1 if ctg_group_dict[c] == 'ungrouped' else groups.setdefault(ctg_group_dict[c], len(groups)) for c in id_names),
                      np.int32, len(id_names))
    group_names = list(groups)
    # the device path holds two dense int64 [contigs x groups] tables: beyond a budget (or with nothing grouped) the reference's
    # O(keys) dict loop is the better tool
    if _original is not None and (not group_names or len(id_names) * len(group_names) > 50_000_000):
        return _original(link_dict, ctg_group_dict, normalize_by_nlinks)
    sums, first = _lib.group_link_sums(fi, fj, val.astype(np.int64), grp, len(group_names))
    ctg_group_link_dict = defaultdict(dict)
    linked_ctg_dict = defaultdict(set)
    rows, cols = np.nonzero(first >= 0)
    order = np.lexsort((first[rows, cols], rows))
    # outer dict order = order in which contigs were first touched by add_ctg_group (smallest `first` of the row)
    row_first = np.full(len(id_names), np.iinfo(np.int64).max, np.int64)
    np.minimum.at(row_first, rows, first[rows, cols])
    for r in np.argsort(row_first, kind='stable').tolist():
        if row_first[r] == np.iinfo(np.int64).max:
            break
        ctg_group_link_dict[id_names[r]] = {}
    for r, c in zip(rows[order].tolist(), cols[order].tolist()):
        ctg_group_link_dict[id_names[r]][group_names[c]] = int(sums[r, c])
    for a, b in zip(fi.tolist(), fj.tolist()):
        linked_ctg_dict[id_names[a]].add(id_names[b])
        linked_ctg_dict[id_names[b]].add(id_names[a])
    return ctg_group_link_dict, linked_ctg_dict


def group_link_dict(link_dict, ctg_group_dict, _original=None):
    """parse_link_dict() of HapHiC_cluster.py :2252-2268 (output_statistics :2279 calls it once per inflation on full_link_dict):
    ctg_group_link_dict[ctg][group] = summed links between the contig and the group's contigs, 'ungrouped' partners skipped; the outer
    dict in the order contigs first receive a contribution, each inner dict in the order its groups do (sorted(...) at :2362 is
    stable, so that order decides ties).  A full_link_dict that is still array-backed is summed from its arrays — one sort of the
    (contig, group) cells — instead of being turned into 10^8 Python tuples first; a real dict goes to the reference's loop."""
    if not (_frozen(link_dict) and link_dict._kind == 'full'):
        if _original is not None:
            return _original(link_dict, ctg_group_dict)
        out = defaultdict(dict)
        for (ctg_i, ctg_j), links in link_dict.items():
            for ctg, group in ((ctg_i, ctg_group_dict[ctg_j]), (ctg_j, ctg_group_dict[ctg_i])):
                if group != 'ungrouped':
                    out[ctg][group] = out[ctg].get(group, 0) + links
        return out
    i, j, v, names = link_dict.arrays()
    gid = {}
    grp = np.fromiter((-1 if g == 'ungrouped' else gid.setdefault(g, len(gid)) for g in map(ctg_group_dict.__getitem__, names)),
                      np.int64, len(names))
    group_names = list(gid)
    G = max(len(group_names), 1)
    # contributions in the reference's order: (ctg_i, group_j) at position 2k, (ctg_j, group_i) at 2k + 1
    ctg = np.stack([i, j], axis=1).ravel().astype(np.int64)
    other = np.stack([grp[j], grp[i]], axis=1).ravel()
    links = np.repeat(np.asarray(v, np.int64), 2)
    keep = other >= 0
    cell = ctg[keep] * G + other[keep]
    uniq, first, inverse = np.unique(cell, return_index=True, return_inverse=True)
    sums = np.bincount(inverse, weights=links[keep], minlength=len(uniq)).astype(np.int64)        # exact: link totals stay far below 2^53
    by_first = np.argsort(first, kind='stable')                      # cells in the order they first receive a contribution
    c_ctg, c_grp, c_sum = uniq[by_first] // G, uniq[by_first] % G, sums[by_first]
    # contigs in the order of their first cell; inside a contig the cells keep that order
    _u, ctg_first = np.unique(c_ctg, return_index=True)
    rank = np.empty(len(names), np.int64)
    rank[_u[np.argsort(ctg_first, kind='stable')]] = np.arange(len(_u))
    order = np.argsort(rank[c_ctg], kind='stable')
    c_ctg, c_grp, c_sum = c_ctg[order], c_grp[order], c_sum[order]
    bounds = np.flatnonzero(np.diff(c_ctg, prepend=-1)).tolist() + [len(c_ctg)]
    gnames = list(map(group_names.__getitem__, c_grp.tolist()))
    totals = c_sum.tolist()
    out = defaultdict(dict)
    for a, b in zip(bounds[:-1], bounds[1:]):
        out[names[c_ctg[a]]] = dict(zip(gnames[a:b], totals[a:b]))
    return out


# ------------------------------------------------------------------ S1 / S3: matrix-level seams
def _to_device(matrix):
    return _lib.DeviceCSR.from_scipy_csc(matrix)


def dot_product_mkl(matrix_a, matrix_b, **_ignored):
    """Drop-in for sparse_dot_mkl.dot_product_mkl on scipy CSC float32 operands (:39-43).
    out = A @ B; on the CSR(T) view that is T_B * T_A."""
    a, b = _to_device(matrix_a), _to_device(matrix_b)
    try:
        return _lib.spgemm(b, a).to_scipy_csc()
    finally:
        a.free()
        b.free()


def normalize(matrix, norm='l1', axis=0):
    """sklearn.preprocessing.normalize as the reference calls it (:2014 :2038 :2144): L1, axis=0, sparse."""
    if norm != 'l1' or axis != 0:
        raise ValueError('only normalize(norm="l1", axis=0) is on the hot path')
    m = _to_device(matrix)
    try:
        return _lib.normalize_l1(m).to_scipy_csc()
    finally:
        m.free()


def prune(matrix, pruning, dense_matrix=False):
    """prune() :1987-2014"""
    if dense_matrix:
        raise ValueError('dense_matrix mode is not on the MI355X path; use the reference function')
    m = _to_device(matrix)
    try:
        return _lib.prune(m, pruning).to_scipy_csc()
    finally:
        m.free()


def mkl_matrix_power(matrix, n):
    """mkl_matrix_power() :2017-2023 — M * M^(n-1), kept on the device between the products"""
    t = _to_device(matrix)
    run = t
    try:
        for _ in range(2, n + 1):
            nxt = _lib.spgemm(run, t)          # T^(e-1) * T  ==  (M * M^(e-1))^T
            if run is not t:
                run.free()
            run = nxt
        return run.to_scipy_csc()
    finally:
        if run is not t:
            run.free()
        t.free()


# ------------------------------------------------------------------ S2: mcl
def mcl_device(pre_expanded, expansion, inflation, iters, pruning, links=False):
    """mcl() on a device-resident pre-expanded matrix (links=True: on the raw link matrix, normalisation and
    pre-expansion fused into iteration 0); logs like the reference (:2047 :2058)."""
    res, n_iter, converged = _lib.mcl(pre_expanded, expansion, inflation, iters, pruning, links=links)
    _log_mcl(n_iter, converged, expansion, inflation, iters, pruning)
    return res


def _log_mcl(n_iter, converged, expansion, inflation, iters, pruning):
    if converged:
        logger.info('The matrix has converged after {} rounds of iterations '
                    '(expansion: {}, inflation: {}, maximum iterations: {}, pruning threshold: {})'.format(
                        n_iter, expansion, inflation, iters, pruning))
    else:
        logger.info('The matrix does not converge after {} rounds of iterations '
                    '(expansion: {}, inflation: {}, maximum iterations: {}, pruning threshold: {})'.format(
                        n_iter, expansion, inflation, iters, pruning))


_DENSE_WARM = None     # the helper thread that takes the sweep's dense block from the driver while the alignments are read (_prewarm_dense_block)


def _prewarm_dense_block(n):
    """The inflation sweep keeps M^2 as ONE float32 block of n x n (hhx_dense_layout: while that is at most a quarter of the device) and a first
    allocation of that size costs the caller's thread 1-3 s (25-75 ms per GB while the file-writer threads are busy with the driver: measured inside the
    whole C3 run, where it was the difference between the 6.7 s of the tail at inflation 1.1 and the 9-11 s of the first round of the sweep).  n is known when
    the alignments start to flow (the fragment table), the block is needed when the link matrix is ready: a helper thread takes it from the driver in between
    (hhx_pool_prewarm: it waits in the pool's cache, survives the trim after the ingest, and goes like any cached block when memory runs out).
    DenseSweep joins the thread before it asks for its block."""
    global _DENSE_WARM
    import threading
    if _DENSE_WARM is not None and _DENSE_WARM.is_alive():
        return
    torch = sys.modules.get('torch')             # several ranks: the sweep is shared out (sharded.sweep_sharded), no rank keeps the whole block — and ranks may share a device
    if torch is not None and torch.distributed.is_available() and torch.distributed.is_initialized() and torch.distributed.get_world_size() > 1:
        return
    free_bytes, total = _lib.mem_info()
    need = 4 * n * ((n + 31) // 32 * 32)
    if n < 20000 or need > total // 4 or free_bytes + _lib.pool_cached_bytes() < 0.75 * total:
        return                                   # small enough to cost nothing / the library would store the triangle or row blocks / the device is in use
    _DENSE_WARM = threading.Thread(target=_lib.pool_prewarm, args=([need],), daemon=True)
    _DENSE_WARM.start()


class DenseSweep:
    """The inflation sweep of run_mcl_clustering :2155-2158 with ONE expansion.  Every inflation restarts mcl() from the matrix
    pre-expanded at :2146-2147; M^2 of a link matrix is nearly dense, so its rows are kept in HBM as float32 row blocks
    (_lib.DenseRows, 4 B x n^2 in total — 40 GB at n = 100k) filled by a single pass over the products, and iteration 0 of each
    inflation (:2037-2042) is the row-local epilogue over those blocks.  When n^2 floats do not fit the budget the blocks are
    produced and consumed one after the other: the first inflations of every block are formed while it is resident and the
    pieces of every inflation wait (pruned: small) until the last block is done."""

    MAX_BLOCKS = 4  # run_mcl_clustering falls back to one fused expansion per inflation beyond this many row blocks
    GROUP = 5       # inflations per pass over the dense block (their candidate pools live side by side: ~2-4 GB each at the low inflations of C3)

    def __init__(self, links, pruning, block_rows=None, budget_bytes=None):
        self.links, self.pruning = links, pruning
        n = links.shape3[0]
        self.n = n
        self._budget_rows = None
        if block_rows is None:
            if budget_bytes is None:
                # what the driver reports free + what the library's pool holds for reuse (no trim: a cached 40 GB block is worth
                # seconds of hipMalloc); the rest: candidate pools, operand stream, the tails of the inflations
                free_bytes, _total = _lib.mem_info()
                budget_bytes = 0.45 * (free_bytes + _lib.pool_cached_bytes())
            block_rows = max(1, min(n, int(budget_bytes // (4 * max(n, 1)))))
            self._budget_rows = block_rows
            if block_rows < n and _lib.links_plan(links)[1] == 2:
                block_rows = n                               # all rows in one block after all: the library stores it as the upper block triangle (half the bytes)
        self.bounds = list(range(0, n, int(block_rows))) + [n]
        self.stage_s = {}                                    # measurement: seconds the caller spent in the sweep's own steps (SWEEP_STAGES)
        self.resident = None                                 # the only block when the whole M^2 fits: kept for the whole sweep
        self.n_products = 0
        self._gen = None

    def first_iterations(self, inflations):
        """iteration 0 of mcl() at every inflation -> one matrix per inflation, as a generator"""
        self._gen = self._first_iterations(inflations)
        return self._gen

    def _first_iterations(self, inflations):
        inflations = [float(x) for x in inflations]
        warm = None
        if len(self.bounds) == 2 and self.resident is None:
            t_0 = time.perf_counter()
            if _DENSE_WARM is not None:
                _DENSE_WARM.join()                # the block it took is in the pool's cache now
            self.stage_s['wait_for_the_block_taken_ahead'] = time.perf_counter() - t_0
            warm = self._prewarm(min(self.GROUP, len(inflations)))
            try:
                t_0 = time.perf_counter()
                self.resident = _lib.DenseRows(self.links, 0, self.n)
                self.stage_s['expansion_call'] = time.perf_counter() - t_0
                self.n_products = self.resident.n_products
            except RuntimeError:
                if warm is not None:
                    warm.join()
                # the one block did not fit after all (free memory moved since the plan, or the library chose the square where the
                # triangle was planned): back to row blocks within the budget
                if not self._budget_rows or self._budget_rows >= self.n:
                    raise
                _lib.load().hhx_pool_trim()
                self.bounds = list(range(0, self.n, int(self._budget_rows))) + [self.n]
        self._warm = warm if len(self.bounds) == 2 and self.resident is not None else None
        if len(self.bounds) == 2:
            # GROUP inflations at a time in one pass over the block (hhx_dense_inflate_prune_multi: the division and the log2 of
            # x^r = exp2(r log2 x) once per entry, the 4 n^2 bytes read once per group); the matrices of a group wait their turn
            for lo in range(0, len(inflations), self.GROUP):
                t_0 = time.perf_counter()
                ready = self.resident.inflate_prune_multi(inflations[lo:lo + self.GROUP], self.pruning)
                self.stage_s.setdefault('epilogue_calls', []).append(time.perf_counter() - t_0)
                try:
                    while ready:
                        yield ready.pop(0)
                finally:
                    for m in ready:                           # the consumer stopped early
                        m.free()
            return
        pieces = [[] for _ in inflations]
        for r0, r1 in zip(self.bounds[:-1], self.bounds[1:]):
            blk = _lib.DenseRows(self.links, r0, r1)
            self.n_products += blk.n_products
            try:
                for lo in range(0, len(inflations), self.GROUP):
                    for k, piece in enumerate(blk.inflate_prune_multi(inflations[lo:lo + self.GROUP], self.pruning)):
                        pieces[lo + k].append(piece)
            finally:
                blk.free()
        for k in range(len(inflations)):
            try:
                yield _lib.vstack(pieces[k])
            finally:
                for p in pieces[k]:
                    p.free()

    def _prewarm(self, group):
        """the candidate / survivor pools and the packed outputs of the first group of inflations, taken from the driver by a helper thread WHILE the
        expansion runs (0.3 s at 100k contigs; fresh device memory costs 12-30 ms per GB): sizes as the library's own first guess
        (hhx_expand_dense_impl: 0.6 survivors and 1.0 candidates per entry of the link matrix).  Only when the device has room to spare."""
        import threading
        nnz, n = self.links.nnz, self.n
        out_b, cand_b = int(4 * (0.6 * nnz + n)), int(4 * (1.0 * nnz + n))
        # the first group's pools, its packed outputs, then the (smaller) outputs of the groups after it: the thread keeps going while the first
        # passes over the block run, and is joined when the sweep's first iterations are done
        # ... and the candidate pools of the first tail's first iteration (hhx_mcl_resume: 3 candidates per entry of T1, + 50 %; T1 at the lowest inflation holds
        # ~0.55 of the link matrix's entries): two blocks of ~3 GB at C3 that cost the caller 0.1-0.6 s when it has to take them itself while the file writers free theirs
        tail_cand = int(4 * 1.5 * 3 * 0.55 * nnz)
        sizes = ([out_b, out_b, cand_b, cand_b] * group) + [tail_cand, tail_cand] + [out_b, out_b] * group + [out_b // 2] * (2 * group) + [out_b // 4] * (4 * group)
        free_bytes, _total = _lib.mem_info()
        if group < 2 or (free_bytes + _lib.pool_cached_bytes()) < 2 * (sum(sizes) + 4 * n * n):
            return None
        t = threading.Thread(target=_lib.pool_prewarm, args=(sizes,), daemon=True)
        t.start()
        return t

    def close(self):
        warm, self._warm = getattr(self, '_warm', None), None
        if warm is not None:
            warm.join()
        if self._gen is not None:                            # a suspended generator holds the matrices of its group: release them now
            self._gen.close()
            self._gen = None
        if self.resident is not None:
            self.resident.free()
            self.resident = None


def mcl_resume_device(first, expansion, inflation, iters, pruning):
    """mcl() :2026-2062 picked up after iteration 0 (`first`, consumed); logs like the reference (:2047 :2058)"""
    if iters <= 1:
        _log_mcl(min(iters, 1), False, expansion, inflation, iters, pruning)
        return first
    try:
        res, n_iter, converged = _lib.mcl_resume(first, 1, expansion, inflation, iters, pruning)
    finally:
        first.free()
    _log_mcl(n_iter, converged, expansion, inflation, iters, pruning)
    return res


def mcl(matrix, expansion, inflation, iters, pruning, dense_matrix=False):
    """mcl() :2026-2062 — scipy CSC in, scipy CSC out; every iteration stays in HBM."""
    if dense_matrix:
        raise ValueError('dense_matrix mode is not on the MI355X path; use the reference function')
    pre = _to_device(matrix)
    try:
        res = mcl_device(pre, expansion, inflation, iters, pruning)
        try:
            return res.to_scipy_csc()
        finally:
            res.free()
    finally:
        pre.free()


def _clusters_from_arrays(att, att_ptr, members, shape):
    # :2073-2095 — the SET of tuples (its iteration order decides group numbering on length ties)
    ptr, mem = np.asarray(att_ptr).tolist(), np.asarray(members).tolist()
    clusters = set()
    for a in range(len(att)):
        clusters.add(tuple(mem[ptr[a]:ptr[a + 1]]))
    if len(clusters) == len(att) and len(att):
        # no two attractors with the same members: "every node in exactly one cluster, all nodes present" (:2086-2093) is a count per node
        used = np.asarray(members[:ptr[len(att)]], np.int64)
        if len(used) != shape or len(np.unique(used)) != len(used):
            return None
        return list(clusters)
    nodes = set()
    for cluster in clusters:
        for node in cluster:
            if node in nodes:
                return None
            nodes.add(node)
    if len(nodes) != shape:
        return None
    return list(clusters)


def _groups_from_arrays(clusters, names_by_index, len_by_index, line_by_index):
    """:2172-2218 for clusters of whole contigs: every cluster's contigs in ascending matrix-index order (:2172-2187), its total length, the
    clusters sorted by total length (descending, stable: :2197), the contigs of each by length (descending, stable: :2208).  Returns
    (result_clusters as the reference builds it — [[contigs], total length] per group —, the body of each group's file)."""
    clusters = [c for c in clusters if c]                                   # an empty cluster never touches `groups` (:2172): no group
    sizes = np.fromiter(map(len, clusters), np.int64, len(clusters))
    flat = np.fromiter((i for c in clusters for i in c), np.int64, int(sizes.sum()))
    start = np.concatenate([[0], np.cumsum(sizes)])
    lens = len_by_index[flat]
    total = np.add.reduceat(lens, start[:-1]) if len(flat) else np.zeros(0, np.int64)
    rank = np.argsort(-total, kind='stable')                                # position in result_clusters -> cluster
    place = np.empty(len(clusters), np.int64)
    place[rank] = np.arange(len(clusters))
    group_of = np.repeat(place, sizes)                                      # where every entry of `flat` ends up
    order = np.lexsort((-lens, group_of))                                   # by group, then length descending; stable in the original (cluster) order
    flat = flat[order]
    bounds = np.concatenate([[0], np.cumsum(sizes[rank])]).tolist()
    names, lines, totals = names_by_index[flat].tolist(), line_by_index[flat].tolist(), total[rank].tolist()
    result = [[names[a:b], t] for a, b, t in zip(bounds[:-1], bounds[1:], totals)]
    return result, [''.join(lines[a:b]) for a, b in zip(bounds[:-1], bounds[1:])]


def interpret_result_device(result):
    att, att_ptr, members = _lib.interpret(result)
    return _clusters_from_arrays(att, att_ptr, members, result.shape3[0])


def interpret_result(result_matrix, dense_matrix=False):
    """interpret_result() :2065-2095"""
    if dense_matrix:
        raise ValueError('dense_matrix mode is not on the MI355X path; use the reference function')
    m = _to_device(result_matrix)
    try:
        return interpret_result_device(m)
    finally:
        m.free()


# ------------------------------------------------------------------ S4: dict_to_matrix
class ResidentMatrix:
    """What dict_to_matrix returns for a link table that never left the device: the CSR(T) handle (== the reference's CSC(M)
    triple) for run_mcl_clustering, which takes it over without a host round trip.  Anything else that is asked of it is
    answered by the scipy CSC matrix the reference would have got (:368), downloaded on first use."""

    def __init__(self, dev):
        self._dev = dev
        self._csc = None

    def take_device(self):
        dev, self._dev = self._dev, None
        if dev is None:
            raise RuntimeError('the device matrix has already been handed over')
        return dev

    def _scipy(self):
        if self._csc is None:
            if self._dev is None:
                raise RuntimeError('the device matrix has already been handed over')
            self._csc = self._dev.to_scipy_csc()
        return self._csc

    def __getattr__(self, name):
        if name in ('_dev', '_csc') or (name.startswith('__') and name.endswith('__')):
            raise AttributeError(name)
        return getattr(self._scipy(), name)

    def __del__(self):
        try:
            if self._dev is not None:
                self._dev.free()
        except Exception:
            pass


def _index_dict(id_names, fidx, n_linked, rest):
    """frag_index_dict: linked fragments in index order (== insertion order :337-349), then the link-less ones (:357-359)"""
    linked = np.flatnonzero(fidx >= 0)
    linked = linked[np.argsort(fidx[linked], kind='stable')]
    frag_index_dict = dict(zip(map(id_names.__getitem__, linked.tolist()), fidx[linked].tolist()))
    frag_index_dict.update(zip(rest, range(n_linked, n_linked + len(rest))))
    return frag_index_dict


def dict_to_matrix(link_dict, frag_set, dense_matrix=True, add_self_loops=False, _device=False):
    """dict_to_matrix() :310-373.  Returns (matrix, frag_index_dict); the matrix is scipy CSC (an ndarray with dense_matrix)
    unless _device=True (then a DeviceCSR).  A flank table that is still frozen (containers.LinkTable: run() has only passed it
    through the seams) is turned into the matrix where it lies, in HBM (hhx_ingest_link_matrix), and comes back as a
    ResidentMatrix."""
    if _frozen(link_dict, 'flank'):
        session = link_dict._session
        id_names = session.table.frag_names
        in_set = np.fromiter(map(frag_set.__contains__, id_names), np.uint8, len(id_names))
        if int(in_set.sum()) == len(frag_set):               # every member is a fragment of the table (else: the generic path)
            m, fidx, n_linked = session.link_matrix(in_set, -1, add_self_loops)
            frags_in_dict = set(map(id_names.__getitem__, np.flatnonzero(fidx >= 0).tolist()))
            rest = frag_set - frags_in_dict                  # :357 — CPython set order, kept in Python
            frag_index_dict = _index_dict(id_names, fidx, n_linked, rest)
            if _device:
                return m, frag_index_dict
            if dense_matrix:
                try:
                    return m.to_scipy_csc().toarray(), frag_index_dict
                finally:
                    m.free()
            return ResidentMatrix(m), frag_index_dict
    names = {}
    ids_i, ids_j, vals = _dict_arrays(link_dict, names)
    # The device builder writes ONE slot per (row, column).  The reference's own parsers only ever produce sorted, unique
    # name pairs, but the seam is public: a dict holding both (a, b) and (b, a) is folded into the first of the two
    # (coo_matrix(...).tocsc() :368 sums duplicates — same matrix, same first-seen index order); a key (f, f) would land on
    # the diagonal twice, next to the self loop — nothing in the reference builds one, so it is refused loudly.
    if len(ids_i):
        if (ids_i == ids_j).any():
            bad = next(k for k in link_dict if k[0] == k[1])
            raise ValueError('dict_to_matrix: key {!r} links a fragment with itself'.format(bad))
        lo, hi = np.minimum(ids_i, ids_j).astype(np.int64), np.maximum(ids_i, ids_j).astype(np.int64)
        und = lo * (len(names) + 1) + hi
        uniq, first_at, inverse = np.unique(und, return_index=True, return_inverse=True)
        if len(uniq) != len(und):
            summed = np.bincount(inverse, weights=vals, minlength=len(uniq))
            keep = np.sort(first_at)                             # first occurrence of every unordered pair, in dict order
            ids_i, ids_j, vals = ids_i[keep], ids_j[keep], summed[inverse[keep]]
    for f in frag_set:
        names.setdefault(f, len(names))
    id_names = list(names)
    in_set = np.fromiter(map(frag_set.__contains__, id_names), np.uint8, len(id_names))
    ok = in_set[ids_i].astype(bool) & in_set[ids_j].astype(bool)
    linked = np.zeros(len(id_names), bool)
    linked[ids_i[ok]] = True
    linked[ids_j[ok]] = True
    frags_in_dict = set(map(id_names.__getitem__, np.flatnonzero(linked).tolist()))
    rest = frag_set - frags_in_dict                      # :357 — CPython set order, kept in Python
    m, fidx, n_linked = _lib.dict_to_matrix(ids_i, ids_j, vals, max(len(id_names), 1), in_set if len(in_set) else
                                            np.zeros(1, np.uint8), len(rest), add_self_loops=add_self_loops)
    fidx = np.where(linked, fidx[:len(id_names)], -1) if len(id_names) else fidx[:0]
    frag_index_dict = _index_dict(id_names, fidx, n_linked, rest)
    if _device:
        return m, frag_index_dict
    try:
        csc = m.to_scipy_csc()
    finally:
        m.free()
    return (csc.toarray() if dense_matrix else csc), frag_index_dict


# ------------------------------------------------------------------ f3 / f2: the files run() writes from the containers
def output_pickle(dict_, from_, to, _original=None):
    """output_pickle() :710-715.  A link table that is still frozen (full_links.pkl, HT_links.pkl) is serialised from its arrays
    by the library's host code — the same `defaultdict(int)` pickle (protocol 4, keys in dict order, int values) without a Python
    object per key — on the library's file-writer thread: the call returns once the file is open and queued, the file is complete
    after `_lib.files_join()` (run() re-bound by patch_reference joins before it returns; HAPHIC_SYNC_FILES=1 writes it here).  Any
    other object is pickled as the reference does."""
    import pickle
    logger.info('Writing {} to {}...'.format(from_, to))
    if _frozen(dict_) and dict_._kind in ('full', 'HT', 'flank'):
        session = dict_._session
        if not (dict_._kind == 'flank' and session.weighted):            # float weights: the generic pickle below
            if _lib.files_async() and session.queue_pickle(dict_._kind, to):
                return
            i, j, v, names = dict_.arrays()
            if v.dtype.kind in 'iu':
                _lib.write_link_pickle(to, i, j, v, names)
                return
    with open(to, 'wb') as fpkl:
        pickle.dump(dict_, fpkl)


def output_clm(clm_dict, _original=None):
    """output_clm() :376-392 — paired_links.clm.  The frozen clm_dict of the S5 mirrors is written from the read pairs kept
    in HBM (grouping, the per-orientation sorts and the text on the device: hhx_ingest_write_clm), on the library's file-writer
    thread like the pickles above; a real dict takes the reference's loop."""
    if _frozen(clm_dict, 'clm'):
        try:
            clm_dict._session.write_clm('paired_links.clm')
            logger.info('Writing clm_dict to paired_links.clm...')
            return
        except RuntimeError as e:
            # the device writer refuses streams with read positions beyond their contig's end (the reference would print negative
            # distances), and may run out of device memory: such a clm_dict takes the reference's loop, like any real dict (the
            # access below thaws it).  A file system that refuses the file would refuse the loop's too.
            if 'cannot open' in str(e):
                raise
            logger.warning('paired_links.clm is written by the host loop: {}'.format(e))
    if _original is not None:
        return _original(clm_dict)
    logger.info('Writing clm_dict to paired_links.clm...')
    signs = ('++', '+-', '-+', '--')
    with open('paired_links.clm', 'w') as fout:
        for (ctg_i, ctg_j), dists in clm_dict.items():
            if len(dists) < 8:                               # fewer than two read pairs (:385)
                continue
            for n, (si, sj) in enumerate(signs):
                ordered = sorted(dists[n::4])
                fout.write('{}{} {}{}\t{}\t{}\n'.format(ctg_i, si, ctg_j, sj, 2 * len(ordered), ' '.join('{0} {0}'.format(d) for d in ordered)))


# ------------------------------------------------------------------ a1: .pairs text
def _pwrite_all(fd, view, offset):
    while len(view):
        k = os.pwrite(fd, view, offset)
        view, offset = view[k:], offset + k


class PairsText:
    """What pairs_generator / pairs_generator_inter_ctgs (:1539-1583) return here: the .pairs file, tokenised on the
    device one chunk of whole lines at a time (hhx_pairs_parse).  parse_alignments* take it as is and push the
    device arrays straight into the ingest; iterating it yields the reference's (ref, mref, pos, mpos) tuples for
    any other consumer.  alignments.bed is written in the working directory as the reference does (:1549)."""

    def __init__(self, pairs, aln_format, inter_only, chunk_bytes=int(os.environ.get('HAPHIC_TEXT_CHUNK_MB', '256')) << 20, bed_path='alignments.bed', bed_writers=8):
        assert aln_format in ('pairs', 'bgzipped_pairs')
        self.path, self.aln_format, self.inter_only = pairs, aln_format, inter_only
        self.chunk_bytes, self.bed_path, self.bed_writers = chunk_bytes, bed_path, bed_writers

    def _chunks(self):
        """byte chunks holding whole lines (cut after the last '\n'; the tail of the file goes as it is).  A plain
        .pairs file is memory-mapped and handed to the device copy without passing through Python bytes objects."""
        if self.aln_format == 'pairs':
            import mmap
            size = os.path.getsize(self.path)
            if size == 0:
                return
            f = open(self.path, 'rb')
            mm = mmap.mmap(f.fileno(), 0, access=mmap.ACCESS_READ)
            try:
                at = 0
                while at < size:
                    end = min(at + self.chunk_bytes, size)
                    cut = end if end == size else mm.rfind(b'\n', at, end) + 1
                    while cut <= at:                             # a line longer than the chunk
                        end = min(end + self.chunk_bytes, size)
                        cut = end if end == size else mm.rfind(b'\n', at, end) + 1
                    view = np.frombuffer(mm, np.uint8, cut - at, at)
                    yield view
                    del view
                    at = cut
            finally:
                try:
                    mm.close()
                except BufferError:                              # a consumer still holds a view of a chunk: the map goes
                    pass                                         # when that view does
                f.close()
            return
        import gzip
        with gzip.open(self.path, 'rb') as f:
            carry = b''
            while True:
                block = f.read(self.chunk_bytes)
                if not block:
                    break
                cut = block.rfind(b'\n') + 1
                if cut == 0:
                    carry += block
                    continue
                yield carry + block[:cut]
                carry = block[cut:]
            if carry:
                yield carry

    def batches(self, names, wide=False):
        """per chunk: (parser, n_lines) with the id / position arrays of the chunk on the device (wide: int64 positions).  The reference writes
        alignments.bed inside its generator loop (:1549-1557) and nothing in run() reads it: the BED bytes of a chunk stay in HBM, where the kernel
        formatted them, and are handed to the library's file-writer thread (hhx_byte_sink: the file is complete after _lib.files_join(), which the
        re-bound run() calls) — a RAM disk takes ~4.5 GB/s into one file, the tokeniser makes 19 GB/s of BED.  HAPHIC_SYNC_FILES=1: the bytes
        come back through the parser's pinned double buffer and are written here by a pool of pwrite() threads while the next chunk is parsed."""
        clock = time.perf_counter
        st = self.stats = {'chunks': 0, 'text_bytes': 0, 'bed_bytes': 0, 'parse_s': 0.0, 'bed_fetch_s': 0.0, 'bed_wait_s': 0.0, 'consumer_s': 0.0,
                           'bed': 'deferred' if _lib.files_async() else 'in place'}
        parser = _lib.PairsParser(names)
        if wide:
            parser.set_wide(True)
        reader = None
        if _lib.files_async():
            # the native front end: the file read ahead into pinned memory by threads of the library (hhx_text_reader), alignments.bed deferred.
            # bgzipped .pairs: the BGZF blocks are inflated by those threads; a plain gzip stream (no block boundaries) keeps Python's gzip below
            threads = int(os.environ.get('HAPHIC_READ_THREADS', '8'))
            try:
                reader = _lib.TextReader(self.path, self.chunk_bytes, threads=threads, bgzf=self.aln_format != 'pairs')
            except RuntimeError as e:
                if self.aln_format == 'pairs' or 'not a BGZF file' not in str(e):
                    raise
        if reader is not None:
            size = os.path.getsize(self.path) * (1 if self.aln_format == 'pairs' else 4)                          # (text deflates ~4 x)
            sink = _lib.ByteSink(self.bed_path, expected_bytes=int(1.45 * size)) if self.bed_path else None       # two BED records ~ 1.35 x the line
            try:
                if sink is not None:
                    parser.set_bed_sink(sink)
                t = clock()
                for host, nbytes in reader:
                    st['read_wait_s'] = st.get('read_wait_s', 0.0) + clock() - t
                    t = clock()
                    st['chunks'] += 1
                    st['text_bytes'] += nbytes
                    n = parser.parse(None, want_bed=sink is not None, host_ptr=host, n_bytes=nbytes)
                    st['bed_bytes'] += parser.bed_bytes
                    st['parse_s'] += clock() - t
                    t = clock()
                    yield parser, n
                    st['consumer_s'] += clock() - t
                    t = clock()
            finally:
                if sink is not None:
                    parser.set_bed_sink(None)
                    sink.close()
                reader.close()
                parser.destroy()
            return
        from concurrent.futures import ThreadPoolExecutor, wait
        st['bed'] = 'in place'
        fd = os.open(self.bed_path, os.O_WRONLY | os.O_CREAT | os.O_TRUNC, 0o644) if self.bed_path else None
        pool = ThreadPoolExecutor(self.bed_writers) if fd is not None else None
        pending = []                                             # [futures of chunk k - 1, futures of chunk k]
        offset = 0
        piece = 8 << 20

        def drain(keep):
            t = clock()
            while len(pending) > keep:
                for f in wait(pending.pop(0)).done:
                    f.result()                                   # a failed write surfaces on the caller's thread
            st['bed_wait_s'] += clock() - t
        try:
            for chunk in self._chunks():
                t = clock()
                st['chunks'] += 1
                st['text_bytes'] += len(chunk)
                n = parser.parse(chunk, want_bed=fd is not None)
                st['parse_s'] += clock() - t
                del chunk
                if fd is not None and parser.bed_bytes:
                    drain(1)                                     # the buffer handed out two calls ago is free again
                    t = clock()
                    buf = memoryview(parser.bed_host())
                    st['bed_fetch_s'] += clock() - t
                    st['bed_bytes'] += len(buf)
                    pending.append([pool.submit(_pwrite_all, fd, buf[a:a + piece], offset + a) for a in range(0, len(buf), piece)])
                    offset += len(buf)
                t = clock()
                yield parser, n
                st['consumer_s'] += clock() - t
            drain(0)
        finally:
            if pool is not None:
                pool.shutdown(wait=True)
            if fd is not None:
                os.close(fd)
            parser.destroy()

    def __iter__(self):
        # name tuples would mean tokenising on the host; patch_reference binds these generators only together with S5
        raise TypeError('PairsText is consumed by haphic_amd.cluster.parse_alignments / parse_alignments_for_ctgs; '
                        'patch_reference(H, ingest=False) keeps the reference generators')


def pairs_generator(pairs, aln_format):
    """pairs_generator() :1539-1559"""
    return PairsText(pairs, aln_format, inter_only=False)


def pairs_generator_inter_ctgs(pairs, aln_format):
    """pairs_generator_inter_ctgs() :1562-1583"""
    return PairsText(pairs, aln_format, inter_only=True)


# ------------------------------------------------------------------ f4: BAM
def check_sorting_order(header_text):
    """check_sorting_order() :1347-1359 on the SAM header text: coordinate-sorted input is an error"""
    order = None
    for line in header_text.splitlines():
        if line.startswith('@HD'):
            for field in line.split('\t')[1:]:
                if field.startswith('SO:'):
                    order = field[3:]
    if order in ('unsorted', 'queryname'):
        logger.info('The sorting order of the BAM file is {}'.format(order))
        return
    if order == 'coordinate':
        logger.error('The sorting order of the BAM file is {}. It should be unsorted or name-sorted'.format(order))
        raise RuntimeError('The sorting order of the BAM file is {}. It should be unsorted or name-sorted'.format(order))
    logger.warning('The sorting order of the BAM file is unknown, but the program will continue')


class BamRecords:
    """What bam_generator (:1586-1593) returns here: the BAM file, inflated by host threads and decoded on the device one
    batch of records at a time (hhx_bam_next).  parse_alignments* push the device arrays straight into the ingest;
    iterating it yields the reference's (ref, mref, pos, mpos) tuples for any other consumer.  Only the two htslib
    filter expressions the reference uses are understood: 'flag.read1' and 'flag.read1 && refid != mrefid'."""

    def __init__(self, bam, threads, format_options, batch_bytes=256 << 20):
        self.path, self.threads, self.batch_bytes = bam, threads, batch_bytes
        self.need_flags, self.drop_same_ref = 0, False
        for opt in format_options or ():
            text = opt.decode() if isinstance(opt, bytes) else str(opt)
            if not text.startswith('filter='):
                raise NotImplementedError('BAM format option {!r} is not on the MI355X path'.format(text))
            for term in (t.strip() for t in text[len('filter='):].split('&&')):
                if term == 'flag.read1':
                    self.need_flags |= 0x40
                elif term.replace(' ', '') == 'refid!=mrefid':
                    self.drop_same_ref = True
                else:
                    raise NotImplementedError('BAM filter term {!r} is not on the MI355X path'.format(term))
        self.inter_only = self.drop_same_ref

    def batches(self, names):
        """per batch: (reader, n_records, [id1, pos1, id2, pos2] device pointers)"""
        reader = _lib.BamReader(self.path, self.threads)
        try:
            check_sorting_order(reader.header_text)
            reader.set_contigs({n: i for i, n in enumerate(names)})
            while True:
                n, ptrs = reader.next_batch(self.need_flags, self.drop_same_ref, self.batch_bytes)
                if n == 0:
                    break
                yield reader, n, ptrs
        finally:
            reader.close()

    def __iter__(self):
        reader = _lib.BamReader(self.path, self.threads)
        try:
            check_sorting_order(reader.header_text)
            names = reader.ref_names
            reader.set_contigs({n: i for i, n in enumerate(names)})        # identity: ids are BAM reference ids
            while True:
                n, _ptrs = reader.next_batch(self.need_flags, self.drop_same_ref, self.batch_bytes)
                if n == 0:
                    break
                id1, p1, id2, p2 = reader.fetch()
                for a, x, b, y in zip(id1.tolist(), p1.tolist(), id2.tolist(), p2.tolist()):
                    if a != -2:                                  # -2: the record failed the filter, htslib never yields it
                        yield (names[a] if a >= 0 else None, names[b] if b >= 0 else None, x, y)
        finally:
            reader.close()


def bam_generator(bam, threads, format_options):
    """bam_generator() :1586-1593"""
    return BamRecords(bam, threads, format_options)


# ------------------------------------------------------------------ S5: ingest
class IdArrays:
    """Alignments that are already integer arrays — contig ids in fa_dict order (-1: a name that is not in the FASTA) and 0-based
    positions — accepted wherever the S5 mirrors take the reference's (ref, mref, pos, mpos) iterator (:1539-1593); iterating
    yields those tuples.  The arrays go to the device ingest as they are, without a pass through Python objects."""

    inter_only = False          # True: the stream stands for pairs_generator_inter_ctgs :1562-1583 (ref == mref is dropped on the device)

    def __init__(self, names, id1, pos1, id2, pos2):
        self.names = list(names)
        self.id1, self.id2 = np.ascontiguousarray(id1, np.int32), np.ascontiguousarray(id2, np.int32)
        self.pos1, self.pos2 = np.ascontiguousarray(pos1), np.ascontiguousarray(pos2)
        if not (len(self.id1) == len(self.id2) == len(self.pos1) == len(self.pos2)):
            raise ValueError('IdArrays: arrays of different lengths')

    def __len__(self):
        return len(self.id1)

    def __iter__(self):
        nm = self.names
        for a, x, b, y in zip(self.id1.tolist(), self.pos1.tolist(), self.id2.tolist(), self.pos2.tolist()):
            yield (nm[a] if a >= 0 else None, nm[b] if b >= 0 else None, x, y)


def _ids_from_alignments(alignments, cid, chunk, wide=False):
    """(ref, mref, pos, mpos) iterator -> id (int32) and position (int32; int64 when wide) arrays, `chunk` pairs at a time"""
    pos_t = np.int64 if wide else np.int32
    b1, b2 = np.empty(chunk, np.int32), np.empty(chunk, np.int32)
    p1, p2 = np.empty(chunk, pos_t), np.empty(chunk, pos_t)
    k = 0
    get = cid.get
    limit = np.iinfo(pos_t).max
    for ref, mref, pos, mpos in alignments:
        if pos >= limit or mpos >= limit:        # int32 positions: every contig is shorter than 2^31 bp (FragTable.wide says otherwise)
            raise RuntimeError('position {} does not fit the {} coordinates of the MI355X ingest'.format(max(pos, mpos), pos_t.__name__))
        b1[k] = get(ref, -1)
        b2[k] = get(mref, -1)
        p1[k] = pos
        p2[k] = mpos
        k += 1
        if k == chunk:
            yield b1, p1, b2, p2, k
            k = 0
    if k:
        yield b1[:k], p1[:k], b2[:k], p2[:k], k


def _ingest_handle(alignments, table, flank, bins, chunk=1 << 22, want_pairs=False, want_frag_pairs=False, sweep_follows=False):
    """Feed an alignment iterator (name tuples, as the reference's generators :1539-1593 yield them, or the device-side front
    ends of this package) through the device ingest; returns the finalized handle (_lib.Ingest), tables resident in HBM."""
    text = isinstance(alignments, PairsText)
    bam = isinstance(alignments, BamRecords)
    ing = _lib.Ingest(table, flank, bins=bins, skip_intra=bool(getattr(alignments, 'inter_only', False)))      # :1582 / refid != mrefid
    try:
        if want_pairs:
            ing.keep_pairs()
        if want_frag_pairs:
            ing.keep_frag_pairs()
        if sweep_follows and (text or bam) and not bins:
            _prewarm_dense_block(table.n_frag)   # run() goes on to run_mcl_clustering: its dense block is taken from the driver while the file is read
        if text:                             # a1 on the device: text chunk -> id arrays -> ingest, nothing returns to the host
            for parser, k in alignments.batches(table.ctg_names, wide=table.wide):
                if k:
                    ing.push_device(k, *parser.device_arrays()[:4], wide=table.wide)
        elif bam:                            # f4: BGZF inflate on host threads, record decode on the device
            for _reader, k, ptrs in alignments.batches(table.ctg_names):
                ing.push_device(k, *ptrs)
        elif isinstance(alignments, IdArrays):
            if alignments.names != list(table.ctg_names):
                raise ValueError('IdArrays: the ids do not refer to the contigs of fa_dict, in its order')
            pos_t = np.int64 if table.wide else np.int32
            if len(alignments) and max(int(alignments.pos1.max()), int(alignments.pos2.max())) >= np.iinfo(pos_t).max:
                raise RuntimeError('a position does not fit the {} coordinates of the MI355X ingest'.format(pos_t.__name__))
            step = max(chunk, 1 << 28)       # whole arrays are at hand: few, large pushes (every push is one aggregated run to merge)
            for lo in range(0, len(alignments), step):
                hi = min(len(alignments), lo + step)
                ing.push(alignments.id1[lo:hi], alignments.pos1[lo:hi].astype(pos_t, copy=False), alignments.id2[lo:hi],
                         alignments.pos2[lo:hi].astype(pos_t, copy=False), wide=table.wide)
        else:
            cid = {n: i for i, n in enumerate(table.ctg_names)}
            for b1, p1, b2, p2, k in _ids_from_alignments(alignments, cid, chunk, wide=table.wide):
                ing.push(b1, p1, b2, p2, wide=table.wide)     # unknown names (-1) and intra-contig pairs are filtered on the device
        t_fin = time.perf_counter()
        ing.finalize()
        if text or bam:
            # the front end's transient blocks (chunk buffers, partition scratch: ~90 GB at 5e8 pairs) sit in the pool's cache, where nothing
            # that follows is of their sizes (lending them to the file-writer thread was tried: its sorts want 4 GB blocks, these are smaller, and
            # it ended up holding both); the device is idle here: back to the driver, before the sweep and the file-writer threads want the room
            # ... except, when the sweep follows, 24 GB of its mid-size blocks: the operand stream and the first pools of the sweep take them instead of fresh ones,
            # which cost the caller 1-1.5 s while the file-writer threads are releasing their own memory (tools/c3_run.py: sweep_stages_s)
            _lib.check(_lib.load().hhx_pool_trim_keep(24 << 30) if sweep_follows else _lib.load().hhx_pool_trim())
        if hasattr(alignments, 'stats'):
            _lib.check(_lib.load().hhx_synchronize())
            alignments.stats['finalize_s'] = time.perf_counter() - t_fin
        return ing
    except BaseException:
        ing.destroy()
        raise


def ingest_links(alignments, table, flank, bins, chunk=1 << 22, want_pairs=False, max_read_pairs=0, want_frag_pairs=False):
    """The same, fetched: the insertion-ordered tables as numpy arrays (+ the CLM distances and the first coordinates of every
    contig pair when want_pairs); the handle is gone when this returns."""
    ing = _ingest_handle(alignments, table, flank, bins, chunk, want_pairs, want_frag_pairs)
    try:
        out = ing.fetch()
        if want_pairs:
            out['clm_ptr'], out['clm'], out['crd_ptr'], out['crd'] = ing.fetch_pairs(max_read_pairs, out['full_cnt'])
            out['ht_first'] = ing.fetch_ht_order()
        if want_frag_pairs:
            out['fp_i'], out['fp_j'] = ing.fetch_frag_pairs()
        return out
    finally:
        ing.destroy()


def _modal_share(values):
    """count of the most frequent value / number of values (scipy.stats.mode(...)[1] / n)"""
    return np.unique(values, return_counts=True)[1].max() / len(values)


def cal_concordance_ratio(coord_list, shorter_len, nwindows):
    """cal_concordance_ratio() :419-428 — the share of read pairs on the best-populated diagonal (y - x) or
    anti-diagonal (y + x) of the contig pair, in windows of shorter_len // nwindows bp"""
    xy = np.asarray(coord_list).reshape(-1, 2)
    width = shorter_len // nwindows
    return max(_modal_share((xy[:, 1] - xy[:, 0]) // width), _modal_share((xy[:, 1] + xy[:, 0]) // width))


def cal_concentration_adj_ratio(coord_list, bin_width=10000):
    """cal_concentration_adj_ratio() :431-451 — per axis, the share of read pairs that sit in bins holding at least ten
    times the median bin; returns (1 - share_x) * (1 - share_y)"""
    xy = np.asarray(coord_list).reshape(-1, 2)
    keep = 1.0
    for axis in (0, 1):
        per_bin = np.unique(xy[:, axis] // bin_width, return_counts=True)[1]
        crowded = per_bin[per_bin >= 10 * np.median(per_bin)]
        keep *= 1 - int(crowded.sum()) / len(xy)
    return keep


class IngestSession:
    """The device-resident result of one S5 call (parse_alignments* :1596-1752): the ingest handle with its link tables (and,
    for the CLM / coordinate side products, the kept read pairs) in HBM, the fragment table that names its ids, and host copies
    of the tables in dict order that are fetched only when a container asks for them.  The containers the mirrors return
    (containers.LinkTable / PairLists) all point here; the handle is destroyed when the last of them lets go."""

    def __init__(self, ing, table, fa_dict, args, pos_int_type, dist_int_type):
        self.ing, self.table = ing, table
        self.record = bool(args.remove_allelic_links or args.remove_concentrated_links)
        self.max_read_pairs = int(args.max_read_pairs) if self.record else 0
        self.allelic, self.concentrated = bool(args.remove_allelic_links), bool(args.remove_concentrated_links)
        self.nwindows = getattr(args, 'nwindows', 50)
        self.dist = ('i', np.int32) if dist_int_type == 'int32' else ('l', np.int64)
        self.pos = ('i', np.int32) if pos_int_type == 'int32' else ('l', np.int64)
        self.weighted = False                   # flank values are float64 weights (normalize_by_nlinks) instead of counts
        self._host = {}
        self._pairs = None
        self._ht_names = None
        self._ht_queued = False

    # ---- host copies, on demand
    def _fetch(self, *keys):
        missing = [k for k in keys if k not in self._host]
        if missing:
            self._host.update(self.ing.fetch(want=missing))
        return [self._host[k] for k in keys]

    def frag_links(self):
        return self._fetch('frag_links')[0]

    def n_keys(self, kind):
        if kind == 'HT':
            return len(self._host['ht_items'][0]) if 'ht_items' in self._host else self.ing.n_ht_items()
        return self.ing.n_full if kind == 'full' else self.ing.n_flank

    def link_arrays(self, kind):
        if kind == 'full':
            return (*self._fetch('full_i', 'full_j', 'full_cnt'), self.table.ctg_names)
        if kind == 'flank':
            fi, fj = self._fetch('flank_i', 'flank_j')
            if self.weighted:
                if 'flank_val' not in self._host:
                    self._host['flank_val'] = self.ing.fetch_flank_values()
                return fi, fj, self._host['flank_val'], self.table.frag_names
            return fi, fj, self._fetch('flank_cnt')[0], self.table.frag_names
        # HT_link_dict: the (contig pair, quadrant) entries in the order of the stream position of their first read pair
        # (update_HT_link_dict :404-416), ordered on the device; names carry the '_H' / '_T' suffix of the quadrant's two ends
        if 'ht_items' not in self._host:
            self._host['ht_items'] = self.ing.fetch_ht_items()
        if self._ht_names is None:
            self._ht_names = [n + s for n in self.table.ctg_names for s in ('_H', '_T')]
        return (*self._host['ht_items'], self._ht_names)

    def pairs(self):
        """(clm_ptr, clm, crd_ptr, crd): hhx_ingest_fetch_pairs, lists in full_link_dict order"""
        if self._pairs is None:
            self._pairs = self.ing.fetch_pairs(self.max_read_pairs, self._fetch('full_cnt')[0])
        return self._pairs

    def pair_items(self, kind):
        """(key, array) items of clm_dict (update_clm_dict :395-401) / ctg_coord_dict (record_coord_pairs :454-471)"""
        from .containers import slices_as_arrays
        cn = np.empty(len(self.table.ctg_names), object)
        cn[:] = self.table.ctg_names
        fi, fj = self._fetch('full_i', 'full_j')
        keys = list(zip(cn[fi].tolist(), cn[fj].tolist()))
        clm_ptr, clm, crd_ptr, crd = self.pairs()
        if kind == 'clm':
            return zip(keys, slices_as_arrays(self.dist[0], clm.astype(self.dist[1], copy=False), clm_ptr, 4))
        values = list(slices_as_arrays(self.pos[0], crd.astype(self.pos[1], copy=False), crd_ptr, 2))
        # a contig pair that reached max_read_pairs was replaced by its [concordance ratio, concentration factor] (:460-471)
        ctg_len = self.table.ctg_len
        for k in np.flatnonzero(np.diff(crd_ptr) >= self.max_read_pairs).tolist():
            c = values[k]
            if self.allelic:
                shorter_len = int(min(ctg_len[fi[k]], ctg_len[fj[k]]))
                values[k] = [cal_concordance_ratio(c, shorter_len, self.nwindows), 1]
            if self.concentrated:
                # as the reference (:466): evaluated on the dict entry AFTER the replacement above, i.e. on the
                # two-element [ratio, 1] when both options are on (which always gives 1.0)
                adj_ratio = cal_concentration_adj_ratio(values[k])
                if self.allelic:
                    values[k][1] = adj_ratio
                else:
                    values[k] = [0, adj_ratio]
        return zip(keys, values)

    def note_thawed(self):
        pass

    # ---- device-side steps on the resident flank table
    def weigh_flank(self, mode, per_frag=None, tag=None, param=0.0):
        """hhx_link_weights over the flank table in HBM (a6); the values become float64"""
        self.ing.weigh_flank(mode, per_frag=per_frag, tag=tag, param=param)
        self.weighted = True
        self._host.pop('flank_val', None)

    def link_matrix(self, in_set, n_rest, add_self_loops):
        return self.ing.link_matrix(in_set, n_rest, add_self_loops=add_self_loops, weighted=self.weighted)

    def write_clm(self, path):
        """paired_links.clm: queued on the library's file-writer thread (or written here with HAPHIC_SYNC_FILES=1).  The kept read pairs are
        released with the file when nothing else of run() can ask for them: no coordinate lists (--remove_allelic_links /
        --remove_concentrated_links) and HT_link_dict already on its way to HT_links.pkl (:2879 precedes :2888)."""
        if not _lib.files_async():
            return self.ing.write_clm(path, self.table.ctg_names)
        return self.ing.write_clm_async(path, self.table.ctg_names, drop_pairs=not self.record and self._ht_queued)

    def queue_pickle(self, kind, path):
        """full_links.pkl / HT_links.pkl of a frozen table on the file-writer thread, from the device tables; False: not possible (the caller
        writes it from the host arrays)"""
        if kind == 'HT':
            if self._ht_names is None:
                self._ht_names = [n + s for n in self.table.ctg_names for s in ('_H', '_T')]
            names = self._ht_names
        else:
            names = self.table.ctg_names if kind == 'full' else self.table.frag_names
        self.ing.write_link_pickle_async(kind, path, names)
        if kind == 'HT':
            self._ht_queued = True
        return True


def ingest_session(alignments, table, fa_dict, args, bins, pos_int_type, dist_int_type, chunk=1 << 22, want_frag_pairs=False):
    """Alignments (the reference's generators :1539-1593, or this package's PairsText / BamRecords / IdArrays) through the device
    ingest; the handle stays alive inside the returned IngestSession."""
    ing = _ingest_handle(alignments, table, int(args.flank * 1000), bins, chunk, want_pairs=True, want_frag_pairs=want_frag_pairs,
                         sweep_follows=not getattr(args, 'skip_clustering', False))
    return IngestSession(ing, table, fa_dict, args, pos_int_type, dist_int_type)


def _s5_containers(session):
    from .containers import LinkTable, PairLists
    full, flank, HT = LinkTable(session, 'full'), LinkTable(session, 'flank'), LinkTable(session, 'HT')
    frag_link = defaultdict(int)
    fn = session.table.frag_names
    links = session.frag_links()
    for f in np.flatnonzero(links).tolist():
        frag_link[fn[f]] = int(links[f])
    clm_dict = PairLists(session, 'clm', session.dist[0])
    ctg_coord_dict = PairLists(session, 'crd', session.pos[0]) if session.record else defaultdict(lambda c=session.pos[0]: array(c))
    return full, flank, HT, clm_dict, frag_link, ctg_coord_dict


def parse_alignments_for_ctgs(alignments, fa_dict, args, ctg_len_dict, Nx_ctg_set, pos_int_type, dist_int_type):
    """parse_alignments_for_ctgs() :1596-1655 — all six containers from the device: link tables, HT counts, the
    CLM distance lists and the first coordinates of every contig pair.  The big ones come back FROZEN (containers.py): their
    entries stay in HBM / numpy arrays until something other than the next seams of run() touches them."""
    logger.info('Parsing input alignments...')
    t_start = time.perf_counter()
    table = FragTable.from_reference(fa_dict, ctg_len_dict, Nx_ctg_set)
    t_table = time.perf_counter()
    session = ingest_session(alignments, table, fa_dict, args, False, pos_int_type, dist_int_type)
    t_session = time.perf_counter()
    out = _s5_containers(session)
    if hasattr(alignments, 'stats'):
        alignments.stats.update(frag_table_s=t_table - t_start, ingest_s=t_session - t_table, containers_s=time.perf_counter() - t_session)
    return out


def parse_alignments(alignments, fa_dict, args, bin_size, frag_len_dict, Nx_frag_set, split_ctg_set, pos_int_type,
                     dist_int_type):
    """parse_alignments() :1658-1752 (some contigs split into bins): all seven containers from the device."""
    logger.info('Parsing input alignments...')
    table = FragTable.from_reference(fa_dict, frag_len_dict, Nx_frag_set, split_ctg_set, bin_size)
    session = ingest_session(alignments, table, fa_dict, args, True, pos_int_type, dist_int_type,
                             want_frag_pairs=bool(args.remove_allelic_links))
    ctg_pair_to_frag = defaultdict(set)                         # :1731-1733
    if args.remove_allelic_links:
        fn, cn = table.frag_names, table.ctg_names
        fp_i, fp_j = session.ing.fetch_frag_pairs()
        ci = np.searchsorted(table.ctg_frag0, fp_i, side='right') - 1     # fragment id -> its contig
        cj = np.searchsorted(table.ctg_frag0, fp_j, side='right') - 1
        for fi, fj, a, b in zip(fp_i.tolist(), fp_j.tolist(), ci.tolist(), cj.tolist()):
            ca, cb = cn[a], cn[b]
            ctg_pair_to_frag[(ca, cb) if ca <= cb else (cb, ca)].add((fn[fi], fn[fj]))
    return _s5_containers(session) + (ctg_pair_to_frag,)


# ------------------------------------------------------------------ S6: run_mcl_clustering
def _inflation_values(min_inflation, max_inflation, inflation_step):
    # numpy.arange over Decimal objects (:2138-2155): start, start+step, ... < max+step
    start, step = Decimal(str(min_inflation)), Decimal(str(inflation_step))
    end = Decimal(str(max_inflation)) + step
    vals = []
    k = 0
    count = int(ceil((end - start) / step))
    while k < count:
        vals.append(start + k * step)
        k += 1
    return vals


def get_main_groups(result_clusters, len_ratio):
    """:2098-2107"""
    main_groups = len(result_clusters)
    for k in range(len(result_clusters) - 1):
        if result_clusters[k + 1][1] / result_clusters[k][1] < len_ratio:
            return k + 1
    return main_groups


def recommend_inflation(result_stat, nchrs, len_ratio):
    """:2110-2129 — the log line below is parsed by HapHiC_pipeline.py:385, wording is API"""
    separated = sorted((infl for infl, groups in result_stat if groups >= nchrs))
    if separated:
        logger.info('You could try inflation from {} (length ratio = {})'.format(separated[0], len_ratio))
        return True
    if len_ratio > 0.5:
        logger.info('The length ratio ({}) might be too strict, trying a lower one...'.format(len_ratio))
        return False
    logger.info('It seems that some chromosomes were grouped together (length ratio = {}) '
                'You could check whether the parameters used are correct / appropriate and '
                'then try to tune the parameters for assembly correction, contig / Hi-C link '
                'filtration, or Markov clustering'.format(len_ratio))
    return True


def _write_inflation_dir(outdir, inflation, result_clusters, group_lines, fa_dict, timing_row):
    """inflation_X/mcl_inflation_X.clusters.txt and the group files (:2200-2218); group_lines: the body of every group file when it was cut from arrays"""
    t0 = time.perf_counter()
    os.makedirs(outdir, exist_ok=True)
    with open(os.path.join(outdir, 'mcl_inflation_{}.clusters.txt'.format(inflation)), 'w') as fout:
        fout.write('#Group\tnContigs\tContigs\n')
        fout.write(''.join('group{}_{}bp\t{}\t{}\n'.format(k, group_len, len(ctgs), ' '.join(ctgs)) for k, (ctgs, group_len) in enumerate(result_clusters, 1)))
    for k, (ctgs, group_len) in enumerate(result_clusters, 1):
        with open(os.path.join(outdir, 'group{}_{}bp.txt'.format(k, group_len)), 'w') as fout:
            if group_lines is not None:
                fout.write('#Contig\tRECounts\tLength\n' + group_lines[k - 1])
            else:
                fout.write('#Contig\tRECounts\tLength\n' + ''.join('{}\t{}\t{}\n'.format(ctg, fa_dict[ctg][2], fa_dict[ctg][1]) for ctg in ctgs))
    timing_row.append(time.perf_counter() - t0)          # [.., seconds the helper thread spent on this directory]


SWEEP_STAGES = {}      # of the last run_mcl_clustering with a DenseSweep: seconds of its steps on the caller's thread (measurement only)
SWEEP_TIMING = []      # of the last run_mcl_clustering: [inflation, seconds of mcl() + interpret_result, seconds of its cluster / group files] (measurement only)


def run_mcl_clustering(link_matrix, bin_set, frag_len_dict, frag_index_dict, expansion, min_inflation,
                       max_inflation, inflation_step, max_iter, pruning, fa_dict, nchrs, dense_matrix=False,
                       outdir_root='.', dist=None, _block_rows=None, _engine=None):
    """run_mcl_clustering() :2132-2242.  link_matrix: scipy CSC or a DeviceCSR.  The normalised,
    pre-expanded matrix is built once and stays in HBM for the whole inflation sweep (_block_rows: rows per float32 block of
    the dense sweep — tests force the blocked path at small orders with it).  With a torch.distributed
    group (`dist`, one process per GPU, every rank holding the link matrix) the sweep is shared out over the ranks
    (sharded.sweep_sharded: one expansion across the ranks, the heavy iterations of the low inflations row-sharded, the light
    remainders dealt by predicted cost; expansion != 2: whole inflations dealt round-robin, sharded.inflation_sweep); every rank
    gets all results, rank 0 writes the files.  _engine: the arithmetic behind sharded.py (default: the HIP library on the
    current device; CPU tests pass an oracle-backed one)."""
    if dense_matrix:
        raise ValueError('dense_matrix mode is not on the MI355X path; use the reference function')
    logger.info('Performing Markov clustering...')
    index_frag = {i: f for f, i in frag_index_dict.items()}
    if isinstance(link_matrix, ResidentMatrix):                      # straight from dict_to_matrix, still in HBM
        m = link_matrix.take_device()
    elif isinstance(link_matrix, _lib.DeviceCSR):
        m = link_matrix.copy()
    else:
        m = _to_device(link_matrix)
    n = m.shape3[0]
    # The reference pre-expands once (:2146-2147) and restarts every inflation from that matrix.  M^e is nearly dense: as a
    # CSR matrix it is materialised only while it is guaranteed to fit scipy's int32 index range (n^2 < 2^31).  Beyond that
    # (and always when a sweep of several inflations is asked for with expansion 2 on a matrix of that size) the rows of M^2
    # are kept as float32 row blocks filled by ONE pass over the products (DenseSweep), and iteration 0 of every inflation is the
    # row-local epilogue over them; a single inflation starts from the link matrix with the pre-expansion fused into its
    # iteration 0 (hhx_mcl_links) — same results, the n^2-entry CSR matrix never exists.
    materialise = expansion > 1 and n * n < 2 ** 31 and not _block_rows
    inflations = _inflation_values(min_inflation, max_inflation, inflation_step)
    pre = None
    sweep = None
    world = dist.get_world_size() if dist is not None else 1
    rank = dist.get_rank() if dist is not None else 0
    across = world > 1 and expansion == 2 and max_iter >= 1          # the sweep shared out over the ranks (sharded.sweep_sharded)
    if across:
        pass
    elif expansion == 2 and not materialise and len(inflations) > 1 and max_iter >= 1:
        sweep = DenseSweep(m, pruning, block_rows=_block_rows)
        if _block_rows is None and len(sweep.bounds) - 1 > DenseSweep.MAX_BLOCKS:
            # under memory pressure the blocks get small and every block pays the normalisation and the operand layout again,
            # without the symmetric half: beyond a handful of blocks one fused expansion per inflation is the cheaper sweep
            sweep = None
    elif materialise:
        _lib.normalize_l1(m)                                         # :2144
        pre = m
        for _ in range(2, expansion + 1):                            # :2146-2147
            nxt = _lib.spgemm(pre, m, fx_shift=52)
            if pre is not m:
                pre.free()
            pre = nxt
    elif expansion <= 1:
        _lib.normalize_l1(m)
        pre = m
    result_clusters_list = []
    mcl_nrounds = 0
    # expansion != 2 with several ranks: the inflations are dealt round-robin; a rank forms iteration 0 only of its own inflations
    mine = [infl for k, infl in enumerate(inflations) if k % world == rank] if world > 1 else inflations
    firsts = sweep.first_iterations(mine) if sweep is not None else None

    def run_one(inflation):
        if sweep is not None:
            first = next(firsts)
            t_0 = time.perf_counter()
            res = mcl_resume_device(first, expansion, float(inflation), max_iter, pruning)
            sweep.stage_s.setdefault('tails', []).append(time.perf_counter() - t_0)
        elif pre is not None:
            res = mcl_device(pre, expansion, float(inflation), max_iter, pruning)
        else:
            res = mcl_device(m, expansion, float(inflation), max_iter, pruning, links=True)
        try:
            return _lib.interpret(res) + (res.shape3[0],)
        finally:
            res.free()

    if across:
        from . import sharded
        if _engine is None:
            import torch
            _engine = sharded.HipEngine('cuda:%d' % torch.cuda.current_device())
        attractor_arrays = []
        for inflation, (att, att_ptr, members, shape, n_iter, converged) in zip(
                inflations, sharded.sweep_sharded(_engine, m, inflations, max_iter, pruning, dist)):
            _log_mcl(n_iter, converged, expansion, float(inflation), max_iter, pruning)      # in the order of the sweep, like the reference
            attractor_arrays.append((att, att_ptr, members, shape))
        write_files = rank == 0
    elif world > 1:
        from . import sharded
        attractor_arrays = sharded.inflation_sweep(run_one, inflations, dist)
        write_files = rank == 0
    else:
        attractor_arrays = None
        write_files = True
    timing = SWEEP_TIMING
    del timing[:]
    file_writer, file_jobs = None, []
    # the per-contig half of :2172-2218 on arrays when no fragment is a bin (the loops below otherwise): names, lengths and the group-file line of
    # every matrix index, gathered once; per inflation the clusters are cut out of them with numpy instead of one Python statement per contig
    by_index = None
    if not bin_set:
        order = sorted(index_frag)
        if order == list(range(len(order))):
            names_by_index = np.empty(len(order), object)
            names_by_index[:] = [index_frag[i] for i in order]
            by_index = (names_by_index, np.fromiter((fa_dict[f][1] for f in names_by_index), np.int64, len(order)),
                        np.array(['{}\t{}\t{}\n'.format(f, fa_dict[f][2], fa_dict[f][1]) for f in names_by_index], object))
    for k_infl, inflation in enumerate(inflations):
        t_start = time.perf_counter()
        att, att_ptr, members, shape = attractor_arrays[k_infl] if attractor_arrays is not None else run_one(inflation)
        timing.append([str(inflation), time.perf_counter() - t_start, 0.0])          # [inflation, seconds of mcl() + interpret, seconds of the files]
        t_start = time.perf_counter()
        mcl_nrounds += 1
        clusters = _clusters_from_arrays(att, att_ptr, members, shape)
        if not clusters:
            logger.info('Some fragments are missing / redundant, result of inflation {} will NOT be output'.format(inflation))
            continue
        outdir = os.path.join(outdir_root, 'inflation_{}'.format(inflation))
        if by_index is not None:
            result_clusters, group_lines = _groups_from_arrays(clusters, *by_index)
        else:
            group_lines = None
            groups = defaultdict(lambda: [[], 0])       # cluster number -> [contigs, total length]
            split_votes = defaultdict(dict)             # split contig -> {cluster number: summed bin length}
            for k, indexes in enumerate(clusters):
                for i in indexes:
                    frag = index_frag[i]
                    if frag in bin_set:                                  # :2176-2183
                        ctg = frag.rsplit('_bin', 1)[0]
                        split_votes[ctg][k] = split_votes[ctg].get(k, 0) + frag_len_dict[frag]
                    else:
                        groups[k][0].append(frag)
                        groups[k][1] += fa_dict[frag][1]
            for ctg, votes in split_votes.items():                       # :2190-2194
                best = sorted(votes.keys(), key=lambda c: votes[c], reverse=True)[0]
                groups[best][0].append(ctg)
                groups[best][1] += fa_dict[ctg][1]
            result_clusters = sorted(tuple(groups.values()), key=lambda g: g[1], reverse=True)   # stable, :2197
            for ctgs, group_len in result_clusters:
                ctgs.sort(key=lambda c: fa_dict[c][1], reverse=True)                            # :2208 (and :2214 for the file below: same order)
        if not write_files:
            result_clusters_list.append((inflation, result_clusters))
            continue
        # the directory of this inflation is written by a helper thread while the next inflation's mcl() waits for the device (the library calls
        # release the GIL): ~0.1-0.7 s of Python and small-file system calls per inflation that the sweep no longer waits for one after the other.
        # Joined (failures re-raised) before this function returns: the files exist when the reference's would.
        if file_writer is None:
            from concurrent.futures import ThreadPoolExecutor
            file_writer = ThreadPoolExecutor(1)
        file_jobs.append(file_writer.submit(_write_inflation_dir, outdir, inflation, result_clusters, group_lines, fa_dict, timing[-1]))
        result_clusters_list.append((inflation, result_clusters))
        timing[-1][2] = time.perf_counter() - t_start
    try:
        for job in file_jobs:
            job.result()
    finally:
        if file_writer is not None:
            file_writer.shutdown(wait=True)
    if pre is not None and pre is not m:
        pre.free()
    if sweep is not None:
        SWEEP_STAGES.clear()
        SWEEP_STAGES.update(sweep.stage_s)
        sweep.close()
    m.free()
    max_nclusters = max([len(rc) for _, rc in result_clusters_list])
    if max_nclusters < nchrs:
        logger.warning('The maximum number of clusters ({}) is even less than the expected number of '
                       'chromosomes ({}). You could try higher inflation.'.format(max_nclusters, nchrs))
    else:
        for len_ratio in (0.75, 0.7, 0.65, 0.6, 0.55, 0.5):
            stat = [(infl, get_main_groups(rc, len_ratio)) for infl, rc in result_clusters_list]
            if recommend_inflation(stat, nchrs, len_ratio):
                break
    return result_clusters_list, mcl_nrounds
