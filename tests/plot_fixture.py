"""Synthetic AGP + read pairs for the `haphic plot` binning tests (f4): scaffolds made of whole contigs and of the pieces
of broken contigs, both orientations, gap lines, unplaced leftovers; read pairs naming placed contigs, contigs the AGP
does not know, and (on request) positions outside what the AGP covers."""
import numpy as np


def make_case(seed, n_scaffolds=4, ctgs_per=6, bin_size=5000, n_pairs=3000, broken=3, partial=0, out_of_agp=False, short_scaffolds=0):
    """-> dict(agp, pairs, contigs {name: length}, records [(ref, pos, mref, mpos)] 1-based)
    broken: contigs cut in two pieces that go to different scaffolds; partial: contigs of which only a prefix is placed
    (read positions stay inside the alignment bins that prefix touches unless out_of_agp); short_scaffolds: extra
    scaffolds of one small contig (the ones --min_len removes)."""
    rng = np.random.default_rng(seed)
    contigs, lines = {}, []
    pieces = [[] for _ in range(n_scaffolds)]            # per scaffold: (ctg, lo, hi)
    k = 0
    for s in range(n_scaffolds):
        for _ in range(ctgs_per):
            name = 'ctg%04d' % k
            k += 1
            contigs[name] = int(rng.integers(3 * bin_size // 5, 8 * bin_size))
            pieces[s].append((name, 1, contigs[name]))
    for _ in range(broken):
        name = 'brk%04d' % k
        k += 1
        contigs[name] = int(rng.integers(2 * bin_size, 9 * bin_size))
        cut = int(rng.integers(bin_size // 2, contigs[name] - bin_size // 2))
        a, b = rng.choice(n_scaffolds, 2, replace=n_scaffolds < 2)
        pieces[a].insert(int(rng.integers(0, len(pieces[a]) + 1)), (name, 1, cut))
        pieces[b].insert(int(rng.integers(0, len(pieces[b]) + 1)), (name, cut + 1, contigs[name]))
    reach = {}                                           # partially placed contigs: last position a read may take
    for _ in range(partial):
        name = 'par%04d' % k
        k += 1
        contigs[name] = int(rng.integers(4 * bin_size, 9 * bin_size))
        placed = int(rng.integers(bin_size + 1, contigs[name] - 2 * bin_size))
        s = int(rng.integers(0, n_scaffolds))
        pieces[s].append((name, 1, placed))
        reach[name] = contigs[name] if out_of_agp else min(contigs[name], ((placed - 1) // bin_size + 1) * bin_size)
    for _ in range(short_scaffolds):
        name = 'tiny%04d' % k
        k += 1
        contigs[name] = int(rng.integers(bin_size // 4, bin_size))
        pieces.append([(name, 1, contigs[name])])
    for s, comp in enumerate(pieces):
        at, part = 1, 1
        for n, (ctg, lo, hi) in enumerate(comp):
            if n:
                lines.append('group%d\t%d\t%d\t%d\tU\t100\tscaffold\tyes\tproximity_ligation' % (s + 1, at, at + 99, part))
                at, part = at + 100, part + 1
            strand = '+-'[int(rng.integers(0, 2))]
            lines.append('group%d\t%d\t%d\t%d\tW\t%s\t%d\t%d\t%s' % (s + 1, at, at + hi - lo, part, ctg, lo, hi, strand))
            at, part = at + hi - lo + 1, part + 1
    agp = '##agp-version\t2.1\n# synthetic\n' + '\n'.join(lines) + '\n'
    names = list(contigs) + ['ghost_a', 'ghost_b']
    lens = [contigs.get(n, 50000) for n in names]
    a, b = rng.integers(0, len(names), n_pairs), rng.integers(0, len(names), n_pairs)
    same = rng.random(n_pairs) < 0.4
    b[same] = a[same]
    records = []
    for x, y in zip(a.tolist(), b.tolist()):
        px = int(rng.integers(1, reach.get(names[x], lens[x]) + 1))
        py = int(rng.integers(1, reach.get(names[y], lens[y]) + 1))
        records.append((names[x], px, names[y], py))
    body = ['read%d\t%s\t%d\t%s\t%d\t+\t-' % (n, r, p, m, q) for n, (r, p, m, q) in enumerate(records)]
    body.insert(len(body) // 2, '')
    body.insert(len(body) // 3, '#a comment inside the body')
    pairs = '## pairs format v1.0\n#columns: readID chr1 pos1 chr2 pos2 strand1 strand2\n' + '\n'.join(body) + '\n'
    return dict(agp=agp, pairs=pairs, contigs=contigs, records=records, bin_size=bin_size)


CASES = {            # name -> (make_case arguments, min_len in Mb, specified_scaffolds)
    'plain': (dict(seed=1), 0, None),
    'min_len': (dict(seed=2, short_scaffolds=3, broken=4), 0.006, None),
    'specified': (dict(seed=3, n_scaffolds=5), 0, 'group4,group2,group1'),
    'partial_inside': (dict(seed=4, partial=4), 0, None),
    'partial_outside': (dict(seed=5, partial=4, out_of_agp=True), 0, None),          # the reference raises
    'one_bin': (dict(seed=6, n_scaffolds=2, ctgs_per=3, bin_size=200000, broken=1), 0, None),
}
