"""Synthetic assemblies and Hi-C read pairs (SURVEY.md §8d pair model).

Genome: `nchrs` equal chromosomes chopped into contigs by the simulation/sim_contigs.py recipe
(lengths ~ N(mean, cv*mean), min_len floor, random orientation, names
`Chr{c}_{k}_{start}_{end}_{ori}_{len}`, reference simulation/sim_contigs.py:43-52,:102).
RE-site counts are synthesised as len/256 + 1 (uniform ACGT gives one GATC per 256 bp; the +1 is the
reference's pseudo-count, HapHiC_cluster.py:75-84) instead of carrying FASTA text.

Pairs: position 1 uniform over the genome; with probability `cis` the mate lies on the same
chromosome at distance d ~ P(d) ∝ 1/d on [1 kb, chromosome length], otherwise uniform over the genome.

Sampling runs through torch so the same code fills host arrays for tests and HBM-resident arrays
for bench.py (torch is plumbing here: RNG + device memory).
"""
from dataclasses import dataclass

import numpy as np


@dataclass
class Genome:
    names: list            # contig names, in FASTA order
    length: np.ndarray     # int64 [n]
    chrom: np.ndarray      # int32 [n] truth chromosome
    start: np.ndarray      # int64 [n] offset of the contig on its chromosome
    rev: np.ndarray        # bool  [n] contig is reverse-complemented
    re_sites: np.ndarray   # int64 [n]
    chr_len: int
    nchrs: int

    @property
    def n(self):
        return len(self.names)

    def lexical_rank(self):
        """Rank of each contig name under Python str ordering (key orientation, HapHiC_cluster.py:1629)."""
        order = sorted(range(self.n), key=lambda i: self.names[i])
        rank = np.empty(self.n, np.int32)
        rank[order] = np.arange(self.n, dtype=np.int32)
        return rank


def make_genome(nchrs, chr_len, mean_len, cv=0.3, min_len=5000, seed=12345):
    rng = np.random.default_rng(seed)
    names, length, chrom, start, rev = [], [], [], [], []
    for c in range(nchrs):
        pos, k = 0, 0
        while pos < chr_len:
            ln = int(rng.normal(mean_len, cv * mean_len))
            ln = max(ln, min_len)
            if chr_len - (pos + ln) < min_len:   # absorb a short tail into the last contig
                ln = chr_len - pos
            k += 1
            r = bool(rng.integers(0, 2))
            names.append('Chr{}_{}_{}_{}_{}_{}'.format(c + 1, k, pos + 1, pos + ln, '-' if r else '+', ln))
            length.append(ln); chrom.append(c); start.append(pos); rev.append(r)
            pos += ln
    length = np.asarray(length, np.int64)
    return Genome(names, length, np.asarray(chrom, np.int32), np.asarray(start, np.int64),
                  np.asarray(rev, bool), length // 256 + 1, int(chr_len), int(nchrs))


def sample_pairs(genome, npairs, seed=12345, cis=0.85, device='cpu', min_dist=1000):
    """Returns torch tensors (id1, pos1, id2, pos2): int32 contig ids (FASTA order) and int32 0-based
    positions, resident on `device`."""
    import torch
    g = torch.Generator(device=device)
    g.manual_seed(int(seed))
    L = genome.chr_len
    n = int(npairs)
    dev = torch.device(device)
    # contig boundaries in global genome coordinates (chromosome c occupies [c*L, (c+1)*L))
    gstart = torch.as_tensor(genome.chrom.astype(np.int64) * L + genome.start, device=dev)
    clen = torch.as_tensor(genome.length, device=dev)
    crev = torch.as_tensor(genome.rev, device=dev)

    c1 = torch.randint(0, genome.nchrs, (n,), generator=g, device=dev)
    x1 = (torch.rand(n, generator=g, device=dev, dtype=torch.float64) * L).long().clamp_(0, L - 1)
    is_cis = torch.rand(n, generator=g, device=dev) < cis
    # power-law distance, P(d) ∝ 1/d on [min_dist, L]
    u = torch.rand(n, generator=g, device=dev, dtype=torch.float64)
    d = (min_dist * torch.pow(torch.tensor(L / min_dist, dtype=torch.float64, device=dev), u)).long()
    sign = torch.randint(0, 2, (n,), generator=g, device=dev) * 2 - 1
    x2 = x1 + sign * d
    bad = (x2 < 0) | (x2 >= L)
    x2 = torch.where(bad, x1 - sign * d, x2)
    bad = (x2 < 0) | (x2 >= L)
    xr = (torch.rand(n, generator=g, device=dev, dtype=torch.float64) * L).long().clamp_(0, L - 1)
    x2 = torch.where(bad, xr, x2)
    c2t = torch.randint(0, genome.nchrs, (n,), generator=g, device=dev)
    c2 = torch.where(is_cis, c1, c2t)
    x2 = torch.where(is_cis, x2, xr)

    def locate(c, x):
        gpos = c * L + x
        cid = torch.searchsorted(gstart, gpos, right=True) - 1
        off = gpos - gstart[cid]
        ln = clen[cid]
        pos = torch.where(crev[cid], ln - 1 - off, off)
        return cid.int(), pos.int()

    id1, p1 = locate(c1, x1)
    id2, p2 = locate(c2, x2)
    return id1, p1, id2, p2


def make_polyploid(base, ploidy):
    """`ploidy` collinear copies of `base` (simulation/sim_haplotypes.py without the divergence: same contig
    layout on every haplotype), contig k of haplotype h = id h * base.n + k, named 'hap{h+1}_' + base name."""
    names = ['hap{}_{}'.format(h + 1, nm) for h in range(ploidy) for nm in base.names]
    tile = lambda a: np.concatenate([a] * ploidy)
    chrom = np.concatenate([base.chrom + h * base.nchrs for h in range(ploidy)]).astype(np.int32)
    return Genome(names, tile(base.length), chrom, tile(base.start), tile(base.rev), tile(base.re_sites), base.chr_len,
                  base.nchrs * ploidy)


def add_allelic_pairs(genome, n_base, ploidy, id1, p1, id2, p2, frac, seed, jitter=2000):
    """turn a fraction of the pairs into allelic contacts: the mate lands on the same contig of
    another haplotype at the homologous position +- jitter.  numpy arrays in, numpy arrays out."""
    rng = np.random.default_rng(seed)
    id1, p1, id2, p2 = (np.array(a) for a in (id1, p1, id2, p2))
    n = len(id1)
    pick = rng.random(n) < frac
    hop = rng.integers(1, ploidy, n)
    k, h = id1 % n_base, id1 // n_base
    mate = ((h + hop) % ploidy) * n_base + k
    pos = np.clip(p1 + rng.integers(-jitter, jitter + 1, n), 0, genome.length[mate] - 1)
    id2 = np.where(pick, mate, id2).astype(id2.dtype)
    p2 = np.where(pick, pos, p2).astype(p2.dtype)
    return id1, p1, id2, p2
