"""BASELINE.json configs[3] at its stated size — a 40k-contig autotetraploid, --remove_allelic_links 4 — shared by the generator
of the fixture (tests/golden/make_golden.py c4_40k, which runs the REFERENCE on it) and the GPU test that replays it
(tests/test_gpu_scale.py::test_c4_40k_cluster_files_against_the_reference)."""
import hashlib

import numpy as np

CFG = dict(nchrs=10, chr_contigs=1000, mean_len=30_000, ploidy=4, npairs=20_000_000, allelic=0.05, seed=77,
           max_read_pairs=200, min_read_pairs=20, concordance_ratio_cutoff=0.2, nwindows=50, flank=500,
           inflations=(1.5, 3.0, 0.5))


def inputs():
    """10k collinear contigs x 4 haplotypes; cis power-law pairs inside a haplotype + 5 % allelic contacts between homologous
    positions.  Sampled with torch's CPU generator (the same stream here and on the GPU box); intra-contig pairs dropped as
    run() does by feeding pairs_generator_inter_ctgs (:2865)."""
    from haphic_amd import synth
    c = CFG
    base = synth.make_genome(c['nchrs'], c['chr_contigs'] * c['mean_len'], c['mean_len'], seed=c['seed'])
    gen = synth.make_polyploid(base, c['ploidy'])
    id1, p1, id2, p2 = [t.numpy() for t in synth.sample_pairs(gen, c['npairs'], seed=c['seed'] + 1, device='cpu')]
    id1, p1, id2, p2 = synth.add_allelic_pairs(gen, base.n, c['ploidy'], id1, p1, id2, p2, c['allelic'], c['seed'] + 2)
    keep = id1 != id2
    return gen, base, id1[keep], p1[keep], id2[keep], p2[keep]


def checksum(id1, p1, id2, p2):
    return int(id1.sum(dtype=np.int64) + p1.sum(dtype=np.int64) + id2.sum(dtype=np.int64) + p2.sum(dtype=np.int64))


def _sha(*arrays):
    h = hashlib.sha256()
    for a in arrays:
        h.update(np.ascontiguousarray(a).tobytes())
    return h.hexdigest()


def coord_digest(coord, cid):
    """ctg_coord_dict as remove_allelic_HiC_links :474 receives it -> four digests: key order, which entries were collapsed to
    [ratio, 1] (:460-465), the ratios, the raw coordinate lists"""
    n = len(coord)
    ki, kj = np.empty(n, np.int32), np.empty(n, np.int32)
    collapsed = np.zeros(n, bool)
    ratio = np.full(n, -1.0)
    raw = []
    raw_len = np.zeros(n, np.int64)
    for k, (pair, v) in enumerate(coord.items()):
        ki[k], kj[k] = cid[pair[0]], cid[pair[1]]
        if isinstance(v, list):
            collapsed[k] = True
            ratio[k] = v[0]
            assert v[1] == 1
        else:
            raw_len[k] = len(v)
            raw.append(np.frombuffer(v.tobytes(), np.int32 if v.itemsize == 4 else np.int64).astype(np.int64))
    raw = np.concatenate(raw) if raw else np.zeros(0, np.int64)
    return dict(coord_keys=_sha(ki, kj), coord_collapsed=_sha(collapsed), coord_ratio=_sha(ratio), coord_raw=_sha(raw_len, raw),
                coord_n=n, coord_n_collapsed=int(collapsed.sum()))


def file_digest(path):
    with open(path, 'rb') as fh:
        return hashlib.sha256(fh.read()).hexdigest()


def group_map(clusters_txt, cid):
    """mcl_inflation_X.clusters.txt -> int32 group rank per contig (-1: not in any group): north_star's 'integer contig -> group map'"""
    out = np.full(len(cid), -1, np.int32)
    for k, line in enumerate(l for l in clusters_txt.splitlines() if not l.startswith('#')):
        for c in line.split('\t')[2].split(' '):
            out[cid[c]] = k
    return out
