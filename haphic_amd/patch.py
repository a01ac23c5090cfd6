"""Re-binding of the reference's own seams (SURVEY §8b S1–S6) onto the MI355X library.

    import HapHiC_cluster as H          # the unmodified reference module
    import haphic_amd.patch
    haphic_amd.patch.patch_reference(H) # then H.run(args) / HapHiC_pipeline as usual

Nothing under /root/reference is edited; the reference resolves these names at call time through its
module globals, so `setattr` on the module is all that is needed."""
from . import cluster

# seam -> (reference line, replacement)
SEAMS = {
    'dot_product_mkl': ('HapHiC_cluster.py:39-43', cluster.dot_product_mkl),            # S1
    'mkl_matrix_power': ('HapHiC_cluster.py:2017-2023', cluster.mkl_matrix_power),      # S1 (keeps M^(e-1) on the device)
    'mcl': ('HapHiC_cluster.py:2026-2062', cluster.mcl),                                # S2
    'prune': ('HapHiC_cluster.py:1987-2014', cluster.prune),                            # S3
    'interpret_result': ('HapHiC_cluster.py:2065-2095', cluster.interpret_result),      # a12
    'run_mcl_clustering': ('HapHiC_cluster.py:2132-2242', cluster.run_mcl_clustering),  # S6
    'count_RE_sites': ('HapHiC_cluster.py:75-84', cluster.count_RE_sites),              # a5
    'parse_fasta': ('HapHiC_cluster.py:87-113', cluster.parse_fasta),                   # a5
    'stat_fragments': ('HapHiC_cluster.py:188-296', cluster.stat_fragments),            # a5
    'filter_fragments': ('HapHiC_cluster.py:741-940', cluster.filter_fragments),        # f1 (rank sums on the device)
    'normalize_by_nlinks': ('HapHiC_cluster.py:718-724', cluster.normalize_by_nlinks),  # a6
    'normalize_by_length': ('HapHiC_cluster.py:727-738', cluster.normalize_by_length),  # a6 (dead code in the reference)
    'reduce_inter_hap_HiC_links': ('HapHiC_cluster.py:695-707', cluster.reduce_inter_hap_HiC_links),   # a6 (GFA phasing)
}
# f2 / f3: what run() does with the S5 containers besides the seams above — the two writers (:2879, :2888, :2929) and the per-group link sums of
# output_statistics (:2354).  They only differ from the reference's functions
# for the array-backed containers of the S5 mirrors (haphic_amd/containers.py), so they travel with `ingest`.
CONTAINER_SEAMS = {
    'output_pickle': ('HapHiC_cluster.py:710-715', cluster.output_pickle),
    'output_clm': ('HapHiC_cluster.py:376-392', cluster.output_clm),
    'parse_link_dict': ('HapHiC_cluster.py:2252-2268', cluster.group_link_dict),     # per-group link sums of output_statistics :2279
}
# S4/S5: dict_to_matrix is also called in dense mode by the filters (:603) — the mirror returns `.toarray()` then;
# the device ingest returns all of the reference's containers (link tables, HT counts, CLM distance lists, first
# coordinates, ctg_pair_to_frag :1731 for split contigs + --remove_allelic_links).
OPTIONAL = {
    'dict_to_matrix': ('HapHiC_cluster.py:310-373', cluster.dict_to_matrix),
    'parse_alignments_for_ctgs': ('HapHiC_cluster.py:1596-1655', cluster.parse_alignments_for_ctgs),
    'parse_alignments': ('HapHiC_cluster.py:1658-1752', cluster.parse_alignments),
    'pairs_generator': ('HapHiC_cluster.py:1539-1559', cluster.pairs_generator),                      # a1
    'pairs_generator_inter_ctgs': ('HapHiC_cluster.py:1562-1583', cluster.pairs_generator_inter_ctgs),
    'bam_generator': ('HapHiC_cluster.py:1586-1593', cluster.bam_generator),                          # f4
}


# position of `dense_matrix` in the reference signatures: --dense_matrix (:2723) is the reference's own numpy mode, which
# stays the reference's code — a seam called with dense_matrix=True is handed back to the original function
DENSE_ARG = {'mcl': 5, 'prune': 2, 'interpret_result': 1, 'run_mcl_clustering': 12}


def _dense_dispatch(ours, original, idx):
    def seam(*args, **kwargs):
        dense = kwargs.get('dense_matrix', args[idx] if len(args) > idx else False)
        if dense and original is not None:
            return original(*args, **kwargs)
        return ours(*args, **kwargs)
    seam.__wrapped__ = ours
    seam.__name__ = getattr(ours, '__name__', 'seam')
    seam.__doc__ = ours.__doc__
    return seam


def _with_original(ours, original):
    def seam(*args, **kwargs):
        return ours(*args, _original=original, **kwargs)
    seam.__wrapped__ = ours
    seam.__name__ = ours.__name__
    seam.__doc__ = ours.__doc__
    return seam


def _run_then_join(original):
    """run() :2738-2959 with the library's file-writer thread joined before it returns: output_pickle / output_clm of the array-backed
    containers only QUEUE HT_links.pkl, paired_links.clm and full_links.pkl (haphic_amd/csrc/hhx_jobs.hip), so the files must be complete —
    and a writer's failure raised, as the reference's own writers would raise inside run() — before the caller (main :2967,
    HapHiC_pipeline.py:358, which starts `haphic reassign` on full_links.pkl next) goes on."""
    from . import _lib

    def run(*args, **kwargs):
        try:
            out = original(*args, **kwargs)
        except BaseException:
            try:
                _lib.files_join()             # the run failed for its own reason: that one is raised, the queue is only drained
            except RuntimeError:
                pass
            raise
        _lib.files_join()
        return out
    run.__wrapped__ = original
    run.__name__ = 'run'
    run.__doc__ = original.__doc__
    return run


def patch_reference(H, ingest=True, matrix_build=True):
    """H: the imported reference module (HapHiC_cluster).  Returns {name: original} so the caller can undo."""
    from . import _lib
    _lib.load()                          # fail loudly here if the HIP library is missing
    saved = {}
    seams = dict(SEAMS)
    if matrix_build:
        seams['dict_to_matrix'] = OPTIONAL['dict_to_matrix']
    if ingest:
        seams['parse_alignments_for_ctgs'] = OPTIONAL['parse_alignments_for_ctgs']
        seams['parse_alignments'] = OPTIONAL['parse_alignments']
        seams['pairs_generator'] = OPTIONAL['pairs_generator']                # a1: only together with S5, which consumes it
        seams['pairs_generator_inter_ctgs'] = OPTIONAL['pairs_generator_inter_ctgs']
        seams['bam_generator'] = OPTIONAL['bam_generator']                    # f4: consumed by the same S5 mirrors
        seams.update(CONTAINER_SEAMS)
    for name, (_cite, fn) in seams.items():
        saved[name] = getattr(H, name, None)
        if name in CONTAINER_SEAMS:
            fn = _with_original(fn, saved[name])
        setattr(H, name, _dense_dispatch(fn, saved[name], DENSE_ARG[name]) if name in DENSE_ARG else fn)
    if ingest and getattr(H, 'run', None) is not None:
        saved['run'] = H.run
        H.run = _run_then_join(H.run)    # main() :2967 and HapHiC_pipeline.py:358 resolve `run` through the module, like every seam
    saved['INTEL_MKL'] = getattr(H, 'INTEL_MKL', None)
    H.INTEL_MKL = True                   # :2764-2768 would otherwise force the dense (numpy) mode
    return saved


def patch_reassign(R):
    """R: the imported HapHiC_reassign module.  f3: parse_link_dict :217-263 (the per-group link sums behind reassign's
    link densities) on the device for integer link counts; float / normalised links go to the original function."""
    from . import _lib
    _lib.load()
    original = R.parse_link_dict

    def parse_link_dict(link_dict, ctg_group_dict, normalize_by_nlinks=False):
        return cluster.parse_link_dict(link_dict, ctg_group_dict, normalize_by_nlinks, _original=original)
    parse_link_dict.__wrapped__ = cluster.parse_link_dict
    R.parse_link_dict = parse_link_dict
    return {'parse_link_dict': original}


def unpatch_reference(H, saved):
    for name, fn in saved.items():
        if fn is None and hasattr(H, name):
            delattr(H, name)
        elif fn is not None:
            setattr(H, name, fn)
