"""`python -m haphic_amd cluster <arguments of "haphic cluster">` — HapHiC's own step 1 with the MI355X library behind
its seams (SURVEY §8b "who calls it"): import the reference's HapHiC_cluster module, re-bind S1-S6 with
haphic_amd.patch.patch_reference, call the reference's run(args, log_file) exactly as its main() does (:2962-2967).
Argument parsing, logging, file formats and every stage outside the hot path are the reference's own code.
`python -m haphic_amd plot <arguments of "haphic plot">` does the same for HapHiC_plot.py: parse_pairs / parse_bam (the read-pair
binning into the scaffold-bin contact matrix) run on the device (haphic_amd.plot.patch_plot), main() is the reference's.

The reference checkout is found through --reference DIR or $HAPHIC_REFERENCE (the repository root or its scripts/
directory).  Extra flags of the wrapper (removed before the reference parses the command line):
  --device N                 HIP device ordinal (default 0)
  --keep-reference-ingest    leave S5 / a1 (parse_alignments*, pairs_generator*) to the reference
  --stub-missing-imports     development boxes only: empty stand-ins for pysam / portion when they are not installed
                             (the .pairs path needs neither; BAM input then fails loudly inside the reference)
"""
import os
import sys
import types


def _take(argv, flag, has_value):
    if flag not in argv:
        return None
    k = argv.index(flag)
    if not has_value:
        del argv[k]
        return True
    if k + 1 >= len(argv):
        raise SystemExit('{} needs a value'.format(flag))
    value = argv[k + 1]
    del argv[k:k + 2]
    return value


def _reference_scripts(path):
    for cand in (path, os.path.join(path or '', 'scripts')):
        if cand and os.path.isfile(os.path.join(cand, 'HapHiC_cluster.py')):
            return cand
    raise SystemExit('HapHiC checkout not found: pass --reference DIR or set HAPHIC_REFERENCE (looked for HapHiC_cluster.py in {!r})'.format(path))


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    if not argv or argv[0] in ('-h', '--help'):
        print(__doc__)
        return 0
    command = argv.pop(0)
    if command not in ('cluster', 'plot'):
        raise SystemExit('haphic_amd wraps the "cluster" and "plot" steps only (got {!r}); run the other steps with the reference'.format(command))
    ref = _take(argv, '--reference', True) or os.environ.get('HAPHIC_REFERENCE')
    device = int(_take(argv, '--device', True) or 0)
    keep_ingest = bool(_take(argv, '--keep-reference-ingest', False))
    stub = bool(_take(argv, '--stub-missing-imports', False))
    scripts = _reference_scripts(ref)
    if stub:
        for name, attrs in (('pysam', {'set_verbosity': lambda *a, **k: None, 'AlignmentFile': None}), ('portion', {'closed': None, 'empty': None})):
            try:
                __import__(name)
            except ImportError:
                m = types.ModuleType(name)
                m.__dict__.update(attrs)
                sys.modules[name] = m
    from . import _lib, patch
    _lib.check(_lib.load().hhx_set_device(device))                 # fails here, loudly, without a GPU or the library
    sys.path.insert(0, scripts)
    if command == 'plot':
        import HapHiC_plot as P                                     # needs pysam / portion / matplotlib, as the reference does
        from . import plot
        plot.patch_plot(P)
        sys.argv = ['haphic plot'] + argv
        P.main()
        return 0
    import HapHiC_cluster as H                                      # the unmodified reference module
    patch.patch_reference(H, ingest=not keep_ingest)
    sys.argv = ['haphic cluster'] + argv
    H.run(H.parse_arguments(), 'HapHiC_cluster.log')                # == HapHiC_cluster.main() :2962-2967
    return 0


if __name__ == '__main__':
    sys.exit(main())
