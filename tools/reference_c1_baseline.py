"""the UNMODIFIED reference `haphic cluster` (HapHiC_cluster.run, dense mode — what the reference selects
itself when MKL is absent, :2764-2766) on configs[0]: ~1 k contigs (4 chr x 25 Mb, mean 100 kb, CV 0.3), 1 M Hi-C pairs, nchrs = 4,
from a real FASTA + .pairs file; wall time of the whole command, and per stage from its own log lines: ingest pairs/s
("Hi-C linking matrix was constructed in"), MCL rounds/s ("round(s) of Markov clustering finished in").  Dev container only
(needs /root/reference; pysam / portion stubbed as in SURVEY App. B).  One JSON object:
    python tools/reference_c1_baseline.py > profiles/r05_reference_c1_baseline.json
bench.py carries it in cpu_baseline.reference_python.c1_haphic_cluster."""
import json
import os
import re
import sys
import tempfile
import time
import types

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
for name, attrs in (('pysam', {'set_verbosity': lambda *a, **k: None, 'AlignmentFile': None}), ('portion', {'closed': None, 'empty': None})):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
sys.path.insert(0, '/root/reference/scripts')
import HapHiC_cluster as H  # noqa: E402
from haphic_amd import synth  # noqa: E402


def main():
    import logging
    gen = synth.make_genome(4, 25_000_000, 100_000, cv=0.3, min_len=5000, seed=12345)           # = tests/golden/make_golden.py gen_pipeline_c1
    id1, p1, id2, p2 = [t.numpy() for t in synth.sample_pairs(gen, 1_000_000, seed=12345)]
    rng = np.random.default_rng(1)
    msgs = []
    handler = logging.Handler()
    handler.emit = lambda rec: msgs.append(rec.getMessage())
    H.logger.addHandler(handler)
    out = {'host': os.uname().nodename, 'host_cpus': os.cpu_count(), 'python': sys.version.split()[0], 'contigs': int(gen.n), 'pairs': 1_000_000,
           'nchrs': 4, 'INTEL_MKL': bool(H.INTEL_MKL),
           'note': 'reference = /root/reference/scripts/HapHiC_cluster.py, run(args) unmodified and unpatched (MKL absent -> its own dense numpy mode); '
                   'the Python loops use 1 core, the dense matrix_power uses the BLAS threads numpy has'}
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
        out['fasta_bytes'], out['pairs_bytes'] = os.path.getsize(fa), os.path.getsize(pairs)
        cwd, argv = os.getcwd(), sys.argv
        os.makedirs(os.path.join(td, 'run'))
        os.chdir(os.path.join(td, 'run'))
        try:
            sys.argv = ['haphic cluster', fa, pairs, '4']                       # every option at the reference's default
            args = H.parse_arguments()
            t0 = time.perf_counter()
            H.run(args)
            out['haphic_cluster_wall_s'] = time.perf_counter() - t0
            out['inflation_dirs'] = sorted(d for d in os.listdir('.') if d.startswith('inflation_'))
        finally:
            os.chdir(cwd)
            sys.argv = argv
    for m_ in msgs:
        g = re.search(r'Hi-C linking matrix was constructed in ([0-9.]+)s', m_)
        if g:
            out['matrix_constructed_s'] = float(g.group(1))
        g = re.search(r'(\d+) round\(s\) of Markov clustering finished in ([0-9.]+)s', m_)
        if g:
            out['mcl_rounds'], out['mcl_s'] = int(g.group(1)), float(g.group(2))
    its = [int(re.search(r'after (\d+) rounds', m_).group(1)) for m_ in msgs if 'rounds of iterations' in m_]
    out['mcl_iterations_total'] = int(sum(its))
    if out.get('mcl_s'):
        out['mcl_iterations_per_s'] = sum(its) / out['mcl_s']
    if out.get('matrix_constructed_s'):
        out['ingest_pairs_per_s_upper_bound'] = 1_000_000 / out['matrix_constructed_s']
        out['ingest_note'] = 'pairs / (time from program start to "Hi-C linking matrix was constructed": FASTA parsing, stat_fragments, the ingest loop, filters, dict_to_matrix)'
    out['log_tail'] = [m_ for m_ in msgs if 'inflation' in m_ or 'finished' in m_][-6:]
    print(json.dumps(out, indent=1))


if __name__ == '__main__':
    main()
