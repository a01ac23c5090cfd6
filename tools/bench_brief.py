#!/usr/bin/env python3
"""print the interesting numbers of a bench.py JSON line"""
import json
import sys
for line in open(sys.argv[1]):
    if not line.startswith('{'):
        if 'rror' in line or 'rc=' in line:
            print(line.rstrip()[:300])
        continue
    d = json.loads(line)
    m = d['mcl']
    print('pairs/s %.3e  ingest_ms %.1f  k_ingest_ms %.1f | mcl it/s %.2f  ms_per_mcl %.1f iters %d  kernels %s | step ms %.1f' % (
        d['value'], d['ingest_ms_per_step'], d['roofline']['avg_launch_ms'], m['iters_per_s'], m['ms_per_mcl'], m['iterations'],
        {k: round(v, 1) for k, v in m['kernel_ms_per_step'].items()}, d['ms_per_step']))
    if 'ingest_kernels_ms' in d:
        print('ingest kernels ms/step', {k: round(v, 2) for k, v in d['ingest_kernels_ms'].items()})
    if 'cpu_baseline' in d:
        print('cpu_baseline', d['cpu_baseline'])
