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
    m, g = d['mcl'], d['ingest']
    print('pairs/s %.3e  ingest_ms %.1f | mcl it/s %.2f  ms_per_mcl %.1f iters %d | step ms %.1f' % (
        d['value'], g['ms_per_step'], m['iters_per_s'], m['ms_per_mcl'], m['iterations'], d['ms_per_step']))
    print('  mcl kernels ms/step', {k: round(v, 1) for k, v in m['kernel_ms_per_step'].items()})
    print('  ingest kernels ms/step', {k: round(v, 2) for k, v in g['kernels_ms_per_step'].items()})
    r = d['roofline']
    if r:
        print('  roofline %s: %.0f GB/s (frac %.3f) avg launch %.2f ms x %.0f/step, traffic %s' % (
            r['kernel'], r['achieved'], r['frac'], r['avg_launch_ms'], r['launches_per_step'], r['traffic']))
    r = g['roofline']
    if r:
        print('  ingest roofline %s: %.0f GB/s (frac %.3f) launch %.2f ms, traffic %s' % (r['kernel'], r['achieved'], r['frac'], r['avg_launch_ms'], r['traffic']))
    if 'parity' in d:
        print('  parity', d['parity'])
    if d.get('sweep'):
        print('  sweep', json.dumps(d['sweep'])[:1800])
    if d.get('transport'):
        print('  transport', d['transport'])
    t = g.get('text')
    if t:
        print('  text: parse %.2e pairs/s, with bed %.2e | file->matrix %.2e pairs/s (%.1f GB file), with alignments.bed %.2e pairs/s' % (
            t['parse']['pairs_per_s'], t['parse_bed']['pairs_per_s'], t['file_to_link_matrix']['pairs_per_s'], t['file_to_link_matrix']['file_bytes'] / 1e9,
            t['file_to_link_matrix_with_bed']['pairs_per_s']))
    if 'cpu_baseline' in d:
        c = dict(d['cpu_baseline'])
        rp = c.pop('reference_python', None)
        print('  cpu_baseline', c)
        if rp:
            print('  reference python: ingest %.0f pairs/s, mcl %.2f it/s at n=%d (%s cores of %s)' % (rp['ingest']['pairs_per_s'], rp['mcl']['iters_per_s'], rp['mcl']['n'], rp['cores_used'], rp['host_cpus']))
