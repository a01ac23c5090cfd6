#!/usr/bin/env python3
"""rocprofv3 --pmc counter_collection.csv files (one pass per counter group, gpurun_out/pmc_*/) ->
profiles/pmc_traffic.json + a per-kernel table.  HBM/fabric bytes follow MI355X_MICROARCH.md §HBM:
FETCH_SIZE / WRITE_SIZE are reported in KB; on gfx950 FETCH_SIZE tallies 128-byte read requests at 64 bytes,
so streaming reads are doubled (the TCC_MISS x 128 B figure of the same run is printed beside it as a check)."""
import collections
import csv
import json
import os
import sys


def load(path):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(path)):
        k = r['Kernel_Name'].replace('(anonymous namespace)::', '')
        k = k.split('(')[0].replace('void ', '')
        agg[k][r['Counter_Name']].append(float(r['Counter_Value']))
    return agg


def main(root, contigs, pairs, out_json, out_txt):
    tables = {}
    for d in sorted(os.listdir(root)):
        f = os.path.join(root, d, 'p_counter_collection.csv')
        if d.startswith('pmc_') and os.path.exists(f):
            for k, cs in load(f).items():
                tables.setdefault(k, {}).update(cs)
    rows = []
    per_launch = {}
    for k, cs in tables.items():
        if not (k.startswith('k_') or '<' in k):
            continue
        n = max(len(v) for v in cs.values())
        fetch = sum(cs.get('FETCH_SIZE', [])) * 1024.0 * 2.0          # KB -> B, gfx950 half-count correction
        write = sum(cs.get('WRITE_SIZE', [])) * 1024.0
        miss = sum(cs.get('TCC_MISS_sum', [])) * 128.0
        hit = sum(cs.get('TCC_HIT_sum', []))
        req = sum(cs.get('TCC_REQ_sum', []))
        rows.append((fetch + write, k, n, fetch, write, miss, hit / req if req else float('nan')))
        if fetch + write > 0 and not k.startswith('at::'):
            per_launch[k] = (fetch + write) / n
    rows.sort(reverse=True)
    with open(out_txt, 'w') as f:
        f.write('kernel, launches, read_bytes(FETCH_SIZE*1024*2), write_bytes(WRITE_SIZE*1024), TCC_MISS*128B, L2_hit_rate\n')
        for _, k, n, fetch, write, miss, hr in rows[:40]:
            f.write('%s, %d, %.4g, %.4g, %.4g, %.3f\n' % (k, n, fetch, write, miss, hr))
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from haphic_amd import build
    json.dump({'contigs': int(contigs), 'pairs_per_gpu': int(pairs), 'kernel_source_sha16': build.source_hash(), 'source': 'rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes',
               'bytes_per_launch': per_launch}, open(out_json, 'w'), indent=1)
    print(open(out_txt).read())


if __name__ == '__main__':
    main(*sys.argv[1:6])
