#!/usr/bin/env python3
"""Per-kernel counter totals of the tools/pmc_calib runs (gpurun_out/calib_*/p_counter_collection.csv) beside the known
byte counts the program printed (gpurun_out/calib_plain.log): the calibration table DESIGN.md §6 quotes."""
import collections
import csv
import json
import os
import sys


def main(root):
    known = []
    for line in open(os.path.join(root, 'calib_plain.log')):
        if line.startswith('{'):
            known.append(json.loads(line))
    per = collections.defaultdict(lambda: collections.defaultdict(list))      # launch ordinal -> counter -> values
    for d in sorted(os.listdir(root)):
        f = os.path.join(root, d, 'p_counter_collection.csv')
        if not (d.startswith('calib_') and os.path.exists(f)):
            continue
        rows = [r for r in csv.DictReader(open(f)) if 'k_read' in r['Kernel_Name'] or 'k_write' in r['Kernel_Name']]
        by_dispatch = collections.OrderedDict()
        for r in rows:
            by_dispatch.setdefault(int(r['Dispatch_Id']), []).append(r)
        for ordinal, (_, rs) in enumerate(sorted(by_dispatch.items())):
            for r in rs:
                per[ordinal][r['Counter_Name']].append(float(r['Counter_Value']))
    names = sorted({c for o in per.values() for c in o})
    print('launch, kernel, buffer_bytes, reps, known_bytes, GB/s, ' + ', '.join(names) + ', FETCH_SIZE*1024/known, WRITE_SIZE*1024/known')
    for o, k in enumerate(known):
        kb = k.get('bytes_read', k.get('bytes_written'))
        c = {n: sum(per[o].get(n, [])) for n in names}
        fr = c.get('FETCH_SIZE', 0) * 1024.0 / kb if kb else 0
        wr = c.get('WRITE_SIZE', 0) * 1024.0 / kb if kb else 0
        print('%d, %s, %d, %d, %.4g, %.0f, %s, %.3f, %.3f' % (o, k['kernel'], k['buffer_bytes'], k.get('reps', 1), kb, k['GBs'],
                                                            ', '.join('%.5g' % c[n] for n in names), fr, wr))


if __name__ == '__main__':
    main(sys.argv[1] if len(sys.argv) > 1 else 'gpurun_out')
