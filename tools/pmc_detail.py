#!/usr/bin/env python3
"""raw rocprofv3 --pmc counters (gpurun_out/pmc_*/p_counter_collection.csv) of the kernels whose name contains one of the
given needles: per-launch averages, one line per (pass, kernel).  Appended to profiles/rNN_pmc_c3.txt next to the traffic table."""
import collections
import csv
import os
import sys


def main(root, needles):
    for d in sorted(os.listdir(root)):
        f = os.path.join(root, d, 'p_counter_collection.csv')
        if not (d.startswith('pmc_') and os.path.exists(f)):
            continue
        agg = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            k = r['Kernel_Name'].replace('(anonymous namespace)::', '').split('(')[0].replace('void ', '')
            agg[k][r['Counter_Name']].append(float(r['Counter_Value']))
        for k in sorted(agg):
            if any(x in k for x in needles):
                print('# %s %s: %s' % (d, k, ', '.join('%s %.4g (x%d)' % (c, sum(v) / len(v), len(v)) for c, v in sorted(agg[k].items()))))


if __name__ == '__main__':
    main(sys.argv[1], sys.argv[2:])
