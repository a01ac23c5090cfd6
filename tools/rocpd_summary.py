#!/usr/bin/env python3
"""Turn a rocprofv3 results .db (rocpd sqlite, the default output of ROCm 7.2's rocprofv3) into the
same per-kernel table `--stats` prints: name, calls, total/avg/min/max duration (us), % of GPU time."""
import sqlite3
import sys


def main(db, out):
    c = sqlite3.connect(db)
    rows = c.execute("select name, count(*), sum(end-start)/1e3, avg(end-start)/1e3, min(end-start)/1e3, "
                     "max(end-start)/1e3 from kernels group by name order by 3 desc").fetchall()
    tot = sum(r[2] for r in rows) or 1.0
    with open(out, 'w') as f:
        f.write('"Name","Calls","TotalDurationUs","AverageUs","MinUs","MaxUs","Percentage"\n')
        for n, k, t, a, mn, mx in rows:
            f.write('"%s",%d,%.3f,%.3f,%.3f,%.3f,%.4f\n' % (n.replace('"', "'"), k, t, a, mn, mx, 100.0 * t / tot))


if __name__ == '__main__':
    main(sys.argv[1], sys.argv[2])
