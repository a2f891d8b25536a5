#!/usr/bin/env python3
"""Turn rocprofv3 (ROCm 7.2, rocpd sqlite output) results into the small text
summaries committed under profiles/.

  python tools/rocprof_summary.py stats  <results.db>            -> kernel stats CSV on stdout
  python tools/rocprof_summary.py pmc    <results.db> [...]      -> per-kernel counter means
"""
import sqlite3
import sys


def stats(db):
    con = sqlite3.connect(db)
    rows = con.execute("select name, count(*), sum(duration), avg(duration), min(duration), max(duration) "
                       "from kernels group by name order by sum(duration) desc").fetchall()
    tot = sum(r[2] for r in rows) or 1
    print("kernel,calls,total_ns,avg_ns,min_ns,max_ns,percent")
    for n, c, t, a, mn, mx in rows:
        print('"%s",%d,%d,%.1f,%d,%d,%.2f' % (n, c, t, a, mn, mx, 100.0 * t / tot))


def pmc(dbs):
    print("kernel,counter,dispatches,mean_value,min_value,max_value,mean_duration_ns")
    for db in dbs:
        con = sqlite3.connect(db)
        rows = con.execute("select kernel_name, counter_name, count(*), avg(value), min(value), max(value), "
                           "avg(duration) from counters_collection group by kernel_name, counter_name "
                           "order by kernel_name").fetchall()
        for r in rows:
            print('"%s",%s,%d,%.3f,%.3f,%.3f,%.1f' % r)


if __name__ == '__main__':
    if sys.argv[1] == 'stats':
        stats(sys.argv[2])
    else:
        pmc(sys.argv[2:])
