#!/usr/bin/env python3
"""Kernel summary of a rocprofv3 results database (rocpd sqlite): name, launches, total and average ms.
    python profiles/rocpd_kernels.py gpurun_out/prof/x_results.db [top]"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
q = ("select s.kernel_name, count(*), sum(d.end-d.start)/1e6, avg(d.end-d.start)/1e6 from %s d join %s s on d.kernel_id=s.id "
     "group by s.kernel_name order by 3 desc limit %d" % (kd, ks, int(sys.argv[2]) if len(sys.argv) > 2 else 16))
for r in cur.execute(q):
    print("%-70s n=%-4d total %9.3f ms  avg %9.3f ms" % (r[0][:70], r[1], r[2], r[3]))
