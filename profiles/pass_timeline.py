#!/usr/bin/env python3
"""Timeline of the LAST pass in a rocprofv3 --kernel-trace database: every kernel dispatch from the last k1_classify_hist on,
with its duration and the gap to the previous dispatch's end - where a small shard's time goes between the kernels.
    python profiles/pass_timeline.py <results.db>"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
rows = list(cur.execute("select s.kernel_name, d.start, d.end from %s d join %s s on d.kernel_id=s.id order by d.start" % (kd, ks)))
last = max(i for i, r in enumerate(rows) if "k1_classify_hist" in r[0] or "nf_k1_classify" in r[0])
seg = rows[last:]
t0 = seg[0][1]
prev_end = None
busy = 0
for name, a, b in seg:
    gap = (a - prev_end) / 1e3 if prev_end is not None else 0.0
    short = name.split("(")[0].replace("void ", "").replace("fhx::", "")[:46]
    print("%9.1f us  +%7.1f us gap  %8.1f us  %s" % ((a - t0) / 1e3, gap, (b - a) / 1e3, short))
    busy += b - a
    prev_end = max(prev_end or b, b)
print("pass span %.1f us, kernels %.1f us, %d dispatches" % ((prev_end - t0) / 1e3, busy / 1e3, len(seg)))
