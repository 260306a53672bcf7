#!/usr/bin/env python3
"""Per-kernel FETCH_SIZE / WRITE_SIZE from two rocprofv3 --pmc runs (rocpd SQLite)."""
import re
import sqlite3
import sys


def per_kernel(path):
    con = sqlite3.connect(path)
    tabs = [r[0] for r in con.execute("select name from sqlite_master where type in ('table','view')")]
    if "counters_collection" not in tabs:
        print("# no counters_collection view in", path, tabs)
        return {}
    cols = [r[1] for r in con.execute("pragma table_info(counters_collection)")]
    print("# columns:", cols)
    out = {}
    q = "select kernel_name, counter_name, sum(value), count(*) from counters_collection group by kernel_name, counter_name"
    for name, counter, total, n in con.execute(q):
        out.setdefault(re.sub(r"\(.*", "", name)[-40:], {})[counter] = (total, n)
    return out


def main(fetch_db, write_db):
    f = per_kernel(fetch_db)
    w = per_kernel(write_db)
    print("%-40s %8s %16s %16s" % ("kernel", "launches", "FETCH_SIZE(sum)", "WRITE_SIZE(sum)"))
    for k in sorted(set(f) | set(w)):
        if "fhx::" not in k:
            continue
        fv = f.get(k, {}).get("FETCH_SIZE", (0, 0))
        wv = w.get(k, {}).get("WRITE_SIZE", (0, 0))
        print("%-40s %8d %16.1f %16.1f" % (k, max(fv[1], wv[1]), fv[0], wv[0]))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
