#!/usr/bin/env python3
"""Per-kernel FETCH_SIZE / WRITE_SIZE from two rocprofv3 --pmc runs (rocpd SQLite)."""
import re
import sqlite3
import sys


def short(name):
    """kernel name without its parameter list; short template arguments are kept (k2_queue<...4> -> k2_queue<4>)"""
    name = re.sub(r"^void ", "", name)
    depth, cut = 0, len(name)
    for i in range(len(name) - 1, -1, -1):          # drop the trailing "(...)" parameter list
        if name[i] == ")":
            depth += 1
        elif name[i] == "(":
            depth -= 1
            if depth == 0:
                cut = i
                break
    name = name[:cut]
    m = re.search(r"<(.*)>$", name)
    if m:
        arg = re.sub(r"\(fhx::dev::BranchClass\)", "", m.group(1))
        name = name[:m.start()] + ("<" + arg + ">" if len(arg) <= 24 else "<...>")
    return name[-44:]


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
        k = short(name)
        prev = out.setdefault(k, {}).get(counter, (0.0, 0))
        out[k][counter] = (prev[0] + total, prev[1] + n)
    return out


def main(fetch_db, write_db):
    f = per_kernel(fetch_db)
    w = per_kernel(write_db)
    print("# FETCH_SIZE / WRITE_SIZE are in KiB.  gfx950 correction (MI355X_MICROARCH.md, HBM section; re-checked here on K1, whose")
    print("# traffic is known exactly: 12 B/row of 16-B/lane loads): FETCH_SIZE counts half of the bytes read -> hbm_read = 2*FETCH*1024.")
    print("%-44s %8s %16s %16s %18s %18s" % ("kernel", "launches", "FETCH_SIZE(sum)", "WRITE_SIZE(sum)", "read MB/launch", "write MB/launch"))
    for k in sorted(set(f) | set(w)):
        if "fhx::" not in k and "krd::" not in k and "cnd::" not in k:
            continue
        fv = f.get(k, {}).get("FETCH_SIZE", (0, 0))
        wv = w.get(k, {}).get("WRITE_SIZE", (0, 0))
        n = max(fv[1], wv[1], 1)
        print("%-44s %8d %16.1f %16.1f %18.1f %18.1f" % (k, n, fv[0], wv[0], 2 * fv[0] * 1024 / n / 1e6, wv[0] * 1024 / n / 1e6))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
