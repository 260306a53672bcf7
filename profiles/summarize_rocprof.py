#!/usr/bin/env python3
"""Condense a rocprofv3 --kernel-trace --stats result (rocpd SQLite .db) into a short per-kernel table.
Usage: python profiles/summarize_rocprof.py gpurun_out/prof_xx/name_results.db > profiles/rNN_xxx_kernel_stats.txt"""
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


def main(path):
    con = sqlite3.connect(path)
    rows = list(con.execute("select name, total_calls, total_duration, average, percentage from top_kernels"))
    print("# source: %s   (durations in microseconds; rocprofv3 --kernel-trace --stats)" % path.split("/")[-1])
    print("%-40s %7s %14s %12s %7s" % ("kernel", "calls", "total_us", "avg_us", "pct"))
    other = 0.0
    for name, calls, total, avg, pct in rows:
        if "fhx::" in name or "krd::" in name or "cnd::" in name or "fhxscan::" in name:
            print("%-40s %7d %14.1f %12.1f %6.2f%%" % (short(name), calls, total, avg, pct))
        else:
            other += pct
    print("%-40s %7s %14s %12s %6.2f%%" % ("(torch data generation, untimed)", "", "", "", other))


if __name__ == "__main__":
    main(sys.argv[1])
