#!/usr/bin/env python3
"""Measurement of the nearby-contact merging (fhx_cni_*): a synthetic table of significant 5 kb cells
(clustered along the diagonal band of 22 chromosomes), arrays handed to the C ABI directly.  One JSON line on stdout.

    python profiles/cni_bench.py [--rows 5000000]          (GPU numbers only)
    python bench.py --path cni                              (the same + cpu_baseline)
"""
import argparse
import gzip
import json
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def table(rng, n_rows, n_chr, span, res):
    import numpy as np
    c = rng.integers(0, n_chr, n_rows).astype(np.int32)
    b1 = rng.integers(0, span, n_rows)
    b2 = np.minimum(b1 + rng.geometric(0.02, n_rows) - 1, span - 1)
    q = np.round(10 ** rng.uniform(-9, -2, n_rows), 12)
    cc = rng.integers(1, 300, n_rows).astype(np.int64)
    return c, (b1 + 1) * res, (b2 + 1) * res, cc, q / 10, q


def measure(rows=5_000_000):
    """-> (result dict, table arrays); the CPU baseline is bench.py's business (oracle/ is imported there only)."""
    import numpy as np
    from fithic_amd import _capi
    rng = np.random.default_rng(7)
    res = 5000
    c, n1, n2, cc, p, q = table(rng, rows, 22, 30000, res)
    cn = _capi.CniContext(0)
    cn.load(c[:1000], n1[:1000], n2[:1000], cc[:1000], p[:1000], q[:1000], res)      # warm-up
    cn.run()
    t0 = time.perf_counter()
    nodes = cn.load(c, n1, n2, cc, p, q, res)
    t_load = time.perf_counter() - t0
    t0 = time.perf_counter()
    rec, info = cn.run(8, 100, 2, 0)
    t_run = time.perf_counter() - t0
    cn.close()
    out = {"metric": "nearby-contact merging (CombineNearbyInteraction path)", "n_gpus": 1, "dtype": "u64 keys / i64 / f64 compare",
           "config": {"workload": "%d significant rows on 22 chromosomes at 5 kb -> %d cells, -c 8 -p 100 -n 2" % (rows, nodes)},
           "info": info.as_dict(), "seconds": {"load_incl_h2d_sort": t_load, "run": t_run},
           "value": nodes / (t_load + t_run), "unit": "cells/s"}
    return out, (c, n1, n2, cc, p, q, res)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=5_000_000)
    args = ap.parse_args()
    out, _ = measure(args.rows)
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
