#!/usr/bin/env python3
"""How much of the converging continued-fraction classes is the loop itself: k_debug_contfrac (loop only, lazy variant) on
2.2e7 realistic (a, b, x) triples of the BCF class, rows grouped by count as k2_queue_by_count hands them to the waves.
Run under rocprofv3 --kernel-trace --stats and read k_debug_contfrac's duration next to k2_queue_by_count<2>'s."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    from fithic_amd import _capi
    rng = np.random.default_rng(5)
    n_total = 4.4e8
    m = 22_000_000
    c = np.minimum(2 + rng.geometric(0.35, m), 60).astype(np.float64)           # counts >= 2, small ones dominate
    hi = (c - 1.0) / (n_total - 1.0)
    lo = 1.0 / (n_total - c + 1.0)
    x = lo + (hi - lo) * rng.uniform(0.05, 0.95, m)
    # tiles of 1024 rows sorted by count (what the kernel's LDS counting sort does)
    order = np.arange(m).reshape(-1, 1024 if m % 1024 == 0 else 1000)
    key = c.reshape(order.shape)
    idx = np.argsort(key, axis=1, kind="stable")
    c = np.take_along_axis(key, idx, axis=1).ravel()
    x = np.take_along_axis(x.reshape(order.shape), idx, axis=1).ravel()
    a = c
    b = n_total - c + 1.0
    ctx = _capi.Context(0)
    for lazy in (1, 0):
        out = ctx.debug_contfrac(0, lazy, a, b, x)
        print("kind 0 lazy %d: %d values, mean %.6g" % (lazy, len(out), float(np.mean(out))))
    ctx.close()


if __name__ == "__main__":
    main()
