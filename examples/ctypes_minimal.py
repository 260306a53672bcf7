#!/usr/bin/env python3
"""The C ABI of include/fithic_mi355x.h driven with nothing but ctypes + numpy (no fithic_amd package): the binding a
maintainer of another host language would write, as a runnable program.  One small spline pass on GPU 0:

    python examples/ctypes_minimal.py [path/to/libfithic_mi355x.so]

Input: two synthetic chromosomes at 10 kb, ~2x10^4 contact rows.  Prints the sums of read_Interactions, the fit
diagnostics and the smallest p / q values.
"""
import ctypes
import os
import sys

import numpy as np

LIB = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "fithic_amd", "libfithic_mi355x.so")
i64, i32, f64 = ctypes.c_int64, ctypes.c_int32, ctypes.c_double


class Params(ctypes.Structure):          # fhx_params
    _fields_ = [("resolution", i64), ("dist_low", i64), ("dist_up", i64), ("n_bins", i32), ("mapp_thres", i32), ("mode", i32),
                ("totals", i32), ("bias_low", f64), ("bias_up", f64)]


class Stats(ctypes.Structure):           # fhx_stats
    _fields_ = [(k, i64) for k in ("n_rows", "inter_count", "inter_sum", "intra_all_count", "intra_all_sum", "in_range_count",
                                   "in_range_sum", "max_count", "n_dist", "n_skipped")]


class FitInfo(ctypes.Structure):         # fhx_fit_info
    _fields_ = [("n_bins_made", i32), ("n_knots", i32), ("spline_ier", i32), ("spline_restarted", i32), ("n_table", i64),
                ("n_frags", i64), ("possible_intra_in_range", i64), ("possible_inter_all", f64), ("possible_intra_all", f64),
                ("max_possible_dist", f64), ("inter_chr_prob", f64), ("baseline_intra_prob", f64), ("spline_s", f64),
                ("spline_fp", f64), ("residual", f64), ("bh_total_tests", f64), ("outlier_thres", f64), ("totals", i32),
                ("totals_narrowed", i32), ("bdtrc_n_intra", i64), ("bdtrc_n_inter", i64)]


def ptr(a, t):
    return a.ctypes.data_as(ctypes.POINTER(t))


def main(lib=LIB, with_inputs=False):
    L = ctypes.CDLL(lib)
    L.fhx_last_error.restype = ctypes.c_char_p
    L.fhx_last_error.argtypes = [ctypes.c_void_p]
    ctx = ctypes.c_void_p()

    def check(rc):
        if rc != 0:
            raise SystemExit("fhx error %d: %s" % (rc, (L.fhx_last_error(ctx) or b"").decode()))

    check(L.fhx_create(0, ctypes.byref(ctx)))                                            # GPU 0
    res = 10000
    check(L.fhx_set_params(ctx, ctypes.byref(Params(res, 20000, 1500000, 30, 1, 0, 0, 0.5, 2.0))))
    # fragments (fithic.py:581-590): chromosome ids are small ints interned by the caller; sort rank = sorted(names) order
    rng = np.random.default_rng(1)
    n_loci = [300, 200]
    f_chr = np.concatenate([np.full(n, c, np.int32) for c, n in enumerate(n_loci)])
    f_mid = np.concatenate([np.arange(n, dtype=np.int32) * res + res // 2 for n in n_loci])
    f_hits = np.ones(len(f_chr), np.int32)
    rank = np.array([0, 1], np.int32)
    check(L.fhx_load_fragments(ctx, ptr(f_chr, i32), ptr(f_mid, i32), ptr(f_hits, i32), i64(len(f_chr)), ptr(rank, i32), i32(2)))
    bias = np.exp(rng.normal(0, 0.25, len(f_chr)))
    check(L.fhx_load_bias(ctx, ptr(f_chr, i32), ptr(f_mid, i32), ptr(bias, f64), i64(len(f_chr))))
    # contact rows (fithic.py:406-417)
    rows = []
    off = 0
    for c, n in enumerate(n_loci):
        for i in range(n):
            for j in range(i + 1, min(n, i + 160)):
                cnt = rng.poisson(40.0 * bias[off + i] * bias[off + j] / (j - i) ** 1.05)
                if cnt:
                    rows.append((c, i * res + res // 2, c, j * res + res // 2, cnt))
        off += n
    a = np.array(rows, np.int32)
    cols = [np.ascontiguousarray(a[:, k]) for k in range(5)]
    check(L.fhx_load_pairs(ctx, *[ptr(v, i32) for v in cols], i64(len(a))))
    st, info = Stats(), FitInfo()
    check(L.fhx_pass_stats(ctx, ctypes.byref(st)))                                       # K1: read_Interactions
    check(L.fhx_fit(ctx, ctypes.byref(info)))                                            # bins, possible pairs, spline, table
    check(L.fhx_pvalues(ctx))                                                            # K2: prior + bdtrc per row
    check(L.fhx_bh(ctx, f64(info.bh_total_tests)))                                       # K3: benjamini_hochberg_correction
    p, q = np.empty(len(a)), np.empty(len(a))
    check(L.fhx_fetch(ctx, ptr(p, f64), ptr(q, f64), None, None, None))
    print("rows %d  observedIntraInRangeSum %d  bins %d  knots %d  N %.0f" % (st.n_rows, st.in_range_sum, info.n_bins_made, info.n_knots,
                                                                            info.bh_total_tests))
    k = np.nanargmin(p)
    print("smallest p %.3e (q %.3e) at row %d: count %d, distance %d" % (p[k], q[k], k, cols[4][k], cols[3][k] - cols[1][k]))
    L.fhx_destroy(ctx)
    if with_inputs:                          # (the test holds p and q against the oracle run on the same tables)
        return p, q, dict(cols=cols, f_chr=f_chr, f_mid=f_mid, bias=bias, res=res, L=20000, U=1500000, n_bins=30)
    return p, q


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else LIB)
