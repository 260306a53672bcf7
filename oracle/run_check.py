"""oracle/run_check.py - TEST INFRASTRUCTURE ONLY (used by tests/ and by bench.py's checking leg, never by fithic_amd).

Checks a finished engine pass against the oracle without re-running the whole oracle pipeline on 10^8 rows:
  (1) p of a sample of rows against the oracle's Cephes bdtrc (cephes_oracle.c) fed with the ENGINE's own fit - its spline
      table, sums and interChrProb - and the rows' biases: the per-pair branch table of fithic/fithic.py:1057-1116;
  (2) q of EVERY row against the oracle's Benjamini-Hochberg (myStats.py:24-48; pruned evaluation, proved and tested equal
      to the plain restatement) of the engine's p.
"""
import time

import numpy as np

from . import fithic_oracle as fo


def check_engine_run(eng, genome, sample, cfg, info, with_bias, p_stride=1):
    """sample = {"rows": row numbers, "cols": [chr1, mid1, chr2, mid2, count] of those rows, "chroms": chromosome ids they
    touch}; cfg = {"res", "L", "U", "mode"}; info = (fit-info dict, stats dict) of the pass.  Every p_stride-th sample row
    is evaluated."""
    from fithic_amd import _capi
    fo.build()
    t0 = time.perf_counter()
    v = eng.fetch(p=True, q=True)
    info, st = info
    res = cfg["res"]
    c1, m1, c2, m2, cnt = [a[::p_stride] for a in sample["cols"]]
    rows = sample["rows"][::p_stride]
    bias = {c: (np.where((genome.bias(c) < 0.5) | (genome.bias(c) > 2.0), -1.0, genome.bias(c)) if with_bias
                else np.ones(genome.n_loci[c])) for c in sample["chroms"]}
    b1 = np.empty(len(rows))
    b2 = np.empty(len(rows))
    for c in sample["chroms"]:
        s1 = c1 == c
        b1[s1] = bias[c][m1[s1] // res]
        s2 = c2 == c
        b2[s2] = bias[c][m2[s2] // res]
    inter = c1 != c2
    d = np.abs(m1.astype(np.int64) - m2.astype(np.int64))
    mode = cfg["mode"]
    want = np.ones(len(rows))
    discard = ((b1 < 0) | (b2 < 0)) & ~inter
    if mode != "interOnly":
        table_x = eng.ctx.get_array(_capi.A_TABLE_X).astype(np.float64)
        table_y = eng.ctx.get_array(_capi.A_TABLE_Y)
        xs = eng.ctx.get_array(_capi.A_X)
        sel = np.flatnonzero(~discard & ~inter & fo.in_range(d, cfg["L"], cfg["U"]))
        look = np.minimum(np.maximum(d[sel].astype(np.float64), xs.min()), xs.max())
        idx = np.minimum(np.searchsorted(table_x, look, side="left"), len(table_x) - 1)
        want[sel] = fo.bdtrc(cnt[sel].astype(np.float64) - 1, float(st["in_range_sum"]), table_y[idx] * (b1[sel] * b2[sel]))
    if mode in ("All", "interOnly"):
        sel = np.flatnonzero(~discard & (inter if mode == "All" else np.ones(len(rows), bool)))
        want[sel] = fo.bdtrc(cnt[sel].astype(np.float64) - 1, float(st["inter_sum"]), info["inter_chr_prob"] * (b1[sel] * b2[sel]))
    got = v["p"][rows]
    nan_equal = bool(np.array_equal(np.isnan(got), np.isnan(want)))
    dp = float(np.nanmax(np.abs(np.where(np.isnan(got), 0, got) - np.where(np.isnan(want), 0, want)))) if len(rows) else 0.0
    q_ref = fo.benjamini_hochberg_pruned(v["p"], info["bh_total_tests"])
    qn = np.isnan(q_ref)
    dq = float(np.max(np.abs(np.where(qn, 0, v["q"]) - np.where(qn, 0, q_ref)))) if len(q_ref) else 0.0
    nan_equal = nan_equal and bool(np.array_equal(np.isnan(v["q"]), qn))
    return {"max_dp": dp, "rows_p": int(len(rows)), "max_dq": dq, "rows_q": int(len(q_ref)), "nan_pattern_equal": nan_equal,
            "tolerance": 1e-10, "ok": bool(dp <= 1e-10 and dq <= 1e-10 and nan_equal),
            "p_bit_identical_frac": float(np.mean(got.view(np.int64) == want.view(np.int64))) if len(rows) else 1.0,
            "how": "p: oracle Cephes bdtrc on the sampled rows with the engine's own fit table; q: oracle BH of the engine's p, all rows",
            "seconds": time.perf_counter() - t0}


