"""oracle/run_check.py - TEST INFRASTRUCTURE ONLY (used by tests/ and by bench.py's checking leg, never by fithic_amd).

Checks a finished engine pass against the oracle without re-running the whole oracle pipeline on 10^8 rows:
  (1) p of a sample of rows against the oracle's Cephes bdtrc (cephes_oracle.c) fed with the ENGINE's own fit - its spline
      table, sums and interChrProb - and the rows' biases: the per-pair branch table of fithic/fithic.py:1057-1116;
  (2) q of EVERY row against the oracle's Benjamini-Hochberg (myStats.py:24-48; pruned evaluation, proved and tested equal
      to the plain restatement) of the engine's p.
"""
import os
import time

import numpy as np

from . import fithic_oracle as fo

GOLDEN = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")


def fit_fixture(config):
    """tests/golden/f14_<config>_fit.npz: what the REAL reference's makeBinsFromInteractions -> generate_FragPairs ->
    calculateProbabilities -> fit_Spline returned on the full-size synth-v1 workload `config` (C3, C3w, C5) - made by
    tests/golden/make_golden.py f14 - or None for a workload without one."""
    path = os.path.join(GOLDEN, "f14_%s_fit.npz" % config)
    return np.load(path) if os.path.exists(path) else None


def _bits(a, b):
    a, b = np.ascontiguousarray(a, np.float64), np.ascontiguousarray(b, np.float64)
    return a.shape == b.shape and bool(np.array_equal(a.view(np.int64), b.view(np.int64)))


def compare_histogram(hist_sumcc, hist_npairs, stats, g, res):
    """The engine's K1 output against the fixture's histogram (computed with torch, without the engine).  -> list of names
    that differ (empty = equal)."""
    bad = []
    idx = g["hist_dist_idx"]
    want_cc = np.zeros(max(len(hist_sumcc), int(idx.max()) + 1), np.int64)
    want_np = np.zeros_like(want_cc)
    want_cc[idx] = g["hist_sumcc"]
    want_np[idx] = g["hist_nrows"]
    got_cc = np.zeros_like(want_cc)
    got_np = np.zeros_like(want_cc)
    got_cc[:len(hist_sumcc)] = hist_sumcc
    got_np[:len(hist_npairs)] = hist_npairs
    for name, got, want in (("hist_sumcc", got_cc, want_cc), ("hist_npairs", got_np, want_np)):
        if not np.array_equal(got, want):
            at = np.flatnonzero(got != want)
            bad.append("%s (%d entries, first at distance index %d: %d, fixture %d)" % (name, len(at), at[0], got[at[0]], want[at[0]]))
    inter_count, inter_sum, intra_all_sum, in_range_sum = [int(v) for v in g["sums"]]
    for k, v in (("inter_count", inter_count), ("inter_sum", inter_sum), ("intra_all_sum", intra_all_sum), ("in_range_sum", in_range_sum)):
        if int(stats[k]) != v:
            bad.append(k)
    return bad


def compare_fit(get_array, info, g):
    """The engine's host fit (fhx_fit) against the reference's stage outputs of fixture g, bit for bit: bins, possible pairs,
    bin means, s, knots, coefficients, spline table before and after the isotonic regression, N.  get_array = Context.get_array,
    info = the fhx_fit_info as a dict.  -> list of names that differ (empty = identical)."""
    from fithic_amd import _capi as A
    bad = []
    for k, w in (("lb", A.A_BIN_LB), ("ub", A.A_BIN_UB), ("s1", A.A_BIN_POSS), ("s2", A.A_BIN_SUMCC), ("s7", A.A_BIN_POSS7)):
        if not np.array_equal(get_array(w), g["bins1_" + k]):
            bad.append("bins1_" + k)
    if not np.array_equal(get_array(A.A_BIN_POSS0), g["bins0_s1"]):
        bad.append("bins0_s1")
    if not _bits(get_array(A.A_BIN_SUMDIST), g["bins1_s3"]):
        bad.append("bins1_s3")
    mine = np.array([info["n_frags"], info["max_possible_dist"], info["possible_intra_in_range"], info["possible_inter_all"],
                     info["inter_chr_prob"], info["baseline_intra_prob"]], np.float64)
    if not _bits(mine, g["frag_scalars"]):
        bad.append("frag_scalars")
    for k, w in (("x", A.A_X), ("y", A.A_Y), ("spl_t", A.A_KNOTS), ("spl_c", A.A_COEFFS), ("splineY", A.A_TABLE_Y0),
                 ("newSplineY", A.A_TABLE_Y)):
        if not _bits(get_array(w), g[k]):
            bad.append(k)
    if not np.array_equal(get_array(A.A_TABLE_X), g["splineX"]):
        bad.append("splineX")
    s, fp, ier = g["spl_s_fp_ier"]
    if not (info["spline_s"] == s and info["spline_fp"] == fp and info["spline_ier"] == int(ier)):
        bad.append("spl_s_fp_ier")
    if info["residual"] != g["residual"][0]:
        bad.append("residual")
    if 1.0 / info["bh_total_tests"] != g["outlierThres"][0]:
        bad.append("outlierThres")
    return bad


class _StreamedColumns:
    """The engine's p and q columns read in chunks of `chunk` rows through torch tensors on the device (copied out of the
    engine's buffers chunk by chunk): what lets a 2e9-row run be checked without 32 GB of host arrays.  The device only
    counts, compares with constants and compacts here - torch operators, none of the engine's kernels; every BH value
    that is compared is computed by the oracle on the host."""

    def __init__(self, eng, torch, n, chunk):
        self.eng, self.torch, self.n, self.chunk = eng, torch, int(n), int(chunk)
        dev = torch.device("cuda", eng.ctx.device)
        self.buf = [torch.empty(min(self.n, self.chunk), dtype=torch.float64, device=dev) for _ in range(2)]

    def chunks(self, which=(0, 1)):
        """-> (lo, hi, [tensor of column w for w in which]); the tensors are reused by the next step"""
        eng = self.eng
        for lo in range(0, self.n, self.chunk):
            hi = min(self.n, lo + self.chunk)
            out = []
            for w in which:
                eng.ctx.memcpy_d2d(self.buf[w].data_ptr(), eng.ctx.device_ptr(w) + 8 * lo, 8 * (hi - lo))
                out.append(self.buf[w][:hi - lo])
            yield lo, hi, out


def _p_and_q_streamed(eng, torch, n, n_tests, sample_rows, chunk):
    """-> (p of the sample rows, dict with the q verdict of EVERY row).  Three sweeps over the columns: NaN count, the oracle's
    pruning threshold (fo.bh_prune_threshold with the counts taken on the device), then per chunk the sampled p, the rows at
    or above the threshold (q must be exactly 1, NaN rows NaN) and the survivors, which go to the host for the oracle's BH."""
    cols = _StreamedColumns(eng, torch, n, chunk)
    N = float(n_tests)
    if not N > 0.0:
        raise ValueError("streamed check needs N > 0 (nothing can be pruned otherwise)")
    n_nan = 0
    for lo, hi, (p_t,) in cols.chunks((0,)):
        n_nan += int(torch.isnan(p_t).sum())
    sweeps = [0]

    def count_below(t):
        sweeps[0] += 1
        c = 0
        for lo, hi, (p_t,) in cols.chunks((0,)):
            c += int((p_t < t).sum())
        return c

    tau = fo.bh_prune_threshold(n - n_nan, N, count_below)
    rows_t = torch.as_tensor(np.ascontiguousarray(sample_rows, np.int64), device=cols.buf[0].device)
    got = np.empty(len(sample_rows), np.float64)
    order = np.argsort(sample_rows, kind="stable")
    sorted_rows = np.asarray(sample_rows)[order]
    bad_pruned = 0
    nan_equal = True
    surv_rows, surv_p, surv_q = [], [], []
    for lo, hi, (p_t, q_t) in cols.chunks((0, 1)):
        a, b = np.searchsorted(sorted_rows, lo), np.searchsorted(sorted_rows, hi)
        if b > a:
            got[order[a:b]] = p_t[rows_t[order[a:b]] - lo].cpu().numpy()
        nan = torch.isnan(p_t)
        nan_equal = nan_equal and bool(torch.equal(nan, torch.isnan(q_t)))
        keep = p_t < tau
        bad_pruned += int(((q_t != 1.0) & ~keep & ~nan).sum())
        idx = torch.nonzero(keep).squeeze(1)
        if idx.numel():
            surv_rows.append((idx + lo).cpu().numpy())
            surv_p.append(p_t[idx].cpu().numpy())
            surv_q.append(q_t[idx].cpu().numpy())
    sp = np.concatenate(surv_p) if surv_p else np.empty(0)
    sq = np.concatenate(surv_q) if surv_q else np.empty(0)
    q_ref = fo.bh_of_survivors(sp, N)                  # chunks come in row order: the stable order of equal values is the reference's
    dq = float(np.max(np.abs(sq - q_ref))) if len(sp) else 0.0
    if bad_pruned:
        dq = max(dq, 1.0)
    return got, {"max_dq": dq, "rows_q": int(n), "nan_pattern_equal": bool(nan_equal), "q_rows_ranked_by_oracle": int(len(sp)),
                 "q_rows_pruned_not_one": int(bad_pruned), "streamed": {"chunk_rows": int(chunk), "threshold_sweeps": sweeps[0], "tau": float(tau)}}


def check_engine_run(eng, genome, sample, cfg, info, with_bias, p_stride=1, fit_fixture_name=None, torch=None, chunk_rows=1 << 27):
    """sample = {"rows": row numbers, "cols": [chr1, mid1, chr2, mid2, count] of those rows, "chroms": chromosome ids they
    touch}; cfg = {"res", "L", "U", "mode"}; info = (fit-info dict, stats dict) of the pass.  Every p_stride-th sample row
    is evaluated.  fit_fixture_name: the f14 fixture of this workload (the run must be the full-size synth-v1 workload of that
    name): the engine's K1 histogram and its fit are then compared with the reference's before the table is used.
    torch: given, p and q are read in chunks of chunk_rows rows through device tensors (_p_and_q_streamed: any size); without
    it both columns are fetched whole (16 B/row of host memory)."""
    from fithic_amd import _capi
    fo.build()
    t0 = time.perf_counter()
    info, st = info
    # the oracle mode that restates what the engine was asked for: "reference" = bdtrc's n narrowed to a C int as scipy does
    # (pinned by the f15 fixtures), "wide" = the true total (the oracle's second mode - nothing pins it above 2^31)
    totals = fo.TOTALS_WIDE if info.get("totals", 0) == _capi.TOTALS_WIDE else fo.TOTALS_REFERENCE
    v = eng.fetch(p=True, q=True) if torch is None else None
    fit_diff = hist_diff = None
    if fit_fixture_name:
        g = fit_fixture(fit_fixture_name)
        if g is None:
            raise FileNotFoundError("tests/golden/f14_%s_fit.npz" % fit_fixture_name)
        hist_diff = compare_histogram(eng.ctx.get_array(_capi.A_HIST_SUMCC), eng.ctx.get_array(_capi.A_HIST_NPAIRS), st, g, cfg["res"])
        fit_diff = compare_fit(eng.ctx.get_array, info, g)
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
        want[sel] = fo.bdtrc(cnt[sel].astype(np.float64) - 1, float(st["in_range_sum"]), table_y[idx] * (b1[sel] * b2[sel]), totals)
    if mode in ("All", "interOnly"):
        sel = np.flatnonzero(~discard & (inter if mode == "All" else np.ones(len(rows), bool)))
        want[sel] = fo.bdtrc(cnt[sel].astype(np.float64) - 1, float(st["inter_sum"]), info["inter_chr_prob"] * (b1[sel] * b2[sel]), totals)
    extra = {}
    if torch is not None:
        got, qv = _p_and_q_streamed(eng, torch, int(eng.n_rows), info["bh_total_tests"], rows, chunk_rows)
        dq, n_q, q_nan_equal = qv.pop("max_dq"), qv.pop("rows_q"), qv.pop("nan_pattern_equal")
        extra = qv
    else:
        got = v["p"][rows]
        q_ref = fo.benjamini_hochberg_pruned(v["p"], info["bh_total_tests"])
        qn = np.isnan(q_ref)
        dq = float(np.max(np.abs(np.where(qn, 0, v["q"]) - np.where(qn, 0, q_ref)))) if len(q_ref) else 0.0
        n_q, q_nan_equal = int(len(q_ref)), bool(np.array_equal(np.isnan(v["q"]), qn))
    nan_equal = bool(np.array_equal(np.isnan(got), np.isnan(want))) and q_nan_equal
    dp = float(np.nanmax(np.abs(np.where(np.isnan(got), 0, got) - np.where(np.isnan(want), 0, want)))) if len(rows) else 0.0
    out = {"max_dp": dp, "rows_p": int(len(rows)), "max_dq": dq, "rows_q": n_q, "nan_pattern_equal": nan_equal,
           "tolerance": 1e-10, "ok": bool(dp <= 1e-10 and dq <= 1e-10 and nan_equal),
           "p_bit_identical_frac": float(np.mean(got.view(np.int64) == want.view(np.int64))) if len(rows) else 1.0}
    out.update(extra)
    narrowed = int(info.get("totals_narrowed", 0))
    out["totals_semantics"] = totals
    out["oracle_mode"] = ("fho_bdtrc (n narrowed to a C int like scipy; pinned by tests/golden/f15_*)" if totals == fo.TOTALS_REFERENCE
                          else "fho_bdtrc_wide (true total)" + ("; a total is >= 2^31: NOT what the reference computes there, unpinned"
                                                                if narrowed else "; totals < 2^31: the same function as the reference's"))
    out["totals_at_or_above_2p31"] = [w for b, w in ((1, "observedIntraInRangeSum"), (2, "observedInterAllSum")) if narrowed & b]
    out["p_nan_in_sample"] = int(np.isnan(want).sum())
    if fit_fixture_name:
        out["fit_vs_reference"] = {"fixture": "tests/golden/f14_%s_fit.npz" % fit_fixture_name, "k1_histogram_differs": hist_diff,
                                   "fit_differs": fit_diff, "bit_identical": not hist_diff and not fit_diff}
        out["ok"] = bool(out["ok"] and not hist_diff and not fit_diff)
        out["how"] = ("K1 histogram and sums == the fixture's (torch bincount of the same rows); bins, possible pairs, x, y, s, knots, "
                      "coefficients, spline table and N bit-identical to what the real reference's makeBinsFromInteractions / "
                      "generate_FragPairs / calculateProbabilities / fit_Spline returned on that histogram (fixture f14); p: oracle Cephes "
                      "bdtrc on the sampled rows with that table; q: oracle BH of the engine's p, all rows")
    else:
        out["how"] = "p: oracle Cephes bdtrc on the sampled rows with the engine's own fit table; q: oracle BH of the engine's p, all rows"
    out["seconds"] = time.perf_counter() - t0
    return out


