"""Host stages of libfithic_mi355x.so (binning, possible pairs, probabilities, FITPACK spline, PAVA, lbeta tables)
against the golden fixtures - runs on CPU through a host-only context (device = -1)."""
import os
import ctypes

import numpy as np
import pytest

from conftest import load_case, case_args, ALL_CASES, bits_equal, GOLDEN
from fithic_amd import _capi, tables
from fithic_amd.engine import MODES


def test_library_exports_every_declared_symbol():
    import re
    hdr = open(os.path.join(os.path.dirname(GOLDEN), "..", "include", "fithic_mi355x.h")).read()
    declared = set(re.findall(r"\b(fhx_[a-z0-9_]+)\s*\(", hdr))
    declared -= {"fhx_ctx", "fhx_params", "fhx_stats", "fhx_fit_info", "fhx_array"}
    L = ctypes.CDLL(_capi.LIB_PATH)
    for name in sorted(declared):
        assert hasattr(L, name), "libfithic_mi355x.so does not export %s" % name
    assert declared == set(_capi.SYMBOLS), declared ^ set(_capi.SYMBOLS)


def test_kernel_entry_points_fail_loudly_without_device():
    ctx = _capi.Context(-1)
    ctx.set_params(40000)
    for call in (ctx.pass_stats, ctx.pvalues, lambda: ctx.bh(10.0), ctx.next_pass, lambda: ctx.bh_array([0.5], 3)):
        with pytest.raises(_capi.FhxError) as e:
            call()
        assert e.value.code == _capi.FHX_ERR_NO_DEVICE


def test_spline_fit_bit_exact_vs_scipy_fixtures():
    g = np.load(os.path.join(GOLDEN, "f4_fitpack.npz"))
    for name in g["names"]:
        s, fp, ier = g[name + "_sfpier"]
        t, c, fp2, ier2, _ = _capi.host_spline_fit(g[name + "_x"], g[name + "_y"], s)
        assert bits_equal(t, g[name + "_t"]) and bits_equal(c, g[name + "_c"]), name
        assert fp2 == fp and ier2 == int(ier), name
        assert bits_equal(_capi.host_spline_eval(t, c, g[name + "_xe"]), g[name + "_ye"]), name


def test_spline_fit_vs_live_scipy_on_random_bins_if_present():
    """The host fit (C++) AND the oracle's fitpack restatement against a live scipy UnivariateSpline on random Hi-C-like bin
    means (s = min(y)^2, with and without the nest restart): knots and coefficients bit for bit.  Found by the differential
    fuzz campaign of round 2: the oracle squared with Python's ** (C pow(), < 1 ulp) where FITPACK multiplies - 0.5 % of
    random inputs gave coefficients 1e-13 away from scipy's, which Cephes' noise turned into |dp| > 1e-10 at counts ~1e4."""
    interp = pytest.importorskip("scipy.interpolate")
    import warnings
    from oracle import fitpack_oracle as fpo
    rng = np.random.default_rng(17)
    done = 0
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        for trial in range(400):
            m = int(rng.integers(5, 40))
            x = np.unique(np.round(np.sort(rng.uniform(1e3, 5e6, m))))
            if len(x) < 5:
                continue
            y = 1e-5 * (x / 1e3) ** -1.1 * np.exp(rng.normal(0, 0.2, len(x)))
            s = float(np.min(y) * np.min(y))
            ius = interp.UnivariateSpline(x, y, s=s)
            ts, cs = np.asarray(ius._eval_args[0]), np.asarray(ius._eval_args[1])
            t, c, _, _, _ = _capi.host_spline_fit(x, y, s)
            assert bits_equal(t, ts) and bits_equal(c, cs[:len(ts) - 4]), ("C++", trial)
            to, co = fpo.univariate_spline(list(x), list(y), s)[:2]
            assert bits_equal(np.asarray(to), ts) and bits_equal(np.asarray(co)[:len(ts) - 4], cs[:len(ts) - 4]), ("oracle", trial)
            xs = np.sort(rng.uniform(x[0], x[-1], 40))
            assert bits_equal(_capi.host_spline_eval(t, c, xs), ius(xs)), ("eval", trial)
            done += 1
    assert done > 300


def test_lbeta_table_bit_exact_vs_scipy_fixtures():
    g = np.load(os.path.join(GOLDEN, "f3_bdtrc.npz"))
    for i, n in enumerate(g["lb_n"]):
        mc = int(min(g["lb_c"].max(), n))
        lb, _ = _capi.host_lbeta_table(float(n), mc)
        sel = g["lb_c"] <= mc
        assert bits_equal(lb[g["lb_c"][sel]], g["lbeta"][i][sel]), n


def test_pava_vs_live_scipy_if_present():
    opt = pytest.importorskip("scipy.optimize")
    rng = np.random.default_rng(4)
    for n in (1, 2, 5, 100, 5000):
        y = np.sort(rng.uniform(0, 1, n))[::-1] * np.exp(rng.normal(0, 0.4, n))
        assert bits_equal(_capi.host_pava_decreasing(y), opt.isotonic_regression(y, increasing=False).x)


@pytest.mark.parametrize("name", ALL_CASES)
def test_host_fit_matches_reference(name):
    """fhx_fit fed with the reference's own distance histogram reproduces bins, x, y, knots, table bit for bit."""
    meta, g = load_case(name)
    kw = case_args(meta)
    chroms = tables.ChromIndex()
    fc, fm, fh = tables.read_fragments(kw["frags"], chroms)
    res = kw["resolution"]
    for pi in range(1, meta["n_passes"] + 1):
        P = "p%d_" % pi
        ctx = _capi.Context(-1)
        ctx.set_params(res, kw["L"], kw["U"], kw["n_bins"], kw["mapp_thres"], MODES[kw["mode"]], kw["tL"], kw["tU"])
        ctx.load_fragments(fc, fm, fh, chroms.sort_rank())
        keys, sumcc = g[P + "dist_keys"], g[P + "dist_sumcc"]
        st = _capi.FhxStats()
        st.inter_count, st.inter_sum, st.intra_all_sum, st.in_range_sum = [int(v) for v in g[P + "sums"]]
        if res == 0 or name.startswith("f11_offgrid"):   # -r 0 / off-grid loci: explicit distance keys and outlier multiset
            ctx.set_dist_keys(keys)
            ctx.set_global_stats(st, sumcc, np.ones(len(keys), np.int64))
            if pi > 1:
                ctx.set_outlier_dists(g["p%d_outliersdist" % (pi - 1)])
        else:
            n_dist = int(keys.max() // res + 1) if len(keys) else 1
            if pi > 1:
                od = g["p%d_outliersdist" % (pi - 1)]
                n_dist = max(n_dist, int(-(-od.max() // res)) + 1)
            hist_cc = np.zeros(n_dist, np.int64)
            hist_np = np.zeros(n_dist, np.int64)
            hist_cc[keys // res] = sumcc
            hist_np[keys // res] = 1
            ctx.set_global_stats(st, hist_cc, hist_np)
            if pi > 1:
                oh = np.zeros(n_dist, np.int64)
                np.add.at(oh, -(-od // res), 1)          # ceil: bins end on grid distances
                ctx.set_outlier_dist_hist(oh)
        info = ctx.fit()
        for k, w in (("lb", _capi.A_BIN_LB), ("ub", _capi.A_BIN_UB), ("s1", _capi.A_BIN_POSS), ("s2", _capi.A_BIN_SUMCC),
                     ("s7", _capi.A_BIN_POSS7)):
            assert np.array_equal(ctx.get_array(w), g[P + "bins1_" + k]), k
        assert bits_equal(ctx.get_array(_capi.A_BIN_SUMDIST), g[P + "bins1_s3"])
        assert np.array_equal(ctx.get_array(_capi.A_BIN_POSS0), g[P + "bins0_s1"])
        fs = g[P + "frag_scalars"]
        mine = np.array([info.n_frags, info.max_possible_dist, info.possible_intra_in_range, info.possible_inter_all,
                         info.inter_chr_prob, info.baseline_intra_prob], np.float64)
        assert bits_equal(mine, fs)
        assert bits_equal(ctx.get_array(_capi.A_X), g[P + "x"]) and bits_equal(ctx.get_array(_capi.A_Y), g[P + "y"])
        assert 1.0 / info.bh_total_tests == g[P + "outlierThres"][0]
        if P + "spl_t" in g:
            assert bits_equal(ctx.get_array(_capi.A_KNOTS), g[P + "spl_t"])
            assert bits_equal(ctx.get_array(_capi.A_COEFFS), g[P + "spl_c"])
            assert np.array_equal(ctx.get_array(_capi.A_TABLE_X), g[P + "splineX"])
            assert bits_equal(ctx.get_array(_capi.A_TABLE_Y0), g[P + "splineY"])
            assert bits_equal(ctx.get_array(_capi.A_TABLE_Y), g[P + "newSplineY"])
            s, fp, ier = g[P + "spl_s_fp_ier"]
            assert info.spline_s == s and info.spline_fp == fp and info.spline_ier == int(ier)
            assert info.residual == g[P + "residual"][0]
        ctx.close()


@pytest.mark.parametrize("name", ["C2", "C3", "C3w", "C5"])
def test_host_fit_matches_reference_at_headline_sizes(name):
    """The fit at the sizes the metric is quoted on - 22 autosomes at 5 kb / 1 kb (576 216 / 2 881 044 loci; 397, 49 734 and
    1 999 distance values): fhx_fit fed with the histogram of the full synth-v1 workload reproduces what the REAL reference's
    makeBinsFromInteractions -> generate_FragPairs -> calculateProbabilities -> fit_Spline returned on it (fixtures f14, made by
    tests/golden/make_golden.py f14) bit for bit: bins, possible pairs, bin means (the 19999.999999999996-type values of SURVEY
    fact 6 occur here), s = min(y)^2, knots, coefficients, the isotonic table, N."""
    import bench
    from fithic_amd import synth
    from oracle import run_check
    g = run_check.fit_fixture(name)
    assert g is not None
    cfg = bench.CONFIGS[name]
    genome = synth.Genome(cfg["res"], cfg["lengths"])
    ctx = _capi.Context(-1)
    ctx.set_params(cfg["res"], cfg["L"], cfg["U"], 100, 1, MODES[cfg["mode"]])
    ctx.load_fragments(*genome.fragments(), genome.sort_rank())
    st = _capi.FhxStats()
    st.inter_count, st.inter_sum, st.intra_all_sum, st.in_range_sum = [int(v) for v in g["sums"]]
    n_dist = int(g["hist_dist_idx"].max()) + 1
    hist_cc, hist_np = np.zeros(n_dist, np.int64), np.zeros(n_dist, np.int64)
    hist_cc[g["hist_dist_idx"]] = g["hist_sumcc"]
    hist_np[g["hist_dist_idx"]] = g["hist_nrows"]
    assert np.array_equal(g["dist_keys"], g["hist_dist_idx"] * cfg["res"]) and np.array_equal(g["dist_sumcc"], g["hist_sumcc"])
    ctx.set_global_stats(st, hist_cc, hist_np)
    info = ctx.fit().as_dict()
    assert run_check.compare_fit(ctx.get_array, info, g) == []
    assert len(g["x"]) == 100 and len(g["splineX"]) > 300
    x = g["x"]
    assert np.any(x != np.round(x))            # bin means that are not whole numbers: the summation order matters here
    ctx.close()


def test_nonfixed_possible_pairs_in_parallel_equal_the_sequential_walk(monkeypatch):
    compared = sum(_nonfixed_case(seed, monkeypatch) for seed in range(16))
    assert compared >= 16            # 16 seeds x 2 thread counts, minus the cases the reference's spline stage exits on


def _nonfixed_case(seed, monkeypatch):
    """-r 0 (fithic.py:691-778): the library counts slots [1] / [7] per run of y in closed form and sums slot [3] as one
    sequential chain per bin, the bins spread over host threads.  Against the oracle's literal pair walk on random
    irregular fragments (duplicate midpoints, unmappable ones, chromosomes without fragments in range, bounds on and
    off): integer slots equal, slot [3] bit for bit, for 1 and 5 threads."""
    from oracle import fithic_oracle as fo
    rng = np.random.default_rng(100 + seed)
    n_chr = int(rng.integers(1, 5))
    frag_rows, f_chr, f_mid, f_hit = [], [], [], []
    for c in range(n_chr):
        n = int(rng.integers(0, 400))
        mids = np.sort(rng.integers(0, 3_000_000, n))
        if n > 10:
            mids[rng.integers(0, n, 5)] = mids[rng.integers(0, n, 5)]          # duplicates: distance 0
        hits = (rng.random(n) > 0.1).astype(np.int64)
        for m, h in zip(rng.permutation(n), hits):                             # file order is not sorted
            frag_rows.append(("c%d" % c, int(mids[m]), int(h)))
            f_chr.append(c)
            f_mid.append(int(mids[m]))
            f_hit.append(int(h))
    L = int(rng.choice([0, 0, 20000, 150000]))
    U = float(rng.choice([np.inf, 400000, 1500000]))
    # observed distances -> bins (any ascending keys with counts will do: the enumeration only needs the bin ends)
    keys = np.unique(rng.integers(max(L, 1), int(min(U, 2_500_000)), 300)).astype(np.int64)
    sumcc = rng.integers(1, 50, len(keys)).astype(np.int64)
    n_bins = int(rng.choice([1, 3, 12, 40]))
    bins = fo.make_bins(keys, sumcc, n_bins, int(sumcc.sum()))
    want = fo.generate_frag_pairs_nonfixed(frag_rows, bins, L, U, 1, 7)
    compared = 0
    for threads in ("1", "5"):
        monkeypatch.setenv("FHX_THREADS", threads)
        ctx = _capi.Context(-1)
        ctx.set_params(0, L, U, n_bins, 1, MODES["intraOnly"])
        rank = np.argsort(np.argsort(["c%d" % c for c in range(n_chr)])).astype(np.int32)
        ctx.load_fragments(np.array(f_chr, np.int32), np.array(f_mid, np.int32), np.array(f_hit, np.int32), rank)
        st = _capi.FhxStats()
        st.in_range_sum, st.inter_count = int(sumcc.sum()), 7
        ctx.set_dist_keys(keys)
        ctx.set_global_stats(st, sumcc, np.ones(len(keys), np.int64))
        try:
            info = ctx.fit()
        except _capi.FhxError as e:                                            # bin means the reference's spline stage exits on
            assert e.code == _capi.FHX_ERR_REFERENCE_EXIT, str(e)
            info = None
        if info is not None:
            compared += 1
            assert np.array_equal(ctx.get_array(_capi.A_BIN_POSS), [b["s1"] for b in bins])
            assert np.array_equal(ctx.get_array(_capi.A_BIN_POSS7), [b["s7"] for b in bins])
            assert bits_equal(ctx.get_array(_capi.A_BIN_SUMDIST), np.array([b["s3"] for b in bins], np.float64))
        if info is not None:
            assert info.possible_intra_in_range == want["poss_in_range"] and info.max_possible_dist == want["max_possible_dist"]
            assert info.possible_inter_all == want["poss_inter"] and info.n_frags == want["n_frags"]
        ctx.close()
    return compared


def test_visual_plots_write_the_reference_figures(tmp_path):
    """-v: the four figure kinds of fithic.py:970-999,1256-1321 are produced from host arrays."""
    pytest.importorskip("matplotlib")
    from fithic_amd import plots
    x = np.linspace(2e4, 2e6, 40)
    y = 1e-5 * (x / 2e4) ** -1.1
    sx = np.arange(20000, 2000001, 5000)
    sy = 1e-5 * (sx / 2e4) ** -1.08
    fdrx = np.arange(0.0, 0.051, 0.001)
    fdry = np.cumsum(np.arange(51))
    base = str(tmp_path / "lib.spline_pass1")
    plots.plot_spline_fit(base, 1, list(x), list(y), [0] * len(x), list(sx), sy, 20000, 2000000)
    plots.plot_qvalues(fdrx, list(fdry), base + ".qplot")
    plots.compare_Spline_FDR(fdrx, list(fdry), fdrx, list(fdry * 2), str(tmp_path / "lib.spline_FDR_comparison"), "2")
    plots.compareFits_Spline(list(sx), sy, list(sx), sy * 0.9, str(tmp_path / "lib.spline_comparison"), "2")
    for name in ("lib.spline_pass1.png", "lib.spline_pass1.qplot.png", "lib.spline_FDR_comparison.png", "lib.spline_comparison.png"):
        f = tmp_path / name
        assert f.exists() and f.read_bytes()[:8] == b"\x89PNG\r\n\x1a\n" and f.stat().st_size > 2000


def test_header_is_plain_c():
    """include/fithic_mi355x.h must be consumable from C (cgo / JNI / FFI bindings read it): C99, pedantic, no warnings."""
    import subprocess
    hdr = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include", "fithic_mi355x.h")
    r = subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-fsyntax-only", "-x", "c", hdr], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr


def test_build_returns_at_once_when_the_library_is_newer_than_every_source():
    """csrc/_obj is neither in git nor on the GPU box; an up-to-date .so must not recompile nine units to find that out"""
    import time
    from fithic_amd import _capi
    _capi.build()
    t0 = time.perf_counter()
    assert _capi.build() == _capi.LIB_PATH
    assert time.perf_counter() - t0 < 0.5


def _python_part_bounds(text, part, n_parts):
    """the rule of fhx_text_part_bounds restated: a cut is the first start of a row at or after the N-th's first byte"""
    T = len(text)

    def cut(at):
        if at <= 0:
            return 0
        k = text.find(b"\n", at - 1)
        return T if k < 0 else k + 1
    lo = cut(T // n_parts * part + T % n_parts * part // n_parts)
    hi = T if part + 1 == n_parts else cut(T // n_parts * (part + 1) + T % n_parts * (part + 1) // n_parts)
    return lo, hi


@pytest.mark.parametrize("shape", ["rows", "no_final_newline", "one_long_row", "no_newline_at_all", "empty", "many_pieces"])
def test_text_parts_tile_the_text_on_row_starts(shape, tmp_path, monkeypatch):
    """fhx_text_part_bounds (what `fithic --gpus N` cuts a plain gzip file's text with): for every N the parts tile the text, every
    part begins at the start of a row, and they are the restated rule's - over one piece and over many (the parallel gunzip's chunks,
    whose borders fall anywhere), rows longer than a part and texts without a newline included."""
    import gzip
    rng = np.random.default_rng(5)
    rows = [b"chr%d\t%d\tchr%d\t%d\t%d\n" % (rng.integers(1, 23), rng.integers(1, 10**9), rng.integers(1, 23), rng.integers(1, 10**9),
                                              rng.integers(1, 500)) for _ in range(4000)]
    text = {"rows": b"".join(rows), "no_final_newline": b"".join(rows)[:-1],
            "one_long_row": b"".join(rows[:10]) + b"x" * 200000 + b"\n" + b"".join(rows[10:40]),
            "no_newline_at_all": b"y" * 5000, "empty": b"", "many_pieces": b"".join(rows * 12)}[shape]
    if shape == "many_pieces":
        monkeypatch.setenv("FHX_PGUNZIP_MIN", "0")
        monkeypatch.setenv("FHX_PGUNZIP_CHUNK", "16384")
    path = str(tmp_path / "t.gz")
    with gzip.open(path, "wb", 6) as f:
        f.write(text)
    ht = _capi.HostText(path, 8)
    try:
        assert len(ht) == len(text)
        for n_parts in (1, 2, 3, 5, 8, 64):
            bounds = [ht.part_bounds(k, n_parts) for k in range(n_parts)]
            assert bounds == [_python_part_bounds(text, k, n_parts) for k in range(n_parts)]
            assert bounds[0][0] == 0 and bounds[-1][1] == len(text)
            for (lo, hi), (lo2, _) in zip(bounds, bounds[1:]):
                assert lo <= hi == lo2
            for lo, hi in bounds:
                assert lo == 0 or lo == len(text) or text[lo - 1:lo] == b"\n"
        with pytest.raises(_capi.FhxError):
            ht.part_bounds(3, 3)
    finally:
        ht.close()
