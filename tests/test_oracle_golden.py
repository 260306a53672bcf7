"""The oracle (oracle/) against the golden fixtures captured from the REAL reference.

Everything here runs on CPU.  It pins the checker that the GPU parity tests rely on:
bit-exact stage intermediates, bit-exact p / q / expCC / bias, sha256 of the full p and q arrays and
md5 of the decompressed .significances.txt text.
"""
import hashlib

import numpy as np
import pytest

from conftest import load_case, case_args, ALL_CASES, bits_equal
from oracle import fithic_oracle as fo
from oracle import fitpack_oracle


@pytest.fixture(scope="module")
def g3():
    import os
    from conftest import GOLDEN
    return np.load(os.path.join(GOLDEN, "f3_bdtrc.npz"))


def test_bdtrc_known_answers_bit_exact(g3):
    out = fo.bdtrc(g3["k"], g3["n"].astype(np.float64), g3["p"])
    ref = g3["val"]
    assert np.array_equal(np.isnan(out), np.isnan(ref))
    ok = np.isnan(ref) | (out.view(np.int64) == ref.view(np.int64))
    assert ok.all(), "%d of %d bdtrc vectors differ from scipy" % ((~ok).sum(), len(ok))


def test_bdtrc_narrows_the_total_to_a_c_int_like_scipy():
    """n at and above 2^31 (fithic.py:1070,1101 pass Python ints; scipy's core takes `int n`): scipy's own values at fifteen
    totals from 2^31 - 1 to 2^52, incl. C3w's 3 215 733 208 and C5's 7 150 761 687 (make_golden.py f15)."""
    import os
    from conftest import GOLDEN
    g = np.load(os.path.join(GOLDEN, "f15_bdtrc_int_n.npz"))
    out = fo.bdtrc(g["k"], g["n"].astype(np.float64), g["p"])
    ref = g["val"]
    assert np.array_equal(np.isnan(out), np.isnan(ref))
    ok = np.isnan(ref) | (out.view(np.int64) == ref.view(np.int64))
    assert ok.all(), "%d of %d vectors differ from scipy" % ((~ok).sum(), len(ok))
    assert np.isnan(ref).sum() > 1000 and (~np.isnan(ref)).sum() > 1000
    # the second mode is a different function there - and the same one below 2^31
    wide = fo.bdtrc(g["k"], g["n"].astype(np.float64), g["p"], totals="wide")
    big = g["n"] >= 2 ** 31
    assert bits_equal(wide[~big], ref[~big])
    assert (np.isnan(ref[big]) & ~np.isnan(wide[big])).sum() > 1000
    assert [fo.int_narrowed(v) for v in (2 ** 31 - 1, 2 ** 31, 2 ** 32 - 1, 2 ** 32, 2 ** 32 + 10 ** 6, 7150761687)] == [
        2 ** 31 - 1, -2 ** 31, -1, 0, 10 ** 6, -1439172905]


def test_bdtrc_branch_coverage(g3):
    _, br, it = fo.bdtrc_stats(g3["k"], g3["n"].astype(np.float64), g3["p"])
    # closed form, pseries, incbcf, incbd, swapped pseries / incbcf / incbd all present
    for b in (0, 1, 2, 3, 5, 6, 7):
        assert (br == b).sum() > 0, "branch %d not covered" % b
    assert it[br == 6].max() == 300          # the swapped continued fraction hits Cephes' cap


def test_lbeta_and_log_bit_exact(g3):
    c = np.repeat(g3["lb_c"][None, :], len(g3["lb_n"]), 0).astype(np.float64)
    b = (g3["lb_n"][:, None] - g3["lb_c"][None, :] + 1).astype(np.float64)
    ok = b > 0
    out = fo.lbeta(c[ok], b[ok])
    assert bits_equal(out, g3["lbeta"][ok])
    import math
    assert bits_equal(np.array([math.log(v) for v in g3["log_x"]]), g3["log_val"])


def test_bdtrc_against_live_scipy_if_present():
    scsp = pytest.importorskip("scipy.special")
    rng = np.random.default_rng(123)
    n = rng.choice([99_991, 6_495_767, 400_000_000], 4000).astype(np.float64)
    cnt = rng.geometric(0.05, 4000).astype(np.float64)
    prior = np.clip(cnt * np.exp(rng.normal(0, 1.0, 4000)) / n, 1e-13, 0.99)
    ref = scsp.bdtrc(cnt - 1, n.astype(np.int64), prior)
    assert bits_equal(fo.bdtrc(cnt - 1, n, prior), ref)


def test_fitpack_known_answers_bit_exact():
    import os
    from conftest import GOLDEN
    g = np.load(os.path.join(GOLDEN, "f4_fitpack.npz"))
    restarts = 0
    for name in g["names"]:
        s, fp, ier = g[name + "_sfpier"]
        t, c, fp2, ier2, restarted = fitpack_oracle.univariate_spline(g[name + "_x"], g[name + "_y"], s)
        restarts += restarted
        assert bits_equal(np.array(t), g[name + "_t"]), name
        assert bits_equal(np.array(c), g[name + "_c"]), name
        assert fp2 == fp and ier2 == int(ier), name
        assert bits_equal(np.array(fitpack_oracle.splev(t, c, g[name + "_xe"])), g[name + "_ye"]), name
    assert restarts >= 3            # the nest-restart route is exercised


def test_pava_against_live_scipy_if_present():
    opt = pytest.importorskip("scipy.optimize")
    rng = np.random.default_rng(9)
    for n in (1, 2, 3, 10, 200, 3000):
        y = np.sort(rng.uniform(0, 1, n))[::-1] * np.exp(rng.normal(0, 0.3, n))
        ref = opt.isotonic_regression(y, increasing=False).x
        assert bits_equal(fo.pava_decreasing(y), ref)


def test_bh_known_answers_bit_exact():
    import os
    from conftest import GOLDEN
    g = np.load(os.path.join(GOLDEN, "f5_bh.npz"))
    for name in g["names"]:
        q = fo.benjamini_hochberg(g[name + "_p"], g[name + "_N"][0])
        assert bits_equal(q, g[name + "_q"]), name


def test_bh_with_zero_or_negative_number_of_tests():
    """fit_Spline can pass a negative N (possible-pair counts go negative with unmappable loci): the reference's loop
    starts its running maximum at 0, so negative bh values give q = 0 (and -0.0 once a p == 0 has passed).  Vectors generated
    by the real myStats.benjamini_hochberg_correction (make_golden.py f12)."""
    import os
    from conftest import GOLDEN
    g = np.load(os.path.join(GOLDEN, "f12_bh_nonpositive_N.npz"))
    assert len(g["names"]) == 6
    for name in g["names"]:
        for f in (fo.benjamini_hochberg, fo.benjamini_hochberg_pruned):
            assert bits_equal(f(g[name + "_p"], g[name + "_N"][0]), g[name + "_q"]), (name, f.__name__)


def test_pruned_bh_is_the_same_function():
    """benjamini_hochberg_pruned (used to check q of 10^8-row GPU runs) equals the plain restatement bit for bit: on the
    reference's own known answers and on random vectors with ties, 1.0s, zeros and NaNs at several N / n ratios."""
    import os
    from conftest import GOLDEN
    g = np.load(os.path.join(GOLDEN, "f5_bh.npz"))
    for name in g["names"]:
        assert bits_equal(fo.benjamini_hochberg_pruned(g[name + "_p"], g[name + "_N"][0]), g[name + "_q"]), name
    rng = np.random.default_rng(12)
    for trial in range(120):
        n = int(rng.integers(1, 4000))
        kind = trial % 5
        if kind == 0:
            p = rng.random(n)
        elif kind == 1:
            p = rng.random(n) ** 8
        elif kind == 2:
            p = np.round(rng.random(n), 2)
        elif kind == 3:
            p = np.where(rng.random(n) < 0.3, 1.0, rng.random(n) ** 4)
        else:
            p = rng.random(n) ** 6
            p[rng.random(n) < 0.05] = np.nan
            p[rng.random(n) < 0.05] = 0.0
        for N in (0.5 * n, n, 3.7 * n, 100 * n, 1):
            assert bits_equal(fo.benjamini_hochberg_pruned(p, N), fo.benjamini_hochberg(p, N)), (trial, N)


def test_skip_mask_freezes_at_duplicate():
    m = fo.effective_skip_mask(12, [2, 5, 5, 7, 9])
    assert m.nonzero()[0].tolist() == [2, 5]          # nothing after the first duplicate is skipped
    m = fo.effective_skip_mask(12, [7, 2, 9])
    assert m.nonzero()[0].tolist() == [2, 7, 9]


@pytest.mark.parametrize("name", ALL_CASES)
def test_pipeline_matches_reference(name):
    meta, g = load_case(name)
    kw = case_args(meta)
    results = fo.run(keep_text=True, **kw)
    assert len(results) == meta["n_passes"]
    sub = meta["subsample"]
    for pi, r in enumerate(results, 1):
        P = "p%d_" % pi
        assert np.array_equal(r.dist_keys, g[P + "dist_keys"])
        assert np.array_equal(r.dist_sumcc, g[P + "dist_sumcc"])
        assert np.array_equal(np.array(r.sums), g[P + "sums"])
        for k in ("lb", "ub", "s1", "s2", "s7"):
            assert np.array_equal(np.array([b[k] for b in r.bins]), g[P + "bins1_" + k]), k
        assert bits_equal(np.array([b["s3"] for b in r.bins]), g[P + "bins1_s3"])
        assert np.array_equal(np.array([b["s1"] for b in r.bins0]), g[P + "bins0_s1"])
        fr = r.frag
        mine = np.array([fr["n_frags"], fr["max_possible_dist"], fr["poss_in_range"], fr["poss_inter"],
                         fr["inter_prob"], fr["base_prob"]], np.float64)
        assert bits_equal(mine, g[P + "frag_scalars"])
        assert bits_equal(np.array(r.x), g[P + "x"]) and bits_equal(np.array(r.y), g[P + "y"])
        if P + "spl_t" in g:
            assert bits_equal(r.t, g[P + "spl_t"]) and bits_equal(r.c, g[P + "spl_c"])
            assert np.array_equal(r.splineX, g[P + "splineX"])
            assert bits_equal(r.splineY, g[P + "splineY"]) and bits_equal(r.newSplineY, g[P + "newSplineY"])
            assert r.residual == g[P + "residual"][0]
        for k, arr in (("p", r.p), ("q", r.q), ("expcc", r.expcc), ("b1", r.b1), ("b2", r.b2)):
            assert bits_equal(arr[::sub], g[P + k]), k
        m = meta["pass%d" % pi]
        assert hashlib.sha256(r.p.tobytes()).hexdigest() == m["p_sha256"]
        assert hashlib.sha256(r.q.tobytes()).hexdigest() == m["q_sha256"]
        assert r.outlier_thres == g[P + "outlierThres"][0]
        assert r.n_outlier_lines_total == g[P + "n_outlier_lines"][0]
        assert np.array_equal(np.array(r.fdr_y), g[P + "fdr_y"])
        assert r.pass_txt == meta["fithic_pass%d_txt" % pi]
        assert hashlib.md5(r.sig_txt.encode()).hexdigest() == meta["sig_md5_pass%d" % pi]
