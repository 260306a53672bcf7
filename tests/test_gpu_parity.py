"""GPU parity tests (run with -m gpu on a real MI355X): the HIP path, called through the C ABI, against
(a) the golden fixtures captured from the real reference and (b) the oracle on the same inputs.

Bars (BASELINE.json north_star): distance-bin indices, pair counts, histograms and all integer results
bit-exact; p- and q-values within 1e-10 absolute of scipy.special.bdtrc-based reference values (the tests
also report how many values are bit-identical - in practice nearly all are).
"""
import os

import numpy as np
import pytest

from conftest import load_case, case_args, ALL_CASES, SMALL_CASES, WRAP_CASES, bits_equal, max_abs_diff, GOLDEN

pytestmark = pytest.mark.gpu

TOL = 1e-10


@pytest.fixture(scope="module")
def ctx():
    from fithic_amd import _capi
    c = _capi.Context(0)          # raises if there is no GPU or the library is missing: no fallback
    yield c
    c.close()


def test_extension_is_the_in_tree_hip_library(ctx):
    from fithic_amd import _capi
    assert os.path.exists(_capi.LIB_PATH)
    with open("/proc/self/maps") as f:
        assert "libfithic_mi355x.so" in f.read()
    assert b"gfx950" in _capi.lib().fhx_version()


def test_bdtrc_known_answers_on_gpu(ctx):
    """K2's arithmetic against scipy.special.bdtrc bit patterns (F3), integer counts, every Cephes branch."""
    g = np.load(os.path.join(GOLDEN, "f3_bdtrc.npz"))
    k, n, p, ref = g["k"], g["n"], g["p"], g["val"]
    integral = (k == np.floor(k))
    worst, n_bits, total = 0.0, 0, 0
    for nt in np.unique(n[integral]):
        sel = integral & (n == nt)
        out = ctx.bdtrc_array(float(nt), (k[sel] + 1).astype(np.int32), p[sel])
        worst = max(worst, max_abs_diff(out, ref[sel]))
        same = (out.view(np.int64) == ref[sel].view(np.int64)) | (np.isnan(out) & np.isnan(ref[sel]))
        n_bits += int(same.sum())
        total += int(sel.sum())
    print("bdtrc KAT: %d/%d bit-identical, max |diff| %.3e" % (n_bits, total, worst))
    assert worst <= TOL
    assert n_bits >= 0.95 * total


def test_bdtrc_narrows_totals_beyond_a_c_int_like_scipy(ctx):
    """scipy's own values at fifteen totals from 2^31 - 1 to 2^52 (fixture f15_bdtrc_int_n, made by make_golden.py f15): NaN
    pattern equal, values within 1e-10.  With FHX_TOTALS_WIDE the same entry keeps the true total (the oracle's second mode)."""
    from oracle import fithic_oracle as fo
    from fithic_amd._capi import Context, TOTALS_WIDE
    g = np.load(os.path.join(GOLDEN, "f15_bdtrc_int_n.npz"))
    k, n, p, ref = g["k"], g["n"], g["p"], g["val"]
    wide = Context(0)
    wide.set_params(10000, totals=TOTALS_WIDE)
    n_nan = 0
    for nt in np.unique(n):
        sel = n == nt
        out = ctx.bdtrc_array(float(nt), (k[sel] + 1).astype(np.int32), p[sel])
        assert max_abs_diff(out, ref[sel]) <= TOL, nt                  # asserts the NaN pattern too
        n_nan += int(np.isnan(out).sum())
        out_w = wide.bdtrc_array(float(nt), (k[sel] + 1).astype(np.int32), p[sel])
        assert max_abs_diff(out_w, fo.bdtrc(k[sel], float(nt), p[sel], totals="wide")) <= TOL, nt
        if nt >= 2 ** 31:                                  # a different function there: other NaNs (negative narrowed n) or other values
            assert not bits_equal(out_w, out)
    assert n_nan == int(np.isnan(ref).sum()) > 1000
    wide.close()


def test_bdtrc_against_oracle_random(ctx):
    from oracle import fithic_oracle as fo
    rng = np.random.default_rng(77)
    for nt in (1234, 150, 6_495_767, 987_654_321):
        cnt = np.minimum(rng.geometric(0.03, 20000), nt).astype(np.int32)
        prior = np.clip(cnt * np.exp(rng.normal(0, 1.3, len(cnt))) / nt, 0, 1)
        prior[::1000] = 0.0
        prior[1::1000] = 1.0
        prior[2::1000] = -0.25
        prior[3::1000] = np.nan
        out = ctx.bdtrc_array(float(nt), cnt, prior)
        ref = fo.bdtrc(cnt.astype(np.float64) - 1, float(nt), prior)
        assert max_abs_diff(out, ref) <= TOL, nt


@pytest.mark.parametrize("kind", [0, 1])
def test_continued_fraction_bit_exact(ctx, kind):
    """The raw Cephes continued fractions on the GPU - literal test and division-free test - against the oracle's C,
    bit for bit (this is the part of bdtrc whose truncated, non-converged value is the answer: SURVEY fact 4)."""
    from oracle import fithic_oracle as fo
    rng = np.random.default_rng(100 + kind)
    n = 400000               # ~1e8 divisions through each of the three division modes (plain, lean, tracked reciprocal)
    ntot = rng.choice([3.0e5, 1.2e6, 6.495767e6, 2.2294127e7, 2.1e8, 1.5e9], n)
    cnt = rng.geometric(0.1, n).astype(np.float64) + 1
    ratio = np.exp(rng.normal(0.3 if kind == 0 else -0.2, 0.8, n))
    prior = np.clip(cnt * ratio / ntot, 1e-12, 0.6)
    swapped = rng.random(n) < 0.6
    a = np.where(swapped, ntot - cnt + 1, cnt)
    b = np.where(swapped, cnt, ntot - cnt + 1)
    x = np.where(swapped, 1.0 - prior, prior)
    # a few degenerate inputs: tiny / huge ratios, x = 0
    a[:5], b[:5], x[:5] = [1, 2, 5, 1e7, 3], [1, 1, 2, 2, 1e7], [0.0, 0.5, 1e-300, 0.999999, 1e-9]
    ref = fo.contfrac(kind, a, b, x)
    plain = ctx.debug_contfrac(kind, 0, a, b, x)
    lazy = ctx.debug_contfrac(kind, 1, a, b, x)
    assert bits_equal(plain, ref)
    assert bits_equal(lazy, ref)


@pytest.mark.parametrize("kind", [0, 1])
def test_continued_fraction_bit_exact_large_n(ctx, kind):
    """Totals of 2e8 .. 4e15 contacts: the loop refines the tracked reciprocal with ONE Newton step per denominator
    (division mode 3); 4e5 fractions x 600 divisions, bit for bit against the oracle's C.  These are Cephes' incbcf / incbd as
    functions of three doubles (a, b, x) - faithful whatever bdtrc's caller did to n; arguments with a + b > 2^31 reach them
    only under FHX_TOTALS_WIDE (the reference's narrowed totals never produce them)."""
    from oracle import fithic_oracle as fo
    rng = np.random.default_rng(300 + kind)
    n = 400000
    ntot = np.floor(10 ** rng.uniform(8.31, 15.6, n))
    ntot[: n // 2] = np.floor(10 ** rng.uniform(8.31, 9.5, n // 2))          # the realistic range, densely
    cnt = rng.geometric(0.08, n).astype(np.float64) + 1
    ratio = np.exp(rng.normal(0.3 if kind == 0 else -0.2, 0.8, n))
    prior = np.clip(cnt * ratio / ntot, 1e-17, 0.6)
    a, b, x = ntot - cnt + 1, cnt, 1.0 - prior                               # swapped orientation: a is the huge one
    ref = fo.contfrac(kind, a, b, x)
    for lazy in (1, 0):
        got = ctx.debug_contfrac(kind, lazy, a, b, x)
        same = (got.view(np.int64) == ref.view(np.int64)) | (np.isnan(got) & np.isnan(ref))      # incbd overflows to NaN on a few
        assert same.all() and np.isnan(ref).sum() < 20


def test_class_threshold_table_equals_incbet_predicates(ctx):
    """k2_classify reads bdtrc_class as five thresholds on the prior per count (dev::cls_row, bisected on the device with the
    predicates of Cephes' incbet).  Table and arithmetic must agree on EVERY double: random priors over twelve decades, the
    domain edges, and - where a wrong threshold would show - each threshold itself with its 3 neighbours on both sides, for
    Hi-C-sized and small binomials (the small ones exercise the direct orientation and both power-series tests)."""
    rng = np.random.default_rng(21)
    for n_total in (645040870.0, 7150761687.0, 1.0e6, 5000.0, 170.0, 3.0):
        counts = np.unique(np.concatenate([np.arange(0, 40), rng.integers(1, 5000, 300), [int(min(n_total, 2e9)) - 1, int(min(n_total, 2e9)),
                                                                                           int(min(n_total, 2e9)) + 1]])).astype(np.int32)
        counts = counts[counts >= 0]
        _, _, thr = ctx.debug_classify(n_total, counts, np.full(len(counts), 0.5), thresholds=True)
        cs, ps = [], []
        for c, row in zip(counts, thr):
            pri = [0.0, 1.0, -0.1, 1.5, np.nan, 5e-324, 1.0 - 2.0 ** -53, 0.95, np.nextafter(0.95, 1), 0.05, np.nextafter(0.05, 0)]
            for t in row:
                if 0.0 <= t <= 1.0:
                    v = t
                    for _ in range(4):
                        pri.append(v)
                        v = np.nextafter(v, 2.0)
                    v = t
                    for _ in range(3):
                        v = np.nextafter(v, -1.0)
                        pri.append(v)
            pri += list(10.0 ** rng.uniform(-12, 0, 24)) + list(1.0 - 10.0 ** rng.uniform(-12, 0, 12))
            cs += [c] * len(pri)
            ps += pri
        by_table, by_arith = ctx.debug_classify(n_total, np.array(cs, np.int32), np.array(ps, np.float64))
        bad = np.flatnonzero(by_table != by_arith)
        assert len(bad) == 0, (n_total, cs[bad[0]], ps[bad[0]], int(by_table[bad[0]]), int(by_arith[bad[0]]), len(bad))
        assert len(np.unique(by_arith)) >= (4 if n_total > 1000 else 2)         # the classes really occur


def test_lean_division_matches_ieee(ctx):
    """K2 divides with the core of hipcc's own f64 division expansion (no scaling scaffolding): must equal IEEE n/d
    bit for bit on the operand window it is used in - numerators 0 or arg*k*k', denominators (a+2n)(a+2n+1)."""
    rng = np.random.default_rng(2024)
    total = 0
    for rep in range(25):
        m = 4_000_000
        a = np.floor(10 ** rng.uniform(0, 9.3, m))
        k = rng.integers(0, 300, m).astype(np.float64)
        d = (a + 2 * k) * (a + 2 * k + 1)
        arg = np.where(rng.random(m) < 0.5, 1.0 - 10 ** rng.uniform(-12, -0.3, m), 10 ** rng.uniform(-14, 2, m))
        nmr = arg * (a + k) * (a + rng.integers(1, 2000, m) + k)
        nmr[::1000] = 0.0
        nmr[1::3] *= -1.0
        if rep % 5 == 0:                 # edges of the window
            nmr *= 10 ** rng.uniform(-180, 150, m)
        got = ctx.debug_lean_div(nmr, d)
        want = nmr / d
        nz = want != 0
        assert np.array_equal(got[nz].view(np.int64), want[nz].view(np.int64))
        assert np.all(got[~nz] == 0)
        total += m
    assert total == 100_000_000


def test_bh_known_answers_on_gpu(ctx):
    g = np.load(os.path.join(GOLDEN, "f5_bh.npz"))
    for name in g["names"]:
        q = ctx.bh_array(g[name + "_p"], g[name + "_N"][0])
        assert bits_equal(q, g[name + "_q"]), name


def test_bh_large_random_bit_exact_vs_oracle(ctx):
    from oracle import fithic_oracle as fo
    rng = np.random.default_rng(8)
    n = 3_000_017                                   # several radix tiles per workgroup, ragged tail
    p = rng.uniform(0, 1, n) ** rng.integers(1, 40, n)
    p[rng.integers(0, n, n // 50)] = 1.0
    p[rng.integers(0, n, 1000)] = np.nan
    p[rng.integers(0, n, 1000)] = 0.0
    p[rng.integers(0, n, n // 10)] = p[rng.integers(0, n, n // 10)]          # ties
    q = ctx.bh_array(p, 7.5e9)
    assert bits_equal(q, fo.benjamini_hochberg(p, 7.5e9))


@pytest.mark.parametrize("n", [1, 2, 3, 63, 64, 65, 255, 4095, 4096, 4097, 8191, 12289, 100001, 1048577])
def test_bh_ragged_sizes_bit_exact(ctx, n):
    """Tile / chunk boundaries of the compaction, radix sort and scan kernels (ragged tails, odd lengths)."""
    from oracle import fithic_oracle as fo
    rng = np.random.default_rng(n)
    p = rng.uniform(0, 1, n) ** rng.integers(1, 30, n)
    p[rng.integers(0, n, max(1, n // 7))] = 1.0
    if n > 10:
        p[rng.integers(0, n, 3)] = np.nan
    assert bits_equal(ctx.bh_array(p, 3.0 * n + 1), fo.benjamini_hochberg(p, 3.0 * n + 1))
    allone = np.ones(n)
    assert bits_equal(ctx.bh_array(allone, 5.0), allone)          # nothing to sort at all


@pytest.mark.parametrize("n", [4095, 4096, 4097, 12288, 50000, 131072, 131073, 300000])
def test_bh_tile_sort_and_radix_sort_agree_with_the_oracle(ctx, n, monkeypatch):
    """Survivor sets of up to 131 072 keys are sorted in LDS tiles and merged by rank (ks_tile_sort / ks_merge_tiles), larger ones
    by the six radix passes; FHX_K3_SMALL=0 forces the radix passes.  Many ties (values drawn from 1 000 levels), some NaN, every
    value below the cutoff (N = 1: nothing saturates, all n values are ranked): both paths must give the oracle's q bit for bit."""
    from oracle import fithic_oracle as fo
    rng = np.random.default_rng(n)
    p = rng.choice(rng.random(1000) * 1e-3, n)
    p[rng.integers(0, n, max(1, n // 500))] = np.nan
    p[rng.integers(0, n, 5)] = 0.0
    want = fo.benjamini_hochberg(p, 1.0)
    for env in (None, "0"):
        if env is None:
            monkeypatch.delenv("FHX_K3_SMALL", raising=False)
        else:
            monkeypatch.setenv("FHX_K3_SMALL", env)
        got = ctx.bh_array(p, 1.0)
        assert np.array_equal(np.isnan(got), np.isnan(want))
        assert bits_equal(np.nan_to_num(got, nan=-1.0), np.nan_to_num(want, nan=-1.0)), (n, env)


@pytest.mark.parametrize("n,shape", [(131073, "uniform"), (1000003, "skewed"), (5000011, "cluster"), (16000000, "uniform"),
                                     (3000000, "levels"), (2000000, "equal")])
def test_bh_large_survivor_sets_bit_exact_vs_oracle(ctx, n, shape):
    """The eight radix passes on 1.3e5 .. 1.6e7 survivors (N = 1: every value is ranked) in the shapes real p-value columns
    take: uniform; p = u^8 (the exponent-heavy skew of small p); a narrow cluster holding a third of the values beside a wide
    background; 1 000 levels and one single value (ties).  q must be the oracle's bit for bit."""
    from oracle import fithic_oracle as fo
    rng = np.random.default_rng(n)
    if shape == "uniform":
        p = rng.random(n) * 1e-2
    elif shape == "skewed":
        p = rng.random(n) ** 8
    elif shape == "cluster":
        p = rng.random(n) * 1e-3
        k = n // 3
        p[rng.integers(0, n, k)] = 3.1e-7 * (1.0 + 1e-9 * rng.random(k))
    elif shape == "levels":
        p = rng.choice(rng.random(1000) * 1e-3, n)
    else:
        p = np.full(n, 1.25e-5)
    p[rng.integers(0, n, 7)] = np.nan
    p[rng.integers(0, n, 5)] = 0.0
    want = fo.benjamini_hochberg(p, 1.0)
    got = ctx.bh_array(p, 1.0)
    assert np.array_equal(np.isnan(got), np.isnan(want))
    assert bits_equal(np.nan_to_num(got, nan=-1.0), np.nan_to_num(want, nan=-1.0)), (n, shape)


def test_bh_values_above_one_and_few_tests_follow_the_reference(ctx):
    """myStats.benjamini_hochberg_correction takes any numbers: with N < rank a p > 1 yields min(p*N/rank, 1) < 1, and only
    p == 1.0 exactly is pinned to 1.0 (myStats.py:33-38).  Nothing saturates for small N, so every row is ranked."""
    from oracle import fithic_oracle as fo
    rng = np.random.default_rng(21)
    n = 50_001
    p = rng.uniform(0, 1, n) ** rng.integers(1, 6, n)
    p[rng.integers(0, n, n // 5)] = 1.0
    p[rng.integers(0, n, n // 7)] = rng.uniform(1.0, 40.0, n // 7)             # not p-values, but legal inputs
    p[rng.integers(0, n, 20)] = np.inf
    p[rng.integers(0, n, 50)] = np.nan
    for N in (1.0, 10.0, 0.37 * n, float(n), 4.0 * n):
        assert bits_equal(ctx.bh_array(p, N), fo.benjamini_hochberg(p, N)), N


def test_bh_with_zero_or_negative_number_of_tests(ctx):
    """fit_Spline can pass a negative N (possible-pair counts go negative with unmappable loci; the fuzz campaign
    found it: seeds 1413, 2006, 2772, 4554): the reference's running maximum starts at 0, so every q is 0 (p == 1.0 stays 1,
    NaN stays NaN).  Vectors generated by the real myStats.benjamini_hochberg_correction (make_golden.py f12); -0.0 == 0.0."""
    g = np.load(os.path.join(GOLDEN, "f12_bh_nonpositive_N.npz"))
    for name in g["names"]:
        q, want = ctx.bh_array(g[name + "_p"], g[name + "_N"][0]), g[name + "_q"]
        assert np.array_equal(np.isnan(q), np.isnan(want)), name
        assert np.array_equal(q[~np.isnan(want)], want[~np.isnan(want)]), name


def test_bh_rejects_negative_values_and_bad_test_counts(ctx):
    """A negative "p-value" would index the key histogram past its end (keys are IEEE bit patterns): refused, loudly."""
    from fithic_amd._capi import FhxError
    p = np.array([0.2, -0.1, 0.5])
    with pytest.raises(FhxError, match="negative p-value at index 1"):
        ctx.bh_array(p, 10.0)
    with pytest.raises(FhxError, match="number of tests must be finite"):
        ctx.bh_array(np.array([0.2, 0.5]), float("nan"))
    assert np.array_equal(ctx.bh_array(np.array([0.2, -0.0, 0.5]), 10.0), np.array([1.0, 0.0, 1.0]))    # -0.0 is zero


def _run_case(name, device=0, totals="reference"):
    from fithic_amd import tables
    from fithic_amd.engine import Engine
    meta, g = load_case(name)
    kw = case_args(meta)
    chroms = tables.ChromIndex()
    con = tables.read_contacts(kw["contacts"], chroms)
    fc, fm, fh = tables.read_fragments(kw["frags"], chroms)
    eng = Engine(device)
    eng.configure(kw["resolution"], kw["L"], kw["U"], kw["n_bins"], kw["mapp_thres"], kw["mode"], kw["tL"], kw["tU"], totals)
    eng.load_fragments(fc, fm, fh, chroms.sort_rank())
    if kw["bias_path"]:
        eng.load_bias(*tables.read_bias(kw["bias_path"], chroms))
    eng.load_contacts(con.chr1, con.mid1, con.chr2, con.mid2, con.count)
    passes = []
    for pi in range(1, meta["n_passes"] + 1):
        out = eng.run_pass()
        out.values = eng.fetch(p=True, q=True, expcc=True, bias=True)
        out.fdr = eng.fdr_counts()
        out.n_outliers_total = eng.next_pass()
        passes.append(out)
    eng.close()
    return meta, g, kw, passes


@pytest.mark.parametrize("name", ALL_CASES)
def test_pipeline_matches_reference_goldens(name):
    meta, g, kw, passes = _run_case(name)
    res = kw["resolution"]
    sub = meta["subsample"]
    for pi, out in enumerate(passes, 1):
        P = "p%d_" % pi
        st = out.stats
        # integer results: bit-exact
        assert [st["inter_count"], st["inter_sum"], st["intra_all_sum"], st["in_range_sum"]] == [int(v) for v in g[P + "sums"]]
        if res and "dist_keys" not in out.arrays:
            keys = np.flatnonzero(out.arrays["hist_npairs"] > 0) * res
            assert np.array_equal(keys, g[P + "dist_keys"])
            assert np.array_equal(out.arrays["hist_sumcc"][keys // res], g[P + "dist_sumcc"])
        else:                                   # -r 0, or -r N on off-grid loci: explicit distinct distances
            assert np.array_equal(out.arrays["dist_keys"], g[P + "dist_keys"])
            assert np.array_equal(out.arrays["hist_sumcc"], g[P + "dist_sumcc"])
        for k, mine in (("lb", "bin_lb"), ("ub", "bin_ub"), ("s1", "bin_poss"), ("s2", "bin_sumcc"), ("s7", "bin_poss7")):
            assert np.array_equal(out.arrays[mine], g[P + "bins1_" + k]), k
        assert bits_equal(out.arrays["x"], g[P + "x"]) and bits_equal(out.arrays["y"], g[P + "y"])
        if P + "spl_t" in g:
            assert bits_equal(out.arrays["knots"], g[P + "spl_t"])
            assert np.array_equal(out.arrays["table_x"], g[P + "splineX"])
            assert bits_equal(out.arrays["table_y"], g[P + "newSplineY"])
        assert out.info["outlier_thres"] == g[P + "outlierThres"][0]
        v = out.values
        for key in ("b1", "b2"):
            assert bits_equal(v[key][::sub], g[P + key]), key
        dp = max_abs_diff(v["p"][::sub], g[P + "p"])
        dq = max_abs_diff(v["q"][::sub], g[P + "q"])
        de = max_abs_diff(v["expcc"][::sub], g[P + "expcc"])
        same_p = np.mean((v["p"][::sub].view(np.int64) == g[P + "p"].view(np.int64)) | np.isnan(g[P + "p"]))
        print("%s pass %d: max|dp| %.2e max|dq| %.2e max|dExpCC| %.2e, p bit-identical %.4f" % (name, pi, dp, dq, de, same_p))
        assert dp <= TOL and dq <= TOL
        assert bits_equal(v["expcc"][::sub], g[P + "expcc"])
        assert out.n_outliers_total == g[P + "n_outlier_lines"][0]
        assert np.array_equal(out.fdr, g[P + "fdr_y"])
        m = meta["pass%d" % pi]
        assert int(np.sum(v["q"] < 0.01)) == m["n_q_lt_0.01"] and int(np.sum(v["q"] < 0.05)) == m["n_q_lt_0.05"]
        assert int(np.sum(v["p"] < 1.0)) == m["n_p_lt_1"] and int(np.isnan(v["p"]).sum()) == m["n_nan_p"]


@pytest.mark.parametrize("name", SMALL_CASES + ["f1_bias"])
def test_pipeline_matches_oracle_every_row(name):
    """Full-length comparison (every row, every pass) with the oracle run on the same files."""
    from oracle import fithic_oracle as fo
    meta, g, kw, passes = _run_case(name)
    ref = fo.run(**kw)
    assert len(ref) == len(passes)
    for out, r in zip(passes, ref):
        v = out.values
        assert max_abs_diff(v["p"], r.p) <= TOL
        assert max_abs_diff(v["q"], r.q) <= TOL
        assert bits_equal(v["expcc"], r.expcc) and bits_equal(v["b1"], r.b1) and bits_equal(v["b2"], r.b2)
        # q is exactly BH of OUR p (the sort/scan path is bit-exact on its own input)
        assert bits_equal(v["q"], fo.benjamini_hochberg(v["p"], out.info["bh_total_tests"]))


@pytest.mark.parametrize("name", WRAP_CASES)
def test_totals_at_and_above_2p31_reference_mode_is_the_reference(name):
    """FHX_TOTALS_REFERENCE (the default): scipy's bdtrc narrows the Python-int totals of fithic.py:1070 / :1101 to a C int, and so
    does the engine - the NaN pattern, p, q and ExpCC of the real reference's run (fixtures made by make_golden.py f15; ExpCC keeps
    the true total, as the reference's `observedIntraInRangeSum * prior_p` does).  fhx_fit_info says what bdtrc was given."""
    from oracle import fithic_oracle as fo
    meta, g, kw, passes = _run_case(name)
    for pi, out in enumerate(passes, 1):
        P = "p%d_" % pi
        v, info, st = out.values, out.info, out.stats
        assert np.array_equal(np.isnan(v["p"]), np.isnan(g[P + "p"])) and np.array_equal(np.isnan(v["q"]), np.isnan(g[P + "q"]))
        assert max_abs_diff(v["p"], g[P + "p"]) <= TOL and max_abs_diff(v["q"], g[P + "q"]) <= TOL
        assert bits_equal(v["expcc"], g[P + "expcc"])
        assert info["totals"] == 0
        assert info["totals_narrowed"] == (1 if st["in_range_sum"] >= 2 ** 31 else 0) + (2 if st["inter_sum"] >= 2 ** 31 else 0)
        assert pi > 1 or info["totals_narrowed"] != 0             # (a later pass has lost its outliers' counts: it may fit again)
        assert info["bdtrc_n_intra"] == fo.int_narrowed(st["in_range_sum"]) and info["bdtrc_n_inter"] == fo.int_narrowed(st["inter_sum"])


@pytest.mark.parametrize("name", WRAP_CASES)
def test_totals_at_and_above_2p31_wide_mode_is_the_oracles_second_mode(name):
    """FHX_TOTALS_WIDE: bdtrc's arithmetic on the true totals.  No reference computes this - the check is against the oracle's
    explicitly named second mode (fho_bdtrc_wide, "parity unpinned" above 2^31) - and it must differ from the reference's output
    exactly where the narrowed total changed the answer."""
    from oracle import fithic_oracle as fo
    meta, g, kw, passes = _run_case(name, totals="wide")
    ref = fo.run(totals="wide", **kw)
    assert len(ref) == len(passes)
    for pi, (out, r) in enumerate(zip(passes, ref), 1):
        v = out.values
        assert max_abs_diff(v["p"], r.p) <= TOL and max_abs_diff(v["q"], r.q) <= TOL
        assert bits_equal(v["expcc"], r.expcc)
        assert out.info["totals"] == 1 and (pi > 1 or out.info["totals_narrowed"] != 0)
        assert out.info["bdtrc_n_intra"] == out.stats["in_range_sum"] and out.info["bdtrc_n_inter"] == out.stats["inter_sum"]
        if pi == 1:
            gp = g["p1_p"]
            differs = np.isnan(gp) != np.isnan(v["p"])
            differs |= ~np.isnan(gp) & ~np.isnan(v["p"]) & (np.abs(gp - v["p"]) > 1e-6)
            assert differs.sum() > 100, "the two modes must disagree on these inputs"


def test_totals_below_2p31_the_two_modes_are_the_same_function():
    a = _run_case("f6_quirk_all")[3]
    b = _run_case("f6_quirk_all", totals="wide")[3]
    for x, y in zip(a, b):
        assert bits_equal(x.values["p"], y.values["p"]) and bits_equal(x.values["q"], y.values["q"])
        assert x.info["totals_narrowed"] == y.info["totals_narrowed"] == 0


def test_off_grid_loci_run_through_the_slotting_path():
    """-r N on midpoints that do not share a grid: accepted (the reference takes abs(mid1 - mid2) of whatever the files hold);
    the distance histogram is keyed by the distinct distances.  The f11_offgrid_* goldens check the values."""
    from fithic_amd import _capi
    c = _capi.Context(0)
    c.set_params(10000)
    c.load_pairs([0, 0, 0], [5000, 15001, 5000], [0, 0, 0], [25000, 35000, 35000], [3, 4, 5])
    st = c.pass_stats()
    assert st.in_range_sum == 12 and st.n_dist == 3
    assert np.array_equal(c.get_array(_capi.A_DIST_KEYS), [19999, 20000, 30000])
    assert np.array_equal(c.get_array(_capi.A_HIST_SUMCC)[:3], [4, 3, 5])
    c.load_pairs([0, 0], [5000, 15000], [0, 0], [25000, 35000], [3, 4])          # on-grid rows again: the dense path
    c.pass_stats()
    assert len(c.get_array(_capi.A_DIST_KEYS)) == 0 and np.array_equal(np.flatnonzero(c.get_array(_capi.A_HIST_NPAIRS)), [2])
    c.close()


@pytest.mark.parametrize("name", ["f6_quirk_all", "f2_all"])
def test_three_passes_match_oracle(name):
    """-p 3: the reference's duplicated outlier line numbers stop the skipping after the first duplicate."""
    from oracle import fithic_oracle as fo
    from fithic_amd import tables
    from fithic_amd.engine import Engine
    meta, g = load_case(name)
    kw = case_args(meta)
    kw["passes"] = 3
    ref = fo.run(**kw)
    chroms = tables.ChromIndex()
    con = tables.read_contacts(kw["contacts"], chroms)
    eng = Engine(0)
    eng.configure(kw["resolution"], kw["L"], kw["U"], kw["n_bins"], kw["mapp_thres"], kw["mode"], kw["tL"], kw["tU"])
    eng.load_fragments(*tables.read_fragments(kw["frags"], chroms), chroms.sort_rank())
    eng.load_bias(*tables.read_bias(kw["bias_path"], chroms))
    eng.load_contacts(con.chr1, con.mid1, con.chr2, con.mid2, con.count)
    for r in ref:
        out = eng.run_pass()
        v = eng.fetch()
        assert [out.stats["inter_count"], out.stats["inter_sum"], out.stats["intra_all_sum"], out.stats["in_range_sum"]] == list(r.sums)
        assert max_abs_diff(v["p"], r.p) <= TOL and max_abs_diff(v["q"], r.q) <= TOL
        assert eng.next_pass() == r.n_outlier_lines_total
    eng.close()


@pytest.mark.parametrize("name", ["f1_bias", "f8_nonfixed_all"])
def test_kernel_seconds_summed_over_passes(name):
    """fhx_kernel_seconds_total: the library adds every pass's HIP-event durations (K1, K2, K3, the heavy launch) to its sums at
    points where it waits for the stream anyway, so that a timing harness does not stop the stream after each pass.  Three passes
    give three samples of each, the sums are at least the last pass's own durations, and a reset clears them."""
    from fithic_amd import tables
    from fithic_amd.engine import Engine
    meta, g = load_case(name)
    kw = case_args(meta)
    chroms = tables.ChromIndex()
    con = tables.read_contacts(kw["contacts"], chroms)
    eng = Engine(0)
    eng.configure(kw["resolution"], kw["L"], kw["U"], kw["n_bins"], kw["mapp_thres"], kw["mode"], kw["tL"], kw["tU"])
    eng.load_fragments(*tables.read_fragments(kw["frags"], chroms), chroms.sort_rank())
    if kw["bias_path"]:
        eng.load_bias(*tables.read_bias(kw["bias_path"], chroms))
    eng.load_contacts(con.chr1, con.mid1, con.chr2, con.mid2, con.count)
    eng.run_pass(collect=False)
    eng.ctx.kernel_seconds_total(reset=True)
    assert eng.ctx.kernel_seconds_total() == ([0.0] * 4, [0] * 4)
    for _ in range(3):
        eng.run_pass(collect=False)
    last = eng.kernel_seconds()
    sums, counts = eng.ctx.kernel_seconds_total()
    assert counts == [3, 3, 3, 3]
    assert all(s >= l > 0.0 for s, l in zip(sums[:3], last)) and 0.0 < sums[3] <= sums[1]
    assert eng.ctx.kernel_seconds_total(reset=True)[1] == [3, 3, 3, 3]      # nothing is counted twice
    assert eng.ctx.kernel_seconds_total()[1] == [0, 0, 0, 0]
    # A kernel group run again before its event pair was read (K2 and K3 repeated with no statistics call in between) loses no
    # pass silently: the pair is read where the stream has passed it, else counted in kernel_events_dropped - folded + dropped is
    # what ran.
    info = eng.ctx.fit()
    for _ in range(4):
        eng.ctx.pvalues()
        eng.ctx.bh(info.bh_total_tests)
    _, counts = eng.ctx.kernel_seconds_total()
    dropped = eng.ctx.kernel_events_dropped()
    assert counts[1] + dropped[1] == 4 and counts[2] + dropped[2] == 4 and dropped[0] == 0
    eng.ctx.kernel_seconds_total(reset=True)
    assert eng.ctx.kernel_events_dropped() == [0, 0, 0, 0]
    eng.close()


@pytest.mark.parametrize("name", ["f1_nobias", "f1_bias", "f2_all", "f6_quirk_all", "f7_pfal_all", "f8_nonfixed_hESC", "f8_nonfixed_all",
                                  "f13_all_p3", "f13_quirk_p4", "f13_hESC_p3"] + WRAP_CASES)
def test_cli_writes_the_reference_files(name, tmp_path, capsys):
    """The drop-in command line: decompressed .significances.txt and .fithic_passN.txt equal the reference's byte for byte."""
    import gzip
    import hashlib
    from fithic_amd import cli
    meta, g = load_case(name)
    kw = case_args(meta)
    argv = ["-i", kw["contacts"], "-f", kw["frags"], "-o", str(tmp_path), "-l", "G"] + meta["argv"]
    if kw["bias_path"]:
        argv += ["-t", kw["bias_path"]]
    cli.main(argv)
    res = kw["resolution"]
    tag = (".res%d" % res) if res else ""
    for pi in range(1, meta["n_passes"] + 1):
        with gzip.open(os.path.join(str(tmp_path), "G.spline_pass%d%s.significances.txt.gz" % (pi, tag)), "rb") as f:
            text = f.read()
        assert text.count(b"\n") - 1 == meta["sig_rows_pass%d" % pi]
        assert hashlib.md5(text).hexdigest() == meta["sig_md5_pass%d" % pi]
        with open(os.path.join(str(tmp_path), "G.fithic_pass%d%s.txt" % (pi, tag))) as f:
            assert f.read() == meta["fithic_pass%d_txt" % pi]
    # the log keeps the reference's text (it is truncated by every read_Interactions); paths differ
    with open(os.path.join(str(tmp_path), "G.fithic.log")) as f:
        mine = [ln for ln in f.read().splitlines() if not ln.startswith("Means and error written")]
    want = [ln for ln in meta["log_txt"].splitlines() if not ln.startswith("Means and error written")]
    assert mine == want
    # a total at or above 2^31: one line per pass on stderr says which semantics ran (and only then)
    err = capsys.readouterr().err
    assert ("scipy's bdtrc narrows n to a C int" in err) == (name in WRAP_CASES)


def test_cli_totals_wide_says_so_and_differs_from_the_reference(tmp_path, capsys):
    import gzip
    import hashlib
    from fithic_amd import cli
    meta, g = load_case("f15_intra_2p31_all")
    kw = case_args(meta)
    cli.main(["-i", kw["contacts"], "-f", kw["frags"], "-o", str(tmp_path), "-l", "G", "-t", kw["bias_path"], "--totals", "wide"] + meta["argv"])
    err = capsys.readouterr().err
    assert "--totals wide" in err and "observedIntraInRangeSum >= 2^31" in err and "observedInterAllSum" not in err
    with gzip.open(os.path.join(str(tmp_path), "G.spline_pass1.res10000.significances.txt.gz"), "rb") as f:
        text = f.read()
    assert text.count(b"\n") - 1 == meta["sig_rows_pass1"]
    assert hashlib.md5(text).hexdigest() != meta["sig_md5_pass1"]
    assert text.count(b"nan") < 100          # the reference's file has 1237 nan rows (x 2 columns) here


@pytest.mark.parametrize("split", ["file", "chromosome"])
@pytest.mark.parametrize("name,gpus", [("f1_bias", 2), ("f2_all", 3), ("f6_quirk_all", 2), ("f13_all_p3", 2), ("f13_quirk_p4", 3),
                                       ("f8_nonfixed_all", 2), ("f8_nonfixed_hESC", 3), ("f11_offgrid_all", 3), ("f15_intra_2p31_all", 2),
                                       ("f15_both_2p32_all", 3)])
def test_cli_gpus_n_writes_the_same_files(name, gpus, split, tmp_path, monkeypatch, capsys):
    """`fithic --gpus N`: rows sharded over N ranks (worker processes), genome-wide steps through the library's communicator, ONE
    output set - byte-identical to the reference's.  The fixtures are plain gzip files: every rank inflates them on the host and
    takes the rows that start in its N-th of the text (split "file"), or parses all of it and keeps its chromosomes
    (FHX_CLI_SPLIT=chromosome).  On this one-GPU box the ranks share GPU 0 and the collectives travel over pipes
    (FHX_CLI_TRANSPORT=pipes; RCCL refuses two ranks on one device)."""
    import gzip
    import hashlib
    from fithic_amd import cli
    monkeypatch.setenv("FHX_CLI_TRANSPORT", "pipes")
    if split == "chromosome":
        monkeypatch.setenv("FHX_CLI_SPLIT", "chromosome")
    monkeypatch.setenv("FHX_CLI_DEVICES", ",".join(["0"] * gpus))
    monkeypatch.setenv("FHX_TIMING", "1")
    meta, g = load_case(name)
    kw = case_args(meta)
    argv = ["-i", kw["contacts"], "-f", kw["frags"], "-o", str(tmp_path), "-l", "G", "--gpus", str(gpus)] + meta["argv"]
    if kw["bias_path"]:
        argv += ["-t", kw["bias_path"]]
    cli.main(argv)
    # no row went through rank 0: every rank read the file on its own GPU and wrote its own stretches of the output
    said = capsys.readouterr().out
    if split == "file":
        assert "every rank: inflate on the host, parse its part of the text" in said and "(device parser)" in said
    else:
        assert "every rank: inflate + parse + keep its chromosomes" in said
        # (two of the files are not sorted by chromosome - a rank's rows would be thousands of separate stretches of the output:
        # split by chromosome those take the route through rank 0 after all, tested on its own below)
        assert ("(device parser)" in said) == (name not in ("f2_all", "f13_all_p3"))
    assert not [f for f in os.listdir(str(tmp_path)) if ".part-" in f or ".fhx-tmp" in f]
    tag = ".res%d" % kw["resolution"] if kw["resolution"] else ""          # -r 0: no resolution in the file names
    for pi in range(1, meta["n_passes"] + 1):
        with gzip.open(os.path.join(str(tmp_path), "G.spline_pass%d%s.significances.txt.gz" % (pi, tag)), "rb") as f:
            text = f.read()
        assert hashlib.md5(text).hexdigest() == meta["sig_md5_pass%d" % pi]
        with open(os.path.join(str(tmp_path), "G.fithic_pass%d%s.txt" % (pi, tag))) as f:
            assert f.read() == meta["fithic_pass%d_txt" % pi]
    with open(os.path.join(str(tmp_path), "G.fithic.log")) as f:
        mine = [ln for ln in f.read().splitlines() if not ln.startswith("Means and error written")]
    want = [ln for ln in meta["log_txt"].splitlines() if not ln.startswith("Means and error written")]
    assert mine == want


@pytest.mark.parametrize("name,gpus,chunk", [("f1_bias", 2, 20000), ("f2_all", 3, 4096), ("f13_all_p3", 4, 10000), ("f7_pfal_all", 3, 3000),
                                             ("f8_nonfixed_hESC", 2, 50000), ("f13_hESC_p3", 5, 8192)])
def test_cli_gpus_n_inflates_one_plain_gzip_stream_in_parts(name, gpus, chunk, tmp_path, monkeypatch, capsys):
    """ONE plain gzip stream (what `gzip` and the reference's own utilities write; fithic.py:404 reads it with gzip.open): rank r
    decodes its N-th of the COMPRESSED bytes without the 32 KB before them, the ranks' tails are chained into every part's window,
    CRC-32 and length are checked against the trailer, and a row belongs to the rank whose text holds its first byte
    (sharded._ingest_stream_parts).  The fixtures are far below the size from which that pays: FHX_PGUNZIP_MIN / _CHUNK let them
    through, in chunks of a few blocks.  The files written are the reference's."""
    import gzip
    import hashlib
    from fithic_amd import cli
    monkeypatch.setenv("FHX_CLI_TRANSPORT", "pipes")
    monkeypatch.setenv("FHX_CLI_DEVICES", ",".join(["0"] * gpus))
    monkeypatch.setenv("FHX_TIMING", "1")
    monkeypatch.setenv("FHX_PGUNZIP_MIN", "0")
    monkeypatch.setenv("FHX_PGUNZIP_CHUNK", str(chunk))
    meta, g = load_case(name)
    kw = case_args(meta)
    argv = ["-i", kw["contacts"], "-f", kw["frags"], "-o", str(tmp_path), "-l", "G", "--gpus", str(gpus)] + meta["argv"]
    if kw["bias_path"]:
        argv += ["-t", kw["bias_path"]]
    cli.main(argv)
    said = capsys.readouterr().out
    assert "every rank: inflate its part of the stream on the host, parse its rows" in said and "(device parser)" in said
    assert not [f for f in os.listdir(str(tmp_path)) if ".part-" in f or ".fhx-tmp" in f]
    tag = ".res%d" % kw["resolution"] if kw["resolution"] else ""
    for pi in range(1, meta["n_passes"] + 1):
        with gzip.open(os.path.join(str(tmp_path), "G.spline_pass%d%s.significances.txt.gz" % (pi, tag)), "rb") as f:
            assert hashlib.md5(f.read()).hexdigest() == meta["sig_md5_pass%d" % pi]
        with open(os.path.join(str(tmp_path), "G.fithic_pass%d%s.txt" % (pi, tag))) as f:
            assert f.read() == meta["fithic_pass%d_txt" % pi]
    with open(os.path.join(str(tmp_path), "G.fithic.log")) as f:
        mine = [ln for ln in f.read().splitlines() if not ln.startswith("Means and error written")]
    assert mine == [ln for ln in meta["log_txt"].splitlines() if not ln.startswith("Means and error written")]


def _tagged_copy(src, dst, n_members, on_rows=True):
    """the gzip file `src` rewritten as `n_members` members that carry their sizes (the "FH" extra field of this library's writers,
    include/fithic_mi355x.h); on_rows=False: cut by bytes, so that members end in the middle of a row"""
    import gzip
    import struct
    import zlib
    with gzip.open(src, "rb") as f:
        text = f.read()
    if on_rows:
        lines = text.splitlines(keepends=True)
        per = max(1, -(-len(lines) // n_members))
        chunks = [b"".join(lines[k:k + per]) for k in range(0, len(lines), per)]
    else:
        per = max(1, -(-len(text) // n_members))
        chunks = [text[k:k + per] for k in range(0, len(text), per)]
    with open(dst, "wb") as out:
        for c in chunks:
            z = zlib.compressobj(6, zlib.DEFLATED, -15)
            body = z.compress(c) + z.flush()
            out.write(b"\x1f\x8b\x08\x04\0\0\0\0\0\xff" + struct.pack("<H2sHQ", 12, b"FH", 8, 24 + len(body) + 8))
            out.write(body + struct.pack("<II", zlib.crc32(c), len(c) & 0xffffffff))
    return dst


@pytest.mark.parametrize("name,gpus,members", [("f1_bias", 2, 7), ("f2_all", 3, 11), ("f13_all_p3", 2, 5), ("f13_quirk_p4", 3, 9),
                                               ("f11_offgrid_all", 3, 8), ("f8_nonfixed_all", 2, 6), ("f6_quirk_all", 4, 3)])
def test_cli_gpus_n_cuts_a_file_of_tagged_members_into_parts(name, gpus, members, tmp_path, monkeypatch, capsys):
    """A contacts file whose gzip members carry their sizes (this library's writers, bgzip) is cut into N parts of whole members:
    every rank inflates and parses only its part, holds one stretch of the output, and a chromosome's rows may lie on several
    ranks (files not sorted by chromosome included: f2_all, f13_all_p3) - the files written are the reference's.  With fewer
    members than ranks (f6_quirk_all on 4) some ranks hold no rows."""
    import gzip
    import hashlib
    from fithic_amd import cli
    monkeypatch.setenv("FHX_CLI_TRANSPORT", "pipes")
    monkeypatch.setenv("FHX_CLI_DEVICES", ",".join(["0"] * gpus))
    monkeypatch.setenv("FHX_TIMING", "1")
    meta, g = load_case(name)
    kw = case_args(meta)
    contacts = _tagged_copy(kw["contacts"], str(tmp_path / "contacts.tagged.gz"), members)
    out = tmp_path / "out"
    out.mkdir()
    argv = ["-i", contacts, "-f", kw["frags"], "-o", str(out), "-l", "G", "--gpus", str(gpus)] + meta["argv"]
    if kw["bias_path"]:
        argv += ["-t", kw["bias_path"]]
    cli.main(argv)
    said = capsys.readouterr().out
    assert "every rank: inflate + parse its part of the file" in said and "(device parser)" in said
    assert not [f for f in os.listdir(str(out)) if ".part-" in f or ".fhx-tmp" in f]
    tag = ".res%d" % kw["resolution"] if kw["resolution"] else ""
    for pi in range(1, meta["n_passes"] + 1):
        with gzip.open(os.path.join(str(out), "G.spline_pass%d%s.significances.txt.gz" % (pi, tag)), "rb") as f:
            assert hashlib.md5(f.read()).hexdigest() == meta["sig_md5_pass%d" % pi]
        with open(os.path.join(str(out), "G.fithic_pass%d%s.txt" % (pi, tag))) as f:
            assert f.read() == meta["fithic_pass%d_txt" % pi]
    with open(os.path.join(str(out), "G.fithic.log")) as f:
        mine = [ln for ln in f.read().splitlines() if not ln.startswith("Means and error written")]
    assert mine == [ln for ln in meta["log_txt"].splitlines() if not ln.startswith("Means and error written")]


def test_cli_gpus_n_cuts_the_text_when_members_end_inside_a_row(tmp_path, monkeypatch, capsys):
    """parts of the FILE are only taken when each ends a row: members cut anywhere (a bgzip file) are inflated by every rank on the
    host, and the TEXT is cut on row starts"""
    import gzip
    import hashlib
    from fithic_amd import cli
    monkeypatch.setenv("FHX_CLI_TRANSPORT", "pipes")
    monkeypatch.setenv("FHX_CLI_DEVICES", "0,0,0")
    monkeypatch.setenv("FHX_TIMING", "1")
    meta, g = load_case("f1_bias")
    kw = case_args(meta)
    contacts = _tagged_copy(kw["contacts"], str(tmp_path / "contacts.tagged.gz"), 5, on_rows=False)
    out = tmp_path / "out"
    out.mkdir()
    cli.main(["-i", contacts, "-f", kw["frags"], "-o", str(out), "-l", "G", "--gpus", "3", "-t", kw["bias_path"]] + meta["argv"])
    said = capsys.readouterr().out
    assert "every rank: inflate on the host, parse its part of the text" in said and "(device parser)" in said
    for pi in range(1, meta["n_passes"] + 1):
        with gzip.open(os.path.join(str(out), "G.spline_pass%d.res%d.significances.txt.gz" % (pi, kw["resolution"])), "rb") as f:
            assert hashlib.md5(f.read()).hexdigest() == meta["sig_md5_pass%d" % pi]


@pytest.mark.parametrize("name,gpus", [("f2_all", 2), ("f13_quirk_p4", 3)])
def test_cli_gpus_n_with_rows_handed_out_by_rank_0(name, gpus, tmp_path, monkeypatch):
    """the older route, still what a file outside the device parser's grammar takes: rank 0 parses on the host, hands the columns
    out and gathers p, q, ExpCC and the biases for the one writer (FHX_CLI_FUNNEL=1 forces it)"""
    import gzip
    import hashlib
    from fithic_amd import cli
    monkeypatch.setenv("FHX_CLI_TRANSPORT", "pipes")
    monkeypatch.setenv("FHX_CLI_DEVICES", ",".join(["0"] * gpus))
    monkeypatch.setenv("FHX_CLI_FUNNEL", "1")
    meta, g = load_case(name)
    kw = case_args(meta)
    argv = ["-i", kw["contacts"], "-f", kw["frags"], "-o", str(tmp_path), "-l", "G", "--gpus", str(gpus)] + meta["argv"]
    if kw["bias_path"]:
        argv += ["-t", kw["bias_path"]]
    cli.main(argv)
    tag = ".res%d" % kw["resolution"] if kw["resolution"] else ""
    for pi in range(1, meta["n_passes"] + 1):
        with gzip.open(os.path.join(str(tmp_path), "G.spline_pass%d%s.significances.txt.gz" % (pi, tag)), "rb") as f:
            assert hashlib.md5(f.read()).hexdigest() == meta["sig_md5_pass%d" % pi]


def test_cli_visual_flag_writes_figures_and_same_tables(tmp_path):
    """-v (fithic.py:225-229, 374-376): the figures appear and the significances stay byte-identical."""
    import gzip
    import hashlib
    pytest.importorskip("matplotlib")
    from fithic_amd import cli
    meta, g = load_case("f1_bias")
    kw = case_args(meta)
    cli.main(["-i", kw["contacts"], "-f", kw["frags"], "-o", str(tmp_path), "-l", "G", "-t", kw["bias_path"], "-v"] + meta["argv"])
    tag = ".res%d" % kw["resolution"]
    want = ["G.spline_pass%d.png" % pi for pi in range(1, meta["n_passes"] + 1)]
    want += ["G.spline_pass%d.qplot.png" % pi for pi in range(1, meta["n_passes"] + 1)]
    if meta["n_passes"] > 1:
        want += ["G.spline_FDR_comparison.png", "G.spline_comparison.png"]
    for name in want:
        f = tmp_path / name
        assert f.exists() and f.read_bytes()[:4] == b"\x89PNG"
    for pi in range(1, meta["n_passes"] + 1):
        with gzip.open(str(tmp_path / ("G.spline_pass%d%s.significances.txt.gz" % (pi, tag))), "rb") as f:
            assert hashlib.md5(f.read()).hexdigest() == meta["sig_md5_pass%d" % pi]


def test_mirror_benjamini_hochberg_signature():
    from fithic_amd import fithic as F
    q = F.benjamini_hochberg_correction([0.03, 0.4, 0.7, 0.01], 10)
    g = np.load(os.path.join(GOLDEN, "f5_bh.npz"))
    assert isinstance(q, list) and bits_equal(np.array(q), g["tiny_q"])
    F.reset_session()


def test_ctypes_only_example_matches_the_oracle(tmp_path):
    """examples/ctypes_minimal.py drives the C ABI with bare ctypes (the binding INTEGRATION.md describes); its p and q are the
    oracle's on the same tables, within 1e-10."""
    import importlib.util
    from oracle import fithic_oracle as fo
    spec = importlib.util.spec_from_file_location("ctypes_minimal", os.path.join(os.path.dirname(GOLDEN), "..", "examples", "ctypes_minimal.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    p, q, x = mod.main(with_inputs=True)
    names = ["c0", "c1"]
    c1, m1, c2, m2, cnt = x["cols"]
    pairs = fo.Pairs(c1, m1, c2, m2, cnt, names)
    frags = [(names[c], int(m), 1) for c, m in zip(x["f_chr"], x["f_mid"])]
    bias = np.where((x["bias"] < 0.5) | (x["bias"] > 2.0), -1.0, x["bias"])
    bias_dic = {n: {} for n in names}
    for c, m, b in zip(x["f_chr"], x["f_mid"], bias):
        bias_dic[names[c]].setdefault(int(m), float(b))
    ref = fo.run(pairs, frags, None, x["res"], n_bins=x["n_bins"], passes=1, mode="intraOnly", L=x["L"], U=x["U"], bias_dic=bias_dic)[0]
    assert len(p) > 10000 and np.nanmin(p) < 1e-3
    assert max_abs_diff(p, ref.p) <= 1e-10 and max_abs_diff(q, ref.q) <= 1e-10


@pytest.mark.parametrize("name", ["f1_bias", "f2_all", "f8_nonfixed_all"])
def test_outlier_rows_compacted_on_the_device_equal_the_flag_bytes(name):
    """fhx_fetch_outlier_rows = the ascending positions of the bytes fhx_fetch_flags sets (fithic.py:1215: p < 1/N)"""
    from test_gpu_writer import _engine_with_case
    eng, kw, chroms, con = _engine_with_case(name)
    eng.run_pass(collect=False)
    flags, _ = eng.ctx.fetch_flags(len(con), outlier=True, skip=False)
    rows = eng.ctx.fetch_outlier_rows()
    assert rows.dtype == np.int64 and np.array_equal(rows, np.flatnonzero(flags))
    eng.close()


def test_mirror_read_biases_returns_the_reference_dictionary_lazily():
    """read_biases gives {chrom: {mid: bias}} (fithic.py:798-837; first occurrence wins, out-of-range -> -1): the engine does not
    need it, so it is built when first looked at - and must then be the oracle's dictionary"""
    from fithic_amd import fithic as F
    from oracle import fithic_oracle as fo
    meta, _ = load_case("f1_bias")
    kw = case_args(meta)
    F.reset_session()
    F.gpus = 1
    F.resolution = kw["resolution"]
    F.biasLowerBound, F.biasUpperBound = kw["tL"], kw["tU"]
    F.distLowThres, F.distUpThres = kw["L"], kw["U"]
    d = F.read_biases(kw["bias_path"])
    assert d and dict.__len__(d) == 0                         # truthy like the reference's dict, nothing built yet
    want = fo.read_biases(kw["bias_path"], kw["tL"], kw["tU"])
    chrom = next(iter(want))
    assert chrom in d and dict.__len__(d) == len(want)        # the first look fills it
    assert d == want and len(d) == len(want) and sorted(d.keys()) == sorted(want.keys())
    mid = next(iter(want[chrom]))
    assert d[chrom][mid] == want[chrom][mid] and d.get("no such chromosome") is None
    assert {k: len(v) for k, v in d.items()} == {k: len(v) for k, v in want.items()}
    F.reset_session()


def _mirror_globals(F, kw):
    F.reset_session()
    F.resolution = kw["resolution"]
    F.biasLowerBound, F.biasUpperBound = kw["tL"], kw["tU"]
    F.distLowThres, F.distUpThres = kw["L"], kw["U"]
    F.mappThres = kw["mapp_thres"]
    F.interOnly, F.allReg = kw["mode"] == "interOnly", kw["mode"] == "All"
    F.logfile, F.visual, F.gpus = None, False, 1


@pytest.mark.parametrize("name", ["f1_bias", "f6_quirk_all"])
def test_mirror_stages_follow_the_mainDic_they_are_handed(name, tmp_path):
    """The reference's stage functions work on their ARGUMENTS (fithic.py:463, 843, 925): a caller may edit mainDic between
    read_Interactions and makeBinsFromInteractions.  The mirrors must then bin / fit / score that dict - checked against the
    oracle's stages run on the same edited histogram - and must refuse x, y or binStats that were changed behind the engine's
    back instead of ignoring them."""
    from fithic_amd import fithic as F
    from oracle import fithic_oracle as fo
    meta, _ = load_case(name)
    kw = case_args(meta)
    _mirror_globals(F, kw)
    try:
        biasDic = F.read_biases(kw["bias_path"]) if kw["bias_path"] else 0
        mainDic, icnt, isum, intra_all, rng_sum = F.read_Interactions(kw["contacts"], kw["bias_path"])
        # the caller's edit: every third distance loses half its contacts, two distances disappear
        keys = sorted(mainDic)
        edited = {k: [0, mainDic[k][1] - (mainDic[k][1] // 2 if i % 3 == 0 else 0)] for i, k in enumerate(keys)}
        for k in (keys[1], keys[len(keys) // 2]):
            del edited[k]
        new_sum = sum(v[1] for v in edited.values())
        assert new_sum != rng_sum
        binStats = F.makeBinsFromInteractions(edited, kw["n_bins"], new_sum)
        out = F.generate_FragPairs(icnt, isum, binStats, kw["frags"], kw["resolution"])
        binStats = out[0]
        x, y, yerr = F.calculateProbabilities(edited, binStats, kw["resolution"], str(tmp_path / "pass1"), new_sum)
        # the oracle on the same histogram
        pairs = fo.read_contacts_file(kw["contacts"])
        frag_rows = fo.read_fragments_file(kw["frags"])
        okeys = np.array(sorted(edited), np.int64)
        osum = np.array([edited[int(k)][1] for k in okeys], np.int64)
        bins = fo.make_bins(okeys, osum, kw["n_bins"], new_sum)
        frag = fo.generate_frag_pairs(frag_rows, bins, kw["resolution"], kw["L"], kw["U"], kw["mapp_thres"], icnt)
        ox, oy, _ = fo.calculate_probabilities(bins, new_sum)
        assert [b["lb"] for b in bins] == [binStats[i][0][0] for i in sorted(binStats)]
        assert [b["ub"] for b in bins] == [binStats[i][0][1] for i in sorted(binStats)]
        assert bits_equal(np.array(x), np.array(ox, float)) and bits_equal(np.array(y), np.array(oy, float))
        bias_dic = fo.read_biases(kw["bias_path"], kw["tL"], kw["tU"]) if kw["bias_path"] else 0
        b1, b2 = fo.gather_bias(pairs, bias_dic)
        R = fo.fit_spline(pairs, okeys, ox, oy, b1, b2, kw["mode"], kw["L"], kw["U"], kw["tL"], kw["tU"],
                          (icnt, isum, intra_all, new_sum), frag)
        # changed x: refused, not ignored
        with pytest.raises(ValueError, match="differs from what the engine computed"):
            F.fit_Spline(edited, [v * 2 for v in x], y, yerr, kw["contacts"], str(tmp_path / "o"), biasDic, [], [], new_sum,
                         out[3], out[4], icnt, intra_all, isum, kw["tL"], kw["tU"], kw["resolution"], 1)
        res = F.fit_Spline(edited, x, y, yerr, kw["contacts"], str(tmp_path / "o"), biasDic, [], [], new_sum,
                           out[3], out[4], icnt, intra_all, isum, kw["tL"], kw["tU"], kw["resolution"], 1)
        if R.newSplineY is not None:
            assert bits_equal(np.asarray(res[1]), np.asarray(R.newSplineY, float))
        v = F._S.engine.fetch(p=True, q=True)
        assert max_abs_diff(v["p"], R.p) <= TOL and max_abs_diff(v["q"], R.q) <= TOL
        assert sorted(res[3]) == sorted(R.outlier_lines.tolist())
        # untouched dict: nothing is replaced (the object and its contents are what read_Interactions handed out)
        mainDic2, *_rest = F.read_Interactions(kw["contacts"], kw["bias_path"])
        F.makeBinsFromInteractions(mainDic2, kw["n_bins"], _rest[-1])
        assert F._S.main_dic is mainDic2
        binStats2 = F.generate_FragPairs(icnt, isum, {}, kw["frags"], kw["resolution"])[0]
        bad = dict(binStats2)
        bad[0] = [(binStats2[0][0][0], binStats2[0][0][1] + kw["resolution"])] + list(binStats2[0][1:])
        with pytest.raises(ValueError, match="binStats"):
            F.calculateProbabilities(mainDic2, bad, kw["resolution"], str(tmp_path / "pass1b"), _rest[-1])
    finally:
        F.reset_session()


@pytest.mark.parametrize("name", ["f6_quirk_all", "f13_hESC_p3"])
def test_mirror_makeBins_follows_the_outlier_distances_it_is_handed(name, tmp_path):
    """Pass 2 of the reference subtracts one possible pair per entry of the `outliersdist` ARGUMENT (fithic.py:533-551).  The
    list fit_Spline returned is the engine's own multiset; an edited copy must replace it for that call (checked against the
    oracle's make_bins on the same list), and read_Interactions must refuse an outlier-line list it cannot honour."""
    from fithic_amd import fithic as F
    from oracle import fithic_oracle as fo
    meta, _ = load_case(name)
    kw = case_args(meta)
    _mirror_globals(F, kw)
    try:
        biasDic = F.read_biases(kw["bias_path"]) if kw["bias_path"] else 0
        mainDic, icnt, isum, intra_all, rng_sum = F.read_Interactions(kw["contacts"], kw["bias_path"])
        binStats = F.makeBinsFromInteractions(mainDic, kw["n_bins"], rng_sum)
        out = F.generate_FragPairs(icnt, isum, binStats, kw["frags"], kw["resolution"])
        x, y, yerr = F.calculateProbabilities(mainDic, out[0], kw["resolution"], str(tmp_path / "pass1"), rng_sum)
        res = F.fit_Spline(mainDic, x, y, yerr, kw["contacts"], str(tmp_path / "o1"), biasDic, [], [], rng_sum, out[3], out[4], icnt,
                           intra_all, isum, kw["tL"], kw["tU"], kw["resolution"], 1)
        outliersline, outliersdist = res[3], res[4]
        assert len(outliersline) > 4 and len(outliersdist) == len(outliersline)
        with pytest.raises(ValueError, match="cannot be honoured"):
            F.read_Interactions(kw["contacts"], kw["bias_path"], list(outliersline)[:-1])
        mainDic2, icnt2, isum2, intra_all2, rng_sum2 = F.read_Interactions(kw["contacts"], kw["bias_path"], outliersline)
        keys = np.array(sorted(mainDic2), np.int64)
        sums = np.array([mainDic2[int(k)][1] for k in keys], np.int64)
        # the engine's own list, passed back untouched
        own = F.makeBinsFromInteractions(mainDic2, kw["n_bins"], rng_sum2, outliersdist)
        want = fo.make_bins(keys, sums, kw["n_bins"], rng_sum2, outliersdist)
        assert [own[b][1] for b in sorted(own)] == [b["s1"] for b in want]
        # an edited copy: every other distance dropped, one far beyond the last bin added
        edited = sorted(list(outliersdist)[::2] + [int(keys[-1]) + 50 * max(kw["resolution"], 1)])
        got = F.makeBinsFromInteractions(mainDic2, kw["n_bins"], rng_sum2, edited)
        want = fo.make_bins(keys, sums, kw["n_bins"], rng_sum2, edited)
        assert [got[b][1] for b in sorted(got)] == [b["s1"] for b in want]
        assert [got[b][1] for b in sorted(got)] != [own[b][1] for b in sorted(own)]
    finally:
        F.reset_session()
