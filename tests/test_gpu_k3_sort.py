"""K3's large sort (fhx_onesweep.inc): one-sweep radix passes over the top key bits + the LDS repair of the rest.  Every case is
held bit for bit to the oracle's Benjamini-Hochberg (itself pinned to myStats.benjamini_hochberg_correction, fithic/myStats.py:24-48)
and the library says which path ran (fhx_bh_sort_stats), so that each branch of the repair is known to have been exercised."""
import numpy as np
import pytest

from conftest import bits_equal

pytestmark = pytest.mark.gpu

LOW = 24                      # fhx_bh_array ranks all 64 key bits: five passes cover bits 24..63


@pytest.fixture(scope="module")
def ctx():
    from fithic_amd import _capi
    c = _capi.Context(0)
    yield c
    c.close()


def _check(ctx, p, N=1.0):
    from oracle import fithic_oracle as fo
    want = fo.benjamini_hochberg(p, N)
    got = ctx.bh_array(p, N)
    assert np.array_equal(np.isnan(got), np.isnan(want))
    assert bits_equal(np.nan_to_num(got, nan=-1.0), np.nan_to_num(want, nan=-1.0))
    return ctx.bh_sort_stats()


def _runs(rng, n, len_lo, len_hi, low_bits=LOW, distinct_low=None):
    """n p-values in runs that share their top (64 - low_bits) key bits and differ below: lengths uniform in [len_lo, len_hi]"""
    lens = []
    total = 0
    while total < n:
        L = int(rng.integers(len_lo, len_hi + 1))
        L = min(L, n - total)
        lens.append(L)
        total += L
    m = len(lens)
    # distinct tops inside (2^-200, 2^-4): exponent field 823..1018, random mantissa bits above `low_bits`
    expo = rng.integers(823, 1019, m).astype(np.uint64)
    mant_hi = rng.integers(0, 1 << (52 - low_bits), m).astype(np.uint64)
    tops = (expo << np.uint64(52)) | (mant_hi << np.uint64(low_bits))
    tops = np.unique(tops)
    if len(tops) < m:                                           # fewer distinct tops than runs: some runs share one (they merge)
        tops = np.concatenate([tops, rng.choice(tops, m - len(tops))])
    tops = rng.permutation(tops[:m])
    keys = np.repeat(tops, lens)
    if distinct_low is None:
        low = rng.integers(0, 1 << low_bits, n).astype(np.uint64)
    else:
        low = rng.choice(rng.integers(0, 1 << low_bits, distinct_low), n).astype(np.uint64)
    return rng.permutation(keys | low).view(np.float64)


@pytest.mark.parametrize("n", [131073, 4096 * 33, 4096 * 33 + 1, 1000003])
def test_short_runs_are_sorted_by_single_threads(ctx, n):
    st = _check(ctx, _runs(np.random.default_rng(n), n, 1, 32))
    assert st["passes"] == 5 and st["low_bit"] == LOW and st["beyond_lists"] == 0 and st["segments"] == 0
    assert st["inversions"] > n // 8 and st["runs_by_thread"] > n // 40 and st["long_run_inversions"] == 0


def _with_long_runs(rng, n, lengths, low_bits=LOW, distinct_low=None):
    """short runs (1..2 keys) with a few long ones of the given lengths somewhere in the order"""
    p = _runs(rng, n, 1, 2)
    where = np.cumsum([0] + [L + 7 for L in lengths[:-1]])          # (where they sit in the input does not matter: permuted below)
    assert where[-1] + lengths[-1] <= n
    for k, (w, L) in enumerate(zip(where, lengths)):
        top = (np.uint64(700 + k) << np.uint64(52)) | (np.uint64(k) << np.uint64(low_bits))
        if distinct_low is None:
            low = rng.integers(0, 1 << low_bits, L).astype(np.uint64)
        else:
            low = rng.choice(rng.integers(0, 1 << low_bits, distinct_low), L).astype(np.uint64)
        p[w:w + L] = (top | low).view(np.float64)
    return rng.permutation(p)


def test_long_runs_are_handed_to_the_host_as_segments(ctx):
    rng = np.random.default_rng(5)
    lengths = [33, 40, 64, 200, 900, 4096, 4097, 9000, 20_000, 140_000]       # the last one is beyond the LDS tile sorts: one-sweep passes
    st = _check(ctx, _with_long_runs(rng, 2_000_003, lengths))
    assert st["beyond_lists"] == 0 and st["segments"] == len(lengths) and st["segment_keys"] == sum(lengths)
    assert st["long_run_inversions"] >= len(lengths)


def test_long_runs_of_few_distinct_values(ctx):
    """runs whose keys take a handful of values below the sorted bits (the atoms of a table-driven run that meet by chance)"""
    rng = np.random.default_rng(6)
    lengths = [150, 900, 5000, 60_000]
    st = _check(ctx, _with_long_runs(rng, 900_001, lengths, distinct_low=3))
    assert st["beyond_lists"] == 0 and st["segments"] == len(lengths) and st["segment_keys"] == sum(lengths)


def test_more_long_runs_than_the_lists_hold_sorts_all_bits(ctx):
    n = 2_000_003
    st = _check(ctx, _runs(np.random.default_rng(13), n, 200, 3000))
    assert st["beyond_lists"] & 2 and st["beyond_lists"] & 8


def test_long_runs_that_are_most_of_the_keys_sort_all_bits(ctx):
    rng = np.random.default_rng(8)
    st = _check(ctx, _with_long_runs(rng, 600_000, [200_000, 150_000]))
    assert st["beyond_lists"] == 8 and st["segment_keys"] == 350_000


def test_the_run_of_zeros_and_the_smallest_subnormals(ctx):
    """what deep maps do: thousands of p-values underflow to 0 and a few stop just above it.  In the raw bit patterns they share
    their leading bits - one long run with inversions at the very start of the order (the bench's lognormal workload met it
    first); the passes sort an image of the key in which subnormals have exponents of their own (os_spread), so nothing is left
    to repair there: the zeros are one run of equal keys."""
    rng = np.random.default_rng(14)
    n = 800_000
    p = rng.random(n) ** 6
    idx = rng.choice(n, 60_000, replace=False)
    p[idx] = 0.0
    p[idx[:300]] = rng.integers(1, 1 << 20, 300).astype(np.uint64).view(np.float64)
    p[idx[300:600]] = rng.integers(1, 1 << 52, 300).astype(np.uint64).view(np.float64)
    p[idx[600:640]] = np.arange(1, 41).astype(np.uint64).view(np.float64)           # the 40 smallest subnormals
    st = _check(ctx, p)
    assert st["beyond_lists"] == 0 and st["segments"] == 0


def test_runs_of_equal_keys_need_no_repair_whatever_their_length(ctx):
    rng = np.random.default_rng(9)
    n = 1_500_000
    p = rng.choice(rng.random(300) ** 6, n)                       # 300 values, 5 000 copies each
    st = _check(ctx, p)
    assert st["beyond_lists"] == 0 and st["inversions"] == 0 and st["runs_by_thread"] == 0 and st["long_run_inversions"] == 0


def test_forced_fallback_and_forced_pass_counts_give_the_same_q(ctx, monkeypatch):
    rng = np.random.default_rng(10)
    n = 700_001
    p = rng.random(n) ** 5
    p[rng.integers(0, n, 11)] = np.nan
    monkeypatch.setenv("FHX_OS_FORCE_FALLBACK", "1")
    assert _check(ctx, p)["beyond_lists"] & 12 == 12
    monkeypatch.delenv("FHX_OS_FORCE_FALLBACK")
    for passes in (3, 4, 6, 8):
        monkeypatch.setenv("FHX_OS_PASSES", str(passes))
        st = _check(ctx, p)
        assert st["passes"] == passes and st["low_bit"] == 64 - 8 * passes
    monkeypatch.delenv("FHX_OS_PASSES")
    monkeypatch.setenv("FHX_K3_SORT", "legacy")
    _check(ctx, p)


def test_values_of_two_and_above_rank_above_every_p_in_large_arrays_too(ctx):
    """with few tests nothing saturates, every value is kept, and a caller's array may hold 2.0 or +inf (bit 62 of
    the pattern set) - the large sort must rank them last like the small one does (n > 131072 keys)."""
    rng = np.random.default_rng(11)
    n = 300_007
    p = rng.random(n)
    p[rng.integers(0, n, 40)] = rng.uniform(2.0, 1e300, 40)
    p[rng.integers(0, n, 5)] = np.inf
    p[rng.integers(0, n, 9)] = 1.0
    for N in (1.0, 25.0, 0.5 * n):
        _check(ctx, p, N)


def test_a_pass_whose_digit_is_constant_only_copies(ctx):
    rng = np.random.default_rng(12)
    n = 500_000
    p = (rng.random(n) + 1.0) * 2.0 ** -20                       # one exponent: the top pass sees a single digit
    st = _check(ctx, p)
    assert st["beyond_lists"] == 0


def _fuzz_seeds(default):
    import os
    spec = os.environ.get("FHX_FUZZ_SEEDS")
    if not spec:
        return list(default)
    lo, hi = (int(v) for v in spec.split(":"))
    return list(range(lo, hi))


@pytest.mark.parametrize("seed", _fuzz_seeds(range(7000, 7008)))
def test_large_sort_fuzz(ctx, seed):
    """random columns of 1.3e5 .. 2.5e6 values for the large sort: continuous tails, atoms with copies, runs that share their
    leading bits (short, long, longer than any list), zeros, subnormals, values of 1 and above, NaN; few or many tests (nothing or
    most of the column cut off).  q bit for bit the oracle's.  FHX_FUZZ_SEEDS=lo:hi widens the campaign."""
    rng = np.random.default_rng(seed)
    n = int(rng.integers(131_073, 2_500_000)) if rng.random() < 0.8 else int(rng.integers(131_073, 140_000))
    kind = rng.integers(0, 5)
    if kind == 0:
        p = rng.random(n) ** rng.integers(1, 12)
    elif kind == 1:
        p = rng.choice(rng.random(int(rng.integers(2, 5000))) ** 6, n)
    elif kind == 2:
        p = _runs(rng, n, 1, int(rng.integers(1, 40)), low_bits=int(rng.integers(20, 30)))
    elif kind == 3:
        lens = [int(v) for v in rng.integers(33, min(n // 4, 300_000), int(rng.integers(1, 6)))]
        while sum(lens) + 8 * len(lens) > n:
            lens.pop()
        p = _with_long_runs(rng, n, lens or [40], distinct_low=None if rng.random() < 0.5 else int(rng.integers(2, 6)))
    else:
        p = np.exp(rng.normal(-12.0, 6.0, n))
        p = np.minimum(p, 1.0)
    m = n // int(rng.integers(20, 2000))
    if rng.random() < 0.6:
        p[rng.integers(0, n, m)] = 0.0
    if rng.random() < 0.5:
        p[rng.integers(0, n, m // 4 + 1)] = rng.integers(1, 1 << int(rng.integers(2, 53)), m // 4 + 1).astype(np.uint64).view(np.float64)
    if rng.random() < 0.5:
        p[rng.integers(0, n, m)] = 1.0
    if rng.random() < 0.3:
        p[rng.integers(0, n, 5)] = rng.uniform(1.0, 1e9, 5)
    if rng.random() < 0.5:
        p[rng.integers(0, n, 9)] = np.nan
    N = float(rng.choice([1.0, 3.0, 0.3 * n, 1.0 * n, 7.5 * n, 1e12]))
    _check(ctx, p, N)


@pytest.mark.parametrize("n", [6000, 300_000])
def test_subnormal_quotients_are_rounded_once(ctx, n):
    """q = (p N) / rank with a SUBNORMAL quotient: the division the compiler expands rounds it twice (at 53 bits, then when it is
    scaled back down) and is one unit of 2^-1074 off in a case in some hundreds - seed 20174 of the fuzz above met p = 1.24e-314 at
    rank 1646.  The %e of such a q shows it in its 7th digit: deep maps put these rows in the file (subnormal p above the zeros)."""
    rng = np.random.default_rng(n)
    p = rng.random(n) ** 4
    k = n // 3
    p[rng.choice(n, k, replace=False)] = rng.integers(1, 1 << 44, k).astype(np.uint64).view(np.float64)
    p[rng.integers(0, n, n // 10)] = 0.0
    for N in (1.0, 3.0, 977.0, 0.37):
        _check(ctx, p, N)


def test_p_rewritten_through_the_device_pointer_is_counted_again():
    """K3 sizes its sort from the key histogram K2 kept while storing p.  A caller that takes fhx_device_ptr(ctx, 0) may write p
    itself: the library must then count the keys of what is THERE (the stale histogram would announce too few survivors and the
    ranking would lose keys - the device's own check, FHX_ERR_INTERNAL, is the backstop).  Here every row gets a small p."""
    import os
    from fithic_amd import tables
    from fithic_amd.engine import Engine
    from oracle import fithic_oracle as fo
    data = os.path.join(os.path.dirname(__file__), "golden", "data")
    chroms = tables.ChromIndex()
    con = tables.read_contacts(os.path.join(data, "quirk.contacts.gz"), chroms)
    eng = Engine(0)
    eng.configure(10000, 20000, 400000, 12, 1, "All")
    fc, fm, fh = tables.read_fragments(os.path.join(data, "quirk.frags.gz"), chroms)
    eng.load_fragments(fc, fm, fh, chroms.sort_rank())
    eng.load_contacts(con.chr1, con.mid1, con.chr2, con.mid2, con.count)
    eng.run_pass()
    n = len(con)
    p_new = np.random.default_rng(5).uniform(1e-12, 1e-6, n)          # all far below any cutoff: every row must be ranked
    eng.ctx.copy(eng.ctx.device_ptr(0), p_new.ctypes.data, 8 * n, 0)
    n_tests = 1.0e6
    eng.ctx.bh(n_tests)
    got = eng.ctx.fetch(n, p=True, q=True)
    assert bits_equal(got["p"], p_new)
    assert bits_equal(got["q"], fo.benjamini_hochberg(p_new, n_tests))
    eng.close()
