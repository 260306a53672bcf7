"""-r 0: generate_FragPairs' walk over every possible fragment pair (fithic/fithic.py:691-778) on the GPU (csrc/fhx_nfpairs.inc).
Slots [1] and [7] of every bin are integers in closed form per (fragment, bin); slot [3] is ONE sequential chain of rounded
additions per bin, evaluated as a scan of (parity -> increment) maps inside a binade and one floating-point addition per binade
crossing.  Held bit for bit to the same library's host walk (a host-only context: itself pinned to the oracle's literal pair walk
in tests/test_host_stages.py) on random irregular fragment sets - duplicate midpoints (distance 0: chains that start with zeros),
unmappable fragments, empty chromosomes, one bin and many, with and without distance bounds - and on chains of 10^7 terms."""
import numpy as np
import pytest

from conftest import bits_equal
from fithic_amd import _capi
from fithic_amd.engine import MODES

pytestmark = pytest.mark.gpu


def _fit(device, f_chr, f_mid, f_hit, rank, keys, sumcc, L, U, n_bins):
    ctx = _capi.Context(device)
    try:
        ctx.set_params(0, L, U, n_bins, 1, MODES["intraOnly"])
        ctx.load_fragments(np.array(f_chr, np.int32), np.array(f_mid, np.int32), np.array(f_hit, np.int32), rank)
        st = _capi.FhxStats()
        st.in_range_sum, st.inter_count = int(sumcc.sum()), 7
        ctx.set_dist_keys(keys)
        ctx.set_global_stats(st, sumcc, np.ones(len(keys), np.int64))
        try:
            info = ctx.fit()
        except _capi.FhxError as e:                                            # bin means the reference's spline stage exits on
            assert e.code == _capi.FHX_ERR_REFERENCE_EXIT, str(e)
            return None
        return (ctx.get_array(_capi.A_BIN_POSS), ctx.get_array(_capi.A_BIN_POSS7), ctx.get_array(_capi.A_BIN_SUMDIST),
                info.possible_intra_in_range, info.max_possible_dist, info.possible_inter_all, info.n_frags)
    finally:
        ctx.close()


def _case(rng, n_chr, n_max, span, n_bins):
    f_chr, f_mid, f_hit = [], [], []
    for c in range(n_chr):
        n = int(rng.integers(0, n_max))
        mids = np.sort(rng.integers(0, span, n))
        if n > 10:
            mids[rng.integers(0, n, 5)] = mids[rng.integers(0, n, 5)]          # duplicates: distance 0
        hits = (rng.random(n) > 0.1).astype(np.int64)
        for m in rng.permutation(n):                                           # file order is not sorted
            f_chr.append(c)
            f_mid.append(int(mids[m]))
            f_hit.append(int(hits[m]))
    L = int(rng.choice([0, 0, 20000, 150000]))
    U = float(rng.choice([np.inf, span // 8, span // 2]))
    keys = np.unique(rng.integers(max(L, 1), int(min(U, span * 0.8)), 300)).astype(np.int64)
    sumcc = rng.integers(1, 50, len(keys)).astype(np.int64)
    rank = np.argsort(np.argsort(["c%d" % c for c in range(n_chr)])).astype(np.int32)
    return f_chr, f_mid, f_hit, rank, keys, sumcc, L, U, n_bins


def _compare(args):
    want = _fit(-1, *args)
    got = _fit(0, *args)
    assert (want is None) == (got is None)
    if want is None:
        return 0
    assert np.array_equal(got[0], want[0]) and np.array_equal(got[1], want[1])
    assert bits_equal(got[2], want[2]), (got[2], want[2])
    assert got[3:] == want[3:]
    return int(want[0].sum())


@pytest.mark.parametrize("seed", range(24))
def test_possible_pair_sums_equal_the_host_walk(seed):
    rng = np.random.default_rng(500 + seed)
    pairs = _compare(_case(rng, int(rng.integers(1, 5)), 400, 3_000_000, int(rng.choice([1, 3, 12, 40]))))
    assert pairs >= 0


@pytest.mark.parametrize("seed,n_bins", [(1, 4), (2, 20), (3, 40), (4, 100)])
def test_long_chains_cross_many_binades(seed, n_bins):
    """thousands of fragments per chromosome, contact counts that fall with the distance (so that the fit behind the walk goes
    through): chains of up to 10^7 terms - hundreds of windows of 16 384 terms, some twenty binade crossings each, ties among them"""
    rng = np.random.default_rng(900 + seed)
    f_chr, f_mid, f_hit = [], [], []
    for c in range(3):
        n = int(rng.integers(2500, 4500))
        mids = np.sort(rng.integers(0, 40_000_000, n))
        f_chr += [c] * n
        f_mid += [int(v) for v in mids]
        f_hit += [1] * n
    L, U = 10000, float(rng.choice([np.inf, 20_000_000]))
    keys = np.unique(rng.integers(L, 30_000_000, 20000)).astype(np.int64)
    sumcc = np.maximum(1, (3e6 * (keys / 1e4) ** -1.08 * rng.random(len(keys))).astype(np.int64))
    rank = np.arange(3, dtype=np.int32)
    pairs = _compare((f_chr, f_mid, f_hit, rank, keys, sumcc, L, U, n_bins))
    assert pairs > 3_000_000


def test_host_walk_on_request(monkeypatch):
    """FHX_NF_HOST_PAIRS=1: the chains on the host threads, as in a host-only context (measurements, and what remains when the
    arrays do not fit)"""
    monkeypatch.setenv("FHX_NF_HOST_PAIRS", "1")
    rng = np.random.default_rng(77)
    assert _compare(_case(rng, 2, 300, 3_000_000, 12)) >= 0
