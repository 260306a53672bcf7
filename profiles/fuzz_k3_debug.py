"""What a failing seed of tests/test_gpu_k3_sort.py::test_large_sort_fuzz looks like: the case rebuilt, run through the default sort,
eight one-sweep passes, round 4's passes and the forced full sort, with the rows that differ from the oracle and the keys around them.
    python profiles/fuzz_k3_debug.py <seed>        (seed 20174 found the twice-rounded subnormal BH quotient)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import test_gpu_k3_sort as T
from fithic_amd import _capi
from oracle import fithic_oracle as fo
seed = int(sys.argv[1])
# replicate the generator
rng = np.random.default_rng(seed)
n = int(rng.integers(131_073, 2_500_000)) if rng.random() < 0.8 else int(rng.integers(131_073, 140_000))
kind = rng.integers(0, 5)
print("n", n, "kind", kind)
if kind == 0:
    p = rng.random(n) ** rng.integers(1, 12)
elif kind == 1:
    p = rng.choice(rng.random(int(rng.integers(2, 5000))) ** 6, n)
elif kind == 2:
    p = T._runs(rng, n, 1, int(rng.integers(1, 40)), low_bits=int(rng.integers(20, 30)))
elif kind == 3:
    lens = [int(v) for v in rng.integers(33, min(n // 4, 300_000), int(rng.integers(1, 6)))]
    while sum(lens) + 8 * len(lens) > n:
        lens.pop()
    print("lens", lens)
    p = T._with_long_runs(rng, n, lens or [40], distinct_low=None if rng.random() < 0.5 else int(rng.integers(2, 6)))
else:
    p = np.exp(rng.normal(-12.0, 6.0, n)); p = np.minimum(p, 1.0)
m = n // int(rng.integers(20, 2000))
if rng.random() < 0.6: p[rng.integers(0, n, m)] = 0.0
if rng.random() < 0.5: p[rng.integers(0, n, m // 4 + 1)] = rng.integers(1, 1 << int(rng.integers(2, 53)), m // 4 + 1).astype(np.uint64).view(np.float64)
if rng.random() < 0.5: p[rng.integers(0, n, m)] = 1.0
if rng.random() < 0.3: p[rng.integers(0, n, 5)] = rng.uniform(1.0, 1e9, 5)
if rng.random() < 0.5: p[rng.integers(0, n, 9)] = np.nan
N = float(rng.choice([1.0, 3.0, 0.3 * n, 1.0 * n, 7.5 * n, 1e12]))
print("N", N, "zeros", int((p == 0).sum()), "subnormal", int(((p > 0) & (p < 2.3e-308)).sum()), "ones", int((p == 1).sum()), ">1", int((p > 1).sum()), "nan", int(np.isnan(p).sum()))
want = fo.benjamini_hochberg(p, N)
ctx = _capi.Context(0)
for env in ({}, {"FHX_OS_PASSES": "8"}, {"FHX_K3_SORT": "legacy"}, {"FHX_OS_FORCE_FALLBACK": "1"}):
    for k in ("FHX_OS_PASSES", "FHX_K3_SORT", "FHX_OS_FORCE_FALLBACK"): os.environ.pop(k, None)
    os.environ.update(env)
    got = ctx.bh_array(p, N)
    st = ctx.bh_sort_stats()
    a = np.nan_to_num(got, nan=-1.0).view(np.int64); b = np.nan_to_num(want, nan=-1.0).view(np.int64)
    bad = np.flatnonzero(a != b)
    print(env, "mismatches", len(bad), st)
    if len(bad):
        i = bad[:5]
        print("  rows", i, "p", p[i], "got", got[i], "want", want[i])
        order = np.argsort(p, kind="stable")
        rank_of = np.empty(n, np.int64); rank_of[order] = np.arange(n)
        print("  ranks of bad rows: min", rank_of[bad].min(), "max", rank_of[bad].max(), "count", len(bad))
        r0 = rank_of[bad].min()
        print("  sorted p around first bad rank:", p[order[max(0, r0 - 3):r0 + 4]].view(np.uint64))
