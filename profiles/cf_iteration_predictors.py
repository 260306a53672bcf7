# cf_iteration_predictors.py (CPU, oracle): how well sort keys homogenise the iteration counts of the converging continued-fraction
# classes inside waves of 64 (tiles of 1024, as k2_queue_by_count sorts them).  python profiles/cf_iteration_predictors.py
import sys, numpy as np
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import fithic_oracle as fo
rng = np.random.default_rng(1)
N = 400000
n_total = 6.5e8
# expected counts: distance decay (lam from 40 down to 0.2) x lognormal bias product
lam = np.exp(rng.uniform(np.log(0.2), np.log(40), N)) * np.exp(rng.normal(0, 0.25, N))
cnt = rng.poisson(lam)
keep = cnt >= 2
lam, cnt = lam[keep], cnt[keep]
prior = lam / n_total
p, br, it = fo.bdtrc_stats(cnt.astype(float) - 1, n_total, prior)   # bdtrc(k-1, n, p)
print("branches", np.unique(br, return_counts=True))
def wave_stats(order, it, name):
    x = it[order]
    m = len(x) // 64 * 64
    g = x[:m].reshape(-1, 64)
    print("%-34s mean it %.2f  mean of per-wave max %.2f" % (name, x.mean(), g.max(1).mean()))
for b in np.unique(br):
    sel = np.flatnonzero(br == b)
    if len(sel) < 5000: continue
    c, l, i = cnt[sel], lam[sel], it[sel]
    print("branch", b, "rows", len(sel))
    # tiles of 1024 in arrival (random) order, sorted within the tile by different keys
    def tiled(keyfun, name):
        tot_max = 0; nw = 0
        for s in range(0, len(sel) - 1023, 1024):
            k = keyfun(c[s:s+1024], l[s:s+1024])
            o = np.argsort(k, kind="stable")
            g = i[s:s+1024][o].reshape(16, 64)
            tot_max += g.max(1).sum(); nw += 16
        print("   %-40s per-wave max %.2f (mean it %.2f)" % (name, tot_max / nw, i.mean()))
    tiled(lambda c, l: np.zeros(len(c)), "queue order")
    tiled(lambda c, l: np.minimum(c, 31), "count (min(c,31))  [current]")
    tiled(lambda c, l: l / c, "ratio lam/count")
    tiled(lambda c, l: np.minimum(c, 31) * 16 + np.minimum((l / c * 8).astype(int), 15), "count, then ratio in 16ths")
    tiled(lambda c, l: np.minimum((l / c * 8).astype(int), 15) * 32 + np.minimum(c, 31), "ratio in 8ths, then count")
    tiled(lambda c, l: i[:0].sum() + 0 * c + 0, "dummy")
    # oracle bound: sort by the true iteration count
    tot_max = 0; nw = 0
    for s in range(0, len(sel) - 1023, 1024):
        g = np.sort(i[s:s+1024]).reshape(16, 64); tot_max += g.max(1).sum(); nw += 16
    print("   %-40s per-wave max %.2f" % ("ideal (sorted by true iterations)", tot_max / nw))
