"""Why the 300-iteration class cannot be replaced by a cheaper, more accurate evaluation: in its regime (n ~ 1.5e8, contact counts
2..40, n * prior of the order of the count) scipy.special.bdtrc - Cephes' swapped continued fraction stopped at its iteration cap -
is itself several 1e-9 away from the exact binomial tail, so an exact evaluation would MISS the north_star's bar (|p - bdtrc| <= 1e-10)
that the bit-faithful loop meets.  CPU only (scipy + decimal):  python profiles/cephes_vs_exact.py > profiles/history/r03_cephes_vs_exact.txt"""
from decimal import Decimal as D, getcontext

import numpy as np
import scipy
import scipy.special as sp

getcontext().prec = 60
rng = np.random.default_rng(1)
n = 147964314                                     # rows of C3-synth = N of its binomial
worst, over = 0.0, 0
cases = 2000
print("scipy", scipy.__version__, " n =", n, " cases =", cases)
for t in range(cases):
    k = int(rng.integers(2, 40))
    lam = rng.uniform(0.06, 3.0) * k
    p = lam / n
    cephes = float(sp.bdtrc(k - 1, n, p))
    P = D(p)
    q = D(1) - P
    term = q ** n
    s = term
    for j in range(1, k):
        term = term * D(n - j + 1) / D(j) * P / q
        s += term
    exact = float(D(1) - s)
    d = abs(cephes - exact)
    over += d > 1e-10
    if d > worst:
        worst = d
        print("count %2d  n*p %9.4f  bdtrc %.17g  exact tail %.17g  |diff| %.3g" % (k, lam, cephes, exact, d))
print("largest |bdtrc - exact| = %.3g; %d of %d cases beyond 1e-10" % (worst, over, cases))
