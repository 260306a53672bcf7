"""The names of fithic/myStats.py, for code that imports them from the package (`from fithic_amd import myStats`).

    benjamini_hochberg_correction(p_values, num_total_tests)    fithic/myStats.py:24-48   -> K3 on the GPU (no CPU path)
    meanAndVariance(a)                                          fithic/myStats.py:53-63   -> E(x^2) - (Ex)^2 on the host
"""
from .fithic import benjamini_hochberg_correction  # noqa: F401  (sort + scan kernels behind the C ABI)


def meanAndVariance(a):
    """(mean, variance) with the variance as E(x^2) - (E x)^2, sums accumulated left to right like the reference's loop."""
    sum_sq = 0
    total = 0
    for x in a:
        sum_sq += x * x
        total += x
    mean = total / float(len(a))
    return (mean, sum_sq / float(len(a)) - mean * mean)
