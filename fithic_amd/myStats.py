"""`from fithic_amd import myStats`: the one name of fithic/myStats.py on the hot path.

    benjamini_hochberg_correction(p_values, num_total_tests)    fithic/myStats.py:24-48   -> K3 on the GPU (no CPU path)

    meanAndVariance(a)                                          fithic/myStats.py:53-63   (no caller in the reference - SURVEY.md
                                                                section 2, #16; kept for scripts that import it)
"""
from .fithic import benjamini_hochberg_correction  # noqa: F401  (sort + scan kernels behind the C ABI)


def meanAndVariance(a):
    """(mean, E[x^2] - mean^2) with the reference's left-to-right sums, so floats round the same way."""
    n = float(len(a))
    total = squares = 0
    for x in a:
        squares += x * x
        total += x
    mean = total / n
    return (mean, squares / n - mean * mean)
