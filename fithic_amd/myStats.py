"""`from fithic_amd import myStats`: the one name of fithic/myStats.py on the hot path.

    benjamini_hochberg_correction(p_values, num_total_tests)    fithic/myStats.py:24-48   -> K3 on the GPU (no CPU path)

(meanAndVariance, fithic/myStats.py:53-63, has no caller in the reference - SURVEY.md section 2, #16 - and is not carried.)
"""
from .fithic import benjamini_hochberg_correction  # noqa: F401  (sort + scan kernels behind the C ABI)
