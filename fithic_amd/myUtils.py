"""The names of fithic/myUtils.py that the hot path uses, for code that imports them from the package
(`from fithic import myUtils` -> `from fithic_amd import myUtils`).

On the engine's path these are not called per row - K1 / K2 classify every contact row on the GPU
(fithic_amd/csrc/fhx_k1.hip: k1_classify_hist; fhx_k2.hip: row_prior) with the same predicates; the Python forms exist for callers that
used them on their own rows.  Thresholds follow the reference's convention: -1 = no bound on that side.

    in_range_check(interactionDistance, distLowThres, distUpThres)      fithic/myUtils.py:85-92
    scale_a_list(somelist, s)                                           fithic/myUtils.py:14-15

(The reference's per-line Interaction objects - fithic/myUtils.py:98-147, 5.5 s of its 67 s profile - have no counterpart: a contact
row is three int32 in HBM and its type is decided where it is used)
"""


def _bounded_below(distance, low):
    return low == -1 or (low > -1 and distance >= low)


def _bounded_above(distance, up):
    return up == -1 or (up > -1 and distance <= up)


def in_range_check(interactionDistance, distLowThres, distUpThres):
    """True when low <= distance <= up, a bound of -1 being absent (both ends inclusive since 2.0.x)."""
    return bool(_bounded_below(interactionDistance, distLowThres) and _bounded_above(interactionDistance, distUpThres))


def scale_a_list(somelist, s):
    return [1.0 * v * s for v in somelist]
