"""The names of fithic/myUtils.py that the hot path uses, for code that imports them from the package
(`from fithic import myUtils` -> `from fithic_amd import myUtils`).

On the engine's path these are not called per row - K1 / K2 classify every contact row on the GPU
(fithic_amd/csrc/fhx_k1.hip: k1_classify_hist; fhx_k2.hip: row_prior) with the same predicates; the Python forms exist for callers that
used them on their own rows.  Thresholds follow the reference's convention: -1 = no bound on that side.

    in_range_check(interactionDistance, distLowThres, distUpThres)      fithic/myUtils.py:85-92
    scale_a_list(somelist, s)                                           fithic/myUtils.py:14-15

    Interaction(locusPair)                                              fithic/myUtils.py:98-147

(The reference builds one Interaction object per input line - 5.5 s of its 67 s profile.  The engine never does: a contact row is three
int32 in HBM and its type is decided where it is used, SURVEY.md section 8 row a1.  The class below is for user scripts that made
such objects themselves: same attributes, same setters / getters, same strings from getType; it is a record over `pair_type`.)
"""

_NO_DISTANCE = -1


def pair_type(same_chromosome, distance, distLowThres, distUpThres):
    """'inter' | 'intraInRange' | 'intraShort' | 'intraLong' | None (no rule applies: the reference leaves the old type in place)."""
    if not same_chromosome:
        return "inter"
    if in_range_check(distance, distLowThres, distUpThres):
        return "intraInRange"
    if distLowThres > -1 and distance <= distLowThres:
        return "intraShort"
    if distUpThres > -1 and distance > distUpThres:
        return "intraLong"
    return None


class Interaction(object):
    """One contact between two loci: (chr1, mid1, chr2, mid2) from a 4-sequence, then count / p / q through the setters."""
    hitCount, pval, qval, dictkey = 0, -1.0, -1.0, "null"

    def __init__(self, locusPair):
        self.chr1, m1, self.chr2, m2 = locusPair[:4]
        self.mid1, self.mid2 = int(m1), int(m2)
        cis = self.chr1 == self.chr2
        self.type = "intra" if cis else "inter"
        self.distance = abs(self.mid1 - self.mid2) if cis else _NO_DISTANCE

    def getType(self, distLowThres, distUpThres):
        found = pair_type(self.type != "inter", self.distance, distLowThres, distUpThres)
        if found is not None:
            self.type = found
        return self.type

    def getDistance(self):
        return self.distance

    def getCount(self):
        return self.hitCount

    def setCount(self, x):
        self.hitCount = int(x)

    def setType(self, x):
        self.type = str(x)

    def setPval(self, x):
        self.pval = float(x)

    def setQval(self, x):
        self.qval = float(x)


def _bounded_below(distance, low):
    return low == -1 or (low > -1 and distance >= low)


def _bounded_above(distance, up):
    return up == -1 or (up > -1 and distance <= up)


def in_range_check(interactionDistance, distLowThres, distUpThres):
    """True when low <= distance <= up, a bound of -1 being absent (both ends inclusive since 2.0.x)."""
    return bool(_bounded_below(interactionDistance, distLowThres) and _bounded_above(interactionDistance, distUpThres))


def scale_a_list(somelist, s):
    return [1.0 * v * s for v in somelist]
