"""The names of fithic/myUtils.py that the hot path uses, for code that imports them from the package
(`from fithic import myUtils` -> `from fithic_amd import myUtils`).

On the engine's path these are not called per row - K1 / K2 classify every contact row on the GPU
(fithic_amd/csrc/fhx_k1.hip: k1_classify_hist; fhx_k2.hip: row_prior) with the same predicates; the Python forms exist for callers that
used them on their own rows.  Thresholds follow the reference's convention: -1 = no bound on that side.

    in_range_check(interactionDistance, distLowThres, distUpThres)      fithic/myUtils.py:85-92
    Interaction([chr1, mid1, chr2, mid2]) .getType(low, up) ...         fithic/myUtils.py:98-147
    scale_a_list(somelist, s)                                           fithic/myUtils.py:14-15
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


class Interaction:
    """A contact between two loci: `type` is 'inter' or 'intra' at construction, refined by getType()."""
    hitCount = 0
    distance = -1
    pval = -1.0
    qval = -1.0
    dictkey = 'null'

    def __init__(self, locusPair):
        self.chr1, self.chr2 = locusPair[0], locusPair[2]
        self.mid1, self.mid2 = int(locusPair[1]), int(locusPair[3])
        if self.chr1 == self.chr2:
            self.type = 'intra'
            self.distance = abs(self.mid1 - self.mid2)
        else:
            self.type = 'inter'

    def setCount(self, x):
        self.hitCount = int(x)

    def setType(self, x):
        self.type = str(x)

    def setPval(self, x):
        self.pval = float(x)

    def setQval(self, x):
        self.qval = float(x)

    def getDistance(self):
        return self.distance

    def getCount(self):
        return self.hitCount

    def getType(self, distLowThres, distUpThres):
        """'inter', 'intraInRange', 'intraShort' (at or below a given lower bound) or 'intraLong' (above a given upper bound);
        the type is left as it was when neither applies (the reference's elif chain)."""
        if self.type == 'inter':
            return self.type
        if in_range_check(self.distance, distLowThres, distUpThres):
            self.type = 'intraInRange'
        elif distLowThres > -1 and self.distance <= distLowThres:
            self.type = 'intraShort'
        elif distUpThres > -1 and self.distance > distUpThres:
            self.type = 'intraLong'
        return self.type
