"""HiCKRy on the MI355X engine: the reference's Knight-Ruiz bias-file generator (fithic/utils/HiCKRy.py) with the same
command line, function names, return shapes and printed messages; the matrix lives in HBM as CSR and every O(nnz)
step (assembly, row removal, the sparse mat-vec of each Newton/CG step) runs in the HIP kernels of csrc/fhx_kr.hip.

    python -m fithic_amd.hickry -i contacts.gz -f fragments.gz -o bias.gz [-x 0.05]

There is no CPU implementation here: without the built library or without a GPU the calls raise.
"""
import argparse
import gzip
import sys
import time

import numpy as np

from . import _capi, tables

device = 0


def parse_args(arguments):
    """HiCKRy.py:10-16 (like the reference, the argument is ignored and sys.argv is parsed)."""
    parser = argparse.ArgumentParser(description="Check help flag")
    parser.add_argument("-i", "--interactions", help="Path to the interactions file to generate bias values", required=True, type=str)
    parser.add_argument("-f", "--fragments", help="Path to the interactions file to generate bias values", required=True, type=str)
    parser.add_argument("-o", "--output", help="Full path to output the generated bias file to", required=True, type=str)
    parser.add_argument("-x", "--percentOfSparseToRemove", help="Percent of diagonal to remove", required=False, type=float, default=0.05)
    return parser.parse_args()


class DeviceMatrix:
    """The symmetric raw contact matrix (or its reduced form) resident on the GPU; stands in for the reference's csr_matrix."""

    def __init__(self, kr, reduced=False):
        self.kr, self.reduced = kr, reduced

    @property
    def shape(self):
        n_full, _, n_red, _ = self.kr.shape()
        n = n_red if self.reduced else n_full
        return (n, n)

    @property
    def nnz(self):
        _, nnz_full, _, nnz_red = self.kr.shape()
        return nnz_red if self.reduced else nnz_full

    def dot(self, x):
        x = np.asarray(x, np.float64)
        y, _ = self.kr.spmv(x.reshape(-1), which=1 if self.reduced else 0)
        return y.reshape(x.shape)

    def sum(self, axis=None):
        sums = self.kr.row_sums()
        return float(np.sum(sums)) if axis is None else sums.reshape(1, -1)


def _read_python(path, fields):
    """The reference's own parse (`lines.rstrip().split()`), used when the native reader rejects the layout."""
    rows = []
    with gzip.open(path, "rt") as f:
        for line in f:
            w = line.rstrip().split()
            rows.append([conv(w[k]) for k, conv in fields])
    return rows


def loadfastfithicInteractions(interactionsFile, fragsFile):
    """HiCKRy.py:18-54.  Returns (rawMatrix, revFrag)."""
    print("Creating sparse matrix...")
    startT = time.time()
    chroms = tables.ChromIndex()
    try:
        f_chr, f_mid, _ = tables.read_fragments(fragsFile, chroms)
    except _capi.FhxError:
        rows = _read_python(fragsFile, [(0, str), (2, int)])
        f_chr = np.array([chroms.intern(r[0]) for r in rows], np.int32)
        f_mid = np.array([r[1] for r in rows], np.int64)
    try:
        con = tables.read_contacts(interactionsFile, chroms, want_raw=True)
        c1, m1, c2, m2, z = con.chr1, con.mid1, con.chr2, con.mid2, con.raw_count
    except _capi.FhxError:
        rows = _read_python(interactionsFile, [(0, str), (1, int), (2, str), (3, int), (4, float)])
        c1 = np.array([chroms.intern(r[0]) for r in rows], np.int32)
        c2 = np.array([chroms.intern(r[2]) for r in rows], np.int32)
        m1 = np.array([r[1] for r in rows], np.int64)
        m2 = np.array([r[3] for r in rows], np.int64)
        z = np.array([r[4] for r in rows], np.float64)
    kr = _capi.KrContext(device)
    kr.load_loci(f_chr, f_mid)
    kr.load_pairs(c1, m1, c2, m2, z)                      # KeyError(row) for a locus missing from the fragments file
    names = chroms.names
    revFrag = [(names[c], int(m)) for c, m in zip(f_chr.tolist(), f_mid.tolist())]
    endT = time.time()
    print("Sparse matrix creation took %s seconds" % (endT - startT))
    return DeviceMatrix(kr), revFrag


def returnBias(rawMatrix, perc):
    """HiCKRy.py:56-72."""
    mtxAndRemoved = removeZeroDiagonalCSR(rawMatrix, perc)
    print("Sparse rows removed")
    initialSize = rawMatrix.shape
    print("Initial matrix size: %s rows and %s columns" % (initialSize[0], initialSize[1]))
    rawMatrix = mtxAndRemoved[0]
    newSize = rawMatrix.shape
    print("New matrix size: %s rows and %s columns" % (newSize[0], newSize[1]))
    print("Normalizing with KR Algorithm")
    knightRuizAlg(rawMatrix)
    return rawMatrix.kr.bias().reshape(-1, 1)            # computeBiasVector + addZeroBiases, on the engine's x


def removeZeroDiagonalCSR(mtx, perc):
    """HiCKRy.py:74-101.  Returns [reduced matrix, removed row indices]."""
    n = mtx.shape[0]
    print("Removing %s percent of most sparse bins" % (perc))
    print("... corresponds to %s total rows" % (int(perc * n)))
    removed, val, _ = mtx.kr.remove_sparse(perc)          # IndexError for perc >= 1 like the reference's list index
    print("... corresponds to all bins with less than or equal to %s total interactions" % val)
    return [DeviceMatrix(mtx.kr, reduced=True), removed.tolist()]


def knightRuizAlg(A, tol=1e-6, f1=False):
    """HiCKRy.py:139-243.  Returns [x (n,1), outer iterations, inner iterations of the last outer step]."""
    x, info = A.kr.balance(tol)
    A.info = info.as_dict()
    return [x.reshape(-1, 1), info.outer_iterations, info.inner_iterations]


def computeBiasVector(x):
    """HiCKRy.py:103-109 (n-sized host arithmetic, numpy like the reference)."""
    one = np.ones((x.shape[0], 1))
    x = one / x
    sums = np.sum(x)
    avg = (1.0 * sums) / x.shape[0]
    return np.divide(x, avg)


def addZeroBiases(lst, vctr):
    """HiCKRy.py:111-114: -1 at every removed index."""
    lst = np.asarray(lst, np.int64)
    n = vctr.shape[0] + len(lst)
    out = np.full((n, 1), -1.0)
    keep = np.ones(n, bool)
    keep[lst] = False
    out[keep, 0] = np.asarray(vctr, np.float64).reshape(-1)
    return out


def checkBias(biasvec):
    """HiCKRy.py:245-263."""
    std = np.std(biasvec)
    mean = np.mean(biasvec)
    median = np.median(biasvec)
    if (mean < 0.5 or mean > 2) or (median < 0.5 or median > 2):
        which = "mean" if (mean < 0.5 or mean > 2) else "median"
        print("WARNING... Bias vector has a %s outside of typical range (0.5, 2)." % which)
        print("Consider running with a larger -x option if problems occur")
        print("Mean\t%s" % mean)
        print("Median\t%s" % median)
        print("Std. Dev.\t%s" % std)
    return


def outputBias(biasCol, revFrag, outputFilePath):
    """HiCKRy.py:265-274: chr <tab> mid <tab> bias (shortest round-trip text of the double, as numpy prints it)."""
    vals = np.asarray(biasCol, np.float64).reshape(-1)
    with gzip.open(outputFilePath, "wt") as biasFile:
        for (chrom, mid), v in zip(revFrag, vals.tolist()):
            biasFile.write("%s\t%s\t%s\n" % (chrom, mid, repr(v)))


def main():
    args = parse_args(sys.argv[3:])
    matrix, revFrag = loadfastfithicInteractions(args.interactions, args.fragments)
    bias = returnBias(matrix, args.percentOfSparseToRemove)
    checkBias(bias)
    outputBias(bias, revFrag, args.output)
    matrix.kr.close()


if __name__ == "__main__":
    main()
