"""TEST INFRASTRUCTURE - CPU oracle of the Knight-Ruiz bias path (fithic/utils/HiCKRy.py), numpy + oracle/kr_oracle.c.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this; the product (fithic_amd.hickry)
never does.  Each function cites the reference lines it restates.  Summation orders are the engine's (see kr_oracle.c):
bit-equal to the GPU, and pinned against the real reference by tests/golden/k*_kr_*.npz within 1e-12 relative (measured <= 7e-15)
(the reference's own last bits depend on its BLAS build, so there is no bit-exact target to hit).
"""
import ctypes
import gzip
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build(force=False):
    so = os.path.join(_HERE, "liboracle_kr.so")
    src = os.path.join(_HERE, "kr_oracle.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["gcc", "-O2", "-ffp-contract=off", "-fno-fast-math", "-fPIC", "-shared", "-o", so, src])
    return so


def _lib():
    global _LIB
    if _LIB is None:
        L = ctypes.CDLL(build())
        p = ctypes.c_void_p
        L.fho_kr_segsum.restype = ctypes.c_int64
        L.fho_kr_segsum.argtypes = [ctypes.c_int64, p, p, p, p]
        L.fho_kr_spmv.restype = None
        L.fho_kr_spmv.argtypes = [ctypes.c_int64, p, p, p, p, p]
        L.fho_kr_dot.restype = ctypes.c_double
        L.fho_kr_dot.argtypes = [ctypes.c_int64, p, p]
        _LIB = L
    return _LIB


def _ptr(a):
    return a.ctypes.data_as(ctypes.c_void_p)


class Csr:
    """Canonical CSR (sorted columns, duplicates summed) of the symmetric raw matrix."""

    def __init__(self, n, indptr, indices, data):
        self.n, self.indptr, self.indices, self.data = int(n), indptr, indices, data

    @property
    def nnz(self):
        return int(self.indptr[-1])

    def dot(self, x):
        x = np.ascontiguousarray(x, np.float64)
        y = np.empty(self.n, np.float64)
        _lib().fho_kr_spmv(self.n, _ptr(self.indptr), _ptr(self.indices), _ptr(self.data), _ptr(x), _ptr(y))
        return y


def dot(a, b=None):
    a = np.ascontiguousarray(a, np.float64)
    if b is not None:
        b = np.ascontiguousarray(b, np.float64)
    return float(_lib().fho_kr_dot(len(a), _ptr(a), _ptr(b) if b is not None else None))


def read_tables(interactionsFile, fragsFile):
    """HiCKRy.py:18-46: locus index = line number of the fragments file (a repeated (chr, mid) keeps its LAST line number,
    revFrag keeps every line); rows with an unknown locus raise KeyError."""
    fragDic, revFrag = {}, []
    with gzip.open(fragsFile, "rt") as f:
        for ctr, line in enumerate(f):
            w = line.rstrip().split()
            fragDic[(w[0], int(w[2]))] = ctr
            revFrag.append((w[0], int(w[2])))
    x, y, z = [], [], []
    with gzip.open(interactionsFile, "rt") as f:
        for line in f:
            w = line.rstrip().split()
            z.append(float(w[4]))
            x.append(fragDic[(w[0], int(w[1]))])
            y.append(fragDic[(w[2], int(w[3]))])
    return np.array(x, np.int64), np.array(y, np.int64), np.array(z, np.float64), len(revFrag), revFrag


def assemble(x, y, z, n):
    """HiCKRy.py:47-52: coo_matrix((z,(x,y))) + its transpose, as canonical CSR.  Entries of one cell are added one by one,
    the file's (x,y) rows first, then the transposed rows, each in file order (exact for integer counts)."""
    keys = np.concatenate([x * n + y, y * n + x]).astype(np.int64)
    vals = np.concatenate([z, z]).astype(np.float64)
    order = np.argsort(keys, kind="stable")
    ks, vs = np.ascontiguousarray(keys[order]), np.ascontiguousarray(vals[order])
    ok, ov = np.empty_like(ks), np.empty_like(vs)
    m = _lib().fho_kr_segsum(len(ks), _ptr(ks), _ptr(vs), _ptr(ok), _ptr(ov))
    ok, ov = ok[:m], ov[:m]
    rows = ok // n
    indptr = np.zeros(n + 1, np.int64)
    np.add.at(indptr, rows + 1, 1)
    return Csr(n, np.cumsum(indptr), np.ascontiguousarray((ok % n).astype(np.int32)), np.ascontiguousarray(ov))


def sparse_rows(A, perc):
    """HiCKRy.py:74-93: indices whose row sum is <= the rem-th smallest sum, rem = int(perc * n); ascending."""
    sums = A.dot(np.ones(A.n))
    rem = int(perc * A.n)
    val = np.sort(sums, kind="stable")[rem]            # IndexError for perc >= 1, like the reference's list index
    return np.flatnonzero(sums <= val).astype(np.int64), float(val), sums


def drop(A, removed):
    """HiCKRy.py:94-101,117-137: drop those rows and columns, renumber."""
    keep = np.ones(A.n, bool)
    keep[removed] = False
    new = np.cumsum(keep) - 1
    rows = np.repeat(np.arange(A.n), np.diff(A.indptr))
    sel = keep[rows] & keep[A.indices]
    n2 = int(keep.sum())
    indptr = np.zeros(n2 + 1, np.int64)
    np.add.at(indptr, new[rows[sel]] + 1, 1)
    return Csr(n2, np.cumsum(indptr), np.ascontiguousarray(new[A.indices[sel]].astype(np.int32)), np.ascontiguousarray(A.data[sel]))


def knight_ruiz(A, tol=1e-6):
    """HiCKRy.py:139-243, statement for statement (vectors are 1-D here; the reference's are (n,1))."""
    n = A.n
    e = np.ones(n)
    Delta, delta, g = 3, 0.1, 0.9
    etamax = eta = 0.1
    stop_tol = tol * 0.5
    x = e.copy()
    rt = tol ** 2.0
    v = x * A.dot(x)
    rk = 1.0 - v
    rho_km1 = dot(rk, rk)
    rho_km2 = rho_km1
    rout = rold = rho_km1
    i = k = 0
    while rout > rt:
        i += 1
        if i > 30:
            break
        k = 0
        y = e.copy()
        innertol = max(eta ** 2.0 * rout, rt)
        while rho_km1 > innertol:
            k += 1
            if k == 1:
                Z = rk / v
                p = Z.copy()
                rho_km1 = dot(rk, Z)
            else:
                beta = rho_km1 / rho_km2
                p = Z + beta * p
            if k > 10:
                break
            w = x * A.dot(x * p) + v * p
            alpha = rho_km1 / dot(p, w)
            ap = alpha * p
            ynew = y + ap
            if np.amin(ynew) <= delta:
                if delta == 0:
                    break
                ind = np.where(ap < 0.0)[0]
                gamma = np.amin((delta - y[ind]) / ap[ind])
                y += gamma * ap
                break
            if np.amax(ynew) >= Delta:
                ind = np.where(ynew > Delta)[0]
                gamma = np.amin((Delta - y[ind]) / ap[ind])
                y += gamma * ap
                break
            y = ynew.copy()
            rk = rk - alpha * w
            rho_km2 = rho_km1
            Z = rk / v
            rho_km1 = dot(rk, Z)
        x = x * y
        v = x * A.dot(x)
        rk = 1.0 - v
        rho_km1 = dot(rk, rk)
        rout = rho_km1
        rat = rout / rold
        rold = rout
        res_norm = rout ** 0.5
        eta_o = eta
        eta = g * rat
        if g * eta_o ** 2.0 > 0.1:
            eta = max(eta, g * eta_o ** 2.0)
        eta = max(min(eta, etamax), stop_tol / res_norm)
    return x, i, k


def bias_vector(x, removed, n_full):
    """HiCKRy.py:103-115: bias = (1/x) / mean(1/x) (np.sum = numpy's pairwise sum), -1 re-inserted at the removed indices."""
    inv = 1.0 / x
    avg = (1.0 * np.sum(inv)) / len(x)
    out = np.full(n_full, -1.0)
    keep = np.ones(n_full, bool)
    keep[removed] = False
    out[keep] = inv / avg
    return out


def run(interactionsFile, fragsFile, perc=0.05):
    x, y, z, n, rev = read_tables(interactionsFile, fragsFile)
    A = assemble(x, y, z, n)
    removed, val, sums = sparse_rows(A, perc)
    R = drop(A, removed)
    xv, i, k = knight_ruiz(R)
    return dict(A=A, removed=removed, val_to_remove=val, row_sums=sums, R=R, x=xv, outer=i, inner=k,
                bias=bias_vector(xv, removed, n), revFrag=rev)
