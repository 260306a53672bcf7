"""oracle/fitpack_oracle.py - TEST INFRASTRUCTURE ONLY (never imported by the product package).

Pure-Python restatement of the smoothing-spline route the reference takes at
/root/reference/fithic/fithic.py:951  (`ius = UnivariateSpline(x, y, s=min(y)**2)`), :961 and :968
(`ius(splineX)`), i.e. scipy.interpolate.UnivariateSpline -> FITPACK (P. Dierckx) `curfit`/`fpcurf`
with k = 3, unit weights, xb = x[0], xe = x[-1], tol = 1e-3, maxit = 20, driven the way
scipy/interpolate/_fitpack2.py does: first call with nest = max(m//2, 2*(k+1)) (nest = m+k+1 when
s == 0), and when FITPACK answers ier == 1 ("nest too small") a *continuation* call (iopt = 1) with
nest = m+k+1.  The dependency (scipy 1.15.3, un-pinned in /root/reference/setup.py:22-28) is not
under /root/reference; the published algorithm (Dierckx, "Curve and Surface Fitting with Splines",
routines fpcurf / fpknot / fpdisc / fpgivs / fprota / fpback / fpbspl / fprati / splev) is restated
here with 1-based index arithmetic kept in comments and plain Python floats (IEEE doubles, no FMA).

Pinned by tests/golden/f4_fitpack.npz (knots, coefficients, fp, ier captured from scipy in the build
container, incl. three nest-restart cases and s = 0) and by the per-pass spline captures in f1_*/f2_*.
"""
import math

K = 3
K1 = K + 1
K2 = K + 2
TOL = 1e-3
MAXIT = 20


def _givens(piv, ww):
    """fpgivs: returns (cos, sin, new diagonal)."""
    store = abs(piv)
    if store >= ww:
        r = ww / piv
        dd = store * math.sqrt(1.0 + r * r)        # r*r, not r ** 2: Python's ** is C pow(), < 1 ulp but not the rounded product
    else:
        r = piv / ww
        dd = ww * math.sqrt(1.0 + r * r)
    return ww / dd, piv / dd, dd


def _bspl(t, x, l):
    """fpbspl: the 4 non-zero cubic B-splines at t[l] <= x < t[l+1] (0-based l)."""
    h = [1.0, 0.0, 0.0, 0.0]
    hh = [0.0, 0.0, 0.0]
    for j in range(1, K + 1):
        for i in range(j):
            hh[i] = h[i]
        h[0] = 0.0
        for i in range(j):
            li = l + 1 + i
            lj = li - j
            if t[li] == t[lj]:
                h[i + 1] = 0.0
                continue
            f = hh[i] / (t[li] - t[lj])
            h[i] = h[i] + f * (t[li] - x)
            h[i + 1] = f * (x - t[lj])
    return h


def _back(a, z, n, bw):
    """fpback: solve the banded upper-triangular system a c = z, band width bw."""
    c = [0.0] * n
    c[n - 1] = z[n - 1] / a[n - 1][0]
    for i in range(n - 2, -1, -1):
        store = z[i]
        i1 = min(bw - 1, n - 1 - i)
        for l in range(1, i1 + 1):
            store = store - c[i + l] * a[i][l]
        c[i] = store / a[i][0]
    return c


def _disc(t, n):
    """fpdisc: discontinuity jumps of the 3rd derivative of the B-splines at the interior knots."""
    nk1 = n - K1
    nrint = nk1 - K
    fac = float(nrint) / (t[nk1] - t[K1 - 1])
    b = []
    h = [0.0] * (2 * K1)
    for l in range(K2, nk1 + 1):            # 1-based l
        for j in range(1, K1 + 1):
            ik = j + K1
            lj = l + j
            lk = lj - K2
            h[j - 1] = t[l - 1] - t[lk - 1]
            h[ik - 1] = t[l - 1] - t[lj - 1]
        row = [0.0] * K2
        lp = l - K1
        for j in range(1, K2 + 1):
            jk = j
            prod = h[j - 1]
            for _ in range(K):
                jk += 1
                prod = prod * h[jk - 1] * fac
            lk = lp + K1
            row[j - 1] = (t[lk - 1] - t[lp - 1]) / prod
            lp += 1
        b.append(row)
    return b


def _rati(p1, f1, p2, f2, p3, f3):
    """fprati: rational interpolation step; returns (p, p1, f1, p3, f3)."""
    if p3 > 0.0:
        h1 = f1 * (f2 - f3)
        h2 = f2 * (f3 - f1)
        h3 = f3 * (f1 - f2)
        p = -(p1 * p2 * h3 + p2 * p3 * h1 + p3 * p1 * h2) / (p1 * h1 + p2 * h2 + p3 * h3)
    else:
        p = (p1 * (f1 - f3) * f2 - p2 * (f2 - f3) * f1) / ((f1 - f2) * f3)
    if f2 < 0.0:
        p3, f3 = p2, f2
    else:
        p1, f1 = p2, f2
    return p, p1, f1, p3, f3


class _State:
    """The FITPACK work arrays that survive between the first call and the continuation call."""

    def __init__(self, m):
        size = m + K1 + 2
        self.t = [0.0] * size
        self.fpint = [0.0] * size
        self.nrdata = [0] * size
        self.n = 0
        self.c = []
        self.fp = 0.0
        self.ier = 0


def _knot(x, m, st, nrint):
    """fpknot: add one knot inside the interval with the largest residual share."""
    t, fpint, nrdata = st.t, st.fpint, st.nrdata
    n = st.n
    k = (n - nrint - 1) // 2
    fpmax = 0.0
    jbegin = 1
    number = maxpt = maxbeg = 0
    for j in range(1, nrint + 1):
        jpoint = nrdata[j - 1]
        if not (fpmax >= fpint[j - 1] or jpoint == 0):
            fpmax = fpint[j - 1]
            number = j
            maxpt = jpoint
            maxbeg = jbegin
        jbegin = jbegin + jpoint + 1
    ihalf = maxpt // 2 + 1
    nrx = maxbeg + ihalf
    nxt = number + 1
    if nxt <= nrint:
        for j in range(nxt, nrint + 1):
            jj = nxt + nrint - j
            fpint[jj] = fpint[jj - 1]
            nrdata[jj] = nrdata[jj - 1]
            jk = jj + k
            t[jk] = t[jk - 1]
    nrdata[number - 1] = ihalf - 1
    nrdata[nxt - 1] = maxpt - ihalf
    am = float(maxpt)
    an = float(nrdata[number - 1])
    fpint[number - 1] = fpmax * an / am
    an = float(nrdata[nxt - 1])
    fpint[nxt - 1] = fpmax * an / am
    jk = nxt + k
    t[jk - 1] = x[nrx - 1]
    st.n = n + 1


def _fpcurf(iopt, x, y, s, nest, st, ier_in):
    """One FITPACK fpcurf call.  Mutates `st`, returns ier."""
    m = len(x)
    xb, xe = x[0], x[-1]
    nmin = 2 * K1
    acc = TOL * s
    nmax = m + K1
    t, fpint, nrdata = st.t, st.fpint, st.nrdata
    ier = ier_in
    fp0 = fpold = 0.0
    nplus = 0

    def interpolation_knots():
        mk1 = m - K1
        i = K2
        j = K // 2 + 2
        for _ in range(mk1):                 # k odd: knots on data points
            t[i - 1] = x[j - 1]
            i += 1
            j += 1

    start_fresh = True
    if s <= 0.0:
        st.n = nmax
        if nmax > nest:
            st.ier = 1
            return 1
        interpolation_knots()
        start_fresh = False
    elif iopt != 0 and st.n != nmin:
        fp0 = fpint[st.n - 1]
        fpold = fpint[st.n - 2]
        nplus = nrdata[st.n - 1]
        if fp0 > s:
            start_fresh = False
    if start_fresh:
        st.n = nmin
        fpold = 0.0
        nplus = 0
        nrdata[0] = m - 2

    a = z = q = None
    nk1 = 0
    fpms = 0.0
    accept = False
    it_outer = 0
    while it_outer < m:
        it_outer += 1
        n = st.n
        if n == nmin:
            ier = -2
        nrint = n - nmin + 1
        nk1 = n - K1
        for j in range(K1):
            t[j] = xb
            t[n - 1 - j] = xe
        # least-squares spline on the current knots: Givens-rotate each observation row into a
        fp = 0.0
        z = [0.0] * nk1
        a = [[0.0] * K1 for _ in range(nk1)]
        q = [None] * m
        l = K1                                # 1-based knot interval index
        for it in range(m):
            xi = x[it]
            yi = y[it]
            while not (xi < t[l] or l == nk1):
                l += 1
            h = _bspl(t, xi, l - 1)
            q[it] = list(h)
            j = l - K1
            for i in range(K1):
                j += 1
                piv = h[i]
                if piv == 0.0:
                    continue
                cs, sn, a[j - 1][0] = _givens(piv, a[j - 1][0])
                yi, z[j - 1] = cs * yi - sn * z[j - 1], cs * z[j - 1] + sn * yi
                if i == K1 - 1:
                    break
                i2 = 0
                for i1 in range(i + 1, K1):
                    i2 += 1
                    h[i1], a[j - 1][i2] = cs * h[i1] - sn * a[j - 1][i2], cs * a[j - 1][i2] + sn * h[i1]
            fp = fp + yi * yi
        if ier == -2:
            fp0 = fp
        fpint[n - 1] = fp0
        fpint[n - 2] = fpold
        nrdata[n - 1] = nplus
        c = _back(a, z, nk1, K1)
        st.c, st.fp = c, fp
        fpms = fp - s
        if abs(fpms) < acc:
            st.ier = ier
            return ier
        if fpms < 0.0:
            accept = True
            break
        if n == nmax:
            st.ier = -1
            return -1
        if n == nest:
            st.ier = 1
            return 1
        if ier == 0:
            npl1 = nplus * 2
            rn = float(nplus)
            if fpold - fp > acc:
                npl1 = int(rn * fpms / (fpold - fp))
            nplus = min(nplus * 2, max(npl1, nplus // 2, 1))
        else:
            nplus = 1
            ier = 0
        fpold = fp
        # residual share of every knot interval
        fpart = 0.0
        i = 1
        l = K2
        new = 0
        for it in range(m):
            if not (x[it] < t[l - 1] or l > nk1):
                new = 1
                l += 1
            term = 0.0
            l0 = l - K2
            for j in range(K1):
                l0 += 1
                term = term + c[l0 - 1] * q[it][j]
            term = (term - y[it]) * (term - y[it])
            fpart = fpart + term
            if new:
                store = term * 0.5
                fpint[i - 1] = fpart - store
                i += 1
                fpart = store
                new = 0
        fpint[nrint - 1] = fpart
        for _ in range(nplus):
            _knot(x, m, st, nrint)
            nrint += 1
            if st.n == nmax:
                interpolation_knots()
                break
            if st.n == nest:
                break
    if not accept:
        st.ier = ier
        return ier
    if ier == -2:
        st.ier = ier
        return ier

    # ---- part 2: smoothing parameter p with F(p) = s -----------------------------------------
    n = st.n
    b = _disc(t, n)
    p1, f1 = 0.0, fp0 - s
    p3, f3 = -1.0, fpms
    p = 0.0
    for i in range(nk1):
        p = p + a[i][0]
    p = float(nk1) / p
    ich1 = ich3 = 0
    n8 = n - nmin
    for it_p in range(1, MAXIT + 1):
        pinv = 1.0 / p
        c = list(z)
        g = [row[:] + [0.0] for row in a]
        for it in range(1, n8 + 1):
            h = [b[it - 1][i] * pinv for i in range(K2)] + [0.0]
            yi = 0.0
            for j in range(it, nk1 + 1):
                piv = h[0]
                cs, sn, g[j - 1][0] = _givens(piv, g[j - 1][0])
                yi, c[j - 1] = cs * yi - sn * c[j - 1], cs * c[j - 1] + sn * yi
                if j == nk1:
                    break
                i2 = K1
                if j > n8:
                    i2 = nk1 - j
                for i in range(1, i2 + 1):
                    hv, gv = h[i], g[j - 1][i]
                    g[j - 1][i] = cs * gv + sn * hv
                    h[i - 1] = cs * hv - sn * gv
                h[i2] = 0.0
        c = _back(g, c, nk1, K2)
        fp = 0.0
        l = K2
        for it in range(m):
            if not (x[it] < t[l - 1] or l > nk1):
                l += 1
            l0 = l - K2
            term = 0.0
            for j in range(K1):
                l0 += 1
                term = term + c[l0 - 1] * q[it][j]
            fp = fp + (term - y[it]) * (term - y[it])
        st.c, st.fp = c, fp
        fpms = fp - s
        if abs(fpms) < acc:
            st.ier = ier
            return ier
        if it_p == MAXIT:
            st.ier = 3
            return 3
        p2, f2 = p, fpms
        if ich3 == 0:
            if (f2 - f3) <= acc:
                p3, f3 = p2, f2
                p = p * 0.04
                if p <= p1:
                    p = p1 * 0.9 + p2 * 0.1
                continue
            if f2 < 0.0:
                ich3 = 1
        if ich1 == 0:
            if (f1 - f2) <= acc:
                p1, f1 = p2, f2
                p = p / 0.04
                if p3 < 0.0:
                    continue
                if p >= p3:
                    p = p2 * 0.1 + p3 * 0.9
                continue
            if f2 > 0.0:
                ich1 = 1
        if f2 >= f1 or f2 <= f3:
            st.ier = 2
            return 2
        p, p1, f1, p3, f3 = _rati(p1, f1, p2, f2, p3, f3)
    st.ier = 3
    return 3


def univariate_spline(x, y, s):
    """(t, c, fp, ier) as scipy.interpolate.UnivariateSpline(x, y, s=s) (k=3, w=None) holds them."""
    x = [float(v) for v in x]
    y = [float(v) for v in y]
    s = float(s)
    m = len(x)
    if m <= K:
        raise ValueError("need more than k data points")
    nest = m + K1 if s == 0.0 else max(m // 2, 2 * K1)
    st = _State(m)
    ier = _fpcurf(0, x, y, s, nest, st, 0)
    restarted = False
    if ier == 1:
        restarted = True
        ier = _fpcurf(1, x, y, s, m + K1, st, 1)
    n = st.n
    return st.t[:n], st.c[:n - K1], st.fp, ier, restarted


def splev(t, c, xs):
    """FITPACK splev (ext = 0) for an ascending or arbitrary list of points inside [t[3], t[n-4]]."""
    n = len(t)
    nk1 = n - K1
    tb, te = t[K1 - 1], t[nk1]
    out = []
    l = K1
    for arg in xs:
        arg = float(arg)
        # ext=0 extrapolates with the end polynomial; the reference never evaluates outside [xb, xe]
        while not (arg >= t[l - 1] or l == K1):
            l -= 1
        while not (arg < t[l] or l == nk1):
            l += 1
        h = _bspl(t, arg, l - 1)
        sp = 0.0
        ll = l - K1
        for j in range(K1):
            sp = sp + c[ll + j] * h[j]
        out.append(sp)
    return out
