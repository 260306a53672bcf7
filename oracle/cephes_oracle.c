/*
 * oracle/cephes_oracle.c - TEST INFRASTRUCTURE ONLY (not product code, never shipped, never on the
 * product path).  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it.
 *
 * Scalar plain-C restatement of the binomial survival function the reference calls once per
 * contact pair:  scipy.special.bdtrc(count-1, n, prior)          (/root/reference/fithic/fithic.py:1070,1101)
 * The arithmetic lives in a third-party dependency that is NOT under /root/reference:
 *   scipy 1.15.3 (unpinned in /root/reference/setup.py:22-28), scipy.special -> xsf/cephes
 *   (Cephes Math Library 2.x, S. L. Moshier): bdtr.c / incbet.c / beta.c / gamma.c / rgamma.c.
 * The published algorithm is restated here operation for operation (same association, same
 * constants, same iteration caps, no FMA contraction: build with -ffp-contract=off), because the
 * acceptance bar is "<= 1e-10 of bdtrc", and bdtrc itself is only ~1e-8 accurate at Hi-C sized n
 * (SURVEY.md facts 2-4).
 *
 * Pinned by: tests/golden/f3_bdtrc.npz (scipy.special.bdtrc / betaln bit patterns generated in the
 * build container by tests/golden/make_golden.py) and, where scipy is importable, live comparison.
 *
 * TOTALS AT AND ABOVE 2^31.  scipy's bdtrc takes n as a C int: the reference's Python-int totals are narrowed to 32 bits
 * (n = 2^31 -> NaN, n = 2^32 + 10^6 -> 10^6).  fho_bdtrc() does the same and is pinned there by
 * tests/golden/f15_bdtrc_int_n.npz and the f15_* whole runs of the real reference; fho_bdtrc_wide() is the second, explicitly
 * named mode (the true total as a real number) - "parity unpinned" above 2^31 by construction: no reference computes it.
 */
#include <math.h>
#include <stdint.h>
#include <stddef.h>

#define FHO_MACHEP 1.11022302462515654042E-16
#define FHO_MAXLOG 7.09782712893383996732E2
#define FHO_MINLOG (-7.451332191019412076235E2)
#define FHO_MAXGAM 171.624376956302725
#define FHO_BIG 4.503599627370496e15
#define FHO_BIGINV 2.22044604925031308085e-16
#define FHO_LS2PI 0.91893853320467274178
#define FHO_SQRTPI 2.50662827463100050242E0
#define FHO_MAXSTIR 143.01608

/* ---- polynomial helpers (Horner, highest power first) ------------------------------------ */
static double horner(double x, const double *c, int degree)
{
    double acc = c[0];
    for (int i = 1; i <= degree; ++i)
        acc = acc * x + c[i];
    return acc;
}

static double horner_monic(double x, const double *c, int n)   /* leading coefficient 1 implied */
{
    double acc = x + c[0];
    for (int i = 1; i < n; ++i)
        acc = acc * x + c[i];
    return acc;
}

static double chebyshev(double x, const double *c, int n)
{
    double b0 = c[0], b1 = 0.0, b2 = 0.0;
    for (int i = 1; i < n; ++i) {
        b2 = b1;
        b1 = b0;
        b0 = x * b1 - b2 + c[i];
    }
    return 0.5 * (b0 - b2);
}


/* ---- Cephes unity.c log1p / expm1 (scipy's bdtrc calls these, not libm's) ------------------- */
static const double U_LP[7] = {4.5270000862445199635215E-5, 4.9854102823193375972212E-1, 6.5787325942061044846969E0,
                               2.9911919328553073277375E1, 6.0949667980987787057556E1, 5.7112963590585538103336E1,
                               2.0039553499201281259648E1};
static const double U_LQ[6] = {1.5062909083469192043167E1, 8.3047565967967209469434E1, 2.2176239823732856465394E2,
                               3.0909872225312059774938E2, 2.1642788614495947685003E2, 6.0118660497603843919306E1};
static const double U_EP[3] = {1.2617719307481059087798E-4, 3.0299440770744196129956E-2, 9.9999999999999999991025E-1};
static const double U_EQ[4] = {3.0019850513866445504159E-6, 2.5244834034968410419224E-3, 2.2726554820815502876593E-1,
                               2.0000000000000000000897E0};

double fho_log1p(double x)
{
    double z = 1.0 + x;
    if (z < 0.70710678118654752440 || z > 1.41421356237309504880)
        return log(z);
    z = x * x;
    z = -0.5 * z + x * (z * horner(x, U_LP, 6) / horner_monic(x, U_LQ, 6));
    return x + z;
}

double fho_expm1(double x)
{
    if (!isfinite(x)) {
        if (isnan(x) || x > 0)
            return x;
        return -1.0;
    }
    if (x < -0.5 || x > 0.5)
        return exp(x) - 1.0;
    double xx = x * x;
    double r = x * horner(xx, U_EP, 2);
    r = r / (horner(xx, U_EQ, 3) - r);
    return r + r;
}

/* ---- log-gamma for x > 0 (gamma.c lgam): Stirling tails + rational on [2,3) ---------------- */
static const double LG_A[5] = {8.11614167470508450300E-4, -5.95061904284301438324E-4,
                               7.93650340457716943945E-4, -2.77777777730099687205E-3,
                               8.33333333333331927722E-2};
static const double LG_B[6] = {-1.37825152569120859100E3, -3.88016315134637840924E4,
                               -3.31612992738871184744E5, -1.16237097492762307383E6,
                               -1.72173700820839662146E6, -8.53555664245765465627E5};
static const double LG_C[6] = {-3.51815701436523470549E2, -1.70642106651881159223E4,
                               -2.20528590553854454839E5, -1.13933444367982507207E6,
                               -2.53252307177582951285E6, -2.01889141433532773231E6};

double fho_lgam(double x)
{
    if (!isfinite(x))
        return x;
    if (x <= 0.0)
        return INFINITY;                       /* poles / negative side are never reached by this path */
    if (x < 13.0) {
        double z = 1.0, p = 0.0, u = x;
        while (u >= 3.0) {
            p -= 1.0;
            u = x + p;
            z *= u;
        }
        while (u < 2.0) {
            if (u == 0.0)
                return INFINITY;
            z /= u;
            p += 1.0;
            u = x + p;
        }
        if (z < 0.0)
            z = -z;
        if (u == 2.0)
            return log(z);
        p -= 2.0;
        x = x + p;
        p = x * horner(x, LG_B, 5) / horner_monic(x, LG_C, 6);
        return log(z) + p;
    }
    if (x > 2.556348e305)
        return INFINITY;
    double q = (x - 0.5) * log(x) - x + FHO_LS2PI;
    if (x >= 1000.0) {
        if (x > 1.0e8)
            return q;
        double p = 1.0 / (x * x);
        p = ((7.9365079365079365079365e-4 * p - 2.7777777777777777777778e-3) * p + 0.0833333333333333333333) / x;
        return q + p;
    }
    double p = 1.0 / (x * x);
    return q + horner(p, LG_A, 4) / x;
}

/* ---- Gamma(x) for 0 < x < MAXGAM and 1/Gamma (gamma.c, rgamma.c), only for the tiny-n path -- */
static const double GM_P[7] = {1.60119522476751861407E-4, 1.19135147006586384913E-3, 1.04213797561761569935E-2,
                               4.76367800457137231464E-2, 2.07448227648435975150E-1, 4.94214826801497100753E-1,
                               9.99999999999999996796E-1};
static const double GM_Q[8] = {-2.31581873324120129819E-5, 5.39605580493303397842E-4, -4.45641913851797240494E-3,
                               1.18139785222060435552E-2, 3.58236398605498653373E-2, -2.34591795718243348568E-1,
                               7.14304917030273074085E-2, 1.00000000000000000320E0};
static const double GM_STIR[5] = {7.87311395793093628397E-4, -2.29549961613378126380E-4, -2.68132617805781232825E-3,
                                  3.47222221605458667310E-3, 8.33333333333482257126E-2};
static const double RG_R[16] = {3.13173458231230000000E-17, -6.70718606477908000000E-16, 2.20039078172259550000E-15,
                                2.47691630348254132600E-13, -6.60074100411295197440E-12, 5.13850186324226978840E-11,
                                1.08965386454418662084E-9, -3.33964630686836942556E-8, 2.68975996440595483619E-7,
                                2.96001177518801696639E-6, -8.04814124978471142852E-5, 4.16609138709688864714E-4,
                                5.06579864028608725080E-3, -6.41925436109158228810E-2, -4.98558728684003594785E-3,
                                1.27546015610523951063E-1};

static double gamma_stirling(double x)
{
    if (x >= FHO_MAXGAM)
        return INFINITY;
    double w = 1.0 / x;
    w = 1.0 + w * horner(w, GM_STIR, 4);
    double y = exp(x);
    if (x > FHO_MAXSTIR) {
        double v = pow(x, 0.5 * x - 0.25);
        y = v * (v / y);
    } else {
        y = pow(x, x - 0.5) / y;
    }
    return FHO_SQRTPI * y * w;
}

double fho_gamma_pos(double x)          /* x > 0 only */
{
    if (!isfinite(x))
        return x;
    if (x > 33.0)
        return gamma_stirling(x);
    double z = 1.0;
    while (x >= 3.0) {
        x -= 1.0;
        z *= x;
    }
    while (x < 2.0) {
        if (x < 1.e-9)
            return z / ((1.0 + 0.5772156649015329 * x) * x);
        z /= x;
        x += 1.0;
    }
    if (x == 2.0)
        return z;
    x -= 2.0;
    return z * horner(x, GM_P, 6) / horner(x, GM_Q, 7);
}

static double rgamma_pos(double x)      /* 1/Gamma(x), x > 0 */
{
    if (x > 4.0)
        return 1.0 / fho_gamma_pos(x);
    double z = 1.0, w = x;
    while (w > 1.0) {
        w -= 1.0;
        z *= w;
    }
    if (w == 0.0)
        return 0.0;
    if (w == 1.0)
        return 1.0 / z;
    return w * (1.0 + chebyshev(4.0 * w - 2.0, RG_R, 16)) / z;
}

/* ---- beta / lbeta for a, b > 0 (beta.c) ---------------------------------------------------- */
static double lbeta_asymptotic(double a, double b)      /* a >> b */
{
    double r = fho_lgam(b);
    r -= b * log(a);
    r += b * (1 - b) / (2 * a);
    r += b * (1 - b) * (1 - 2 * b) / (12 * a * a);
    r += -b * b * (1 - b) * (1 - b) / (12 * a * a * a);
    return r;
}

double fho_lbeta(double a, double b)
{
    if (a <= 0.0 || b <= 0.0)
        return INFINITY;
    if (fabs(a) < fabs(b)) {
        double s = a;
        a = b;
        b = s;
    }
    if (fabs(a) > 1e6 * fabs(b) && a > 1e6)
        return lbeta_asymptotic(a, b);
    double y = a + b;
    if (fabs(y) > FHO_MAXGAM || fabs(a) > FHO_MAXGAM || fabs(b) > FHO_MAXGAM) {
        y = fho_lgam(y);
        y = fho_lgam(b) - y;
        y = fho_lgam(a) + y;
        return y;
    }
    y = rgamma_pos(y);
    a = fho_gamma_pos(a);
    b = fho_gamma_pos(b);
    if (isinf(y))
        return INFINITY;
    if (fabs(fabs(a * y) - 1.0) > fabs(fabs(b * y) - 1.0)) {
        y = b * y;
        y *= a;
    } else {
        y = a * y;
        y *= b;
    }
    if (y < 0)
        y = -y;
    return log(y);
}

double fho_beta(double a, double b)
{
    if (a <= 0.0 || b <= 0.0)
        return INFINITY;
    if (fabs(a) < fabs(b)) {
        double s = a;
        a = b;
        b = s;
    }
    if (fabs(a) > 1e6 * fabs(b) && a > 1e6)
        return exp(lbeta_asymptotic(a, b));
    double y = a + b;
    if (fabs(y) > FHO_MAXGAM || fabs(a) > FHO_MAXGAM || fabs(b) > FHO_MAXGAM) {
        y = fho_lgam(y);
        y = fho_lgam(b) - y;
        y = fho_lgam(a) + y;
        if (y > FHO_MAXLOG)
            return INFINITY;
        return exp(y);
    }
    y = rgamma_pos(y);
    a = fho_gamma_pos(a);
    b = fho_gamma_pos(b);
    if (isinf(y))
        return INFINITY;
    if (fabs(fabs(a * y) - 1.0) > fabs(fabs(b * y) - 1.0)) {
        y = b * y;
        y *= a;
    } else {
        y = a * y;
        y *= b;
    }
    return y;
}

/* ---- incomplete beta pieces (incbet.c) ------------------------------------------------------ */
/* iteration counters for diagnostics (branch statistics in tests / DESIGN.md appendix) */
static int g_last_branch;      /* 0 k==0 closed form, 1 pseries, 2 incbcf, 3 incbd, +4 when swapped */
static int g_last_iters;

static double power_series(double a, double b, double x)
{
    double ai = 1.0 / a;
    double u = (1.0 - b) * x;
    double v = u / (a + 1.0);
    double t1 = v;
    double t = u;
    double n = 2.0;
    double s = 0.0;
    double z = FHO_MACHEP * ai;
    int it = 0;
    while (fabs(v) > z) {
        u = (n - b) * x / n;
        t *= u;
        v = t / (a + n);
        s += v;
        n += 1.0;
        ++it;
    }
    g_last_iters = it;
    s += t1;
    s += ai;
    u = a * log(x);
    if ((a + b) < FHO_MAXGAM && fabs(u) < FHO_MAXLOG) {
        t = 1.0 / fho_beta(a, b);
        s = s * t * pow(x, a);
    } else {
        t = -fho_lbeta(a, b) + u + log(s);
        s = (t < FHO_MINLOG) ? 0.0 : exp(t);
    }
    return s;
}

/* Both continued fractions share one skeleton; `which` = 0 -> incbcf, 1 -> incbd. */
static double continued_fraction(double a, double b, double x, int which)
{
    double k1, k2, k3, k4, k5, k6, k7, k8, d2, d6, arg;
    if (which == 0) {
        arg = x;
        k1 = a; k2 = a + b; k3 = a; k4 = a + 1.0; k5 = 1.0; k6 = b - 1.0; k7 = k4; k8 = a + 2.0;
        d2 = 1.0; d6 = -1.0;
    } else {
        arg = x / (1.0 - x);
        k1 = a; k2 = b - 1.0; k3 = a; k4 = a + 1.0; k5 = 1.0; k6 = a + b; k7 = a + 1.0; k8 = a + 2.0;
        d2 = -1.0; d6 = 1.0;
    }
    double pkm2 = 0.0, qkm2 = 1.0, pkm1 = 1.0, qkm1 = 1.0, ans = 1.0, r = 1.0;
    const double thresh = 3.0 * FHO_MACHEP;
    int n = 0;
    do {
        double xk = -(arg * k1 * k2) / (k3 * k4);
        double pk = pkm1 + pkm2 * xk;
        double qk = qkm1 + qkm2 * xk;
        pkm2 = pkm1; pkm1 = pk; qkm2 = qkm1; qkm1 = qk;

        xk = (arg * k5 * k6) / (k7 * k8);
        pk = pkm1 + pkm2 * xk;
        qk = qkm1 + qkm2 * xk;
        pkm2 = pkm1; pkm1 = pk; qkm2 = qkm1; qkm1 = qk;

        double t;
        if (qk != 0)
            r = pk / qk;
        if (r != 0) {
            t = fabs((ans - r) / r);
            ans = r;
        } else {
            t = 1.0;
        }
        if (t < thresh)
            break;

        k1 += 1.0; k2 += d2; k3 += 2.0; k4 += 2.0; k5 += 1.0; k6 += d6; k7 += 2.0; k8 += 2.0;

        if ((fabs(qk) + fabs(pk)) > FHO_BIG) {
            pkm2 *= FHO_BIGINV; pkm1 *= FHO_BIGINV; qkm2 *= FHO_BIGINV; qkm1 *= FHO_BIGINV;
        }
        if ((fabs(qk) < FHO_BIGINV) || (fabs(pk) < FHO_BIGINV)) {
            pkm2 *= FHO_BIG; pkm1 *= FHO_BIG; qkm2 *= FHO_BIG; qkm1 *= FHO_BIG;
        }
    } while (++n < 300);
    g_last_iters = n;
    return ans;
}

double fho_incbet(double aa, double bb, double xx)
{
    if (aa <= 0.0 || bb <= 0.0)
        return NAN;
    if (xx <= 0.0 || xx >= 1.0) {
        if (xx == 0.0)
            return 0.0;
        if (xx == 1.0)
            return 1.0;
        return NAN;
    }
    if (bb * xx <= 1.0 && xx <= 0.95) {
        g_last_branch = 1;
        return power_series(aa, bb, xx);
    }
    double w = 1.0 - xx;
    double a, b, x, xc, t, y;
    int flag;
    if (xx > aa / (aa + bb)) {
        flag = 1; a = bb; b = aa; xc = xx; x = w;
    } else {
        flag = 0; a = aa; b = bb; xc = w; x = xx;
    }
    if (flag == 1 && b * x <= 1.0 && x <= 0.95) {
        g_last_branch = 1 + 4;
        t = power_series(a, b, x);
    } else {
        y = x * (a + b - 2.0) - (a - 1.0);
        if (y < 0.0) {
            g_last_branch = 2 + 4 * flag;
            w = continued_fraction(a, b, x, 0);
        } else {
            g_last_branch = 3 + 4 * flag;
            w = continued_fraction(a, b, x, 1) / xc;
        }
        y = a * log(x);
        t = b * log(xc);
        if ((a + b) < FHO_MAXGAM && fabs(y) < FHO_MAXLOG && fabs(t) < FHO_MAXLOG) {
            t = pow(xc, b);
            t *= pow(x, a);
            t /= a;
            t *= w;
            t *= 1.0 / fho_beta(a, b);
        } else {
            y += t - fho_lbeta(a, b);
            y += log(w / a);
            t = (y < FHO_MINLOG) ? 0.0 : exp(y);
        }
    }
    if (flag == 1) {
        if (t <= FHO_MACHEP)
            t = 1.0 - FHO_MACHEP;
        else
            t = 1.0 - t;
    }
    return t;
}

/* Cephes' bdtrc with n as a real number: the arithmetic of bdtr.c on a total that is NOT narrowed to a C int.  This is the
 * oracle's SECOND mode ("wide totals"): for n < 2^31 it is scipy.special.bdtrc bit for bit; at and above 2^31 it is what the
 * formula would give on the true total - NOT what the reference writes there (see fho_bdtrc below). */
double fho_bdtrc_wide(double k, double n, double p)
{
    g_last_branch = -1;
    g_last_iters = 0;
    if (isnan(p) || isnan(k))
        return NAN;
    double fk = floor(k);
    if (p < 0.0 || p > 1.0 || n < fk)
        return NAN;
    if (fk < 0)
        return 1.0;
    if (fk == n)
        return 0.0;
    double dn = n - fk;
    if (k == 0) {
        g_last_branch = 0;
        if (p < 0.01)
            return -fho_expm1(dn * fho_log1p(-p));
        return 1.0 - pow(1.0 - p, dn);
    }
    return fho_incbet(fk + 1, dn, p);
}

/* What a C `int` holds after scipy's ufunc loop `dld->d` narrows the C long it received: the value modulo 2^32, in
 * [-2^31, 2^31).  n must be integral and |n| < 2^63 (the reference's totals are Python ints of counts: exact in a double up
 * to 2^53). */
double fho_int_narrowed(double n)
{
    if (!(fabs(n) < 9.2e18))
        return n;                                 /* NaN / out of a C long: not the reference's call, left alone */
    return (double)(int32_t)(uint32_t)(uint64_t)(int64_t)n;
}

/* scipy.special.bdtrc(k, n, p) as the reference calls it (fithic.py:1070,1101: count-1, a Python int total, prior):
 * scipy 1.15.3 binds Cephes' `double bdtrc(double k, int n, double p)`, so the total is narrowed to 32 bits first -
 * n = 2^31 arrives as -2^31 (n < k: NaN), n = 2^32 + 10^6 as 10^6.  Pinned by tests/golden/f15_bdtrc_int_n.npz (scipy's own
 * values at fifteen totals from 2^31 - 1 to 2^52) and the f15_* whole-run fixtures.  This is the DEFAULT mode of the oracle. */
double fho_bdtrc(double k, double n, double p)
{
    return fho_bdtrc_wide(k, fho_int_narrowed(n), p);
}

/* ---- vector entry points used through ctypes ------------------------------------------------ */
void fho_bdtrc_vec(const double *k, const double *n, const double *p, double *out, int64_t len)
{
    for (int64_t i = 0; i < len; ++i)
        out[i] = fho_bdtrc(k[i], n[i], p[i]);
}

void fho_bdtrc_wide_vec(const double *k, const double *n, const double *p, double *out, int64_t len)
{
    for (int64_t i = 0; i < len; ++i)
        out[i] = fho_bdtrc_wide(k[i], n[i], p[i]);
}

void fho_bdtrc_vec_stats(const double *k, const double *n, const double *p, double *out,
                         int32_t *branch, int32_t *iters, int64_t len)
{
    for (int64_t i = 0; i < len; ++i) {
        out[i] = fho_bdtrc(k[i], n[i], p[i]);
        branch[i] = g_last_branch;
        iters[i] = g_last_iters;
    }
}

void fho_contfrac_vec(int which, const double *a, const double *b, const double *x, double *out, int64_t len)
{
    for (int64_t i = 0; i < len; ++i)
        out[i] = continued_fraction(a[i], b[i], x[i], which);
}

void fho_lbeta_vec(const double *a, const double *b, double *out, int64_t len)
{
    for (int64_t i = 0; i < len; ++i)
        out[i] = fho_lbeta(a[i], b[i]);
}

void fho_log_vec(const double *x, double *out, int64_t len)
{
    for (int64_t i = 0; i < len; ++i)
        out[i] = log(x[i]);
}
