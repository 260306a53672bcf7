// fhx_host.cpp - host stages of the MI355X Fit-Hi-C engine (see fhx_host.hpp).
// Build with -ffp-contract=off -fno-fast-math: results depend on the exact sequence of IEEE operations.
#include "fhx_host.hpp"

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstring>
#include <limits>
#include <thread>
#include <vector>

#include "fhx_cpus.hpp"

#include "../../include/fithic_mi355x.h"

namespace fhx {

// ===================================================================================================
// Cephes pieces needed on the host (gamma.c lgam, beta.c lbeta/beta, rgamma.c)
// ===================================================================================================
namespace {

constexpr double kMaxGam = 171.624376956302725;
constexpr double kMaxLog = 7.09782712893383996732E2;
constexpr double kLogSqrt2Pi = 0.91893853320467274178;
constexpr double kSqrt2Pi = 2.50662827463100050242E0;   // Cephes calls it SQRTPI
constexpr double kMaxStirling = 143.01608;
constexpr double kInf = std::numeric_limits<double>::infinity();

template <int N>
inline double poly(double x, const double (&c)[N]) {      // c[0] x^(N-1) + ... + c[N-1]
    double r = c[0];
    for (int i = 1; i < N; ++i) r = r * x + c[i];
    return r;
}
template <int N>
inline double poly_monic(double x, const double (&c)[N]) {  // x^N + c[0] x^(N-1) + ... + c[N-1]
    double r = x + c[0];
    for (int i = 1; i < N; ++i) r = r * x + c[i];
    return r;
}

constexpr double kLgA[5] = {8.11614167470508450300E-4, -5.95061904284301438324E-4, 7.93650340457716943945E-4,
                            -2.77777777730099687205E-3, 8.33333333333331927722E-2};
constexpr double kLgB[6] = {-1.37825152569120859100E3, -3.88016315134637840924E4, -3.31612992738871184744E5,
                            -1.16237097492762307383E6, -1.72173700820839662146E6, -8.53555664245765465627E5};
constexpr double kLgC[6] = {-3.51815701436523470549E2, -1.70642106651881159223E4, -2.20528590553854454839E5,
                            -1.13933444367982507207E6, -2.53252307177582951285E6, -2.01889141433532773231E6};
constexpr double kGamP[7] = {1.60119522476751861407E-4, 1.19135147006586384913E-3, 1.04213797561761569935E-2,
                             4.76367800457137231464E-2, 2.07448227648435975150E-1, 4.94214826801497100753E-1,
                             9.99999999999999996796E-1};
constexpr double kGamQ[8] = {-2.31581873324120129819E-5, 5.39605580493303397842E-4, -4.45641913851797240494E-3,
                             1.18139785222060435552E-2, 3.58236398605498653373E-2, -2.34591795718243348568E-1,
                             7.14304917030273074085E-2, 1.00000000000000000320E0};
constexpr double kStir[5] = {7.87311395793093628397E-4, -2.29549961613378126380E-4, -2.68132617805781232825E-3,
                             3.47222221605458667310E-3, 8.33333333333482257126E-2};
constexpr double kRgam[16] = {3.13173458231230000000E-17, -6.70718606477908000000E-16, 2.20039078172259550000E-15,
                              2.47691630348254132600E-13, -6.60074100411295197440E-12, 5.13850186324226978840E-11,
                              1.08965386454418662084E-9, -3.33964630686836942556E-8, 2.68975996440595483619E-7,
                              2.96001177518801696639E-6, -8.04814124978471142852E-5, 4.16609138709688864714E-4,
                              5.06579864028608725080E-3, -6.41925436109158228810E-2, -4.98558728684003594785E-3,
                              1.27546015610523951063E-1};

double gamma_positive(double x) {        // Cephes Gamma() restricted to x > 0
    if (!std::isfinite(x)) return x;
    if (x > 33.0) {                       // Stirling
        if (x >= kMaxGam) return kInf;
        double w = 1.0 / x;
        w = 1.0 + w * poly(w, kStir);
        double y = std::exp(x);
        if (x > kMaxStirling) {
            double v = std::pow(x, 0.5 * x - 0.25);
            y = v * (v / y);
        } else {
            y = std::pow(x, x - 0.5) / y;
        }
        return kSqrt2Pi * y * w;
    }
    double z = 1.0;
    while (x >= 3.0) {
        x -= 1.0;
        z *= x;
    }
    while (x < 2.0) {
        if (x < 1.e-9) return z / ((1.0 + 0.5772156649015329 * x) * x);
        z /= x;
        x += 1.0;
    }
    if (x == 2.0) return z;
    x -= 2.0;
    return z * poly(x, kGamP) / poly(x, kGamQ);
}

double rgamma_positive(double x) {       // 1 / Gamma(x), x > 0
    if (x > 4.0) return 1.0 / gamma_positive(x);
    double z = 1.0, w = x;
    while (w > 1.0) {
        w -= 1.0;
        z *= w;
    }
    if (w == 0.0) return 0.0;
    if (w == 1.0) return 1.0 / z;
    // Chebyshev series on [0,1] (Clenshaw)
    const double arg = 4.0 * w - 2.0;
    double b0 = kRgam[0], b1 = 0.0, b2 = 0.0;
    for (int i = 1; i < 16; ++i) {
        b2 = b1;
        b1 = b0;
        b0 = arg * b1 - b2 + kRgam[i];
    }
    return w * (1.0 + 0.5 * (b0 - b2)) / z;
}

double lbeta_large_ratio(double a, double b) {   // a > 1e6 * b: avoids lgam(a+b) - lgam(a) cancellation
    double r = cephes_lgam(b);
    r -= b * std::log(a);
    r += b * (1 - b) / (2 * a);
    r += b * (1 - b) * (1 - 2 * b) / (12 * a * a);
    r += -b * b * (1 - b) * (1 - b) / (12 * a * a * a);
    return r;
}

}  // namespace

double cephes_lgam(double x) {
    if (!std::isfinite(x)) return x;
    if (x <= 0.0) return kInf;            // never reached: arguments are counts >= 1
    if (x < 13.0) {
        double z = 1.0, p = 0.0, u = x;
        while (u >= 3.0) {
            p -= 1.0;
            u = x + p;
            z *= u;
        }
        while (u < 2.0) {
            if (u == 0.0) return kInf;
            z /= u;
            p += 1.0;
            u = x + p;
        }
        if (z < 0.0) z = -z;
        if (u == 2.0) return std::log(z);
        p -= 2.0;
        x = x + p;
        p = x * poly(x, kLgB) / poly_monic(x, kLgC);
        return std::log(z) + p;
    }
    if (x > 2.556348e305) return kInf;
    const double q = (x - 0.5) * std::log(x) - x + kLogSqrt2Pi;
    if (x >= 1000.0) {
        if (x > 1.0e8) return q;
        double p = 1.0 / (x * x);
        p = ((7.9365079365079365079365e-4 * p - 2.7777777777777777777778e-3) * p + 0.0833333333333333333333) / x;
        return q + p;
    }
    const double p = 1.0 / (x * x);
    return q + poly(p, kLgA) / x;
}

static double beta_small_args(double a, double b, bool take_log) {   // a+b, a, b all < MAXGAM
    double y = rgamma_positive(a + b);
    a = gamma_positive(a);
    b = gamma_positive(b);
    if (std::isinf(y)) return kInf;
    if (std::fabs(std::fabs(a * y) - 1.0) > std::fabs(std::fabs(b * y) - 1.0)) {
        y = b * y;
        y *= a;
    } else {
        y = a * y;
        y *= b;
    }
    if (!take_log) return y;
    if (y < 0) y = -y;
    return std::log(y);
}

double cephes_lbeta(double a, double b) {
    if (a <= 0.0 || b <= 0.0) return kInf;
    if (std::fabs(a) < std::fabs(b)) std::swap(a, b);
    if (std::fabs(a) > 1e6 * std::fabs(b) && a > 1e6) return lbeta_large_ratio(a, b);
    double y = a + b;
    if (std::fabs(y) > kMaxGam || std::fabs(a) > kMaxGam || std::fabs(b) > kMaxGam) {
        y = cephes_lgam(y);
        y = cephes_lgam(b) - y;
        y = cephes_lgam(a) + y;
        return y;
    }
    return beta_small_args(a, b, true);
}

double cephes_beta(double a, double b) {
    if (a <= 0.0 || b <= 0.0) return kInf;
    if (std::fabs(a) < std::fabs(b)) std::swap(a, b);
    if (std::fabs(a) > 1e6 * std::fabs(b) && a > 1e6) return std::exp(lbeta_large_ratio(a, b));
    double y = a + b;
    if (std::fabs(y) > kMaxGam || std::fabs(a) > kMaxGam || std::fabs(b) > kMaxGam) {
        y = cephes_lgam(y);
        y = cephes_lgam(b) - y;
        y = cephes_lgam(a) + y;
        if (y > kMaxLog) return kInf;
        return std::exp(y);
    }
    return beta_small_args(a, b, false);
}

// entries [c_lo, c_hi] of the two tables (zero-filled by the caller): every entry on its own, so ranges can be built side by side
void fill_lbeta_table(double n_total, int64_t c_lo, int64_t c_hi, double* lbeta, double* inv_beta) {
    const bool small = (n_total + 1.0) < kMaxGam;      // incbet's "a + b < MAXGAM" with a + b = n + 1
    for (int64_t c = std::max<int64_t>(c_lo, 1); c <= c_hi; ++c) {
        const double a = static_cast<double>(c);
        const double b = n_total - a + 1.0;            // dn = n - (count-1)
        if (b <= 0.0) break;                           // count > n: bdtrc returns NaN / 0 before incbet
        lbeta[c] = cephes_lbeta(a, b);
        if (small) inv_beta[c] = 1.0 / cephes_beta(a, b);
    }
}

void build_lbeta_table(double n_total, int64_t max_count, std::vector<double>& lbeta, std::vector<double>& inv_beta) {
    if (max_count < 0) max_count = 0;
    lbeta.assign(static_cast<size_t>(max_count) + 1, 0.0);
    inv_beta.assign(static_cast<size_t>(max_count) + 1, 0.0);
    fill_lbeta_table(n_total, 1, max_count, lbeta.data(), inv_beta.data());
}

// ===================================================================================================
// FITPACK fpcurf (k = 3, unit weights), restated from Dierckx's published routines.
// Index arithmetic is 1-based in comments where it mirrors the original description.
// ===================================================================================================
namespace {

constexpr int K = 3, K1 = 4, K2 = 5;
constexpr double kTol = 1e-3;
constexpr int kMaxIt = 20;
constexpr int kLanes = 4;            // rows whose Givens chains run interleaved (fpcurf below)

struct Rot {
    double cs, sn;
};
inline Rot givens(double piv, double& ww) {
    const double store = std::fabs(piv);
    double dd;
    if (store >= ww)
        dd = store * std::sqrt(1.0 + (ww / piv) * (ww / piv));
    else
        dd = ww * std::sqrt(1.0 + (piv / ww) * (piv / ww));
    Rot r{ww / dd, piv / dd};
    ww = dd;
    return r;
}
inline void rotate(const Rot& r, double& a, double& b) {
    const double s1 = a, s2 = b;
    b = r.cs * s2 + r.sn * s1;
    a = r.cs * s1 - r.sn * s2;
}

// the 4 non-zero cubic B-splines on t[l] <= x < t[l+1] (0-based l)
inline void bsplines(const double* t, double x, int l, double h[4]) {
    double hh[3];
    h[0] = 1.0;
    for (int j = 1; j <= K; ++j) {
        for (int i = 0; i < j; ++i) hh[i] = h[i];
        h[0] = 0.0;
        for (int i = 0; i < j; ++i) {
            const int li = l + 1 + i, lj = li - j;
            if (t[li] == t[lj]) {
                h[i + 1] = 0.0;
                continue;
            }
            const double f = hh[i] / (t[li] - t[lj]);
            h[i] = h[i] + f * (t[li] - x);
            h[i + 1] = f * (x - t[lj]);
        }
    }
}

// banded back substitution: rows of `a` hold the diagonal in column 0 and `bw-1` super-diagonals
void back_substitute(const std::vector<double>& a, int stride, const std::vector<double>& z, int n, int bw,
                     std::vector<double>& c) {
    c.assign(n, 0.0);
    c[n - 1] = z[n - 1] / a[(n - 1) * stride];
    for (int i = n - 2; i >= 0; --i) {
        double store = z[i];
        const int i1 = std::min(bw - 1, n - 1 - i);
        for (int l = 1; l <= i1; ++l) store = store - c[i + l] * a[i * stride + l];
        c[i] = store / a[i * stride];
    }
}

struct CurfitState {       // arrays FITPACK keeps between the first call and the iopt = 1 continuation
    std::vector<double> t, fpint, c;
    std::vector<int> nrdata;
    int n = 0;
    double fp = 0.0;
};

void add_knot(const double* x, CurfitState& st, int nrint) {        // fpknot
    auto& t = st.t;
    auto& fpint = st.fpint;
    auto& nrdata = st.nrdata;
    const int n = st.n;
    const int k = (n - nrint - 1) / 2;
    double fpmax = 0.0;
    int jbegin = 1, number = 0, maxpt = 0, maxbeg = 0;
    for (int j = 1; j <= nrint; ++j) {
        const int jpoint = nrdata[j - 1];
        if (!(fpmax >= fpint[j - 1] || jpoint == 0)) {
            fpmax = fpint[j - 1];
            number = j;
            maxpt = jpoint;
            maxbeg = jbegin;
        }
        jbegin = jbegin + jpoint + 1;
    }
    const int ihalf = maxpt / 2 + 1;
    const int nrx = maxbeg + ihalf;
    const int next = number + 1;
    for (int j = next; j <= nrint; ++j) {
        const int jj = next + nrint - j;
        fpint[jj] = fpint[jj - 1];
        nrdata[jj] = nrdata[jj - 1];
        const int jk = jj + k;
        t[jk] = t[jk - 1];
    }
    nrdata[number - 1] = ihalf - 1;
    nrdata[next - 1] = maxpt - ihalf;
    const double am = maxpt;
    double an = nrdata[number - 1];
    fpint[number - 1] = fpmax * an / am;
    an = nrdata[next - 1];
    fpint[next - 1] = fpmax * an / am;
    t[next + k - 1] = x[nrx - 1];
    st.n = n + 1;
}

// jumps of the third derivative of the B-splines at the interior knots, one row of 5 per knot (fpdisc)
void discontinuity_rows(const std::vector<double>& t, int n, std::vector<double>& b) {
    const int nk1 = n - K1;
    const int nrint = nk1 - K;
    const double fac = static_cast<double>(nrint) / (t[nk1] - t[K1 - 1]);
    b.assign(static_cast<size_t>(std::max(0, nk1 - K1)) * K2, 0.0);
    double h[2 * K1];
    for (int l = K2; l <= nk1; ++l) {
        for (int j = 1; j <= K1; ++j) {
            const int ik = j + K1, lj = l + j, lk = lj - K2;
            h[j - 1] = t[l - 1] - t[lk - 1];
            h[ik - 1] = t[l - 1] - t[lj - 1];
        }
        int lp = l - K1;
        for (int j = 1; j <= K2; ++j) {
            int jk = j;
            double prod = h[j - 1];
            for (int i = 0; i < K; ++i) {
                ++jk;
                prod = prod * h[jk - 1] * fac;
            }
            const int lk = lp + K1;
            b[static_cast<size_t>(l - K2) * K2 + (j - 1)] = (t[lk - 1] - t[lp - 1]) / prod;
            ++lp;
        }
    }
}

int fpcurf(int iopt, const double* x, const double* y, int m, double s, int nest, CurfitState& st, int ier_in) {
    const double xb = x[0], xe = x[m - 1];
    const int nmin = 2 * K1;
    const double acc = kTol * s;
    const int nmax = m + K1;
    auto& t = st.t;
    auto& fpint = st.fpint;
    auto& nrdata = st.nrdata;
    int ier = ier_in;
    double fp0 = 0.0, fpold = 0.0;
    int nplus = 0;

    auto interpolation_knots = [&]() {
        int i = K2, j = K / 2 + 2;
        for (int l = 0; l < m - K1; ++l) {
            t[i - 1] = x[j - 1];
            ++i;
            ++j;
        }
    };

    bool fresh = true;
    if (s <= 0.0) {
        st.n = nmax;
        if (nmax > nest) return 1;
        interpolation_knots();
        fresh = false;
    } else if (iopt != 0 && st.n != nmin) {
        fp0 = fpint[st.n - 1];
        fpold = fpint[st.n - 2];
        nplus = nrdata[st.n - 1];
        if (fp0 > s) fresh = false;
    }
    if (fresh) {
        st.n = nmin;
        fpold = 0.0;
        nplus = 0;
        nrdata[0] = m - 2;
    }

    std::vector<double> a, z, q(static_cast<size_t>(m) * K1), hs(static_cast<size_t>(m) * K1), yis(m);
    std::vector<int> first_row(m), started(m);
    int nk1 = 0;
    double fpms = 0.0;
    bool accepted = false;
    for (int iter = 0; iter < m; ++iter) {
        const int n = st.n;
        if (n == nmin) ier = -2;
        int nrint = n - nmin + 1;
        nk1 = n - K1;
        for (int j = 0; j < K1; ++j) {
            t[j] = xb;
            t[n - 1 - j] = xe;
        }
        // least-squares spline for the current knots: rotate each observation row into the band matrix
        double fp = 0.0;
        z.assign(nk1, 0.0);
        a.assign(static_cast<size_t>(nk1) * K1, 0.0);
        // (a) the knot interval and the four B-spline values of every observation: no observation waits for another here
        int l = K1;
        for (int it = 0; it < m; ++it) {
            const double xi = x[it];
            while (!(xi < t[l] || l == nk1)) ++l;
            first_row[it] = l - K1;
            bsplines(t.data(), xi, l - 1, &q[static_cast<size_t>(it) * K1]);
        }
        // (b) FITPACK rotates observation after observation into the band matrix, four Givens steps each (band rows first_row ..
        // first_row + 3), every step waiting for the one before it: a division, a square root, two divisions - some 20 ns of latency
        // and next to no work.  Observation it + 1 needs a band row only after observation `it` has left it, so it may run
        // 1 + (first_row[it + 1] - first_row[it]) steps behind: up to four observations are under way at a time, oldest first within
        // a step - the same operations on the same operands as fpcurf's loop, in an order the core can overlap (bit-identical by
        // construction; pinned by f4_fitpack and f14_*).
        {
            int head = 0, tail = 0, next_start = 0;
            for (int s = 0; head < m; ++s) {
                if (tail < m && next_start <= s) {
                    for (int i = 0; i < K1; ++i) hs[static_cast<size_t>(tail) * K1 + i] = q[static_cast<size_t>(tail) * K1 + i];
                    yis[tail] = y[tail];
                    started[tail] = s;
                    if (tail + 1 < m) next_start = s + 1 + (first_row[tail + 1] - first_row[tail]);
                    ++tail;
                }
                for (int ob = head; ob < tail; ++ob) {
                    const int i = s - started[ob];
                    double* h = &hs[static_cast<size_t>(ob) * K1];
                    const double piv = h[i];
                    if (piv != 0.0) {
                        const int j = first_row[ob] + i;
                        double* row = &a[static_cast<size_t>(j) * K1];
                        const Rot r = givens(piv, row[0]);
                        rotate(r, yis[ob], z[j]);
                        for (int i1 = i + 1; i1 < K1; ++i1) rotate(r, h[i1], row[i1 - i]);
                    }
                }
                if (s - started[head] == K1 - 1) ++head;      // the oldest one has taken its last step
            }
        }
        for (int it = 0; it < m; ++it) fp = fp + yis[it] * yis[it];
        if (ier == -2) fp0 = fp;
        fpint[n - 1] = fp0;
        fpint[n - 2] = fpold;
        nrdata[n - 1] = nplus;
        back_substitute(a, K1, z, nk1, K1, st.c);
        st.fp = fp;
        fpms = fp - s;
        if (std::fabs(fpms) < acc) return ier;
        if (fpms < 0.0) {
            accepted = true;
            break;
        }
        if (n == nmax) return -1;
        if (n == nest) return 1;
        if (ier == 0) {
            int npl1 = nplus * 2;
            const double rn = nplus;
            if (fpold - fp > acc) npl1 = static_cast<int>(rn * fpms / (fpold - fp));
            nplus = std::min(nplus * 2, std::max(std::max(npl1, nplus / 2), 1));
        } else {
            nplus = 1;
            ier = 0;
        }
        fpold = fp;
        // share of the residual sum of squares that falls into every knot interval
        double fpart = 0.0;
        int i = 1;
        l = K2;
        bool crossed = false;
        for (int it = 0; it < m; ++it) {
            if (!(x[it] < t[l - 1] || l > nk1)) {
                crossed = true;
                ++l;
            }
            double term = 0.0;
            int l0 = l - K2;
            for (int j = 0; j < K1; ++j) {
                ++l0;
                term = term + st.c[l0 - 1] * q[static_cast<size_t>(it) * K1 + j];
            }
            term = (term - y[it]) * (term - y[it]);
            fpart = fpart + term;
            if (crossed) {
                const double store = term * 0.5;
                fpint[i - 1] = fpart - store;
                ++i;
                fpart = store;
                crossed = false;
            }
        }
        fpint[nrint - 1] = fpart;
        for (int rep = 0; rep < nplus; ++rep) {
            add_knot(x, st, nrint);
            ++nrint;
            if (st.n == nmax) {
                interpolation_knots();
                break;
            }
            if (st.n == nest) break;
        }
    }
    if (!accepted) return ier;
    if (ier == -2) return ier;

    // ---- part 2: find the smoothing parameter p with F(p) = s ------------------------------------
    const int n = st.n;
    std::vector<double> b;
    discontinuity_rows(t, n, b);
    double p1 = 0.0, f1 = fp0 - s;
    double p3 = -1.0, f3 = fpms;
    double p = 0.0;
    for (int i = 0; i < nk1; ++i) p = p + a[static_cast<size_t>(i) * K1];
    p = static_cast<double>(nk1) / p;
    int ich1 = 0, ich3 = 0;
    const int n8 = n - nmin;
    std::vector<double> g(static_cast<size_t>(nk1) * K2), c;
    for (int iter = 1; iter <= kMaxIt; ++iter) {
        const double pinv = 1.0 / p;
        c = z;
        for (int i = 0; i < nk1; ++i) {
            for (int j = 0; j < K1; ++j) g[static_cast<size_t>(i) * K2 + j] = a[static_cast<size_t>(i) * K1 + j];
            g[static_cast<size_t>(i) * K2 + K1] = 0.0;
        }
        // Row `it` of the discontinuity matrix is rotated into band rows it, it + 1, ..., nk1 - one Givens step per band row, each
        // waiting for the one before it (a division, a square root, two divisions: ~25 ns of latency and next to no work).  Row
        // it + 1 needs band row j only after row `it` has left it, so kLanes rows go down the band together, each two band rows
        // behind the one before it: the same operations on the same operands as FITPACK's loop nest (fpcurf's statement 260..300),
        // in an order the core can overlap.  Bit-identical by construction; pinned by f4_fitpack / f14_*.
        for (int it0 = 1; it0 <= n8; it0 += kLanes) {
            double h[kLanes][K2 + 1];
            double yi[kLanes];
            const int lanes = std::min(kLanes, n8 - it0 + 1);
            for (int r = 0; r < lanes; ++r) {
                for (int i = 0; i < K2; ++i) h[r][i] = b[static_cast<size_t>(it0 + r - 1) * K2 + i] * pinv;
                h[r][K2] = 0.0;
                yi[r] = 0.0;
            }
            const int steps = (nk1 - it0) + (lanes - 1) + 1;       // lane r takes band row it0 + s - r at step s, from s = 2 r on
            for (int s = 0; s < steps; ++s) {
#pragma GCC unroll 4
                for (int r = 0; r < kLanes; ++r) {
                    const int j = it0 + s - r;
                    if (r >= lanes || s < 2 * r || j > nk1) continue;
                    double* row = &g[static_cast<size_t>(j - 1) * K2];
                    const Rot rot = givens(h[r][0], row[0]);
                    rotate(rot, yi[r], c[j - 1]);
                    if (j == nk1) continue;
                    const int i2 = (j > n8) ? nk1 - j : K1;
                    for (int i = 1; i <= i2; ++i) {
                        rotate(rot, h[r][i], row[i]);
                        h[r][i - 1] = h[r][i];
                    }
                    h[r][i2] = 0.0;
                }
            }
        }
        std::vector<double> sol;
        back_substitute(g, K2, c, nk1, K2, sol);
        double fp = 0.0;
        int l = K2;
        for (int it = 0; it < m; ++it) {
            if (!(x[it] < t[l - 1] || l > nk1)) ++l;
            int l0 = l - K2;
            double term = 0.0;
            for (int j = 0; j < K1; ++j) {
                ++l0;
                term = term + sol[l0 - 1] * q[static_cast<size_t>(it) * K1 + j];
            }
            fp = fp + (term - y[it]) * (term - y[it]);
        }
        st.c = sol;
        st.fp = fp;
        fpms = fp - s;
        if (std::fabs(fpms) < acc) return ier;
        if (iter == kMaxIt) return 3;
        const double p2 = p, f2 = fpms;
        if (ich3 == 0) {
            if ((f2 - f3) <= acc) {            // initial p too large
                p3 = p2;
                f3 = f2;
                p = p * 0.04;
                if (p <= p1) p = p1 * 0.9 + p2 * 0.1;
                continue;
            }
            if (f2 < 0.0) ich3 = 1;
        }
        if (ich1 == 0) {
            if ((f1 - f2) <= acc) {            // initial p too small
                p1 = p2;
                f1 = f2;
                p = p / 0.04;
                if (p3 < 0.0) continue;
                if (p >= p3) p = p2 * 0.1 + p3 * 0.9;
                continue;
            }
            if (f2 > 0.0) ich1 = 1;
        }
        if (f2 >= f1 || f2 <= f3) return 2;
        // rational interpolation through (p1,f1), (p2,f2), (p3,f3); p3 < 0 encodes p3 = infinity
        if (p3 > 0.0) {
            const double h1 = f1 * (f2 - f3), h2 = f2 * (f3 - f1), h3 = f3 * (f1 - f2);
            p = -(p1 * p2 * h3 + p2 * p3 * h1 + p3 * p1 * h2) / (p1 * h1 + p2 * h2 + p3 * h3);
        } else {
            p = (p1 * (f1 - f3) * f2 - p2 * (f2 - f3) * f1) / ((f1 - f2) * f3);
        }
        if (f2 < 0.0) {
            p3 = p2;
            f3 = f2;
        } else {
            p1 = p2;
            f1 = f2;
        }
    }
    return 3;
}

}  // namespace

int spline_fit(const double* x, const double* y, int m, double s, Spline& out) {
    if (m <= K) return FHX_ERR_ARG;
    for (int i = 1; i < m; ++i)
        if (!(x[i] > x[i - 1])) return FHX_ERR_ARG;
    CurfitState st;
    const size_t cap = static_cast<size_t>(m) + K1 + 2;
    st.t.assign(cap, 0.0);
    st.fpint.assign(cap, 0.0);
    st.nrdata.assign(cap, 0);
    const int nest = (s == 0.0) ? m + K1 : std::max(m / 2, 2 * K1);
    int ier = fpcurf(0, x, y, m, s, nest, st, 0);
    out.restarted = false;
    if (ier == 1) {                       // "nest too small": continue with room for interpolation
        out.restarted = true;
        ier = fpcurf(1, x, y, m, s, m + K1, st, 1);
    }
    out.t.assign(st.t.begin(), st.t.begin() + st.n);
    out.c.assign(st.c.begin(), st.c.begin() + (st.n - K1));
    out.fp = st.fp;
    out.ier = ier;
    return FHX_OK;
}

void spline_eval(const Spline& sp, const double* xs, int64_t nx, double* out) {
    const int n = static_cast<int>(sp.t.size());
    const int nk1 = n - K1;
    const double* t = sp.t.data();
    int l = K1;
    for (int64_t i = 0; i < nx; ++i) {
        const double arg = xs[i];
        while (!(arg >= t[l - 1] || l == K1)) --l;
        while (!(arg < t[l] || l == nk1)) ++l;
        double h[4];
        bsplines(t, arg, l - 1, h);
        double v = 0.0;
        const int ll = l - K1;
        for (int j = 0; j < K1; ++j) v = v + sp.c[ll + j] * h[j];
        out[i] = v;
    }
}

// ===================================================================================================
// Pool-adjacent-violators, antitonic, unit weights (Busing 2022 as scipy.optimize implements it,
// run on the reversed sequence: that is what increasing=False does).
// ===================================================================================================
void pava_decreasing(const double* y, int64_t n, double* out) {
    if (n <= 0) return;
    std::vector<double> x(n), w(n, 1.0);
    std::vector<int64_t> r(n + 1, 0);
    for (int64_t i = 0; i < n; ++i) x[i] = y[n - 1 - i];
    r[0] = 0;
    if (n > 0) r[1] = 1;
    int64_t b = 0;
    double xb_prev = x[0], wb_prev = w[0];
    for (int64_t i = 1; i < n; ++i) {
        ++b;
        double xb = x[i], wb = w[i];
        if (xb_prev >= xb) {
            --b;
            double sb = wb_prev * xb_prev + wb * xb;
            wb += wb_prev;
            xb = sb / wb;
            while (i < n - 1 && xb >= x[i + 1]) {
                ++i;
                sb += w[i] * x[i];
                wb += w[i];
                xb = sb / wb;
            }
            while (b > 0 && x[b - 1] >= xb) {
                --b;
                sb += w[b] * x[b];
                wb += w[b];
                xb = sb / wb;
            }
        }
        x[b] = xb_prev = xb;
        w[b] = wb_prev = wb;
        r[b + 1] = i + 1;
    }
    int64_t f = n - 1;
    for (int64_t k = b; k >= 0; --k) {
        const int64_t t = r[k];
        const double xk = x[k];
        for (int64_t i = f; i >= t; --i) x[i] = xk;
        f = t - 1;
    }
    for (int64_t i = 0; i < n; ++i) out[i] = x[n - 1 - i];
}

// ===================================================================================================
// Stage logic of one pass
// ===================================================================================================
namespace {

// numpy's pairwise summation (np.sum of a contiguous double array), used for `residual` only
double numpy_pairwise(const double* a, int64_t n) {
    if (n < 8) {
        double r = -0.0;
        for (int64_t i = 0; i < n; ++i) r += a[i];
        return r;
    }
    if (n <= 128) {
        double r[8];
        for (int j = 0; j < 8; ++j) r[j] = a[j];
        int64_t i = 8;
        for (; i < n - (n % 8); i += 8)
            for (int j = 0; j < 8; ++j) r[j] += a[i + j];
        double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
        for (; i < n; ++i) res += a[i];
        return res;
    }
    int64_t n2 = n / 2;
    n2 -= n2 % 8;
    return numpy_pairwise(a, n2) + numpy_pairwise(a + n2, n - n2);
}

// np.sum of a contiguous double array: the reduction runs over buffers of 8192 elements (numpy's default buffer size), each
// summed pairwise, the buffer sums added from left to right (checked against numpy 2.2 up to 10^6 elements; a single pairwise
// pass over the whole array differs in the last bit for about half of the arrays beyond 8192 elements)
double numpy_sum(const double* a, int64_t n) {
    const int64_t B = 8192;
    if (n <= B) return numpy_pairwise(a, n);
    double acc = numpy_pairwise(a, B);
    for (int64_t i = B; i < n; i += B) acc += numpy_pairwise(a + i, std::min<int64_t>(B, n - i));
    return acc;
}

inline bool dist_in_range(int64_t d, int64_t lo, int64_t hi) { return d >= lo && d <= hi; }

// forward-only bin cursor that sticks at the last bin (fithic.py:535-546 and :627-638)
inline size_t advance_cursor(const std::vector<Bin>& bins, size_t cur, int64_t d) {
    while (!(bins[cur].lb <= d && d <= bins[cur].ub)) {
        ++cur;
        if (cur >= bins.size()) {
            --cur;
            break;
        }
    }
    return cur;
}

}  // namespace

void make_bins_stage(const PassInputs& in, PassFit& out) {
    out = PassFit();
    const int64_t res = in.resolution;
    // ---- makeBinsFromInteractions (fithic.py:463-553) --------------------------------------------
    // keys of mainDic = distances with at least one observed in-range row, ascending
    {
        int64_t termination = 0, so_far = 0;
        int n = 0;
        // desiredPerBin starts as int/int true division (fithic.py:476)
        double desired = static_cast<double>(in.in_range_sum) / static_cast<double>(in.n_bins);
        Bin cur;
        bool open = false;
        int64_t prev_ub = -1;
        for (int64_t i = 0; i < in.n_dist; ++i) {
            if (in.hist_npairs[i] <= 0) continue;
            const int64_t d = in.dist_keys ? in.dist_keys[i] : i * res;
            const int64_t cc = in.hist_sumcc[i];
            so_far += cc;
            bool full = false;
            if (static_cast<double>(cc) >= desired) {
                termination = 0;
                full = true;
            } else if (static_cast<double>(termination + cc) >= desired) {
                termination = 0;
                full = true;
            } else {
                termination += cc;
            }
            if (!open) {
                cur = Bin();
                open = true;
            }
            cur.sumcc += cc;
            cur.ub = d;
            if (full) {
                ++n;
                if (n < in.n_bins)
                    desired = 1.0 * static_cast<double>(in.in_range_sum - so_far) / static_cast<double>(in.n_bins - n);
                cur.lb = out.bins.empty() ? 0 : prev_ub + 1;
                prev_ub = cur.ub;
                out.bins.push_back(cur);
                termination = 0;
                open = false;
            }
        }
        // distances that never filled a bin are dropped, exactly as the reference drops them (A10)
    }
    if (in.outlier_dists != nullptr && !out.bins.empty()) {          // -r 0: explicit ascending list
        size_t cur = 0;
        for (int64_t i = 0; i < in.n_outlier_dists; ++i) {
            cur = advance_cursor(out.bins, cur, in.outlier_dists[i]);
            out.bins[cur].poss7 -= 1;
            out.bins[cur].poss -= 1;
        }
    } else if (in.outlier_dist_hist != nullptr && !out.bins.empty()) {
        size_t cur = 0;
        for (int64_t i = 0; i < in.n_dist; ++i) {
            const int64_t mult = in.outlier_dist_hist[i];
            if (mult <= 0) continue;
            cur = advance_cursor(out.bins, cur, i * res);
            out.bins[cur].poss7 -= mult;
            out.bins[cur].poss -= mult;
        }
    }
    for (auto& b : out.bins) b.poss0 = b.poss;
}


int run_host_pass(const PassInputs& in, const FragTable& frags, PassFit& out, std::string& err) {
    static const bool stage_times = std::getenv("FHX_FIT_TIMES") != nullptr;      // measurements
    double tm[8] = {0};
    int tk = 0;
    auto tlast = std::chrono::steady_clock::now();
    auto T = [&]() { const auto n = std::chrono::steady_clock::now(); if (tk < 8) tm[tk++] = std::chrono::duration<double, std::micro>(n - tlast).count(); tlast = n; };
    make_bins_stage(in, out);
    T();
    const int64_t res = in.resolution;

    // ---- generate_FragPairs ---------------------------------------------------------------------------
    if (res == 0) {
        // non-fixed-size branch (fithic.py:691-778): every in-range pair of mappable fragments in (x, y) order, bin cursor
        // restarted for every x, npairs = n - (#in-range y seen so far for this x) weights slots [7] and [3], slot [1]
        // counts pairs; slot [3] is a sequential double sum in visiting order.  Chromosomes without mappable fragments are
        // skipped here (no error in this branch).
        //
        // The reference visits all pairs from one thread (O(n * window) Python iterations: hours on a restriction-fragment
        // genome).  What it computes per bin separates cleanly:
        //   * slot [1], slot [7] and the counters are integers: for one x the y of a bin are a contiguous run (the mids are
        //     sorted, distances ascend in y, the cursor only moves forward), so a run contributes its length and an arithmetic
        //     series of npairs - no pair is touched;
        //   * slot [3] is a double accumulated pair by pair: every bin's sum is its own sequential chain over the bin's pairs in
        //     (chromosome, x, y) order.  The chains of different bins are independent, so they run on different host threads
        //     (largest distances first: those bins hold the most possible pairs), each with the reference's rounding sequence.
        // The bin of a distance does not depend on the cursor's history: bins are contiguous from 0 and distances ascend, so it
        // is the first bin whose upper end is >= the distance, or the last bin (the cursor sticks there, fithic.py:723-735).
        int64_t n_frags = 0;
        for (const auto& m : frags.mids) n_frags += (int64_t)m.size();
        int64_t poss_in_range = 0, poss_inter2 = 0, poss_intra_all = 0;
        double max_possible = 0.0;
        const int64_t lo_i = in.dist_low;
        const bool bounded = in.dist_up != INT64_MAX;
        const int64_t hi_i = in.dist_up;
        const int64_t nb = (int64_t)out.bins.size();
        // integer mids: |float(a) - float(b)| is the exact integer distance for |mid| < 2^53, and its comparisons with the
        // integer thresholds / bin ends are exact too
        for (const auto& m : frags.mids) {                       // already in sorted(name) order, each sorted ascending
            const int64_t n = (int64_t)m.size();
            if (n == 0) continue;
            poss_inter2 += (n_frags - n) * n;
            int64_t y0 = 0, yend = 0;                            // first y with dist >= lo; first y with dist > hi
            for (int64_t x = 0; x < n; ++x) {
                const int64_t mx = m[x];
                if (y0 < x + 1) y0 = x + 1;
                while (y0 < n && (int64_t)m[y0] - mx < lo_i) ++y0;
                if (yend < y0) yend = y0;
                if (bounded)
                    while (yend < n && (int64_t)m[yend] - mx <= hi_i) ++yend;
                else
                    yend = n;
                if (yend > y0) {
                    poss_in_range += yend - y0;
                    max_possible = std::max(max_possible, static_cast<double>((int64_t)m[yend - 1] - mx));
                }
            }
        }
        NfPairSums given;
        const bool have_given = nb > 0 && in.nf_pairs && in.nf_pairs(out.bins, given) && (int64_t)given.poss.size() == nb &&
                                (int64_t)given.poss7.size() == nb && (int64_t)given.sumdist.size() == nb;
        if (have_given) {                                        // the caller walked the pairs (fhx_fit: on the GPU)
            for (int64_t b = 0; b < nb; ++b) {
                out.bins[(size_t)b].poss7 += given.poss7[(size_t)b];
                out.bins[(size_t)b].poss += given.poss[(size_t)b];
                out.bins[(size_t)b].sumdist = given.sumdist[(size_t)b];
                poss_intra_all += given.poss[(size_t)b];
            }
        } else if (nb > 0) {
            struct BinSums {
                int64_t poss7 = 0, poss = 0;
                double sumdist = 0.0;
            };
            std::vector<BinSums> acc((size_t)nb);
            std::atomic<int64_t> next{nb - 1};
            constexpr int64_t TERM_BLOCK = 256;
            auto work = [&]() {
                for (;;) {
                    const int64_t b = next.fetch_sub(1);
                    if (b < 0) return;
                    const bool last = b == nb - 1;
                    const int64_t below = b == 0 ? -1 : out.bins[(size_t)b - 1].ub;      // the bin takes below < dist <= ub
                    const int64_t ub = out.bins[(size_t)b].ub;
                    BinSums A;
                    A.sumdist = out.bins[(size_t)b].sumdist;                               // 0.0: the chain starts here
                    for (const auto& m : frags.mids) {
                        const int64_t n = (int64_t)m.size();
                        int64_t y0 = 0, yend = 0, ya = 0, yb = 0;
                        for (int64_t x = 0; x < n; ++x) {
                            const int64_t mx = m[x];
                            if (y0 < x + 1) y0 = x + 1;
                            while (y0 < n && (int64_t)m[y0] - mx < lo_i) ++y0;
                            if (yend < y0) yend = y0;
                            if (bounded)
                                while (yend < n && (int64_t)m[yend] - mx <= hi_i) ++yend;
                            else
                                yend = n;
                            if (ya < y0) ya = y0;
                            while (ya < yend && (int64_t)m[ya] - mx <= below) ++ya;
                            if (ya > yend) ya = yend;
                            if (last) {
                                yb = yend;
                            } else {
                                if (yb < ya) yb = ya;
                                while (yb < yend && (int64_t)m[yb] - mx <= ub) ++yb;
                                if (yb > yend) yb = yend;
                            }
                            const int64_t cnt = yb - ya;
                            if (cnt <= 0) continue;
                            // npairs of pair (x, y) = n - (y - y0): the k-th in-range y of this x weighs n - k (fithic.py:714-715)
                            const int64_t k0 = ya - y0, k1 = yb - 1 - y0;
                            A.poss += cnt;
                            A.poss7 += cnt * n - (k0 + k1) * cnt / 2;
                            // terms (dist / 1e6) * npairs of a block first - independent of each other: the divisions
                            // pipeline (and vectorise) - then the block is added in visiting order, one rounded add at a time
                            const int32_t* my = m.data();
                            for (int64_t y = ya; y < yb; y += TERM_BLOCK) {
                                const int64_t len = std::min<int64_t>(TERM_BLOCK, yb - y);
                                double term[TERM_BLOCK];
                                for (int64_t j = 0; j < len; ++j)
                                    term[j] = (static_cast<double>((int64_t)my[y + j] - mx) / 1000000.0) *
                                              static_cast<double>(n - (y + j - y0));
                                double acc_s = A.sumdist;
                                for (int64_t j = 0; j < len; ++j) acc_s += term[j];
                                A.sumdist = acc_s;
                            }
                        }
                    }
                    acc[(size_t)b] = A;
                }
            };
            const int nt = (int)std::max<int64_t>(1, std::min<int64_t>(nb, (int64_t)fhx::usable_cpus()));
            std::vector<std::thread> pool;
            for (int t = 1; t < nt; ++t) pool.emplace_back(work);
            work();
            for (auto& th : pool) th.join();
            for (int64_t b = 0; b < nb; ++b) {
                out.bins[(size_t)b].poss7 += acc[(size_t)b].poss7;
                out.bins[(size_t)b].poss += acc[(size_t)b].poss;
                out.bins[(size_t)b].sumdist = acc[(size_t)b].sumdist;
                poss_intra_all += acc[(size_t)b].poss;
            }
        }
        out.n_frags = n_frags;
        out.max_possible_dist = max_possible;
        out.poss_intra_in_range = poss_in_range;
        out.poss_inter_all = static_cast<double>(poss_inter2) / 2;
        out.poss_intra_all = static_cast<double>(poss_intra_all);
        out.inter_chr_prob = in.inter_count > 0 ? 1.0 / static_cast<double>(in.inter_count) : 0.0;
        out.baseline_intra_prob = poss_intra_all > 0 ? 1.0 / static_cast<double>(poss_intra_all) : 0.0;
    } else {
        // fixed-size branch (fithic.py:592-689)
        int64_t n_frags = 0;
        double max_possible = 0.0;
        for (size_t c = 0; c < frags.n_mappable.size(); ++c) {
            if (frags.n_mappable[c] == 0) {
                err = "a chromosome in the fragments file has no mappable fragment; the reference raises TypeError "
                      "at fithic.py:600";
                return FHX_ERR_REFERENCE_EXIT;
            }
            n_frags += frags.n_mappable[c];
            const double mf = static_cast<double>(frags.max_mid[c]) - static_cast<double>(res) / 2;
            max_possible = std::max(max_possible, mf);
        }
        int64_t poss_in_range = 0;
        // Python ints are exact; n*(noOfFrags-n) stays below 2^63 for any realistic genome
        int64_t poss_inter2 = 0;
        double poss_intra_all = 0.0;
        for (size_t c = 0; c < frags.n_mappable.size(); ++c) {   // already in sorted(name) order
            const int64_t n = frags.n_mappable[c];
            const double max_frag = static_cast<double>(frags.max_mid[c]) - static_cast<double>(res) / 2;
            const int64_t stop = static_cast<int64_t>(max_frag + 1);      // int(maxFrag+1): truncation
            int64_t k = 0;
            size_t cur = 0;
            int64_t per_chr = 0;
            // the reference walks every distance 0, res, 2 res, ... < stop and tests the range each time; only the in-range
            // ones do anything, and k is just d / res, so start at the first in-range multiple and stop after the last
            // (same statements in the same order; O(#in-range distances) instead of O(loci) per chromosome and pass)
            int64_t d_first = in.dist_low <= 0 ? 0 : ((in.dist_low + res - 1) / res) * res;
            const int64_t d_stop = in.dist_up == INT64_MAX ? stop : std::min<int64_t>(stop, in.dist_up + 1);
            k = d_first / res;
            for (int64_t d = d_first; d < d_stop; d += res) {
                const int64_t npairs = n - k;
                ++k;
                if (!dist_in_range(d, in.dist_low, in.dist_up)) continue;
                per_chr += npairs;
                if (!out.bins.empty()) {
                    cur = advance_cursor(out.bins, cur, d);
                    Bin& b = out.bins[cur];
                    b.poss7 += npairs;
                    b.poss += npairs;
                    b.sumdist += (static_cast<double>(d) / 1000000.0) * static_cast<double>(npairs);
                    per_chr += npairs;                               // the reference counts twice (A5)
                }
            }
            poss_inter2 += n * (n_frags - n);
            poss_intra_all += static_cast<double>(n * (n + 1)) / 2;   // Python: (n*(n+1))/2 true division
            poss_in_range += per_chr;
        }
        out.n_frags = n_frags;
        out.max_possible_dist = max_possible;
        out.poss_intra_in_range = poss_in_range;
        out.poss_inter_all = static_cast<double>(poss_inter2) / 2;
        out.poss_intra_all = poss_intra_all;
        out.inter_chr_prob = in.inter_count > 0 ? 1.0 / static_cast<double>(in.inter_count) : 0.0;
        out.baseline_intra_prob = poss_intra_all > 0 ? 1.0 / poss_intra_all : 0.0;
    }

    T();
    // ---- calculateProbabilities (fithic.py:869-908) -------------------------------------------------
    out.x.clear();
    out.y.clear();
    for (const auto& b : out.bins) {
        double avg_cc = 0.0;
        if (b.poss > 0 && in.in_range_sum > 0)
            avg_cc = (1.0 * static_cast<double>(b.sumcc) / static_cast<double>(b.poss)) / static_cast<double>(in.in_range_sum);
        double avg_dist = 0.0;
        if (b.poss7 != 0) avg_dist = 1000000.0 * (b.sumdist / static_cast<double>(b.poss7));
        out.x.push_back(avg_dist);
        out.y.push_back(avg_cc);
    }

    // ---- BH denominator for this mode (fithic.py:1126-1163) -----------------------------------------
    if (in.mode == FHX_MODE_ALL)
        out.bh_total_tests = static_cast<double>(out.poss_intra_in_range + in.inter_count);
    else if (in.mode == FHX_MODE_INTER_ONLY)
        out.bh_total_tests = static_cast<double>(in.inter_count);
    else
        out.bh_total_tests = static_cast<double>(out.poss_intra_in_range);
    if (out.bh_total_tests == 0.0) {
        err = "the number of tests N is zero; the reference raises ZeroDivisionError at fithic.py:1138-1162";
        return FHX_ERR_REFERENCE_EXIT;
    }
    out.prior_lut.assign(static_cast<size_t>(std::max<int64_t>(in.n_dist, 1)), 1.0);
    if (in.mode == FHX_MODE_INTER_ONLY) return FHX_OK;      // no spline in interOnly mode (fithic.py:936)

    // ---- fit_Spline, fit + table (fithic.py:936-968) ---------------------------------------------
    const int m = static_cast<int>(out.x.size());
    std::vector<int> order(m);
    for (int i = 0; i < m; ++i) order[i] = i;
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return out.x[a] < out.x[b]; });
    std::vector<double> xs(m), ys(m);
    for (int i = 0; i < m; ++i) {
        xs[i] = out.x[order[i]];
        ys[i] = out.y[order[i]];
    }
    for (int i = 1; i < m; ++i) {
        if (xs[i] <= xs[i - 1]) {
            err = "ERROR in spline fitting. Distances do not decrease across bins (the reference exits with status 2 "
                  "at fithic.py:941-945)";
            return FHX_ERR_REFERENCE_EXIT;
        }
    }
    if (m <= K) {
        err = "fewer than 4 bins: scipy's UnivariateSpline raises (m > k must hold), fithic.py:951";
        return FHX_ERR_REFERENCE_EXIT;
    }
    const double ymin = *std::min_element(ys.begin(), ys.end());
    out.spline_s = ymin * ymin;                 // splineError = min(y)*min(y) (fithic.py:948): the product, not pow()
    if (spline_fit(xs.data(), ys.data(), m, out.spline_s, out.spline) != FHX_OK) {
        err = "spline fit rejected its input";
        return FHX_ERR_REFERENCE_EXIT;
    }
    T();
    out.min_x = xs.front();
    out.max_x = xs.back();
    // splineX: observed distances d with min(x) <= d <= max(x) (int vs float comparison is exact here)
    std::vector<double> tx;
    for (int64_t i = 0; i < in.n_dist; ++i) {
        if (in.hist_npairs[i] <= 0) continue;
        const int64_t key = in.dist_keys ? in.dist_keys[i] : i * res;
        const double d = static_cast<double>(key);
        if (out.min_x <= d && d <= out.max_x) {
            out.table_x.push_back(key);
            tx.push_back(d);
        }
    }
    if (tx.empty()) {
        err = "no observed distance lies inside [min(x), max(x)]: the reference fails on an empty spline table "
              "(fithic.py:961-966)";
        return FHX_ERR_REFERENCE_EXIT;
    }
    out.table_y0.resize(tx.size());
    spline_eval(out.spline, tx.data(), static_cast<int64_t>(tx.size()), out.table_y0.data());
    T();
    out.table_y.resize(tx.size());
    pava_decreasing(out.table_y0.data(), static_cast<int64_t>(tx.size()), out.table_y.data());
    {
        std::vector<double> fitted(m), sq(m);
        spline_eval(out.spline, xs.data(), m, fitted.data());
        for (int i = 0; i < m; ++i) {
            const double r = ys[i] - fitted[i];
            sq[i] = r * r;
        }
        out.residual = numpy_sum(sq.data(), m);
    }
    T();
    // dense LUT over distance indices: clamp, bisect_left, cap (fithic.py:1066-1069); with explicit distance keys (-r 0,
    // or -r N on off-grid loci) the device does the same search per row
    if (res > 0 && !in.dist_keys) {
        const size_t nt = tx.size();
        size_t pos = 0;                                       // bisect_left is monotone in the clamped distance
        for (int64_t i = 0; i < in.n_dist; ++i) {
            double look = static_cast<double>(i * res);
            if (look < out.min_x) look = out.min_x;           // max(d, min(x))
            if (look > out.max_x) look = out.max_x;           // min(., max(x))
            while (pos < nt && tx[pos] < look) ++pos;
            out.prior_lut[static_cast<size_t>(i)] = out.table_y[std::min(pos, nt - 1)];
        }
    }
    if (stage_times) { T(); std::fprintf(stderr, "run_host_pass: bins %.1f us, possible pairs %.1f, spline fit %.1f, table eval %.1f, isotonic + residual %.1f, lut %.1f\n", tm[0], tm[1], tm[2], tm[3], tm[4], tm[5]); }
    return FHX_OK;
}

}  // namespace fhx
