// CPU check of fithic_amd/csrc/fhx_fmt.hpp (the formatting the GPU writer runs) against the C library: "%e" and "%f" of
// random and adversarial doubles must be the same characters.  Usage: fmt_check <n_random> ; exit status 0 = all equal.
#include <charconv>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>

#include "../../fithic_amd/csrc/fhx_fmt.hpp"

static long bad = 0, checked = 0, unsupported = 0;

static void check(double v) {
    char a[512], b[512];
    ++checked;
    int nb = fhx::fmt::fmt_e6(v, b);
    if (nb < 0) {
        ++unsupported;
    } else {
        int na = std::isfinite(v) ? snprintf(a, sizeof a, "%e", v) : snprintf(a, sizeof a, "%s", std::isnan(v) ? "nan" : (v > 0 ? "inf" : "-inf"));
        if (na != nb || memcmp(a, b, na)) {
            if (bad < 10) printf("E %.17g: want '%s' got '%.*s'\n", v, a, nb, b);
            ++bad;
        }
    }
    nb = fhx::fmt::fmt_f6(v, b);
    if (nb < 0) {
        ++unsupported;
    } else {
        int na = std::isfinite(v) ? snprintf(a, sizeof a, "%f", v) : snprintf(a, sizeof a, "%s", std::isnan(v) ? "nan" : (v > 0 ? "inf" : "-inf"));
        if (na != nb || memcmp(a, b, na)) {
            if (bad < 10) printf("F %.17g: want '%s' got '%.*s'\n", v, a, nb, b);
            ++bad;
        }
    }
}

int main(int argc, char** argv) {
    const long n = argc > 1 ? atol(argv[1]) : 1000000;
    std::mt19937_64 g(20260928);
    // edge cases
    const double edges[] = {0.0, -0.0, 1.0, -1.0, 0.5, 9.9999995, 9.9999994999999, 9.99999950000001, 0.9999995, 0.99999949999, 1e-7, 5e-7, 4.9999999e-7,
                            5.0000001e-7, 1.5e-6, 2.5e-6, 0.0000005, 0.0000015, 0.0000025, 123456.5, 1234567.5, 12345678.5, 1e15, 1e16, 9007199254740991.0, 9007199254740992.0,
                            4.9e-324, 2.2250738585072014e-308, 2.2250738585072009e-308, 1.7976931348623157e308, 1e22, 1e23, 1.8446744073709552e19,
                            1.8446744073709550e19, 9.2233720368547758e18, 9.2233720368547748e18, 1e-300, 1e-310, 3.1e-318, 0.30000000000000004, 2.5, 3.5, 0.125, 0.0625,
                            1000000.0, 999999.5, 9999999.5, 99999.95, 1e6 - 1e-7, 0.1 + 0.2, 1.0 / 3.0, 2.0 / 3.0, NAN, INFINITY, -INFINITY};
    for (double v : edges) {
        check(v);
        check(-v);
        check(std::nextafter(v, INFINITY));
        check(std::nextafter(v, -INFINITY));
    }
    // exact decimal ties: (2k+1) / 2^j scaled so that the 7th / the 6th decimal digit sits on a half
    for (int j = 1; j < 40; ++j)
        for (int k = 0; k < 2000; ++k) {
            check(std::ldexp((double)(2 * k + 1), -j));
            check(std::ldexp((double)(2 * k + 1), -j) * 1e-6);
            check((double)(2 * k + 1) * 0.5e-6);
            check((double)k + 0.5);
        }
    for (int e = -330; e <= 25; ++e) {
        const double p = std::pow(10.0, e);
        for (int d = -3; d <= 3; ++d) {
            double v = p;
            for (int t = 0; t < (d < 0 ? -d : d); ++t) v = std::nextafter(v, d < 0 ? 0.0 : INFINITY);
            check(v);
            check(v * 9.9999995);
            check(v * 1.0000005);
        }
    }
    // random: all bit patterns, p-value-like, bias-like, ExpCC-like
    std::uniform_real_distribution<double> U(0.0, 1.0);
    for (long i = 0; i < n; ++i) {
        unsigned long long bits = g();
        double v;
        memcpy(&v, &bits, 8);
        check(v);
        check(std::pow(U(g), 1.0 + 40.0 * U(g)));
        check(std::exp(0.25 * (U(g) + U(g) + U(g) - 1.5) * 3.0));
        check(U(g) * 500.0);
        check(std::ldexp(U(g), -(int)(g() % 1070)));
        check((double)(g() % 100000000ull) * 1e-6);
        check((double)(g() % 10000000000ull) / 1024.0);
    }
    printf("checked %ld values, %ld outside the covered range, %ld differences\n", checked, unsupported, bad);
    return bad ? 1 : 0;
}
