// fhx_fmt.hpp - printf("%e") / printf("%f") / "%d" for the significances writer, usable on the host AND in a kernel.
//
// The reference writes every output row with "%s\t%d\t%s\t%d\t%d\t%e\t%e\t%e\t%e\t%f\n" (fithic/fithic.py:1202,1212): C's
// correctly rounded decimal conversion (round-half-even on the EXACT binary value), six digits after the point.  These
// functions produce the same characters with exact integer arithmetic - a double is m * 2^e, so m * 10^j / 2^s is an integer
// division whose remainder decides the rounding; nothing is approximated - and are checked against std::to_chars / snprintf
// on 10^8 values (tests/test_fmt.py drives tools/fmt_check.cpp on the CPU; the GPU test formats on the device).
// Ranges the device path does not cover return -1 and the caller formats that row on the host: |v| >= 2^63 for %f,
// |v| >= 2^64 for %e (ExpCC, p, q and the biases of a Fit-Hi-C run never get there).
#pragma once
#include <cstdint>
#include <cstring>

#if defined(__HIPCC__)          // hipcc: usable in kernels; g++ (tests/native/fmt_check.cpp): plain inline functions
#define FHX_HD __host__ __device__ __forceinline__
#else
#define FHX_HD inline
#endif

namespace fhx {
namespace fmt {

FHX_HD int put_u64(char* dst, unsigned long long v) {
    char tmp[24];
    int n = 0;
    do {
        tmp[n++] = (char)('0' + (int)(v % 10ull));
        v /= 10ull;
    } while (v);
    for (int k = 0; k < n; ++k) dst[k] = tmp[n - 1 - k];
    return n;
}

FHX_HD int put_i64(char* dst, long long v) {
    if (v < 0) {
        dst[0] = '-';
        return 1 + put_u64(dst + 1, 0ull - (unsigned long long)v);
    }
    return put_u64(dst, (unsigned long long)v);
}

FHX_HD int put_special(char* dst, unsigned long long bits) {     // nan / inf as Python prints them
    const bool neg = (bits >> 63) != 0;
    const bool is_nan = (bits & 0x000FFFFFFFFFFFFFull) != 0;
    int n = 0;
    if (is_nan) {
        dst[0] = 'n'; dst[1] = 'a'; dst[2] = 'n';
        return 3;
    }
    if (neg) dst[n++] = '-';
    dst[n++] = 'i'; dst[n++] = 'n'; dst[n++] = 'f';
    return n;
}

constexpr int kLimbs = 40;               // 32-bit limbs: m * 10^j up to 53 + 1100 bits

// N = m * 10^j, little-endian limbs; returns the number of limbs in use
FHX_HD int big_m_pow10(unsigned long long m, int j, unsigned int* N) {
    int n = 2;
    N[0] = (unsigned int)m;
    N[1] = (unsigned int)(m >> 32);
    while (j > 0) {
        const int step = j >= 9 ? 9 : j;
        unsigned int mul = 1;
        for (int k = 0; k < step; ++k) mul *= 10u;
        unsigned long long carry = 0;
        for (int i = 0; i < n; ++i) {
            const unsigned long long t = (unsigned long long)N[i] * mul + carry;
            N[i] = (unsigned int)t;
            carry = t >> 32;
        }
        if (carry) N[n++] = (unsigned int)carry;
        j -= step;
    }
    return n;
}

// q = round-half-even(N / 2^s), assuming the quotient fits 64 bits
FHX_HD unsigned long long big_shift_round(const unsigned int* N, int n, int s) {
    const int w = s >> 5, b = s & 31;
    unsigned long long q = 0;
    for (int k = 2; k >= 0; --k) {       // bits s .. s+95 -> up to three limbs, then realigned
        const int i = w + k;
        const unsigned long long limb = i < n ? N[i] : 0u;
        if (k == 2)
            q = b ? (limb << (64 - b)) : 0ull;           // only its low b bits can land inside 64 bits
        else if (k == 1)
            q |= b ? (limb << (32 - b)) : (limb << 32);
        else
            q |= limb >> b;
    }
    if (s == 0) return q;
    // remainder = low s bits: half bit is bit s-1, sticky = any lower bit
    const int hw = (s - 1) >> 5, hb = (s - 1) & 31;
    const bool half = hw < n && ((N[hw] >> hb) & 1u);
    if (!half) return q;
    bool sticky = hw < n && (N[hw] & ((1u << hb) - 1u)) != 0;
    for (int i = 0; i < hw && i < n && !sticky; ++i) sticky = N[i] != 0;
    if (sticky || (q & 1ull)) ++q;
    return q;
}

// "%e": sign, d.dddddd, e, sign, at least two exponent digits.  Returns the length, or -1 (not covered here).
FHX_HD int fmt_e6(double v, char* dst) {
    unsigned long long bits;
    memcpy(&bits, &v, sizeof(bits));
    const int bexp = (int)((bits >> 52) & 0x7FF);
    if (bexp == 0x7FF) return put_special(dst, bits);
    int len = 0;
    if (bits >> 63) dst[len++] = '-';
    unsigned long long m = bits & 0x000FFFFFFFFFFFFFull;
    int e2;
    if (bexp == 0) {
        if (m == 0) {
            const char* z = "0.000000e+00";
            for (int k = 0; k < 12; ++k) dst[len + k] = z[k];
            return len + 12;
        }
        e2 = -1074;
    } else {
        m |= 1ull << 52;
        e2 = bexp - 1075;
    }
    int blen = 64 - __builtin_clzll(m);
    // k = floor(log10 |v|) or one less: floor((e2 + blen - 1) * log10(2)) with 78913 / 2^18 = 0.30102920...
    const int p2 = e2 + blen - 1;
    int k = (int)(((long long)p2 * 78913ll) >> 18);
    unsigned long long D = 0;
    for (int attempt = 0; attempt < 4; ++attempt) {
        const int j = 6 - k;                             // D = round(|v| * 10^j)
        if (e2 >= 0) {
            if (e2 > 11) return -1;                      // |v| >= 2^64
            const unsigned long long iv = m << e2;       // an integer
            if (j >= 0) {
                unsigned long long mul = 1;
                for (int t = 0; t < j; ++t) mul *= 10ull;
                D = iv * mul;                            // k <= 6 here, no overflow
            } else {
                unsigned long long den = 1;
                for (int t = 0; t < -j; ++t) den *= 10ull;
                D = iv / den;
                const unsigned long long rem = iv - D * den;
                if (rem * 2 > den || (rem * 2 == den && (D & 1ull))) ++D;
            }
        } else {
            const int s = -e2;
            if (j >= 0) {
                unsigned int N[kLimbs];
                const int n = big_m_pow10(m, j, N);
                D = big_shift_round(N, n, s);
            } else {                                     // 10^7 <= |v| < 2^53: the divisor 10^-j * 2^s is below 2^34
                unsigned long long den = 1;
                for (int t = 0; t < -j; ++t) den *= 10ull;
                den <<= s;
                D = m / den;
                const unsigned long long rem = m - D * den;
                if (rem * 2 > den || (rem * 2 == den && (D & 1ull))) ++D;
            }
        }
        if (D >= 10000000ull) {
            if (D == 10000000ull) {
                // either rounding carried 9.9999995.. up, or k was one too small; both print 1.000000 with k + 1 - but only
                // the first is right when the true value is below 10^(k+1): recomputing with k + 1 decides exactly
            }
            ++k;
            continue;
        }
        if (D < 1000000ull) {
            --k;
            continue;
        }
        break;
    }
    // digits
    char dig[7];
    unsigned long long t = D;
    for (int i = 6; i >= 0; --i) {
        dig[i] = (char)('0' + (int)(t % 10ull));
        t /= 10ull;
    }
    dst[len++] = dig[0];
    dst[len++] = '.';
    for (int i = 1; i < 7; ++i) dst[len++] = dig[i];
    dst[len++] = 'e';
    int ke = k;
    if (ke < 0) {
        dst[len++] = '-';
        ke = -ke;
    } else {
        dst[len++] = '+';
    }
    if (ke >= 100) {
        dst[len++] = (char)('0' + ke / 100);
        ke %= 100;
    }
    dst[len++] = (char)('0' + ke / 10);
    dst[len++] = (char)('0' + ke % 10);
    return len;
}

// "%f": sign, integer digits, '.', six digits.  Returns the length, or -1 for |v| >= 2^63.
FHX_HD int fmt_f6(double v, char* dst) {
    unsigned long long bits;
    memcpy(&bits, &v, sizeof(bits));
    const int bexp = (int)((bits >> 52) & 0x7FF);
    if (bexp == 0x7FF) return put_special(dst, bits);
    int len = 0;
    if (bits >> 63) dst[len++] = '-';
    unsigned long long m = bits & 0x000FFFFFFFFFFFFFull;
    int e2;
    if (bexp == 0) {
        e2 = -1074;
    } else {
        m |= 1ull << 52;
        e2 = bexp - 1075;
    }
    unsigned long long ip = 0, frac = 0;
    if (m == 0) {
        // 0.000000
    } else if (e2 >= 0) {
        if (e2 > 10) return -1;                          // |v| >= 2^63
        ip = m << e2;
    } else {
        const int s = -e2;
        // F = round(m * 10^6 / 2^s) (m * 10^6 < 2^73), then split by 10^6
        unsigned int N[kLimbs];
        const int n = big_m_pow10(m, 6, N);
        if (s > 96 + 32) {
            ip = 0;
            frac = 0;
        } else {
            // the quotient can exceed 64 bits only when s < 9; then |v| >= 2^44 has no fractional bits beyond 2^-9: still exact below
            if (s <= 9) {
                // F = (m * 10^6) >> s is up to 73 bits: divide first: ip = m >> s exactly, fractional part from the low s bits
                ip = m >> s;
                const unsigned long long low = m & ((1ull << s) - 1ull);            // < 2^9
                const unsigned long long num = low * 1000000ull;                     // < 2^29
                frac = num >> s;
                const unsigned long long rem = num & ((1ull << s) - 1ull);
                const unsigned long long half = 1ull << (s - 1);
                if (rem > half || (rem == half && (frac & 1ull))) ++frac;
                if (frac == 1000000ull) {
                    frac = 0;
                    ++ip;
                }
            } else {
                const unsigned long long F = big_shift_round(N, n, s);               // < 2^64 since s >= 10
                ip = F / 1000000ull;
                frac = F % 1000000ull;
            }
        }
    }
    len += put_u64(dst + len, ip);
    dst[len++] = '.';
    char dig[6];
    for (int i = 5; i >= 0; --i) {
        dig[i] = (char)('0' + (int)(frac % 10ull));
        frac /= 10ull;
    }
    for (int i = 0; i < 6; ++i) dst[len++] = dig[i];
    return len;
}

}  // namespace fmt
}  // namespace fhx
