// fhx_deflate.hpp - the pieces of the device-side deflate of the significances file that do not depend on HIP: the row
// tokeniser (host + device), the length / distance code maps of RFC 1951, the CRC-32 combine step, and - host only - the
// length-limited Huffman codes (package-merge), canonical code assignment and the dynamic block header.  fhx_emit.inc uses
// them in kernels; tests/native/deflate_check.cpp runs the same functions sequentially on the CPU and inflates the result
// with zlib.
#pragma once
#include <algorithm>
#include <cstdint>
#include <vector>

#include "fhx_fmt.hpp"

namespace fhx {
namespace emit {

constexpr int N_LIT = 286, N_DIST = 30, N_SYM = N_LIT + N_DIST;

// ---- tokens --------------------------------------------------------------------------------------------------------
FHX_HD void len_code(int len, int& sym, int& ebits, int& eval) {
    if (len <= 10) {
        sym = 254 + len;                       // 257 + (len - 3)
        ebits = 0;
        eval = 0;
    } else if (len == 258) {
        sym = 285;
        ebits = 0;
        eval = 0;
    } else {
        const int l = len - 3;
        ebits = 29 - __builtin_clz((unsigned int)l);            // floor(log2 l) - 2
        sym = 257 + 4 * ebits + 4 + ((l >> ebits) & 3);
        eval = l & ((1 << ebits) - 1);
    }
}
FHX_HD void dist_code(int dist, int& sym, int& ebits, int& eval) {
    const int x = dist - 1;
    if (x < 4) {
        sym = x;
        ebits = 0;
        eval = 0;
    } else {
        ebits = 30 - __builtin_clz((unsigned int)x);            // floor(log2 x) - 1
        sym = 2 * ebits + 2 + ((x >> ebits) & 1);
        eval = x & ((1 << ebits) - 1);
    }
}


// Tokens of one row against the previous one.  Sink: literal(byte), match(len, dist).
template <class Sink>
FHX_HD void tokenise(const unsigned char* cur, int lc, const unsigned char* prev, int lp, Sink& S) {
    int pos = 0;
    if (lp > 0) {
        const int mx = lc < lp ? lc : lp;
        int l = 0;
        while (l < mx && cur[l] == prev[l]) ++l;
        if (l >= 3) {
            S.match(l, lp);
            pos = l;
        }
    }
    // field by field: cs / ps = start of the current field in the two rows
    int cs = 0, ps = 0;
    while (cs < lc) {
        int ce = cs;
        while (ce < lc && cur[ce] != '\t' && cur[ce] != '\n') ++ce;
        if (ce < lc) ++ce;                                     // the delimiter belongs to the field
        int pe = ps;
        if (lp > 0) {
            while (pe < lp && prev[pe] != '\t' && prev[pe] != '\n') ++pe;
            if (pe < lp) ++pe;
        }
        if (ce > pos) {
            const int from = cs > pos ? cs : pos;
            bool same = lp > 0 && from == cs && (ce - cs) == (pe - ps) && (ce - cs) >= 3;
            for (int k = 0; same && k < ce - cs; ++k) same = cur[cs + k] == prev[ps + k];
            if (same) {
                S.match(ce - cs, lp + cs - ps);
            } else {
                for (int k = from; k < ce; ++k) S.literal(cur[k]);
            }
            pos = ce;
        }
        cs = ce;
        ps = pe;
    }
}


FHX_HD unsigned int crc_multmodp(unsigned int a, unsigned int b) {       // zlib crc32.c multmodp
    unsigned int m = 1u << 31, p = 0;
    for (;;) {
        if (a & m) {
            p ^= b;
            if ((a & (m - 1)) == 0) break;
        }
        m >>= 1;
        b = (b & 1u) ? (b >> 1) ^ 0xedb88320u : b >> 1;
    }
    return p;
}


// Mem: how shared counters / boundary words are updated (atomics in a kernel, plain operations in the sequential check)
template <class Mem>
struct HistSink {
    unsigned int* h;                           // LDS, N_SYM counters
    FHX_HD void literal(unsigned char b) { Mem::add(&h[b], 1u); }
    FHX_HD void match(int len, int dist) {
        int s, eb, ev;
        len_code(len, s, eb, ev);
        Mem::add(&h[s], 1u);
        dist_code(dist, s, eb, ev);
        Mem::add(&h[N_LIT + s], 1u);
    }
};

struct CodeTab {                                 // per member, uploaded by the host
    unsigned short code[N_SYM];                  // bit-reversed Huffman codes (emitted LSB first)
    unsigned char len[N_SYM];
};

struct BitsSink {
    const unsigned char* len;                    // LDS copy of the member's code lengths
    unsigned int bits;
    FHX_HD void literal(unsigned char b) { bits += len[b]; }
    FHX_HD void match(int l, int d) {
        int s, eb, ev;
        len_code(l, s, eb, ev);
        bits += len[s] + eb;
        dist_code(d, s, eb, ev);
        bits += len[N_LIT + s] + eb;
    }
};

// A row's bits: plain stores for the 32-bit words it owns entirely, Mem::bit_or for its first and last word (shared with
// the neighbouring rows; the buffer starts zeroed)
template <class Mem>
struct EncodeSink {
    const unsigned short* code;                  // LDS
    const unsigned char* len;
    unsigned int* out;                           // the output buffer as 32-bit words
    unsigned long long acc;
    int nacc;
    unsigned long long word;
    bool first;
    FHX_HD void put(unsigned int bits, int n) {
        acc |= (unsigned long long)bits << nacc;
        nacc += n;
        if (nacc >= 32) {
            const unsigned int wv = (unsigned int)acc;
            if (first) {
                Mem::bit_or(&out[word], wv);
                first = false;
            } else {
                out[word] = wv;
            }
            ++word;
            acc >>= 32;
            nacc -= 32;
        }
    }
    FHX_HD void literal(unsigned char b) { put(code[b], len[b]); }
    FHX_HD void match(int l, int d) {
        int s, eb, ev;
        len_code(l, s, eb, ev);
        put(code[s], len[s]);
        if (eb) put((unsigned int)ev, eb);
        dist_code(d, s, eb, ev);
        put(code[N_LIT + s], len[N_LIT + s]);
        if (eb) put((unsigned int)ev, eb);
    }
    FHX_HD void finish() {
        if (nacc > 0) Mem::bit_or(&out[word], (unsigned int)acc);
    }
};


// ---- host side: length-limited Huffman codes (package-merge) and the deflate block header ---------------------------------
inline void package_merge(const std::vector<unsigned long long>& freq, int max_len, std::vector<unsigned char>& len_out) {
    const int n = (int)freq.size();
    len_out.assign(n, 0);
    std::vector<int> used;
    for (int i = 0; i < n; ++i)
        if (freq[i]) used.push_back(i);
    if (used.empty()) return;
    if (used.size() == 1) {
        len_out[used[0]] = 1;
        return;
    }
    std::sort(used.begin(), used.end(), [&](int a, int b) { return freq[a] != freq[b] ? freq[a] < freq[b] : a < b; });
    const int m = (int)used.size();
    struct Node {
        unsigned long long w;
        std::vector<int> leaves;                  // indices into `used` (with multiplicity over levels)
    };
    std::vector<Node> prev;
    for (int level = 0; level < max_len; ++level) {
        std::vector<Node> cur;
        cur.reserve(m + prev.size() / 2);
        for (int i = 0; i < m; ++i) cur.push_back(Node{freq[used[i]], {i}});
        for (size_t k = 0; k + 1 < prev.size(); k += 2) {
            Node p{prev[k].w + prev[k + 1].w, prev[k].leaves};
            p.leaves.insert(p.leaves.end(), prev[k + 1].leaves.begin(), prev[k + 1].leaves.end());
            cur.push_back(std::move(p));
        }
        std::stable_sort(cur.begin(), cur.end(), [](const Node& a, const Node& b) { return a.w < b.w; });
        prev.swap(cur);
    }
    std::vector<int> cl(m, 0);
    for (int k = 0; k < 2 * m - 2 && k < (int)prev.size(); ++k)
        for (int leaf : prev[k].leaves) ++cl[leaf];
    for (int i = 0; i < m; ++i) len_out[used[i]] = (unsigned char)cl[i];
}

inline void canonical_codes(const std::vector<unsigned char>& len, std::vector<unsigned short>& rev_code) {
    const int n = (int)len.size();
    rev_code.assign(n, 0);
    int bl_count[17] = {0};
    for (int i = 0; i < n; ++i) bl_count[len[i]]++;
    bl_count[0] = 0;
    int next[17] = {0}, code = 0;
    for (int b = 1; b <= 16; ++b) {
        code = (code + bl_count[b - 1]) << 1;
        next[b] = code;
    }
    for (int i = 0; i < n; ++i) {
        if (!len[i]) continue;
        const int c = next[len[i]]++;
        int r = 0;
        for (int b = 0; b < len[i]; ++b)
            if (c & (1 << b)) r |= 1 << (len[i] - 1 - b);
        rev_code[i] = (unsigned short)r;
    }
}

struct BitString {
    std::vector<unsigned char> bytes;
    unsigned int nbits = 0;
    void put(unsigned int v, int n) {
        for (int k = 0; k < n; ++k) {
            if ((nbits & 7u) == 0) bytes.push_back(0);
            if (v & (1u << k)) bytes.back() |= (unsigned char)(1u << (nbits & 7u));
            ++nbits;
        }
    }
};

// dynamic-Huffman block header (RFC 1951 3.2.7) for the given code lengths, BFINAL = 1; code lengths are sent one by one
// (symbols 0..15 of the code-length alphabet only: ~160 bytes per multi-megabyte member)
inline void block_header(const std::vector<unsigned char>& lit_len, const std::vector<unsigned char>& dist_len, BitString& out) {
    out.put(1, 1);                                   // BFINAL
    out.put(2, 2);                                   // BTYPE = dynamic Huffman
    out.put(N_LIT - 257, 5);
    out.put(N_DIST - 1, 5);
    std::vector<unsigned long long> cl_freq(19, 0);
    for (unsigned char l : lit_len) cl_freq[l]++;
    for (unsigned char l : dist_len) cl_freq[l]++;
    std::vector<unsigned char> cl_len;
    package_merge(cl_freq, 7, cl_len);
    std::vector<unsigned short> cl_code;
    canonical_codes(cl_len, cl_code);
    static const int order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
    out.put(19 - 4, 4);
    for (int k = 0; k < 19; ++k) out.put(cl_len[order[k]], 3);
    for (unsigned char l : lit_len) out.put(cl_code[l], cl_len[l]);
    for (unsigned char l : dist_len) out.put(cl_code[l], cl_len[l]);
}

const unsigned char kLenExtra[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
const unsigned char kDistExtra[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};

}  // namespace emit
}  // namespace fhx
