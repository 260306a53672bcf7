// deflate_check.cpp - the device-side deflate of the significances file, run sequentially on the CPU.
//
// Uses the SAME functions the kernels of fhx_emit.inc use (fhx_deflate.hpp: tokenise, the sinks, len_code / dist_code,
// crc_multmodp, package_merge, canonical_codes, block_header), with the rows of a member encoded in a shuffled order at the
// bit offsets a scan gives them - the situation on the GPU - and inflates every member with zlib, which also checks the
// CRC-32 (combined from per-row CRCs as the kernel does) and ISIZE.  Build + run: tests/test_fmt.py.
//
//   g++ -O2 -std=c++17 -I fithic_amd/csrc tests/native/deflate_check.cpp -lz -o deflate_check && ./deflate_check
#include <zlib.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <numeric>
#include <random>
#include <string>
#include <vector>

#include "fhx_deflate.hpp"

using namespace fhx::emit;

struct HostMem {
    static void add(unsigned int* p, unsigned int v) { *p += v; }
    static void bit_or(unsigned int* p, unsigned int v) { *p |= v; }
};

static unsigned int crc_tab[256], x2n_tab[32];

static void init_tables() {
    for (unsigned int i = 0; i < 256; ++i) {
        unsigned int c = i;
        for (int k = 0; k < 8; ++k) c = (c & 1u) ? 0xedb88320u ^ (c >> 1) : c >> 1;
        crc_tab[i] = c;
    }
    unsigned int p = 1u << 30;
    x2n_tab[0] = p;
    for (int k = 1; k < 32; ++k) x2n_tab[k] = p = crc_multmodp(p, p);
}

static bool encode_member(const std::vector<std::string>& rows, std::mt19937_64& rng, std::vector<unsigned char>& member) {
    const int n = (int)rows.size();
    auto prev_of = [&](int r, const unsigned char*& prev, int& lp) {
        prev = nullptr;
        lp = 0;
        if (r > 0) {
            prev = (const unsigned char*)rows[r - 1].data();
            lp = (int)rows[r - 1].size();
        }
    };
    // pass 1: histogram, text offsets, CRC
    std::vector<unsigned int> h(N_SYM, 0);
    unsigned long long text = 0;
    for (const auto& r : rows) text += r.size();
    unsigned int crc = 0;
    unsigned long long before = 0;
    for (int r = 0; r < n; ++r) {
        const unsigned char* prev;
        int lp;
        prev_of(r, prev, lp);
        const int lc = (int)rows[r].size();
        if (lc > 0) {
            HistSink<HostMem> S{h.data()};
            tokenise((const unsigned char*)rows[r].data(), lc, prev, lp, S);
            unsigned int c = 0xffffffffu;
            for (int k = 0; k < lc; ++k) c = crc_tab[(c ^ (unsigned char)rows[r][k]) & 0xffu] ^ (c >> 8);
            c ^= 0xffffffffu;
            unsigned long long nb = text - before - (unsigned long long)lc;
            unsigned int op = 1u << 31, k = 3;
            while (nb) {
                if (nb & 1ull) op = crc_multmodp(x2n_tab[k & 31u], op);
                nb >>= 1;
                ++k;
            }
            crc ^= crc_multmodp(op, c);
        }
        before += (unsigned long long)lc;
    }
    // codes + header
    std::vector<unsigned long long> lf(N_LIT, 0), df(N_DIST, 0);
    for (int s = 0; s < N_LIT; ++s) lf[s] = h[s];
    for (int s = 0; s < N_DIST; ++s) df[s] = h[N_LIT + s];
    lf[256] = 1;
    std::vector<unsigned char> ll, dl;
    package_merge(lf, 15, ll);
    package_merge(df, 15, dl);
    bool any = false;
    for (unsigned char l : dl) any = any || l;
    if (!any) dl[0] = 1;
    for (unsigned char l : ll)
        if (l > 15) return false;
    {   // Kraft sums: complete or under-full codes only
        unsigned long long k = 0;
        for (unsigned char l : ll)
            if (l) k += 1ull << (15 - l);
        if (k > (1ull << 15)) {
            std::printf("literal code over-subscribed\n");
            return false;
        }
    }
    std::vector<unsigned short> lc, dc;
    canonical_codes(ll, lc);
    canonical_codes(dl, dc);
    CodeTab tab;
    for (int s = 0; s < N_LIT; ++s) {
        tab.code[s] = lc[s];
        tab.len[s] = ll[s];
    }
    for (int s = 0; s < N_DIST; ++s) {
        tab.code[N_LIT + s] = dc[s];
        tab.len[N_LIT + s] = dl[s];
    }
    BitString hb;
    block_header(ll, dl, hb);
    unsigned long long bits_from_hist = 0;
    for (int s = 0; s < 256; ++s) bits_from_hist += lf[s] * ll[s];
    for (int s = 257; s < N_LIT; ++s) bits_from_hist += lf[s] * (ll[s] + kLenExtra[s - 257]);
    for (int s = 0; s < N_DIST; ++s) bits_from_hist += df[s] * (dl[s] + kDistExtra[s]);
    // pass 2: bits per row, scan
    std::vector<unsigned long long> bit_at(n + 1, 0);
    for (int r = 0; r < n; ++r) {
        const unsigned char* prev;
        int lp;
        prev_of(r, prev, lp);
        BitsSink B{tab.len, 0u};
        if (!rows[r].empty()) tokenise((const unsigned char*)rows[r].data(), (int)rows[r].size(), prev, lp, B);
        bit_at[r + 1] = bit_at[r] + B.bits;
    }
    if (bit_at[n] != bits_from_hist) {
        std::printf("bits from the histogram %llu, from the rows %llu\n", bits_from_hist, bit_at[n]);
        return false;
    }
    const int head = 24;
    const unsigned long long bit0 = (unsigned long long)head * 8 + hb.nbits;
    const unsigned long long deflate_bits = hb.nbits + bit_at[n] + ll[256];
    const unsigned long long size = head + (deflate_bits + 7) / 8 + 8;
    std::vector<unsigned int> out((size + 11) / 4, 0u);
    // pass 3: rows in a shuffled order
    std::vector<int> order(n);
    std::iota(order.begin(), order.end(), 0);
    std::shuffle(order.begin(), order.end(), rng);
    for (int r : order) {
        if (rows[r].empty()) continue;
        const unsigned char* prev;
        int lp;
        prev_of(r, prev, lp);
        const unsigned long long at = bit0 + bit_at[r];
        EncodeSink<HostMem> E{tab.code, tab.len, out.data(), 0ull, (int)(at & 31ull), at >> 5, true};
        tokenise((const unsigned char*)rows[r].data(), (int)rows[r].size(), prev, lp, E);
        E.finish();
    }
    // frame (what em_frame does)
    {
        unsigned long long at = bit0 - hb.nbits;
        for (unsigned int k = 0; k < hb.nbits; k += 8) {
            const unsigned int nb = hb.nbits - k < 8 ? hb.nbits - k : 8;
            const unsigned long long v = (unsigned long long)(hb.bytes[k >> 3] & ((1u << nb) - 1u)) << (at & 31ull);
            out[at >> 5] |= (unsigned int)v;
            if (v >> 32) out[(at >> 5) + 1] |= (unsigned int)(v >> 32);
            at += nb;
        }
        at = bit0 + bit_at[n];
        const unsigned long long v = (unsigned long long)tab.code[256] << (at & 31ull);
        out[at >> 5] |= (unsigned int)v;
        if (v >> 32) out[(at >> 5) + 1] |= (unsigned int)(v >> 32);
    }
    member.assign((unsigned char*)out.data(), (unsigned char*)out.data() + size);
    const unsigned char gz[24] = {0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 255, 12, 0, 'F', 'H', 8, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    std::memcpy(member.data(), gz, 24);
    for (int k = 0; k < 8; ++k) member[16 + k] = (unsigned char)((size >> (8 * k)) & 0xFF);
    for (int k = 0; k < 4; ++k) {
        member[size - 8 + k] = (unsigned char)((crc >> (8 * k)) & 0xFF);
        member[size - 4 + k] = (unsigned char)(((unsigned int)text >> (8 * k)) & 0xFF);
    }
    return true;
}

static bool inflate_equals(const std::vector<unsigned char>& member, const std::string& want) {
    z_stream zs;
    std::memset(&zs, 0, sizeof(zs));
    if (inflateInit2(&zs, 15 + 16) != Z_OK) return false;
    std::string got(want.size() + 64, '\0');
    zs.next_in = (Bytef*)member.data();
    zs.avail_in = (uInt)member.size();
    zs.next_out = (Bytef*)&got[0];
    zs.avail_out = (uInt)got.size();
    const int rc = inflate(&zs, Z_FINISH);
    const bool ok = rc == Z_STREAM_END && zs.total_out == want.size() && zs.avail_in == 0 && std::memcmp(got.data(), want.data(), want.size()) == 0;
    if (!ok) std::printf("inflate rc %d (%s), %lu of %zu bytes, %u input bytes left\n", rc, zs.msg ? zs.msg : "", zs.total_out, want.size(), zs.avail_in);
    inflateEnd(&zs);
    return ok;
}

int main() {
    init_tables();
    std::mt19937_64 rng(20260928);
    long long members = 0, rows_total = 0, in_bytes = 0, out_bytes = 0;
    auto run = [&](const std::vector<std::string>& rows, const char* what) {
        std::vector<unsigned char> member;
        std::string want;
        for (const auto& r : rows) want += r;
        if (!encode_member(rows, rng, member) || !inflate_equals(member, want)) {
            std::printf("FAILED: %s (%zu rows)\n", what, rows.size());
            std::exit(1);
        }
        ++members;
        rows_total += (long long)rows.size();
        in_bytes += (long long)want.size();
        out_bytes += (long long)member.size();
    };
    // table-like rows as the writer produces them
    for (int rep = 0; rep < 40; ++rep) {
        std::vector<std::string> rows;
        const int n = rep < 4 ? 1 + rep : (int)(rng() % 3000) + 1;
        int mid1 = 5000;
        for (int i = 0; i < n; ++i) {
            if (rng() % 7 == 0) mid1 += 10000;
            const int mid2 = mid1 + 10000 * (int)(rng() % 300);
            char buf[256];
            int k = 0;
            const char* c1 = rep % 3 ? "chr1" : "chrUn_gl000220";
            k += std::snprintf(buf + k, sizeof(buf) - k, "%s\t%d\t%s\t%d\t%d\t", c1, mid1, rep % 5 ? c1 : "chrX", mid2, (int)(rng() % 50) + 1);
            const double p = std::ldexp((double)(rng() >> 11), -53 - (int)(rng() % 200));
            const double vals[4] = {rng() % 4 ? p : 1.0, rng() % 3 ? 1.0 : p * 3, 0.5 + (double)(rng() % 1000) / 997.0, 0.5 + (double)(rng() % 1000) / 997.0};
            for (double v : vals) {
                k += fhx::fmt::fmt_e6(v, buf + k);
                buf[k++] = '\t';
            }
            k += fhx::fmt::fmt_f6((double)(rng() % 100000) / 771.0, buf + k);
            buf[k++] = '\n';
            if (rep % 4 == 1 && rng() % 5 == 0) k = 0;                 // rows that are not emitted
            rows.emplace_back(buf, buf + k);
        }
        run(rows, "table rows");
    }
    // adversarial shapes: repeats, tiny rows, empty fields, one symbol only, no match at all, binary bytes
    run({"a\n"}, "one tiny row");
    run({"abc\tdef\n", "abc\tdef\n", "abc\tdef\n", "", "abc\tdef\n"}, "identical rows with a gap");
    run({"\t\t\t\n", "\t\t\t\n", "x\t\ty\t\n", "x\t\ty\tz\n"}, "empty fields");
    run({std::string(127, 'q'), std::string(127, 'q'), std::string(126, 'q') + "\n"}, "long runs without delimiters");
    {
        std::vector<std::string> rows;
        for (int i = 0; i < 500; ++i) {
            std::string r;
            const int l = (int)(rng() % 127) + 1;
            for (int k = 0; k < l; ++k) r.push_back((char)(rng() & 0xFF));
            rows.push_back(r);
        }
        run(rows, "random bytes");
    }
    {
        std::vector<std::string> rows;                                 // skewed literal frequencies: forces the 15-bit limit
        unsigned long long f = 1;
        for (int s = 0; s < 40; ++s) {
            for (unsigned long long k = 0; k < f && k < 3000; ++k) rows.push_back(std::string(1, (char)('!' + s)) + std::string(1, (char)(200 - s)));
            f = f + f / 2 + 1;
        }
        std::shuffle(rows.begin(), rows.end(), rng);
        run(rows, "skewed frequencies");
    }
    std::printf("inflated %lld members, %lld rows, %lld text bytes -> %lld bytes (ratio %.3f), 0 differences\n", members, rows_total, in_bytes,
                out_bytes, (double)out_bytes / (double)in_bytes);
    return 0;
}
