// io_sanitize.cpp - the host side of libfithic_mi355x.so that consumes untrusted bytes (gzip containers, deflate streams, table
// text) and the host fit, driven with random and mutated inputs under AddressSanitizer + UndefinedBehaviorSanitizer.
//
// Linked against fhx_io.cpp, fhx_gunzip.cpp and fhx_host.cpp compiled with
//     g++ -std=c++17 -O1 -g -fsanitize=address,undefined -fno-sanitize-recover=all -ffp-contract=off
// (tests/test_sanitizers.py builds and runs it).  A memory error or undefined behaviour aborts the process; the harness itself
// checks what must hold whatever the input: an accepted file yields as many rows as Python-style line splitting of its text,
// a call never reports success AND an error text, and the three inflate routes (member chain with sizes, one plain stream cut
// at block starts on all cores, zlib on one thread) return the same bytes.
//
//     io_sanitize <scratch dir> <cases> [seed]
#include <zlib.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <string>
#include <vector>

#include "../../include/fithic_mi355x.h"
#include "fhx_host.hpp"

static std::mt19937_64 rng;
static uint64_t rnd(uint64_t n) { return n ? rng() % n : 0; }

static std::string gz_member(const std::string& text, int level, int strategy) {
    z_stream zs;
    std::memset(&zs, 0, sizeof(zs));
    deflateInit2(&zs, level, Z_DEFLATED, 15 + 16, 8, strategy);
    std::string out(deflateBound(&zs, (uLong)text.size()) + 64, '\0');
    zs.next_in = (Bytef*)text.data();
    zs.avail_in = (uInt)text.size();
    zs.next_out = (Bytef*)&out[0];
    zs.avail_out = (uInt)out.size();
    // a few sync / full flushes on the way: stored and empty blocks, byte-aligned block starts
    while (zs.avail_in > 0) {
        const uInt step = (uInt)std::min<uint64_t>(zs.avail_in, 1 + rnd(1 << 15));
        const uInt rest = zs.avail_in - step;
        zs.avail_in = step;
        deflate(&zs, rnd(4) == 0 ? (rnd(2) ? Z_SYNC_FLUSH : Z_FULL_FLUSH) : Z_NO_FLUSH);
        zs.avail_in += rest;
    }
    deflate(&zs, Z_FINISH);
    out.resize(out.size() - zs.avail_out);
    deflateEnd(&zs);
    return out;
}

// a member that carries its own compressed size in an "FH" extra subfield, as this library's writers emit them (fhx_io.cpp
// scan_members): such files are inflated member by member on all cores, trusting - and therefore checking - those sizes
static std::string fh_member(const std::string& text, int level) {
    const std::string m = gz_member(text, level, 0);           // zlib writes a 10-byte header without optional fields
    std::string out = m.substr(0, 10);
    out[3] = (char)(out[3] | 4);                               // FLG.FEXTRA
    const uint64_t size = m.size() + 2 + 12;
    const unsigned char extra[14] = {12, 0, 'F', 'H', 8, 0, (unsigned char)size, (unsigned char)(size >> 8), (unsigned char)(size >> 16),
                                     (unsigned char)(size >> 24), (unsigned char)(size >> 32), (unsigned char)(size >> 40),
                                     (unsigned char)(size >> 48), (unsigned char)(size >> 56)};
    out.append((const char*)extra, sizeof(extra));
    out += m.substr(10);
    return out;
}

static const char* kNames[] = {"chr1", "chr2", "chrX", "10", "scaffold_12|a", "c"};
static const char* kOdd[] = {"nan", "inf", "-inf", "1e400", "0x10", "1_000", "1__0", "_1", "2147483648", "-2147483649", "99999999999999999999",
                             "1.5e3", ".5", "5.", "+7", "--7", "", "\xc2\xa0", "1e", "Infinity", "nan(1)", "1" "0000000000000000000000000000000000000000000000000000000000000000000000"};

// odd: adversarial tokens, separators and line ends, about two per file (so that half of the odd files still parse)
static std::string table_text(int kind, size_t rows, bool odd) {
    std::string t;
    char b[256];
    int64_t mid = 5000;
    const uint64_t every = 2 * rows + 2;
    for (size_t i = 0; i < rows; ++i) {
        const char* a = kNames[rnd(odd ? 6 : 3)];
        mid += (int64_t)rnd(30) * 500;
        const char* sep = odd && rnd(20) == 0 ? (rnd(2) ? "  " : "\x1c") : "\t";
        if (kind == 0)
            std::snprintf(b, sizeof(b), "%s%s%lld%s%s%s%lld%s%d", a, sep, (long long)mid, sep, a, sep, (long long)(mid + 5000 * (1 + (int64_t)rnd(400))), sep,
                          1 + (int)rnd(40));
        else if (kind == 1)
            std::snprintf(b, sizeof(b), "%s%s0%s%lld%s%d%s0", a, sep, sep, (long long)mid, sep, (int)rnd(3), sep);
        else
            std::snprintf(b, sizeof(b), "%s%s%lld%s%.6f", a, sep, (long long)mid, sep, 0.2 + 0.002 * (double)rnd(1000));
        t += b;
        if (odd && rnd(every) == 0) {              // append an adversarial token
            t += rnd(2) ? "\t" : " ";
            t += kOdd[rnd(sizeof(kOdd) / sizeof(kOdd[0]))];
        }
        if (odd && rnd(every) == 0) {              // replace the tail of the line by one
            const size_t cut = t.size() - std::min<size_t>(t.size(), 1 + rnd(12));
            t.resize(cut);
            t += kOdd[rnd(sizeof(kOdd) / sizeof(kOdd[0]))];
        }
        t += odd && rnd(30) == 0 ? (rnd(2) ? "\r\n" : "\r") : "\n";
    }
    if (odd && rnd(4) == 0 && !t.empty()) t.pop_back();          // no newline at the end of the file
    return t;
}

static void put(const std::string& path, const std::string& bytes) {
    FILE* f = std::fopen(path.c_str(), "wb");
    if (!f) std::abort();
    std::fwrite(bytes.data(), 1, bytes.size(), f);
    std::fclose(f);
}

static size_t python_lines(const std::string& text) {          // universal newlines
    size_t n = 0;
    bool open = false;
    for (size_t i = 0; i < text.size(); ++i) {
        open = true;
        if (text[i] == '\n' || (text[i] == '\r' && !(i + 1 < text.size() && text[i + 1] == '\n'))) {
            ++n;
            open = false;
        }
    }
    return n + (open ? 1 : 0);
}

static long failures = 0;
#define EXPECT(cond, what)                                                                   \
    do {                                                                                     \
        if (!(cond)) {                                                                       \
            std::fprintf(stderr, "CHECK FAILED line %d: %s (%s)\n", __LINE__, #cond, what); \
            ++failures;                                                                      \
        }                                                                                    \
    } while (0)

static std::string inflate_all(const std::string& path, int threads, int* rc_out) {
    fhx_text* x = nullptr;
    const int rc = fhx_host_inflate(path.c_str(), threads, &x);
    std::string text;
    if (rc == FHX_OK) {
        text.resize((size_t)fhx_text_bytes(x));
        EXPECT(fhx_text_copy(x, text.empty() ? nullptr : &text[0], (int64_t)text.size()) == FHX_OK, "text copy");
        // the parts a sharded run cuts the text into: they tile it, and each starts a row
        const int n_parts = 1 + (int)rnd(9);
        int64_t end = 0;
        for (int k = 0; k < n_parts; ++k) {
            int64_t lo = -1, hi = -1;
            EXPECT(fhx_text_part_bounds(x, k, n_parts, &lo, &hi) == FHX_OK, "part bounds");
            EXPECT(lo == end && hi >= lo && hi <= (int64_t)text.size(), "parts follow each other");
            EXPECT(lo == 0 || lo == (int64_t)text.size() || text[(size_t)lo - 1] == '\n', "a part starts a row");
            end = hi;
        }
        EXPECT(end == (int64_t)text.size(), "parts cover the text");
        int64_t lo = 0, hi = 0;
        EXPECT(fhx_text_part_bounds(x, n_parts, n_parts, &lo, &hi) == FHX_ERR_ARG, "part out of range");
    } else {
        EXPECT(x == nullptr || std::strlen(fhx_text_error(x)) > 0, "an error carries a message");
    }
    fhx_text_free(x);
    *rc_out = rc;
    return text;
}

// The stream inflated in parts (fhx_host_inflate_part .. fhx_text_part_resolve), as sharded._ingest_stream_parts drives it: the
// windows chained from the tails, the parts' CRC-32s combined.  true + text: every part accepted AND the combined CRC-32 and length
// are the trailer's (what the caller requires before it believes the text).
static bool inflate_parts(const std::string& path, const std::string& file_bytes, int n_parts, int threads, std::string* text) {
    std::vector<fhx_text_part*> parts((size_t)n_parts, nullptr);
    bool ok = true;
    for (int r = 0; r < n_parts && ok; ++r) {
        const int rc = fhx_host_inflate_part(path.c_str(), threads, r, n_parts, &parts[(size_t)r]);
        if (rc != FHX_OK) {
            EXPECT(rc == FHX_ERR_UNSUPPORTED || rc == FHX_ERR_REFERENCE_EXIT || rc == FHX_ERR_ARG || rc == FHX_ERR_NOMEM, "part error code");
            EXPECT(parts[(size_t)r] == nullptr || std::strlen(fhx_text_part_error(parts[(size_t)r])) > 0, "a refused part carries a message");
            ok = false;
        }
    }
    std::vector<uint8_t> window(32768, 0);
    uint32_t crc = 0;
    int64_t total = 0;
    text->clear();
    for (int r = 0; r < n_parts && ok; ++r) {
        std::vector<uint16_t> tail(32768);
        EXPECT(fhx_text_part_tail(parts[(size_t)r], tail.data(), 32768) == FHX_OK, "tail");
        EXPECT(fhx_text_part_tail(parts[(size_t)r], tail.data(), 100) == FHX_ERR_ARG, "tail buffer too small");
        fhx_text* x = nullptr;
        uint32_t c = 0;
        const int rc = fhx_text_part_resolve(parts[(size_t)r], window.data(), 32768, &x, &c);
        if (rc != FHX_OK) {
            ok = false;
            fhx_text_free(x);
            break;
        }
        std::string piece((size_t)fhx_text_bytes(x), '\0');
        EXPECT(fhx_text_copy(x, piece.empty() ? nullptr : &piece[0], (int64_t)piece.size()) == FHX_OK, "part text copy");
        const int64_t first = fhx_text_first_row_end(x, nullptr, 0);
        const size_t nl = piece.find('\n');
        EXPECT(first == (nl == std::string::npos ? -1 : (int64_t)nl + 1), "first row end");
        EXPECT((fhx_text_ends_with_newline(x) != 0) == (!piece.empty() && piece.back() == '\n'), "ends with newline");
        fhx_text_free(x);
        crc = r ? fhx_crc32_combine(crc, c, (int64_t)piece.size()) : c;
        total += (int64_t)piece.size();
        *text += piece;
        std::vector<uint8_t> next(32768);
        for (int j = 0; j < 32768; ++j) next[(size_t)j] = tail[(size_t)j] < 256 ? (uint8_t)tail[(size_t)j] : window[(size_t)(tail[(size_t)j] - 256)];
        window.swap(next);
    }
    for (fhx_text_part* p : parts) fhx_text_part_free(p);
    if (!ok || file_bytes.size() < 8) return false;
    const unsigned char* t = (const unsigned char*)file_bytes.data() + file_bytes.size() - 8;
    const uint32_t want_crc = t[0] | ((uint32_t)t[1] << 8) | ((uint32_t)t[2] << 16) | ((uint32_t)t[3] << 24);
    const uint32_t want_n = t[4] | ((uint32_t)t[5] << 8) | ((uint32_t)t[6] << 16) | ((uint32_t)t[7] << 24);
    return crc == want_crc && (uint32_t)total == want_n;
}

int main(int argc, char** argv) {
    if (argc < 3) return 2;
    const std::string dir = argv[1];
    const long cases = std::atol(argv[2]);
    rng.seed(argc > 3 ? (uint64_t)std::atoll(argv[3]) : 1);
    const std::string path = dir + "/case.gz";
    long accepted = 0, refused = 0, routes = 0, part_routes = 0;
    for (long it = 0; it < cases; ++it) {
        const int kind = (int)rnd(3);
        const bool odd = rnd(3) != 0;
        const size_t rows = rnd(8) == 0 ? (size_t)(20000 + rnd(60000)) : (size_t)rnd(400);
        const std::string text = table_text(kind, rows, odd);
        // container: one member, several members, zero padding, or members that carry their size ("FH", this library's writers)
        std::string gz;
        const int shape = (int)rnd(6);
        if (shape <= 2) {
            gz = gz_member(text, (int)rnd(10), (int)rnd(5));
        } else if (shape == 3) {
            size_t at = 0;
            while (at < text.size() || gz.empty()) {
                const size_t n = std::min<size_t>(text.size() - at, 1 + rnd(text.size() + 1));
                gz += gz_member(text.substr(at, n), (int)rnd(10), (int)rnd(5));
                at += n;
                if (rnd(3) == 0) gz += std::string(rnd(9), '\0');
                if (text.empty()) break;
            }
        } else if (shape == 4) {
            gz = gz_member(text, 1 + (int)rnd(9), 0) + std::string(rnd(40), '\0');
        } else {
            size_t at = 0;
            do {
                const size_t n = std::min<size_t>(text.size() - at, 1 + rnd(text.size() / 2 + 1));
                gz += fh_member(text.substr(at, n), 1 + (int)rnd(9));
                at += n;
            } while (at < text.size());
        }
        // mutations
        std::string bytes = gz;
        const int mut = (int)rnd(8);
        if (mut == 1 && !bytes.empty()) bytes.resize(rnd(bytes.size()));
        if (mut == 2 && !bytes.empty())
            for (int k = 0, n = 1 + (int)rnd(4); k < n; ++k) bytes[rnd(bytes.size())] ^= (char)(1u << rnd(8));
        if (mut == 3) bytes += std::string(1 + rnd(30), (char)rnd(256));
        if (mut == 4 && bytes.size() > 20) bytes.erase(rnd(bytes.size() - 10), 1 + rnd(9));
        if (mut == 5) bytes.insert(rnd(bytes.size() + 1), std::string(1 + rnd(9), (char)rnd(256)));
        put(path, bytes);
        // small streams through the all-core gunzip too, in small chunks
        if (rnd(2)) {
            setenv("FHX_PGUNZIP_MIN", "0", 1);
            setenv("FHX_PGUNZIP_CHUNK", rnd(2) ? "2048" : "30000", 1);
        } else {
            unsetenv("FHX_PGUNZIP_MIN");
            unsetenv("FHX_PGUNZIP_CHUNK");
        }
        const int threads = 1 + (int)rnd(6);
        fhx_table* t = nullptr;
        const int rc = fhx_host_read_table(path.c_str(), kind, threads, &t);
        if (rc == FHX_OK) {
            ++accepted;
            EXPECT(t != nullptr && std::strlen(fhx_table_error(t)) == 0, "accepted file without an error text");
            const int64_t n = fhx_table_rows(t);
            if (mut == 0) EXPECT((size_t)n == python_lines(text), "rows == lines of the text");
            std::vector<int32_t> col((size_t)std::max<int64_t>(n, 1));
            std::vector<double> dv((size_t)std::max<int64_t>(n, 1));
            for (int c = 0; c < 5; ++c) (void)fhx_table_copy(t, c, col.data());
            (void)fhx_table_copy(t, 5, dv.data());
            for (int i = 0; i < fhx_table_n_names(t); ++i) EXPECT(fhx_table_name(t, i) != nullptr, "names");
        } else {
            ++refused;
            EXPECT(rc == FHX_ERR_REFERENCE_EXIT || rc == FHX_ERR_UNSUPPORTED || rc == FHX_ERR_ARG || rc == FHX_ERR_NOMEM, "error code");
            EXPECT(t == nullptr || std::strlen(fhx_table_error(t)) > 0, "a refusal carries a message");
            if (mut == 0 && !odd) EXPECT(false, "a clean table was refused");
        }
        fhx_table_free(t);
        // the inflate routes agree: all cores vs zlib on one thread
        if (rnd(3) == 0) {
            int rc_a = 0, rc_b = 0;
            setenv("FHX_PGUNZIP_MIN", "0", 1);
            setenv("FHX_PGUNZIP_CHUNK", "4096", 1);
            const std::string a = inflate_all(path, 4, &rc_a);
            setenv("FHX_SERIAL_GUNZIP", "1", 1);
            const std::string b = inflate_all(path, 1, &rc_b);
            unsetenv("FHX_SERIAL_GUNZIP");
            EXPECT((rc_a == FHX_OK) == (rc_b == FHX_OK), "both inflate routes accept or both refuse");
            if (rc_a == FHX_OK && rc_b == FHX_OK) EXPECT(a == b, "both inflate routes return the same bytes");
            if (rc_a == FHX_OK && mut == 0) EXPECT(a == text, "inflated text == the text that was compressed");
            // ... and the stream inflated in parts: whatever it accepts (with its CRC-32 and length check) is zlib's text
            std::string c;
            if (inflate_parts(path, bytes, 1 + (int)rnd(5), 1 + (int)rnd(4), &c)) {
                EXPECT(rc_b == FHX_OK && c == b, "the parts route returns zlib's bytes");
                ++part_routes;
            }
            ++routes;
        }
    }
    // the host fit on random histograms (binning, possible pairs, FITPACK, PAVA): inputs from a file are integers, but the
    // combinations (few distances, one bin, huge counts, unmappable loci, a distance range without loci) are the caller's
    long fits = 0, fit_refused = 0;
    for (long it = 0; it < cases / 4 + 8; ++it) {
        fhx::FragTable ft;
        const int n_chr = 1 + (int)rnd(4);
        const int64_t res = (int64_t)(1 + rnd(4)) * 5000;
        int64_t longest = 1;
        for (int c = 0; c < n_chr; ++c) {
            const int64_t n = (int64_t)rnd(3000);
            std::vector<int32_t> mids;
            for (int64_t i = 0; i < n; ++i)
                if (rnd(50) != 0) mids.push_back((int32_t)(i * res + res / 2));
            ft.chr_id.push_back(c);
            ft.n_mappable.push_back((int64_t)mids.size());
            ft.max_mid.push_back(mids.empty() ? 0 : mids.back());
            ft.mids.push_back(mids);
            longest = std::max<int64_t>(longest, n);
        }
        std::vector<int64_t> cc((size_t)longest + 1, 0), np((size_t)longest + 1, 0);
        fhx::PassInputs in;
        in.resolution = res;
        in.dist_low = rnd(3) ? (int64_t)rnd(5) * res : 0;
        in.dist_up = rnd(3) ? in.dist_low + (int64_t)rnd((uint64_t)longest + 1) * res : INT64_MAX;
        in.n_bins = 1 + (int32_t)rnd(rnd(2) ? 120 : 8);
        in.mode = (int32_t)rnd(3);
        int64_t sum = 0;
        for (int64_t d = 0; d <= longest; ++d) {
            const int64_t dist = d * res;
            if (dist < in.dist_low || dist > in.dist_up || rnd(6) == 0) continue;
            const int64_t rows = (int64_t)rnd(1000);
            if (!rows) continue;
            np[(size_t)d] = rows;
            cc[(size_t)d] = rows * (int64_t)(1 + rnd(rnd(10) == 0 ? 2000000 : 6)) / (1 + d / 4);
            sum += cc[(size_t)d];
        }
        in.hist_sumcc = cc.data();
        in.hist_npairs = np.data();
        in.n_dist = (int64_t)cc.size();
        in.in_range_sum = sum;
        in.inter_count = (int64_t)rnd(1000);
        in.inter_sum = in.inter_count * 2;
        fhx::PassFit fit;
        std::string err;
        const int rc = fhx::run_host_pass(in, ft, fit, err);
        if (rc == FHX_OK) {
            ++fits;
            EXPECT(fit.table_y.size() == fit.table_x.size(), "one table value per table distance");
            for (double v : fit.table_y) EXPECT(!(v != v), "no NaN in the isotonic table");
        } else {
            ++fit_refused;
            EXPECT(!err.empty(), "a refused fit says why");
        }
    }
    std::printf("io_sanitize: %ld table files (%ld accepted, %ld refused), %ld route comparisons (%ld through the parts route), %ld fits (%ld refused), %ld check failures\n",
                cases, accepted, refused, routes, part_routes, fits, fit_refused, failures);
    return failures ? 1 : 0;
}
