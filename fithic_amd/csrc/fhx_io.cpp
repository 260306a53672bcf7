// fhx_io.cpp - native ingest / emit of the reference's gzip text formats (51 % of the
// reference's wall time is text I/O).  Host-only code, no GPU involved:
//   writer  <- the output loop of fit_Spline          (fithic/fithic.py:1167-1220): "%s\t%d\t%s\t%d\t%d\t%e\t%e\t%e\t%e\t%f\n"
//   reader  <- the three gzip readers                 (fithic/fithic.py:406-417, :581-590, :805-808, :818-821)
// Rows are formatted and deflated in parallel, one gzip member per block of rows (a concatenation of gzip members is a
// valid gzip file: zcat, Python's gzip module and the reference's own gzip.open read it as one stream).  Every member the
// writers emit carries its own compressed size in a gzip extra subfield ("FH", 8 bytes - the idea of BGZF's "BC"), so a
// reader can find all member starts without inflating anything; the reader inflates such files - and bgzip output - on all
// cores.  A plain single-member .gz (what `gzip` writes) has one deflate stream and is inflated by one thread.
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <zlib.h>

#include <charconv>
#include <chrono>
#include <condition_variable>
#include <mutex>

#include <algorithm>
#include <atomic>
#include <cerrno>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <unordered_map>
#include <vector>

#include "../../include/fithic_mi355x.h"
#include "fhx_cpus.hpp"
#include "fhx_io_internal.hpp"

namespace {

// Python's '%e' / '%f' of a float: C's printf with the same precision, except that Python prints nan/inf without sign quirks
inline int put_e(char* dst, double v) {
    if (std::isnan(v)) {
        std::memcpy(dst, "nan", 3);
        return 3;
    }
    if (std::isinf(v)) {
        const char* s = v > 0 ? "inf" : "-inf";
        const int n = v > 0 ? 3 : 4;
        std::memcpy(dst, s, n);
        return n;
    }
    // == printf("%e"): both are the correctly rounded 6-digit decimal expansion with an exponent of at least two digits
    // (checked against snprintf on 2e7 random doubles incl. subnormals); ~8x faster than the locale-aware printf path
    return (int)(std::to_chars(dst, dst + 32, v, std::chars_format::scientific, 6).ptr - dst);
}
inline int put_f(char* dst, double v) {
    if (std::isnan(v)) {
        std::memcpy(dst, "nan", 3);
        return 3;
    }
    if (std::isinf(v)) {
        const char* s = v > 0 ? "inf" : "-inf";
        const int n = v > 0 ? 3 : 4;
        std::memcpy(dst, s, n);
        return n;
    }
    return (int)(std::to_chars(dst, dst + 400, v, std::chars_format::fixed, 6).ptr - dst);
}
inline int put_int(char* dst, long long v) {
    char tmp[24];
    int n = 0;
    unsigned long long u = v < 0 ? 0ull - (unsigned long long)v : (unsigned long long)v;
    do {
        tmp[n++] = char('0' + u % 10);
        u /= 10;
    } while (u);
    int k = 0;
    if (v < 0) dst[k++] = '-';
    while (n) dst[k++] = tmp[--n];
    return k;
}

// one gzip member holding `text`; its total size goes into the "FH" extra subfield (patched after deflate: the header is not
// covered by the CRC)
bool deflate_member(const std::string& text, int level, std::string& out) {
    z_stream zs;
    std::memset(&zs, 0, sizeof(zs));
    if (deflateInit2(&zs, level, Z_DEFLATED, 15 + 16, 8, Z_DEFAULT_STRATEGY) != Z_OK) return false;   // 15+16: gzip wrapper
    unsigned char extra[12] = {'F', 'H', 8, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    gz_header hd;
    std::memset(&hd, 0, sizeof(hd));
    hd.os = 255;
    hd.extra = extra;
    hd.extra_len = sizeof(extra);
    if (deflateSetHeader(&zs, &hd) != Z_OK) {
        deflateEnd(&zs);
        return false;
    }
    out.resize(deflateBound(&zs, (uLong)text.size()) + 64);
    zs.next_in = (Bytef*)text.data();
    zs.avail_in = (uInt)text.size();
    zs.next_out = (Bytef*)&out[0];
    zs.avail_out = (uInt)out.size();
    const int rc = deflate(&zs, Z_FINISH);
    out.resize(zs.total_out);
    deflateEnd(&zs);
    if (rc != Z_STREAM_END || out.size() < 24) return false;
    const unsigned long long total = out.size();           // fixed header 10 B, XLEN 2 B, subfield id + length 4 B, then the value
    for (int k = 0; k < 8; ++k) out[16 + k] = (char)((total >> (8 * k)) & 0xFF);
    return true;
}

// blocks 0..n_blocks-1 produced by `produce(block, scratch, out)` on n_threads workers, written to `f` in block order while
// later blocks are still being produced; at most `window` finished blocks wait in memory.  Every buffer is reused: a worker
// keeps its text scratch, the window keeps its output strings - fresh multi-megabyte allocations per block (mmap, page
// faults, munmap under one address-space lock) were what 256 threads spent their time on (15.7 s -> see profiles/ for C3).
template <typename Produce>
bool ordered_parallel_write(std::FILE* f, int64_t n_blocks, int n_threads, Produce produce) {
    if (n_blocks <= 0) return true;
    n_threads = (int)std::max<int64_t>(1, std::min<int64_t>(n_threads, n_blocks));
    const int64_t window = 2 * (int64_t)n_threads;
    std::vector<std::string> slot((size_t)window);
    std::vector<char> ready((size_t)window, 0);
    std::mutex mu;
    std::condition_variable cv_ready, cv_space;
    std::atomic<int64_t> next{0};
    int64_t written = 0;                                   // guarded by mu
    std::atomic<bool> fine{true};
    auto worker = [&]() {
        std::string scratch;
        for (;;) {
            const int64_t b = next.fetch_add(1);
            if (b >= n_blocks || !fine) return;
            {
                std::unique_lock<std::mutex> lk(mu);
                cv_space.wait(lk, [&] { return b < written + window || !fine; });
            }
            if (!fine) return;
            std::string& out = slot[(size_t)(b % window)];   // free: block b - window has been written
            if (!produce(b, scratch, out)) fine = false;
            {
                std::lock_guard<std::mutex> lk(mu);
                ready[(size_t)(b % window)] = 1;
            }
            cv_ready.notify_all();
        }
    };
    std::vector<std::thread> pool;
    for (int k = 0; k < n_threads; ++k) pool.emplace_back(worker);
    bool ok = true;
    for (int64_t b = 0; b < n_blocks; ++b) {
        {
            std::unique_lock<std::mutex> lk(mu);
            cv_ready.wait(lk, [&] { return ready[(size_t)(b % window)] || !fine; });
            if (!ready[(size_t)(b % window)]) {
                ok = false;
                break;
            }
        }
        const std::string& z = slot[(size_t)(b % window)];
        if (!z.empty() && std::fwrite(z.data(), 1, z.size(), f) != z.size()) {
            ok = false;
            fine = false;
        }
        {
            std::lock_guard<std::mutex> lk(mu);
            ready[(size_t)(b % window)] = 0;
            written = b + 1;
        }
        cv_space.notify_all();
        if (!ok) break;
    }
    if (!ok) fine = false;
    cv_space.notify_all();
    cv_ready.notify_all();
    for (auto& t : pool) t.join();
    return ok && fine;
}

}  // namespace

extern "C" {

int fhx_host_write_significances(const char* path, const char* const* chr_names, int32_t n_names, const int32_t* chr1,
                                 const int32_t* mid1, const int32_t* chr2, const int32_t* mid2, const int32_t* count,
                                 const double* p, const double* q, const double* bias1, const double* bias2,
                                 const double* expcc, int64_t n_rows, int32_t mode, int64_t dist_low, int64_t dist_up,
                                 int32_t gzip_level, int32_t n_threads, int64_t* rows_written) {
    if (!path || !chr_names || n_names <= 0 || n_rows < 0) return FHX_ERR_ARG;
    if (n_rows > 0 && (!chr1 || !mid1 || !chr2 || !mid2 || !count || !p || !q || !bias1 || !bias2 || !expcc)) return FHX_ERR_ARG;
    if (gzip_level < 0 || gzip_level > 9) gzip_level = 6;
    if (n_threads <= 0) n_threads = fhx::usable_cpus();
    std::FILE* f = std::fopen(path, "wb");
    if (!f) return FHX_ERR_ARG;
    const bool all_reg = mode == FHX_MODE_ALL, inter_only = mode == FHX_MODE_INTER_ONLY;
    const int64_t block = 1 << 16;                                   // rows per gzip member
    const int64_t n_blocks = (n_rows + block - 1) / block;
    std::vector<size_t> name_len(n_names);
    for (int i = 0; i < n_names; ++i) name_len[i] = std::strlen(chr_names[i]);
    bool ok = true;
    for (int i = 0; i < n_names; ++i)
        if (name_len[i] > 256) {                                         // rows are assembled in a fixed buffer
            std::fclose(f);
            return FHX_ERR_ARG;
        }
    {
        std::string hdr = "chr1\tfragmentMid1\tchr2\tfragmentMid2\tcontactCount\tp-value\tq-value\tbias1\tbias2\tExpCC\n", z;
        ok = deflate_member(hdr, gzip_level, z) && std::fwrite(z.data(), 1, z.size(), f) == z.size();
    }
    std::vector<int64_t> rows((size_t)n_blocks, 0);
    auto produce = [&](int64_t blk, std::string& text, std::string& zipped) -> bool {
        const int64_t lo = blk * block, hi = std::min(n_rows, lo + block);
        text.clear();
        zipped.clear();
        if (text.capacity() < (size_t)(hi - lo) * 110) text.reserve((size_t)(hi - lo) * 110);
        char buf[1400];                                                   // 2 names <= 512, 3 ints <= 36, 4 x %e <= 100, %f <= 320
        for (int64_t i = lo; i < hi; ++i) {
            const bool inter = chr1[i] != chr2[i];
            bool emit;
            if (inter) {
                emit = all_reg || inter_only;                                    // fithic.py:1197
            } else {
                const int64_t d = std::llabs((long long)mid1[i] - (long long)mid2[i]);
                emit = (all_reg || !inter_only) && d >= dist_low && d <= dist_up;  // fithic.py:1205-1207
            }
            if (!emit) continue;
            if (chr1[i] < 0 || chr1[i] >= n_names || chr2[i] < 0 || chr2[i] >= n_names) return false;
            int n = 0;
            std::memcpy(buf + n, chr_names[chr1[i]], name_len[chr1[i]]);
            n += (int)name_len[chr1[i]];
            buf[n++] = '\t';
            n += put_int(buf + n, mid1[i]);
            buf[n++] = '\t';
            std::memcpy(buf + n, chr_names[chr2[i]], name_len[chr2[i]]);
            n += (int)name_len[chr2[i]];
            buf[n++] = '\t';
            n += put_int(buf + n, mid2[i]);
            buf[n++] = '\t';
            n += put_int(buf + n, count[i]);
            buf[n++] = '\t';
            n += put_e(buf + n, p[i]);
            buf[n++] = '\t';
            n += put_e(buf + n, q[i]);
            buf[n++] = '\t';
            n += put_e(buf + n, bias1[i]);
            buf[n++] = '\t';
            n += put_e(buf + n, bias2[i]);
            buf[n++] = '\t';
            n += put_f(buf + n, expcc[i]);
            buf[n++] = '\n';
            text.append(buf, (size_t)n);
            ++rows[(size_t)blk];
        }
        return text.empty() || deflate_member(text, gzip_level, zipped);
    };
    if (ok) ok = ordered_parallel_write(f, n_blocks, n_threads, produce);
    int64_t written = 0;
    for (int64_t v : rows) written += v;
    if (std::fclose(f) != 0) ok = false;
    if (rows_written) *rows_written = written;
    return ok ? FHX_OK : FHX_ERR_ARG;
}

}  // extern "C"

// =====================================================================================================================
// reader
// =====================================================================================================================
namespace {

struct Chunk {
    std::vector<std::string> names;
    std::unordered_map<std::string, int32_t> index;
    std::vector<int32_t> ci[2], mi[2], iv;
    std::vector<double> dv;
    int64_t bad_line = -1;                        // chunk-relative
    bool bad_is_range = false;                    // the bad line is well-formed for the reference; a number does not fit int32 here
    int64_t n_lines = 0;
    int32_t last_id[2] = {-1, -1};                // id of the previous line's name, per column
};

}  // namespace

// the parsed file: per-range chunks in file order; fhx_table_copy gathers a column from them on all cores, straight into the
// caller's array (no merged intermediate copy)
struct fhx_table {
    int kind = 0;
    bool has_float = true;                       // contacts read with FHX_TABLE_NO_FLOAT keep no column 5
    std::vector<std::string> names;
    std::vector<Chunk> chunks;
    std::vector<std::vector<int32_t>> remap;     // chunk-local chromosome id -> file-wide id (order of first appearance)
    std::vector<size_t> row0;                    // first row of every chunk, then the total
    std::string error;
};

namespace {

// str.split() of a line read in text mode: ASCII whitespace as Python's str sees it (\x1c-\x1f are separators too; \n ends the
// line, and so does a lone \r: text mode's universal newlines turn \r and \r\n into \n before the reference sees the line).
// Deviation, documented in INTEGRATION.md: non-ASCII whitespace (NBSP, U+2000...) is not a separator here.
inline bool is_space(char c) { return c == ' ' || (c >= '\t' && c <= '\r') || (c >= '\x1c' && c <= '\x1f'); }

// Python's int() / float() grammar on an ASCII token: underscores are allowed singly BETWEEN digits (PEP 515).  Copies the token
// without them; false if an underscore is misplaced, or the token holds a non-ASCII byte (Python accepts any Unicode decimal
// digit there - not supported: such a line is refused, which the caller reports).
inline bool strip_underscores(const char* b, const char* e, char* dst, size_t cap, size_t& n) {
    n = 0;
    for (const char* p = b; p < e; ++p) {
        const unsigned char ch = (unsigned char)*p;
        if (ch >= 0x80) return false;
        if (ch == '_') {
            const bool left = p > b && p[-1] >= '0' && p[-1] <= '9', right = p + 1 < e && p[1] >= '0' && p[1] <= '9';
            if (!left || !right) return false;
            continue;
        }
        if (n + 1 >= cap) return false;
        dst[n++] = (char)ch;
    }
    dst[n] = 0;
    return true;
}

// `range` is set when the token IS an integer for Python's int() (which has no upper limit) but does not fit the int32 columns
inline bool parse_i32_plain(const char* b, const char* e, int32_t& out, bool& range) {      // optional sign, ASCII digits
    if (b == e) return false;
    bool neg = false;
    if (*b == '+' || *b == '-') {
        neg = *b == '-';
        ++b;
    }
    if (b == e) return false;
    long long v = 0;
    bool big = false;
    for (; b < e; ++b) {
        if (*b < '0' || *b > '9') return false;
        if (!big) v = v * 10 + (*b - '0');
        if (v > 4000000000ll) big = true;
    }
    if (neg) v = -v;
    if (big || v < INT32_MIN || v > INT32_MAX) {
        range = true;
        return false;
    }
    out = (int32_t)v;
    return true;
}

inline bool parse_i32(const char* b, const char* e, int32_t& out, bool& range) {      // Python int(): optional sign, digits, single '_' between digits
    if (parse_i32_plain(b, e, out, range)) return true;
    if (range || std::memchr(b, '_', (size_t)(e - b)) == nullptr) return false;
    char tmp[64];
    size_t n = 0;
    if (!strip_underscores(b, e, tmp, sizeof(tmp), n)) {
        // longer than tmp and nothing but digits and well-placed underscores: an integer of more than 60 digits
        if (e - b >= (long)sizeof(tmp)) {
            bool digits = true;
            for (const char* p = b; p < e && digits; ++p)
                digits = (*p >= '0' && *p <= '9') || (*p == '_' && p > b && p + 1 < e && p[-1] != '_' && p[1] != '_') ||
                         (p == b && (*p == '+' || *p == '-'));
            range = digits;
        }
        return false;
    }
    return parse_i32_plain(tmp, tmp + n, out, range);
}

inline bool parse_f64(const char* b, const char* e, double& out) {       // Python float()
    // counts are nearly always plain digits: up to 15 of them are exact in a double and need no strtod
    if (e - b >= 1 && e - b <= 15) {
        long long v = 0;
        const char* q = b;
        for (; q < e && *q >= '0' && *q <= '9'; ++q) v = v * 10 + (*q - '0');
        if (q == e) {
            out = (double)v;
            return true;
        }
    }
    // the rest through strtod, restricted to what float() takes: no hexadecimal form, no "nan(...)", underscores by PEP 515
    char tmp[400];
    size_t n = 0;
    if (!strip_underscores(b, e, tmp, sizeof(tmp), n) || n == 0) return false;
    for (size_t i = 0; i < n; ++i)
        if (tmp[i] == 'x' || tmp[i] == 'X' || tmp[i] == '(' || tmp[i] == 'p' || tmp[i] == 'P') return false;
    char* end = nullptr;
    out = std::strtod(tmp, &end);
    return end == tmp + n;
}

void parse_chunk(const char* b, const char* e, int kind, bool keep_float, Chunk& c) {
    const char* fld_b[8];
    const char* fld_e[8];
    {   // one allocation per column instead of a dozen doublings: a contacts line is >= 16 bytes, the others >= 8
        const size_t est = (size_t)(e - b) / (kind == 0 ? 16 : 8) + 16;
        c.ci[0].reserve(est);
        c.mi[0].reserve(est);
        if (kind == 0) {
            c.ci[1].reserve(est);
            c.mi[1].reserve(est);
        }
        if (kind != 2) c.iv.reserve(est);
        if (kind == 2 || (kind == 0 && keep_float)) c.dv.reserve(est);
    }
    while (b < e) {
        const char* nl = (const char*)std::memchr(b, '\n', (size_t)(e - b));
        const char* le = nl ? nl : e;
        // universal newlines (gzip.open(..., 'rt'), fithic.py:406): a \r that is not the first half of \r\n ends the line too
        const char* next = nl ? nl + 1 : e;
        if (const char* cr = (const char*)std::memchr(b, '\r', (size_t)(le - b))) {
            if (cr + 1 < le || !nl) {
                le = cr;
                next = cr + 1;
            }
        }
        bool range = false;
        int nf = 0, total = 0;
        const char* p = b;
        while (p < le) {
            while (p < le && is_space(*p)) ++p;
            if (p >= le) break;
            const char* s = p;
            while (p < le && !is_space(*p)) ++p;
            if (nf < 8) {
                fld_b[nf] = s;
                fld_e[nf] = p;
                ++nf;
            }
            ++total;
        }
        bool ok = true;
        // sorted contact files repeat the chromosome of the previous line: compare with the last name of each column first
        auto intern = [&](const char* s, const char* t, int col) -> int32_t {
            const size_t len = (size_t)(t - s);
            if (c.last_id[col] >= 0) {
                const std::string& last = c.names[(size_t)c.last_id[col]];
                if (last.size() == len && std::memcmp(last.data(), s, len) == 0) return c.last_id[col];
            }
            std::string key(s, len);
            auto it = c.index.find(key);
            int32_t id;
            if (it != c.index.end()) {
                id = it->second;
            } else {
                id = (int32_t)c.names.size();
                c.names.push_back(key);
                c.index.emplace(std::move(key), id);
            }
            c.last_id[col] = id;
            return id;
        };
        if (kind == 0) {                                   // ch1 mid1 ch2 mid2 count: exactly 5 fields (fithic.py:413)
            int32_t m1, m2;
            double raw;
            // every field is checked even after one has failed: a line with an int32 overflow AND a malformed field is malformed
            bool r1 = false, r2 = false;
            const bool ok1 = total == 5 && parse_i32(fld_b[1], fld_e[1], m1, r1), ok2 = total == 5 && parse_i32(fld_b[3], fld_e[3], m2, r2);
            const bool ok3 = total == 5 && parse_f64(fld_b[4], fld_e[4], raw);
            ok = ok1 && ok2 && ok3;
            range = total == 5 && ok3 && (ok1 || r1) && (ok2 || r2) && std::isfinite(raw);     // int(float) of nan / inf raises in Python
            if (ok) {
                const double tr = std::trunc(raw);
                ok = tr >= INT32_MIN && tr <= INT32_MAX;    // NaN fails both comparisons: int(float('nan')) raises in Python
                if (ok) {
                    c.ci[0].push_back(intern(fld_b[0], fld_e[0], 0));
                    c.mi[0].push_back(m1);
                    c.ci[1].push_back(intern(fld_b[2], fld_e[2], 1));
                    c.mi[1].push_back(m2);
                    c.iv.push_back((int32_t)tr);
                    if (keep_float) c.dv.push_back(raw);
                }
            }
        } else if (kind == 1) {                            // words[0], int(words[2]), int(words[3]) (fithic.py:583-586)
            int32_t mid, hits;
            bool r1 = false, r2 = false;
            const bool ok1 = total >= 4 && parse_i32(fld_b[2], fld_e[2], mid, r1), ok2 = total >= 4 && parse_i32(fld_b[3], fld_e[3], hits, r2);
            ok = ok1 && ok2;
            range = total >= 4 && (ok1 || r1) && (ok2 || r2);
            if (ok) {
                c.ci[0].push_back(intern(fld_b[0], fld_e[0], 0));
                c.mi[0].push_back(mid);
                c.iv.push_back(hits);
            }
        } else {                                           // words[0], int(words[1]), float(words[2]) (fithic.py:807-808)
            int32_t mid;
            double bias;
            bool r1 = false;
            const bool ok1 = total >= 3 && parse_i32(fld_b[1], fld_e[1], mid, r1), ok2 = total >= 3 && parse_f64(fld_b[2], fld_e[2], bias);
            ok = ok1 && ok2;
            range = total >= 3 && ok2 && (ok1 || r1);
            if (ok) {
                c.ci[0].push_back(intern(fld_b[0], fld_e[0], 0));
                c.mi[0].push_back(mid);
                c.dv.push_back(bias);
            }
        }
        if (!ok) {
            c.bad_line = c.n_lines;
            c.bad_is_range = range;
            return;
        }
        ++c.n_lines;
        b = next;
    }
}

using Member = fhx::GzMember;

// Walks the gzip members of a file whose members all carry their compressed size: "FH" (8 bytes, this library's writers)
// or "BC" (2 bytes, BGZF: bgzip / htslib).  false: some member has no size field (plain gzip) or the chain does not end
// exactly at the end of the file.
bool scan_members(const unsigned char* d, size_t n, std::vector<Member>& out) {
    out.clear();
    size_t o = 0;
    while (o < n) {
        if (n - o < 18 || d[o] != 0x1f || d[o + 1] != 0x8b || d[o + 2] != 8 || !(d[o + 3] & 4)) return false;
        const size_t xlen = d[o + 10] | ((size_t)d[o + 11] << 8);
        if (o + 12 + xlen > n) return false;
        size_t size = 0;
        for (size_t p = o + 12; p + 4 <= o + 12 + xlen;) {
            const size_t len = d[p + 2] | ((size_t)d[p + 3] << 8);
            if (p + 4 + len > o + 12 + xlen) return false;
            if (d[p] == 'F' && d[p + 1] == 'H' && len == 8) {
                for (int k = 7; k >= 0; --k) size = (size << 8) | d[p + 4 + k];
            } else if (d[p] == 'B' && d[p + 1] == 'C' && len == 2) {
                size = (size_t)(d[p + 4] | ((size_t)d[p + 5] << 8)) + 1;
            }
            p += 4 + len;
        }
        if (size < 18 || size > n - o) return false;
        const unsigned char* tail = d + o + size - 4;
        const size_t isize = tail[0] | ((size_t)tail[1] << 8) | ((size_t)tail[2] << 16) | ((size_t)tail[3] << 24);
        out.push_back(Member{o, size, isize});
        o += size;
    }
    return !out.empty();
}

bool inflate_one(const unsigned char* src, size_t n, char* dst, size_t want) {
    z_stream zs;
    std::memset(&zs, 0, sizeof(zs));
    if (inflateInit2(&zs, 15 + 16) != Z_OK) return false;
    zs.next_in = const_cast<Bytef*>(src);
    zs.avail_in = (uInt)n;
    char dummy = 0;
    zs.next_out = (Bytef*)(want ? dst : &dummy);
    zs.avail_out = (uInt)(want ? want : 1);
    const int rc = inflate(&zs, Z_FINISH);
    const bool ok = rc == Z_STREAM_END && zs.total_out == want;
    inflateEnd(&zs);
    return ok;
}

// any gzip file (one or more members without size fields): one stream, one thread.  A stream that ends early is an error,
// as it is for Python's gzip module ("Compressed file ended before the end-of-stream marker was reached").
bool inflate_stream(const unsigned char* src, size_t n, std::string& text, std::string& err) {
    z_stream zs;
    std::memset(&zs, 0, sizeof(zs));
    if (inflateInit2(&zs, 15 + 16) != Z_OK) {
        err = "zlib failure";
        return false;
    }
    text.clear();
    text.resize(std::max<size_t>(n * 5, 1 << 16));
    size_t at = 0, in_at = 0;
    bool ended = false;
    while (in_at < n) {
        zs.next_in = const_cast<Bytef*>(src + in_at);
        const size_t in_now = std::min<size_t>(n - in_at, 1u << 30);
        zs.avail_in = (uInt)in_now;
        int rc = Z_OK;
        do {
            if (at == text.size()) text.resize(text.size() + text.size() / 2);
            const size_t room = std::min<size_t>(text.size() - at, 1u << 30);
            zs.next_out = (Bytef*)&text[at];
            zs.avail_out = (uInt)room;
            rc = inflate(&zs, Z_NO_FLUSH);
            at += room - zs.avail_out;
            if (rc == Z_STREAM_END) {
                // What Python's gzip module does after a member's trailer (gzip.py _GzipReader.read / _read_gzip_header; the
                // reference reads through gzip.open(..., 'rt'), fithic.py:406): zero padding is skipped; nothing left = end of
                // file; anything else must be the magic of another member, or it raises BadGzipFile("Not a gzipped file").
                ended = true;
                size_t at_in = in_at + (in_now - zs.avail_in);
                while (at_in < n && src[at_in] == 0) ++at_in;
                if (at_in == n) {
                    in_at = n;
                    goto finished;
                }
                if (n - at_in < 2 || src[at_in] != 0x1f || src[at_in + 1] != 0x8b) {
                    inflateEnd(&zs);
                    err = "bytes that are neither zero padding nor another gzip member follow the last member";
                    return false;
                }
                in_at = at_in;                                          // the next member of a concatenation
                inflateReset(&zs);
                ended = false;
                goto next_input;
            }
            if (rc != Z_OK && rc != Z_BUF_ERROR) {
                inflateEnd(&zs);
                err = "corrupt gzip stream";
                return false;
            }
        } while (zs.avail_in > 0 || zs.avail_out == 0);
        in_at += in_now - zs.avail_in;
    next_input:;
    }
finished:
    inflateEnd(&zs);
    if (!ended) {
        err = "gzip stream ended before its end-of-stream marker";
        return false;
    }
    text.resize(at);
    return true;
}

}  // namespace

bool fhx::io_scan_members(const unsigned char* d, size_t n, std::vector<fhx::GzMember>& out) { return scan_members(d, n, out); }

// The compressed file, whole: mapped, so that the inflating threads page it in themselves (a 600 MB read() into a zero-filled
// vector was 0.24 s of one core); read() for what cannot be mapped (a pipe, an empty file, a file system without mmap).
fhx::FileBytes::~FileBytes() {
    if (map) ::munmap(map, n);
}

int fhx::io_read_file(const char* path, fhx::FileBytes& gz, std::string& error) {
    const int fd = ::open(path, O_RDONLY | O_CLOEXEC);
    if (fd < 0) {
        error = std::string("cannot open ") + path;
        return FHX_ERR_ARG;
    }
    struct stat st;
    bool mapped = false;
    if (::fstat(fd, &st) == 0 && S_ISREG(st.st_mode) && st.st_size > 0) {
        void* m = ::mmap(nullptr, (size_t)st.st_size, PROT_READ, MAP_PRIVATE, fd, 0);
        if (m != MAP_FAILED) {
            (void)::madvise(m, (size_t)st.st_size, MADV_WILLNEED);
            gz.map = m;
            gz.p = (const unsigned char*)m;
            gz.n = (size_t)st.st_size;
            mapped = true;
        }
    }
    if (!mapped) {
        unsigned char buf[1 << 16];
        for (;;) {
            const ssize_t got = ::read(fd, buf, sizeof(buf));
            if (got < 0 && errno == EINTR) continue;
            if (got < 0) {
                ::close(fd);
                error = std::string("read error on ") + path;
                return FHX_ERR_ARG;
            }
            if (got == 0) break;
            gz.owned.insert(gz.owned.end(), buf, buf + got);
        }
        gz.p = gz.owned.data();
        gz.n = gz.owned.size();
    }
    ::close(fd);
    if (gz.size() < 18 || gz.data()[0] != 0x1f || gz.data()[1] != 0x8b) {
        error = std::string("not a gzip file: ") + path + " (the reference's gzip.open raises on it)";
        return FHX_ERR_REFERENCE_EXIT;
    }
    return FHX_OK;
}

// The file read whole and inflated into `pieces` (the text, in file order): on n_threads cores when every gzip member carries
// its size ("FH" of this library's writers, "BC" of bgzip), by one thread otherwise.  seconds[0] = read, seconds[1] = inflate.
int fhx::io_inflate_file(const char* path, int n_threads, std::vector<fhx::TextPiece>& pieces, std::string& error, double* seconds) {
    const auto t_begin = std::chrono::steady_clock::now();
    fhx::FileBytes gz;
    {
        const int rc = fhx::io_read_file(path, gz, error);
        if (rc != FHX_OK) return rc;
    }
    if (seconds) seconds[0] = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_begin).count();
    // ---- inflate: all cores when every member carries its size ("FH" of this library's writers, "BC" of bgzip) ------
    pieces.clear();                                        // the text, in order
    std::vector<Member> members;
    if (scan_members(gz.data(), gz.size(), members) && members.size() > 1) {
        const int nt = (int)std::min<size_t>((size_t)n_threads, members.size());
        std::vector<size_t> first((size_t)nt + 1, members.size());
        {
            size_t total = 0, k = 0;
            for (const auto& m : members) total += m.size;
            size_t acc = 0;
            first[0] = 0;
            for (size_t i = 0; i < members.size(); ++i) {          // contiguous ranges of about total / nt compressed bytes
                while (k + 1 < (size_t)nt && acc >= total / nt * (k + 1)) first[++k] = i;
                acc += members[i].size;
            }
            for (++k; k <= (size_t)nt; ++k) first[k] = members.size();
        }
        pieces.resize((size_t)nt);
        std::atomic<long long> bad_member{-1};
        auto work = [&](int k) {
            size_t bytes = 0;
            for (size_t i = first[k]; i < first[k + 1]; ++i) bytes += members[i].isize;
            fhx::TextPiece& text = pieces[(size_t)k];
            if (!text.allocate(bytes)) {
                bad_member = (long long)first[k];
                return;
            }
            size_t at = 0;
            for (size_t i = first[k]; i < first[k + 1]; ++i) {
                if (!inflate_one(gz.data() + members[i].off, members[i].size, text.data() + at, members[i].isize)) {
                    bad_member = (long long)i;
                    return;
                }
                at += members[i].isize;
            }
        };
        std::vector<std::thread> pool;
        for (int k = 1; k < nt; ++k) pool.emplace_back(work, k);
        work(0);
        for (auto& th : pool) th.join();
        if (bad_member >= 0) {
            error = "corrupt gzip member " + std::to_string((long long)bad_member) + " in " + path;
            return FHX_ERR_REFERENCE_EXIT;
        }
    } else if (!std::getenv("FHX_SERIAL_GUNZIP") && fhx::io_parallel_gunzip(gz.data(), gz.size(), n_threads, pieces, error)) {
        error.clear();                                     // one plain stream, inflated on all cores and checked against its trailer
    } else {
        if (std::getenv("FHX_TIMING") && !error.empty()) std::fprintf(stderr, "parallel gunzip of %s not used: %s\n", path, error.c_str());
        error.clear();
        pieces.clear();
        pieces.resize(1);
        std::string err;
        if (!inflate_stream(gz.data(), gz.size(), pieces[0].grown, err)) {
            error = err + " in " + path + " (the reference's gzip module raises on it)";
            return FHX_ERR_REFERENCE_EXIT;
        }
    }
    if (seconds) seconds[1] = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_begin).count() - seconds[0];
    return FHX_OK;
}

// the parse stage of fhx_host_read_table over an inflated text; t_report carries the stage clocks of the stages before it
static int parse_pieces(const std::vector<fhx::TextPiece>& pieces, const char* path, int32_t kind, int32_t n_threads, fhx_table** out,
                        std::string t_report) {
    const bool keep_float = !(kind & FHX_TABLE_NO_FLOAT);
    kind &= ~FHX_TABLE_NO_FLOAT;
    if (!out || kind < 0 || kind > 2 || (!keep_float && kind != 0)) return FHX_ERR_ARG;
    *out = nullptr;
    fhx_table* t = new (std::nothrow) fhx_table();
    if (!t) return FHX_ERR_NOMEM;
    t->kind = kind;
    t->has_float = keep_float;
    *out = t;
    if (n_threads <= 0) n_threads = fhx::usable_cpus();
    const bool timing = std::getenv("FHX_TIMING") != nullptr;          // stage clocks on stderr
    auto t_last = std::chrono::steady_clock::now();
    auto mark = [&](const char* what) {
        if (!timing) return;
        const auto now = std::chrono::steady_clock::now();
        char b[96];
        std::snprintf(b, sizeof(b), " %s %.3f s;", what, std::chrono::duration<double>(now - t_last).count());
        t_report += b;
        t_last = now;
    };
    // ---- parse ranges, in file order: lines that straddle two pieces are glued, the rest is cut on newlines -------------
    struct Range {
        const char *b, *e;
    };
    std::vector<Range> ranges;
    std::vector<std::string> glued;                        // storage of the straddling lines
    glued.reserve(pieces.size() + 1);
    {
        std::string carry;
        size_t total_text = 0;
        for (const fhx::TextPiece& text : pieces) total_text += text.size();
        // ~8 MB of text per parse task; a small table (fragments, biases: 10-20 MB) is still cut into a few tasks per thread
        const size_t target = std::min<size_t>(8u << 20, std::max<size_t>(256u << 10, total_text / (size_t)(4 * std::max(n_threads, 1))));
        for (const fhx::TextPiece& text : pieces) {
            const char* b = text.data();
            const char* e = b + text.size();
            const char* first_nl = (const char*)std::memchr(b, '\n', text.size());
            if (!first_nl) {
                carry.append(text.data(), text.size());
                continue;
            }
            const char* mid_b = b;
            if (!carry.empty()) {
                carry.append(b, (size_t)(first_nl + 1 - b));
                glued.push_back(std::move(carry));
                carry.clear();
                ranges.push_back(Range{glued.back().data(), glued.back().data() + glued.back().size()});
                mid_b = first_nl + 1;
            }
            const char* last_nl = e;
            while (last_nl > mid_b && last_nl[-1] != '\n') --last_nl;        // one past the last newline
            for (const char* p = mid_b; p < last_nl;) {
                const char* q = p + target < last_nl ? p + target : last_nl;
                if (q < last_nl) {
                    const char* nl = (const char*)std::memchr(q, '\n', (size_t)(last_nl - q));
                    q = nl ? nl + 1 : last_nl;
                }
                ranges.push_back(Range{p, q});
                p = q;
            }
            carry.assign(last_nl, (size_t)(e - last_nl));
        }
        if (!carry.empty()) {
            glued.push_back(std::move(carry));
            ranges.push_back(Range{glued.back().data(), glued.back().data() + glued.back().size()});
        }
    }
    std::vector<Chunk> chunks(ranges.size());
    {
        std::atomic<size_t> next{0};
        auto work = [&]() {
            for (;;) {
                const size_t i = next.fetch_add(1);
                if (i >= ranges.size()) return;
                parse_chunk(ranges[i].b, ranges[i].e, kind, keep_float, chunks[i]);
            }
        };
        const int nt = (int)std::max<size_t>(1, std::min<size_t>((size_t)n_threads, ranges.size()));
        std::vector<std::thread> pool;
        for (int k = 1; k < nt; ++k) pool.emplace_back(work);
        work();
        for (auto& th : pool) th.join();
    }
    mark("parse");
    int64_t line0 = 0;
    for (auto& c : chunks) {
        if (c.bad_line >= 0) {
            if (c.bad_is_range) {               // the reference's int() takes it: a limit of this library, not an input error
                t->error = "line " + std::to_string(line0 + c.bad_line + 1) + " in " + path +
                           ": a coordinate or count outside the int32 range of this library's columns (the reference accepts it)";
                return FHX_ERR_UNSUPPORTED;
            }
            t->error = "malformed line " + std::to_string(line0 + c.bad_line + 1) + " in " + path +
                       " (the reference raises ValueError on it)";
            return FHX_ERR_REFERENCE_EXIT;
        }
        line0 += c.n_lines;
    }
    // ---- names in order of first appearance over the whole file; the rows stay in their chunks until fhx_table_copy --------
    std::unordered_map<std::string, int32_t> index;
    t->remap.resize(chunks.size());
    t->row0.assign(chunks.size() + 1, 0);
    for (size_t k = 0; k < chunks.size(); ++k) {
        Chunk& c = chunks[k];
        t->remap[k].resize(c.names.size());
        for (size_t i = 0; i < c.names.size(); ++i) {
            auto it = index.find(c.names[i]);
            if (it == index.end()) {
                t->remap[k][i] = (int32_t)t->names.size();
                index.emplace(c.names[i], t->remap[k][i]);
                t->names.push_back(c.names[i]);
            } else {
                t->remap[k][i] = it->second;
            }
        }
        t->row0[k + 1] = t->row0[k] + c.mi[0].size();
    }
    t->chunks = std::move(chunks);
    mark("name index");
    if (timing) std::fprintf(stderr, "fhx_host_read_table(%s): %d threads:%s\n", path, n_threads, t_report.c_str());
    return FHX_OK;
}

static std::string inflate_report(const fhx_text* x) {
    char b[96];
    std::snprintf(b, sizeof(b), " file read %.3f s; inflate %.3f s;", x->seconds[0], x->seconds[1]);
    return b;
}

extern "C" {

int fhx_host_inflate(const char* path, int32_t n_threads, fhx_text** out) {
    if (!path || !out) return FHX_ERR_ARG;
    *out = nullptr;
    fhx_text* x = new (std::nothrow) fhx_text();
    if (!x) return FHX_ERR_NOMEM;
    *out = x;
    x->path = path;
    if (n_threads <= 0) n_threads = fhx::usable_cpus();
    const int rc = fhx::io_inflate_file(path, n_threads, x->pieces, x->error, x->seconds);
    if (rc != FHX_OK) return rc;
    for (const fhx::TextPiece& piece : x->pieces) x->bytes += (int64_t)piece.size();
    return FHX_OK;
}

int64_t fhx_text_bytes(const fhx_text* x) { return x ? x->bytes : 0; }
int fhx_text_copy(const fhx_text* x, void* dst, int64_t cap) {
    if (!x || (!dst && cap > 0) || cap < x->bytes) return FHX_ERR_ARG;
    char* out = (char*)dst;
    for (const fhx::TextPiece& piece : x->pieces) {
        std::memcpy(out, piece.data(), piece.size());
        out += piece.size();
    }
    return FHX_OK;
}
const char* fhx_text_error(const fhx_text* x) { return x ? x->error.c_str() : "null text"; }
void fhx_text_free(fhx_text* x) { delete x; }

// Bytes [*lo, *hi) of an inflated text that part `part` of `n_parts` takes: the rows that START in the part-th N-th of its bytes (a
// cut moves forward to the next start of a row, so every row is in exactly one part whatever its length; parts can be empty).
int fhx_text_part_bounds(const fhx_text* text, int32_t part, int32_t n_parts, int64_t* lo_out, int64_t* hi_out) {
    if (!text || !lo_out || !hi_out || n_parts < 1 || part < 0 || part >= n_parts) return FHX_ERR_ARG;
    const int64_t T = text->bytes;
    // the first start of a row at or after byte `at`: 0, T, or the byte after a newline
    auto cut = [&](int64_t at) -> int64_t {
        if (at <= 0) return 0;
        int64_t base = 0;
        bool searching = false;                             // true: looking for the first newline from the start of this piece on
        for (const fhx::TextPiece& piece : text->pieces) {
            const int64_t n = (int64_t)piece.size();
            if (!searching && at - 1 >= base + n) {
                base += n;
                continue;
            }
            const int64_t from = searching ? 0 : at - 1 - base;              // the row starting at `at` needs a newline at at - 1
            const void* hit = n > from ? std::memchr(piece.data() + from, '\n', (size_t)(n - from)) : nullptr;
            if (hit) return base + ((const char*)hit - piece.data()) + 1;
            searching = true;
            base += n;
        }
        return T;
    };
    *lo_out = cut(T / n_parts * part + T % n_parts * part / n_parts);
    *hi_out = part + 1 == n_parts ? T : cut(T / n_parts * (part + 1) + T % n_parts * (part + 1) / n_parts);
    return FHX_OK;
}

// length of the text's first row with its newline (what a part of a stream that begins inside a row must leave to the part before
// it - and hand over as that part's `extra`); -1 when the text holds no newline.  row (cap bytes) receives it when it fits.
int64_t fhx_text_first_row_end(const fhx_text* text, char* row, int64_t cap) {
    if (!text) return -1;
    int64_t base = 0;
    for (const fhx::TextPiece& piece : text->pieces) {
        const void* hit = piece.size() ? std::memchr(piece.data(), '\n', piece.size()) : nullptr;
        if (hit) {
            const int64_t len = base + ((const char*)hit - piece.data()) + 1;
            if (row && len <= cap) {
                int64_t at = 0;
                for (const fhx::TextPiece& q : text->pieces) {
                    const int64_t take = std::min<int64_t>((int64_t)q.size(), len - at);
                    if (take <= 0) break;
                    std::memcpy(row + at, q.data(), (size_t)take);
                    at += take;
                }
            }
            return len;
        }
        base += (int64_t)piece.size();
    }
    return -1;
}

int32_t fhx_text_ends_with_newline(const fhx_text* text) {
    if (!text) return 0;
    for (size_t k = text->pieces.size(); k-- > 0;)
        if (text->pieces[k].size()) return text->pieces[k].data()[text->pieces[k].size() - 1] == '\n' ? 1 : 0;
    return 0;
}

int fhx_host_parse_text(const fhx_text* text, int32_t kind, int32_t n_threads, fhx_table** out) {
    if (!text || !out) return FHX_ERR_ARG;
    return parse_pieces(text->pieces, text->path.c_str(), kind, n_threads, out, inflate_report(text));
}

int fhx_host_read_table(const char* path, int32_t kind, int32_t n_threads, fhx_table** out) {
    if (!path || !out) return FHX_ERR_ARG;
    const int k = kind & ~FHX_TABLE_NO_FLOAT;
    if (k < 0 || k > 2 || ((kind & FHX_TABLE_NO_FLOAT) && k != 0)) return FHX_ERR_ARG;
    fhx_text* x = nullptr;
    int rc = fhx_host_inflate(path, n_threads, &x);
    if (rc == FHX_OK) {
        rc = parse_pieces(x->pieces, path, kind, n_threads, out, inflate_report(x));
    } else if (x) {                                        // the message travels in a table, as before
        fhx_table* t = new (std::nothrow) fhx_table();
        if (t) t->error = x->error;
        *out = t;
    }
    fhx_text_free(x);
    return rc;
}

// The contacts table as the reference reads it ("%s\t%d\t%s\t%d\t%d\n", fithic/fithic.py:413-417), formatted and deflated
// on all cores, one size-tagged gzip member per 2^18 rows.  Tooling for synthetic workloads and for re-sharding inputs.
int fhx_host_write_contacts(const char* path, const char* const* chr_names, int32_t n_names, const int32_t* chr1,
                            const int32_t* mid1, const int32_t* chr2, const int32_t* mid2, const int32_t* count, int64_t n_rows,
                            int32_t gzip_level, int32_t n_threads) {
    if (!path || !chr_names || n_names <= 0 || n_rows < 0) return FHX_ERR_ARG;
    if (n_rows > 0 && (!chr1 || !mid1 || !chr2 || !mid2 || !count)) return FHX_ERR_ARG;
    if (gzip_level < 0 || gzip_level > 9) gzip_level = 6;
    if (n_threads <= 0) n_threads = fhx::usable_cpus();
    std::vector<size_t> name_len(n_names);
    for (int i = 0; i < n_names; ++i) {
        name_len[i] = std::strlen(chr_names[i]);
        if (name_len[i] > 256) return FHX_ERR_ARG;
    }
    std::FILE* f = std::fopen(path, "wb");
    if (!f) return FHX_ERR_ARG;
    const int64_t block = 1 << 18;
    const int64_t n_blocks = std::max<int64_t>(1, (n_rows + block - 1) / block);
    auto produce = [&](int64_t blk, std::string& text, std::string& zipped) -> bool {
        const int64_t lo = blk * block, hi = std::min(n_rows, lo + block);
        text.clear();
        zipped.clear();
        if (text.capacity() < (size_t)(hi - lo) * 40) text.reserve((size_t)(hi - lo) * 40);
        char buf[600];
        for (int64_t i = lo; i < hi; ++i) {
            if (chr1[i] < 0 || chr1[i] >= n_names || chr2[i] < 0 || chr2[i] >= n_names) return false;
            int n = 0;
            std::memcpy(buf + n, chr_names[chr1[i]], name_len[chr1[i]]);
            n += (int)name_len[chr1[i]];
            buf[n++] = '\t';
            n += put_int(buf + n, mid1[i]);
            buf[n++] = '\t';
            std::memcpy(buf + n, chr_names[chr2[i]], name_len[chr2[i]]);
            n += (int)name_len[chr2[i]];
            buf[n++] = '\t';
            n += put_int(buf + n, mid2[i]);
            buf[n++] = '\t';
            n += put_int(buf + n, count[i]);
            buf[n++] = '\n';
            text.append(buf, (size_t)n);
        }
        return deflate_member(text, gzip_level, zipped);
    };
    bool ok = ordered_parallel_write(f, n_blocks, n_threads, produce);
    if (std::fclose(f) != 0) ok = false;
    return ok ? FHX_OK : FHX_ERR_ARG;
}

int64_t fhx_table_rows(const fhx_table* t) { return t ? (t->row0.empty() ? 0 : (int64_t)t->row0.back()) : -1; }
int32_t fhx_table_n_names(const fhx_table* t) { return t ? (int32_t)t->names.size() : -1; }
const char* fhx_table_name(const fhx_table* t, int32_t i) {
    return (t && i >= 0 && i < (int32_t)t->names.size()) ? t->names[i].c_str() : nullptr;
}
const char* fhx_table_error(const fhx_table* t) { return t ? t->error.c_str() : "null table"; }

// ids[i] = the caller's id of name i: columns 0 and 2 of fhx_table_copy then come out in the caller's id space (the
// file-local order of first appearance is composed away here, once per chunk, instead of per row by the caller)
int fhx_table_map_names(fhx_table* t, const int32_t* ids, int32_t n_ids) {
    if (!t || !ids || n_ids != (int32_t)t->names.size()) return FHX_ERR_ARG;
    for (auto& m : t->remap)
        for (auto& v : m) v = ids[v];
    return FHX_OK;
}

// columns: 0 chr1, 1 mid1, 2 chr2, 3 mid2, 4 count / hits (int32); 5 raw count / bias (double)
int fhx_table_copy(const fhx_table* t, int32_t column, void* dst) {
    if (!t || !dst || column < 0 || column > 5) return FHX_ERR_ARG;
    const int kind = t->kind;
    if ((column == 2 || column == 3) && kind != 0) return FHX_ERR_ARG;
    if (column == 4 && kind == 2) return FHX_ERR_ARG;
    if (column == 5 && (kind == 1 || !t->has_float)) return FHX_ERR_ARG;
    std::atomic<size_t> next{0};
    auto work = [&]() {
        for (;;) {
            const size_t k = next.fetch_add(1);
            if (k >= t->chunks.size()) return;
            const Chunk& c = t->chunks[k];
            const size_t n = c.mi[0].size(), at = t->row0[k];
            if (!n) continue;
            if (column == 0 || column == 2) {
                const std::vector<int32_t>& src = c.ci[column == 0 ? 0 : 1];
                const std::vector<int32_t>& map = t->remap[k];
                int32_t* out = static_cast<int32_t*>(dst) + at;
                for (size_t i = 0; i < n; ++i) out[i] = map[(size_t)src[i]];
            } else if (column == 1 || column == 3) {
                std::memcpy(static_cast<int32_t*>(dst) + at, c.mi[column == 1 ? 0 : 1].data(), n * sizeof(int32_t));
            } else if (column == 4) {
                std::memcpy(static_cast<int32_t*>(dst) + at, c.iv.data(), n * sizeof(int32_t));
            } else {
                std::memcpy(static_cast<double*>(dst) + at, c.dv.data(), n * sizeof(double));
            }
        }
    };
    const int nt = (int)std::max<size_t>(1, std::min<size_t>((size_t)fhx::usable_cpus(), t->chunks.size()));
    std::vector<std::thread> pool;
    for (int k = 1; k < nt; ++k) pool.emplace_back(work);
    work();
    for (auto& th : pool) th.join();
    return FHX_OK;
}

void fhx_table_free(fhx_table* t) { delete t; }

}  // extern "C"
