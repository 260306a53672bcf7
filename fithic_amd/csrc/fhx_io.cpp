// fhx_io.cpp - native ingest / emit of the reference's gzip text formats (51 % of the
// reference's wall time is text I/O).  Host-only code, no GPU involved:
//   writer  <- the output loop of fit_Spline          (fithic/fithic.py:1167-1220): "%s\t%d\t%s\t%d\t%d\t%e\t%e\t%e\t%e\t%f\n"
//   reader  <- the three gzip readers                 (fithic/fithic.py:406-417, :581-590, :805-808, :818-821)
// Rows are formatted and deflated in parallel, one gzip member per block of rows (a concatenation of gzip members is a
// valid gzip file: zcat, Python's gzip module and the reference's own gzip.open read it as one stream).
#include <zlib.h>

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
    return std::snprintf(dst, 32, "%e", v);
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
    return std::snprintf(dst, 400, "%f", v);
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

bool deflate_member(const std::string& text, int level, std::string& out) {
    z_stream zs;
    std::memset(&zs, 0, sizeof(zs));
    if (deflateInit2(&zs, level, Z_DEFLATED, 15 + 16, 8, Z_DEFAULT_STRATEGY) != Z_OK) return false;   // 15+16: gzip wrapper
    out.resize(deflateBound(&zs, (uLong)text.size()) + 64);
    zs.next_in = (Bytef*)text.data();
    zs.avail_in = (uInt)text.size();
    zs.next_out = (Bytef*)&out[0];
    zs.avail_out = (uInt)out.size();
    const int rc = deflate(&zs, Z_FINISH);
    out.resize(zs.total_out);
    deflateEnd(&zs);
    return rc == Z_STREAM_END;
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
    if (n_threads <= 0) n_threads = (int)std::max(1u, std::thread::hardware_concurrency());
    std::FILE* f = std::fopen(path, "wb");
    if (!f) return FHX_ERR_ARG;
    const bool all_reg = mode == FHX_MODE_ALL, inter_only = mode == FHX_MODE_INTER_ONLY;
    const int64_t block = 1 << 16;                                   // rows per gzip member
    const int64_t n_blocks = (n_rows + block - 1) / block;
    std::vector<size_t> name_len(n_names);
    for (int i = 0; i < n_names; ++i) name_len[i] = std::strlen(chr_names[i]);
    int64_t written = 0;
    bool ok = true;
    {
        std::string hdr = "chr1\tfragmentMid1\tchr2\tfragmentMid2\tcontactCount\tp-value\tq-value\tbias1\tbias2\tExpCC\n", z;
        ok = deflate_member(hdr, gzip_level, z) && std::fwrite(z.data(), 1, z.size(), f) == z.size();
    }
    // waves of n_threads blocks: format + deflate in parallel, write in order
    for (int64_t b0 = 0; ok && b0 < n_blocks; b0 += n_threads) {
        const int nb = (int)std::min<int64_t>(n_threads, n_blocks - b0);
        std::vector<std::string> zipped(nb);
        std::vector<int64_t> rows(nb, 0);
        std::atomic<bool> fine{true};
        auto work = [&](int k) {
            const int64_t lo = (b0 + k) * block, hi = std::min(n_rows, lo + block);
            std::string text;
            text.reserve((size_t)(hi - lo) * 110);
            char buf[700];
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
                if (chr1[i] < 0 || chr1[i] >= n_names || chr2[i] < 0 || chr2[i] >= n_names) {
                    fine = false;
                    return;
                }
                int n = 0;
                std::memcpy(buf + n, chr_names[chr1[i]], name_len[chr1[i]]);
                n += (int)name_len[chr1[i]];
                buf[n++] = '\t';
                n += put_int(buf + n, mid1[i]);
                buf[n++] = '\t';
                if (n + name_len[chr2[i]] > 300) {
                    fine = false;
                    return;
                }
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
                ++rows[k];
            }
            if (!text.empty() && !deflate_member(text, gzip_level, zipped[k])) fine = false;
        };
        std::vector<std::thread> pool;
        for (int k = 1; k < nb; ++k) pool.emplace_back(work, k);
        work(0);
        for (auto& t : pool) t.join();
        if (!fine) ok = false;
        for (int k = 0; ok && k < nb; ++k) {
            if (!zipped[k].empty() && std::fwrite(zipped[k].data(), 1, zipped[k].size(), f) != zipped[k].size()) ok = false;
            written += rows[k];
        }
    }
    if (std::fclose(f) != 0) ok = false;
    if (rows_written) *rows_written = written;
    return ok ? FHX_OK : FHX_ERR_ARG;
}

}  // extern "C"

// =====================================================================================================================
// reader
// =====================================================================================================================
struct fhx_table {
    int kind = 0;
    std::vector<std::string> names;
    std::vector<int32_t> ci[2], mi[2], iv;       // chr ids / mids of locus 1 and 2; iv = count (contacts) or hits (fragments)
    std::vector<double> dv;                      // raw count (contacts) or bias (bias table)
    std::string error;
};

namespace {

struct Chunk {
    std::vector<std::string> names;
    std::unordered_map<std::string, int32_t> index;
    std::vector<int32_t> ci[2], mi[2], iv;
    std::vector<double> dv;
    int64_t bad_line = -1;                        // chunk-relative
    int64_t n_lines = 0;
};

inline bool is_space(char c) { return c == ' ' || c == '\t' || c == '\r' || c == '\v' || c == '\f'; }

inline bool parse_i32(const char* b, const char* e, int32_t& out) {      // Python int(): optional sign, digits
    if (b == e) return false;
    bool neg = false;
    if (*b == '+' || *b == '-') {
        neg = *b == '-';
        ++b;
    }
    if (b == e) return false;
    long long v = 0;
    for (; b < e; ++b) {
        if (*b < '0' || *b > '9') return false;
        v = v * 10 + (*b - '0');
        if (v > 4000000000ll) return false;
    }
    if (neg) v = -v;
    if (v < INT32_MIN || v > INT32_MAX) return false;
    out = (int32_t)v;
    return true;
}

inline bool parse_f64(const char* b, const char* e, double& out) {       // Python float()
    char tmp[64];
    const size_t n = (size_t)(e - b);
    if (n == 0 || n >= sizeof(tmp)) return false;
    std::memcpy(tmp, b, n);
    tmp[n] = 0;
    char* end = nullptr;
    out = std::strtod(tmp, &end);
    return end == tmp + n;
}

void parse_chunk(const char* b, const char* e, int kind, Chunk& c) {
    const char* fld_b[8];
    const char* fld_e[8];
    while (b < e) {
        const char* nl = (const char*)std::memchr(b, '\n', (size_t)(e - b));
        const char* le = nl ? nl : e;
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
        auto intern = [&](const char* s, const char* t) -> int32_t {
            std::string key(s, (size_t)(t - s));
            auto it = c.index.find(key);
            if (it != c.index.end()) return it->second;
            const int32_t id = (int32_t)c.names.size();
            c.names.push_back(key);
            c.index.emplace(std::move(key), id);
            return id;
        };
        if (kind == 0) {                                   // ch1 mid1 ch2 mid2 count: exactly 5 fields (fithic.py:413)
            int32_t m1, m2;
            double raw;
            ok = total == 5 && parse_i32(fld_b[1], fld_e[1], m1) && parse_i32(fld_b[3], fld_e[3], m2) &&
                 parse_f64(fld_b[4], fld_e[4], raw);
            if (ok) {
                const double tr = std::trunc(raw);
                ok = tr >= INT32_MIN && tr <= INT32_MAX;    // NaN fails both comparisons: int(float('nan')) raises in Python
                if (ok) {
                    c.ci[0].push_back(intern(fld_b[0], fld_e[0]));
                    c.mi[0].push_back(m1);
                    c.ci[1].push_back(intern(fld_b[2], fld_e[2]));
                    c.mi[1].push_back(m2);
                    c.iv.push_back((int32_t)tr);
                    c.dv.push_back(raw);
                }
            }
        } else if (kind == 1) {                            // words[0], int(words[2]), int(words[3]) (fithic.py:583-586)
            int32_t mid, hits;
            ok = total >= 4 && parse_i32(fld_b[2], fld_e[2], mid) && parse_i32(fld_b[3], fld_e[3], hits);
            if (ok) {
                c.ci[0].push_back(intern(fld_b[0], fld_e[0]));
                c.mi[0].push_back(mid);
                c.iv.push_back(hits);
            }
        } else {                                           // words[0], int(words[1]), float(words[2]) (fithic.py:807-808)
            int32_t mid;
            double bias;
            ok = total >= 3 && parse_i32(fld_b[1], fld_e[1], mid) && parse_f64(fld_b[2], fld_e[2], bias);
            if (ok) {
                c.ci[0].push_back(intern(fld_b[0], fld_e[0]));
                c.mi[0].push_back(mid);
                c.dv.push_back(bias);
            }
        }
        if (!ok) {
            c.bad_line = c.n_lines;
            return;
        }
        ++c.n_lines;
        b = nl ? nl + 1 : e;
    }
}

}  // namespace

extern "C" {

int fhx_host_read_table(const char* path, int32_t kind, int32_t n_threads, fhx_table** out) {
    if (!path || !out || kind < 0 || kind > 2) return FHX_ERR_ARG;
    *out = nullptr;
    fhx_table* t = new (std::nothrow) fhx_table();
    if (!t) return FHX_ERR_NOMEM;
    t->kind = kind;
    *out = t;
    gzFile g = gzopen(path, "rb");
    if (!g) {
        t->error = std::string("cannot open ") + path;
        return FHX_ERR_ARG;
    }
    gzbuffer(g, 1 << 20);
    std::string text;
    {
        std::vector<char> buf(8 << 20);
        for (;;) {
            const int n = gzread(g, buf.data(), (unsigned)buf.size());
            if (n < 0) {
                t->error = "gzip read error";
                gzclose(g);
                return FHX_ERR_ARG;
            }
            if (n == 0) break;
            text.append(buf.data(), (size_t)n);
        }
    }
    gzclose(g);
    if (n_threads <= 0) n_threads = (int)std::max(1u, std::thread::hardware_concurrency());
    n_threads = (int)std::min<size_t>((size_t)n_threads, text.size() / (1 << 20) + 1);
    // chunk boundaries on newlines
    std::vector<size_t> cut(n_threads + 1, text.size());
    cut[0] = 0;
    for (int k = 1; k < n_threads; ++k) {
        size_t pos = text.size() / n_threads * k;
        const void* nl = std::memchr(text.data() + pos, '\n', text.size() - pos);
        cut[k] = nl ? (size_t)((const char*)nl - text.data()) + 1 : text.size();
        if (cut[k] < cut[k - 1]) cut[k] = cut[k - 1];
    }
    std::vector<Chunk> chunks(n_threads);
    {
        std::vector<std::thread> pool;
        for (int k = 1; k < n_threads; ++k)
            pool.emplace_back(parse_chunk, text.data() + cut[k], text.data() + cut[k + 1], kind, std::ref(chunks[k]));
        parse_chunk(text.data() + cut[0], text.data() + cut[1], kind, chunks[0]);
        for (auto& th : pool) th.join();
    }
    int64_t line0 = 0;
    for (auto& c : chunks) {
        if (c.bad_line >= 0) {
            t->error = "malformed line " + std::to_string(line0 + c.bad_line + 1) + " in " + path +
                       " (the reference raises ValueError on it)";
            return FHX_ERR_REFERENCE_EXIT;
        }
        line0 += c.n_lines;
    }
    // merge: names in order of first appearance over the whole file
    std::unordered_map<std::string, int32_t> index;
    size_t rows = 0;
    for (auto& c : chunks) rows += c.mi[0].size();
    for (int s = 0; s < 2; ++s) {
        t->ci[s].reserve(rows);
        t->mi[s].reserve(rows);
    }
    t->iv.reserve(rows);
    t->dv.reserve(rows);
    for (auto& c : chunks) {
        std::vector<int32_t> remap(c.names.size());
        for (size_t i = 0; i < c.names.size(); ++i) {
            auto it = index.find(c.names[i]);
            if (it == index.end()) {
                remap[i] = (int32_t)t->names.size();
                index.emplace(c.names[i], remap[i]);
                t->names.push_back(c.names[i]);
            } else {
                remap[i] = it->second;
            }
        }
        for (int s = 0; s < 2; ++s) {
            for (int32_t v : c.ci[s]) t->ci[s].push_back(remap[v]);
            t->mi[s].insert(t->mi[s].end(), c.mi[s].begin(), c.mi[s].end());
        }
        t->iv.insert(t->iv.end(), c.iv.begin(), c.iv.end());
        t->dv.insert(t->dv.end(), c.dv.begin(), c.dv.end());
    }
    return FHX_OK;
}

int64_t fhx_table_rows(const fhx_table* t) { return t ? (int64_t)t->mi[0].size() : -1; }
int32_t fhx_table_n_names(const fhx_table* t) { return t ? (int32_t)t->names.size() : -1; }
const char* fhx_table_name(const fhx_table* t, int32_t i) {
    return (t && i >= 0 && i < (int32_t)t->names.size()) ? t->names[i].c_str() : nullptr;
}
const char* fhx_table_error(const fhx_table* t) { return t ? t->error.c_str() : "null table"; }

// columns: 0 chr1, 1 mid1, 2 chr2, 3 mid2, 4 count / hits (int32); 5 raw count / bias (double)
int fhx_table_copy(const fhx_table* t, int32_t column, void* dst) {
    if (!t || !dst) return FHX_ERR_ARG;
    const std::vector<int32_t>* iv = nullptr;
    switch (column) {
        case 0: iv = &t->ci[0]; break;
        case 1: iv = &t->mi[0]; break;
        case 2: iv = &t->ci[1]; break;
        case 3: iv = &t->mi[1]; break;
        case 4: iv = &t->iv; break;
        case 5:
            if (!t->dv.empty()) std::memcpy(dst, t->dv.data(), t->dv.size() * sizeof(double));
            return FHX_OK;
        default: return FHX_ERR_ARG;
    }
    if (!iv->empty()) std::memcpy(dst, iv->data(), iv->size() * sizeof(int32_t));
    return FHX_OK;
}

void fhx_table_free(fhx_table* t) { delete t; }

}  // extern "C"
