// fhx_gunzip.cpp - ONE plain gzip stream inflated on all host cores (what `gzip` writes: no member sizes, nothing to split on).
//
// The reference reads its inputs through Python's gzip module (fithic/fithic.py:404); a deflate stream is serial, and zlib
// needs 11 s for the 4.6 GB of a C3-size contacts file.  Two facts make it parallel (the scheme of pugz / rapidgzip, written
// here from RFC 1951):
//   * a dynamic-Huffman block header is recognisable: of 2^17 random bit patterns a handful pass the field ranges, and next to
//     none the requirement that the code-length code, the literal/length code and the distance code it describes are all
//     complete prefix codes with an end-of-block symbol.  So a thread can FIND a block start somewhere in the middle of the file;
//   * decoding from there lacks only the 32 KB of text before it.  The decoder writes 16-bit symbols: a byte, or "byte j of the
//     unknown window" (256 + j); matches copy these symbols like any others.  When the chunk before has been resolved, its last
//     32 KB turn every such symbol into its byte.
// Stages: (1) chunk starts = found block starts; (2) every chunk decoded in parallel up to the next chunk's start, which it must
// hit exactly; (3) the windows handed down the chain (32 KB per chunk, serial); (4) all chunks resolved into the text in
// parallel; (5) CRC-32 (zlib's crc32 per chunk + crc32_combine) and ISIZE against the trailer.  Any disagreement - a guessed
// block start that was none, a stream this decoder does not accept, a second member behind the first - and the caller inflates
// the file with zlib as before: the result is zlib's, or it is checked against the file's own CRC.
#include <sys/mman.h>
#include <zlib.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <string>
#include <thread>
#include <vector>

#include "../../include/fithic_mi355x.h"
#include "fhx_cpus.hpp"
#include "fhx_io_internal.hpp"

namespace {

constexpr int kWindow = 32768;
constexpr int kFastL = 11, kFastD = 9;

const uint16_t kLenBase[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
const uint8_t kLenBits[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
const uint16_t kDistBase[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073,
                                4097, 6145, 8193, 12289, 16385, 24577};
const uint8_t kDistBits[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};
const uint8_t kClOrder[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};

// bits of the stream [data, data + n), least significant first; reads past the end give zeros and set `over`
struct Bits {
    const uint8_t* data;
    size_t n;
    size_t byte = 0;             // next byte to load
    uint64_t buf = 0;
    int cnt = 0;
    bool over = false;
    void seek(uint64_t bitpos) {
        byte = (size_t)(bitpos >> 3);
        buf = 0;
        cnt = 0;
        over = false;
        refill();
        const int skip = (int)(bitpos & 7);
        buf >>= skip;
        cnt -= skip;
    }
    uint64_t pos() const { return (uint64_t)byte * 8 - (uint64_t)cnt; }
    void refill() {
        if (byte + 8 <= n) {                                   // 8 bytes at once: the bits that fit are taken, whole bytes counted
            uint64_t w;
            std::memcpy(&w, data + byte, 8);
            buf |= w << cnt;
            const int take = (63 - cnt) >> 3;
            byte += (size_t)take;
            cnt += take * 8;
            return;
        }
        while (cnt <= 56) {
            uint64_t b = 0;
            if (byte < n)
                b = data[byte];
            else
                over = true;
            ++byte;
            buf |= b << cnt;
            cnt += 8;
        }
    }
    uint32_t peek(int k) const { return (uint32_t)(buf & ((1ull << k) - 1)); }
    void drop(int k) {
        buf >>= k;
        cnt -= k;
    }
    uint32_t get(int k) {
        const uint32_t v = peek(k);
        drop(k);
        return v;
    }
};

// canonical code of `n` lengths: counts, symbols in code order; returns the code space left (0 = complete, < 0 = over-subscribed)
int construct(const uint8_t* lens, int n, uint16_t* cnt, uint16_t* sym) {
    for (int l = 0; l < 16; ++l) cnt[l] = 0;
    for (int s = 0; s < n; ++s) cnt[lens[s]]++;
    int left = 1;
    for (int l = 1; l < 16; ++l) {
        left <<= 1;
        left -= cnt[l];
        if (left < 0) return left;
    }
    uint16_t offs[16];
    offs[1] = 0;
    for (int l = 1; l < 15; ++l) offs[l + 1] = (uint16_t)(offs[l] + cnt[l]);
    for (int s = 0; s < n; ++s)
        if (lens[s]) sym[offs[lens[s]]++] = (uint16_t)s;
    return left;
}

// entries: bits 0-3 code length (0 = longer than the table), 4-7 extra bits, 8-9 kind (0 literal / distance, 1 length,
// 2 end of block, 3 not a symbol), 16-31 the byte or the base value
uint32_t lit_entry(uint32_t s, uint32_t len) {
    if (s < 256) return (s << 16) | len;
    if (s == 256) return (2u << 8) | len;
    if (s >= 286) return (3u << 8) | len;
    return ((uint32_t)kLenBase[s - 257] << 16) | (1u << 8) | ((uint32_t)kLenBits[s - 257] << 4) | len;
}
uint32_t dist_entry(uint32_t s, uint32_t len) {
    if (s >= 30) return (3u << 8) | len;
    return ((uint32_t)kDistBase[s] << 16) | ((uint32_t)kDistBits[s] << 4) | len;
}
uint32_t reverse_bits(uint32_t v, int n) {
    uint32_t r = 0;
    for (int i = 0; i < n; ++i) r |= ((v >> i) & 1u) << (n - 1 - i);
    return r;
}
template <typename F>
void fill_fast(const uint16_t* cnt, const uint16_t* sym, uint32_t* fast, int bits, F entry) {
    std::memset(fast, 0, sizeof(uint32_t) << bits);
    uint32_t code = 0, index = 0;
    for (int l = 1; l <= bits; ++l) {
        for (uint32_t j = 0; j < cnt[l]; ++j) {
            const uint32_t e = entry(sym[index + j], (uint32_t)l), rev = reverse_bits(code + j, l);
            for (uint32_t k = 0; k < (1u << (bits - l)); ++k) fast[rev | (k << l)] = e;
        }
        index += cnt[l];
        code = (code + cnt[l]) << 1;
    }
}
template <typename F>
uint32_t decode_slow(uint64_t b, const uint16_t* cnt, const uint16_t* sym, F entry) {
    int code = 0, first = 0, index = 0;
    for (int l = 1; l <= 15; ++l) {
        code |= (int)(b & 1);
        b >>= 1;
        const int c = cnt[l];
        if (code - c < first) return entry(sym[index + (code - first)], (uint32_t)l);
        index += c;
        first += c;
        first <<= 1;
        code <<= 1;
    }
    return 0;
}

struct Tables {
    uint32_t lit_fast[1 << kFastL], dist_fast[1 << kFastD];
    uint16_t lit_cnt[16], dist_cnt[16], lit_sym[288], dist_sym[32];
};

// The header of a dynamic block at the reader's position (after the 3 block bits): lengths -> tables.  false = not a header a
// deflate encoder writes (and not one zlib accepts): used both to decode and to recognise block starts.
bool dynamic_tables(Bits& B, Tables& T) {
    B.refill();
    const int n_lit = (int)B.get(5) + 257, n_dist = (int)B.get(5) + 1, n_cl = (int)B.get(4) + 4;
    if (n_lit > 286 || n_dist > 30) return false;
    uint8_t lens[320];
    std::memset(lens, 0, sizeof(lens));
    for (int k = 0; k < n_cl; ++k) {
        B.refill();
        lens[kClOrder[k]] = (uint8_t)B.get(3);
    }
    uint16_t cl_cnt[16], cl_sym[19];
    if (construct(lens, 19, cl_cnt, cl_sym) != 0) return false;            // the code length code must be complete
    uint32_t cl_fast[128];
    fill_fast(cl_cnt, cl_sym, cl_fast, 7, [](uint32_t s, uint32_t l) { return (s << 16) | l; });
    std::memset(lens, 0, sizeof(lens));
    int at = 0;
    const int total = n_lit + n_dist;
    uint32_t prev = 0;
    while (at < total) {
        B.refill();
        const uint32_t e = cl_fast[B.peek(7)];
        if ((e & 15u) == 0) return false;
        B.drop((int)(e & 15u));
        const uint32_t s = e >> 16;
        if (s < 16) {
            lens[at++] = (uint8_t)s;
            prev = s;
        } else {
            uint32_t v = 0, rep;
            if (s == 16) {
                if (at == 0) return false;
                v = prev;
                rep = 3 + B.get(2);
            } else if (s == 17) {
                rep = 3 + B.get(3);
            } else {
                rep = 11 + B.get(7);
            }
            if (at + (int)rep > total) return false;
            for (uint32_t k = 0; k < rep; ++k) lens[at++] = (uint8_t)v;
            prev = v;
        }
    }
    if (B.over || lens[256] == 0) return false;
    uint8_t dl[32];
    std::memset(dl, 0, sizeof(dl));
    std::memcpy(dl, lens + n_lit, (size_t)n_dist);
    std::memset(lens + n_lit, 0, (size_t)(320 - n_lit));
    const int left_l = construct(lens, 288, T.lit_cnt, T.lit_sym);
    if (left_l < 0 || (left_l > 0 && !(288 - T.lit_cnt[0] == 1 && T.lit_cnt[1] == 1))) return false;
    const int left_d = construct(dl, 32, T.dist_cnt, T.dist_sym);
    if (left_d < 0 || (left_d > 0 && !(32 - T.dist_cnt[0] == 1 && T.dist_cnt[1] == 1))) return false;
    fill_fast(T.lit_cnt, T.lit_sym, T.lit_fast, kFastL, lit_entry);
    fill_fast(T.dist_cnt, T.dist_sym, T.dist_fast, kFastD, dist_entry);
    return true;
}

void fixed_tables(Tables& T) {
    uint8_t lens[288], dl[32];
    for (int s = 0; s < 288; ++s) lens[s] = s < 144 ? 8 : s < 256 ? 9 : s < 280 ? 7 : 8;
    std::memset(dl, 5, sizeof(dl));
    construct(lens, 288, T.lit_cnt, T.lit_sym);
    construct(dl, 32, T.dist_cnt, T.dist_sym);
    fill_fast(T.lit_cnt, T.lit_sym, T.lit_fast, kFastL, lit_entry);
    fill_fast(T.dist_cnt, T.dist_sym, T.dist_fast, kFastD, dist_entry);
}

// first bit position >= from (and < limit) where a non-final dynamic block with complete codes begins; UINT64_MAX if none
uint64_t find_block(const uint8_t* data, size_t n, uint64_t from, uint64_t limit, Tables& scratch) {
    Bits B{data, n};
    for (uint64_t at = from; at < limit; ++at) {
        // cheap look first: BFINAL = 0, BTYPE = 10, HLIT <= 29, HDIST <= 29
        const size_t by = (size_t)(at >> 3);
        if (by + 4 > n) return UINT64_MAX;
        uint32_t w;
        std::memcpy(&w, data + by, 4);
        w >>= (at & 7);
        if ((w & 7u) != 4u) continue;                                     // bits: 0 (not final), then 0 1 = type 2 (LSB first: 100b)
        if (((w >> 3) & 31u) > 29u || ((w >> 8) & 31u) > 29u) continue;
        B.seek(at + 3);
        if (dynamic_tables(B, scratch) && !B.over) return at;
    }
    return UINT64_MAX;
}

// 16-bit symbols of a chunk: an anonymous mapping on huge pages (nothing is zero-filled by us, first touch is cheap), grown by
// mremap when a chunk expands more than expected
struct SymBuf {
    uint16_t* p = nullptr;
    size_t cap = 0;
    SymBuf() = default;
    SymBuf(const SymBuf&) = delete;
    SymBuf& operator=(const SymBuf&) = delete;
    SymBuf(SymBuf&& o) noexcept : p(o.p), cap(o.cap) { o.p = nullptr; o.cap = 0; }
    SymBuf& operator=(SymBuf&& o) noexcept {
        if (this != &o) {
            release();
            p = o.p;
            cap = o.cap;
            o.p = nullptr;
            o.cap = 0;
        }
        return *this;
    }
    ~SymBuf() { release(); }
    void release() {
        if (p) ::munmap(p, cap * sizeof(uint16_t));
        p = nullptr;
        cap = 0;
    }
    bool reserve(size_t want) {
        if (want <= cap) return true;
        const size_t bytes = ((want * sizeof(uint16_t) + ((size_t)2 << 20) - 1) >> 21) << 21;
        void* m = p ? ::mremap(p, cap * sizeof(uint16_t), bytes, MREMAP_MAYMOVE)
                    : ::mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
        if (m == MAP_FAILED) return false;
        (void)::madvise(m, bytes, MADV_HUGEPAGE);
        p = (uint16_t*)m;
        cap = bytes / sizeof(uint16_t);
        return true;
    }
    uint16_t* data() { return p; }
    const uint16_t* data() const { return p; }
    size_t size() const { return cap; }
};

struct Chunk {
    uint64_t start_bit = 0, stop_bit = 0;      // decodes blocks from start_bit until a block would begin at stop_bit (or the final block ends)
    SymBuf* sym = nullptr;                     // kWindow markers, then the output symbols (a buffer of the batch's pool)
    size_t n_out = 0;
    bool ok = false, saw_final = false;
    uint64_t end_bit = 0;
};

// blocks from C.start_bit on; first = the stream's own start (no unknown window: a distance beyond the output is an error)
// expect_out: the first reservation; a chunk that grows beyond 8 x that (text compresses 3-10 x; 64 x its compressed bytes) is
// given up - sixteen chunks of a file of zeros would otherwise ask for hundreds of GB of symbols - and zlib streams the file
void decode_chunk(const uint8_t* data, size_t n, Chunk& C, bool first, size_t expect_out) {
    const size_t out_limit = kWindow + std::max<size_t>(expect_out * 8, (size_t)64 << 20);
    Tables* T = new Tables();
    struct Free {
        Tables* t;
        ~Free() { delete t; }
    } guard{T};
    SymBuf& buf16 = *C.sym;
    if (!buf16.reserve(kWindow + std::max<size_t>(expect_out, 1 << 16))) return;
    uint16_t* out = buf16.data();
    for (int j = 0; j < kWindow; ++j) out[(size_t)j] = (uint16_t)(256 + j);
    size_t pos = kWindow;
    Bits B{data, n};
    B.seek(C.start_bit);
    for (;;) {
        const uint64_t here = B.pos();
        if (here >= C.stop_bit) {
            if (here != C.stop_bit) return;                                // ran past the next chunk's start: one of the two is wrong
            break;
        }
        B.refill();
        const bool last = B.get(1) != 0;
        const uint32_t type = B.get(2);
        if (type == 3) return;
        if (type == 0) {
            B.drop(B.cnt & 7);
            B.refill();
            const uint32_t len = B.get(16), nlen = B.get(16);
            if ((len ^ 0xffffu) != nlen) return;
            if (buf16.size() < pos + len + 512) {
                if (pos + len > out_limit || !buf16.reserve((pos + len) * 2)) return;
                out = buf16.data();
            }
            for (uint32_t i = 0; i < len; ++i) {
                B.refill();
                out[pos++] = (uint16_t)B.get(8);
            }
            if (B.over) return;
        } else {
            if (type == 1) {
                fixed_tables(*T);
            } else if (!dynamic_tables(B, *T)) {
                return;
            }
            for (;;) {
                if (buf16.size() < pos + 600) {
                    if (pos > out_limit || !buf16.reserve(buf16.size() * 2)) return;
                    out = buf16.data();
                }
                B.refill();
                uint32_t e = T->lit_fast[B.peek(kFastL)];
                if ((e & 15u) == 0) {
                    e = decode_slow(B.buf, T->lit_cnt, T->lit_sym, lit_entry);
                    if (e == 0) return;
                }
                B.drop((int)(e & 15u));
                const uint32_t kind = (e >> 8) & 3u;
                if (kind == 0) {
                    out[pos++] = (uint16_t)(e >> 16);
                } else if (kind == 1) {
                    const uint32_t len = (e >> 16) + B.get((int)((e >> 4) & 15u));
                    B.refill();
                    uint32_t d = T->dist_fast[B.peek(kFastD)];
                    if ((d & 15u) == 0) {
                        d = decode_slow(B.buf, T->dist_cnt, T->dist_sym, dist_entry);
                        if (d == 0) return;
                    }
                    if ((d >> 8) & 3u) return;
                    B.drop((int)(d & 15u));
                    const uint32_t dist = (d >> 16) + B.get((int)((d >> 4) & 15u));
                    if (first && dist > pos - kWindow) return;
                    const uint16_t* src = out + pos - dist;
                    uint16_t* dst = out + pos;
                    for (uint32_t i = 0; i < len; ++i) dst[i] = src[i];     // forward, element by element: overlapping copies repeat
                    pos += len;
                } else if (kind == 2) {
                    break;
                } else {
                    return;
                }
                if (B.over) return;
            }
        }
        if (last) {
            C.saw_final = true;
            break;
        }
    }
    C.end_bit = B.pos();
    C.n_out = pos - kWindow;
    C.ok = true;
}

// offset of the deflate stream behind the header of the file's first member; false + why: not a gzip header / truncated
bool member_payload(const unsigned char* gz, size_t n, size_t& o, std::string& why) {
    if (n < 18 || gz[0] != 0x1f || gz[1] != 0x8b || gz[2] != 8 || (gz[3] & 0xe0)) {
        why = "not a gzip header";
        return false;
    }
    o = 10;
    const unsigned flg = gz[3];
    if (flg & 4) {
        if (o + 2 > n) {
            why = "truncated";
            return false;
        }
        o += 2 + (gz[o] | ((size_t)gz[o + 1] << 8));
    }
    for (int name = 0; name < 2; ++name)
        if (flg & (name == 0 ? 8u : 16u)) {
            while (o < n && gz[o]) ++o;
            ++o;
        }
    if (flg & 2) o += 2;
    if (o + 8 >= n) {
        why = "truncated";
        return false;
    }
    return true;
}

template <typename Job>
void run_jobs(size_t lo, size_t hi, int n_threads, const Job& job) {
    std::atomic<size_t> next{lo};
    auto work = [&]() {
        for (;;) {
            const size_t k = next.fetch_add(1);
            if (k >= hi) break;
            job(k);
        }
    };
    std::vector<std::thread> pool;
    for (int t = 1; t < n_threads && (size_t)t < hi - lo; ++t) pool.emplace_back(work);
    work();
    for (auto& th : pool) th.join();
}

}  // namespace

// gz = a whole gzip file.  true: `pieces` hold its text (one piece per chunk, in order), checked against the trailer's CRC-32
// and ISIZE.  false: not done (why says what stood in the way) - the caller uses zlib.
bool fhx::io_parallel_gunzip(const unsigned char* gz, size_t n, int n_threads, std::vector<fhx::TextPiece>& pieces, std::string& why) {
    // FHX_PGUNZIP_MIN / FHX_PGUNZIP_CHUNK (bytes): test knobs for the size from which the scheme is used (default 4 MB: below
    // that zlib is done in a few hundredths of a second) and for the compressed bytes per chunk (default 512 KB at least)
    size_t min_bytes = (size_t)4 << 20, chunk_bytes = (size_t)512 << 10;
    if (const char* e = std::getenv("FHX_PGUNZIP_MIN")) min_bytes = (size_t)std::max(0ll, std::atoll(e));
    if (const char* e = std::getenv("FHX_PGUNZIP_CHUNK")) chunk_bytes = (size_t)std::max(1024ll, std::atoll(e));
    if (n_threads < 2 || n < min_bytes) {
        why = "small file or one thread";
        return false;
    }
    size_t o = 0;
    if (!member_payload(gz, n, o, why)) return false;
    const uint8_t* data = gz + o;
    const size_t len = n - o - 8;                                         // if this is the only member: deflate stream, then CRC-32 and ISIZE
    const unsigned char* tail = gz + n - 8;
    const uint32_t want_crc = tail[0] | ((uint32_t)tail[1] << 8) | ((uint32_t)tail[2] << 16) | ((uint32_t)tail[3] << 24);
    const uint32_t want_isize = tail[4] | ((uint32_t)tail[5] << 8) | ((uint32_t)tail[6] << 16) | ((uint32_t)tail[7] << 24);
    const bool timing = std::getenv("FHX_TIMING") != nullptr;
    const auto t_begin = std::chrono::steady_clock::now();
    auto since = [&]() { return std::chrono::duration<double>(std::chrono::steady_clock::now() - t_begin).count(); };
    double t_find = 0, t_decode = 0, t_chain = 0, t_resolve = 0;
    // ---- (1) chunk starts ------------------------------------------------------------------------------------------------
    const size_t n_chunks = std::min<size_t>((size_t)n_threads * 4, std::max<size_t>(2, len / chunk_bytes));
    std::vector<Chunk> chunks(n_chunks);
    {
        std::atomic<size_t> next{1};
        auto work = [&]() {
            Tables* scratch = new Tables();
            for (;;) {
                const size_t k = next.fetch_add(1);
                if (k >= n_chunks) break;
                const uint64_t from = (uint64_t)(len / n_chunks * k) * 8, limit = (uint64_t)(len / n_chunks * (k + 1)) * 8;
                chunks[k].start_bit = find_block(data, len + 8, from, limit, *scratch);
            }
            delete scratch;
        };
        std::vector<std::thread> pool;
        for (int t = 1; t < n_threads; ++t) pool.emplace_back(work);
        work();
        for (auto& th : pool) th.join();
    }
    t_find = since();
    chunks[0].start_bit = 0;
    {
        std::vector<Chunk> kept;
        for (auto& c : chunks)
            if (c.start_bit != UINT64_MAX) kept.push_back(std::move(c));   // a stretch without a dynamic block start joins the chunk before it
        chunks.swap(kept);
    }
    if (chunks.size() < 2) {
        why = "no block starts found (stored or fixed-code blocks only)";
        return false;
    }
    for (size_t k = 0; k < chunks.size(); ++k) chunks[k].stop_bit = k + 1 < chunks.size() ? chunks[k + 1].start_bit : UINT64_MAX;
    // ---- (2)-(5) in batches of n_threads chunks, so that the 16-bit symbols of at most that many chunks exist at a time:
    // decode the batch in parallel, hand the windows down its chain, resolve + CRC in parallel ------------------------------
    const size_t readable = len + 8;                                      // the decoder may look at (never consume) the trailer
    pieces.clear();
    pieces.resize(chunks.size());
    std::vector<uint32_t> crcs(chunks.size(), 0);
    std::vector<uint8_t> carry(kWindow, 0);                               // the 32 KB of text before the batch's first chunk
    uint64_t total = 0;
    auto run_parallel = [&](size_t lo, size_t hi, const std::function<void(size_t)>& job) {
        std::atomic<size_t> next{lo};
        auto work = [&]() {
            for (;;) {
                const size_t k = next.fetch_add(1);
                if (k >= hi) break;
                job(k);
            }
        };
        std::vector<std::thread> pool;
        for (int t = 1; t < n_threads && (size_t)t < hi - lo; ++t) pool.emplace_back(work);
        work();
        for (auto& th : pool) th.join();
    };
    std::vector<SymBuf> pool((size_t)n_threads);                         // reused by every batch: mapping and unmapping GBs is not free
    for (size_t lo = 0; lo < chunks.size(); lo += (size_t)n_threads) {
        const size_t hi = std::min(chunks.size(), lo + (size_t)n_threads);
        for (size_t k = lo; k < hi; ++k) chunks[k].sym = &pool[k - lo];
        double t0 = since();
        run_parallel(lo, hi, [&](size_t k) {
            const uint64_t span = (k + 1 < chunks.size() ? chunks[k + 1].start_bit : (uint64_t)len * 8) - chunks[k].start_bit;
            decode_chunk(data, readable, chunks[k], k == 0, (size_t)(span / 8) * 8);
        });
        t_decode += since() - t0;
        t0 = since();
        for (size_t k = lo; k < hi; ++k) {
            if (!chunks[k].ok) {
                pieces.clear();
                why = "a chunk did not decode from its guessed block start to the next one";
                return false;
            }
            if (chunks[k].saw_final != (k + 1 == chunks.size())) {
                pieces.clear();
                why = "the final block is not where the stream ends (more than one member?)";
                return false;
            }
        }
        if (hi == chunks.size() && (chunks.back().end_bit + 7) / 8 != len) {
            pieces.clear();
            why = "the stream does not end where the file's trailer begins (more than one member?)";
            return false;
        }
        // windows down the chain: window[k] = the 32 KB of text before chunk k
        std::vector<std::vector<uint8_t>> window(hi - lo + 1);
        window[0] = carry;
        for (size_t k = lo; k < hi; ++k) {
            const Chunk& c = chunks[k];
            const std::vector<uint8_t>& before = window[k - lo];
            std::vector<uint8_t>& w = window[k - lo + 1];
            w.resize(kWindow);
            const uint16_t* end = c.sym->data() + kWindow + c.n_out;      // the last kWindow symbols of [markers of the window, output]
            for (int j = 0; j < kWindow; ++j) {
                const uint16_t v = end[j - kWindow];
                w[(size_t)j] = v < 256 ? (uint8_t)v : before[(size_t)(v - 256)];
            }
        }
        carry = window[hi - lo];
        t_chain += since() - t0;
        t0 = since();
        std::atomic<bool> bad{false};
        run_parallel(lo, hi, [&](size_t k) {
            Chunk& c = chunks[k];
            if (!pieces[k].allocate(c.n_out)) {
                bad = true;
                return;
            }
            uint8_t* dst = (uint8_t*)pieces[k].data();
            const uint16_t* src = c.sym->data() + kWindow;
            const uint8_t* w = window[k - lo].data();
            size_t i = 0;
            for (; i + 4 <= c.n_out; i += 4) {                          // four symbols at a time when none of them is a window reference
                uint64_t q;
                std::memcpy(&q, src + i, 8);
                if ((q & 0xff00ff00ff00ff00ull) == 0) {
                    dst[i] = (uint8_t)q;
                    dst[i + 1] = (uint8_t)(q >> 16);
                    dst[i + 2] = (uint8_t)(q >> 32);
                    dst[i + 3] = (uint8_t)(q >> 48);
                } else {
                    for (int j = 0; j < 4; ++j) {
                        const uint16_t v = src[i + j];
                        dst[i + j] = v < 256 ? (uint8_t)v : w[v - 256];
                    }
                }
            }
            for (; i < c.n_out; ++i) {
                const uint16_t v = src[i];
                dst[i] = v < 256 ? (uint8_t)v : w[v - 256];
            }
            uint32_t crc = (uint32_t)crc32(0L, Z_NULL, 0);
            for (size_t i = 0; i < c.n_out; i += (size_t)1 << 30)
                crc = (uint32_t)crc32(crc, dst + i, (uInt)std::min<size_t>((size_t)1 << 30, c.n_out - i));
            crcs[k] = crc;
        });
        if (bad) {
            pieces.clear();
            why = "out of memory";
            return false;
        }
        for (size_t k = lo; k < hi; ++k) total += chunks[k].n_out;
        t_resolve += since() - t0;
    }
    if (timing)
        std::fprintf(stderr, "parallel gunzip: %zu chunks on %d threads: block starts %.3f s; decode %.3f s; windows %.3f s; resolve + CRC-32 %.3f s\n",
                     chunks.size(), n_threads, t_find, t_decode, t_chain, t_resolve);
    if ((uint32_t)total != want_isize) {
        pieces.clear();
        why = "ISIZE differs";
        return false;
    }
    uint32_t crc = crcs[0];
    for (size_t k = 1; k < chunks.size(); ++k) crc = (uint32_t)crc32_combine(crc, crcs[k], (z_off_t)chunks[k].n_out);
    if (crc != want_crc) {
        pieces.clear();
        why = "CRC-32 differs";
        return false;
    }
    return true;
}


// ---- ONE plain gzip stream inflated by N processes, each its N-th of the compressed bytes -----------------------------------------
// (sharded runs: fithic --gpus N on a file `gzip` wrote.  Every rank inflating the whole file is N times the work and N copies of the
// text in host memory.)  The scheme above already decodes a chunk without the 32 KB before it; here the unknown window is the one
// before the PART, known only when the ranks before have decoded theirs.  So in two calls:
//   decode   the part's chunks -> 16-bit symbols, batch after batch as above; of every chunk only its last 32 K symbols are kept,
//            chained into the part's last 32 K symbols in terms of the window before the part, which go to the caller;
//   resolve  the caller chains those tails rank after rank (part 0 knows its window: none) and hands every part the 32 KB before it:
//            the chunks are decoded once more, narrowed to bytes through their windows, and the part's CRC-32 is taken.
// Part 0 does both in the first call.  Why decode twice instead of keeping the symbols: in text of this kind two thirds of the
// symbols of a chunk still refer to the unknown window after megabytes ("chr1", the shared digits of neighbouring coordinates are
// copied on and on, never written as literals again) - the symbols are 2 bytes per byte of text, on every rank at once, where a
// second decode is 1/N of the file's on 1/N of the cores.  The caller checks the combined CRC-32 (crc32_combine) and the summed
// length against the file's trailer - as above, the text is zlib's or it is found out.  Chunk grid: n_parts x per_part stretches
// of the compressed bytes, the same on every rank; a stretch without a block start joins the chunk before it, whichever rank
// holds that.
struct fhx_text_part {
    std::string path, error;
    int part = 0, n_parts = 1, n_threads = 1;
    fhx::FileBytes gz;
    size_t payload = 0, len = 0;           // the deflate stream: gz[payload, payload + len)
    std::vector<Chunk> chunks;             // start / stop bits of this part's chunks
    std::vector<uint16_t> tail;            // the part's last kWindow symbols in terms of the window before the part
    std::vector<fhx::TextPiece> pieces;    // the text, once the window was known
    uint32_t crc = 0;
    int64_t bytes = 0;
    bool last = false;                     // holds the final block, which ends where the trailer begins
    bool resolved = false;
    double seconds[3] = {0, 0, 0};         // block starts, first decode, second decode
};

// All chunks of the part decoded in batches of n_threads.  window0 = the kWindow bytes before the part: pieces, crc, bytes are
// produced; nullptr: only P.tail (symbolic) and P.bytes.
static bool part_run(fhx_text_part& P, const uint8_t* window0, std::string& why) {
    const uint8_t* data = P.gz.data() + P.payload;
    const size_t len = P.len, readable = len + 8;
    const int n_threads = P.n_threads;
    std::vector<Chunk>& chunks = P.chunks;
    std::vector<SymBuf> pool((size_t)n_threads);
    std::vector<uint16_t> sym_tail(kWindow), w16(kWindow);
    for (int j = 0; j < kWindow; ++j) sym_tail[(size_t)j] = (uint16_t)(256 + j);
    std::vector<uint8_t> carry(kWindow, 0);
    if (window0) std::memcpy(carry.data(), window0, kWindow);
    std::vector<uint32_t> crcs(chunks.size(), 0);
    if (window0) {
        P.pieces.clear();
        P.pieces.resize(chunks.size());
    }
    int64_t total = 0;
    for (size_t lo = 0; lo < chunks.size(); lo += (size_t)n_threads) {
        const size_t hi = std::min(chunks.size(), lo + (size_t)n_threads);
        for (size_t k = lo; k < hi; ++k) {
            chunks[k].sym = &pool[k - lo];
            chunks[k].ok = chunks[k].saw_final = false;
        }
        run_jobs(lo, hi, n_threads, [&](size_t k) {
            Chunk& c = chunks[k];
            const uint64_t span = (c.stop_bit == UINT64_MAX ? (uint64_t)len * 8 : c.stop_bit) - c.start_bit;
            decode_chunk(data, readable, c, c.start_bit == 0, (size_t)(span / 8) * 8);
        });
        for (size_t k = lo; k < hi; ++k) {
            if (!chunks[k].ok) {
                why = "a chunk did not decode from its guessed block start to the next one";
                return false;
            }
            if (chunks[k].saw_final != (P.last && k + 1 == chunks.size())) {
                why = "the final block is not where the stream ends (more than one member?)";
                return false;
            }
        }
        // windows down the batch: symbolic (in terms of the window before the part) or, when that window is known, bytes
        std::vector<std::vector<uint8_t>> window(window0 ? hi - lo : 0);
        for (size_t k = lo; k < hi; ++k) {
            const Chunk& c = chunks[k];
            const uint16_t* end = c.sym->data() + kWindow + c.n_out;      // the last kWindow symbols of [markers of the window, output]
            if (window0) {
                window[k - lo] = carry;
                for (int j = 0; j < kWindow; ++j) {
                    const uint16_t v = end[j - kWindow];
                    carry[(size_t)j] = v < 256 ? (uint8_t)v : window[k - lo][(size_t)(v - 256)];
                }
            } else {
                for (int j = 0; j < kWindow; ++j) {
                    const uint16_t v = end[j - kWindow];
                    w16[(size_t)j] = v < 256 ? v : sym_tail[(size_t)(v - 256)];
                }
                sym_tail.swap(w16);
            }
            total += (int64_t)c.n_out;
        }
        if (!window0) continue;
        std::atomic<bool> bad{false};
        run_jobs(lo, hi, n_threads, [&](size_t k) {
            Chunk& c = chunks[k];
            if (!P.pieces[k].allocate(c.n_out)) {
                bad = true;
                return;
            }
            uint8_t* dst = (uint8_t*)P.pieces[k].data();
            const uint16_t* src = c.sym->data() + kWindow;
            const uint8_t* w = window[k - lo].data();
            for (size_t i = 0; i < c.n_out; ++i) {
                const uint16_t v = src[i];
                dst[i] = v < 256 ? (uint8_t)v : w[v - 256];
            }
            uint32_t crc = (uint32_t)crc32(0L, Z_NULL, 0);
            for (size_t i = 0; i < c.n_out; i += (size_t)1 << 30)
                crc = (uint32_t)crc32(crc, dst + i, (uInt)std::min<size_t>((size_t)1 << 30, c.n_out - i));
            crcs[k] = crc;
        });
        if (bad) {
            why = "out of memory";
            return false;
        }
    }
    if (P.last && !chunks.empty() && (chunks.back().end_bit + 7) / 8 != len) {
        why = "the stream does not end where the file's trailer begins (more than one member?)";
        return false;
    }
    P.bytes = total;
    if (window0) {
        P.tail.resize(kWindow);
        for (int j = 0; j < kWindow; ++j) P.tail[(size_t)j] = carry[(size_t)j];
        uint32_t crc = (uint32_t)crc32(0L, Z_NULL, 0);
        for (size_t k = 0; k < chunks.size(); ++k) crc = (uint32_t)crc32_combine(crc, crcs[k], (z_off_t)chunks[k].n_out);
        P.crc = crc;
        P.resolved = true;
    } else {
        P.tail = sym_tail;
    }
    for (Chunk& c : chunks) c.sym = nullptr;
    return true;
}

// the part's chunks: block starts inside its stretches of the grid, the last one ending at the first start of a later part
static bool part_chunks(fhx_text_part& P, std::string& why) {
    size_t min_bytes = (size_t)4 << 20, chunk_bytes = 0;
    if (const char* e = std::getenv("FHX_PGUNZIP_MIN")) min_bytes = (size_t)std::max(0ll, std::atoll(e));
    if (const char* e = std::getenv("FHX_PGUNZIP_CHUNK")) chunk_bytes = (size_t)std::max(1024ll, std::atoll(e));
    const unsigned char* gz = P.gz.data();
    const size_t n = P.gz.size();
    if (n < min_bytes) {
        why = "small file";
        return false;
    }
    if (!member_payload(gz, n, P.payload, why)) return false;
    const uint8_t* data = gz + P.payload;
    P.len = n - P.payload - 8;
    const size_t len = P.len, readable = len + 8;
    // chunks small enough that the symbols of n_threads of them (2 B per byte of text, in flight at a time) are a small share of
    // the part's text
    const size_t part_len = len / (size_t)P.n_parts;
    if (!chunk_bytes) chunk_bytes = std::min<size_t>((size_t)4 << 20, std::max<size_t>((size_t)512 << 10, part_len / ((size_t)P.n_threads * 10)));
    const size_t per_part = std::max<size_t>(1, part_len / chunk_bytes), total = per_part * (size_t)P.n_parts;
    auto range_bit = [&](size_t k) { return (uint64_t)(len / total * k) * 8; };
    auto range_end = [&](size_t k) { return k + 1 < total ? range_bit(k + 1) : (uint64_t)len * 8; };
    const size_t k0 = per_part * (size_t)P.part, k1 = k0 + per_part;
    std::vector<uint64_t> starts(per_part, UINT64_MAX);
    run_jobs(0, per_part, P.n_threads, [&](size_t i) {
        const size_t k = k0 + i;
        if (k == 0) {
            starts[i] = 0;
            return;
        }
        Tables* scratch = new Tables();
        starts[i] = find_block(data, readable, range_bit(k), range_end(k), *scratch);
        delete scratch;
    });
    uint64_t next_start = UINT64_MAX;                                   // where the first chunk of a later part begins
    {
        Tables* scratch = new Tables();
        for (size_t k = k1; k < total && next_start == UINT64_MAX; ++k) next_start = find_block(data, readable, range_bit(k), range_end(k), *scratch);
        delete scratch;
    }
    P.chunks.clear();
    for (size_t i = 0; i < per_part; ++i)
        if (starts[i] != UINT64_MAX) {
            Chunk c;
            c.start_bit = starts[i];
            P.chunks.push_back(std::move(c));
        }
    for (size_t k = 0; k < P.chunks.size(); ++k) P.chunks[k].stop_bit = k + 1 < P.chunks.size() ? P.chunks[k + 1].start_bit : next_start;
    P.last = next_start == UINT64_MAX;
    if (P.last && P.chunks.empty()) {                                     // (the final block lies in an earlier part's last chunk)
        why = "no block start in the last part";
        return false;
    }
    return true;
}

extern "C" {

int fhx_host_inflate_part(const char* path, int32_t n_threads, int32_t part, int32_t n_parts, fhx_text_part** out) {
    if (!path || !out || n_parts < 1 || part < 0 || part >= n_parts) return FHX_ERR_ARG;
    *out = nullptr;
    fhx_text_part* P = new (std::nothrow) fhx_text_part();
    if (!P) return FHX_ERR_NOMEM;
    *out = P;
    P->path = path;
    P->part = part;
    P->n_parts = n_parts;
    if (n_threads <= 0) n_threads = fhx::usable_cpus();
    P->n_threads = n_threads;
    {
        const int rc = fhx::io_read_file(path, P->gz, P->error);
        if (rc != FHX_OK) return rc;
    }
    std::vector<fhx::GzMember> members;
    if (fhx::io_scan_members(P->gz.data(), P->gz.size(), members) && members.size() > 1) {
        P->error = "a chain of size-tagged members, not one stream";
        return FHX_ERR_UNSUPPORTED;
    }
    const auto t_begin = std::chrono::steady_clock::now();
    auto since = [&]() { return std::chrono::duration<double>(std::chrono::steady_clock::now() - t_begin).count(); };
    std::string why;
    bool ok = part_chunks(*P, why);
    P->seconds[0] = since();
    if (ok) {
        const std::vector<uint8_t> none(kWindow, 0);
        ok = part_run(*P, part == 0 ? none.data() : nullptr, why);       // part 0 knows the window before it: there is none
    }
    P->seconds[1] = since() - P->seconds[0];
    if (!ok) {
        P->chunks.clear();
        P->pieces.clear();
        P->error = "not inflated in parts: " + why;
        return FHX_ERR_UNSUPPORTED;
    }
    if (std::getenv("FHX_TIMING"))
        std::fprintf(stderr, "fhx_host_inflate_part(%s, %d of %d): %zu chunks on %d threads: block starts %.3f s; decode %.3f s; %.1f MB of text%s\n",
                     path, part, n_parts, P->chunks.size(), n_threads, P->seconds[0], P->seconds[1], P->bytes / 1e6,
                     P->resolved ? "" : " (tails only)");
    return FHX_OK;
}

int64_t fhx_text_part_bytes(const fhx_text_part* P) { return P ? P->bytes : 0; }
int32_t fhx_text_part_is_last(const fhx_text_part* P) { return P && P->last ? 1 : 0; }
const char* fhx_text_part_error(const fhx_text_part* P) { return P ? P->error.c_str() : "null part"; }
void fhx_text_part_free(fhx_text_part* P) { delete P; }

int fhx_text_part_tail(const fhx_text_part* P, uint16_t* tail, int64_t cap) {
    if (!P || !tail || cap < kWindow || P->tail.size() != (size_t)kWindow) return FHX_ERR_ARG;
    std::memcpy(tail, P->tail.data(), kWindow * sizeof(uint16_t));
    return FHX_OK;
}

uint32_t fhx_crc32_combine(uint32_t crc_a, uint32_t crc_b, int64_t len_b) { return (uint32_t)crc32_combine(crc_a, crc_b, (z_off_t)len_b); }

// window: the kWindow bytes of text before the part (ignored by part 0).  The part's pieces move into a new fhx_text (fhx_text_free);
// *crc32_out = the CRC-32 of the part's text.
int fhx_text_part_resolve(fhx_text_part* P, const uint8_t* window, int64_t window_bytes, fhx_text** text_out, uint32_t* crc32_out) {
    if (!P || !window || window_bytes != kWindow || !text_out || !crc32_out) return FHX_ERR_ARG;
    *text_out = nullptr;
    const auto t_begin = std::chrono::steady_clock::now();
    if (!P->resolved) {
        std::string why;
        const int64_t announced = P->bytes;
        if (!part_run(*P, window, why) || P->bytes != announced) {
            P->error = "second decode: " + (why.empty() ? std::string("another length than the first") : why);
            return FHX_ERR_UNSUPPORTED;
        }
    }
    fhx_text* x = new (std::nothrow) fhx_text();
    if (!x) return FHX_ERR_NOMEM;
    x->path = P->path;
    x->bytes = P->bytes;
    x->pieces = std::move(P->pieces);
    P->pieces.clear();
    P->chunks.clear();
    P->seconds[2] = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_begin).count();
    x->seconds[1] = P->seconds[0] + P->seconds[1] + P->seconds[2];
    if (std::getenv("FHX_TIMING")) std::fprintf(stderr, "fhx_text_part_resolve(%d of %d): %.3f s\n", P->part, P->n_parts, P->seconds[2]);
    *crc32_out = P->crc;
    *text_out = x;
    return FHX_OK;
}

}  // extern "C"
