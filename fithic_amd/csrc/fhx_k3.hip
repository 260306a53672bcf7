// fhx_k3.hip - K3: myStats.benjamini_hochberg_correction (fithic/myStats.py:24-48) - exact early cutoff, compaction of the surviving p,
// LSD radix sort of their IEEE bit patterns, min(p*N/rank, 1), inclusive max-scan, scatter to row order
// (one of the device translation units of libfithic_mi355x.so; shared declarations: fhx_ctx.hpp)
#include "fhx_ctx.hpp"

namespace fhx {

// ===================================================================================================
// K3: Benjamini-Hochberg as the reference defines it
// ===================================================================================================
// ---- early cutoff -----------------------------------------------------------------------------------------------
// The reference's monotonisation is a FORWARD running max of min(p*N/rank, 1) over ascending p (fithic/myStats.py:31-46):
// once one element reaches 1 every later element has q = 1.  An element with p >= t whose rank is at most C certainly has
// fl(fl(p*N)/rank) >= fl(fl(t*N)/C) (rounding is monotone), so from a coarse histogram of the keys (top 14 bits: sign,
// exponent, 2 mantissa bits) we can name a key T* such that every element >= T* has q = 1 exactly - those rows are not
// sorted at all.  On Hi-C data N (possible pairs) exceeds the number of observed rows, so only the enriched small-p tail
// (typically 10-20 % of the rows) survives the cutoff.  The result is bit-identical to sorting everything.
// Every value that is not NaN is counted (NaN rows take no rank: they sort last and get q = NaN).  p == 1.0 - most rows of
// a Hi-C run - goes through a per-thread counter instead of 64 lanes hitting one LDS word.  Negative values do not occur
// (fhx_bh_array rejects them; bdtrc never returns one); a stray sign bit is clamped into the last bin rather than indexing
// past the table.
__device__ __forceinline__ void top_hist_one(double v, unsigned int* h, unsigned int& ones) {
    if (v == 1.0)
        ++ones;
    else if (v == v)
        atomicAdd(&h[min((unsigned int)(pvalue_key(v) >> TOP_SHIFT), (unsigned int)TOP_BINS - 1u)], 1u);
}

__global__ __launch_bounds__(512) void k3_top_hist(const double* __restrict__ p, int64_t n, unsigned long long* __restrict__ hist) {
    __shared__ unsigned int h[TOP_BINS];
    for (int i = threadIdx.x; i < TOP_BINS; i += 512) h[i] = 0;
    __syncthreads();
    const int64_t n2 = n >> 1;
    const double2* p2 = reinterpret_cast<const double2*>(p);
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    unsigned int ones = 0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n2; i += stride) {
        const double2 v = p2[i];
        top_hist_one(v.x, h, ones);
        top_hist_one(v.y, h, ones);
    }
    if (blockIdx.x == 0 && threadIdx.x == 0 && (n & 1)) top_hist_one(p[n - 1], h, ones);
    ones = (unsigned int)wave_sum_i64((long long)ones);
    if ((threadIdx.x & 63) == 0 && ones) atomicAdd(&h[KEY_ONE >> TOP_SHIFT], ones);
    __syncthreads();
    for (int i = threadIdx.x; i < TOP_BINS; i += 512)
        if (h[i]) atomicAdd(&hist[i], (unsigned long long)h[i]);
}

// smallest bin b (non-empty) with fl(fl(lower_edge(b) * N) / (#keys in bins 0..b)) >= 1 -> cutoff key = b << TOP_SHIFT
__host__ __device__ inline bool bin_saturates(int b, unsigned long long cum_incl, double n_tests) {
    const unsigned long long edge_bits = (unsigned long long)b << TOP_SHIFT;
    double edge;
    memcpy(&edge, &edge_bits, sizeof(edge));
    const double v = edge * n_tests / (double)cum_incl;
    return v >= 1.0;
}

// (dense_flag: 1 when at least dense_min values lie below the cutoff - the pass then takes q through the dense array, k3_compact<true>)
// (host: the engine's own pass - the survivors' number goes straight to the host's pinned words, followed by the ticket the host
// waits for (fhx_ctx::h_flags: no copy dispatched behind this kernel), and the compaction's counter is zeroed here (no fill before it))
struct CutoffToHost {
    volatile unsigned long long* words = nullptr;    // [0] the ticket, [1] the number of values below the cutoff
    unsigned long long ticket = 0;
    unsigned int* done = nullptr;
    unsigned long long* zero_me = nullptr;
};
__global__ __launch_bounds__(1024) void k3_cutoff(const unsigned long long* __restrict__ hist, double n_tests,
                                                  unsigned long long* __restrict__ cutoff_key, unsigned long long* __restrict__ n_below,
                                                  unsigned long long dense_min = ~0ull, unsigned long long* __restrict__ dense_flag = nullptr,
                                                  CutoffToHost host = CutoffToHost{}) {
    __shared__ unsigned long long part[1024];
    __shared__ unsigned long long below;
    __shared__ unsigned int best;
    constexpr int PER = TOP_BINS / 1024;
    unsigned long long local[PER];
    unsigned long long sum = 0;
    for (int k = 0; k < PER; ++k) {
        local[k] = hist[threadIdx.x * PER + k];
        sum += local[k];
    }
    // exclusive prefix of the 1024 partial sums: wave scan + the 16 wave totals (one thread walking all 1024 took 12 of this
    // kernel's 16 us - a fixed cost of every pass)
    if (threadIdx.x == 0) {
        best = TOP_BINS;
        below = 0ull;
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    unsigned long long incl = sum;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const unsigned long long o = __shfl_up(incl, off, 64);
        if (lane >= off) incl += o;
    }
    if (lane == 63) part[wave] = incl;
    __syncthreads();
    unsigned long long cum = incl - sum;
    for (int w = 0; w < wave; ++w) cum += part[w];
    const unsigned long long cum0 = cum;
    for (int k = 0; k < PER; ++k) {
        cum += local[k];
        const int b = threadIdx.x * PER + k;
        if (local[k] && bin_saturates(b, cum, n_tests)) atomicMin(&best, (unsigned int)b);
    }
    __syncthreads();
    // the values below the cutoff key = the counts of the bins below the cutoff bin (all counted values when nothing saturates):
    // what k3_compact will keep, known before it has run (the host sizes the sort while the compaction is still under way)
    if (n_below) {
        unsigned long long mine = 0;
        cum = cum0;
        for (int k = 0; k < PER; ++k)
            if ((unsigned int)(threadIdx.x * PER + k) < best) mine += local[k];
        if (mine) atomicAdd(&below, mine);
        __syncthreads();
        if (threadIdx.x == 0) {
            *n_below = below;
            if (dense_flag) *dense_flag = below >= dense_min ? 1ull : 0ull;
            if (host.words) const_cast<unsigned long long*>(host.words)[1] = below;
            if (host.zero_me) *host.zero_me = 0ull;
        }
    }
    if (threadIdx.x == 0)
        *cutoff_key = (best < (unsigned int)TOP_BINS) ? ((unsigned long long)best << TOP_SHIFT) : KEY_KEEP_ALL;
    if (host.words) publish_ticket(host.done, host.words, host.ticket);
}

// compaction: keys of the rows below the cutoff key (IEEE bit pattern: all p are >= 0, so unsigned order is
// numeric order); rows at or above it get q = 1 and NaN rows get q = NaN right here.
//
// DENSE (round 6; the engine's own pass when many rows survive - rank_and_adjust): q is not touched here.  A survivor's value is its
// COMPACT INDEX (its slot in keys[0]) instead of its row, so that K3c scatters the q of the survivors into a dense array of their
// number - an eighth to a half of the q column: the scattered 8-byte stores stay in the L2s / the 256 MB Infinity Cache instead of
// each filling and writing back a 32-byte sector of HBM - and k3_fill_q then writes the whole q column once, in row order, from
// two bits per row left here (kept / NaN: `mask`, four ballots per 128 rows) + the wave's first slot (`wave_slot`).
struct DenseQ {
    double* dense = nullptr;                 // q of the survivors by compact index
    unsigned long long* mask = nullptr;      // per wave chunk (1024 rows) and step h: keep even rows, keep odd rows, NaN even, NaN odd
    unsigned long long* wave_slot = nullptr; // per wave chunk: compact index of its first survivor
    const unsigned long long* flag = nullptr; // the device's decision (k3_cutoff): non-zero = this pass goes through the dense array
};

template <bool DENSE>
__global__ __launch_bounds__(CP_THREADS) void k3_compact(const double* __restrict__ p, int64_t n,
                                                           unsigned long long* __restrict__ keys,
                                                           unsigned int* __restrict__ vals, double* __restrict__ q,
                                                           unsigned long long* __restrict__ counter,
                                                           const unsigned long long* __restrict__ cutoff_key, DenseQ dq,
                                                           bool q_is_ones = false, int tiles_per_wg = 1) {
    // rows at or above the cutoff key (see above) have q = 1 and are not sorted; NaN rows get q = NaN.
    // One returning atomic on the survivors' counter per tile - and those retire at ~11 ns each on one address (a same-address
    // atomic per 256 rows capped this kernel at 88 M atomics/s; per 4096 rows it still was 0.40 of 0.42 ms on C3).  A workgroup
    // therefore takes tiles_per_wg consecutive tiles of 16 384 rows, and a tile with few survivors (the rule on a Hi-C map: 8 per
    // tile on C3) leaves them in an LDS strip instead of asking for slots; the strip goes out behind the workgroup's last tile
    // with ONE atomic (a tile that does not fit it asks as before).  The order of the survivors among themselves is free.
    __shared__ unsigned int wave_cnt[CP_WAVES];
    __shared__ unsigned long long block_base;
    __shared__ unsigned long long strip_k[CP_STRIP];
    __shared__ unsigned int strip_v[CP_STRIP];
    __shared__ unsigned int strip_n, strip_at;                   // entries in the strip; this tile's first one, or ~0u: global slots
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const unsigned long long lane_lt = (1ull << lane) - 1ull;
    const int64_t tiles = (n + CP_TILE - 1) / CP_TILE;
    if (dq.flag && (*dq.flag != 0ull) != DENSE) return;          // both variants are launched; the one the device chose runs
    const unsigned long long cutoff = *cutoff_key;
    const double2* p2 = reinterpret_cast<const double2*>(p);
    double2* q2 = reinterpret_cast<double2*>(q);
    const bool strip_ok = !DENSE && tiles_per_wg > 1;            // (the dense variant's payload IS the slot: it must be known here)
    if (threadIdx.x == 0) strip_n = 0u;
    const int64_t groups = (tiles + tiles_per_wg - 1) / tiles_per_wg;
    for (int64_t grp = blockIdx.x; grp < groups; grp += gridDim.x) {
    for (int64_t t = grp * tiles_per_wg; t < tiles && t < (grp + 1) * tiles_per_wg; ++t) {
        const int64_t wave_base = t * CP_TILE + (int64_t)wave * (64 * CP_ITEMS);
        // two consecutive rows per lane and step: 16-byte loads of p and (for the rows that are not ranked: nearly all) 16-byte
        // stores of q
        double v[CP_ITEMS];
        unsigned int before[CP_ITEMS];
        unsigned long long keepmask = 0;          // bit r: this lane keeps item r
        unsigned int run = 0;
#pragma unroll
        for (int h = 0; h < CP_ITEMS / 2; ++h) {
            const int64_t i = wave_base + (int64_t)(h * 64 + lane) * 2;
            bool keep0 = false, keep1 = false;
            v[2 * h] = v[2 * h + 1] = 1.0;
            if (i + 1 < n) {
                const double2 w = p2[i >> 1];
                v[2 * h] = w.x;
                v[2 * h + 1] = w.y;
                keep0 = (w.x == w.x) && (pvalue_key(w.x) < cutoff);        // false for NaN; p >= 1 stays when nothing saturates
                keep1 = (w.y == w.y) && (pvalue_key(w.y) < cutoff);
                // both q of the pair in one 16-byte store, kept rows included: bh_apply overwrites those later on this stream
                // (partial 8-byte stores around every kept row cost 0.13 ms per 1.2e8 rows with 12 % of them kept)
                // (q_is_ones: the column was filled with 1.0 behind K1 - k1_prezero -, only a NaN is left to store)
                if (!DENSE && (!q_is_ones || w.x != w.x || w.y != w.y)) q2[i >> 1] = make_double2((w.x == w.x) ? 1.0 : w.x, (w.y == w.y) ? 1.0 : w.y);
            } else if (i < n) {                                            // the last row of an odd count
                v[2 * h] = p[i];
                keep0 = (v[2 * h] == v[2 * h]) && (pvalue_key(v[2 * h]) < cutoff);
                if (!DENSE && !keep0 && !(q_is_ones && v[2 * h] == v[2 * h])) q[i] = (v[2 * h] == v[2 * h]) ? 1.0 : v[2 * h];
            }
            const unsigned long long m0 = __ballot(keep0), m1 = __ballot(keep1);
            if (DENSE) {
                const unsigned long long n0 = __ballot(v[2 * h] != v[2 * h]), n1 = __ballot(v[2 * h + 1] != v[2 * h + 1]);
                if (lane == 0) {
                    ulonglong2* m = reinterpret_cast<ulonglong2*>(dq.mask + ((t * CP_WAVES + wave) * (CP_ITEMS / 2) + h) * 4);
                    m[0] = make_ulonglong2(m0, m1);
                    m[1] = make_ulonglong2(n0, n1);
                }
            }
            before[2 * h] = run + __popcll(m0 & lane_lt);
            run += __popcll(m0);
            before[2 * h + 1] = run + __popcll(m1 & lane_lt);
            run += __popcll(m1);
            if (keep0) keepmask |= (1ull << (2 * h));
            if (keep1) keepmask |= (1ull << (2 * h + 1));
        }
        if (lane == 0) wave_cnt[wave] = run;
        __syncthreads();
        if (threadIdx.x == 0) {
            unsigned int tot = 0;
            for (int w = 0; w < CP_WAVES; ++w) {
                const unsigned int c = wave_cnt[w];
                wave_cnt[w] = tot;
                tot += c;
            }
            if (strip_ok && strip_n + tot <= (unsigned int)CP_STRIP) {
                strip_at = strip_n;
                strip_n += tot;
            } else {
                strip_at = ~0u;
                block_base = tot ? atomicAdd(counter, (unsigned long long)tot) : 0ull;
            }
        }
        __syncthreads();
        if (strip_at != ~0u) {
            const unsigned int at = strip_at + wave_cnt[wave];
#pragma unroll
            for (int r = 0; r < CP_ITEMS; ++r) {
                if ((keepmask >> r) & 1ull) {
                    strip_k[at + before[r]] = pvalue_key(v[r]);
                    strip_v[at + before[r]] = (unsigned int)(wave_base + (int64_t)((r >> 1) * 64 + lane) * 2 + (r & 1));
                }
            }
        } else {
            const unsigned long long base = block_base + wave_cnt[wave];
            if (DENSE && lane == 0) dq.wave_slot[t * CP_WAVES + wave] = base;
#pragma unroll
            for (int r = 0; r < CP_ITEMS; ++r) {
                if ((keepmask >> r) & 1ull) {
                    keys[base + before[r]] = pvalue_key(v[r]);
                    vals[base + before[r]] = DENSE ? (unsigned int)(base + before[r])
                                                   : (unsigned int)(wave_base + (int64_t)((r >> 1) * 64 + lane) * 2 + (r & 1));
                }
            }
        }
        __syncthreads();
    }
        // the strip of this group of tiles: one atomic, then out
        if (strip_ok) {
            const unsigned int held = strip_n;
            if (held) {
                if (threadIdx.x == 0) block_base = atomicAdd(counter, (unsigned long long)held);
                __syncthreads();
                for (unsigned int i = threadIdx.x; i < held; i += CP_THREADS) {
                    keys[block_base + i] = strip_k[i];
                    vals[block_base + i] = strip_v[i];
                }
                __syncthreads();
                if (threadIdx.x == 0) strip_n = 0u;
                __syncthreads();
            }
        }
    }
}

// The whole q column in row order, once: 1.0, the row's own NaN, or - for the rows k3_compact<true> kept - the survivor's q from the
// dense array (consecutive kept rows of a wave step read consecutive entries).  Same tiles and wave chunks as k3_compact.
__global__ __launch_bounds__(CP_THREADS) void k3_fill_q(const double* __restrict__ p, int64_t n, DenseQ dq, double* __restrict__ q) {
    if (dq.flag && *dq.flag == 0ull) return;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const unsigned long long lane_lt = (1ull << lane) - 1ull;
    const int64_t tiles = (n + CP_TILE - 1) / CP_TILE;
    double2* q2 = reinterpret_cast<double2*>(q);
    for (int64_t t = blockIdx.x; t < tiles; t += gridDim.x) {
        const int64_t chunk = t * CP_WAVES + wave;
        const int64_t wave_base = chunk * (64 * CP_ITEMS);
        if (wave_base >= n) continue;
        const double* mine = dq.dense + dq.wave_slot[chunk];
        const ulonglong2* m = reinterpret_cast<const ulonglong2*>(dq.mask + chunk * (CP_ITEMS / 2) * 4);
        unsigned int run = 0;
#pragma unroll
        for (int h = 0; h < CP_ITEMS / 2; ++h) {
            const ulonglong2 keep = m[2 * h], nan = m[2 * h + 1];
            const int64_t i = wave_base + (int64_t)(h * 64 + lane) * 2;
            const unsigned int b0 = run + __popcll(keep.x & lane_lt);
            run += __popcll(keep.x);
            const unsigned int b1 = run + __popcll(keep.y & lane_lt);
            run += __popcll(keep.y);
            double q0 = 1.0, q1 = 1.0;
            if ((keep.x >> lane) & 1ull) q0 = mine[b0];
            if ((keep.y >> lane) & 1ull) q1 = mine[b1];
            if ((nan.x >> lane) & 1ull) q0 = p[i];
            if ((nan.y >> lane) & 1ull) q1 = p[i + 1];
            if (i + 1 < n)
                q2[i >> 1] = make_double2(q0, q1);
            else if (i < n)
                q[i] = q0;
        }
    }
}

// per-workgroup digit counts for one radix pass; workgroup b owns the contiguous chunk [b*chunk, (b+1)*chunk)
template <int BITS>
__global__ __launch_bounds__(SORT_THREADS) void rs_count(const unsigned long long* __restrict__ keys,
                                                         const unsigned long long* __restrict__ n_ptr, int shift,
                                                         unsigned int* __restrict__ block_hist) {
    constexpr int RADIX = 1 << BITS;                  // (shadows the header's 11-bit constant)
    __shared__ unsigned int h[RADIX];
    const int64_t n = (int64_t)*n_ptr;
    const int64_t nblk = gridDim.x;                   // the launch decides how many chunks there are (sort_blocks_for)
    const int64_t chunk = ((n + nblk - 1) / nblk + SORT_TILE - 1) / SORT_TILE * SORT_TILE;
    const int64_t beg = (int64_t)blockIdx.x * chunk, end = min(n, beg + chunk);
    for (int d = threadIdx.x; d < RADIX; d += SORT_THREADS) h[d] = 0;
    __syncthreads();
    // two keys per lane per step (16-byte loads); chunk starts are multiples of SORT_TILE, so they are aligned
    const int64_t len = end > beg ? end - beg : 0;           // workgroups past the end own nothing
    const int64_t n2 = len >> 1;
    const ulonglong2* k2 = reinterpret_cast<const ulonglong2*>(keys + beg);
    for (int64_t i = threadIdx.x; i < n2; i += SORT_THREADS) {
        const ulonglong2 k = k2[i];
        atomicAdd(&h[(k.x >> shift) & (RADIX - 1)], 1u);
        atomicAdd(&h[(k.y >> shift) & (RADIX - 1)], 1u);
    }
    if (threadIdx.x == 0 && (len & 1)) atomicAdd(&h[(keys[end - 1] >> shift) & (RADIX - 1)], 1u);
    __syncthreads();
    for (int d = threadIdx.x; d < RADIX; d += SORT_THREADS) block_hist[(size_t)d * nblk + blockIdx.x] = h[d];   // digit-major
}

// exclusive scan of the digit-major (RADIX x SORT_BLOCKS) count matrix along the workgroup axis: one
// workgroup per digit, one thread per sorting workgroup (coalesced row access); digit totals go to
// digit_total[], their own exclusive scan is folded into rs_scatter's prologue.
__global__ __launch_bounds__(SORT_BLOCKS) void rs_scan(unsigned int* __restrict__ block_hist,
                                                       unsigned int* __restrict__ digit_total, int nblk) {
    __shared__ unsigned int wsum[SORT_BLOCKS / 64];
    unsigned int* row = block_hist + (size_t)blockIdx.x * nblk;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const bool live = (int)threadIdx.x < nblk;         // blockDim.x = nblk rounded up to whole waves
    const unsigned int mine = live ? row[threadIdx.x] : 0u;
    unsigned int incl = mine;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const unsigned int o = __shfl_up(incl, off, 64);
        if (lane >= off) incl += o;
    }
    if (lane == 63) wsum[wave] = incl;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned int acc = 0;
        const int waves = (blockDim.x + 63) / 64;
        for (int w = 0; w < waves; ++w) {
            const unsigned int c = wsum[w];
            wsum[w] = acc;
            acc += c;
        }
        digit_total[blockIdx.x] = acc;
    }
    __syncthreads();
    if (live) row[threadIdx.x] = wsum[wave] + incl - mine;
}

// stable scatter of one radix pass.  Each wave owns a contiguous sub-range of the tile and ranks its keys
// with wave-private LDS digit counters (no atomics: one leader lane per distinct digit, found with RADIX_BITS+1
// ballots); the tile is then reordered through LDS so that the global writes of equal-digit runs are contiguous.
//
// LDS decides how many workgroups a CU holds, and with them how much of the ranking's latency (dependent LDS reads and writes,
// 12 ballots per key) is hidden.  Round 3's layout - counters 16 KB + keys 32 KB + payloads 16 KB + two offset tables - came to
// 81 936 B: ONE 256-thread workgroup per CU, one wave per SIMD, 271 us per pass over 1.5e7 keys (1.3 TB/s,
// profiles/history/r04_b_od1_kernel_stats.txt).  Now 512 threads per tile of 4096 keys and the per-wave counters share their 32 KB with
// the staged keys (a key's slot is in a register by the time the counters die): 64 KB, two workgroups = 16 waves per CU.
// (Measured and dropped, profiles/history/r04_e_rs_ab.txt: no staging at all - keys written straight from registers to
// global_base[digit] + rank - is 45 % slower, the tile-wide reordering is what coalesces the writes of the passes over the
// exponent bits; squeezing the kernel to 80 VGPRs for a third workgroup per CU spills and is slower still.)

template <int SCAT_THREADS, int WPE, int BITS>
__global__ __launch_bounds__(SCAT_THREADS) __attribute__((amdgpu_waves_per_eu(WPE, WPE))) void rs_scatter(
    const unsigned long long* __restrict__ keys_in, const unsigned int* __restrict__ vals_in, unsigned long long* __restrict__ keys_out,
    unsigned int* __restrict__ vals_out, const unsigned long long* __restrict__ n_ptr, int shift,
    const unsigned int* __restrict__ block_hist, const unsigned int* __restrict__ digit_total) {
    constexpr int RADIX = 1 << BITS, RADIX_BITS = BITS;       // (shadow the header's 11-bit constants)
    constexpr int SCAT_WAVES = SCAT_THREADS / 64, TILE = SCAT_THREADS * SCAT_ITEMS, PER = RADIX >= SCAT_THREADS ? RADIX / SCAT_THREADS : 1;
    static_assert(SORT_TILE % TILE == 0, "a workgroup's chunk (a multiple of SORT_TILE keys) is whole tiles");
    static_assert(SCAT_WAVES * RADIX * 2 <= TILE * 8, "the per-wave counters fit the block that later stages the keys");
    __shared__ __attribute__((aligned(16))) unsigned char stage_raw[TILE * 8];      // per-wave counters, then the tile's keys
    __shared__ unsigned int s_vals[TILE];
    __shared__ unsigned int tile_start[RADIX];                 // first tile-local slot of each digit
    __shared__ unsigned int global_base[RADIX];                // running global offset of each digit
    __shared__ unsigned int wave_tmp[SCAT_WAVES];
    unsigned short (*wave_digit)[RADIX] = reinterpret_cast<unsigned short (*)[RADIX]>(stage_raw);   // counts (<= 512) -> exclusive offsets over the waves
    unsigned long long* s_keys = reinterpret_cast<unsigned long long*>(stage_raw);
    const int64_t n = (int64_t)*n_ptr;
    const int64_t nblk = gridDim.x;
    const int64_t chunk = ((n + nblk - 1) / nblk + SORT_TILE - 1) / SORT_TILE * SORT_TILE;
    const int64_t beg = (int64_t)blockIdx.x * chunk, end = min(n, beg + chunk);
    if (beg >= end) return;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const unsigned long long lane_lt = (1ull << lane) - 1ull;
    // exclusive scan of RADIX entries of `a` (PER consecutive ones per thread); two barriers inside
    auto scan_radix = [&](unsigned int* a) {
        unsigned int v[PER];
        unsigned int mine = 0;
#pragma unroll
        for (int k = 0; k < PER; ++k) {
            v[k] = (int)threadIdx.x * PER + k < RADIX ? a[threadIdx.x * PER + k] : 0u;
            mine += v[k];
        }
        const unsigned int incl = wave_incl_sum_u32(mine);
        if (lane == 63) wave_tmp[wave] = incl;
        __syncthreads();
        unsigned int excl = incl - mine;
        for (int w = 0; w < wave; ++w) excl += wave_tmp[w];
#pragma unroll
        for (int k = 0; k < PER; ++k) {
            if ((int)threadIdx.x * PER + k < RADIX) a[threadIdx.x * PER + k] = excl;
            excl += v[k];
        }
        __syncthreads();
    };
    for (int d = threadIdx.x; d < RADIX; d += SCAT_THREADS) tile_start[d] = digit_total[d];
    __syncthreads();
    scan_radix(tile_start);
    for (int d = threadIdx.x; d < RADIX; d += SCAT_THREADS)
        global_base[d] = tile_start[d] + block_hist[(size_t)d * nblk + blockIdx.x];
    __syncthreads();
    for (int64_t tile = beg; tile < end; tile += TILE) {
        for (int i = threadIdx.x; i < SCAT_WAVES * RADIX / 2; i += SCAT_THREADS) reinterpret_cast<unsigned int*>(stage_raw)[i] = 0u;
        unsigned long long key[SCAT_ITEMS];
        unsigned int val[SCAT_ITEMS];
        unsigned int slot[SCAT_ITEMS];
        const int64_t wave_base = tile + (int64_t)wave * (64 * SCAT_ITEMS);
#pragma unroll
        for (int r = 0; r < SCAT_ITEMS; ++r) {
            const int64_t i = wave_base + r * 64 + lane;
            const bool live = i < end;
            key[r] = live ? keys_in[i] : ~0ull;
            val[r] = live ? vals_in[i] : 0u;
        }
        __syncthreads();
#pragma unroll
        for (int r = 0; r < SCAT_ITEMS; ++r) {
            const bool live = (wave_base + r * 64 + lane) < end;
            const unsigned int digit = (unsigned int)(key[r] >> shift) & (RADIX - 1);
            // lanes holding the same digit (dead lanes form their own group through the extra bit)
            unsigned long long same = ~0ull;
            const unsigned int tag = digit | (live ? 0u : RADIX);
#pragma unroll
            for (int b = 0; b <= RADIX_BITS; ++b) {
                const unsigned long long m = __ballot((tag >> b) & 1u);
                same &= ((tag >> b) & 1u) ? m : ~m;
            }
            const unsigned int before = __popcll(same & lane_lt);
            unsigned int old = 0;
            if (live) old = wave_digit[wave][digit];          // every lane of the group reads the same counter ...
            slot[r] = old + before;                           // rank among the wave's keys of this digit, so far
            __builtin_amdgcn_wave_barrier();
            if (live && before == 0) wave_digit[wave][digit] = (unsigned short)(old + __popcll(same));   // ... its leader bumps it
            __builtin_amdgcn_wave_barrier();
        }
        __syncthreads();
        // exclusive offsets: per digit across waves, then across digits
        for (int d = threadIdx.x * PER; d < (threadIdx.x + 1) * PER && d < RADIX; ++d) {
            unsigned int acc = 0;
#pragma unroll
            for (int w = 0; w < SCAT_WAVES; ++w) {
                const unsigned int c = wave_digit[w][d];
                wave_digit[w][d] = (unsigned short)acc;
                acc += c;
            }
            tile_start[d] = acc;                              // digit total for now
        }
        __syncthreads();
        scan_radix(tile_start);
        const int live_in_tile = (int)min<int64_t>(TILE, end - tile);
#pragma unroll
        for (int r = 0; r < SCAT_ITEMS; ++r) {
            const unsigned int digit = (unsigned int)(key[r] >> shift) & (RADIX - 1);
            if (wave_base + r * 64 + lane < end) slot[r] += tile_start[digit] + wave_digit[wave][digit];
        }
        __syncthreads();                                      // the counters are dead: their block now stages the keys
#pragma unroll
        for (int r = 0; r < SCAT_ITEMS; ++r)
            if (wave_base + r * 64 + lane < end) {
                s_keys[slot[r]] = key[r];
                s_vals[slot[r]] = val[r];
            }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < SCAT_ITEMS; ++j) {
            const int s = threadIdx.x + j * SCAT_THREADS;
            if (s < live_in_tile) {
                const unsigned long long k = s_keys[s];
                const unsigned int digit = (unsigned int)(k >> shift) & (RADIX - 1);
                const unsigned int dst = global_base[digit] + (s - tile_start[digit]);
                keys_out[dst] = k;
                vals_out[dst] = s_vals[s];
            }
        }
        __syncthreads();
        // advance the running global offsets by this tile's digit totals
        for (int d = threadIdx.x; d < RADIX; d += SCAT_THREADS) {
            const unsigned int nxt = (d + 1 < RADIX) ? tile_start[d + 1] : (unsigned int)live_in_tile;
            global_base[d] += nxt - tile_start[d];
        }
        __syncthreads();
    }
}

#include "fhx_onesweep.inc"

// ---- small survivor sets: two launches instead of eighteen ---------------------------------------------------------------
// On Hi-C data whose counts follow the model closely (C3-synth: 77 k of 1.5e8 rows below the cutoff) and on every shard of a
// strong-scaling run the radix sort is all fixed cost: 18 launches of 5-15 us for microseconds of work (0.16 ms per pass,
// profiles/history/r03_z_c2_timeline.txt - the launches already run back to back, it is the kernels' own floor).  Up to KS_MAX_KEYS
// survivors are instead sorted tile by tile in LDS (ks_tile_sort: a bitonic network over 4096 (key, row) pairs, the strides
// below 8 in registers) and the sorted tiles merged by rank (ks_merge_tiles: an element's place is its position in its own tile
// plus, for every other tile, the number of elements below it - at or below it for earlier tiles).  Neither step is stable and
// neither needs to be: equal p-values share one q whatever their order (the first of them has the largest p*N/rank and the
// running maximum carries it over the rest), and no later pass depends on the order - unlike the passes of the LSD sort.
constexpr int KS_TILE = 4096;
constexpr int KS_THREADS = 512;
constexpr int KS_PER = KS_TILE / KS_THREADS;                    // 8 consecutive elements per thread in the register stages
constexpr int KS_MAX_TILES = 32;
constexpr int KS_MAX_KEYS = KS_MAX_TILES * KS_TILE;             // 131 072

__global__ __launch_bounds__(KS_THREADS) void ks_tile_sort(const unsigned long long* __restrict__ keys_in, const unsigned int* __restrict__ vals_in,
                                                           const unsigned long long* __restrict__ n_ptr, unsigned long long* __restrict__ keys_out,
                                                           unsigned int* __restrict__ vals_out) {
    // The tile lives in LDS; a compare-exchange at stride j pairs element i with i + j.  Strides of 8 and more: four pairs per
    // thread and a barrier per stride; the strides 4, 2, 1 that end every block size: one visit of a thread's eight consecutive
    // elements, in registers.  (Measured, profiles/history/r04_l_small_ab.txt: keeping the elements in registers throughout and reaching
    // the partner lane with wave shuffles for strides 8..256 - 39 of the 78 stages without a barrier - is slower, 75 us
    // against 45 for three tiles: 24 ds_bpermute per stage cost more than the barriers they save.)
    __shared__ unsigned long long sk[KS_TILE];
    __shared__ unsigned int sv[KS_TILE];
    const int64_t n = (int64_t)*n_ptr;
    const int64_t base = (int64_t)blockIdx.x * KS_TILE;
    if (base >= n) return;
#pragma unroll
    for (int r = 0; r < KS_PER; ++r) {
        const int s = threadIdx.x + r * KS_THREADS;
        const bool live = base + s < n;
        sk[s] = live ? keys_in[base + s] : ~0ull;                // padding sorts last (keys are IEEE patterns of p >= 0: never all ones)
        sv[s] = live ? vals_in[base + s] : 0u;
    }
    __syncthreads();
    auto register_stages = [&](int k) {
        unsigned long long a[KS_PER];
        unsigned int b[KS_PER];
        const int first = threadIdx.x * KS_PER;
#pragma unroll
        for (int e = 0; e < KS_PER; ++e) {
            a[e] = sk[first + e];
            b[e] = sv[first + e];
        }
#pragma unroll
        for (int j = KS_PER / 2; j > 0; j >>= 1) {
            if (j < k) {
#pragma unroll
                for (int e = 0; e < KS_PER; ++e) {
                    if ((e & j) == 0) {
                        const bool up = ((first + e) & k) == 0;
                        if ((a[e] > a[e + j]) == up) {
                            const unsigned long long ta = a[e]; a[e] = a[e + j]; a[e + j] = ta;
                            const unsigned int tb = b[e]; b[e] = b[e + j]; b[e + j] = tb;
                        }
                    }
                }
            }
        }
#pragma unroll
        for (int e = 0; e < KS_PER; ++e) {
            sk[first + e] = a[e];
            sv[first + e] = b[e];
        }
    };
    for (int k = 2; k <= KS_TILE; k <<= 1) {
        for (int j = k >> 1; j >= KS_PER; j >>= 1) {             // partners in other threads: through LDS, four pairs per thread
#pragma unroll
            for (int r = 0; r < KS_PER / 2; ++r) {
                const int t = threadIdx.x + r * KS_THREADS;
                const int i = ((t & ~(j - 1)) << 1) | (t & (j - 1));
                const int l = i + j;
                const unsigned long long x = sk[i], y = sk[l];
                const bool up = (i & k) == 0;
                if ((x > y) == up) {
                    sk[i] = y;
                    sk[l] = x;
                    const unsigned int vx = sv[i];
                    sv[i] = sv[l];
                    sv[l] = vx;
                }
            }
            __syncthreads();
        }
        register_stages(k);
        __syncthreads();
    }
#pragma unroll
    for (int r = 0; r < KS_PER; ++r) {
        const int s = threadIdx.x + r * KS_THREADS;
        if (base + s < n) {
            keys_out[base + s] = sk[s];
            vals_out[base + s] = sv[s];
        }
    }
}

// sorted tiles of KS_TILE elements -> one sorted array
__global__ __launch_bounds__(256) void ks_merge_tiles(const unsigned long long* __restrict__ keys_in, const unsigned int* __restrict__ vals_in,
                                                      const unsigned long long* __restrict__ n_ptr, unsigned long long* __restrict__ keys_out,
                                                      unsigned int* __restrict__ vals_out) {
    const int64_t n = (int64_t)*n_ptr;
    const int tiles = (int)((n + KS_TILE - 1) / KS_TILE);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const unsigned long long k = keys_in[i];
        const int mine = (int)(i / KS_TILE);
        int64_t pos = i - (int64_t)mine * KS_TILE;
        for (int t = 0; t < tiles; ++t) {
            if (t == mine) continue;
            int64_t lo = (int64_t)t * KS_TILE, hi = min(n, lo + (int64_t)KS_TILE);
            const int64_t first = lo;
            const bool take_equal = t < mine;                    // equal keys: the earlier tile's come first
            while (lo < hi) {
                const int64_t mid = (lo + hi) >> 1;
                const unsigned long long v = keys_in[mid];
                if (v < k || (take_equal && v == k))
                    lo = mid + 1;
                else
                    hi = mid;
            }
            pos += lo - first;
        }
        keys_out[pos] = k;
        vals_out[pos] = vals_in[i];
    }
}

// BH value of sorted position i (0-based, global rank = rank0 + i + 1): min(p*N/rank, 1), myStats.py:35-38
// x / r when the quotient is subnormal.  The division the compiler expands (v_div_scale / v_rcp / v_div_fmas / v_div_fixup) is
// correctly rounded for normal quotients; a subnormal one is rounded twice - once at 53 bits, once when v_div_fmas scales it back
// down - and comes out one unit of 2^-1074 off in a case in some hundreds (the large-sort fuzz found it: p = 1.24e-314, N = 1,
// rank 1646).  Deep maps put such quotients in the file: the handful of subnormal p above thousands of exact zeros.  Here the
// quotient is rounded ONCE, in integers: with g = 2^-1074, x = q0 r + rem exactly (one fma: the remainder is a multiple of g
// below 2^34 g), and the nearest multiple of g to x / r is q0 + round_half_even(rem / (r g)) g.
__device__ __noinline__ double div_to_subnormal(double x, double r, double q0) {
    const double rem = __builtin_fma(-q0, r, x);
    const double up = 0x1p537;                                 // 2^1074 in two exact steps
    const long long R = (long long)(rem * up * up);            // rem / g
    const long long Q = (long long)(fabs(q0) * up * up);       // |q0| / g  (< 2^52)
    const long long ri = (long long)r;
    const long long Rs = x < 0.0 ? -R : R;                     // work on |x| / r
    long long kf = Rs / ri;
    long long rr = Rs - kf * ri;
    if (rr < 0) {
        rr += ri;
        --kf;
    }
    long long k = kf;
    if (2 * rr > ri || (2 * rr == ri && ((Q + kf) & 1ll))) ++k;
    const double mag = (double)(Q + k) * 0x1p-537 * 0x1p-537;  // exact: an integer below 2^53 times g
    return x < 0.0 ? -mag : mag;
}

__device__ __forceinline__ double bh_value(unsigned long long key_bits, double n_tests, double rank) {
    const double pv = __longlong_as_double((long long)key_bits);
    const double x = pv * n_tests;
    double v = x / rank;                      // (p*N)/(i+1): mul then div, never fused
    if (__builtin_expect(fabs(v) < 0x1p-1022 && x != 0.0 && fabs(x) < 0x1p-900 && rank >= 1.0 && rank < 0x1p53, 0))
        v = div_to_subnormal(x, rank, v);
    if (1.0 < v || pv == 1.0) v = 1.0;        // min(bh, 1); p == 1.0 is 1.0 whatever N / rank says (myStats.py:33-34)
    // The reference's running maximum starts at 0 (myStats.py:30): invisible while N > 0, but fit_Spline can pass a NEGATIVE
    // number of tests (possible-pair counts go negative with unmappable loci) and then every bh value is negative
    // and every q is 0 (tests/golden/f12_bh_nonpositive_N.npz).  Clamping the values is the same running maximum.
    if (v < 0.0) v = 0.0;
    return v;
}


__device__ __forceinline__ double wave_incl_max(double v, int lane) {
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const double o = __shfl_up(v, off, 64);
        if (lane >= off) v = fmax(v, o);
    }
    return v;
}

__global__ __launch_bounds__(BH_THREADS) void bh_tile_max(const unsigned long long* __restrict__ keys,
                                                          const unsigned long long* __restrict__ n_ptr, int64_t n_fixed,
                                                          double n_tests, double rank0, double* __restrict__ tile_max) {
    __shared__ double wmax[BH_THREADS / 64];
    const int64_t n = n_ptr ? (int64_t)*n_ptr : n_fixed;
    const int64_t base = (int64_t)blockIdx.x * BH_TILE;
    if (base >= n) return;
    double m = 0.0;
#pragma unroll
    for (int r = 0; r < BH_ITEMS; ++r) {
        const int64_t i = base + r * BH_THREADS + threadIdx.x;
        if (i < n) m = fmax(m, bh_value(keys[i], n_tests, rank0 + (double)(i + 1)));
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) m = fmax(m, __shfl_down(m, off, 64));
    if ((threadIdx.x & 63) == 0) wmax[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = wmax[0];
        for (int w = 1; w < BH_THREADS / 64; ++w) t = fmax(t, wmax[w]);
        tile_max[blockIdx.x] = t;
    }
}

// exclusive running max over the tile maxima (carry-in of every tile); one workgroup
// (n_bound / fault: the host sized the sort and this grid for n_bound keys - the histogram's count when it did not wait for the
// device counter; more keys than that would have been dropped without a word, so the one workgroup here leaves the counter's value in
// the context's fault word - pinned memory, read by every call that hands results to the host: check_fault)
__global__ __launch_bounds__(1024) void bh_scan_tiles(double* __restrict__ tile_max,
                                                      const unsigned long long* __restrict__ n_ptr, int64_t n_fixed,
                                                      double carry_in, double* __restrict__ total_max, int64_t n_bound = -1,
                                                      unsigned long long* __restrict__ fault = nullptr) {
    __shared__ double part[1024];
    const int64_t n = n_ptr ? (int64_t)*n_ptr : n_fixed;
    if (fault && threadIdx.x == 0 && n > n_bound)
        __hip_atomic_store(fault, (unsigned long long)n, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    const int64_t tiles = (n + BH_TILE - 1) / BH_TILE;
    const int64_t per = (tiles + 1023) / 1024;
    const int64_t beg = (int64_t)threadIdx.x * per, end = min(tiles, beg + per);
    double m = 0.0;
    for (int64_t t = beg; t < end; ++t) m = fmax(m, tile_max[t]);
    // exclusive running max over the 1024 partial maxima (wave scan + the 16 wave maxima; was one thread walking all of them)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const double incl = wave_incl_max(m, lane);
    if (lane == 63) part[wave] = incl;
    __syncthreads();
    double run = __shfl_up(incl, 1, 64);
    if (lane == 0) run = 0.0;
    run = fmax(run, carry_in);
    for (int w = 0; w < wave; ++w) run = fmax(run, part[w]);
    if (total_max && threadIdx.x == 1023) *total_max = fmax(run, m);
    for (int64_t t = beg; t < end; ++t) {
        const double c = tile_max[t];
        tile_max[t] = run;
        run = fmax(run, c);
    }
}

// q = inclusive running max of the BH values; written either scattered to row order (vals != null)
// or in sorted order (distributed path)
__global__ __launch_bounds__(BH_THREADS) void bh_apply(const unsigned long long* __restrict__ keys,
                                                       const unsigned int* __restrict__ vals,
                                                       const unsigned long long* __restrict__ n_ptr, int64_t n_fixed,
                                                       double n_tests, double rank0, const double* __restrict__ tile_carry,
                                                       const double* __restrict__ extra_carry, double* __restrict__ q_out,
                                                       double* __restrict__ dense = nullptr,
                                                       const unsigned long long* __restrict__ dense_flag = nullptr) {
    __shared__ double wtot[BH_THREADS / 64];
    const bool into_dense = dense && (!dense_flag || *dense_flag != 0ull);    // the values are compact indices: q goes to the dense array
    if (into_dense) q_out = dense;
    const int64_t n = n_ptr ? (int64_t)*n_ptr : n_fixed;
    const int64_t base = (int64_t)blockIdx.x * BH_TILE;
    if (base >= n) return;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // blocked arrangement: thread t owns BH_ITEMS consecutive sorted positions
    const int64_t first = base + (int64_t)threadIdx.x * BH_ITEMS;
    double v[BH_ITEMS];
    double run = 0.0;
#pragma unroll
    for (int r = 0; r < BH_ITEMS; ++r) {
        const int64_t i = first + r;
        const double b = (i < n) ? bh_value(keys[i], n_tests, rank0 + (double)(i + 1)) : 0.0;
        run = fmax(run, b);
        v[r] = run;
    }
    const double incl = wave_incl_max(run, lane);
    double excl = __shfl_up(incl, 1, 64);
    if (lane == 0) excl = 0.0;
    if (lane == 63) wtot[wave] = incl;
    __syncthreads();
    double carry = tile_carry[blockIdx.x];
    if (extra_carry) carry = fmax(carry, *extra_carry);          // sharded runs: the running max of the lower ranks' slices
    for (int w = 0; w < wave; ++w) carry = fmax(carry, wtot[w]);
    carry = fmax(carry, excl);
#pragma unroll
    for (int r = 0; r < BH_ITEMS; ++r) {
        const int64_t i = first + r;
        if (i < n) {
            const double qv = fmax(v[r], carry);
            if (vals && into_dense)
                q_out[vals[i]] = qv;                                     // the dense array: k3_fill_q reads it right behind this launch
            else if (vals)
                __builtin_nontemporal_store(qv, q_out + vals[i]);       // one 8-byte store into a line nobody else touches soon: no allocate
            else
                q_out[i] = qv;
        }
    }
}

// (Measured and dropped in round 4, profiles/history/r04_f_small_ab.txt: ONE resident launch for small survivor sets - 64 workgroups,
// a tile each, the six passes and the BH scan separated by device-wide barriers instead of 21 launches.  The kernels of a small
// sort already run back to back without gaps (profiles/history/r03_z_c2_timeline.txt); what a launch boundary costs is what a
// device-scope barrier costs too - the eight XCDs' L2s are made coherent by writing them back - and 19 such barriers took
// 0.58 ms where the 21 launches take 0.22 ms for the same 77 k keys.)

__global__ void k_iota_u32(unsigned int* __restrict__ v, int64_t n) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) v[i] = (unsigned int)i;
}

__global__ void k_scatter_q(const unsigned int* __restrict__ rows, const double* __restrict__ q_sorted,
                            const unsigned long long* __restrict__ n_ptr, double* __restrict__ q) {
    const int64_t n = (int64_t)*n_ptr;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) q[rows[i]] = q_sorted[i];
}

// plot_qvalues' 51 buckets (fithic.py:1235-1254): counts of floor(q/0.001), NaN -> bucket of 1.0
__global__ void k_fdr_hist(const double* __restrict__ q, int64_t n, unsigned long long* __restrict__ buckets) {
    __shared__ unsigned int h[64];
    if (threadIdx.x < 64) h[threadIdx.x] = 0;
    __syncthreads();
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        double v = q[i];
        if (v != v) v = 1.0;
        const double b = floor(v / 0.001);
        if (b < 51.0) atomicAdd(&h[(int)b], 1u);
    }
    __syncthreads();
    if (threadIdx.x < 51 && h[threadIdx.x]) atomicAdd(&buckets[threadIdx.x], (unsigned long long)h[threadIdx.x]);
}


// ---- host side ----------------------------------------------------------------------------------------------------------
// LSD radix sort of (u64 key, u32 payload) pairs over the low `passes`*11 key bits; n lives in *counter (device)
// Chunks (= workgroups) of a sort of about n keys: a count matrix of RADIX x blocks is scanned in every pass, so a small sort
// must not pay for 1024 of them (6 passes over ~10^6 keys: 0.26 ms with 1024 blocks, a third of that with 64).  n_hint < 0:
// the size is only known on the device.
int sort_blocks_for(int64_t n_hint) {
    if (n_hint < 0) return SORT_BLOCKS;
    const int64_t want = (n_hint + 4 * SORT_TILE - 1) / (4 * SORT_TILE);          // >= four tiles per workgroup
    return (int)std::max<int64_t>(64, std::min<int64_t>(SORT_BLOCKS, (want + 63) / 64 * 64));
}

// LSD radix passes over the low `key_bits` bits of the keys, `bits` bits per pass (8..11), starting in buffer pair `src`;
// *result_buf = the pair that holds the sorted keys and payloads
static int radix_passes(fhx_ctx* ctx, unsigned long long* keys[2], unsigned int* vals[2], const unsigned long long* counter, int nblk,
                        int key_bits, int bits, int src, int* result_buf) {
    const int passes = (key_bits + bits - 1) / bits;
    ctx->k2_prezeroed = false;                        // the count matrix is K2's heavy-class histogram too (fhx_pass_stats zeroes it ahead)
    for (int pass = 0; pass < passes; ++pass) {
        const int shift = pass * bits;
#define FHX_RS_PASS(B)                                                                                                               \
    do {                                                                                                                             \
        hipLaunchKernelGGL(HIP_KERNEL_NAME(rs_count<B>), dim3(nblk), dim3(SORT_THREADS), 0, ctx->stream, keys[src], counter, shift,    \
                           ctx->d_block_hist);                                                                                       \
        hipLaunchKernelGGL(rs_scan, dim3(1 << B), dim3((nblk + 63) / 64 * 64), 0, ctx->stream, ctx->d_block_hist, ctx->d_digit_total, \
                           nblk);                                                                                                    \
        hipLaunchKernelGGL(HIP_KERNEL_NAME(rs_scatter<512, 4, B>), dim3(nblk), dim3(512), 0, ctx->stream, keys[src], vals[src],        \
                           keys[1 - src], vals[1 - src], counter, shift, (const unsigned int*)ctx->d_block_hist,                      \
                           (const unsigned int*)ctx->d_digit_total);                                                                 \
    } while (0)
        if (bits == 8) FHX_RS_PASS(8);
        else if (bits == 9) FHX_RS_PASS(9);
        else if (bits == 10) FHX_RS_PASS(10);
        else FHX_RS_PASS(11);
#undef FHX_RS_PASS
        src = 1 - src;
    }
    FHX_HIP(hipGetLastError());
    *result_buf = src;
    return FHX_OK;
}

int radix_sort_pairs(fhx_ctx* ctx, unsigned long long* keys[2], unsigned int* vals[2], const unsigned long long* counter,
                     int passes, int* result_buf, int64_t n_hint) {
    return radix_passes(ctx, keys, vals, counter, sort_blocks_for(n_hint), passes * RADIX_BITS, RADIX_BITS, 0, result_buf);
}

// ---- one-sweep sort of the top bits + repair of the rest (fhx_onesweep.inc) ---------------------------------------------------
// ctrl: os_ctrl_words(n, 8) 32-bit words of device scratch.  key_hi: 62 for the engine's own p (bdtrc returns values in [0, 1]:
// bits 62 and 63 of the pattern are clear), 64 for a caller's array (fhx_bh_array takes whatever is >= 0: 2.0, +inf).
struct OsPlan {
    int passes, lo;
};
static size_t os_repair_offset(int64_t n) { return (os_ctrl_words(n, OS_MAX_PASSES) + 3) / 4 * 4; }      // words; 16-byte aligned
static size_t os_scratch_bytes(int64_t n) {          // control block + descriptors of eight passes + the repair's lists
    return (os_repair_offset(n) + os_repair_words(n)) * sizeof(unsigned int);
}
static OsPlan os_plan(int64_t n, int key_hi) {
    const char* e = std::getenv("FHX_OS_PASSES");    // measurements and tests (8 = all bits, nothing to repair)
    const int fv = e ? std::atoi(e) : 0;
    const int forced = (fv >= 1 && fv <= OS_MAX_PASSES) ? fv : 0;
    int passes = forced ? forced : (n <= 250000000ll ? 5 : 6);
    passes = std::min(passes, (key_hi + OS_BITS - 1) / OS_BITS);
    return OsPlan{passes, std::max(0, key_hi - passes * OS_BITS)};
}

static int os_run_passes(fhx_ctx* ctx, unsigned long long* keys[2], unsigned int* vals[2], const unsigned long long* counter, int64_t n,
                         OsPlan plan, unsigned int* ctrl, int* src_io, bool n_into_slot = false) {
    const int tiles = (int)((n + OS_TILE - 1) / OS_TILE);
    FHX_HIP(hipMemsetAsync(ctrl, 0, os_ctrl_words(n, plan.passes) * sizeof(unsigned int), ctx->stream));
    if (n_into_slot) {                                     // a segment: its key count lives in the control block
        unsigned long long* slot = reinterpret_cast<unsigned long long*>(ctrl + OSC_SEG_N);
        hipLaunchKernelGGL(os_set_n, dim3(1), dim3(1), 0, ctx->stream, slot, (unsigned long long)n);
        counter = slot;
    }
    const int hgrid = std::max(1, std::min(tiles / 2, 1024));
    switch (plan.passes) {
#define FHX_OS_HIST(P)                                                                                                             \
    case P:                                                                                                                        \
        hipLaunchKernelGGL(HIP_KERNEL_NAME(os_hist<P>), dim3(hgrid), dim3(OS_HIST_THREADS), 0, ctx->stream, (const unsigned long long*)keys[*src_io], \
                           counter, plan.lo, ctrl + OSC_HIST);                                                                     \
        break
        FHX_OS_HIST(1); FHX_OS_HIST(2); FHX_OS_HIST(3); FHX_OS_HIST(4); FHX_OS_HIST(5); FHX_OS_HIST(6); FHX_OS_HIST(7); FHX_OS_HIST(8);
#undef FHX_OS_HIST
    }
    int src = *src_io;
    const char* we = std::getenv("FHX_OS_WPE");            // waves per SIMD the scatter is compiled for (measurements): 4 or 6
    const int wpe = we ? std::atoi(we) : 4;
    const char* le = std::getenv("FHX_OS_LB");             // descriptors per look-back step: 8 (default), 16, 32
    const int lb = le ? std::atoi(le) : 8;
    const bool persist = std::getenv("FHX_OS_PERSIST") != nullptr;      // resident workgroups taking tiles in a loop (measured slower)
    for (int p = 0; p < plan.passes; ++p) {
#define FHX_OS_SCATTER(W, P, L)                                                                                                       \
    hipLaunchKernelGGL(HIP_KERNEL_NAME(os_scatter<W, P, L>), dim3(P ? std::min(tiles, 256 * (W / 2)) : tiles), dim3(OS_THREADS), 0, ctx->stream, (const unsigned long long*)keys[src], \
                       (const unsigned int*)vals[src], keys[1 - src], vals[1 - src], counter, plan.lo + p * OS_BITS,                \
                       (const unsigned int*)(ctrl + OSC_HIST + p * OS_RADIX), ctrl + OSC_DESC + (size_t)p * tiles * OS_RADIX,        \
                       ctrl + OSC_TICKET + p, ctrl + OSC_COPY_PASSES)
        if (wpe == 6) FHX_OS_SCATTER(6, false, 8);
        else if (persist) FHX_OS_SCATTER(4, true, 8);
        else if (lb == 8) FHX_OS_SCATTER(4, false, 8);
        else if (lb == 32) FHX_OS_SCATTER(4, false, 32);
        else FHX_OS_SCATTER(4, false, 16);
#undef FHX_OS_SCATTER
        src = 1 - src;
    }
    FHX_HIP(hipGetLastError());
    *src_io = src;
    return FHX_OK;
}

// One-sweep passes + repair, in two halves: onesweep_launch enqueues the passes, the repair and a copy of the repair's verdict
// into pinned memory; onesweep_finish waits for that copy alone (an event, not the stream) and - rarely - sorts what the repair
// could not reach.  Between the two a caller may enqueue whatever only READS the sorted keys (the BH scan): it runs while the
// host looks at the verdict, instead of the GPU idling through a host round trip; *moved tells it to run that work again.
struct OsPending {
    bool active = false;
    OsPlan plan{0, 0};
    int src = 0, key_hi = 62;
    int64_t n = 0;
    unsigned long long* keys[2] = {nullptr, nullptr};
    unsigned int* vals[2] = {nullptr, nullptr};
    const unsigned long long* counter = nullptr;
    unsigned int* ctrl = nullptr;
};

static int ensure_k3_host(fhx_ctx* ctx) {
    if (!ctx->h_k3) FHX_HIP(hipHostMalloc((void**)&ctx->h_k3, (20 + RP_SEG_CAP * 4 + 4) * sizeof(unsigned int), hipHostMallocDefault));
    if (!ctx->ev_k3) FHX_HIP(hipEventCreateWithFlags(&ctx->ev_k3, hipEventDisableTiming));
    return FHX_OK;
}

static int onesweep_launch(fhx_ctx* ctx, unsigned long long* keys[2], unsigned int* vals[2], const unsigned long long* counter, int64_t n,
                           int key_hi, unsigned int* ctrl, OsPending* pend) {
    int rc = ensure_k3_host(ctx);
    if (rc != FHX_OK) return rc;
    key_hi = std::min(64, key_hi + 1);                     // os_spread moves every exponent up by 52: p <= 1 reaches bit 62
    pend->plan = os_plan(n, key_hi);
    pend->src = 0;
    pend->key_hi = key_hi;
    pend->n = n;
    pend->counter = counter;
    pend->ctrl = ctrl;
    for (int b = 0; b < 2; ++b) {
        pend->keys[b] = keys[b];
        pend->vals[b] = vals[b];
    }
    rc = os_run_passes(ctx, keys, vals, counter, n, pend->plan, ctrl, &pend->src);
    if (rc != FHX_OK) return rc;
    for (int k = 0; k < 8; ++k) ctx->h_k3[8 + k] = 0u;
    if (pend->plan.lo > 0) {
        unsigned int* rp = ctrl + os_repair_offset(n);
        const unsigned int region_cap = (unsigned int)os_region_cap(n);
        uint2* d_runs = reinterpret_cast<uint2*>(rp);
        OsFindCounts* d_counts = reinterpret_cast<OsFindCounts*>(rp + (size_t)region_cap * RP_FIND_WGS * 2);
        unsigned int* d_longs = reinterpret_cast<unsigned int*>(d_counts + RP_FIND_WGS);
        long long* d_segs = reinterpret_cast<long long*>(d_longs + RP_LONG_CAP);
        unsigned long long* k = keys[pend->src];
        unsigned int* v = vals[pend->src];
        static_assert(RP_FIND_WGS == RP_FIND_GRID, "one count slot and one region per workgroup");
        hipLaunchKernelGGL(os_find_runs, dim3(RP_FIND_GRID), dim3(256), 0, ctx->stream, (const unsigned long long*)k, counter, pend->plan.lo, ctrl,
                           d_runs, region_cap, d_counts, d_longs);
        hipLaunchKernelGGL(os_tally, dim3(1), dim3(RP_FIND_GRID), 0, ctx->stream, (const OsFindCounts*)d_counts, ctrl);
        hipLaunchKernelGGL(os_fix_runs, dim3(RP_FIND_GRID), dim3(256), 0, ctx->stream, k, v, (const uint2*)d_runs, region_cap,
                           (const OsFindCounts*)d_counts);
        hipLaunchKernelGGL(os_long_runs, dim3(64), dim3(512), 0, ctx->stream, (const unsigned long long*)k, counter, pend->plan.lo, ctrl,
                           (const unsigned int*)d_longs, d_segs);
        hipLaunchKernelGGL(os_fix_long, dim3(64), dim3(512), 0, ctx->stream, k, v, (const unsigned int*)ctrl, (const long long*)d_segs);
        FHX_HIP(hipGetLastError());
        FHX_HIP(hipMemcpyAsync(ctx->h_k3 + 8, ctrl + OSC_FALLBACK, 8 * sizeof(unsigned int), hipMemcpyDeviceToHost, ctx->stream));
        // (n may be the histogram's upper bound, see compact_pvalues: the exact count comes back with the verdict)
        FHX_HIP(hipMemcpyAsync(ctx->h_k3 + 16, counter, sizeof(unsigned long long), hipMemcpyDeviceToHost, ctx->stream));
        FHX_HIP(hipMemcpyAsync(ctx->h_k3 + 20, d_segs, RP_SEG_CAP * 2 * sizeof(long long), hipMemcpyDeviceToHost, ctx->stream));
        FHX_HIP(hipEventRecord(ctx->ev_k3, ctx->stream));
    }
    pend->active = true;
    return FHX_OK;
}

static int onesweep_finish(fhx_ctx* ctx, OsPending* pend, bool* moved) {
    *moved = false;
    if (!pend->active) return FHX_OK;
    pend->active = false;
    const OsPlan plan = pend->plan;
    int64_t n = pend->n;                                   // what the launch was sized for: the exact count or an upper bound
    const int key_hi = pend->key_hi;
    unsigned int* ctrl = pend->ctrl;
    int src = pend->src;
    unsigned int st[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    int64_t seg_count = 0, seg_keys = 0;
    bool everything = false;
    int rc = FHX_OK;
    if (plan.lo > 0) {
        FHX_HIP(hipEventSynchronize(ctx->ev_k3));
        for (int k = 0; k < 8; ++k) st[k] = ctx->h_k3[8 + k];
        const int64_t n_exact = (int64_t)*reinterpret_cast<volatile unsigned long long*>(ctx->h_k3 + 16);
        if (n_exact > n) return fail(ctx, FHX_ERR_HIP, "internal: more keys were compacted than the histogram counted");
        n = n_exact;
        if (std::getenv("FHX_OS_FORCE_FALLBACK")) st[0] |= 4u;      // tests
        const unsigned int n_segs = st[OSC_SEGMENTS - OSC_FALLBACK];
        if (st[0]) everything = true;                      // a list overflowed, a run's ends were out of reach, or forced
        else if (n_segs) {
            struct OsSegment {
                int64_t s, e;
            };
            std::vector<OsSegment> segs;
            const long long* hs = reinterpret_cast<const long long*>(ctx->h_k3 + 20);
            for (unsigned int k = 0; k < n_segs && k < (unsigned int)RP_SEG_CAP; ++k) {
                const OsSegment g{hs[2 * k], hs[2 * k + 1]};
                if (g.s < 0 || g.e > n || g.e <= g.s) return fail(ctx, FHX_ERR_HIP, "internal: a long run's bounds are off");
                seg_keys += g.e - g.s;
                ++seg_count;
                if (g.e - g.s > RP_WG_RUN) segs.push_back(g);      // (the shorter ones are sorted already: os_fix_long)
            }
            if (seg_keys > n / 2) everything = true;       // no longer an exception: sort all bits in one go
            for (size_t i = 0; i < segs.size() && !everything; ++i) {
                const OsSegment& g = segs[i];
                if (g.e - g.s < 2) continue;
                unsigned long long* kk[2] = {pend->keys[src] + g.s, pend->keys[1 - src] + g.s};
                unsigned int* vv[2] = {pend->vals[src] + g.s, pend->vals[1 - src] + g.s};
                const int bits = plan.lo;
                const int64_t m = g.e - g.s;
                int where = 0;
                if (m <= KS_MAX_KEYS) {                    // tiles sorted in LDS + merge by rank (whole keys: the run's top bits are equal anyway)
                    unsigned long long* slot = reinterpret_cast<unsigned long long*>(ctrl + OSC_SEG_N);
                    hipLaunchKernelGGL(os_set_n, dim3(1), dim3(1), 0, ctx->stream, slot, (unsigned long long)m);
                    const int tiles = (int)((m + KS_TILE - 1) / KS_TILE);
                    hipLaunchKernelGGL(ks_tile_sort, dim3(tiles), dim3(KS_THREADS), 0, ctx->stream, (const unsigned long long*)kk[0],
                                       (const unsigned int*)vv[0], (const unsigned long long*)slot, kk[1], vv[1]);
                    if (tiles > 1)
                        hipLaunchKernelGGL(ks_merge_tiles, dim3(grid_for(m, 256)), dim3(256), 0, ctx->stream, (const unsigned long long*)kk[1],
                                           (const unsigned int*)vv[1], (const unsigned long long*)slot, kk[0], vv[0]);
                    FHX_HIP(hipGetLastError());
                    where = tiles > 1 ? 0 : 1;
                } else {
                    rc = os_run_passes(ctx, kk, vv, nullptr, m, OsPlan{(bits + OS_BITS - 1) / OS_BITS, 0}, ctrl, &where, true);
                    if (rc != FHX_OK) return rc;
                }
                if (where == 1) {                          // an odd number of passes: back into the array's own buffer
                    FHX_HIP(hipMemcpyAsync(kk[0], kk[1], (size_t)m * sizeof(unsigned long long), hipMemcpyDeviceToDevice, ctx->stream));
                    FHX_HIP(hipMemcpyAsync(vv[0], vv[1], (size_t)m * sizeof(unsigned int), hipMemcpyDeviceToDevice, ctx->stream));
                }
                *moved = true;
            }
        }
    }
    ctx->sort_stats[0] = plan.passes;
    ctx->sort_stats[1] = plan.lo;
    ctx->sort_stats[2] = (int64_t)st[0] | (everything ? 8 : 0);      // bit 3: every bit of every key was sorted after all
    ctx->sort_stats[3] = st[OSC_INVERSIONS - OSC_FALLBACK];
    ctx->sort_stats[4] = st[OSC_SMALL_RUNS - OSC_FALLBACK];
    ctx->sort_stats[5] = st[OSC_LONG_RUNS - OSC_FALLBACK];
    if (std::getenv("FHX_OS_DEBUG"))
        std::fprintf(stderr, "[onesweep] %lld keys, %d passes from bit %d: look-back steps of digit 0 per tile and pass %.2f, inversions %u\n", (long long)n,
                     plan.passes, plan.lo, (double)st[OSC_LB_STEPS - OSC_FALLBACK] / std::max<double>(1.0, (double)plan.passes * ((n + OS_TILE - 1) / OS_TILE)),
                     st[OSC_INVERSIONS - OSC_FALLBACK]);
    ctx->sort_stats[6] = seg_count;
    ctx->sort_stats[7] = seg_keys;
    if (everything) {
        // whatever order the keys are in now is as good a start as any (eight passes: the result is in the same buffer pair)
        rc = os_run_passes(ctx, pend->keys, pend->vals, pend->counter, pend->n, OsPlan{(key_hi + OS_BITS - 1) / OS_BITS, 0}, ctrl, &src);
        if (rc != FHX_OK) return rc;
        *moved = true;
        if (src != pend->src) return fail(ctx, FHX_ERR_HIP, "internal: the full sort left the keys in the other buffer pair");
    }
    return FHX_OK;
}

// launches for the other translation units (the sharded schedule, the heavy class's bucket sort, the FDR counts)
void launch_rs_scan(fhx_ctx* ctx, int nblk) {
    hipLaunchKernelGGL(rs_scan, dim3(RADIX), dim3((nblk + 63) / 64 * 64), 0, ctx->stream, ctx->d_block_hist, ctx->d_digit_total, nblk);
}
void launch_k3_cutoff(fhx_ctx* ctx, double n_tests, unsigned long long* d_cutoff) {
    hipLaunchKernelGGL(k3_cutoff, dim3(1), dim3(1024), 0, ctx->stream, (const unsigned long long*)ctx->d_top_hist, n_tests, d_cutoff,
                       (unsigned long long*)nullptr);
}
void launch_bh_tile_max(fhx_ctx* ctx, int tiles, const unsigned long long* keys, const unsigned long long* n_ptr, int64_t n_fixed,
                        double n_tests, double rank0, double* tile_max) {
    hipLaunchKernelGGL(bh_tile_max, dim3(tiles), dim3(BH_THREADS), 0, ctx->stream, keys, n_ptr, n_fixed, n_tests, rank0, tile_max);
}
void launch_bh_scan_tiles(fhx_ctx* ctx, double* tile_max, const unsigned long long* n_ptr, int64_t n_fixed, double carry_in,
                          double* total_max) {
    hipLaunchKernelGGL(bh_scan_tiles, dim3(1), dim3(1024), 0, ctx->stream, tile_max, n_ptr, n_fixed, carry_in, total_max);
}
void launch_bh_apply(fhx_ctx* ctx, int tiles, const unsigned long long* keys, const unsigned int* vals, const unsigned long long* n_ptr,
                     int64_t n_fixed, double n_tests, double rank0, const double* tile_carry, const double* extra_carry, double* q_out) {
    hipLaunchKernelGGL(bh_apply, dim3(tiles), dim3(BH_THREADS), 0, ctx->stream, keys, vals, n_ptr, n_fixed, n_tests, rank0, tile_carry,
                       extra_carry, q_out);
}
void launch_scatter_q(fhx_ctx* ctx, int64_t n_rows, const unsigned int* rows, const double* q_sorted, const unsigned long long* n_ptr,
                      double* q) {
    hipLaunchKernelGGL(k_scatter_q, dim3(grid_for(n_rows, 256)), dim3(256), 0, ctx->stream, rows, q_sorted, n_ptr, q);
}
void launch_fdr_hist(fhx_ctx* ctx, const double* q, int64_t n, unsigned long long* buckets) {
    hipLaunchKernelGGL(k_fdr_hist, dim3(grid_for(n, 256)), dim3(256), 0, ctx->stream, q, n, buckets);
}

}  // namespace fhx

// ---- C ABI -----------------------------------------------------------------------------------------------------------------
// compact p < 1 and LSD-radix-sort (key, row); returns the index (0/1) of the buffer pair holding the result
// cutoff key from a device-local histogram of the p-values (single-GPU path; sharded runs all-reduce the histogram)
// d_top_hist <- key histogram of the context's p: what K2 gathered while storing them, else one more read of p
int fhx::fill_top_hist(fhx_ctx* ctx) {
    if (ctx->k2_hist_valid) {
        FHX_HIP(hipMemcpyAsync(ctx->d_top_hist, ctx->d_k2_hist, TOP_BINS * sizeof(unsigned long long), hipMemcpyDeviceToDevice, ctx->stream));
        return FHX_OK;
    }
    FHX_HIP(hipMemsetAsync(ctx->d_top_hist, 0, TOP_BINS * sizeof(unsigned long long), ctx->stream));
    hipLaunchKernelGGL(k3_top_hist, dim3(grid_for((ctx->n_rows + 1) / 2, 512, 256 * 4)), dim3(512), 0, ctx->stream, ctx->d_p,
                       ctx->n_rows, ctx->d_top_hist);
    FHX_HIP(hipGetLastError());
    return FHX_OK;
}

// (k2_done: an event the caller has just recorded on the stream - the host sleeps on it before it spins for k3_cutoff's ticket;
// nullptr: one is recorded here)
static int auto_cutoff(fhx_ctx* ctx, const double* d_p, int64_t n, double n_total_tests, unsigned long long* d_cutoff,
                       unsigned long long dense_min = ~0ull, unsigned long long* counter_to_zero = nullptr, hipEvent_t k2_done = nullptr) {
    const unsigned long long* hist = ctx->d_top_hist;
    if (d_p == ctx->d_p && ctx->k2_hist_valid) {
        hist = ctx->d_k2_hist;                          // what K2 counted while it stored p: read where it lies (no copy into d_top_hist)
    } else if (d_p == ctx->d_p) {
        const int rc = fill_top_hist(ctx);
        if (rc != FHX_OK) return rc;
    } else {
        FHX_HIP(hipMemsetAsync(ctx->d_top_hist, 0, TOP_BINS * sizeof(unsigned long long), ctx->stream));
        hipLaunchKernelGGL(k3_top_hist, dim3(grid_for((n + 1) / 2, 512, 256 * 4)), dim3(512), 0, ctx->stream, d_p, n, ctx->d_top_hist);
    }
    // the survivors' number by the histogram (d_misc[8]) goes to pinned memory right behind the cutoff: compact_pvalues waits for
    // that copy alone, while the compaction it has already enqueued runs
    {
        int rc = ensure_k3_host(ctx);
        if (rc == FHX_OK) rc = ensure_flags(ctx);
        if (rc != FHX_OK) return rc;
    }
    // the stream up to here (K2) has an event the host can sleep on; what follows it - this one workgroup - is waited for by ticket
    if (!k2_done) {
        FHX_HIP(hipEventRecord(ctx->ev_k3, ctx->stream));
        k2_done = ctx->ev_k3;
    }
    ctx->k3_wait_ev = k2_done;
    CutoffToHost host;
    host.words = ctx->h_flags + FLAG_K3;
    host.ticket = ++ctx->ticket;
    host.done = ctx->d_done + 1;
    host.zero_me = counter_to_zero;
    hipLaunchKernelGGL(k3_cutoff, dim3(1), dim3(1024), 0, ctx->stream, hist, n_total_tests, d_cutoff, ctx->d_misc + 8, dense_min,
                       ctx->d_misc + MISC_K3_DENSE, host);
    FHX_HIP(hipGetLastError());
    ctx->k3_ticket = host.ticket;
    ctx->k3_kept_by_hist = true;
    ctx->k3_counter_zeroed = counter_to_zero != nullptr;
    return FHX_OK;
}

// rows below the cutoff -> keys[0] / vals[0] (their number in *counter and, read back, in *n_kept); every other row gets its q here
// (dq: k3_compact<true> - no q here, compact indices as values, masks for k3_fill_q)
static int compact_pvalues(fhx_ctx* ctx, const double* d_p, int64_t n, unsigned long long* keys[2], unsigned int* vals[2], double* d_q,
                           unsigned long long* counter, const unsigned long long* d_cutoff, int64_t* n_kept_out,
                           const DenseQ* dq = nullptr) {
    if (!ctx->k3_counter_zeroed) FHX_HIP(hipMemsetAsync(counter, 0, sizeof(unsigned long long), ctx->stream));
    ctx->k3_counter_zeroed = false;                   // (auto_cutoff had k3_cutoff zero it)
    // one workgroup per tile, not a resident grid walking the column: 0.507 -> 0.451 ms on C3 (profiles/history/r03_x_k3_grid.txt); the
    // plain copy kernel of profiles/hbm_rate.hip shows the same (4.9 TB/s with 2048 grid-striding workgroups, 5.6 with one per
    // chunk).  FHX_K3_GRID caps the grid for measurements.
    static const int k3_cap = std::getenv("FHX_K3_GRID") ? std::atoi(std::getenv("FHX_K3_GRID")) : (1 << 30);
    if (dq)
        hipLaunchKernelGGL(k3_compact<true>, dim3(grid_for(n, CP_TILE, k3_cap)), dim3(CP_THREADS), 0, ctx->stream, d_p, n,
                           keys[0], vals[0], d_q, counter, d_cutoff, *dq, false, 1);
    if (!dq || dq->flag) {                           // (with a flag the device picks one of the two; the other returns at once)
        DenseQ off;
        if (dq) off.flag = dq->flag;
        const bool ones = ctx->q_prefilled && d_q == ctx->d_q && d_p == ctx->d_p;
        // tiles per workgroup: as many as leave every CU its two workgroups several times over (a shard keeps one tile per workgroup)
        static const int per_env = std::getenv("FHX_K3_TILES_PER_WG") ? std::atoi(std::getenv("FHX_K3_TILES_PER_WG")) : 0;
        const int64_t n_tiles = (n + CP_TILE - 1) / CP_TILE;
        const int per = per_env > 0 ? per_env : (int)std::max<int64_t>(1, std::min<int64_t>(4, n_tiles / 2048));
        hipLaunchKernelGGL(k3_compact<false>, dim3(grid_for((n_tiles + per - 1) / per, 1, k3_cap)), dim3(CP_THREADS), 0, ctx->stream, d_p, n,
                           keys[0], vals[0], d_q, counter, d_cutoff, off, ones, per);
    }
    if (d_q == ctx->d_q) ctx->q_prefilled = false;      // from here on the column holds this pass's q
    // how many keys survived decides the shape of the sort.  When the cutoff came from this GPU's own histogram (auto_cutoff) the
    // number is already on its way - the histogram's bins below the cutoff bin hold exactly the rows kept here - and the host goes
    // on to enqueue the sort while the compaction runs; otherwise (sharded runs: the histogram is the all-reduced one) the counter
    // is read back behind the compaction.
    unsigned long long n_kept = 0;
    ctx->k3_n_is_bound = false;
    if (ctx->k3_kept_by_hist) {
        // (an UPPER BOUND when the histogram is K2's: it counts a wave's values in the bin of the smallest, fhx_k2.hip FusedHist -
        // every launch below takes its true count from the device counter and sizes its grid for the bound)
        ctx->k3_kept_by_hist = false;
        FHX_HIP(hipEventSynchronize(ctx->k3_wait_ev));     // K2 is through (the host sleeps: milliseconds) ...
        FHX_HIP(wait_ticket(ctx, FLAG_K3, ctx->k3_ticket));   // ... and k3_cutoff's one workgroup (microseconds: spin)
        const unsigned long long by_hist = ctx->h_flags[FLAG_K3 + 1];
        n_kept = std::min<unsigned long long>(by_hist, (unsigned long long)n);
        ctx->k3_n_is_bound = true;
    } else {
        FHX_HIP(hipMemcpyAsync(&n_kept, counter, sizeof(n_kept), hipMemcpyDeviceToHost, ctx->stream));
        FHX_HIP(hipStreamSynchronize(ctx->stream));
    }
    *n_kept_out = (int64_t)n_kept;
    return FHX_OK;
}

// sort of the n_kept compacted keys; the result is in buffer pair *sorted_buf.  key_hi: the bits that can be set (62 / 64, see
// onesweep_sort); ctrl: device scratch of os_ctrl_words(n_kept, 8) words, or nullptr (then the three-launch passes run)
static int sort_kept(fhx_ctx* ctx, unsigned long long* keys[2], unsigned int* vals[2], const unsigned long long* counter, int64_t n_kept,
                     int* sorted_buf, int key_hi, unsigned int* ctrl, OsPending* defer = nullptr) {
    const char* small_env = std::getenv("FHX_K3_SMALL");          // "0": the radix passes whatever the size (A/B runs, tests)
    const bool small_off = small_env && std::atoi(small_env) == 0;
    for (int k = 0; k < 8; ++k) ctx->sort_stats[k] = 0;
    if (n_kept <= KS_MAX_KEYS && !small_off) {                     // tile sort in LDS + merge by rank: two launches (see ks_tile_sort)
        const int tiles = (int)std::max<int64_t>(1, (n_kept + KS_TILE - 1) / KS_TILE);
        if (n_kept > 0) {
            hipLaunchKernelGGL(ks_tile_sort, dim3(tiles), dim3(KS_THREADS), 0, ctx->stream, (const unsigned long long*)keys[0],
                               (const unsigned int*)vals[0], counter, keys[1], vals[1]);
            if (tiles > 1)
                hipLaunchKernelGGL(ks_merge_tiles, dim3(grid_for(n_kept, 256)), dim3(256), 0, ctx->stream, (const unsigned long long*)keys[1],
                                   (const unsigned int*)vals[1], counter, keys[0], vals[0]);
        }
        FHX_HIP(hipGetLastError());
        *sorted_buf = (n_kept > 0 && tiles == 1) ? 1 : 0;
        return FHX_OK;
    }
    const char* se = std::getenv("FHX_K3_SORT");                  // "legacy": round 4's count / scan / scatter passes (A/B runs)
    const bool legacy = se && std::strcmp(se, "legacy") == 0;
    if (ctrl && !legacy && n_kept <= OS_MAX_KEYS) {
        OsPending mine;
        OsPending* pend = defer ? defer : &mine;
        int rc = onesweep_launch(ctx, keys, vals, counter, n_kept, key_hi, ctrl, pend);
        if (rc != FHX_OK) return rc;
        *sorted_buf = pend->src;                          // (the finish never changes the buffer pair)
        if (defer) return FHX_OK;                          // the caller finishes, after enqueueing what only reads the keys
        bool moved = false;
        return onesweep_finish(ctx, pend, &moved);
    }
    // FHX_RS_BITS: digit width of these passes (measurements)
    const char* be = std::getenv("FHX_RS_BITS");
    const int bits = (be && std::atoi(be) >= 8 && std::atoi(be) <= 11) ? std::atoi(be) : SORT_BITS_LARGE;
    return radix_passes(ctx, keys, vals, counter, sort_blocks_for(n_kept), key_hi, bits, 0, sorted_buf);
}

static int sort_pvalues(fhx_ctx* ctx, const double* d_p, int64_t n, unsigned long long* keys[2], unsigned int* vals[2],
                        double* d_q, unsigned long long* counter, const unsigned long long* d_cutoff, int* sorted_buf,
                        int64_t* n_sorted_out, int key_hi, unsigned int* ctrl) {
    int64_t n_kept = 0;
    const int rc = compact_pvalues(ctx, d_p, n, keys, vals, d_q, counter, d_cutoff, &n_kept);
    if (rc != FHX_OK) return rc;
    if (n_sorted_out) *n_sorted_out = n_kept;
    return sort_kept(ctx, keys, vals, counter, n_kept, sorted_buf, key_hi, ctrl);
}

// The engine's own sort scratch: the workspace behind the K3 view (alloc_row_arrays: K3 uses 24 of its >= 48 bytes per row; the
// descriptors of eight passes take 2).
// nullptr - the sort then takes round 4's three-launch passes, which need no scratch - if the workspace ever stops covering it (a
// changed queue layout, bucket padding or tile size): never a write past the block.
static unsigned int* engine_sort_ctrl(fhx_ctx* ctx) {
    const size_t cap = std::max<size_t>(4, ((size_t)ctx->n_rows + 3) / 4 * 4);
    if (cap * 24 + os_scratch_bytes(ctx->n_rows) > ctx->work_bytes) return nullptr;
    return reinterpret_cast<unsigned int*>(ctx->d_work + cap * 24);
}
// The dense-q arrays of the engine's own pass, behind the sort's scratch: 8 B per row at most (every row a survivor) + 2 bits per
// row + 8 B per 1024 rows - 24 + 2.7 + 8.3 of the workspace's >= 48 B per row.  false: they do not fit (then q is scattered as before).
static bool engine_dense_q(fhx_ctx* ctx, DenseQ* dq) {
    const size_t cap = std::max<size_t>(4, ((size_t)ctx->n_rows + 3) / 4 * 4);
    const size_t chunks = (cap + 64 * CP_ITEMS - 1) / (64 * CP_ITEMS) + CP_WAVES;
    size_t at = (cap * 24 + os_scratch_bytes(ctx->n_rows) + 255) / 256 * 256;
    const size_t dense_bytes = cap * 8, mask_bytes = chunks * (CP_ITEMS / 2) * 4 * 8, slot_bytes = chunks * 8;
    if (at + dense_bytes + mask_bytes + slot_bytes > ctx->work_bytes) return false;
    dq->dense = reinterpret_cast<double*>(ctx->d_work + at);
    at += dense_bytes;
    dq->mask = reinterpret_cast<unsigned long long*>(ctx->d_work + at);
    at += mask_bytes;
    dq->wave_slot = reinterpret_cast<unsigned long long*>(ctx->d_work + at);
    return true;
}

// n_keys: the number of sorted keys when the host knows it (the grids then cover the keys, not the rows), else an upper bound
// (dq: the values are compact indices - q goes to the dense array and k3_fill_q writes the column: p_rows / n_rows name it)
static int bh_from_sorted(fhx_ctx* ctx, const unsigned long long* keys, const unsigned int* vals, int64_t n_keys,
                          const unsigned long long* counter, double n_total_tests, double* tile_max, double* d_q,
                          const DenseQ* dq = nullptr, const double* p_rows = nullptr, int64_t n_rows = 0) {
    const int tiles = (int)std::max<int64_t>(1, (n_keys + BH_TILE - 1) / BH_TILE);
    unsigned long long* fault = nullptr;
    if (ctx->k3_n_is_bound) {                           // n_keys came from the histogram: the device checks it against its counter
        const int rc = ensure_flags(ctx);
        if (rc != FHX_OK) return rc;
        fault = const_cast<unsigned long long*>(ctx->h_flags + FLAG_FAULT);
    }
    hipLaunchKernelGGL(bh_tile_max, dim3(tiles), dim3(BH_THREADS), 0, ctx->stream, keys, counter, (int64_t)0, n_total_tests,
                       0.0, tile_max);
    hipLaunchKernelGGL(bh_scan_tiles, dim3(1), dim3(1024), 0, ctx->stream, tile_max, counter, (int64_t)0, 0.0,
                       (double*)nullptr, n_keys, fault);
    hipLaunchKernelGGL(bh_apply, dim3(tiles), dim3(BH_THREADS), 0, ctx->stream, keys, vals, counter, (int64_t)0,
                       n_total_tests, 0.0, tile_max, (const double*)nullptr, d_q, dq ? dq->dense : (double*)nullptr,
                       dq ? dq->flag : (const unsigned long long*)nullptr);
    if (dq) {
        static const int fill_cap = std::getenv("FHX_K3_GRID") ? std::atoi(std::getenv("FHX_K3_GRID")) : (1 << 30);
        hipLaunchKernelGGL(k3_fill_q, dim3(grid_for(n_rows, CP_TILE, fill_cap)), dim3(CP_THREADS), 0, ctx->stream, p_rows, n_rows, *dq,
                           d_q);
    }
    FHX_HIP(hipGetLastError());
    return FHX_OK;
}

// compaction, sort and BH of one p column -> q in row order
static int rank_and_adjust(fhx_ctx* ctx, const double* d_p, int64_t n, unsigned long long* keys[2], unsigned int* vals[2], double* d_q,
                           unsigned long long* counter, const unsigned long long* d_cutoff, double n_total_tests, double* tile_max,
                           int* sorted_buf, int64_t* n_sorted_out, int key_hi, unsigned int* ctrl, const DenseQ* dq = nullptr) {
    int64_t n_kept = 0;
    int rc = compact_pvalues(ctx, d_p, n, keys, vals, d_q, counter, d_cutoff, &n_kept, dq);
    if (rc != FHX_OK) return rc;
    if (n_sorted_out) *n_sorted_out = n_kept;
    OsPending pend;
    rc = sort_kept(ctx, keys, vals, counter, n_kept, sorted_buf, key_hi, ctrl, &pend);
    if (rc != FHX_OK) return rc;
    rc = bh_from_sorted(ctx, keys[*sorted_buf], vals[*sorted_buf], n_kept, counter, n_total_tests, tile_max, d_q, dq, d_p, n);
    if (rc != FHX_OK) return rc;
    bool moved = false;
    rc = onesweep_finish(ctx, &pend, &moved);          // the BH scan above runs while the host reads the repair's verdict
    if (rc != FHX_OK) return rc;
    if (moved) rc = bh_from_sorted(ctx, keys[*sorted_buf], vals[*sorted_buf], n_kept, counter, n_total_tests, tile_max, d_q, dq, d_p, n);
    return rc;
}

int fhx::ensure_sort_scratch(fhx_ctx* ctx) {
    if (!ctx->d_block_hist) FHX_HIP(hipMalloc(&ctx->d_block_hist, (size_t)RADIX * SORT_BLOCKS * sizeof(unsigned int)));
    if (!ctx->d_digit_total) FHX_HIP(hipMalloc(&ctx->d_digit_total, RADIX * sizeof(unsigned int)));
    if (!ctx->d_misc) FHX_HIP(hipMalloc(&ctx->d_misc, 192 * sizeof(unsigned long long)));
    if (!ctx->d_top_hist) FHX_HIP(hipMalloc(&ctx->d_top_hist, TOP_BINS * sizeof(unsigned long long)));
    return FHX_OK;
}

int fhx_bh_top_hist(fhx_ctx* ctx, int64_t* hist_out, int64_t capacity) {
    if (!ctx || !hist_out || capacity < TOP_BINS) return FHX_ERR_ARG;
    if (ctx->device < 0) return fail(ctx, FHX_ERR_NO_DEVICE, "host-only context");
    if (!ctx->have_p) return fail(ctx, FHX_ERR_ARG, "fhx_pvalues must run first");
    FHX_HIP(hipSetDevice(ctx->device));
    {
        const int rc = fill_top_hist(ctx);
        if (rc != FHX_OK) return rc;
    }
    FHX_HIP(hipMemcpyAsync(hist_out, ctx->d_top_hist, TOP_BINS * sizeof(int64_t), hipMemcpyDeviceToHost, ctx->stream));
    FHX_HIP(hipStreamSynchronize(ctx->stream));
    return FHX_OK;
}

// Device-resident variant for sharded runs: the histogram stays in HBM (fhx_device_ptr(ctx, 4)), the caller all-reduces it
// in place (RCCL) and fhx_bh_set_cutoff_device derives the cutoff from it - no host round trip of the 64 KiB table.
int fhx_bh_top_hist_device(fhx_ctx* ctx) {
    if (!ctx) return FHX_ERR_ARG;
    if (ctx->device < 0) return fail(ctx, FHX_ERR_NO_DEVICE, "host-only context");
    if (!ctx->have_p) return fail(ctx, FHX_ERR_ARG, "fhx_pvalues must run first");
    FHX_HIP(hipSetDevice(ctx->device));
    {
        const int rc = fill_top_hist(ctx);
        if (rc != FHX_OK) return rc;
    }
    FHX_HIP(hipStreamSynchronize(ctx->stream));           // the caller's collective runs on another stream
    return FHX_OK;
}

int fhx_bh_set_cutoff_device(fhx_ctx* ctx, double n_total_tests) {
    if (!ctx || !(n_total_tests > 0)) return FHX_ERR_ARG;
    if (ctx->device < 0) return fail(ctx, FHX_ERR_NO_DEVICE, "host-only context");
    FHX_HIP(hipSetDevice(ctx->device));
    hipLaunchKernelGGL(k3_cutoff, dim3(1), dim3(1024), 0, ctx->stream, (const unsigned long long*)ctx->d_top_hist, n_total_tests,
                       ctx->d_misc + 6, (unsigned long long*)nullptr);
    FHX_HIP(hipGetLastError());
    return FHX_OK;
}

int fhx_bh_set_cutoff(fhx_ctx* ctx, const int64_t* global_hist, int64_t n_bins, double n_total_tests) {
    if (!ctx || !global_hist || n_bins != TOP_BINS || !(n_total_tests > 0)) return FHX_ERR_ARG;
    if (ctx->device < 0) return fail(ctx, FHX_ERR_NO_DEVICE, "host-only context");
    FHX_HIP(hipSetDevice(ctx->device));
    unsigned long long cutoff = KEY_KEEP_ALL, cum = 0;
    for (int b = 0; b < TOP_BINS; ++b) {
        cum += (unsigned long long)global_hist[b];
        if (global_hist[b] > 0 && bin_saturates(b, cum, n_total_tests)) {
            cutoff = (unsigned long long)b << TOP_SHIFT;
            break;
        }
    }
    FHX_HIP(hipMemcpyAsync(ctx->d_misc + 6, &cutoff, sizeof(cutoff), hipMemcpyHostToDevice, ctx->stream));
    FHX_HIP(hipStreamSynchronize(ctx->stream));
    return FHX_OK;
}

int fhx_bh_sort_stats(fhx_ctx* ctx, int64_t* out8) {
    if (!ctx || !out8) return FHX_ERR_ARG;
    for (int k = 0; k < 8; ++k) out8[k] = ctx->sort_stats[k];
    return FHX_OK;
}

int fhx_bh_local_sort(fhx_ctx* ctx) {
    if (!ctx) return FHX_ERR_ARG;
    ctx->k3_kept_by_hist = false;            // (sharded runs: the cutoff comes from the all-reduced histogram, the count from the counter)
    ctx->k3_counter_zeroed = false;          // (... and nothing has zeroed the counter ahead of the compaction)
    if (ctx->device < 0) return fail(ctx, FHX_ERR_NO_DEVICE, "host-only context");
    if (!ctx->have_p) return fail(ctx, FHX_ERR_ARG, "fhx_pvalues must run first");
    FHX_HIP(hipSetDevice(ctx->device));
    int64_t kept = 0;
    const int rc = sort_pvalues(ctx, ctx->d_p, ctx->n_rows, ctx->d_keys, ctx->d_vals, ctx->d_q, ctx->d_misc, ctx->d_misc + 6,
                                &ctx->sorted_buf, &kept, 62, engine_sort_ctrl(ctx));
    if (rc != FHX_OK) return rc;
    ctx->n_sorted = kept;
    return FHX_OK;
}

int fhx_bh_array(fhx_ctx* ctx, const double* p, int64_t n, double n_total_tests, double* q) {
    if (!ctx || !p || !q || n < 0) return FHX_ERR_ARG;
    if (ctx->device < 0) return fail(ctx, FHX_ERR_NO_DEVICE, "host-only context");
    if (n == 0) return FHX_OK;
    if (n >= (1ll << 32)) return fail(ctx, FHX_ERR_UNSUPPORTED, "more than 2^32 p-values");
    if (!std::isfinite(n_total_tests)) return fail(ctx, FHX_ERR_ARG, "number of tests must be finite");
    for (int64_t i = 0; i < n; ++i)
        if (p[i] < 0.0) return fail(ctx, FHX_ERR_ARG, "negative p-value at index " + std::to_string(i) + " (p-values must be >= 0 or NaN)");
    FHX_HIP(hipSetDevice(ctx->device));
    int rc = ensure_sort_scratch(ctx);
    if (rc != FHX_OK) return rc;
    double *d_p = nullptr, *d_q = nullptr, *tile_max = nullptr;
    unsigned long long* keys[2] = {nullptr, nullptr};
    unsigned int* vals[2] = {nullptr, nullptr};
    DeviceScratch tmp;
    const size_t cap = (size_t)n;
    FHX_HIP(tmp.get(&d_p, cap * sizeof(double)));
    FHX_HIP(tmp.get(&d_q, cap * sizeof(double)));
    FHX_HIP(tmp.get(&tile_max, (cap / BH_TILE + 2) * sizeof(double)));
    for (int b = 0; b < 2; ++b) {
        FHX_HIP(tmp.get(&keys[b], cap * sizeof(unsigned long long)));
        FHX_HIP(tmp.get(&vals[b], cap * sizeof(unsigned int)));
    }
    // a caller's p may be anything >= 0 (2.0, +inf: the reference ranks them like any other value): all 64 key bits count
    unsigned int* ctrl = nullptr;
    if (n > KS_MAX_KEYS && n <= OS_MAX_KEYS) FHX_HIP(tmp.get(&ctrl, os_scratch_bytes(n)));
    FHX_HIP(hipMemcpyAsync(d_p, p, cap * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    int buf = 0;
    unsigned long long* counter = ctx->d_misc + 2;
    unsigned long long* cutoff = ctx->d_misc + 7;
    rc = auto_cutoff(ctx, d_p, n, n_total_tests, cutoff, ~0ull, counter);
    if (rc == FHX_OK) rc = rank_and_adjust(ctx, d_p, n, keys, vals, d_q, counter, cutoff, n_total_tests, tile_max, &buf, nullptr, 64, ctrl);
    if (rc == FHX_OK) {
        FHX_HIP(hipMemcpyAsync(q, d_q, cap * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
        FHX_HIP(hipStreamSynchronize(ctx->stream));
    }
    return rc;
}

int fhx_bh(fhx_ctx* ctx, double n_total_tests) {
    if (!ctx) return FHX_ERR_ARG;
    if (ctx->device < 0) return fail(ctx, FHX_ERR_NO_DEVICE, "host-only context");
    if (!ctx->have_p) return fail(ctx, FHX_ERR_ARG, "fhx_pvalues must run first");
    if (!std::isfinite(n_total_tests)) return fail(ctx, FHX_ERR_ARG, "number of tests must be finite");
    FHX_HIP(hipSetDevice(ctx->device));
    before_rerecord(ctx, 2);
    FHX_HIP(hipEventRecord(ctx->ev[4], ctx->stream));
    // q of the survivors through a dense array + one fill of the q column (k3_compact<true>) when at least K3_DENSE_PERCENT of the
    // rows survive the cutoff - the device decides, from the count k3_cutoff takes off the histogram, while the host goes on
    // enqueueing: below that the two extra passes over the rows cost more than the scattered stores into the q column
    // (profiles/r06_k3_dense_q.txt).  FHX_K3_DENSE: 1 = always, 0 = never, otherwise the percentage (measurements, tests).
    static const int dense_env = std::getenv("FHX_K3_DENSE") ? std::atoi(std::getenv("FHX_K3_DENSE")) : K3_DENSE_PERCENT;
    DenseQ dq;
    // (a row set whose last pass left under a tenth of its rows below the cutoff will not leave 35 % now: the dense variant's two
    // launches - which would return at once - are not even enqueued then; the decision itself stays the device's whenever they are)
    const bool far_below = dense_env != 1 && ctx->k3_last_rows == ctx->n_rows && ctx->k3_last_kept >= 0 &&
                           ctx->k3_last_kept * 10 < ctx->n_rows;
    const bool dense = dense_env != 0 && !far_below && engine_dense_q(ctx, &dq);
    unsigned long long dense_min = ~0ull;
    if (dense) {
        dense_min = dense_env == 1 ? 0ull : (unsigned long long)(((long double)ctx->n_rows * dense_env + 99) / 100);
        dq.flag = ctx->d_misc + MISC_K3_DENSE;
    }
    int rc = auto_cutoff(ctx, ctx->d_p, ctx->n_rows, n_total_tests, ctx->d_misc + 6, dense_min, ctx->d_misc, ctx->ev[4]);
    if (rc != FHX_OK) return rc;
    int64_t kept = 0;
    rc = rank_and_adjust(ctx, ctx->d_p, ctx->n_rows, ctx->d_keys, ctx->d_vals, ctx->d_q, ctx->d_misc, ctx->d_misc + 6, n_total_tests,
                         ctx->d_tile_max, &ctx->sorted_buf, &kept, 62, engine_sort_ctrl(ctx), dense ? &dq : nullptr);
    if (rc != FHX_OK) return rc;
    ctx->n_sorted = ctx->k3_n_is_bound ? -2 : kept;      // -2: fhx_n_sorted reads the device counter when somebody asks
    ctx->k3_last_kept = kept;                            // (the histogram's bound or the exact number: either serves the guess above)
    ctx->k3_last_rows = ctx->n_rows;
    FHX_HIP(hipEventRecord(ctx->ev[5], ctx->stream));
    ctx->ev_valid[2] = true;
    ctx->ev_folded[2] = false;
    ctx->have_q = true;
    return FHX_OK;
}

int fhx_bh_apply_sorted(fhx_ctx* ctx, const void* d_sorted_keys, int64_t n, int64_t global_rank0, double carry_in,
                        double n_total_tests, void* d_q_sorted, double* block_max_out) {
    if (!ctx || n < 0) return FHX_ERR_ARG;
    if (ctx->device < 0) return fail(ctx, FHX_ERR_NO_DEVICE, "host-only context");
    FHX_HIP(hipSetDevice(ctx->device));
    if (n == 0) {
        if (block_max_out) *block_max_out = carry_in;
        return FHX_OK;
    }
    const int tiles = (int)((n + BH_TILE - 1) / BH_TILE);
    double* tile_max = nullptr;
    FHX_HIP(hipMalloc(&tile_max, ((size_t)tiles + 1) * sizeof(double)));
    const unsigned long long* keys = (const unsigned long long*)d_sorted_keys;
    hipLaunchKernelGGL(bh_tile_max, dim3(tiles), dim3(BH_THREADS), 0, ctx->stream, keys, (const unsigned long long*)nullptr, n,
                       n_total_tests, (double)global_rank0, tile_max);
    hipLaunchKernelGGL(bh_scan_tiles, dim3(1), dim3(1024), 0, ctx->stream, tile_max, (const unsigned long long*)nullptr, n,
                       carry_in, tile_max + tiles);
    if (d_q_sorted)
        hipLaunchKernelGGL(bh_apply, dim3(tiles), dim3(BH_THREADS), 0, ctx->stream, keys, (const unsigned int*)nullptr,
                           (const unsigned long long*)nullptr, n, n_total_tests, (double)global_rank0, tile_max,
                           (const double*)nullptr, (double*)d_q_sorted);
    double total = 0.0;
    FHX_HIP(hipMemcpyAsync(&total, tile_max + tiles, sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    FHX_HIP(hipStreamSynchronize(ctx->stream));
    dev_free(tile_max);
    if (block_max_out) *block_max_out = total;
    return FHX_OK;
}

int fhx_sort_u64(fhx_ctx* ctx, const void* d_keys_in, int64_t n, void* d_keys_out, void* d_perm_out) {
    if (!ctx || n < 0 || (n > 0 && (!d_keys_in || !d_keys_out || !d_perm_out))) return FHX_ERR_ARG;
    if (ctx->device < 0) return fail(ctx, FHX_ERR_NO_DEVICE, "host-only context");
    if (n == 0) return FHX_OK;
    if (n >= (1ll << 32)) return fail(ctx, FHX_ERR_UNSUPPORTED, "more than 2^32 keys");
    FHX_HIP(hipSetDevice(ctx->device));
    int rc = ensure_sort_scratch(ctx);
    if (rc != FHX_OK) return rc;
    unsigned long long* keys[2] = {nullptr, (unsigned long long*)d_keys_out};
    unsigned int* vals[2] = {nullptr, (unsigned int*)d_perm_out};
    DeviceScratch tmp;                               // freed on every return path
    FHX_HIP(tmp.get(&keys[0], (size_t)n * sizeof(unsigned long long)));
    FHX_HIP(tmp.get(&vals[0], (size_t)n * sizeof(unsigned int)));
    unsigned long long* counter = ctx->d_misc + 3;
    const unsigned long long n_host = (unsigned long long)n;
    FHX_HIP(hipMemcpyAsync(counter, &n_host, sizeof(n_host), hipMemcpyHostToDevice, ctx->stream));
    // an even number of ping-pong passes: start in the caller's output pair so that the result lands there
    FHX_HIP(hipMemcpyAsync(keys[1], d_keys_in, (size_t)n * sizeof(unsigned long long), hipMemcpyDeviceToDevice, ctx->stream));
    hipLaunchKernelGGL(k_iota_u32, dim3(grid_for(n, 256)), dim3(256), 0, ctx->stream, vals[1], n);
    int where = 1;
    rc = radix_passes(ctx, keys, vals, counter, sort_blocks_for(n), SORT_PASSES * RADIX_BITS, RADIX_BITS, 1, &where);   // six passes: back in pair [1]
    if (rc != FHX_OK) return rc;
    FHX_HIP(hipStreamSynchronize(ctx->stream));
    return FHX_OK;                                   // even number of swaps: the result is in pair [1]
}

int fhx_bh_scatter(fhx_ctx* ctx, const void* d_q_sorted_local) {
    if (!ctx) return FHX_ERR_ARG;
    if (ctx->device < 0) return fail(ctx, FHX_ERR_NO_DEVICE, "host-only context");
    if (ctx->n_sorted == -1) return fail(ctx, FHX_ERR_ARG, "fhx_bh_local_sort must run first");
    FHX_HIP(hipSetDevice(ctx->device));
    if (d_q_sorted_local)                       // NULL is legal when this rank holds no p < 1 at all
        hipLaunchKernelGGL(k_scatter_q, dim3(grid_for(ctx->n_rows, 256)), dim3(256), 0, ctx->stream, ctx->d_vals[ctx->sorted_buf],
                       (const double*)d_q_sorted_local, ctx->d_misc, ctx->d_q);
    FHX_HIP(hipGetLastError());
    FHX_HIP(hipStreamSynchronize(ctx->stream));
    ctx->have_q = true;
    return FHX_OK;
}

