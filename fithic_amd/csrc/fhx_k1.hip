// fhx_k1.hip - K0 ingest (raw rows -> 12-byte SoA rows), K1 classify + sums + distance histogram (fithic.read_Interactions,
// fithic/fithic.py:389-454), the -r 0 / off-grid slotting by sort + run detection, the per-row device arrays
// (one of the device translation units of libfithic_mi355x.so; shared declarations: fhx_ctx.hpp)
#include "fhx_ctx.hpp"

namespace fhx {

// ===================================================================================================
// K0: ingest.  Slot of a locus = chr_base[chr] + mid / res; a chromosome's loci must share mid % res
// (that is what "fixed-size" data looks like: createFitHiCFragments-fixedsize.py writes mid = i*res + res/2).
// ===================================================================================================
__global__ void k0_extent(const int32_t* __restrict__ chr, const int32_t* __restrict__ mid, int64_t n, int res,
                          int n_chr, int32_t* __restrict__ max_idx, int32_t* __restrict__ min_off,
                          int32_t* __restrict__ max_off, int32_t* __restrict__ bad) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const int c = chr[i];
        const int m = mid[i];
        if (c < 0 || c >= n_chr || m < 0) {
            atomicOr(bad, 1);
            continue;
        }
        const int idx = m / res, off = m - idx * res;
        // rows of one wave nearly always share the chromosome (contact files are sorted): with all 64 lanes active and one
        // chromosome, reduce over the wave and let lane 0 talk to memory (per-lane atomics on the running maximum of a
        // sorted column cost 12 ms per 1.5e8 rows)
        const bool full = __ballot(1) == ~0ull;
        const int c0 = __shfl(c, 0, 64);
        if (full && __ballot(c != c0) == 0ull) {
            int hi = idx, lo_off = off, hi_off = off;
            for (int s = 32; s >= 1; s >>= 1) {
                hi = max(hi, __shfl_xor(hi, s, 64));
                lo_off = min(lo_off, __shfl_xor(lo_off, s, 64));
                hi_off = max(hi_off, __shfl_xor(hi_off, s, 64));
            }
            if ((threadIdx.x & 63) == 0) {
                if (hi > max_idx[c]) atomicMax(&max_idx[c], hi);
                if (lo_off < min_off[c]) atomicMin(&min_off[c], lo_off);
                if (hi_off > max_off[c]) atomicMax(&max_off[c], hi_off);
            }
        } else {
            if (idx > max_idx[c]) atomicMax(&max_idx[c], idx);
            if (off < min_off[c]) atomicMin(&min_off[c], off);
            if (off > max_off[c]) atomicMax(&max_off[c], off);
        }
    }
}

__global__ void k0_slots(const int32_t* __restrict__ chr1, const int32_t* __restrict__ mid1,
                         const int32_t* __restrict__ chr2, const int32_t* __restrict__ mid2,
                         const int32_t* __restrict__ cnt, int64_t n, int res, const ChrGrid* __restrict__ grid,
                         int32_t* __restrict__ loc1, int32_t* __restrict__ loc2, int32_t* __restrict__ count) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const int c1 = chr1[i], c2 = chr2[i];
        const int s1 = grid[c1].base + mid1[i] / res;
        const int s2 = grid[c2].base + mid2[i] / res;
        loc1[i] = s1;
        loc2[i] = (c1 == c2) ? s2 : ~s2;          // sign bit carries "inter-chromosomal"
        count[i] = cnt[i];
    }
}

// ===================================================================================================
// K1: classification + sums + distance histogram
// ===================================================================================================
constexpr int K1_THREADS = 512;
constexpr int K1_LDS_BINS = 6144;      // 6144 * (8 + 4) B = 72 KiB -> two workgroups per CU

// WIDE = false: the window holds 6144 bins as (u64 sum, u32 rows): every Hi-C run with a distance cap (C3: 397 bins).
// WIDE = true (more distance values than that, e.g. no -U: 49 847 at 5 kb): one 1024-thread workgroup per CU owns 144 KB =
// 24 576 bins as (u32 sum, 15-bit row count + guard bit); a sum that wraps adds 2^32 to the bin in HBM (exactly one thread
// sees the wrap), a row count that reaches 2^15 sets the guard bit, which cannot carry into the neighbouring half-word, and
// the thread that set it moves 2^15 to HBM and clears it.  With 12 B/bin only a quarter of the bins of that run were in LDS
// and the rest took two device-scope atomics per row: K1 24.8 ms per 1.06e9 rows (profiles/history/r02_p_c3w_bench.json).
constexpr int K1_WIDE_BINS = 24576;
constexpr int K1_WIDE_THREADS = 1024;

template <int THREADS, bool WIDE>
__global__ __launch_bounds__(THREADS) void k1_classify_hist(
    const int32_t* __restrict__ loc1, const int32_t* __restrict__ loc2, const int32_t* __restrict__ count,
    const uint8_t* __restrict__ skip, int64_t skip_limit, const long long* __restrict__ grow, int64_t n, int lo_idx,
    int hi_idx,
    unsigned long long* __restrict__ hist_sumcc, unsigned long long* __restrict__ hist_npairs,
    K1Sums* __restrict__ sums) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int BINS = WIDE ? K1_WIDE_BINS : K1_LDS_BINS;
    unsigned long long* lds_cc = reinterpret_cast<unsigned long long*>(smem);
    unsigned int* lds_np = reinterpret_cast<unsigned int*>(smem + sizeof(unsigned long long) * K1_LDS_BINS);
    unsigned int* wide_sum = reinterpret_cast<unsigned int*>(smem);                          // WIDE: BINS x u32
    unsigned int* wide_cnt = reinterpret_cast<unsigned int*>(smem) + K1_WIDE_BINS;           // WIDE: BINS / 2 x (2 x 16 bit)
    if (WIDE) {
        for (int i = threadIdx.x; i < BINS + BINS / 2; i += THREADS) wide_sum[i] = 0u;
    } else {
        for (int i = threadIdx.x; i < BINS; i += THREADS) {
            lds_cc[i] = 0ull;
            lds_np[i] = 0u;
        }
    }
    __syncthreads();

    long long inter_count = 0, inter_sum = 0, intra_cnt = 0, intra_sum = 0, rng_cnt = 0, rng_sum = 0, skipped = 0;
    int max_count = 0;

    auto one = [&](int l1, int l2, int c, int sk) {
        max_count = max(max_count, c);
        if (sk) {
            ++skipped;
            return;
        }
        if (l2 < 0) {
            ++inter_count;
            inter_sum += c;
            return;
        }
        ++intra_cnt;
        intra_sum += c;
        const int d = abs(l1 - l2);
        if (d >= lo_idx && d <= hi_idx) {
            ++rng_cnt;
            rng_sum += c;
            const int b = d - lo_idx;
            if (WIDE && b < BINS && c >= 0) {
                const unsigned int old = atomicAdd(&wide_sum[b], (unsigned int)c);
                if (old + (unsigned int)c < old) atomicAdd(&hist_sumcc[d], 1ull << 32);
                const int sh = (b & 1) * 16;
                const unsigned int was = (atomicAdd(&wide_cnt[b >> 1], 1u << sh) >> sh) & 0xFFFFu;
                if (was == 0x7FFFu) {                       // this add set the guard bit: move 2^15 rows to HBM
                    atomicSub(&wide_cnt[b >> 1], 0x8000u << sh);
                    atomicAdd(&hist_npairs[d], 32768ull);
                }
            } else if (!WIDE && b < BINS) {
                atomicAdd(&lds_cc[b], (unsigned long long)(long long)c);
                atomicAdd(&lds_np[b], 1u);
            } else {
                atomicAdd(&hist_sumcc[d], (unsigned long long)(long long)c);
                atomicAdd(&hist_npairs[d], 1ull);
            }
        }
    };

    // 4 rows per lane per step: 16-byte coalesced loads of each of the three columns
    const int64_t n4 = n >> 2;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    const int4* a4 = reinterpret_cast<const int4*>(loc1);
    const int4* b4 = reinterpret_cast<const int4*>(loc2);
    const int4* c4 = reinterpret_cast<const int4*>(count);
    const uchar4* s4 = reinterpret_cast<const uchar4*>(skip);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
        const int4 a = a4[i], b = b4[i], c = c4[i];
        uchar4 s = make_uchar4(0, 0, 0, 0);
        // rows after the first duplicated outlier line are not skipped any more (fithic.py:408-412)
        if (skip && grow) {                       // shard: compare file positions, not local positions
            s = s4[i];
            const int64_t r = i << 2;
            if (grow[r] > skip_limit) s.x = 0;
            if (grow[r + 1] > skip_limit) s.y = 0;
            if (grow[r + 2] > skip_limit) s.z = 0;
            if (grow[r + 3] > skip_limit) s.w = 0;
        } else if (skip && (i << 2) <= skip_limit) {
            s = s4[i];
            const int64_t r = i << 2;
            if (r + 1 > skip_limit) s.y = 0;
            if (r + 2 > skip_limit) s.z = 0;
            if (r + 3 > skip_limit) s.w = 0;
        }
        one(a.x, b.x, c.x, s.x);
        one(a.y, b.y, c.y, s.y);
        one(a.z, b.z, c.z, s.z);
        one(a.w, b.w, c.w, s.w);
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
        const int64_t i = (n4 << 2) + threadIdx.x;
        one(loc1[i], loc2[i], count[i], (skip && (grow ? grow[i] : i) <= skip_limit) ? skip[i] : 0);
    }

    __syncthreads();
    // flush the LDS window; every workgroup starts at a different bin so that the 512 workgroups, which finish together,
    // do not queue up on the same L2 atomic address (same-address atomics retire at ~88 M/s, MI355X_MICROARCH.md)
    const int rot = (int)((blockIdx.x * 389u) % (unsigned)BINS);
    for (int k = threadIdx.x; k < BINS; k += THREADS) {
        int i = k + rot;
        if (i >= BINS) i -= BINS;
        if (WIDE) {
            const unsigned int sum = wide_sum[i], np = (wide_cnt[i >> 1] >> ((i & 1) * 16)) & 0xFFFFu;
            if (sum) atomicAdd(&hist_sumcc[lo_idx + i], (unsigned long long)sum);
            if (np) atomicAdd(&hist_npairs[lo_idx + i], (unsigned long long)np);
        } else {
            const unsigned int np = lds_np[i];
            if (np) {
                atomicAdd(&hist_sumcc[lo_idx + i], lds_cc[i]);
                atomicAdd(&hist_npairs[lo_idx + i], (unsigned long long)np);
            }
        }
    }
    // sums: wave reduce, combine the waves in LDS, then ONE atomic per field per workgroup (one per wave was 32 768
    // same-cache-line atomics at the end of the kernel: a ~0.35 ms tail on a 0.33 ms kernel)
    inter_count = wave_sum_i64(inter_count);
    inter_sum = wave_sum_i64(inter_sum);
    intra_cnt = wave_sum_i64(intra_cnt);
    intra_sum = wave_sum_i64(intra_sum);
    rng_cnt = wave_sum_i64(rng_cnt);
    rng_sum = wave_sum_i64(rng_sum);
    skipped = wave_sum_i64(skipped);
    max_count = wave_max_i32(max_count);
    __syncthreads();                                   // the histogram window is free now: reuse its first bytes
    long long* part = reinterpret_cast<long long*>(smem);
    constexpr int WAVES = THREADS / 64;
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) {
        part[w * 8 + 0] = inter_count;
        part[w * 8 + 1] = inter_sum;
        part[w * 8 + 2] = intra_cnt;
        part[w * 8 + 3] = intra_sum;
        part[w * 8 + 4] = rng_cnt;
        part[w * 8 + 5] = rng_sum;
        part[w * 8 + 6] = skipped;
        part[w * 8 + 7] = max_count;
    }
    __syncthreads();
    if (threadIdx.x < 8) {
        long long v = part[threadIdx.x];
        for (int k = 1; k < WAVES; ++k) v = threadIdx.x == 7 ? max(v, part[k * 8 + 7]) : v + part[k * 8 + threadIdx.x];
        unsigned long long* dst = reinterpret_cast<unsigned long long*>(sums);      // seven int64 fields, then max_count
        if (threadIdx.x < 7) {
            if (v) atomicAdd(dst + threadIdx.x, (unsigned long long)v);
        } else {
            atomicMax(&sums->max_count, (int)v);
        }
    }
}

// [7 sums | max_count | sumCC[a .. a + w) | rows[a .. a + w)] in one block: what the host fit needs of K1's output leaves the
// device in ONE small copy (the histograms are as long as the longest chromosome - 400 KB each at 5 kb - but only the distance
// window of the run can be non-zero: 397 entries of each on C3)
__global__ void k1_pack_window(const K1Sums* __restrict__ sums, const unsigned long long* __restrict__ hist_cc,
                               const unsigned long long* __restrict__ hist_np, int a, int w, long long* __restrict__ pack,
                               unsigned int* done, volatile unsigned long long* flag, unsigned long long ticket) {
    const int total = 8 + 2 * w;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        long long v;
        if (i < 7)
            v = reinterpret_cast<const long long*>(sums)[i];
        else if (i == 7)
            v = (long long)sums->max_count;
        else if (i < 8 + w)
            v = (long long)hist_cc[a + (i - 8)];
        else
            v = (long long)hist_np[a + (i - 8 - w)];
        pack[i] = v;
    }
    // `pack` is the host's pinned block when a flag is given: the ticket tells the spinning host that every part has arrived
    if (flag) publish_ticket(done, flag, ticket);
}

// K2's two histograms (the fused top-bits histogram of p, the count matrix of the heavy class) zeroed behind K1, while the host
// fits: two fill dispatches less between the fit and k2_classify
// ... and the q column filled with 1.0 - what nearly every row ends with (myStats.py:33-38: min(p N / rank, 1) of everything behind
// the exact cutoff): k3_compact then only reads p and stores the NaN rows' q, instead of storing 8 bytes for every row while K3
// is on the critical path; here the stores run while the GPU would idle behind K1
__global__ void k1_prezero(unsigned long long* __restrict__ a, int64_t na, uint4* __restrict__ b, int64_t nb16, double* __restrict__ q,
                           int64_t n_q) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nb16; i += stride) b[i] = make_uint4(0u, 0u, 0u, 0u);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < na; i += stride) a[i] = 0ull;
    double2* q2 = reinterpret_cast<double2*>(q);
    const int64_t n2 = n_q >> 1;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n2; i += stride) q2[i] = make_double2(1.0, 1.0);
    if ((n_q & 1) && blockIdx.x == 0 && threadIdx.x == 0) q[n_q - 1] = 1.0;
}

// ===================================================================================================
// non-fixed-size mode (-r 0): loci and distances are arbitrary integers, so the dense index arithmetic of the
// fixed-size path is replaced by sort + run detection (reusing the radix sort above)
// ===================================================================================================

// locus keys (chr << 32 | mid) of both ends of every row: element i = locus 1 of row i, element n + i = locus 2
__global__ void nf_locus_keys(const int32_t* __restrict__ c1, const int32_t* __restrict__ m1, const int32_t* __restrict__ c2,
                              const int32_t* __restrict__ m2, int64_t n, unsigned long long* __restrict__ keys,
                              unsigned int* __restrict__ vals, int* __restrict__ bad) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        if (c1[i] < 0 || c2[i] < 0 || m1[i] < 0 || m2[i] < 0) atomicOr(bad, 1);
        keys[i] = ((unsigned long long)(unsigned int)c1[i] << 32) | (unsigned int)m1[i];
        keys[n + i] = ((unsigned long long)(unsigned int)c2[i] << 32) | (unsigned int)m2[i];
        vals[i] = (unsigned int)i;
        vals[n + i] = (unsigned int)(n + i);
    }
}

// run heads of a sorted key array: per-tile head counts, then (after the scan of the tile counts) the run id of every element
__global__ __launch_bounds__(SEG_THREADS) void seg_count_heads(const unsigned long long* __restrict__ keys, int64_t n,
                                                               unsigned int* __restrict__ tile_heads) {
    __shared__ unsigned int cnt;
    if (threadIdx.x == 0) cnt = 0;
    __syncthreads();
    const int64_t base = (int64_t)blockIdx.x * SEG_TILE;
    unsigned int mine = 0;
    for (int r = 0; r < SEG_ITEMS; ++r) {
        const int64_t i = base + r * SEG_THREADS + threadIdx.x;
        if (i < n && (i == 0 || keys[i] != keys[i - 1])) ++mine;
    }
    mine = (unsigned int)wave_sum_i64((long long)mine);
    if ((threadIdx.x & 63) == 0 && mine) atomicAdd(&cnt, mine);
    __syncthreads();
    if (threadIdx.x == 0) tile_heads[blockIdx.x] = cnt;
}

__global__ __launch_bounds__(1024) void seg_scan_tiles(unsigned int* __restrict__ tile_heads, int64_t tiles,
                                                       unsigned long long* __restrict__ total) {
    __shared__ unsigned int part[1024];
    const int64_t per = (tiles + 1023) / 1024;
    const int64_t beg = (int64_t)threadIdx.x * per, end = min(tiles, beg + per);
    unsigned int sum = 0;
    for (int64_t t = beg; t < end; ++t) sum += tile_heads[t];
    part[threadIdx.x] = sum;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned int acc = 0;
        for (int i = 0; i < 1024; ++i) {
            const unsigned int c = part[i];
            part[i] = acc;
            acc += c;
        }
        *total = acc;
    }
    __syncthreads();
    unsigned int run = part[threadIdx.x];
    for (int64_t t = beg; t < end; ++t) {
        const unsigned int c = tile_heads[t];
        tile_heads[t] = run;
        run += c;
    }
}

// run id of every sorted element (0-based), blocked arrangement: thread t owns SEG_ITEMS consecutive elements
__global__ __launch_bounds__(SEG_THREADS) void seg_ids(const unsigned long long* __restrict__ keys, int64_t n,
                                                       const unsigned int* __restrict__ tile_base,
                                                       unsigned int* __restrict__ ids) {
    __shared__ unsigned int wtot[SEG_THREADS / 64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t first = (int64_t)blockIdx.x * SEG_TILE + (int64_t)threadIdx.x * SEG_ITEMS;
    unsigned int heads = 0;
    unsigned int flag[SEG_ITEMS];
#pragma unroll
    for (int r = 0; r < SEG_ITEMS; ++r) {
        const int64_t i = first + r;
        flag[r] = (i < n && (i == 0 || keys[i] != keys[i - 1])) ? 1u : 0u;
        heads += flag[r];
    }
    unsigned int incl = heads;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const unsigned int o = __shfl_up(incl, off, 64);
        if (lane >= off) incl += o;
    }
    if (lane == 63) wtot[wave] = incl;
    __syncthreads();
    unsigned int before = tile_base[blockIdx.x] + incl - heads;
    for (int w = 0; w < wave; ++w) before += wtot[w];
#pragma unroll
    for (int r = 0; r < SEG_ITEMS; ++r) {
        const int64_t i = first + r;
        before += flag[r];
        if (i < n) ids[i] = before - 1;                  // heads so far, including this element's own head
    }
}

// locus slots: element -> run id; the run's key goes to the slot table
__global__ void nf_assign_slots(const unsigned long long* __restrict__ keys, const unsigned int* __restrict__ vals,
                                const unsigned int* __restrict__ ids, int64_t n2, int32_t* __restrict__ loc,
                                unsigned long long* __restrict__ slot_key) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n2; i += stride) {
        loc[vals[i]] = (int32_t)ids[i];
        if (i == 0 || keys[i] != keys[i - 1]) slot_key[ids[i]] = keys[i];
    }
}

// rows: (loc1, loc2) with the inter flag in the sign of loc2, plus the slot tables
__global__ void nf_finish_rows(const int32_t* __restrict__ c1, const int32_t* __restrict__ c2, const int32_t* __restrict__ cnt,
                               const int32_t* __restrict__ loc, int64_t n, int32_t* __restrict__ loc1,
                               int32_t* __restrict__ loc2, int32_t* __restrict__ count) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        loc1[i] = loc[i];
        loc2[i] = (c1[i] == c2[i]) ? loc[n + i] : ~loc[n + i];
        count[i] = cnt[i];
    }
}

__global__ void nf_slot_tables(const unsigned long long* __restrict__ slot_key, int64_t n_slots, int32_t* __restrict__ slot_mid,
                               int16_t* __restrict__ slot_chr) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_slots; i += stride) {
        slot_mid[i] = (int32_t)(slot_key[i] & 0xFFFFFFFFull);
        slot_chr[i] = (int16_t)(slot_key[i] >> 32);
    }
}

// K1 for -r 0: the same classification and sums as k1_classify_hist; in-range rows emit (distance, count) for the sort
__global__ __launch_bounds__(SORT_THREADS) void nf_k1_classify(
    const int32_t* __restrict__ loc1, const int32_t* __restrict__ loc2, const int32_t* __restrict__ count,
    const uint8_t* __restrict__ skip, int64_t skip_limit, const long long* __restrict__ grow, int64_t n,
    const int32_t* __restrict__ slot_mid, long long dist_low, long long dist_up, unsigned long long* __restrict__ keys,
    unsigned int* __restrict__ vals, unsigned long long* __restrict__ counter, K1Sums* __restrict__ sums) {
    __shared__ unsigned int wave_cnt[SORT_WAVES];
    __shared__ unsigned long long block_base;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const unsigned long long lane_lt = (1ull << lane) - 1ull;
    long long inter_count = 0, inter_sum = 0, intra_cnt = 0, intra_sum = 0, rng_cnt = 0, rng_sum = 0, skipped = 0;
    int max_count = 0;
    const int64_t tiles = (n + SORT_TILE - 1) / SORT_TILE;
    for (int64_t t = blockIdx.x; t < tiles; t += gridDim.x) {
        const int64_t wave_base = t * SORT_TILE + (int64_t)wave * (64 * SORT_ITEMS);
        unsigned long long d_of[SORT_ITEMS];
        unsigned int c_of[SORT_ITEMS], before[SORT_ITEMS];
        unsigned long long keepmask = 0;
        unsigned int run = 0;
#pragma unroll
        for (int r = 0; r < SORT_ITEMS; ++r) {
            const int64_t i = wave_base + r * 64 + lane;
            bool keep = false;
            d_of[r] = 0;
            c_of[r] = 0;
            if (i < n) {
                const int l1 = loc1[i], l2 = loc2[i], c = count[i];
                max_count = max(max_count, c);
                const bool sk = skip && skip[i] && ((grow ? grow[i] : i) <= skip_limit);
                if (sk) {
                    ++skipped;
                } else if (l2 < 0) {
                    ++inter_count;
                    inter_sum += c;
                } else {
                    ++intra_cnt;
                    intra_sum += c;
                    const long long dist = llabs((long long)slot_mid[l1] - (long long)slot_mid[l2]);
                    if (dist >= dist_low && dist <= dist_up) {
                        ++rng_cnt;
                        rng_sum += c;
                        keep = true;
                        d_of[r] = (unsigned long long)dist;
                        c_of[r] = (unsigned int)c;
                    }
                }
            }
            const unsigned long long m = __ballot(keep);
            before[r] = run + __popcll(m & lane_lt);
            run += __popcll(m);
            if (keep) keepmask |= (1ull << r);
        }
        if (lane == 0) wave_cnt[wave] = run;
        __syncthreads();
        if (threadIdx.x == 0) {
            unsigned int tot = 0;
            for (int w = 0; w < SORT_WAVES; ++w) {
                const unsigned int c = wave_cnt[w];
                wave_cnt[w] = tot;
                tot += c;
            }
            block_base = tot ? atomicAdd(counter, (unsigned long long)tot) : 0ull;
        }
        __syncthreads();
        const unsigned long long base = block_base + wave_cnt[wave];
#pragma unroll
        for (int r = 0; r < SORT_ITEMS; ++r) {
            if ((keepmask >> r) & 1ull) {
                keys[base + before[r]] = d_of[r];
                vals[base + before[r]] = c_of[r];
            }
        }
        __syncthreads();
    }
    inter_count = wave_sum_i64(inter_count);
    inter_sum = wave_sum_i64(inter_sum);
    intra_cnt = wave_sum_i64(intra_cnt);
    intra_sum = wave_sum_i64(intra_sum);
    rng_cnt = wave_sum_i64(rng_cnt);
    rng_sum = wave_sum_i64(rng_sum);
    skipped = wave_sum_i64(skipped);
    max_count = wave_max_i32(max_count);
    if (lane == 0) {
        atomicAdd((unsigned long long*)&sums->inter_count, (unsigned long long)inter_count);
        atomicAdd((unsigned long long*)&sums->inter_sum, (unsigned long long)inter_sum);
        atomicAdd((unsigned long long*)&sums->intra_all_count, (unsigned long long)intra_cnt);
        atomicAdd((unsigned long long*)&sums->intra_all_sum, (unsigned long long)intra_sum);
        atomicAdd((unsigned long long*)&sums->in_range_count, (unsigned long long)rng_cnt);
        atomicAdd((unsigned long long*)&sums->in_range_sum, (unsigned long long)rng_sum);
        atomicAdd((unsigned long long*)&sums->n_skipped, (unsigned long long)skipped);
        atomicMax(&sums->max_count, max_count);
    }
}

// distinct distances: key, sum of counts and number of rows per run of the sorted (distance, count) array
__global__ void nf_accumulate_runs(const unsigned long long* __restrict__ keys, const unsigned int* __restrict__ vals,
                                   const unsigned int* __restrict__ ids, int64_t n, unsigned long long* __restrict__ out_key,
                                   unsigned long long* __restrict__ out_sum, unsigned long long* __restrict__ out_cnt) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const unsigned int s = ids[i];
        if (i == 0 || keys[i] != keys[i - 1]) out_key[s] = keys[i];
        atomicAdd(&out_sum[s], (unsigned long long)vals[i]);
        atomicAdd(&out_cnt[s], 1ull);
    }
}


// ---- host side ----------------------------------------------------------------------------------------------------------
// bias rows -> per-slot table (first occurrence wins, bounds applied: fithic.py:818-832)
int build_slot_tables_nonfixed(fhx_ctx* ctx) {
    // exact (chr, mid) match against the sorted distinct loci of the rows; first occurrence wins (fithic.py:829-832)
    std::vector<double> bias((size_t)std::max<int64_t>(ctx->n_slots, 1), ctx->have_bias ? -1.0 : 1.0);
    if (ctx->have_bias) {
        std::vector<uint8_t> seen(bias.size(), 0);
        const auto& keys = ctx->h_slot_keys;
        for (size_t i = 0; i < ctx->bias_val.size(); ++i) {
            const int32_t c = ctx->bias_chr[i], m = ctx->bias_mid[i];
            if (c < 0 || m < 0) continue;
            const unsigned long long k = ((unsigned long long)(unsigned int)c << 32) | (unsigned int)m;
            const auto it = std::lower_bound(keys.begin(), keys.end(), k);
            if (it == keys.end() || *it != k) continue;
            const size_t s = (size_t)(it - keys.begin());
            if (seen[s]) continue;
            seen[s] = 1;
            double b = ctx->bias_val[i];
            if (b < ctx->prm.bias_low || std::isnan(b))
                b = -1;
            else if (b > ctx->prm.bias_up)
                b = -1;
            bias[s] = b;
        }
    }
    dev_free(ctx->d_slot_bias);
    FHX_HIP(hipMalloc(&ctx->d_slot_bias, bias.size() * sizeof(double)));
    FHX_HIP(hipMemcpyAsync(ctx->d_slot_bias, bias.data(), bias.size() * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    FHX_HIP(hipStreamSynchronize(ctx->stream));
    ctx->tables_dirty = false;
    return FHX_OK;
}

int build_slot_tables(fhx_ctx* ctx) {
    if (ctx->nonfixed) return build_slot_tables_nonfixed(ctx);
    const int64_t res = ctx->prm.resolution;
    std::vector<double> bias((size_t)std::max<int64_t>(ctx->n_slots, 1), ctx->have_bias ? -1.0 : 1.0);
    std::vector<int16_t> slot_chr((size_t)std::max<int64_t>(ctx->n_slots, 1), 0);
    for (size_t c = 0; c < ctx->grid.size(); ++c)
        for (int32_t s = 0; s < ctx->grid[c].nslots; ++s) slot_chr[(size_t)ctx->grid[c].base + s] = (int16_t)c;
    if (ctx->have_bias) {
        std::vector<uint8_t> seen(bias.size(), 0);
        for (size_t i = 0; i < ctx->bias_val.size(); ++i) {
            const int32_t c = ctx->bias_chr[i], m = ctx->bias_mid[i];
            if (c < 0 || c >= (int32_t)ctx->grid.size() || m < 0) continue;
            const ChrGrid& g = ctx->grid[c];
            if (g.off < 0) continue;                                  // chromosome has no contact rows
            const int64_t idx = m / res;
            if (m - idx * res != g.off || idx >= g.nslots) continue;  // no row can match this exact midpoint
            const size_t s = (size_t)g.base + (size_t)idx;
            if (seen[s]) continue;
            seen[s] = 1;
            double b = ctx->bias_val[i];
            if (b < ctx->prm.bias_low || std::isnan(b))
                b = -1;
            else if (b > ctx->prm.bias_up)
                b = -1;
            bias[s] = b;
        }
    }
    dev_free(ctx->d_slot_bias);
    dev_free(ctx->d_slot_chr);
    FHX_HIP(hipMalloc(&ctx->d_slot_bias, bias.size() * sizeof(double)));
    FHX_HIP(hipMalloc(&ctx->d_slot_chr, slot_chr.size() * sizeof(int16_t)));
    FHX_HIP(hipMemcpyAsync(ctx->d_slot_bias, bias.data(), bias.size() * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    FHX_HIP(hipMemcpyAsync(ctx->d_slot_chr, slot_chr.data(), slot_chr.size() * sizeof(int16_t), hipMemcpyHostToDevice,
                           ctx->stream));
    FHX_HIP(hipStreamSynchronize(ctx->stream));
    ctx->tables_dirty = false;
    return FHX_OK;
}

// run ids of a sorted key array of n elements (n known on the host); returns the number of runs
int run_ids(fhx_ctx* ctx, const unsigned long long* keys, int64_t n, unsigned int* ids, unsigned int* tile_scratch,
            int64_t* n_runs) {
    *n_runs = 0;
    if (n == 0) return FHX_OK;
    const int tiles = (int)((n + SEG_TILE - 1) / SEG_TILE);
    unsigned long long* total = ctx->d_misc + 9;
    hipLaunchKernelGGL(seg_count_heads, dim3(tiles), dim3(SEG_THREADS), 0, ctx->stream, keys, n, tile_scratch);
    hipLaunchKernelGGL(seg_scan_tiles, dim3(1), dim3(1024), 0, ctx->stream, tile_scratch, (int64_t)tiles, total);
    hipLaunchKernelGGL(seg_ids, dim3(tiles), dim3(SEG_THREADS), 0, ctx->stream, keys, n, (const unsigned int*)tile_scratch, ids);
    FHX_HIP(hipGetLastError());
    unsigned long long t = 0;
    FHX_HIP(hipMemcpyAsync(&t, total, sizeof(t), hipMemcpyDeviceToHost, ctx->stream));
    FHX_HIP(hipStreamSynchronize(ctx->stream));
    *n_runs = (int64_t)t;
    return FHX_OK;
}

int alloc_row_arrays(fhx_ctx* ctx, int64_t n, int64_t n_dist);

// -r 0: loci are arbitrary (chr, mid) pairs.  Slot = rank of the locus among the sorted distinct loci of the rows.
int ingest_device_rows_nonfixed(fhx_ctx* ctx, const int32_t* c1, const int32_t* m1, const int32_t* c2, const int32_t* m2,
                                const int32_t* cnt, int64_t n) {
    if (!ctx->have_params) return fail(ctx, FHX_ERR_ARG, "fhx_set_params must be called before fhx_load_pairs");
    if (n < 0) return fail(ctx, FHX_ERR_ARG, "negative row count");
    if (2 * n >= (1ll << 32)) return fail(ctx, FHX_ERR_UNSUPPORTED, "more than 2^31 rows per GPU: shard the contacts");
    int rc = ensure_sort_scratch(ctx);
    if (rc != FHX_OK) return rc;
    const int64_t n2 = 2 * n;
    const size_t cap2 = std::max<size_t>(4, (size_t)n2);
    DeviceScratch tmp;
    unsigned long long* keys[2] = {nullptr, nullptr};
    unsigned int* vals[2] = {nullptr, nullptr};
    unsigned int *ids = nullptr, *tiles = nullptr;
    int32_t* loc = nullptr;
    unsigned long long* slot_key = nullptr;
    int* bad = nullptr;
    for (int b = 0; b < 2; ++b) {
        FHX_HIP(tmp.get(&keys[b], cap2 * sizeof(unsigned long long)));
        FHX_HIP(tmp.get(&vals[b], cap2 * sizeof(unsigned int)));
    }
    FHX_HIP(tmp.get(&ids, cap2 * sizeof(unsigned int)));
    FHX_HIP(tmp.get(&tiles, (cap2 / SEG_TILE + 2) * sizeof(unsigned int)));
    FHX_HIP(tmp.get(&loc, cap2 * sizeof(int32_t)));
    FHX_HIP(tmp.get(&slot_key, cap2 * sizeof(unsigned long long)));
    FHX_HIP(tmp.get(&bad, sizeof(int)));
    FHX_HIP(hipMemsetAsync(bad, 0, sizeof(int), ctx->stream));
    hipLaunchKernelGGL(nf_locus_keys, dim3(grid_for(n, 256)), dim3(256), 0, ctx->stream, c1, m1, c2, m2, n, keys[0], vals[0], bad);
    unsigned long long* counter = ctx->d_misc + 3;
    const unsigned long long n2u = (unsigned long long)n2;
    FHX_HIP(hipMemcpyAsync(counter, &n2u, sizeof(n2u), hipMemcpyHostToDevice, ctx->stream));
    int buf = 0;
    rc = radix_sort_pairs(ctx, keys, vals, counter, SORT_PASSES, &buf, n2);
    int64_t n_slots = 0;
    if (rc == FHX_OK) rc = run_ids(ctx, keys[buf], n2, ids, tiles, &n_slots);
    int h_bad = 0;
    if (rc == FHX_OK) {
        hipLaunchKernelGGL(nf_assign_slots, dim3(grid_for(n2, 256)), dim3(256), 0, ctx->stream, keys[buf], vals[buf],
                           (const unsigned int*)ids, n2, loc, slot_key);
        FHX_HIP(hipMemcpyAsync(&h_bad, bad, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
        FHX_HIP(hipStreamSynchronize(ctx->stream));
        if (h_bad) rc = fail(ctx, FHX_ERR_ARG, "contact rows hold a negative midpoint or chromosome id");
    }
    if (rc == FHX_OK && n_slots >= (1ll << 31)) rc = fail(ctx, FHX_ERR_UNSUPPORTED, "more than 2^31 loci");
    if (rc == FHX_OK) {
        ctx->n_slots = n_slots;
        ctx->n_dist = 1;
        ctx->grid.clear();
        rc = alloc_row_arrays(ctx, n, 1);
    }
    if (rc == FHX_OK) {
        hipLaunchKernelGGL(nf_finish_rows, dim3(grid_for(n, 256)), dim3(256), 0, ctx->stream, c1, c2, cnt, (const int32_t*)loc, n,
                           ctx->d_loc1, ctx->d_loc2, ctx->d_count);
        dev_free(ctx->d_slot_mid);
        dev_free(ctx->d_slot_chr);
        const size_t ns = (size_t)std::max<int64_t>(n_slots, 1);
        FHX_HIP(hipMalloc(&ctx->d_slot_mid, ns * sizeof(int32_t)));
        FHX_HIP(hipMalloc(&ctx->d_slot_chr, ns * sizeof(int16_t)));
        hipLaunchKernelGGL(nf_slot_tables, dim3(grid_for(n_slots, 256)), dim3(256), 0, ctx->stream,
                           (const unsigned long long*)slot_key, n_slots, ctx->d_slot_mid, ctx->d_slot_chr);
        ctx->h_slot_keys.assign((size_t)n_slots, 0ull);
        if (n_slots)
            FHX_HIP(hipMemcpyAsync(ctx->h_slot_keys.data(), slot_key, (size_t)n_slots * sizeof(unsigned long long),
                                   hipMemcpyDeviceToHost, ctx->stream));
        dev_free(ctx->d_seg_ids);
        dev_free(ctx->d_seg_tiles);
        const size_t cap = std::max<size_t>(4, (size_t)n);
        FHX_HIP(hipMalloc(&ctx->d_seg_ids, cap * sizeof(unsigned int)));
        FHX_HIP(hipMalloc(&ctx->d_seg_tiles, (cap / SEG_TILE + 2) * sizeof(unsigned int)));
        FHX_HIP(hipGetLastError());
        FHX_HIP(hipStreamSynchronize(ctx->stream));
        ctx->h_outlier_dists.clear();
    ctx->h_outlier_dists_global.clear();
    ctx->outlier_dists_are_global = false;
        ctx->h_dist_keys.clear();
    }
    return rc;
}

int ingest_device_rows(fhx_ctx* ctx, const int32_t* c1, const int32_t* m1, const int32_t* c2, const int32_t* m2,
                       const int32_t* cnt, int64_t n) {
    if (!ctx->have_params) return fail(ctx, FHX_ERR_ARG, "fhx_set_params must be called before fhx_load_pairs");
    if (n < 0) return fail(ctx, FHX_ERR_ARG, "negative row count");
    if (n >= (1ll << 32)) return fail(ctx, FHX_ERR_UNSUPPORTED, "more than 2^32 rows per GPU: shard the contacts");
    const int res = (int)ctx->prm.resolution;
    int n_chr = std::max(ctx->n_chr, 1);
    // FHX_TIMING=1: where the call's time goes (at 2e9 rows the arrays below are 158 GB: profiles/r06_cli_c5.txt)
    const bool timing = std::getenv("FHX_TIMING") != nullptr;
    auto t_last = std::chrono::steady_clock::now();
    std::string t_report;
    auto mark = [&](const char* what) {
        if (!timing) return;
        (void)hipStreamSynchronize(ctx->stream);
        const auto t = std::chrono::steady_clock::now();
        char b[96];
        std::snprintf(b, sizeof(b), " %s %.3f s;", what, std::chrono::duration<double>(t - t_last).count());
        t_report += b;
        t_last = t;
    };
    // the caller's chromosome id space may be larger than the fragments file's: scan for the maximum id is
    // folded into the extent kernel by giving it a generous table
    n_chr = std::max(n_chr, 4096);
    int32_t *d_maxidx = nullptr, *d_minoff = nullptr, *d_maxoff = nullptr, *d_bad = nullptr;
    DeviceScratch tmp;
    FHX_HIP(tmp.get(&d_maxidx, n_chr * sizeof(int32_t)));
    FHX_HIP(tmp.get(&d_minoff, n_chr * sizeof(int32_t)));
    FHX_HIP(tmp.get(&d_maxoff, n_chr * sizeof(int32_t)));
    FHX_HIP(tmp.get(&d_bad, sizeof(int32_t)));
    FHX_HIP(hipMemsetAsync(d_maxidx, 0xFF, n_chr * sizeof(int32_t), ctx->stream));       // -1
    FHX_HIP(hipMemsetAsync(d_minoff, 0x7F, n_chr * sizeof(int32_t), ctx->stream));       // large
    FHX_HIP(hipMemsetAsync(d_maxoff, 0xFF, n_chr * sizeof(int32_t), ctx->stream));       // -1
    FHX_HIP(hipMemsetAsync(d_bad, 0, sizeof(int32_t), ctx->stream));
    const int blocks = grid_for(n, 256);
    hipLaunchKernelGGL(k0_extent, dim3(blocks), dim3(256), 0, ctx->stream, c1, m1, n, res, n_chr, d_maxidx, d_minoff,
                       d_maxoff, d_bad);
    hipLaunchKernelGGL(k0_extent, dim3(blocks), dim3(256), 0, ctx->stream, c2, m2, n, res, n_chr, d_maxidx, d_minoff,
                       d_maxoff, d_bad);
    std::vector<int32_t> maxidx(n_chr), minoff(n_chr), maxoff(n_chr);
    int32_t bad = 0;
    FHX_HIP(hipMemcpyAsync(maxidx.data(), d_maxidx, n_chr * sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream));
    FHX_HIP(hipMemcpyAsync(minoff.data(), d_minoff, n_chr * sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream));
    FHX_HIP(hipMemcpyAsync(maxoff.data(), d_maxoff, n_chr * sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream));
    FHX_HIP(hipMemcpyAsync(&bad, d_bad, sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream));
    FHX_HIP(hipStreamSynchronize(ctx->stream));
    mark("extents of the chromosomes");
    if (bad) return fail(ctx, FHX_ERR_ARG, "contact rows hold a negative midpoint or a chromosome id outside [0, 4096)");
    int used = 0;
    for (int c = 0; c < n_chr; ++c)
        if (maxidx[c] >= 0) used = c + 1;
    used = std::max(used, ctx->n_chr);
    ctx->grid.assign(used, ChrGrid{0, -1, 0, 0});
    int64_t base = 0, n_dist = 1;          // histogram length = longest chromosome in slots + 1 spare index
    for (int c = 0; c < used; ++c) {
        ctx->grid[c].base = (int32_t)base;
        if (maxidx[c] >= 0) {
            if (minoff[c] != maxoff[c]) {
                // midpoints of one chromosome are not on one grid (mid % resolution differs): the reference still takes
                // abs(mid1 - mid2) of whatever the files hold (myUtils.py:112-124), so these rows go through the slotting of
                // the -r 0 path (sort + run detection) while the host keeps the fixed-size possible pairs
                ctx->offgrid = ctx->nonfixed = true;
                return ingest_device_rows_nonfixed(ctx, c1, m1, c2, m2, cnt, n);
            }
            ctx->grid[c].off = minoff[c];
            ctx->grid[c].nslots = maxidx[c] + 1;
            base += ctx->grid[c].nslots;
            n_dist = std::max<int64_t>(n_dist, (int64_t)ctx->grid[c].nslots + 1);
        }
    }
    if (base >= (1ll << 31)) return fail(ctx, FHX_ERR_UNSUPPORTED, "more than 2^31 loci");
    ctx->n_slots = base;
    ctx->n_dist = n_dist;
    dev_free(ctx->d_grid);
    FHX_HIP(hipMalloc(&ctx->d_grid, std::max<size_t>(1, ctx->grid.size()) * sizeof(ChrGrid)));
    FHX_HIP(hipMemcpyAsync(ctx->d_grid, ctx->grid.data(), ctx->grid.size() * sizeof(ChrGrid), hipMemcpyHostToDevice,
                           ctx->stream));
    {
        const int rc = alloc_row_arrays(ctx, n, n_dist);
        if (rc != FHX_OK) return rc;
    }
    mark("row arrays + workspace allocated");
    hipLaunchKernelGGL(k0_slots, dim3(blocks), dim3(256), 0, ctx->stream, c1, m1, c2, m2, cnt, n, res, ctx->d_grid,
                       ctx->d_loc1, ctx->d_loc2, ctx->d_count);
    FHX_HIP(hipGetLastError());
    FHX_HIP(hipStreamSynchronize(ctx->stream));
    mark("locus slots");
    if (timing) std::fprintf(stderr, "fhx_load_pairs_device: %lld rows:%s\n", (long long)n, t_report.c_str());
    return FHX_OK;
}

// per-row device arrays, histograms and workspaces for n rows; resets the pass state
int alloc_row_arrays(fhx_ctx* ctx, int64_t n, int64_t n_dist) {
    // row arrays (padded to a multiple of 4 rows for the 16-byte loads)
    const size_t cap = std::max<size_t>(4, ((size_t)n + 3) / 4 * 4);
    dev_free(ctx->d_loc1);
    dev_free(ctx->d_loc2);
    dev_free(ctx->d_count);
    dev_free(ctx->d_skip);
    dev_free(ctx->d_outlier);
    dev_free(ctx->d_seen_twice);
    dev_free(ctx->d_p);
    dev_free(ctx->d_q);
    dev_free(ctx->d_grow);
    FHX_HIP(hipMalloc(&ctx->d_loc1, cap * sizeof(int32_t)));
    FHX_HIP(hipMalloc(&ctx->d_loc2, cap * sizeof(int32_t)));
    FHX_HIP(hipMalloc(&ctx->d_count, cap * sizeof(int32_t)));
    FHX_HIP(hipMalloc(&ctx->d_skip, cap));
    FHX_HIP(hipMalloc(&ctx->d_outlier, cap));
    FHX_HIP(hipMalloc(&ctx->d_seen_twice, cap));
    FHX_HIP(hipMalloc(&ctx->d_p, cap * sizeof(double)));
    FHX_HIP(hipMalloc(&ctx->d_q, cap * sizeof(double)));
    ctx->q_prefilled = false;
    FHX_HIP(hipMemsetAsync(ctx->d_skip, 0, cap, ctx->stream));
    FHX_HIP(hipMemsetAsync(ctx->d_outlier, 0, cap, ctx->stream));
    FHX_HIP(hipMemsetAsync(ctx->d_seen_twice, 0, cap, ctx->stream));
    // histograms
    dev_free(ctx->d_hist_cc);
    dev_free(ctx->d_hist_np);
    dev_free(ctx->d_out_hist);
    const size_t hist_len = ctx->nonfixed ? cap : (size_t)n_dist;       // -r 0: at most one distinct distance per row
    FHX_HIP(hipMalloc(&ctx->d_hist_cc, hist_len * sizeof(unsigned long long)));
    FHX_HIP(hipMalloc(&ctx->d_hist_np, hist_len * sizeof(unsigned long long)));
    FHX_HIP(hipMalloc(&ctx->d_out_hist, hist_len * sizeof(unsigned long long)));
    FHX_HIP(hipMemsetAsync(ctx->d_out_hist, 0, hist_len * sizeof(unsigned long long), ctx->stream));
    if (!ctx->d_sums) FHX_HIP(hipMalloc(&ctx->d_sums, sizeof(K1Sums)));
    if (!ctx->d_misc) FHX_HIP(hipMalloc(&ctx->d_misc, 192 * sizeof(unsigned long long)));
    // One workspace, two views that are never live together (K2 and K3 run back to back on one stream):
    //   K2: queue[0] (16 B/row) | queue[1] (16 B/row) | the bucketed 300-iteration queue (16 B/row + bucket padding)
    //   K3: keys[0], keys[1] (8 B/row each)            | vals[0], vals[1] (4 B/row each)
    // (-r 0 sorts distances in K1 and lists outlier distances after K3 through the K3 view.)
    dev_free(ctx->d_work);
    const size_t qcap = std::max<size_t>(cap, (size_t)k2_classify_grid((int64_t)cap) * (size_t)k2_shard_capacity((int64_t)cap));   // sharded queues: k2_classify
    const size_t work_bytes = qcap * 32 + std::max(qcap, cap + (size_t)K2H_BUCKETS * 64 * K2H_MAX_ROWS) * sizeof(QEntry);   // queue 0 | queue 1 | sorted heavy queue / closed-form queue
    FHX_HIP(hipMalloc(&ctx->d_work, work_bytes));
    ctx->work_bytes = work_bytes;
    ctx->queue_cap = (int64_t)qcap;
    ctx->d_queue[0] = reinterpret_cast<QEntry*>(ctx->d_work);
    ctx->d_queue[1] = reinterpret_cast<QEntry*>(ctx->d_work + qcap * 16);
    ctx->d_queue_sorted = reinterpret_cast<QEntry*>(ctx->d_work + qcap * 32);
    ctx->d_keys[0] = reinterpret_cast<unsigned long long*>(ctx->d_work);
    ctx->d_keys[1] = reinterpret_cast<unsigned long long*>(ctx->d_work + cap * 8);
    ctx->d_vals[0] = reinterpret_cast<unsigned int*>(ctx->d_work + cap * 16);
    ctx->d_vals[1] = reinterpret_cast<unsigned int*>(ctx->d_work + cap * 20);
    if (!ctx->d_block_hist) FHX_HIP(hipMalloc(&ctx->d_block_hist, (size_t)RADIX * SORT_BLOCKS * sizeof(unsigned int)));
    if (!ctx->d_digit_total) FHX_HIP(hipMalloc(&ctx->d_digit_total, RADIX * sizeof(unsigned int)));
    if (!ctx->d_top_hist) FHX_HIP(hipMalloc(&ctx->d_top_hist, TOP_BINS * sizeof(unsigned long long)));
    if (!ctx->d_cf_tab) FHX_HIP(hipMalloc(&ctx->d_cf_tab, (size_t)K2H_GENERIC * dev::kCfIters * sizeof(dev::CfRow)));
    if (!ctx->d_k2h_off) FHX_HIP(hipMalloc(&ctx->d_k2h_off, (K2H_BUCKETS + 1) * sizeof(unsigned int)));
    dev_free(ctx->d_tile_max);
    FHX_HIP(hipMalloc(&ctx->d_tile_max, ((size_t)n / BH_TILE + 2) * sizeof(double)));
    FHX_HIP(hipStreamSynchronize(ctx->stream));
    ctx->n_rows = n;
    ctx->pass_no = 0;
    ctx->skip_active = false;
    ctx->have_stats = ctx->have_fit = ctx->have_p = ctx->have_q = false;
    ctx->n_outliers_total = 0;
    ctx->skip_limit = INT64_MAX;
    ctx->outlier_hist_nonempty = false;
    ctx->h_out_hist.assign((size_t)n_dist, 0);
    ctx->tables_dirty = true;
    ctx->n_sorted = -1;
    ctx->dist_ndist_agreed = false;
    ctx->dist_ndist_global = -1;
    ctx->dist_any_nonfixed = false;
    ctx->h_outlier_dists_global.clear();
    ctx->outlier_dists_are_global = false;
    if (!ctx->nonfixed) ctx->h_dist_keys.clear();
    return FHX_OK;
}


}  // namespace fhx

// ---- C ABI -----------------------------------------------------------------------------------------------------------------
// -r 0: classification + sums as K1, then the in-range (distance, count) pairs are radix-sorted by distance and the runs
// are reduced to (distinct distance, sum of counts, rows): the reference's mainDic for arbitrary distances
int fhx::pass_stats_nonfixed(fhx_ctx* ctx, fhx_stats* out) {
    unsigned long long* counter = ctx->d_misc + 10;
    FHX_HIP(hipMemsetAsync(counter, 0, sizeof(unsigned long long), ctx->stream));
    FHX_HIP(hipMemsetAsync(ctx->d_sums, 0, sizeof(K1Sums), ctx->stream));
    before_rerecord(ctx, 0);
    FHX_HIP(hipEventRecord(ctx->ev[0], ctx->stream));
    hipLaunchKernelGGL(nf_k1_classify, dim3(grid_for(ctx->n_rows, SORT_TILE, 256 * 8)), dim3(SORT_THREADS), 0, ctx->stream,
                       ctx->d_loc1, ctx->d_loc2, ctx->d_count, ctx->skip_active ? ctx->d_skip : (const uint8_t*)nullptr,
                       ctx->skip_limit, (ctx->skip_limit != INT64_MAX) ? (const long long*)ctx->d_grow : (const long long*)nullptr,
                       ctx->n_rows, (const int32_t*)ctx->d_slot_mid, (long long)ctx->prm.dist_low, (long long)ctx->prm.dist_up,
                       ctx->d_keys[0], ctx->d_vals[0], counter, ctx->d_sums);
    int buf = 0;
    int rc = radix_sort_pairs(ctx, ctx->d_keys, ctx->d_vals, counter, 3, &buf);       // distances < 2^31: 33 key bits
    if (rc != FHX_OK) return rc;
    K1Sums s{};
    unsigned long long n_keys = 0;
    FHX_HIP(hipMemcpyAsync(&s, ctx->d_sums, sizeof(K1Sums), hipMemcpyDeviceToHost, ctx->stream));
    FHX_HIP(hipMemcpyAsync(&n_keys, counter, sizeof(n_keys), hipMemcpyDeviceToHost, ctx->stream));
    FHX_HIP(hipStreamSynchronize(ctx->stream));
    int64_t n_runs = 0;
    rc = run_ids(ctx, ctx->d_keys[buf], (int64_t)n_keys, ctx->d_seg_ids, ctx->d_seg_tiles, &n_runs);
    if (rc != FHX_OK) return rc;
    ctx->h_dist_keys.assign((size_t)n_runs, 0);
    ctx->h_hist_cc.assign((size_t)n_runs, 0);
    ctx->h_hist_np.assign((size_t)n_runs, 0);
    if (n_runs) {
        FHX_HIP(hipMemsetAsync(ctx->d_hist_cc, 0, (size_t)n_runs * sizeof(unsigned long long), ctx->stream));
        FHX_HIP(hipMemsetAsync(ctx->d_hist_np, 0, (size_t)n_runs * sizeof(unsigned long long), ctx->stream));
        hipLaunchKernelGGL(nf_accumulate_runs, dim3(grid_for((int64_t)n_keys, 256)), dim3(256), 0, ctx->stream,
                           (const unsigned long long*)ctx->d_keys[buf], (const unsigned int*)ctx->d_vals[buf],
                           (const unsigned int*)ctx->d_seg_ids, (int64_t)n_keys, ctx->d_out_hist, ctx->d_hist_cc, ctx->d_hist_np);
        FHX_HIP(hipGetLastError());
        FHX_HIP(hipEventRecord(ctx->ev[1], ctx->stream));
        FHX_HIP(hipMemcpyAsync(ctx->h_dist_keys.data(), ctx->d_out_hist, (size_t)n_runs * sizeof(int64_t), hipMemcpyDeviceToHost, ctx->stream));
        FHX_HIP(hipMemcpyAsync(ctx->h_hist_cc.data(), ctx->d_hist_cc, (size_t)n_runs * sizeof(int64_t), hipMemcpyDeviceToHost, ctx->stream));
        FHX_HIP(hipMemcpyAsync(ctx->h_hist_np.data(), ctx->d_hist_np, (size_t)n_runs * sizeof(int64_t), hipMemcpyDeviceToHost, ctx->stream));
        FHX_HIP(hipStreamSynchronize(ctx->stream));
    } else {
        FHX_HIP(hipEventRecord(ctx->ev[1], ctx->stream));
        FHX_HIP(hipStreamSynchronize(ctx->stream));
    }
    ctx->ev_valid[0] = true;
    ctx->ev_folded[0] = false;
    fold_kernel_events(ctx);
    fhx_stats& st = ctx->stats;
    st.n_rows = ctx->n_rows;
    st.inter_count = s.inter_count;
    st.inter_sum = s.inter_sum;
    st.intra_all_count = s.intra_all_count;
    st.intra_all_sum = s.intra_all_sum;
    st.in_range_count = s.in_range_count;
    st.in_range_sum = s.in_range_sum;
    st.max_count = s.max_count;
    st.n_dist = n_runs;
    st.n_skipped = s.n_skipped;
    ctx->have_stats = true;
    ctx->have_fit = ctx->have_bins = ctx->have_p = ctx->have_q = false;
    if (out) *out = st;
    return FHX_OK;
}

namespace fhx {
__global__ __launch_bounds__(256) void k1_zero(unsigned long long* __restrict__ hist_cc, unsigned long long* __restrict__ hist_np, int64_t n_dist,
                                               K1Sums* __restrict__ sums) {
    const int64_t i0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (int64_t i = i0; i < n_dist; i += (int64_t)gridDim.x * blockDim.x) {
        hist_cc[i] = 0ull;
        hist_np[i] = 0ull;
    }
    static_assert(sizeof(K1Sums) % sizeof(unsigned long long) == 0, "K1Sums is a block of 8-byte words");
    if (i0 < (int64_t)(sizeof(K1Sums) / sizeof(unsigned long long))) reinterpret_cast<unsigned long long*>(sums)[i0] = 0ull;
}
}  // namespace fhx

// K1 of the fixed-size path on the context's stream: histograms and sums stay in HBM
int fhx::launch_k1(fhx_ctx* ctx) {
    const int64_t res = ctx->prm.resolution;
    const int64_t lo = (ctx->prm.dist_low + res - 1) / res;
    const int64_t hi = std::min<int64_t>(ctx->prm.dist_up / res, ctx->n_dist - 1);
    // both histograms and the sums zeroed by ONE launch (three memsets were three fill kernels of ~4.5 us each in front of K1, and
    // three enqueues on the host while the GPU waits for the pass to start)
    hipLaunchKernelGGL(k1_zero, dim3(grid_for(std::max<int64_t>(ctx->n_dist, 1), 256, 256)), dim3(256), 0, ctx->stream, ctx->d_hist_cc,
                       ctx->d_hist_np, (int64_t)ctx->n_dist, ctx->d_sums);
    before_rerecord(ctx, 0);
    FHX_HIP(hipEventRecord(ctx->ev[0], ctx->stream));
    const uint8_t* skip = ctx->skip_active ? ctx->d_skip : (const uint8_t*)nullptr;
    const long long* grow = (ctx->skip_limit != INT64_MAX) ? (const long long*)ctx->d_grow : (const long long*)nullptr;
    const int lo_i = (int)std::min<int64_t>(lo, INT32_MAX);
    static const bool force_narrow = std::getenv("FHX_K1_NARROW") != nullptr;      // measurements only
    if (hi - lo + 1 > K1_LDS_BINS && !force_narrow) {                                // more distance values than the 12-B window holds
        const size_t lds = (size_t)K1_WIDE_BINS * 6;
        static bool attr_done = false;
        if (!attr_done) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k1_classify_hist<K1_WIDE_THREADS, true>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            attr_done = true;
        }
        const int blocks = grid_for((ctx->n_rows + 3) / 4, K1_WIDE_THREADS, 256);
        hipLaunchKernelGGL((k1_classify_hist<K1_WIDE_THREADS, true>), dim3(blocks), dim3(K1_WIDE_THREADS), lds, ctx->stream, ctx->d_loc1,
                           ctx->d_loc2, ctx->d_count, skip, ctx->skip_limit, grow, ctx->n_rows, lo_i, (int)hi, ctx->d_hist_cc,
                           ctx->d_hist_np, ctx->d_sums);
    } else {
        const size_t lds = (size_t)K1_LDS_BINS * (sizeof(unsigned long long) + sizeof(unsigned int));
        const int blocks = grid_for((ctx->n_rows + 3) / 4, K1_THREADS, 512);
        hipLaunchKernelGGL((k1_classify_hist<K1_THREADS, false>), dim3(blocks), dim3(K1_THREADS), lds, ctx->stream, ctx->d_loc1,
                           ctx->d_loc2, ctx->d_count, skip, ctx->skip_limit, grow, ctx->n_rows, lo_i, (int)hi, ctx->d_hist_cc,
                           ctx->d_hist_np, ctx->d_sums);
    }
    FHX_HIP(hipGetLastError());
    FHX_HIP(hipEventRecord(ctx->ev[1], ctx->stream));
    ctx->ev_valid[0] = true;
    ctx->ev_folded[0] = false;
    return FHX_OK;
}

// Adds the event pairs that have not been counted yet to the context's sums.  The caller has just synchronised the stream, and
// every pair recorded so far lies before that point.
void fhx::fold_kernel_events(fhx_ctx* ctx) {
    for (int k = 0; k < 4; ++k) {
        if (ctx->ev_folded[k] || !ctx->ev_valid[k < 3 ? k : 1]) continue;
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, ctx->ev[k < 3 ? 2 * k : 6], ctx->ev[k < 3 ? 2 * k + 1 : 7]) == hipSuccess) {
            ctx->ev_sum[k] += (double)ms * 1e-3;
            ctx->ev_count[k] += 1;
        }
        ctx->ev_folded[k] = true;
    }
}

// Each kernel group has ONE event pair, recorded again by every pass.  A caller that runs a group twice without a statistics call
// or fhx_kernel_seconds_total in between (pvalues() repeated, bh re-run) would overwrite a pair nobody has read: the pair is
// read here if the stream has passed it (no waiting), else counted in ev_dropped - fhx_kernel_events_dropped tells a timing
// harness that its mean covers fewer passes than it ran.
void fhx::before_rerecord(fhx_ctx* ctx, int group) {
    for (int k : {group, group == 1 ? 3 : -1}) {
        if (k < 0 || ctx->ev_folded[k] || !ctx->ev_valid[k < 3 ? k : 1]) continue;
        hipEvent_t a = ctx->ev[k < 3 ? 2 * k : 6], b = ctx->ev[k < 3 ? 2 * k + 1 : 7];
        float ms = 0.f;
        if (hipEventQuery(b) == hipSuccess && hipEventElapsedTime(&ms, a, b) == hipSuccess) {
            ctx->ev_sum[k] += (double)ms * 1e-3;
            ctx->ev_count[k] += 1;
        } else {
            (void)hipGetLastError();               // hipErrorNotReady is not an error of the pass
            ctx->ev_dropped[k] += 1;
        }
        ctx->ev_folded[k] = true;
    }
}

int fhx_pass_stats(fhx_ctx* ctx, fhx_stats* out) {
    if (!ctx) return FHX_ERR_ARG;
    if (ctx->device < 0) return fail(ctx, FHX_ERR_NO_DEVICE, "host-only context");
    if (!ctx->d_loc1) return fail(ctx, FHX_ERR_ARG, "no contact rows loaded");
    FHX_HIP(hipSetDevice(ctx->device));
    if (ctx->nonfixed) return pass_stats_nonfixed(ctx, out);
    {
        const int rc = launch_k1(ctx);
        if (rc != FHX_OK) return rc;
    }
    // the sums and the in-range window of the two histograms (K1 touches no bin outside it) packed on the device and copied
    // in one piece into pinned memory: three pageable copies of 64 B + 2 x n_dist x 8 B were ~100 us of a small shard's pass
    const int64_t res = ctx->prm.resolution;
    const int64_t nd = ctx->n_dist;
    const int64_t a = std::min<int64_t>(std::max<int64_t>(0, (ctx->prm.dist_low + res - 1) / res), nd);
    const int64_t b = (ctx->prm.dist_up == INT64_MAX) ? nd : std::max(a, std::min<int64_t>(nd, ctx->prm.dist_up / res + 1));
    const int w = (int)(b - a);
    const size_t pack_len = 8 + 2 * (size_t)w;
    if (pack_len > ctx->stats_stage_cap) {
        FHX_HIP(hipStreamSynchronize(ctx->stream));
        if (ctx->h_stats_stage) (void)hipHostFree(ctx->h_stats_stage);
        ctx->h_stats_stage = nullptr;
        ctx->stats_stage_cap = 0;
        FHX_HIP(hipHostMalloc((void**)&ctx->h_stats_stage, (pack_len + 1024) * sizeof(long long), hipHostMallocCoherent | hipHostMallocMapped));
        ctx->stats_stage_cap = pack_len + 1024;
    }
    {
        const int rc = ensure_flags(ctx);
        if (rc != FHX_OK) return rc;
    }
    // the kernel stores the block straight into the pinned host buffer and publishes a ticket behind it; the host spins on the
    // ticket (wait_ticket): no copy dispatch, no interrupt + wake-up between K1 and the fit (a 1/8 shard of C3: the gap between
    // K1 and k2_classify 203 -> 155 us, profiles/r06_a_tl_shard8.txt)
    long long* d_pack = nullptr;
    FHX_HIP(hipHostGetDevicePointer((void**)&d_pack, ctx->h_stats_stage, 0));
    unsigned long long* d_flag = nullptr;
    FHX_HIP(hipHostGetDevicePointer((void**)&d_flag, (void*)ctx->h_flags, 0));
    const unsigned long long ticket = ++ctx->ticket;
    hipLaunchKernelGGL(k1_pack_window, dim3(grid_for((int64_t)pack_len, 256, 64)), dim3(256), 0, ctx->stream, (const K1Sums*)ctx->d_sums,
                       (const unsigned long long*)ctx->d_hist_cc, (const unsigned long long*)ctx->d_hist_np, (int)a, w, d_pack, ctx->d_done,
                       (volatile unsigned long long*)(d_flag + FLAG_K1), ticket);
    FHX_HIP(hipGetLastError());
    // behind it, off the host's critical path: what fhx_pvalues would otherwise have to zero before it can classify
    ctx->k2_prezeroed = false;
    if (ctx->d_block_hist && ctx->d_k2_hist) {
        static const bool prefill = !(std::getenv("FHX_Q_PREFILL") && std::atoi(std::getenv("FHX_Q_PREFILL")) == 0);     // 0: measurements
        hipLaunchKernelGGL(k1_prezero, dim3(prefill ? 2048 : 1024), dim3(256), 0, ctx->stream, ctx->d_k2_hist, (int64_t)TOP_BINS,
                           reinterpret_cast<uint4*>(ctx->d_block_hist), (int64_t)K2H_BUCKETS * K2H_BLOCKS / 4, ctx->d_q,
                           prefill ? ctx->n_rows : (int64_t)0);
        FHX_HIP(hipGetLastError());
        ctx->k2_prezeroed = true;
        ctx->q_prefilled = prefill;
    }
    FHX_HIP(wait_ticket(ctx, FLAG_K1, ticket));
    fold_kernel_events(ctx);               // this pass's K1 and, behind it on the stream, the previous pass's K2 and K3
    const long long* pk = ctx->h_stats_stage;
    ctx->h_hist_cc.assign((size_t)nd, 0);
    ctx->h_hist_np.assign((size_t)nd, 0);
    for (int i = 0; i < w; ++i) {
        ctx->h_hist_cc[(size_t)a + i] = pk[8 + i];
        ctx->h_hist_np[(size_t)a + i] = pk[8 + w + i];
    }
    fhx_stats& st = ctx->stats;
    st.n_rows = ctx->n_rows;
    st.inter_count = pk[0];
    st.inter_sum = pk[1];
    st.intra_all_count = pk[2];
    st.intra_all_sum = pk[3];
    st.in_range_count = pk[4];
    st.in_range_sum = pk[5];
    st.max_count = pk[7];
    st.n_dist = ctx->n_dist;
    st.n_skipped = pk[6];
    ctx->have_stats = true;
    ctx->have_fit = ctx->have_bins = ctx->have_p = ctx->have_q = false;
    if (out) *out = st;
    return FHX_OK;
}

