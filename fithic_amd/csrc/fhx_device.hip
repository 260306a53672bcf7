// fhx_device.hip - gfx950 kernels + the C-ABI context of libfithic_mi355x.so.
//
//   K0 ingest          raw (chr, mid) rows -> 12-byte SoA rows (slot1, slot2|inter-flag, count)
//   K1 classify_hist   fithic.read_Interactions     (fithic/fithic.py:389-454)
//   K2 pvalue          fithic.fit_Spline pair loop  (fithic/fithic.py:1017-1124) + Cephes bdtrc
//   K3 bh_*            myStats.benjamini_hochberg_correction (fithic/myStats.py:24-48):
//                      compact p < 1 -> LSD radix sort of the IEEE bit patterns -> min(p*N/rank,1) ->
//                      inclusive max-scan -> scatter
//
// CDNA4 notes: wave64 everywhere (ballots are 64-bit); pair arrays are streamed with 16-byte-per-lane
// coalesced loads; the distance histogram is privatised in LDS (int64 sums + int32 row counts) and flushed
// with one global atomic per touched bin per workgroup; there is no dense contraction on this path, so no
// MFMA; the whole TU is compiled with -ffp-contract=off (see fhx_bdtrc.hpp for why).
#include <hip/hip_runtime.h>

#include <atomic>
#include <chrono>
#include <thread>
#include <memory>
#include <sys/stat.h>
#include <unistd.h>
#include <deque>
#include <mutex>
#include <condition_variable>
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/fithic_mi355x.h"
#include "fhx_bdtrc.hpp"
#include "fhx_host.hpp"
#include "fhx_io_internal.hpp"
#include "fhx_scan.hpp"

namespace fhx {

// ===================================================================================================
// small device helpers
// ===================================================================================================
__device__ __forceinline__ long long wave_sum_i64(long long v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;
}
__device__ __forceinline__ int wave_max_i32(int v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = max(v, __shfl_down(v, off, 64));
    return v;
}

// inclusive add-scan over the 64 lanes of a wave in six DPP adds (row shifts inside each row of 16 lanes, then the last lane of
// rows 0 / 2 broadcast into rows 1 / 3 and lane 31 into the upper half); lanes a shift leaves without a source add 0
__device__ __forceinline__ unsigned int wave_incl_sum_u32(unsigned int v) {
    v += (unsigned int)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, false);     // row_shr:1
    v += (unsigned int)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, false);     // row_shr:2
    v += (unsigned int)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, false);     // row_shr:4
    v += (unsigned int)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, false);     // row_shr:8
    v += (unsigned int)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false);     // row_bcast:15 -> rows 1, 3
    v += (unsigned int)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false);     // row_bcast:31 -> rows 2, 3
    return v;
}

// ===================================================================================================
// K0: ingest.  Slot of a locus = chr_base[chr] + mid / res; a chromosome's loci must share mid % res
// (that is what "fixed-size" data looks like: createFitHiCFragments-fixedsize.py writes mid = i*res + res/2).
// ===================================================================================================
struct ChrGrid {          // per chromosome id, device copy
    int32_t base;         // first slot
    int32_t off;          // mid % res shared by the chromosome's loci (-1: chromosome unseen)
    int32_t nslots;
    int32_t pad;
};

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

struct K1Sums {           // device accumulator block (int64 each)
    long long inter_count, inter_sum, intra_all_count, intra_all_sum, in_range_count, in_range_sum, n_skipped;
    int max_count, pad;
};

// WIDE = false: the window holds 6144 bins as (u64 sum, u32 rows): every Hi-C run with a distance cap (C3: 397 bins).
// WIDE = true (more distance values than that, e.g. no -U: 49 847 at 5 kb): one 1024-thread workgroup per CU owns 144 KB =
// 24 576 bins as (u32 sum, 15-bit row count + guard bit); a sum that wraps adds 2^32 to the bin in HBM (exactly one thread
// sees the wrap), a row count that reaches 2^15 sets the guard bit, which cannot carry into the neighbouring half-word, and
// the thread that set it moves 2^15 to HBM and clears it.  With 12 B/bin only a quarter of the bins of that run were in LDS
// and the rest took two device-scope atomics per row: K1 24.8 ms per 1.06e9 rows (profiles/r02_p_c3w_bench.json).
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
                               const unsigned long long* __restrict__ hist_np, int a, int w, long long* __restrict__ pack) {
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
}

// ===================================================================================================
// K2: per-pair prior + binomial survival p-value
// ===================================================================================================
struct K2Params {
    const int32_t* loc1;
    const int32_t* loc2;
    const int32_t* count;
    const double* slot_bias;      // -1 = discarded / missing; all 1.0 without a bias file
    bool no_bias;                 // no bias table was loaded: slot_bias is all 1.0 and need not be read
    const double* prior_lut;      // newSplineY by distance index (clamp + bisect_left folded in)
    int lut_len;                  // entries of prior_lut (= length of the distance histogram)
    dev::BinomTables intra, inter;
    double inter_chr_prob;
    double outlier_thres;         // 1/N
    int lo_idx, hi_idx;
    int mode;
    int64_t n;
    double* p;
    // K3's key histogram accumulated where p is stored (4096 bins = key >> 50 of p <= 1): LDS-privatised per workgroup, one
    // flush per workgroup; nullptr = not collected (k3_top_hist reads p again instead)
    unsigned long long* top_hist;
    uint8_t* outlier;             // p < 1/N, feeds the next pass
    // non-fixed-size mode (-r 0): loci are ranks into the sorted distinct (chr, mid) list, distances come from slot_mid,
    // and the prior is found by bisect_left over the spline table (fithic.py:1066-1069) instead of a dense LUT
    int nonfixed;
    const int32_t* slot_mid;
    const double* table_x;
    const double* table_y;
    int n_table;
    double min_x, max_x;
    long long dist_low, dist_up;
};

__device__ __forceinline__ double prior_by_search(const K2Params& P, long long dist) {
    double look = (double)dist;
    if (look < P.min_x) look = P.min_x;                                   // max(d, min(x))
    if (look > P.max_x) look = P.max_x;                                   // min(., max(x))
    int lo = 0, hi = P.n_table;                                           // bisect_left
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (P.table_x[mid] < look)
            lo = mid + 1;
        else
            hi = mid;
    }
    return P.table_y[min(lo, P.n_table - 1)];
}

// prior and which binomial a row uses; returns false when the row's p-value is the constant 1.0.
// NF: 0 = fixed-size loci on a grid, 1 = arbitrary loci (-r 0 / off-grid), -1 = decided at run time (P.nonfixed); the
// specialised forms keep the other mode's fields out of the kernel (SGPRs, and the search loop's code).
template <int NF = -1>
__device__ __forceinline__ bool row_prior(const K2Params& P, int l1, int l2, double& prior, bool& is_inter) {
    const bool inter = l2 < 0;
    const int s2 = inter ? ~l2 : l2;
    // no bias table: every slot holds 1.0 - skip the two gathers
    const double b1 = P.no_bias ? 1.0 : P.slot_bias[l1], b2 = P.no_bias ? 1.0 : P.slot_bias[s2];
    if ((b1 < 0 || b2 < 0) && !inter) return false;                        // fithic.py:1057-1064
    if (!inter && P.mode != FHX_MODE_INTER_ONLY) {
        if (NF == 1 || (NF == -1 && P.nonfixed)) {
            const long long dist = llabs((long long)P.slot_mid[l1] - (long long)P.slot_mid[s2]);
            if (dist < P.dist_low || dist > P.dist_up) return false;
            prior = prior_by_search(P, dist) * (b1 * b2);
            is_inter = false;
            return true;
        }
        const int d = abs(l1 - s2);
        if (d < P.lo_idx || d > P.hi_idx) return false;                   // intraShort / intraLong: p = 1
        prior = P.prior_lut[d] * (b1 * b2);                               // fithic.py:1069
        is_inter = false;
        return true;
    }
    if (P.mode == FHX_MODE_INTRA_ONLY) return false;                      // inter row in intraOnly mode
    prior = P.inter_chr_prob * (b1 * b2);                                 // fithic.py:1100 (also intra rows when interOnly)
    is_inter = true;
    return true;
}

// row_prior<0> for the four rows a lane of k2_classify holds, with every gather issued up front: the three table reads of a row
// (two biases, the prior by distance index) do not depend on the row's fate, so all twelve go out back to back - unconditionally,
// on clamped indices - and the branch table of fithic.py:1057-1116 is applied to the values afterwards.  (Evaluated row by row,
// each row's gathers sat behind the previous row's classification: four exposed round trips per step at four waves per SIMD.)
// Same values, same order of the two multiplications: prior = table * (b1 * b2).
template <int ITEMS>
__device__ __forceinline__ void rows_prior_fixed(const K2Params& P, const int (&l1)[ITEMS], const int (&l2)[ITEMS], double (&prior)[ITEMS],
                                                 bool (&is_inter)[ITEMS], bool (&live)[ITEMS]) {
    double b1[ITEMS], b2[ITEMS], tab[ITEMS];
    int dist[ITEMS];
#pragma unroll
    for (int r = 0; r < ITEMS; ++r) {
        const bool inter = l2[r] < 0;
        const int s2 = inter ? ~l2[r] : l2[r];
        dist[r] = abs(l1[r] - s2);
        b1[r] = P.no_bias ? 1.0 : P.slot_bias[l1[r]];
        b2[r] = P.no_bias ? 1.0 : P.slot_bias[s2];
        tab[r] = P.prior_lut[min(dist[r], P.lut_len - 1)];             // inter rows: any entry, unused
    }
#pragma unroll
    for (int r = 0; r < ITEMS; ++r) {
        const bool inter = l2[r] < 0;
        const double bb = b1[r] * b2[r];
        const bool as_intra = !inter && P.mode != FHX_MODE_INTER_ONLY;
        const bool discarded = (b1[r] < 0 || b2[r] < 0) && !inter;                      // fithic.py:1057-1064
        const bool in_range = dist[r] >= P.lo_idx && dist[r] <= P.hi_idx;
        live[r] = !discarded && (as_intra ? in_range : P.mode != FHX_MODE_INTRA_ONLY);
        is_inter[r] = !as_intra;
        prior[r] = live[r] ? (as_intra ? tab[r] : P.inter_chr_prob) * bb : 1.0;        // fithic.py:1069 / :1100
    }
}

// The top-bits histogram of K3's early cutoff (k3_top_hist) gathered by the kernels that store p: bdtrc values are NaN or in
// [0, 1], so key >> 50 < 4096; p == 1.0 (most rows) goes through a per-thread counter.  One LDS table per workgroup.
constexpr int K2_HIST_BINS = 4096;
struct FusedHist {
    unsigned int* h;
    unsigned int ones;
    bool on;
    __device__ __forceinline__ void init(unsigned int* lds, const unsigned long long* global) {
        h = lds;
        ones = 0;
        on = global != nullptr;
        if (on) {
            for (int i = threadIdx.x; i < K2_HIST_BINS; i += blockDim.x) h[i] = 0;
            __syncthreads();
        }
    }
    __device__ __forceinline__ void add(double v) {
        if (!on) return;
        if (v == 1.0)
            ++ones;
        else if (v == v) {
            unsigned long long bits = (unsigned long long)__double_as_longlong(v);
            if (bits == 0x8000000000000000ull) bits = 0ull;
            atomicAdd(&h[min((unsigned int)(bits >> 50), (unsigned int)K2_HIST_BINS - 1u)], 1u);
        }
    }
    // All values of the wave counted in the bin of its SMALLEST one: a single LDS atomic instead of 64 on a handful of words
    // (the 300-iteration class yields p in [0.5, 1): three or four bins for a whole launch).  Counting a value in a lower bin
    // than its own is exact for the cutoff: cumulative counts only grow, `bin_saturates` is decreasing in the count, so a
    // bin found saturating this way saturates with the true counts too, and the first true value at or above its edge has a
    // rank within the inflated count.  At worst a few more rows are sorted.
    __device__ __forceinline__ void add_wave_min(double v, bool valid) {
        if (!on) return;
        valid = valid && v == v;
        unsigned long long key = valid ? (unsigned long long)__double_as_longlong(v) : ~0ull;
        if (key == 0x8000000000000000ull) key = 0ull;
#pragma unroll
        for (int s = 32; s >= 1; s >>= 1) {
            const unsigned long long o = __shfl_xor(key, s, 64);
            key = o < key ? o : key;
        }
        const unsigned int n = (unsigned int)__popcll(__ballot(valid));
        if ((threadIdx.x & 63) == 0 && n) atomicAdd(&h[min((unsigned int)(key >> 50), (unsigned int)K2_HIST_BINS - 1u)], n);
    }
    __device__ __forceinline__ void flush(unsigned long long* global) {      // every thread of the workgroup must call it
        if (!on) return;
        const unsigned int w = (unsigned int)wave_sum_i64((long long)ones);
        if ((threadIdx.x & 63) == 0 && w) atomicAdd(&h[0x3FF0000000000000ull >> 50], w);
        __syncthreads();
        for (int i = threadIdx.x; i < K2_HIST_BINS; i += blockDim.x)
            if (h[i]) atomicAdd(&global[i], (unsigned long long)h[i]);
    }
};

constexpr int K2_THREADS = 256;

// K2 runs as one classification launch plus one launch per branch class so that waves are branch-homogeneous
// (SURVEY appendix C / F): the iteration count of Cephes' incbet is multi-modal - none for the closed form, ~15 for the
// power series, ~9 for the converging continued fractions and (practically always) the full 300 for the swapped
// continued fraction ("observed < expected").  k2_classify finishes the loop-free class in place and appends every
// other row to the queue of its class (wave-aggregated: one atomic per wave and class); k2_queue then runs one
// class at a time with every lane on the same code path and nearly the same trip count.
constexpr int K2_QUEUES = dev::BC_COUNT - 1;        // classes 1..4

// p of a queued row goes to p[row]: an 8-byte store into a line nobody reads again before K3.  In the queue-order kernels the
// rows of a wave are neighbours and the L2 merges their stores into whole lines; the bucket-sorted heavy class scatters them over
// the whole column, and a plain store then makes the L2 FETCH every line it partially writes (PMC, k2h_heavy: 1639 MB read per
// launch for 427 MB of entries; 713 MB with nontemporal stores, which write through without allocating - and 0.5 ms less per
// pass, profiles/r02_y_*).  The queue-order kernels keep plain stores (nontemporal ones cost them 10-28 % more write traffic).
template <bool SCATTERED>
__device__ __forceinline__ void store_p(double* dst, double v) {
    if (SCATTERED)
        __builtin_nontemporal_store(v, dst);
    else
        *dst = v;
}

// one queued row: everything the per-class kernel needs, so that it streams 16 B/row instead of re-gathering the
// three pair columns, two biases and the prior LUT through a row index (measured 66 B/row of HBM traffic that way)
struct QEntry {
    unsigned int row;
    int count;                  // negative: the row uses the inter-chromosomal binomial (n = observedInterAllSum)
    double prior;
};

// Queues are SHARDED BY WORKGROUP: workgroup b of k2_classify appends only to shard b of every class queue, so its slot counters
// are its own (LDS, kept across its tiles and written to HBM once at the end) and nothing in its tile loop waits for another
// workgroup - or for another wave: a wave reserves its slots with one LDS atomic per class and goes on.  (Round 2 took one
// returning GLOBAL atomic per class and tile, eight counters per class: all workgroups adding to one address retire at ~88 M
// atomics/s on this chip, and the two block barriers around that round trip left the kernel at 29 % of HBM and 49 % VALU busy -
// bound by neither, profiles/r02_z_pmc.txt.)  A shard's tiles are known in advance (tile t belongs to workgroup t % grid), so a
// region of ceil(tiles / grid) tiles per shard can never overflow; two classes share a buffer, growing towards each other
// inside every shard's region.  Consumers walk the queue shard by shard (a consumer workgroup takes whole shards: no index
// arithmetic over shard boundaries).
constexpr int K2_MAX_SHARDS = 2048;                      // the largest k2_classify grid: 256 CUs x 8

struct QSpan {
    QEntry* base;                      // slot 0 of shard 0 (queues that grow downwards: the LAST entry of shard 0's region)
    long long cap_s;                   // entries per shard region
    int dir;                           // +1 / -1
    int n_shards;
    const unsigned long long* count;   // n_shards counters
};
__device__ __forceinline__ QEntry* qentry(const QSpan& q, int shard, long long j) {
    return q.base + (long long)shard * q.cap_s + (long long)q.dir * j;
}

struct K2Queues {
    QSpan q[K2_QUEUES + 1];            // classes 1..4, then the closed-form class (count == 1, prior >= 0.01)
    unsigned long long* count;         // (K2_QUEUES + 1) x K2_MAX_SHARDS counters: [class * K2_MAX_SHARDS + shard]
    unsigned int* heavy_hist;          // K2H_BUCKETS x K2H_BLOCKS bucket counts of the swapped-fraction queue (zeroed before the
                                       // launch; column = shard % K2H_BLOCKS, the workgroup of k2h_scatter that will move the shard);
                                       // nullptr = not collected
};

// bucket of a swapped-continued-fraction row in the count sort that feeds k2h_heavy (defined with that sort, below); k2_classify
// counts its shard's rows per bucket while it queues them, so that the sort needs no counting pass of its own
constexpr int K2H_BUCKETS = 2048;                       // == RADIX: the radix sort's count matrix and scan are reused
constexpr int K2H_BLOCKS = 1024;                        // == SORT_BLOCKS
__device__ __forceinline__ int k2h_bucket(int signed_count);

constexpr int K2_CL_ITEMS = 4;
constexpr int K2_CL_TILE = K2_THREADS * K2_CL_ITEMS;     // 1024 rows per workgroup step: four waves of 256 consecutive rows

constexpr int K2_CLOSED = K2_QUEUES + 1;                 // count == 1 rows with prior >= 0.01 (Cephes takes pow there): queued, k2_closed
constexpr int K2_CLOSED_LOCAL = K2_QUEUES + 2;           // count == 1 rows with prior < 0.01: wave-local, evaluated densely from LDS
constexpr int K2_CLASSES = K2_QUEUES + 2;

// TABLE: 0 = incbet's predicates evaluated per row (bdtrc_class); 3 = the same predicates with their one division - the orientation
// threshold aa / (aa + bb), a function of the count alone - read from an LDS table the workgroup fills for counts < K2_TB_COUNTS
// (larger counts divide, as before).  Rounds 2-3 measured a per-count row of all five thresholds in HBM (slower: the dependent
// 64-byte gather cost more than the arithmetic) and the orientation threshold alone from that table (no change): DESIGN.md 4.
// HOIST: all gathers of the four rows up front (rows_prior_fixed).  PACK: slots reserved with two packed DPP prefix sums
// instead of 24 ballots.
constexpr int K2_TB_COUNTS = 128;
template <int NF, int WPE, int TABLE, bool HOIST = false, bool PACK = false>
__global__ __launch_bounds__(K2_THREADS) __attribute__((amdgpu_waves_per_eu(WPE, WPE))) void k2_classify(K2Params P, K2Queues Q) {
    static_assert(!HOIST || NF == 0, "the hoisted gathers are the fixed-size path's");
    static_assert(TABLE == 0 || TABLE == 3, "table variants 1 and 2 were measured and dropped");
    constexpr int ITEMS = K2_CL_ITEMS, WAVES = K2_THREADS / 64, WAVE_ROWS = 64 * ITEMS;
    // Per wave and step: 256 consecutive rows, four per lane (16-byte loads of the three columns).  Every looping row becomes a
    // 16-byte entry of its class queue, in this workgroup's shard: the wave counts its rows per class with ballots, reserves the
    // slots with ONE LDS atomic instruction (lane k adds class k's total to the workgroup's running counter) and writes - no
    // barrier, no global atomic.  The closed-form rows (count == 1: a third of a Hi-C run) are not evaluated where they are met -
    // with a third of the lanes active that costs the wave the full price four times per step - but compacted into the wave's
    // own LDS strip and evaluated with all lanes busy: -expm1(n * log1p(-prior)), ~150 fp64 instructions and few registers.
    // Cephes' other branch (prior >= 0.01: 1 - pow(1 - prior, n); practically never on Hi-C data) would bring pow's ~90 VGPRs
    // into this kernel: those rows are queued for k2_closed instead.
    __shared__ unsigned int cnt[K2_QUEUES + 1];                 // entries of this shard per queued class, so far
    __shared__ double cf_prior[WAVES][WAVE_ROWS];
    __shared__ unsigned short cf_idx[WAVES][WAVE_ROWS];         // row within the wave's 256 | 0x8000 for the inter-chromosomal binomial
    __shared__ unsigned int hist_lds[K2_HIST_BINS];
    __shared__ unsigned int heavy_lds[K2H_BUCKETS];             // this shard's swapped-fraction rows per bucket of the count sort
    __shared__ double tb_lds[TABLE == 3 ? 2 * K2_TB_COUNTS : 1];   // aa / (aa + bb) of counts 0..127: intra binomial, then inter
    if (TABLE == 3) {
        static_assert(2 * K2_TB_COUNTS <= K2_THREADS, "one thread per table entry");
        if (threadIdx.x < 2 * K2_TB_COUNTS) {
            const int c = threadIdx.x & (K2_TB_COUNTS - 1);
            const double n_total = threadIdx.x < K2_TB_COUNTS ? P.intra.n : P.inter.n;
            const double fk = (double)c - 1.0;                   // bdtrc_class's own statements
            const double aa = fk + 1.0, bb = n_total - fk;
            tb_lds[threadIdx.x] = aa / (aa + bb);
        }
    }
    if (threadIdx.x <= K2_QUEUES) cnt[threadIdx.x] = 0;
    const bool count_heavy = Q.heavy_hist != nullptr;
    if (count_heavy)
        for (int d = threadIdx.x; d < K2H_BUCKETS; d += K2_THREADS) heavy_lds[d] = 0;
    FusedHist H;
    H.init(hist_lds, P.top_hist);
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const unsigned long long lane_lt = (1ull << lane) - 1ull;
    const int shard = (int)blockIdx.x;
    const int64_t tiles = (P.n + K2_CL_TILE - 1) / K2_CL_TILE;
    const int4* a4 = reinterpret_cast<const int4*>(P.loc1);
    const int4* b4 = reinterpret_cast<const int4*>(P.loc2);
    const int4* c4 = reinterpret_cast<const int4*>(P.count);
    for (int64_t t = blockIdx.x; t < tiles; t += gridDim.x) {
        const int64_t wave_row0 = t * K2_CL_TILE + (int64_t)wave * WAVE_ROWS;
        const int64_t row0 = wave_row0 + lane * ITEMS;
        int l1_of[ITEMS] = {0, 0, 0, 0}, l2_of[ITEMS] = {0, 0, 0, 0}, count_of[ITEMS] = {0, 0, 0, 0};
        if (row0 < P.n) {                                       // the columns are padded to a multiple of four rows
            const int4 a = a4[row0 >> 2], b = b4[row0 >> 2], c = c4[row0 >> 2];
            l1_of[0] = a.x; l1_of[1] = a.y; l1_of[2] = a.z; l1_of[3] = a.w;
            l2_of[0] = b.x; l2_of[1] = b.y; l2_of[2] = b.z; l2_of[3] = b.w;
            count_of[0] = c.x; count_of[1] = c.y; count_of[2] = c.z; count_of[3] = c.w;
            if (HOIST && row0 + ITEMS > P.n) {                  // the padding rows of the last group hold whatever the allocation held:
#pragma unroll
                for (int r = 1; r < ITEMS; ++r)                 // their (unconditional) gathers must stay inside the tables
                    if (row0 + r >= P.n) l1_of[r] = l2_of[r] = 0;
            }
        }
        int cls_of[ITEMS];
        double prior_of[ITEMS];
        bool inter_of[ITEMS], live_of[ITEMS];
        if (HOIST) rows_prior_fixed<ITEMS>(P, l1_of, l2_of, prior_of, inter_of, live_of);
#pragma unroll
        for (int r = 0; r < ITEMS; ++r) {
            const int64_t i = row0 + r;
            int cls = -1;                                              // -1: no row, 0: done here, 1..4: queued, 5 / 6: closed form
            double prior = 1.0;
            int c = count_of[r];
            if (i < P.n) {
                double pv = 1.0;
                bool is_inter = false;
                cls = 0;
                bool live;
                if (HOIST) {
                    live = live_of[r];
                    prior = prior_of[r];
                    is_inter = inter_of[r];
                } else {
                    live = row_prior<NF>(P, l1_of[r], l2_of[r], prior, is_inter);
                }
                if (live) {
                    const dev::BinomTables& T = is_inter ? P.inter : P.intra;
                    if (TABLE == 3) {
                        double tB;
                        if (c >= 0 && c < K2_TB_COUNTS) {
                            tB = tb_lds[(is_inter ? K2_TB_COUNTS : 0) + c];
                        } else {                                 // a wave without such a count skips the division
                            const double fk = (double)c - 1.0;
                            const double aa = fk + 1.0, bb = T.n - fk;
                            tB = aa / (aa + bb);
                        }
                        cls = dev::bdtrc_class_tb(c, T.n, prior, tB);
                    } else {
                        cls = dev::bdtrc_class(c, T.n, prior);
                    }
                    if (cls == dev::BC_TRIVIAL) {
                        if (dev::bdtrc_is_closed_form(c, T.n, prior))
                            cls = prior < 0.01 ? K2_CLOSED_LOCAL : K2_CLOSED;
                        else
                            pv = dev::bdtrc_count_trivial_open(c, T, prior);  // constants and NaN only
                    }
                    if (is_inter) c = -c;
                }
                if (cls == 0) {
                    P.p[i] = pv;
                    H.add(pv);
                }
            }
            cls_of[r] = cls;
            count_of[r] = c;
            prior_of[r] = prior;
        }
        // slot reservation for the whole wave at once: 24 ballots (4 items x 6 classes), then ONE LDS atomic instruction
        // (lane k reserves class k's total in the workgroup's running counter) and the broadcasts
        unsigned int before_cls[ITEMS];          // rank of this lane's item r among the wave's items of its class
        unsigned int tot[K2_CLASSES] = {0u, 0u, 0u, 0u, 0u, 0u};
        static_assert(K2_CLASSES == 6 && K2_QUEUES == 4, "lane k reserves class k; the sixth class is wave-local");
        if (PACK) {
            // Six counters of 10 bits (a wave holds 256 rows) in two words - classes 1..3 and 4..6 - summed over the lanes by
            // two DPP prefix scans; an item's rank is the field of its class in the lanes' exclusive prefix plus the lane's own
            // earlier items of that class.  Order inside a class is (lane, item) instead of (item, lane): queue order is free.
            unsigned int mine[2] = {0u, 0u};
            unsigned int shift_of[ITEMS];
#pragma unroll
            for (int r = 0; r < ITEMS; ++r) {
                const int k = cls_of[r] - 1;                      // 0..5 counted; -2, -1: no rank
                const int f = k >= 3 ? k - 3 : k;
                shift_of[r] = 10u * (unsigned int)(f < 0 ? 0 : f);
                const unsigned int one = k >= 0 ? (1u << shift_of[r]) : 0u;
                mine[0] += k < 3 ? one : 0u;
                mine[1] += k >= 3 ? one : 0u;
            }
            const unsigned int incl0 = wave_incl_sum_u32(mine[0]), incl1 = wave_incl_sum_u32(mine[1]);
            const unsigned int all0 = (unsigned int)__builtin_amdgcn_readlane((int)incl0, 63), all1 = (unsigned int)__builtin_amdgcn_readlane((int)incl1, 63);
            tot[0] = all0 & 1023u; tot[1] = (all0 >> 10) & 1023u; tot[2] = (all0 >> 20) & 1023u;
            tot[3] = all1 & 1023u; tot[4] = (all1 >> 10) & 1023u; tot[5] = (all1 >> 20) & 1023u;
            unsigned int run[2] = {incl0 - mine[0], incl1 - mine[1]};
#pragma unroll
            for (int r = 0; r < ITEMS; ++r) {
                const int k = cls_of[r] - 1;
                const unsigned int word = k >= 3 ? run[1] : run[0];
                before_cls[r] = (word >> shift_of[r]) & 1023u;
                const unsigned int one = k >= 0 ? (1u << shift_of[r]) : 0u;
                run[0] += k < 3 ? one : 0u;
                run[1] += k >= 3 ? one : 0u;
            }
        } else {
#pragma unroll
            for (int r = 0; r < ITEMS; ++r) {
                before_cls[r] = 0;
#pragma unroll
                for (int k = 1; k <= K2_CLASSES; ++k) {
                    const unsigned long long m = __ballot(cls_of[r] == k);
                    if (cls_of[r] == k) before_cls[r] = tot[k - 1] + (unsigned int)__popcll(m & lane_lt);
                    tot[k - 1] += (unsigned int)__popcll(m);
                }
            }
        }
        const unsigned int my_tot = lane == 0 ? tot[0] : (lane == 1 ? tot[1] : (lane == 2 ? tot[2] : (lane == 3 ? tot[3] : tot[4])));
        unsigned int my_base = 0;
        if (lane <= K2_QUEUES && my_tot) my_base = atomicAdd(&cnt[lane], my_tot);
        unsigned int wave_base[K2_QUEUES + 1];
#pragma unroll
        for (int k = 0; k <= K2_QUEUES; ++k) wave_base[k] = __shfl(my_base, k, 64);
#pragma unroll
        for (int r = 0; r < ITEMS; ++r) {
            const int k = cls_of[r] - 1;
            if (k >= 0 && k <= K2_QUEUES) {
                const unsigned int wb = k == 0 ? wave_base[0] : (k == 1 ? wave_base[1] : (k == 2 ? wave_base[2] : (k == 3 ? wave_base[3] : wave_base[4])));
                QEntry e;
                e.row = (unsigned int)(row0 + r);
                e.count = count_of[r];
                e.prior = prior_of[r];
                *qentry(Q.q[k], shard, (long long)(wb + before_cls[r])) = e;
                if (k == dev::BC_CF_SWAPPED - 1 && count_heavy) atomicAdd(&heavy_lds[k2h_bucket(e.count)], 1u);
            } else if (cls_of[r] == K2_CLOSED_LOCAL) {
                cf_prior[wave][before_cls[r]] = prior_of[r];
                cf_idx[wave][before_cls[r]] = (unsigned short)((lane * ITEMS + r) | (count_of[r] < 0 ? 0x8000 : 0));
            }
        }
        // the small-prior closed-form rows of this wave, all lanes busy.  The strip is the wave's own: LDS operations of one
        // wave complete in order, so its reads below see its writes above without a barrier (the fence keeps the compiler from
        // moving them)
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        const unsigned int n_local = tot[K2_CLASSES - 1];
        for (unsigned int j = lane; j < n_local; j += 64) {
            const unsigned int ix = cf_idx[wave][j];
            const double n_total = (ix & 0x8000u) ? P.inter.n : P.intra.n;
            const double pv = -dev::cephes_expm1(n_total * dev::cephes_log1p(-cf_prior[wave][j]));        // bdtrc_closed_form, prior < 0.01
            P.p[wave_row0 + (ix & 0x7FFFu)] = pv;
            H.add(pv);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");       // the strip is rewritten in the next step
        __builtin_amdgcn_wave_barrier();
    }
    __syncthreads();
    if (threadIdx.x <= K2_QUEUES) Q.count[(size_t)threadIdx.x * K2_MAX_SHARDS + shard] = cnt[threadIdx.x];
    if (count_heavy)                    // a handful of counts are met in a shard: only those words of the matrix are touched
        for (int d = threadIdx.x; d < K2H_BUCKETS; d += K2_THREADS)
            if (heavy_lds[d]) atomicAdd(&Q.heavy_hist[(size_t)d * K2H_BLOCKS + (shard & (K2H_BLOCKS - 1))], heavy_lds[d]);
    H.flush(P.top_hist);
}

// count == 1: p = 1 - (1 - prior)^n through Cephes' log1p / expm1 (or pow): bdtrc_closed_form
__global__ __launch_bounds__(K2_THREADS) void k2_closed(K2Params P, QSpan q) {
    __shared__ unsigned int hist_lds[K2_HIST_BINS];
    FusedHist H;
    H.init(hist_lds, P.top_hist);
    for (int sh = blockIdx.x; sh < q.n_shards; sh += gridDim.x) {
        const long long n = (long long)q.count[sh];
        for (long long j = threadIdx.x; j < n; j += blockDim.x) {
            const QEntry e = *qentry(q, sh, j);
            const double pv = dev::bdtrc_closed_form(e.count < 0 ? P.inter.n : P.intra.n, e.prior);
            store_p<false>(P.p + e.row, pv);
            H.add(pv);
        }
    }
    H.flush(P.top_hist);
}

template <int CLS, bool SMALL_N>
__global__ __launch_bounds__(K2_THREADS) void k2_queue(K2Params P, QSpan q) {
    __shared__ unsigned int hist_lds[K2_HIST_BINS];
    FusedHist H;
    H.init(hist_lds, P.top_hist);
    for (int sh = blockIdx.x; sh < q.n_shards; sh += gridDim.x) {
        const long long n = (long long)q.count[sh];
        for (long long j = threadIdx.x; j < n; j += blockDim.x) {
            const QEntry e = *qentry(q, sh, j);
            const bool is_inter = e.count < 0;
            const int c = is_inter ? -e.count : e.count;
            const double pv = dev::bdtrc_count_class<CLS, SMALL_N>(c, is_inter ? P.inter : P.intra, e.prior);
            store_p<false>(P.p + e.row, pv);
            H.add(pv);
        }
    }
    H.flush(P.top_hist);
}

// The two converging continued-fraction classes need ~5..17 iterations, growing with the contact count: in queue order a
// wave waits for its slowest lane (measured: mean 9.3 iterations, mean of the per-wave maximum 20.3).  Each workgroup
// therefore takes a tile of 1024 entries, counting-sorts it by min(count, 31) in LDS (one LDS atomic per entry) and hands
// every wave 64 neighbours of that order (per-wave maximum 11.4).  Results go to P.p[row], so the order is free.
constexpr int K2_SORT_TILE = 1024;
constexpr int K2_SORT_BUCKETS = 32;
template <int CLS, bool SMALL_N, int WPE>
__global__ __launch_bounds__(K2_THREADS) __attribute__((amdgpu_waves_per_eu(WPE))) void k2_queue_by_count(K2Params P, QSpan q) {
    static_assert(K2_SORT_TILE == 4 * K2_THREADS, "four entries per thread");
    __shared__ QEntry tile[K2_SORT_TILE];
    __shared__ unsigned int bucket_cnt[K2_SORT_BUCKETS], bucket_off[K2_SORT_BUCKETS];
    __shared__ unsigned int hist_lds[K2_HIST_BINS];
    FusedHist H;
    H.init(hist_lds, P.top_hist);
    for (int sh = blockIdx.x; sh < q.n_shards; sh += gridDim.x) {
      const int64_t n = (int64_t)q.count[sh];
      const int64_t tiles = (n + K2_SORT_TILE - 1) / K2_SORT_TILE;
      for (int64_t t = 0; t < tiles; ++t) {
        if (threadIdx.x < K2_SORT_BUCKETS) bucket_cnt[threadIdx.x] = 0;
        __syncthreads();
        QEntry e[4];
        int bucket[4];
        unsigned int slot[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int64_t j = t * K2_SORT_TILE + r * K2_THREADS + threadIdx.x;
            bucket[r] = -1;
            if (j < n) {
                e[r] = *qentry(q, sh, j);
                const int c = e[r].count < 0 ? -e[r].count : e[r].count;
                bucket[r] = c < K2_SORT_BUCKETS - 1 ? c : K2_SORT_BUCKETS - 1;
                slot[r] = atomicAdd(&bucket_cnt[bucket[r]], 1u);
            }
        }
        __syncthreads();
        if (threadIdx.x < K2_SORT_BUCKETS) {
            unsigned int off = 0;
            for (int b = 0; b < (int)threadIdx.x; ++b) off += bucket_cnt[b];
            bucket_off[threadIdx.x] = off;
        }
        __syncthreads();
#pragma unroll
        for (int r = 0; r < 4; ++r)
            if (bucket[r] >= 0) tile[bucket_off[bucket[r]] + slot[r]] = e[r];
        __syncthreads();
        const int m = (int)min((int64_t)K2_SORT_TILE, n - t * K2_SORT_TILE);
#pragma unroll 1
        for (int r = 0; r < 4; ++r) {
            const int idx = r * K2_THREADS + threadIdx.x;
            if (idx < m) {
                const QEntry x = tile[idx];
                const bool is_inter = x.count < 0;
                const int c = is_inter ? -x.count : x.count;
                const double pv = dev::bdtrc_count_class<CLS, SMALL_N>(c, is_inter ? P.inter : P.intra, x.prior);
                store_p<false>(P.p + x.row, pv);
                H.add(pv);
            }
        }
        __syncthreads();
      }
    }
    H.flush(P.top_hist);
}

// ---- the 300-iteration class in count-homogeneous waves -----------------------------------------------------------
// The swapped-continued-fraction queue is counting-sorted by (binomial, contact count) so that every wave of k2h_heavy
// holds 64 rows of ONE count: all per-iteration constants of Cephes' loop then come from a table row per iteration
// through scalar loads (cf_swapped_uniform, fhx_bdtrc.hpp).  Bucket = count for intra rows, K2H_KCAP + count for
// rows of the inter-chromosomal binomial, one last bucket for counts >= K2H_KCAP (evaluated by the per-lane k2_queue
// kernel).  Every bucket starts at a multiple of 64 entries in the sorted queue, so a wave never straddles two counts.
constexpr int K2H_KCAP = 1023;
constexpr int K2H_GENERIC = 2 * K2H_KCAP;               // 2046
constexpr int K2H_THREADS = 256;

__device__ __forceinline__ int k2h_bucket(int signed_count) {
    const bool inter = signed_count < 0;
    const int c = inter ? -signed_count : signed_count;
    return c < K2H_KCAP ? (inter ? K2H_KCAP + c : c) : K2H_GENERIC;
}

// per-workgroup bucket counts of its shards of the queue (digit-major matrix, as rs_count writes it): workgroup b takes the
// shards b, b + K2H_BLOCKS, ...
__global__ __launch_bounds__(K2H_THREADS) void k2h_count(QSpan q, unsigned int* __restrict__ block_hist) {
    __shared__ unsigned int h[K2H_BUCKETS];
    for (int d = threadIdx.x; d < K2H_BUCKETS; d += K2H_THREADS) h[d] = 0;
    __syncthreads();
    for (int sh = blockIdx.x; sh < q.n_shards; sh += K2H_BLOCKS) {
        const long long n = (long long)q.count[sh];
        for (long long i = threadIdx.x; i < n; i += K2H_THREADS) atomicAdd(&h[k2h_bucket(qentry(q, sh, i)->count)], 1u);
    }
    __syncthreads();
    for (int d = threadIdx.x; d < K2H_BUCKETS; d += K2H_THREADS) block_hist[(size_t)d * K2H_BLOCKS + blockIdx.x] = h[d];
}

// bucket starts, each rounded up to a multiple of `granule` entries (64 x the rows a lane of k2h_heavy takes): off[b] for
// b = 0..K2H_BUCKETS (the last one = padded total)
__global__ __launch_bounds__(1024) void k2h_offsets(const unsigned int* __restrict__ digit_total, unsigned int* __restrict__ off,
                                                    unsigned int granule) {
    __shared__ unsigned int part[1024];
    const unsigned int g1 = granule - 1u;
    const unsigned int a = (digit_total[2 * threadIdx.x] + g1) / granule * granule, b = (digit_total[2 * threadIdx.x + 1] + g1) / granule * granule;
    part[threadIdx.x] = a + b;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned int acc = 0;
        for (int i = 0; i < 1024; ++i) {
            const unsigned int c = part[i];
            part[i] = acc;
            acc += c;
        }
        off[K2H_BUCKETS] = acc;
    }
    __syncthreads();
    off[2 * threadIdx.x] = part[threadIdx.x];
    off[2 * threadIdx.x + 1] = part[threadIdx.x] + a;
}

__global__ __launch_bounds__(K2H_THREADS) void k2h_scatter(QSpan q, const unsigned int* __restrict__ block_hist,
                                                           const unsigned int* __restrict__ off, QEntry* __restrict__ out) {
    __shared__ unsigned int cursor[K2H_BUCKETS];
    for (int d = threadIdx.x; d < K2H_BUCKETS; d += K2H_THREADS) cursor[d] = off[d] + block_hist[(size_t)d * K2H_BLOCKS + blockIdx.x];
    __syncthreads();
    for (int sh = blockIdx.x; sh < q.n_shards; sh += K2H_BLOCKS) {
        const long long n = (long long)q.count[sh];
        for (long long i = threadIdx.x; i < n; i += K2H_THREADS) {
            const QEntry e = *qentry(q, sh, i);
            out[atomicAdd(&cursor[k2h_bucket(e.count)], 1u)] = e;       // order inside a bucket is free: results go to p[row]
        }
    }
}

// one workgroup per non-empty (binomial, count) bucket, one thread per iteration: the 300 rows of iteration constants
constexpr int K2H_TABLE_THREADS = 320;
static_assert(K2H_TABLE_THREADS >= dev::kCfIters, "one thread per table row");
__global__ __launch_bounds__(K2H_TABLE_THREADS) void k2h_tables(const unsigned int* __restrict__ digit_total, double n_intra, double n_inter,
                                                                dev::CfRow* __restrict__ tab) {
    const int b = blockIdx.x;
    if (b >= K2H_GENERIC || digit_total[b] == 0 || (int)threadIdx.x >= dev::kCfIters) return;
    const bool inter = b >= K2H_KCAP;
    tab[(size_t)b * dev::kCfIters + threadIdx.x] = dev::cf_swapped_row(inter ? n_inter : n_intra, inter ? b - K2H_KCAP : b, (int)threadIdx.x);
}

// Lanes cf_swapped_uniform cannot take (unusual inputs or states, see fhx_bdtrc.hpp) are appended to `redo` - the space the
// unsorted queue occupied, free once k2h_scatter has run - and k2h_generic evaluates them with the per-lane loop; keeping that
// loop out of this kernel keeps it at 8 waves per SIMD (38 VGPRs instead of 102).
struct K2HeavyParams {            // the few fields of K2Params this kernel reads: its SGPR count decides how many waves a CU admits
    dev::BinomTables intra, inter;
    double* p;
    unsigned long long* top_hist;
};

// R rows per lane (a task = 64 R consecutive entries of one bucket: every bucket starts at a multiple of that), WPE waves per
// SIMD: see cf_swapped_uniform for why more rows per lane beat more waves.
constexpr int K2H_MAX_ROWS = 4;
template <int R, int WPE, bool SMALL_N>
__global__ __launch_bounds__(K2H_THREADS) __attribute__((amdgpu_waves_per_eu(WPE, WPE))) void k2h_heavy(
    K2HeavyParams P, const QEntry* __restrict__ sorted, const unsigned int* __restrict__ off,
    const unsigned int* __restrict__ digit_total, const dev::CfRow* __restrict__ tab, QEntry* __restrict__ redo,
    unsigned long long* __restrict__ n_redo) {
    static_assert(R >= 1 && R <= K2H_MAX_ROWS, "the sorted queue is padded for at most K2H_MAX_ROWS rows per lane");
    __shared__ unsigned int hist_lds[K2_HIST_BINS];
    FusedHist H;
    H.init(hist_lds, P.top_hist);
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int lane = threadIdx.x & 63;
    constexpr unsigned int TASK = 64u * R;
    const unsigned int n_tasks = off[K2H_GENERIC] / TASK;        // tasks in front of the generic bucket
    const unsigned int stride = gridDim.x * (K2H_THREADS / 64);
    for (unsigned int task = blockIdx.x * (K2H_THREADS / 64) + wave; task < n_tasks; task += stride) {
        const unsigned int first = task * TASK;
        // bucket of this task: the last b with off[b] <= first (empty buckets share their successor's start: skip them)
        int lo = 0, hi = K2H_GENERIC;                             // invariant: off[lo] <= first < off[hi]
        while (hi - lo > 1) {
            const int mid = (lo + hi) >> 1;
            if (off[mid] <= first)
                lo = mid;
            else
                hi = mid;
        }
        const int b = __builtin_amdgcn_readfirstlane(lo);
        const unsigned int live = off[b] + digit_total[b];        // entries of the bucket end here, padding follows
        const bool is_inter = b >= K2H_KCAP;
        const int c = is_inter ? b - K2H_KCAP : b;
        const dev::BinomTables& T = is_inter ? P.inter : P.intra;
        // bdtrc_count_class<BC_CF_SWAPPED>: incbet_finish(bb, aa, 1 - xx, xx, incbcf(bb, aa, 1 - xx), flag = 1, ...)
        const double fk = (double)c - 1.0;
        const double aa = fk + 1.0, bb = T.n - fk;
        QEntry e[R];
        bool have[R], irregular[R];
        double w1[R], cf[R];
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const unsigned int j = first + (unsigned int)r * 64u + (unsigned int)lane;
            have[r] = j < live;
            e[r].row = 0u;
            e[r].count = is_inter ? -c : c;
            e[r].prior = 0.5;
            if (have[r]) e[r] = sorted[j];
            w1[r] = 1.0 - e[r].prior;
            irregular[r] = !have[r] || !dev::cf_swapped_regular(bb, aa, w1[r]);
        }
        const dev::CfRowConstPtr rows = (dev::CfRowConstPtr)(uintptr_t)(tab + (size_t)b * dev::kCfIters);
        dev::cf_swapped_uniform<R>(rows, w1, irregular, cf);                  // every lane of the wave takes part
#pragma unroll
        for (int r = 0; r < R; ++r) {
            double pv = 0.0;
            const bool mine = have[r] && !irregular[r];
            if (have[r]) {
                if (__builtin_expect(irregular[r], 0))
                    redo[atomicAdd(n_redo, 1ull)] = e[r];
                else {
                    pv = dev::incbet_finish<SMALL_N>(bb, aa, w1[r], e[r].prior, cf[r], 1, T.lbeta[c], (SMALL_N && T.small_n) ? T.inv_beta[c] : 0.0);
                    store_p<true>(P.p + e[r].row, pv);
                }
            }
            H.add_wave_min(pv, mine);
        }
    }
    H.flush(P.top_hist);
}

// counts >= K2H_KCAP (the last bucket) and the rows k2h_heavy handed back: per-lane evaluation, the k2_queue<BC_CF_SWAPPED> body
__global__ __launch_bounds__(K2_THREADS) void k2h_generic(K2Params P, const QEntry* __restrict__ sorted,
                                                          const unsigned int* __restrict__ off,
                                                          const unsigned int* __restrict__ digit_total,
                                                          const QEntry* __restrict__ redo,
                                                          const unsigned long long* __restrict__ n_redo) {
    const QEntry* base = sorted + off[K2H_GENERIC];
    const int64_t n_generic = (int64_t)digit_total[K2H_GENERIC];
    const int64_t n = n_generic + (int64_t)*n_redo;
    __shared__ unsigned int hist_lds[K2_HIST_BINS];
    FusedHist H;
    H.init(hist_lds, P.top_hist);
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; j < n; j += stride) {
        const QEntry e = j < n_generic ? base[j] : redo[j - n_generic];
        const bool is_inter = e.count < 0;
        const int c = is_inter ? -e.count : e.count;
        const double pv = dev::bdtrc_count_class<dev::BC_CF_SWAPPED>(c, is_inter ? P.inter : P.intra, e.prior);
        store_p<true>(P.p + e.row, pv);
        H.add(pv);
    }
    H.flush(P.top_hist);
}

// (Measured and dropped in round 4, profiles/r04_h_cfu_ab.txt: the two CONVERGING classes through the heavy class's machinery -
// counting-sorted by (binomial, orientation, count), iteration constants from a table row per iteration through scalar loads,
// four rows per lane, the loop stopped in blocks of eight iterations.  Bit-identical (same digest of all p and q), and the
// loop kernels were 15 % (incbcf: 1.12 -> 0.95 ms) and 2 % (incbd: 0.96 -> 0.94 ms) faster than k2_queue_by_count - a wave of 256
// rows runs until its slowest row converges, 16-32 iterations where a row needs 9 on average, and the transcendental epilogue is
// the same - but the count sort in front of them (count 0.10-0.11 ms, scatter 0.14-0.18 ms, tables, offsets per class) costs more
// than that: 12.2-12.4 ms per pass against 11.9.)

// test hook: class of (count, prior) by the table and by bdtrc_class's arithmetic, and the five thresholds of the count
__global__ void k_debug_classify(double n_total, const int32_t* __restrict__ count, const double* __restrict__ prior, int64_t n,
                                 int32_t* __restrict__ by_table, int32_t* __restrict__ by_arith, double* __restrict__ thr5) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const dev::ClsRow r = dev::cls_row(n_total, count[i]);
    by_table[i] = dev::cls_is_trivial(count[i], n_total, prior[i]) ? (int)dev::BC_TRIVIAL : dev::cls_lookup(r, prior[i]);
    by_arith[i] = dev::bdtrc_class(count[i], n_total, prior[i]);
    if (thr5) {
        thr5[5 * i] = r.tA;
        thr5[5 * i + 1] = r.tB;
        thr5[5 * i + 2] = r.tC;
        thr5[5 * i + 3] = r.tD;
        thr5[5 * i + 4] = r.tE;
    }
}

// outlier flags (p < 1/N, fithic.py:1215) are derived from p when somebody asks: K2 never writes per-row bytes
__global__ void k_outlier_flags(const double* __restrict__ p, double thres, int64_t n, uint8_t* __restrict__ flags) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) flags[i] = (p[i] < thres) ? 1 : 0;
}

// ordered compaction of the set flag bytes: counts per tile of 1024 rows (thread t takes rows 4t..4t+3), then the row numbers
__global__ __launch_bounds__(fhxscan::THREADS) void k_flag_count(const unsigned char* __restrict__ flag, int64_t n, unsigned int* __restrict__ tile_counts) {
    const int64_t base = (int64_t)blockIdx.x * fhxscan::TILE + (int64_t)threadIdx.x * fhxscan::SCAN_ITEMS;
    unsigned int c = 0;
    for (int k = 0; k < fhxscan::SCAN_ITEMS; ++k)
        if (base + k < n && flag[base + k]) ++c;
    unsigned int total;
    fhxscan::block_exclusive_scan(c, &total);
    if (threadIdx.x == 0) tile_counts[blockIdx.x] = total;
}
__global__ __launch_bounds__(fhxscan::THREADS) void k_flag_rows(const unsigned char* __restrict__ flag, int64_t n,
                                                                 const unsigned long long* __restrict__ tile_offsets, int64_t* __restrict__ rows) {
    const int64_t base = (int64_t)blockIdx.x * fhxscan::TILE + (int64_t)threadIdx.x * fhxscan::SCAN_ITEMS;
    unsigned int c = 0;
    for (int k = 0; k < fhxscan::SCAN_ITEMS; ++k)
        if (base + k < n && flag[base + k]) ++c;
    unsigned int total;
    unsigned long long at = tile_offsets[blockIdx.x] + fhxscan::block_exclusive_scan(c, &total);
    for (int k = 0; k < fhxscan::SCAN_ITEMS; ++k)
        if (base + k < n && flag[base + k]) rows[at++] = base + k;
}

__global__ void k_bdtrc_array(dev::BinomTables T, const int32_t* __restrict__ count, const double* __restrict__ prior,
                              int64_t n, double* __restrict__ out) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
        out[i] = dev::bdtrc_count(count[i], T, prior[i]);
}

template <int KIND, int LAZY>
__global__ void k_debug_contfrac(const double* __restrict__ a, const double* __restrict__ b, const double* __restrict__ x,
                                 int64_t n, double* __restrict__ out) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
        out[i] = LAZY ? dev::contfrac_lazy<KIND>(a[i], b[i], x[i]) : dev::contfrac<KIND>(a[i], b[i], x[i]);
}

__global__ void k_debug_lean_div(const double* __restrict__ n, const double* __restrict__ d, int64_t len, double* __restrict__ out) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < len; i += stride) out[i] = dev::lean_div(n[i], d[i]);
}

// ---- no bias file: p depends on (distance index, count) only ------------------------------------------------------
// Without a bias table every in-range intra row has prior = prior_lut[d] exactly (b1 = b2 = 1.0, fithic.py:1069) and every
// inter row prior = interChrProb, so bdtrc is a function of (d, count) / of count.  K2 then runs on a TABLE of virtual rows
// - one per (d, count), count <= cap, plus one per count for inter rows - through the same classify + queue kernels, and the
// real rows gather.  Same function of the same inputs: bit-identical to evaluating every row (a few hundred thousand
// evaluations instead of one per contact pair).  Rows whose count exceeds the table evaluate in place.
__global__ __launch_bounds__(256) void k2_memo_rows(int n_d, int lo_idx, int cap, int with_inter, int32_t* __restrict__ loc1,
                                                   int32_t* __restrict__ loc2, int32_t* __restrict__ count, int64_t n_v) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    const int64_t intra = (int64_t)n_d * (cap + 1);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_v; i += stride) {
        if (i < intra) {
            const int d = lo_idx + (int)(i / (cap + 1));
            loc1[i] = 0;
            loc2[i] = d;                                    // |slot 0 - slot d| = d; both slots carry bias 1.0
            count[i] = (int)(i % (cap + 1));
        } else {
            loc1[i] = 0;
            loc2[i] = ~0;                                   // bit 31: inter-chromosomal
            count[i] = (int)(i - intra);
        }
        (void)with_inter;
    }
}

__global__ __launch_bounds__(256) void k2_memo_gather(K2Params P, const double* __restrict__ table, int n_d, int cap, int has_intra,
                                                     int has_inter, unsigned int* __restrict__ overflow_rows,
                                                     unsigned long long* __restrict__ n_overflow, unsigned long long overflow_cap) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    const int64_t intra = has_intra ? (int64_t)n_d * (cap + 1) : 0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < P.n; i += stride) {
        const int l1 = P.loc1[i], l2 = P.loc2[i];
        const int c = P.count[i];
        double prior = 1.0, pv = 1.0;
        bool is_inter = false;
        if (row_prior(P, l1, l2, prior, is_inter)) {
            if (c >= 0 && c <= cap && (is_inter ? has_inter : has_intra)) {
                pv = is_inter ? table[intra + c] : table[(int64_t)(abs(l1 - l2) - P.lo_idx) * (cap + 1) + c];
            } else {                                        // beyond the table (rare): k2_memo_overflow evaluates these rows
                const unsigned long long at = atomicAdd(n_overflow, 1ull);
                if (at < overflow_cap) overflow_rows[at] = (unsigned int)i;
                pv = -1.0;                                  // never a p-value: marks the row if the list overflowed
            }
        }
        P.p[i] = pv;
    }
}

// rows beyond the table: evaluate in place.  from_list = 0: the list overflowed, scan for the -1 marks instead.
__global__ __launch_bounds__(256) void k2_memo_overflow(K2Params P, const unsigned int* __restrict__ rows, int64_t n_list, int from_list) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    const int64_t n = from_list ? n_list : P.n;
    for (int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; j < n; j += stride) {
        const int64_t i = from_list ? (int64_t)rows[j] : j;
        if (!from_list && !(P.p[i] == -1.0)) continue;
        double prior = 1.0;
        bool is_inter = false;
        if (row_prior(P, P.loc1[i], P.loc2[i], prior, is_inter)) P.p[i] = dev::bdtrc_count(P.count[i], is_inter ? P.inter : P.intra, prior);
    }
}

// expected contact count and the two biases, recomputed on demand for the writer (fithic.py:1075-1078, :1105-1108)
__global__ void k2_extras(K2Params P, double bias_low, double bias_up, double* __restrict__ expcc,
                          double* __restrict__ ob1, double* __restrict__ ob2) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < P.n; i += stride) {
        const int l1 = P.loc1[i], l2 = P.loc2[i];
        const int s2 = l2 < 0 ? ~l2 : l2;
        const double b1 = P.slot_bias[l1], b2 = P.slot_bias[s2];
        double prior = 1.0, e = 0.0;
        bool is_inter = false;
        if (row_prior(P, l1, l2, prior, is_inter)) {
            const bool within = b1 >= bias_low && b1 <= bias_up && b2 >= bias_low && b2 <= bias_up;
            if (within) e = (is_inter ? P.inter.n : P.intra.n) * prior;
        }
        if (expcc) expcc[i] = e;
        if (ob1) ob1[i] = b1;
        if (ob2) ob2[i] = b2;
    }
}

// outlier bookkeeping for the next pass: skip mask |= outlier, and the multiset of outlier distances
__global__ void k_fold_outliers(const int32_t* __restrict__ loc1, const int32_t* __restrict__ loc2,
                                const double* __restrict__ pvals, double thres, uint8_t* __restrict__ skip,
                                uint8_t* __restrict__ seen_twice, int64_t n, int res, int n_dist,
                                const int16_t* __restrict__ slot_chr, const ChrGrid* __restrict__ grid,
                                unsigned long long* __restrict__ out_hist, unsigned long long* __restrict__ n_out,
                                unsigned long long* __restrict__ first_dup, const long long* __restrict__ grow) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    unsigned long long mine = 0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        if (!(pvals[i] < thres)) continue;             // NaN is not an outlier (p_val < outlierThres is False)
        ++mine;
        if (skip[i]) {                             // duplicated line number in the reference's SortedList (A17)
            seen_twice[i] = 1;
            atomicMin(first_dup, (unsigned long long)(grow ? grow[i] : i));
        }
        skip[i] = 1;
        const int l1 = loc1[i], l2 = loc2[i];
        long long idx;
        if (l2 >= 0) {
            idx = abs(l1 - l2);
        } else {                                    // inter row: the reference still records abs(mid1 - mid2)
            const int s2 = ~l2;
            const ChrGrid g1 = grid[slot_chr[l1]], g2 = grid[slot_chr[s2]];
            const long long m1 = (long long)(l1 - g1.base) * res + g1.off;
            const long long m2 = (long long)(s2 - g2.base) * res + g2.off;
            const long long d = m1 > m2 ? m1 - m2 : m2 - m1;
            idx = (d + res - 1) / res;              // bins end on grid distances: rounding up keeps the bin
        }
        if (idx > n_dist - 1) idx = n_dist - 1;
        atomicAdd(&out_hist[idx], 1ull);
    }
    mine = (unsigned long long)wave_sum_i64((long long)mine);
    if ((threadIdx.x & 63) == 0 && mine) atomicAdd(n_out, mine);
}

// ===================================================================================================
// K3: Benjamini-Hochberg as the reference defines it
// ===================================================================================================
constexpr int SORT_THREADS = 256;
constexpr int SORT_WAVES = SORT_THREADS / 64;
constexpr int SORT_ITEMS = 16;                        // per thread
constexpr int SORT_TILE = SORT_THREADS * SORT_ITEMS;  // 4096 keys per tile
constexpr int SORT_BLOCKS = 1024;                     // persistent: 4 workgroups per CU
constexpr int RADIX_BITS = 11;                       // 6 passes cover 66 >= 64 key bits
constexpr int RADIX = 1 << RADIX_BITS;
constexpr int SORT_PASSES = 6;                        // even: the result lands in the buffer pair it started in

// ---- early cutoff -----------------------------------------------------------------------------------------------
// The reference's monotonisation is a FORWARD running max of min(p*N/rank, 1) over ascending p (fithic/myStats.py:31-46):
// once one element reaches 1 every later element has q = 1.  An element with p >= t whose rank is at most C certainly has
// fl(fl(p*N)/rank) >= fl(fl(t*N)/C) (rounding is monotone), so from a coarse histogram of the keys (top 14 bits: sign,
// exponent, 2 mantissa bits) we can name a key T* such that every element >= T* has q = 1 exactly - those rows are not
// sorted at all.  On Hi-C data N (possible pairs) exceeds the number of observed rows, so only the enriched small-p tail
// (typically 10-20 % of the rows) survives the cutoff.  The result is bit-identical to sorting everything.
constexpr int TOP_SHIFT = 50;
constexpr int TOP_BINS = 8192;                         // keys of p < 1 are < 2^62, so key >> 50 < 4096 (kept at 8192 for slack)

__device__ __forceinline__ unsigned long long pvalue_key(double v) {
    unsigned long long bits = (unsigned long long)__double_as_longlong(v);
    if (bits == 0x8000000000000000ull) bits = 0ull;    // -0.0 sorts with +0.0
    return bits;
}

constexpr unsigned long long KEY_ONE = 0x3FF0000000000000ull;          // bits of 1.0
constexpr unsigned long long KEY_KEEP_ALL = 0x7FF0000000000001ull;     // above +inf: no value is cut

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

__global__ __launch_bounds__(1024) void k3_cutoff(const unsigned long long* __restrict__ hist, double n_tests,
                                                  unsigned long long* __restrict__ cutoff_key) {
    __shared__ unsigned long long part[1024];
    __shared__ unsigned int best;
    constexpr int PER = TOP_BINS / 1024;
    unsigned long long local[PER];
    unsigned long long sum = 0;
    for (int k = 0; k < PER; ++k) {
        local[k] = hist[threadIdx.x * PER + k];
        sum += local[k];
    }
    part[threadIdx.x] = sum;
    if (threadIdx.x == 0) best = TOP_BINS;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned long long acc = 0;
        for (int i = 0; i < 1024; ++i) {
            const unsigned long long c = part[i];
            part[i] = acc;
            acc += c;
        }
    }
    __syncthreads();
    unsigned long long cum = part[threadIdx.x];
    for (int k = 0; k < PER; ++k) {
        cum += local[k];
        const int b = threadIdx.x * PER + k;
        if (local[k] && bin_saturates(b, cum, n_tests)) atomicMin(&best, (unsigned int)b);
    }
    __syncthreads();
    if (threadIdx.x == 0)
        *cutoff_key = (best < (unsigned int)TOP_BINS) ? ((unsigned long long)best << TOP_SHIFT) : KEY_KEEP_ALL;
}

// compaction: keys of the rows below the cutoff key (IEEE bit pattern: all p are >= 0, so unsigned order is
// numeric order); rows at or above it get q = 1 and NaN rows get q = NaN right here.
__global__ __launch_bounds__(SORT_THREADS) void k3_compact(const double* __restrict__ p, int64_t n,
                                                           unsigned long long* __restrict__ keys,
                                                           unsigned int* __restrict__ vals, double* __restrict__ q,
                                                           unsigned long long* __restrict__ counter,
                                                           const unsigned long long* __restrict__ cutoff_key) {
    // rows at or above the cutoff key (see above) have q = 1 and are not sorted; NaN rows get q = NaN.
    // one global atomic per 4096-row tile (a same-address atomic per 256 rows capped this kernel at ~88 M atomics/s)
    __shared__ unsigned int wave_cnt[SORT_WAVES];
    __shared__ unsigned long long block_base;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const unsigned long long lane_lt = (1ull << lane) - 1ull;
    const int64_t tiles = (n + SORT_TILE - 1) / SORT_TILE;
    const unsigned long long cutoff = *cutoff_key;
    const double2* p2 = reinterpret_cast<const double2*>(p);
    double2* q2 = reinterpret_cast<double2*>(q);
    for (int64_t t = blockIdx.x; t < tiles; t += gridDim.x) {
        const int64_t wave_base = t * SORT_TILE + (int64_t)wave * (64 * SORT_ITEMS);
        // two consecutive rows per lane and step: 16-byte loads of p and (for the rows that are not ranked: nearly all) 16-byte
        // stores of q
        double v[SORT_ITEMS];
        unsigned int before[SORT_ITEMS];
        unsigned long long keepmask = 0;          // bit r: this lane keeps item r
        unsigned int run = 0;
#pragma unroll
        for (int h = 0; h < SORT_ITEMS / 2; ++h) {
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
                q2[i >> 1] = make_double2((w.x == w.x) ? 1.0 : w.x, (w.y == w.y) ? 1.0 : w.y);
            } else if (i < n) {                                            // the last row of an odd count
                v[2 * h] = p[i];
                keep0 = (v[2 * h] == v[2 * h]) && (pvalue_key(v[2 * h]) < cutoff);
                if (!keep0) q[i] = (v[2 * h] == v[2 * h]) ? 1.0 : v[2 * h];
            }
            const unsigned long long m0 = __ballot(keep0), m1 = __ballot(keep1);
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
                keys[base + before[r]] = pvalue_key(v[r]);
                vals[base + before[r]] = (unsigned int)(wave_base + (int64_t)((r >> 1) * 64 + lane) * 2 + (r & 1));
            }
        }
        __syncthreads();
    }
}

// per-workgroup digit counts for one radix pass; workgroup b owns the contiguous chunk [b*chunk, (b+1)*chunk)
__global__ __launch_bounds__(SORT_THREADS) void rs_count(const unsigned long long* __restrict__ keys,
                                                         const unsigned long long* __restrict__ n_ptr, int shift,
                                                         unsigned int* __restrict__ block_hist) {
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
// profiles/r04_b_od1_kernel_stats.txt).  Now 512 threads per tile of 4096 keys and the per-wave counters share their 32 KB with
// the staged keys (a key's slot is in a register by the time the counters die): 64 KB, two workgroups = 16 waves per CU.
// (Measured and dropped, profiles/r04_e_rs_ab.txt: no staging at all - keys written straight from registers to
// global_base[digit] + rank - is 45 % slower, the tile-wide reordering is what coalesces the writes of the passes over the
// exponent bits; squeezing the kernel to 80 VGPRs for a third workgroup per CU spills and is slower still.)
constexpr int SCAT_ITEMS = 8;                                   // keys per thread; a tile = SCAT_THREADS x 8 keys

template <int SCAT_THREADS, int WPE>
__global__ __launch_bounds__(SCAT_THREADS) __attribute__((amdgpu_waves_per_eu(WPE, WPE))) void rs_scatter(
    const unsigned long long* __restrict__ keys_in, const unsigned int* __restrict__ vals_in, unsigned long long* __restrict__ keys_out,
    unsigned int* __restrict__ vals_out, const unsigned long long* __restrict__ n_ptr, int shift,
    const unsigned int* __restrict__ block_hist, const unsigned int* __restrict__ digit_total) {
    constexpr int SCAT_WAVES = SCAT_THREADS / 64, TILE = SCAT_THREADS * SCAT_ITEMS, PER = RADIX / SCAT_THREADS;
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
            v[k] = a[threadIdx.x * PER + k];
            mine += v[k];
        }
        const unsigned int incl = wave_incl_sum_u32(mine);
        if (lane == 63) wave_tmp[wave] = incl;
        __syncthreads();
        unsigned int excl = incl - mine;
        for (int w = 0; w < wave; ++w) excl += wave_tmp[w];
#pragma unroll
        for (int k = 0; k < PER; ++k) {
            a[threadIdx.x * PER + k] = excl;
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
        for (int d = threadIdx.x * PER; d < (threadIdx.x + 1) * PER; ++d) {
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

// BH value of sorted position i (0-based, global rank = rank0 + i + 1): min(p*N/rank, 1), myStats.py:35-38
__device__ __forceinline__ double bh_value(unsigned long long key_bits, double n_tests, double rank) {
    const double pv = __longlong_as_double((long long)key_bits);
    double v = pv * n_tests / rank;           // (p*N)/(i+1): mul then div, never fused
    if (1.0 < v || pv == 1.0) v = 1.0;        // min(bh, 1); p == 1.0 is 1.0 whatever N / rank says (myStats.py:33-34)
    // The reference's running maximum starts at 0 (myStats.py:30): invisible while N > 0, but fit_Spline can pass a NEGATIVE
    // number of tests (possible-pair counts go negative with unmappable loci) and then every bh value is negative
    // and every q is 0 (tests/golden/f12_bh_nonpositive_N.npz).  Clamping the values is the same running maximum.
    if (v < 0.0) v = 0.0;
    return v;
}

constexpr int BH_THREADS = 256;
constexpr int BH_ITEMS = 8;
constexpr int BH_TILE = BH_THREADS * BH_ITEMS;

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
__global__ __launch_bounds__(1024) void bh_scan_tiles(double* __restrict__ tile_max,
                                                      const unsigned long long* __restrict__ n_ptr, int64_t n_fixed,
                                                      double carry_in, double* __restrict__ total_max) {
    __shared__ double part[1024];
    const int64_t n = n_ptr ? (int64_t)*n_ptr : n_fixed;
    const int64_t tiles = (n + BH_TILE - 1) / BH_TILE;
    const int64_t per = (tiles + 1023) / 1024;
    const int64_t beg = (int64_t)threadIdx.x * per, end = min(tiles, beg + per);
    double m = 0.0;
    for (int64_t t = beg; t < end; ++t) m = fmax(m, tile_max[t]);
    part[threadIdx.x] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        double run = carry_in;
        for (int i = 0; i < 1024; ++i) {
            const double c = part[i];
            part[i] = run;
            run = fmax(run, c);
        }
        if (total_max) *total_max = run;
    }
    __syncthreads();
    double run = part[threadIdx.x];
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
                                                       const double* __restrict__ extra_carry, double* __restrict__ q_out) {
    __shared__ double wtot[BH_THREADS / 64];
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
            if (vals)
                __builtin_nontemporal_store(qv, q_out + vals[i]);       // one 8-byte store into a line nobody else touches soon: no allocate
            else
                q_out[i] = qv;
        }
    }
}

// (Measured and dropped in round 4, profiles/r04_f_small_ab.txt: ONE resident launch for small survivor sets - 64 workgroups,
// a tile each, the six passes and the BH scan separated by device-wide barriers instead of 21 launches.  The kernels of a small
// sort already run back to back without gaps (profiles/r03_z_c2_timeline.txt); what a launch boundary costs is what a
// device-scope barrier costs too - the eight XCDs' L2s are made coherent by writing them back - and 19 such barriers took
// 0.58 ms where the 21 launches take 0.22 ms for the same 77 k keys.)

// ===================================================================================================
// non-fixed-size mode (-r 0): loci and distances are arbitrary integers, so the dense index arithmetic of the
// fixed-size path is replaced by sort + run detection (reusing the radix sort above)
// ===================================================================================================
constexpr int SEG_THREADS = 256;
constexpr int SEG_ITEMS = 16;
constexpr int SEG_TILE = SEG_THREADS * SEG_ITEMS;

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

// outliers of a -r 0 pass: skip mask + the list of their distances (the reference's SortedList outliersdist)
__global__ void nf_fold_outliers(const int32_t* __restrict__ loc1, const int32_t* __restrict__ loc2,
                                 const double* __restrict__ pvals, double thres, uint8_t* __restrict__ skip,
                                 uint8_t* __restrict__ seen_twice, int64_t n, const int32_t* __restrict__ slot_mid,
                                 unsigned long long* __restrict__ dist_list, unsigned long long* __restrict__ n_out,
                                 unsigned long long* __restrict__ first_dup, const long long* __restrict__ grow) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        if (!(pvals[i] < thres)) continue;
        if (skip[i]) {
            seen_twice[i] = 1;
            atomicMin(first_dup, (unsigned long long)(grow ? grow[i] : i));
        }
        skip[i] = 1;
        const int l1 = loc1[i], l2 = loc2[i];
        const int s2 = l2 < 0 ? ~l2 : l2;
        const long long dist = llabs((long long)slot_mid[l1] - (long long)slot_mid[s2]);   // also for inter rows (fithic.py:1217)
        dist_list[atomicAdd(n_out, 1ull)] = (unsigned long long)dist;
    }
}

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

}  // namespace fhx

// =====================================================================================================
// Context + C ABI
// =====================================================================================================
using namespace fhx;

namespace fhx {
struct DistState;
}

struct FhxPinnedPair;                            // fhx_emit.inc: two pinned 64 MB buffers + events, kept for the context's life
void fhx_pinned_pair_free(FhxPinnedPair* p);

struct fhx_ctx {
    FhxPinnedPair* pinned = nullptr;
    struct TextIngest;                           // fhx_ingest.inc: a parsed contacts text waiting for its chromosome ids
    TextIngest* text_ingest = nullptr;
    int device = -1;
    hipStream_t stream = nullptr;
    hipEvent_t ev[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};   // [6],[7]: around the heavy K2 launch
    long long n_heavy_last = 0;
    bool ev_valid[3] = {false, false, false};
    std::string err;
    fhx_params prm{};
    bool have_params = false;

    // fragments
    FragTable frags;
    bool have_frags = false;
    int n_chr = 0;

    // bias rows as loaded (resolved onto the slot grid when pairs are known)
    std::vector<int32_t> bias_chr, bias_mid;
    std::vector<double> bias_val;
    bool have_bias = false;

    // pairs (device) + grid
    int64_t n_rows = 0;
    int32_t *d_loc1 = nullptr, *d_loc2 = nullptr, *d_count = nullptr;
    std::vector<ChrGrid> grid;
    ChrGrid* d_grid = nullptr;
    int16_t* d_slot_chr = nullptr;
    double* d_slot_bias = nullptr;
    int64_t n_slots = 0;
    int64_t n_dist = 0;
    bool tables_dirty = true;

    // pass state
    int pass_no = 0;                  // passes completed so far
    uint8_t *d_skip = nullptr, *d_outlier = nullptr, *d_seen_twice = nullptr;
    bool skip_active = false;
    unsigned long long *d_hist_cc = nullptr, *d_hist_np = nullptr, *d_out_hist = nullptr, *d_misc = nullptr;
    K1Sums* d_sums = nullptr;
    fhx_stats stats{};
    bool have_stats = false;
    std::vector<int64_t> h_hist_cc, h_hist_np, h_out_hist;
    int64_t n_outliers_total = 0;
    long long* d_grow = nullptr;      // file position of every local row (shards, -p >= 3 only)
    int64_t skip_limit = INT64_MAX;   // row of the first duplicated outlier line: later rows are no longer skipped
    bool outlier_hist_nonempty = false;
    PassFit fit;
    bool have_fit = false;
    bool have_bins = false;
    double* d_lut = nullptr;
    double *d_lbeta_intra = nullptr, *d_invb_intra = nullptr, *d_lbeta_inter = nullptr, *d_invb_inter = nullptr;
    // the tables of one fit (prior LUT | four per-count tables | -r 0: spline table x, y) live in ONE device buffer filled by ONE
    // copy from a pinned staging buffer: the pointers above point into it
    double* d_fit_tables = nullptr;
    double* h_fit_stage = nullptr;              // pinned
    size_t fit_tables_cap = 0;                  // doubles
    hipEvent_t ev_fit_copy = nullptr;           // the last copy out of h_fit_stage
    double *d_p = nullptr, *d_q = nullptr;
    bool have_p = false, have_q = false;

    // sort workspace
    unsigned long long *d_keys[2] = {nullptr, nullptr};
    unsigned int *d_vals[2] = {nullptr, nullptr};
    unsigned int* d_block_hist = nullptr;
    unsigned int* d_digit_total = nullptr;
    unsigned long long* d_top_hist = nullptr;
    unsigned long long* d_k2_hist = nullptr;          // K3's key histogram as K2 gathered it while storing p (4096 bins)
    bool k2_hist_valid = false;
    unsigned char* d_work = nullptr;                  // the K2 queues and the K3 sort buffers are views into this block
    QEntry* d_queue[2] = {nullptr, nullptr};          // K2's per-class row queues (sharded, see QSpan)
    int64_t queue_cap = 0;                            // entries per queue buffer
    unsigned long long* d_k2_counts = nullptr;        // (K2_QUEUES + 1) x K2_MAX_SHARDS queue counters
    QEntry* d_queue_sorted = nullptr;                 // the 300-iteration class, bucketed by (binomial, count), 64-aligned buckets
    dev::CfRow* d_cf_tab = nullptr;                   // K2H_GENERIC x 300 rows of iteration constants
    long long *d_stats_stage = nullptr, *h_stats_stage = nullptr;   // K1's sums + histogram window: device block, pinned host copy
    size_t stats_stage_cap = 0;
    int k2_shards = 0;                                // shards (= k2_classify workgroups) of the last fhx_pvalues
    unsigned int* d_k2h_off = nullptr;                // K2H_BUCKETS + 1 bucket starts
    unsigned char* d_memo = nullptr;                  // no-bias table path: virtual rows, table, overflow list (kept across passes)
    size_t memo_bytes = 0;
    // non-fixed-size mode (-r 0), and -r N > 0 on loci that do not share one grid per chromosome (offgrid): arbitrary
    // midpoints, distinct observed distances as histogram keys, table lookup by search; offgrid keeps the fixed-size
    // possible-pair enumeration at multiples of the resolution (fithic.py:592-689)
    bool nonfixed = false;
    bool offgrid = false;
    int32_t* d_slot_mid = nullptr;
    std::vector<unsigned long long> h_slot_keys;      // sorted distinct (chr << 32 | mid) of every locus the rows touch
    std::vector<int64_t> h_dist_keys;                 // distinct in-range distances of the current pass, ascending
    std::vector<int64_t> h_outlier_dists;             // outlier distances of all earlier passes, ascending (a multiset)
    std::vector<int64_t> h_outlier_dists_global;      // sharded runs with explicit distances: the multiset over all ranks
    bool outlier_dists_are_global = false;
    double *d_table_x = nullptr, *d_table_y = nullptr;
    unsigned int* d_seg_ids = nullptr;                // run ids / tile counts scratch
    unsigned int* d_seg_tiles = nullptr;
    double* d_tile_max = nullptr;
    int sorted_buf = 0;
    int64_t n_sorted = -1;
    std::vector<int64_t> fdr_counts;
    fhx::DistState* dist = nullptr;                 // communicator + exchange buffers of sharded runs (fhx_dist.inc)
    bool dist_ndist_agreed = false;                   // sharded runs: the histogram length was made equal on all ranks
    long long dist_ndist_global = -1;                 // ... the all-reduced answer (max length | non-fixed bit), -1 = not asked yet
    bool dist_any_nonfixed = false;                   // ... and some rank holds off-grid / -r 0 rows (agreed in the same all-reduce)
};

namespace {

int fail(fhx_ctx* c, int code, const std::string& msg) {
    if (c) c->err = msg;
    return code;
}

#define FHX_HIP(call)                                                                                     \
    do {                                                                                                  \
        hipError_t e_ = (call);                                                                           \
        if (e_ != hipSuccess)                                                                             \
            return fail(ctx, FHX_ERR_HIP, std::string(#call) + ": " + hipGetErrorString(e_));            \
    } while (0)

template <typename T>
void dev_free(T*& p) {
    if (p) (void)hipFree(p);
    p = nullptr;
}

// temporary device allocations of one call: freed on every return path (FHX_HIP returns early on errors)
// Device blocks kept between the batches of one call: allocating and freeing GBs per batch stalls behind the other thread's
// hipFree (a batch of the device writer waited up to 0.5 s in its allocations); a block goes back here instead and the next
// batch, which asks for the same sizes in the same order, takes it again.  Freed when the pool goes out of scope.
struct ScratchPool {
    std::mutex mu;
    std::vector<std::pair<void*, size_t>> idle;
    ~ScratchPool() {
        for (auto& b : idle) (void)hipFree(b.first);
    }
    void* take(size_t bytes, size_t* real) {           // the smallest idle block that is large enough, or nullptr
        std::lock_guard<std::mutex> g(mu);
        size_t best = idle.size();
        for (size_t i = 0; i < idle.size(); ++i)
            if (idle[i].second >= bytes && (best == idle.size() || idle[i].second < idle[best].second)) best = i;
        if (best == idle.size()) return nullptr;
        void* p = idle[best].first;
        *real = idle[best].second;
        idle.erase(idle.begin() + (long)best);
        return p;
    }
    void give(void* p, size_t bytes) {
        std::lock_guard<std::mutex> g(mu);
        idle.emplace_back(p, bytes);
    }
};

struct DeviceScratch {
    std::vector<std::pair<void*, size_t>> v;
    ScratchPool* pool = nullptr;                       // where the blocks go at the end instead of hipFree
    DeviceScratch() = default;
    explicit DeviceScratch(ScratchPool* p) : pool(p) {}
    ~DeviceScratch() {
        for (auto& b : v) {
            if (pool)
                pool->give(b.first, b.second);
            else
                (void)hipFree(b.first);
        }
    }
    template <typename T>
    hipError_t get(T** p, size_t bytes) {
        bytes = std::max<size_t>(bytes, 16);
        if (pool) {
            size_t real = 0;
            if (void* q = pool->take(bytes, &real)) {
                *p = (T*)q;
                v.emplace_back(q, real);
                return hipSuccess;
            }
        }
        const hipError_t e = hipMalloc((void**)p, bytes);
        if (e == hipSuccess) v.emplace_back((void*)*p, bytes);
        return e;
    }
};

int grid_for(int64_t n, int threads, int max_blocks = 256 * 8) {
    const int64_t b = (n + threads - 1) / threads;
    return (int)std::max<int64_t>(1, std::min<int64_t>(b, max_blocks));
}

// k2_classify over n rows: workgroup b of `grid` takes tiles b, b + grid, ... and queues into shard b, so a shard receives at
// most ceil(tiles / grid) tiles of rows, whatever their classes
int k2_classify_grid(int64_t n) { return grid_for(n, K2_CL_TILE, K2_MAX_SHARDS); }
long long k2_shard_capacity(int64_t n) {              // entries per shard region: the rows one workgroup of k2_classify can meet
    const long long tiles = std::max<long long>(1, (n + K2_CL_TILE - 1) / K2_CL_TILE), grid = k2_classify_grid(n);
    return ((tiles + grid - 1) / grid) * (long long)K2_CL_TILE;
}

K2Params make_k2_params(fhx_ctx* c) {
    K2Params P{};
    P.loc1 = c->d_loc1;
    P.loc2 = c->d_loc2;
    P.count = c->d_count;
    P.slot_bias = c->d_slot_bias;
    P.no_bias = !c->have_bias;
    P.prior_lut = c->d_lut;
    P.lut_len = (int)std::min<size_t>(std::max<size_t>(c->fit.prior_lut.size(), 1), (size_t)INT32_MAX);
    const double n_intra = (double)c->stats.in_range_sum, n_inter = (double)c->stats.inter_sum;
    P.intra = dev::BinomTables{c->d_lbeta_intra, c->d_invb_intra, n_intra, (n_intra + 1.0) < dev::kMaxGam};
    P.inter = dev::BinomTables{c->d_lbeta_inter, c->d_invb_inter, n_inter, (n_inter + 1.0) < dev::kMaxGam};
    P.inter_chr_prob = c->fit.inter_chr_prob;
    P.outlier_thres = 1.0 / c->fit.bh_total_tests;
    const int64_t res = std::max<int64_t>(c->prm.resolution, 1);          // -r 0 does not use the index window
    P.lo_idx = (int)std::min<int64_t>((c->prm.dist_low + res - 1) / res, INT32_MAX);
    P.hi_idx = (int)std::min<int64_t>(c->prm.dist_up / res, INT32_MAX);
    P.mode = c->prm.mode;
    P.n = c->n_rows;
    P.p = c->d_p;
    P.top_hist = nullptr;
    P.outlier = c->d_outlier;
    P.nonfixed = c->nonfixed ? 1 : 0;
    P.slot_mid = c->d_slot_mid;
    P.table_x = c->d_table_x;
    P.table_y = c->d_table_y;
    P.n_table = (int)c->fit.table_x.size();
    P.min_x = c->fit.min_x;
    P.max_x = c->fit.max_x;
    P.dist_low = c->prm.dist_low;
    P.dist_up = c->prm.dist_up;
    return P;
}

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

int ensure_sort_scratch_early(fhx_ctx* ctx) {
    if (!ctx->d_block_hist) FHX_HIP(hipMalloc(&ctx->d_block_hist, (size_t)RADIX * SORT_BLOCKS * sizeof(unsigned int)));
    if (!ctx->d_digit_total) FHX_HIP(hipMalloc(&ctx->d_digit_total, RADIX * sizeof(unsigned int)));
    if (!ctx->d_misc) FHX_HIP(hipMalloc(&ctx->d_misc, 192 * sizeof(unsigned long long)));
    if (!ctx->d_top_hist) FHX_HIP(hipMalloc(&ctx->d_top_hist, TOP_BINS * sizeof(unsigned long long)));
    return FHX_OK;
}

// LSD radix sort of (u64 key, u32 payload) pairs over the low `passes`*11 key bits; n lives in *counter (device)
// Chunks (= workgroups) of a sort of about n keys: a count matrix of RADIX x blocks is scanned in every pass, so a small sort
// must not pay for 1024 of them (6 passes over ~10^6 keys: 0.26 ms with 1024 blocks, a third of that with 64).  n_hint < 0:
// the size is only known on the device.
int sort_blocks_for(int64_t n_hint) {
    if (n_hint < 0) return SORT_BLOCKS;
    const int64_t want = (n_hint + 4 * SORT_TILE - 1) / (4 * SORT_TILE);          // >= four tiles per workgroup
    return (int)std::max<int64_t>(64, std::min<int64_t>(SORT_BLOCKS, (want + 63) / 64 * 64));
}

// one scatter pass
void launch_rs_scatter(fhx_ctx* ctx, int nblk, const unsigned long long* keys_in, const unsigned int* vals_in, unsigned long long* keys_out,
                       unsigned int* vals_out, const unsigned long long* counter, int shift) {
    hipLaunchKernelGGL(HIP_KERNEL_NAME(rs_scatter<512, 4>), dim3(nblk), dim3(512), 0, ctx->stream, keys_in, vals_in, keys_out, vals_out,
                       counter, shift, (const unsigned int*)ctx->d_block_hist, (const unsigned int*)ctx->d_digit_total);
}

int radix_sort_pairs(fhx_ctx* ctx, unsigned long long* keys[2], unsigned int* vals[2], const unsigned long long* counter,
                     int passes, int* result_buf, int64_t n_hint = -1) {
    const int nblk = sort_blocks_for(n_hint);
    int src = 0;
    for (int pass = 0; pass < passes; ++pass) {
        const int shift = pass * RADIX_BITS;
        hipLaunchKernelGGL(rs_count, dim3(nblk), dim3(SORT_THREADS), 0, ctx->stream, keys[src], counter, shift,
                           ctx->d_block_hist);
        hipLaunchKernelGGL(rs_scan, dim3(RADIX), dim3((nblk + 63) / 64 * 64), 0, ctx->stream, ctx->d_block_hist, ctx->d_digit_total, nblk);
        launch_rs_scatter(ctx, nblk, keys[src], vals[src], keys[1 - src], vals[1 - src], counter, shift);
        src = 1 - src;
    }
    FHX_HIP(hipGetLastError());
    *result_buf = src;
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
    int rc = ensure_sort_scratch_early(ctx);
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
    hipLaunchKernelGGL(k0_slots, dim3(blocks), dim3(256), 0, ctx->stream, c1, m1, c2, m2, cnt, n, res, ctx->d_grid,
                       ctx->d_loc1, ctx->d_loc2, ctx->d_count);
    FHX_HIP(hipGetLastError());
    FHX_HIP(hipStreamSynchronize(ctx->stream));
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

}  // namespace

extern "C" {

const char* fhx_version(void) { return "fithic-mi355x 0.1.0 (gfx950)"; }

int fhx_create(int device, fhx_ctx** out) {
    if (!out) return FHX_ERR_ARG;
    *out = nullptr;
    fhx_ctx* ctx = new (std::nothrow) fhx_ctx();
    if (!ctx) return FHX_ERR_NOMEM;
    ctx->device = device;
    if (device >= 0) {
        int count = 0;
        if (hipGetDeviceCount(&count) != hipSuccess || device >= count) {
            delete ctx;
            return FHX_ERR_NO_DEVICE;
        }
        if (hipSetDevice(device) != hipSuccess || hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking) != hipSuccess) {
            delete ctx;
            return FHX_ERR_HIP;
        }
        for (auto& e : ctx->ev)
            if (hipEventCreate(&e) != hipSuccess) {
                delete ctx;
                return FHX_ERR_HIP;
            }
    }
    *out = ctx;
    return FHX_OK;
}

int fhx_warmup(int device) {
    if (device < 0) return FHX_ERR_ARG;
    if (hipSetDevice(device) != hipSuccess) return FHX_ERR_NO_DEVICE;
    return hipFree(nullptr) == hipSuccess ? FHX_OK : FHX_ERR_HIP;
}

void fhx_destroy(fhx_ctx* ctx) {
    if (!ctx) return;
    (void)fhx_comm_destroy(ctx);
    if (ctx->device >= 0) {
        (void)hipSetDevice(ctx->device);
        if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
        fhx_ingest_contacts_discard(ctx);
        fhx_pinned_pair_free(ctx->pinned);
        ctx->pinned = nullptr;
        dev_free(ctx->d_loc1);
        dev_free(ctx->d_loc2);
        dev_free(ctx->d_count);
        dev_free(ctx->d_grid);
        dev_free(ctx->d_slot_chr);
        dev_free(ctx->d_slot_bias);
        dev_free(ctx->d_skip);
        dev_free(ctx->d_outlier);
        dev_free(ctx->d_seen_twice);
        dev_free(ctx->d_grow);
        dev_free(ctx->d_hist_cc);
        dev_free(ctx->d_hist_np);
        dev_free(ctx->d_out_hist);
        dev_free(ctx->d_misc);
        dev_free(ctx->d_sums);
        dev_free(ctx->d_fit_tables);            // d_lut, d_lbeta_*, d_invb_*, d_table_x / y point into it
        if (ctx->h_fit_stage) (void)hipHostFree(ctx->h_fit_stage);
        if (ctx->ev_fit_copy) (void)hipEventDestroy(ctx->ev_fit_copy);
        dev_free(ctx->d_p);
        dev_free(ctx->d_q);
        dev_free(ctx->d_work);
        dev_free(ctx->d_block_hist);
        dev_free(ctx->d_digit_total);
        dev_free(ctx->d_top_hist);
        dev_free(ctx->d_k2_hist);
        dev_free(ctx->d_cf_tab);
        dev_free(ctx->d_stats_stage);
        if (ctx->h_stats_stage) (void)hipHostFree(ctx->h_stats_stage);
        dev_free(ctx->d_k2h_off);
        dev_free(ctx->d_k2_counts);
        dev_free(ctx->d_memo);
        dev_free(ctx->d_slot_mid);
        dev_free(ctx->d_seg_ids);
        dev_free(ctx->d_seg_tiles);
        dev_free(ctx->d_tile_max);
        for (auto& e : ctx->ev)
            if (e) (void)hipEventDestroy(e);
        if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
    }
    delete ctx;
}

const char* fhx_last_error(fhx_ctx* ctx) { return ctx ? ctx->err.c_str() : "null context"; }

int fhx_set_params(fhx_ctx* ctx, const fhx_params* p) {
    if (!ctx || !p) return FHX_ERR_ARG;
    if (p->resolution < 0) return fail(ctx, FHX_ERR_ARG, "resolution must be >= 0 (0 = non-fixed-size data)");
    if (p->resolution > INT32_MAX) return fail(ctx, FHX_ERR_ARG, "resolution too large");
    if (p->n_bins <= 0 || p->mapp_thres < 0 || p->mode < 0 || p->mode > 2) return fail(ctx, FHX_ERR_ARG, "bad parameter");
    if (p->bias_low > p->bias_up)
        return fail(ctx, FHX_ERR_REFERENCE_EXIT, "bias lower bound is greater than bias upper bound (fithic.py:261-263)");
    if (ctx->n_rows > 0 && ctx->have_params && p->resolution != ctx->prm.resolution)
        return fail(ctx, FHX_ERR_ARG, "the resolution cannot change after the contact rows were loaded");
    // the per-slot bias table depends on the grid and the bias bounds only: the drop-in layer re-sends unchanged parameters
    // before every stage, which must not cost a rebuild + upload of the table
    if (!ctx->have_params || p->resolution != ctx->prm.resolution || p->bias_low != ctx->prm.bias_low || p->bias_up != ctx->prm.bias_up)
        ctx->tables_dirty = true;
    ctx->prm = *p;
    ctx->nonfixed = p->resolution == 0 || ctx->offgrid;
    ctx->have_params = true;
    return FHX_OK;
}

int fhx_load_fragments(fhx_ctx* ctx, const int32_t* chr, const int32_t* mid, const int32_t* hits, int64_t n,
                       const int32_t* chr_sort_rank, int32_t n_chr) {
    if (!ctx || !chr || !mid || !hits || !chr_sort_rank || n < 0 || n_chr <= 0) return FHX_ERR_ARG;
    if (!ctx->have_params) return fail(ctx, FHX_ERR_ARG, "fhx_set_params must be called first");
    std::vector<int64_t> cnt(n_chr, 0), mx(n_chr, -1);
    std::vector<uint8_t> present(n_chr, 0);
    std::vector<std::vector<int32_t>> mids_of(ctx->nonfixed ? n_chr : 0);
    for (int64_t i = 0; i < n; ++i) {
        const int c = chr[i];
        if (c < 0 || c >= n_chr) return fail(ctx, FHX_ERR_ARG, "fragment chromosome id out of range");
        present[c] = 1;
        if (hits[i] >= ctx->prm.mapp_thres) {
            ++cnt[c];
            mx[c] = std::max<int64_t>(mx[c], mid[i]);
            if (ctx->nonfixed) mids_of[c].push_back(mid[i]);
        }
    }
    std::vector<int> order;
    for (int c = 0; c < n_chr; ++c)
        if (present[c]) order.push_back(c);
    std::sort(order.begin(), order.end(), [&](int a, int b) { return chr_sort_rank[a] < chr_sort_rank[b]; });
    ctx->frags = FragTable();
    for (int c : order) {
        ctx->frags.chr_id.push_back(c);
        ctx->frags.n_mappable.push_back(cnt[c]);
        ctx->frags.max_mid.push_back(mx[c]);
        if (ctx->nonfixed) {
            std::sort(mids_of[c].begin(), mids_of[c].end());
            ctx->frags.mids.push_back(std::move(mids_of[c]));
        }
    }
    ctx->n_chr = std::max(ctx->n_chr, n_chr);
    ctx->have_frags = true;
    return FHX_OK;
}

int fhx_load_bias(fhx_ctx* ctx, const int32_t* chr, const int32_t* mid, const double* bias, int64_t n) {
    if (!ctx || n < 0 || (n > 0 && (!chr || !mid || !bias))) return FHX_ERR_ARG;
    ctx->bias_chr.assign(chr, chr + n);
    ctx->bias_mid.assign(mid, mid + n);
    ctx->bias_val.assign(bias, bias + n);
    ctx->have_bias = n > 0;                 // an empty bias dictionary is falsy in the reference (fithic.py:1026)
    ctx->tables_dirty = true;
    return FHX_OK;
}

int fhx_load_pairs_device(fhx_ctx* ctx, const void* c1, const void* m1, const void* c2, const void* m2, const void* cnt,
                          int64_t n, void* stream) {
    if (!ctx || n < 0 || (n > 0 && (!c1 || !m1 || !c2 || !m2 || !cnt))) return FHX_ERR_ARG;
    if (ctx->device < 0) return fail(ctx, FHX_ERR_NO_DEVICE, "host-only context");
    FHX_HIP(hipSetDevice(ctx->device));
    // the producer's stream; nullptr = the legacy default stream, which is where PyTorch computes unless told otherwise.  The
    // context's own stream is non-blocking (no implicit ordering with the default stream), so this wait is what makes rows that
    // were just written by the caller's kernels safe to read here.
    FHX_HIP(hipStreamSynchronize((hipStream_t)stream));
    ctx->offgrid = false;
    ctx->nonfixed = ctx->have_params && ctx->prm.resolution == 0;
    if (ctx->have_params && ctx->nonfixed)
        return ingest_device_rows_nonfixed(ctx, (const int32_t*)c1, (const int32_t*)m1, (const int32_t*)c2, (const int32_t*)m2,
                                           (const int32_t*)cnt, n);
    return ingest_device_rows(ctx, (const int32_t*)c1, (const int32_t*)m1, (const int32_t*)c2, (const int32_t*)m2,
                              (const int32_t*)cnt, n);
}

int fhx_load_pairs(fhx_ctx* ctx, const int32_t* chr1, const int32_t* mid1, const int32_t* chr2, const int32_t* mid2,
                   const int32_t* count, int64_t n) {
    if (!ctx || n < 0 || (n > 0 && (!chr1 || !mid1 || !chr2 || !mid2 || !count))) return FHX_ERR_ARG;
    if (ctx->device < 0) return fail(ctx, FHX_ERR_NO_DEVICE, "host-only context cannot hold contact rows");
    FHX_HIP(hipSetDevice(ctx->device));
    int32_t* d[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
    const int32_t* h[5] = {chr1, mid1, chr2, mid2, count};
    DeviceScratch tmp;
    for (int k = 0; k < 5; ++k) {
        FHX_HIP(tmp.get(&d[k], (size_t)std::max<int64_t>(n, 1) * sizeof(int32_t)));         // an empty shard is legal
        if (n) FHX_HIP(hipMemcpyAsync(d[k], h[k], (size_t)n * sizeof(int32_t), hipMemcpyHostToDevice, ctx->stream));
    }
    FHX_HIP(hipStreamSynchronize(ctx->stream));
    ctx->offgrid = false;
    ctx->nonfixed = ctx->have_params && ctx->prm.resolution == 0;
    const int rc = (ctx->have_params && ctx->nonfixed) ? ingest_device_rows_nonfixed(ctx, d[0], d[1], d[2], d[3], d[4], n)
                                                       : ingest_device_rows(ctx, d[0], d[1], d[2], d[3], d[4], n);
    return rc;
}

// -r 0: classification + sums as K1, then the in-range (distance, count) pairs are radix-sorted by distance and the runs
// are reduced to (distinct distance, sum of counts, rows): the reference's mainDic for arbitrary distances
static int pass_stats_nonfixed(fhx_ctx* ctx, fhx_stats* out) {
    unsigned long long* counter = ctx->d_misc + 10;
    FHX_HIP(hipMemsetAsync(counter, 0, sizeof(unsigned long long), ctx->stream));
    FHX_HIP(hipMemsetAsync(ctx->d_sums, 0, sizeof(K1Sums), ctx->stream));
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
    }
    ctx->ev_valid[0] = true;
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

// K1 of the fixed-size path on the context's stream: histograms and sums stay in HBM
static int launch_k1(fhx_ctx* ctx) {
    const int64_t res = ctx->prm.resolution;
    const int64_t lo = (ctx->prm.dist_low + res - 1) / res;
    const int64_t hi = std::min<int64_t>(ctx->prm.dist_up / res, ctx->n_dist - 1);
    FHX_HIP(hipMemsetAsync(ctx->d_hist_cc, 0, ctx->n_dist * sizeof(unsigned long long), ctx->stream));
    FHX_HIP(hipMemsetAsync(ctx->d_hist_np, 0, ctx->n_dist * sizeof(unsigned long long), ctx->stream));
    FHX_HIP(hipMemsetAsync(ctx->d_sums, 0, sizeof(K1Sums), ctx->stream));
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
    return FHX_OK;
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
        if (ctx->d_stats_stage) (void)hipFree(ctx->d_stats_stage);
        if (ctx->h_stats_stage) (void)hipHostFree(ctx->h_stats_stage);
        ctx->d_stats_stage = ctx->h_stats_stage = nullptr;
        ctx->stats_stage_cap = 0;
        FHX_HIP(hipMalloc(&ctx->d_stats_stage, (pack_len + 1024) * sizeof(long long)));
        FHX_HIP(hipHostMalloc((void**)&ctx->h_stats_stage, (pack_len + 1024) * sizeof(long long), hipHostMallocDefault));
        ctx->stats_stage_cap = pack_len + 1024;
    }
    hipLaunchKernelGGL(k1_pack_window, dim3(grid_for((int64_t)pack_len, 256, 64)), dim3(256), 0, ctx->stream, (const K1Sums*)ctx->d_sums,
                       (const unsigned long long*)ctx->d_hist_cc, (const unsigned long long*)ctx->d_hist_np, (int)a, w, ctx->d_stats_stage);
    FHX_HIP(hipGetLastError());
    FHX_HIP(hipMemcpyAsync(ctx->h_stats_stage, ctx->d_stats_stage, pack_len * sizeof(long long), hipMemcpyDeviceToHost, ctx->stream));
    FHX_HIP(hipStreamSynchronize(ctx->stream));
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

int fhx_get_stats(fhx_ctx* ctx, fhx_stats* out) {
    if (!ctx || !out) return FHX_ERR_ARG;
    if (!ctx->have_stats) return fail(ctx, FHX_ERR_ARG, "no pass statistics yet");
    *out = ctx->stats;
    return FHX_OK;
}

int fhx_set_global_stats(fhx_ctx* ctx, const fhx_stats* g, const int64_t* hist_sumcc, const int64_t* hist_npairs,
                         int64_t n_dist) {
    if (!ctx || !g || !hist_sumcc || !hist_npairs || n_dist <= 0) return FHX_ERR_ARG;
    if (ctx->nonfixed && ctx->h_dist_keys.size() != (size_t)n_dist)
        return fail(ctx, FHX_ERR_UNSUPPORTED, "-r 0: call fhx_set_dist_keys with the distinct distances first (their number must match the histograms)");
    const int64_t rows = ctx->n_rows;
    ctx->stats = *g;
    ctx->stats.n_rows = rows > 0 ? rows : g->n_rows;
    const int64_t len = std::max(ctx->n_dist, n_dist);      // the device histograms keep their local length
    ctx->stats.n_dist = len;
    ctx->h_hist_cc.assign(hist_sumcc, hist_sumcc + n_dist);
    ctx->h_hist_np.assign(hist_npairs, hist_npairs + n_dist);
    ctx->h_hist_cc.resize((size_t)len, 0);
    ctx->h_hist_np.resize((size_t)len, 0);
    ctx->have_stats = true;
    ctx->have_fit = false;
    return FHX_OK;
}

int fhx_set_dist_keys(fhx_ctx* ctx, const int64_t* keys, int64_t n) {
    if (!ctx || n < 0 || (n > 0 && !keys)) return FHX_ERR_ARG;
    if (!ctx->nonfixed) {                          // -r N with explicit distances: loci off the grid (host-side callers)
        if (!ctx->have_params) return fail(ctx, FHX_ERR_ARG, "fhx_set_params must be called first");
        ctx->offgrid = ctx->nonfixed = true;
    }
    ctx->h_dist_keys.assign(keys, keys + n);
    return FHX_OK;
}

int fhx_set_outlier_dists(fhx_ctx* ctx, const int64_t* dists, int64_t n) {
    if (!ctx || n < 0 || (n > 0 && !dists)) return FHX_ERR_ARG;
    ctx->h_outlier_dists.assign(dists, dists + n);
    std::sort(ctx->h_outlier_dists.begin(), ctx->h_outlier_dists.end());
    if (ctx->pass_no < 1) ctx->pass_no = 1;
    return FHX_OK;
}

int fhx_set_outlier_dist_hist(fhx_ctx* ctx, const int64_t* hist, int64_t n_dist) {
    if (!ctx || !hist || n_dist <= 0) return FHX_ERR_ARG;
    ctx->h_out_hist.assign(hist, hist + n_dist);
    if (ctx->pass_no < 1) ctx->pass_no = 1;          // an outlier multiset exists: this is pass >= 2
    return FHX_OK;
}

static void fill_pass_inputs(fhx_ctx* ctx, PassInputs& in) {
    in.resolution = ctx->prm.resolution;
    in.dist_low = ctx->prm.dist_low;
    in.dist_up = ctx->prm.dist_up;
    in.n_bins = ctx->prm.n_bins;
    in.mode = ctx->prm.mode;
    in.hist_sumcc = ctx->h_hist_cc.data();
    in.hist_npairs = ctx->h_hist_np.data();
    in.n_dist = (int64_t)ctx->h_hist_cc.size();
    in.in_range_sum = ctx->stats.in_range_sum;
    in.inter_count = ctx->stats.inter_count;
    in.inter_sum = ctx->stats.inter_sum;
    if (ctx->nonfixed) {
        in.dist_keys = ctx->h_dist_keys.data();
        const std::vector<int64_t>& od = ctx->outlier_dists_are_global ? ctx->h_outlier_dists_global : ctx->h_outlier_dists;
        in.outlier_dists = ctx->pass_no > 0 ? od.data() : nullptr;
        in.n_outlier_dists = (int64_t)od.size();
        static const int64_t none = 0;
        if (ctx->pass_no > 0 && od.empty()) in.outlier_dists = &none;      // an empty multiset, not "pass 1"
        in.outlier_dist_hist = nullptr;
        return;
    }
    if (ctx->h_out_hist.size() < ctx->h_hist_cc.size()) ctx->h_out_hist.resize(ctx->h_hist_cc.size(), 0);
    in.outlier_dist_hist = ctx->pass_no > 0 ? ctx->h_out_hist.data() : nullptr;
}

int fhx_make_bins(fhx_ctx* ctx, int32_t* n_bins_made) {
    if (!ctx) return FHX_ERR_ARG;
    if (!ctx->have_params) return fail(ctx, FHX_ERR_ARG, "fhx_set_params must be called first");
    if (!ctx->have_stats) return fail(ctx, FHX_ERR_ARG, "fhx_pass_stats (or fhx_set_global_stats) must run first");
    PassInputs in;
    fill_pass_inputs(ctx, in);
    make_bins_stage(in, ctx->fit);
    ctx->have_fit = false;                 // bins only: K2 still needs fhx_fit
    ctx->have_bins = true;
    if (n_bins_made) *n_bins_made = (int32_t)ctx->fit.bins.size();
    return FHX_OK;
}

int fhx_fit(fhx_ctx* ctx, fhx_fit_info* out) {
    if (!ctx) return FHX_ERR_ARG;
    if (!ctx->have_params || !ctx->have_frags) return fail(ctx, FHX_ERR_ARG, "parameters and fragments must be loaded");
    if (!ctx->have_stats) return fail(ctx, FHX_ERR_ARG, "fhx_pass_stats (or fhx_set_global_stats) must run first");
    PassInputs in;
    fill_pass_inputs(ctx, in);
    std::string err;
    const int rc = run_host_pass(in, ctx->frags, ctx->fit, err);
    if (rc != FHX_OK) return fail(ctx, rc, err);
    ctx->have_fit = true;
    ctx->have_bins = true;
    const PassFit& f = ctx->fit;
    if (ctx->device >= 0) {
        FHX_HIP(hipSetDevice(ctx->device));
        if (ctx->tables_dirty) {
            const int r2 = build_slot_tables(ctx);
            if (r2 != FHX_OK) return r2;
        }
        // prior LUT (fixed-size) or the spline table itself (-r 0) + the two pairs of per-count tables: packed into the pinned
        // staging buffer and sent with one copy; nothing waits for it here (the stream orders it before K2)
        const int64_t mc = std::max<int64_t>(ctx->stats.max_count, 1);
        std::vector<double> lb_a, ib_a, lb_e, ib_e;
        build_lbeta_table((double)ctx->stats.in_range_sum, mc, lb_a, ib_a);
        build_lbeta_table((double)ctx->stats.inter_sum, mc, lb_e, ib_e);
        const size_t n_lut = std::max<size_t>(f.prior_lut.size(), 1), n_tab = (size_t)(mc + 1);
        const size_t n_xy = ctx->nonfixed ? std::max<size_t>(f.table_x.size(), 1) : 0;
        const size_t need = n_lut + 4 * n_tab + 2 * n_xy;
        if (need > ctx->fit_tables_cap) {
            FHX_HIP(hipStreamSynchronize(ctx->stream));                // kernels of an earlier pass may still read the old buffer
            dev_free(ctx->d_fit_tables);
            if (ctx->h_fit_stage) (void)hipHostFree(ctx->h_fit_stage);
            ctx->h_fit_stage = nullptr;
            ctx->fit_tables_cap = need + need / 2 + 1024;
            FHX_HIP(hipMalloc(&ctx->d_fit_tables, ctx->fit_tables_cap * sizeof(double)));
            FHX_HIP(hipHostMalloc((void**)&ctx->h_fit_stage, ctx->fit_tables_cap * sizeof(double), hipHostMallocDefault));
        }
        if (!ctx->ev_fit_copy) FHX_HIP(hipEventCreateWithFlags(&ctx->ev_fit_copy, hipEventDisableTiming));
        else FHX_HIP(hipEventSynchronize(ctx->ev_fit_copy));           // the previous fit's copy has left the staging buffer
        double* h = ctx->h_fit_stage;
        size_t at = 0;
        auto put = [&](const double* src, size_t n_src, size_t n_slot) {
            if (n_src) std::memcpy(h + at, src, n_src * sizeof(double));
            double* d = ctx->d_fit_tables + at;
            at += n_slot;
            return d;
        };
        ctx->d_lut = put(f.prior_lut.data(), f.prior_lut.size(), n_lut);
        ctx->d_lbeta_intra = put(lb_a.data(), lb_a.size(), n_tab);
        ctx->d_invb_intra = put(ib_a.data(), ib_a.size(), n_tab);
        ctx->d_lbeta_inter = put(lb_e.data(), lb_e.size(), n_tab);
        ctx->d_invb_inter = put(ib_e.data(), ib_e.size(), n_tab);
        ctx->d_table_x = ctx->d_table_y = nullptr;
        if (ctx->nonfixed) {
            const std::vector<double> tx(f.table_x.begin(), f.table_x.end());
            ctx->d_table_x = put(tx.data(), tx.size(), n_xy);
            ctx->d_table_y = put(f.table_y.data(), std::min(f.table_y.size(), tx.size()), n_xy);
        }
        FHX_HIP(hipMemcpyAsync(ctx->d_fit_tables, h, at * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
        FHX_HIP(hipEventRecord(ctx->ev_fit_copy, ctx->stream));
    }
    if (out) {
        std::memset(out, 0, sizeof(*out));
        out->n_bins_made = (int32_t)f.bins.size();
        out->n_knots = (int32_t)f.spline.t.size();
        out->spline_ier = f.spline.ier;
        out->spline_restarted = f.spline.restarted ? 1 : 0;
        out->n_table = (int64_t)f.table_x.size();
        out->n_frags = f.n_frags;
        out->possible_intra_in_range = f.poss_intra_in_range;
        out->possible_inter_all = f.poss_inter_all;
        out->possible_intra_all = f.poss_intra_all;
        out->max_possible_dist = f.max_possible_dist;
        out->inter_chr_prob = f.inter_chr_prob;
        out->baseline_intra_prob = f.baseline_intra_prob;
        out->spline_s = f.spline_s;
        out->spline_fp = f.spline.fp;
        out->residual = f.residual;
        out->bh_total_tests = f.bh_total_tests;
        out->outlier_thres = 1.0 / f.bh_total_tests;
    }
    return FHX_OK;
}

int fhx_pvalues(fhx_ctx* ctx) {
    if (!ctx) return FHX_ERR_ARG;
    if (ctx->device < 0) return fail(ctx, FHX_ERR_NO_DEVICE, "host-only context");
    if (!ctx->have_fit) return fail(ctx, FHX_ERR_ARG, "fhx_fit must run first");
    FHX_HIP(hipSetDevice(ctx->device));
    if (ctx->tables_dirty) {               // e.g. the bias table arrived after the fit (the reference's call order)
        const int r2 = build_slot_tables(ctx);
        if (r2 != FHX_OK) return r2;
    }
    K2Params P = make_k2_params(ctx);
    FHX_HIP(hipEventRecord(ctx->ev[2], ctx->stream));
    // no bias table, fixed-size loci: evaluate a (distance, count) table instead of every row (see k2_memo_rows)
    int32_t *v_loc1 = nullptr, *v_loc2 = nullptr, *v_count = nullptr;
    double* v_table = nullptr;
    unsigned int* over_rows = nullptr;
    unsigned long long over_cap = 0;
    int memo_cap = -1, memo_nd = 0;
    const bool memo_intra = ctx->prm.mode != FHX_MODE_INTER_ONLY, memo_inter = ctx->prm.mode != FHX_MODE_INTRA_ONLY;
    if (!ctx->have_bias && !ctx->nonfixed && !getenv("FHX_NO_MEMO")) {
        memo_nd = memo_intra ? (int)(P.hi_idx - P.lo_idx + 1) : 0;
        const int64_t budget = std::min<int64_t>(1ll << 24, ctx->n_rows / 4);
        const int64_t per_count = (int64_t)memo_nd + (memo_inter ? 1 : 0);
        if (per_count > 0 && memo_nd >= 0) {
            const int64_t cap = std::min<int64_t>(ctx->stats.max_count, budget / per_count - 1);
            if (cap >= 8) memo_cap = (int)cap;
        }
    }
    // K3's key histogram rides on K2's stores of p - except on the table path, whose class kernels store table entries
    ctx->k2_hist_valid = false;
    if (memo_cap < 0 && !getenv("FHX_NO_FUSED_HIST")) {
        if (!ctx->d_k2_hist) FHX_HIP(hipMalloc(&ctx->d_k2_hist, TOP_BINS * sizeof(unsigned long long)));
        FHX_HIP(hipMemsetAsync(ctx->d_k2_hist, 0, TOP_BINS * sizeof(unsigned long long), ctx->stream));
        P.top_hist = ctx->d_k2_hist;
        ctx->k2_hist_valid = true;
    }
    const K2Params P_rows = P;
    int64_t k2_n = ctx->n_rows;
    if (memo_cap >= 0) {
        k2_n = (int64_t)memo_nd * (memo_cap + 1) + (memo_inter ? (memo_cap + 1) : 0);
        over_cap = (unsigned long long)std::max<int64_t>(ctx->n_rows / 16, 1024);
        const size_t col = ((size_t)k2_n + 3) / 4 * 4;                               // the three columns are read 16 bytes at a time
        const size_t need = col * (4 + 4 + 4 + 8) + (size_t)over_cap * 4 + 64;
        if (need > ctx->memo_bytes) {
            dev_free(ctx->d_memo);
            ctx->memo_bytes = 0;
            FHX_HIP(hipMalloc(&ctx->d_memo, need));
            ctx->memo_bytes = need;
        }
        v_loc1 = reinterpret_cast<int32_t*>(ctx->d_memo);
        v_loc2 = v_loc1 + col;
        v_count = v_loc2 + col;
        v_table = reinterpret_cast<double*>(v_count + col);
        over_rows = reinterpret_cast<unsigned int*>(v_table + col);
        hipLaunchKernelGGL(k2_memo_rows, dim3(grid_for(k2_n, 256)), dim3(256), 0, ctx->stream, memo_nd, P.lo_idx, memo_cap, (int)memo_inter,
                           v_loc1, v_loc2, v_count, k2_n);
        P.loc1 = v_loc1;
        P.loc2 = v_loc2;
        P.count = v_count;
        P.p = v_table;
        P.n = k2_n;
    }
    // queues live in the sort workspace, which is idle until K3: 2 x u32[n] + 2 x u64[n]
    // two entry buffers of n_rows each: [swapped CF up | power series down] and [incbcf up | incbd down]
    K2Queues Q;
    const long long cap_s = k2_shard_capacity(k2_n);
    const int n_shards = k2_classify_grid(k2_n);
    if ((int64_t)n_shards * cap_s > ctx->queue_cap) return fail(ctx, FHX_ERR_HIP, "internal: queue workspace smaller than the shard layout");
    if (!ctx->d_k2_counts) FHX_HIP(hipMalloc(&ctx->d_k2_counts, (size_t)(K2_QUEUES + 1) * K2_MAX_SHARDS * sizeof(unsigned long long)));
    Q.count = ctx->d_k2_counts;
    static const bool own_count_pass = std::getenv("FHX_K2H_COUNT") != nullptr;          // measurements: the separate k2h_count launch
    const bool heavy_sorted = getenv("FHX_K2_LEGACY") == nullptr;
    Q.heavy_hist = nullptr;
    if (heavy_sorted && !own_count_pass) {
        static_assert((K2H_BLOCKS & (K2H_BLOCKS - 1)) == 0, "shard -> column by masking");
        FHX_HIP(hipMemsetAsync(ctx->d_block_hist, 0, (size_t)K2H_BUCKETS * K2H_BLOCKS * sizeof(unsigned int), ctx->stream));
        Q.heavy_hist = ctx->d_block_hist;
    }
    ctx->k2_shards = n_shards;
    auto span = [&](int cls, QEntry* buf, int dir) {
        QSpan& q = Q.q[cls - 1];
        q.base = dir > 0 ? buf : buf + cap_s - 1;
        q.cap_s = cap_s;
        q.dir = dir;
        q.n_shards = n_shards;
        q.count = Q.count + (size_t)(cls - 1) * K2_MAX_SHARDS;
    };
    span(dev::BC_CF_SWAPPED, ctx->d_queue[0], 1);
    span(dev::BC_PSERIES, ctx->d_queue[0], -1);
    span(dev::BC_CF_BCF, ctx->d_queue[1], 1);
    span(dev::BC_CF_BD, ctx->d_queue[1], -1);
    span(K2_CLOSED, ctx->d_queue_sorted, 1);             // the sorted heavy queue is written after k2_closed has run
    // (no reset of the counters: every workgroup of k2_classify writes its own shard's counts)
    {
        const dim3 cgrid(k2_classify_grid(k2_n)), cblock(K2_THREADS);
        // FHX_CL_BASE=1: round 3's kernel (gathers row by row, 24 ballots, the division per row) for A/B runs; FHX_CL_PACK=0 /
        // FHX_CL_TB=0 switch the two later steps off one at a time
        static const bool cl_base = std::getenv("FHX_CL_BASE") != nullptr;
        static const bool cl_pack = !(std::getenv("FHX_CL_PACK") && std::atoi(std::getenv("FHX_CL_PACK")) == 0);
        static const bool cl_tb = !(std::getenv("FHX_CL_TB") && std::atoi(std::getenv("FHX_CL_TB")) == 0);
        if (P.nonfixed)
            hipLaunchKernelGGL(HIP_KERNEL_NAME(k2_classify<1, 4, 0>), cgrid, cblock, 0, ctx->stream, P, Q);
        else if (cl_base)
            hipLaunchKernelGGL(HIP_KERNEL_NAME(k2_classify<0, 4, 0>), cgrid, cblock, 0, ctx->stream, P, Q);
        else if (cl_pack && cl_tb)
            hipLaunchKernelGGL(HIP_KERNEL_NAME(k2_classify<0, 4, 3, true, true>), cgrid, cblock, 0, ctx->stream, P, Q);
        else if (cl_pack)
            hipLaunchKernelGGL(HIP_KERNEL_NAME(k2_classify<0, 4, 0, true, true>), cgrid, cblock, 0, ctx->stream, P, Q);
        else if (cl_tb)
            hipLaunchKernelGGL(HIP_KERNEL_NAME(k2_classify<0, 4, 3, true, false>), cgrid, cblock, 0, ctx->stream, P, Q);
        else
            hipLaunchKernelGGL(HIP_KERNEL_NAME(k2_classify<0, 4, 0, true, false>), cgrid, cblock, 0, ctx->stream, P, Q);
    }
    const dim3 qgrid(256 * 8), qblock(K2_THREADS);
    hipLaunchKernelGGL(k2_closed, qgrid, qblock, 0, ctx->stream, P, Q.q[K2_CLOSED - 1]);
    // totals below 171: the kernels that carry Cephes' pow branch (a binomial without a single contact - no inter-chromosomal
    // rows - classifies every row as trivial and reaches no class kernel: it does not count)
    const bool small_n = (P.intra.small_n && P.intra.n >= 1.0) || (P.inter.small_n && P.inter.n >= 1.0);
#define FHX_LAUNCH_QUEUE(CLS)                                                                                         \
    do {                                                                                                              \
        if (small_n)                                                                                                  \
            hipLaunchKernelGGL(HIP_KERNEL_NAME(k2_queue<CLS, true>), qgrid, qblock, 0, ctx->stream, P, Q.q[(CLS) - 1]);  \
        else                                                                                                          \
            hipLaunchKernelGGL(HIP_KERNEL_NAME(k2_queue<CLS, false>), qgrid, qblock, 0, ctx->stream, P, Q.q[(CLS) - 1]); \
    } while (0)
    const bool legacy_heavy = getenv("FHX_K2_LEGACY") != nullptr;      // A/B and tests: the per-lane kernel of round 1
    if (legacy_heavy) {
        FHX_HIP(hipEventRecord(ctx->ev[6], ctx->stream));
        FHX_LAUNCH_QUEUE(dev::BC_CF_SWAPPED);            // longest-running class first
        FHX_HIP(hipEventRecord(ctx->ev[7], ctx->stream));
    } else {
        static_assert(K2H_BUCKETS == RADIX && K2H_BLOCKS == SORT_BLOCKS, "the radix sort's count matrix and scan are reused");
        const QSpan hs = Q.q[dev::BC_CF_SWAPPED - 1];
        QEntry* hq = ctx->d_queue[0];                    // the handed-back rows: this buffer is dead once it is scattered and the
                                                         // power-series class (its other tenant) has run
        unsigned long long* n_redo = ctx->d_misc + 11;
        FHX_HIP(hipMemsetAsync(n_redo, 0, sizeof(unsigned long long), ctx->stream));
        if (!Q.heavy_hist)              // otherwise k2_classify has counted while it queued
            hipLaunchKernelGGL(k2h_count, dim3(K2H_BLOCKS), dim3(K2H_THREADS), 0, ctx->stream, hs, ctx->d_block_hist);
        hipLaunchKernelGGL(rs_scan, dim3(RADIX), dim3(SORT_BLOCKS), 0, ctx->stream, ctx->d_block_hist, ctx->d_digit_total, (int)SORT_BLOCKS);
        // rows per lane: 4 at 4 waves/SIMD (7.43 -> 6.66 ms per 2.7e7 rows against one row per lane at 8 waves/SIMD; 2 x 8, 2 x 6,
        // 3 x 5, 4 x 3 are within 3 % of each other, profiles/r03_c_heavy_variants.txt); FHX_K2H_ROWS / FHX_K2H_WAVES: measurements
        // rows per lane: 4 at 4 waves/SIMD - C3 (2.7e7 rows in the class) 7.43 -> 6.66 ms, a 1/18 shard (1.5e6 rows) 0.87 -> 0.78 ms of
        // K2 against one row per lane at 8 waves/SIMD; 2 x 8, 3 x 5 and 4 x 3 are within 3 % (profiles/r03_c_*heavy_variants.txt).
        // FHX_K2H_ROWS (1, 2) / FHX_K2H_WAVES (3) select the instantiations kept for measurements.
        static const int heavy_rows = std::getenv("FHX_K2H_ROWS") ? std::atoi(std::getenv("FHX_K2H_ROWS")) : 4;
        static const int heavy_wpe = std::getenv("FHX_K2H_WAVES") ? std::atoi(std::getenv("FHX_K2H_WAVES")) : 0;
        const int hr = (heavy_rows == 1 || heavy_rows == 2) ? heavy_rows : 4;     // the instantiations below: 1, 2 or 4 rows per lane - the
                                                                                   // bucket granule must be the launched kernel's task size
        hipLaunchKernelGGL(k2h_offsets, dim3(1), dim3(1024), 0, ctx->stream, (const unsigned int*)ctx->d_digit_total, ctx->d_k2h_off,
                           64u * (unsigned int)hr);
        hipLaunchKernelGGL(k2h_tables, dim3(K2H_GENERIC), dim3(K2H_TABLE_THREADS), 0, ctx->stream, (const unsigned int*)ctx->d_digit_total,
                           P.intra.n, P.inter.n, ctx->d_cf_tab);
        hipLaunchKernelGGL(k2h_scatter, dim3(K2H_BLOCKS), dim3(K2H_THREADS), 0, ctx->stream, hs,
                           (const unsigned int*)ctx->d_block_hist, (const unsigned int*)ctx->d_k2h_off, ctx->d_queue_sorted);
        FHX_LAUNCH_QUEUE(dev::BC_PSERIES);               // before the redo list reuses the buffer it shares with the heavy queue
        FHX_HIP(hipEventRecord(ctx->ev[6], ctx->stream));
        const K2HeavyParams HP{P.intra, P.inter, P.p, P.top_hist};
#define FHX_HEAVY_N(R, W, S)                                                                                                        \
    hipLaunchKernelGGL(HIP_KERNEL_NAME(k2h_heavy<R, W, S>), dim3(256 * W), dim3(K2H_THREADS), 0, ctx->stream, HP,                     \
                       (const QEntry*)ctx->d_queue_sorted, (const unsigned int*)ctx->d_k2h_off, (const unsigned int*)ctx->d_digit_total, \
                       (const dev::CfRow*)ctx->d_cf_tab, hq, n_redo)
#define FHX_HEAVY(R, W)           \
    do {                          \
        if (small_n)              \
            FHX_HEAVY_N(R, W, true);  \
        else                      \
            FHX_HEAVY_N(R, W, false); \
    } while (0)
        if (hr == 1) FHX_HEAVY(1, 8);
        else if (hr == 2) FHX_HEAVY(2, 8);
        else if (heavy_wpe == 3) FHX_HEAVY(4, 3);
        else FHX_HEAVY(4, 4);
#undef FHX_HEAVY
#undef FHX_HEAVY_N
        FHX_HIP(hipEventRecord(ctx->ev[7], ctx->stream));
        hipLaunchKernelGGL(k2h_generic, dim3(256 * 4), dim3(K2_THREADS), 0, ctx->stream, P, (const QEntry*)ctx->d_queue_sorted,
                           (const unsigned int*)ctx->d_k2h_off, (const unsigned int*)ctx->d_digit_total, (const QEntry*)hq,
                           (const unsigned long long*)n_redo);
    }
    static const int cf_wpe = std::getenv("FHX_CF_WAVES") ? std::atoi(std::getenv("FHX_CF_WAVES")) : 0;              // measurements
#define FHX_LAUNCH_QUEUE_BY_COUNT(CLS)                                                                                                \
    do {                                                                                                                              \
        if (small_n)                                                                                                                  \
            hipLaunchKernelGGL(HIP_KERNEL_NAME(k2_queue_by_count<CLS, true, 4>), qgrid, qblock, 0, ctx->stream, P, Q.q[(CLS) - 1]);      \
        else if (cf_wpe == 4)                                                                                                         \
            hipLaunchKernelGGL(HIP_KERNEL_NAME(k2_queue_by_count<CLS, false, 4>), qgrid, qblock, 0, ctx->stream, P, Q.q[(CLS) - 1]);     \
        else                                                                                                                          \
            hipLaunchKernelGGL(HIP_KERNEL_NAME(k2_queue_by_count<CLS, false, 5>), qgrid, qblock, 0, ctx->stream, P, Q.q[(CLS) - 1]);     \
    } while (0)
    FHX_LAUNCH_QUEUE_BY_COUNT(dev::BC_CF_BD);
    FHX_LAUNCH_QUEUE_BY_COUNT(dev::BC_CF_BCF);
    if (legacy_heavy) FHX_LAUNCH_QUEUE(dev::BC_PSERIES);
#undef FHX_LAUNCH_QUEUE_BY_COUNT
#undef FHX_LAUNCH_QUEUE
    if (memo_cap >= 0) {
        unsigned long long* n_over = ctx->d_misc + 5;
        FHX_HIP(hipMemsetAsync(n_over, 0, sizeof(unsigned long long), ctx->stream));
        hipLaunchKernelGGL(k2_memo_gather, dim3(grid_for(ctx->n_rows, 256, 256 * 16)), dim3(256), 0, ctx->stream, P_rows,
                           (const double*)v_table, memo_nd, memo_cap, (int)memo_intra, (int)memo_inter, over_rows, n_over, over_cap);
        FHX_HIP(hipGetLastError());
        unsigned long long h_over = 0;
        FHX_HIP(hipMemcpyAsync(&h_over, n_over, sizeof(h_over), hipMemcpyDeviceToHost, ctx->stream));
        FHX_HIP(hipStreamSynchronize(ctx->stream));
        if (h_over > 0) {
            const bool listed = h_over <= over_cap;
            const int64_t work = listed ? (int64_t)h_over : ctx->n_rows;
            hipLaunchKernelGGL(k2_memo_overflow, dim3(grid_for(work, 256, 256 * 16)), dim3(256), 0, ctx->stream, P_rows,
                               (const unsigned int*)over_rows, (int64_t)h_over, listed ? 1 : 0);
            FHX_HIP(hipGetLastError());
            FHX_HIP(hipStreamSynchronize(ctx->stream));
        }
    }
    FHX_HIP(hipGetLastError());
    FHX_HIP(hipEventRecord(ctx->ev[3], ctx->stream));
    ctx->ev_valid[1] = true;
    ctx->have_p = true;
    ctx->have_q = false;
    ctx->n_sorted = -1;
    {
        const unsigned long long one = KEY_KEEP_ALL;               // until a cutoff is computed: keep every p
        FHX_HIP(hipMemcpyAsync(ctx->d_misc + 6, &one, sizeof(one), hipMemcpyHostToDevice, ctx->stream));
    }
    return FHX_OK;
}

// compact p < 1 and LSD-radix-sort (key, row); returns the index (0/1) of the buffer pair holding the result
// cutoff key from a device-local histogram of the p-values (single-GPU path; sharded runs all-reduce the histogram)
// d_top_hist <- key histogram of the context's p: what K2 gathered while storing them, else one more read of p
static int fill_top_hist(fhx_ctx* ctx) {
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

static int auto_cutoff(fhx_ctx* ctx, const double* d_p, int64_t n, double n_total_tests, unsigned long long* d_cutoff) {
    if (d_p == ctx->d_p) {
        const int rc = fill_top_hist(ctx);
        if (rc != FHX_OK) return rc;
    } else {
        FHX_HIP(hipMemsetAsync(ctx->d_top_hist, 0, TOP_BINS * sizeof(unsigned long long), ctx->stream));
        hipLaunchKernelGGL(k3_top_hist, dim3(grid_for((n + 1) / 2, 512, 256 * 4)), dim3(512), 0, ctx->stream, d_p, n, ctx->d_top_hist);
    }
    hipLaunchKernelGGL(k3_cutoff, dim3(1), dim3(1024), 0, ctx->stream, (const unsigned long long*)ctx->d_top_hist, n_total_tests,
                       d_cutoff);
    FHX_HIP(hipGetLastError());
    return FHX_OK;
}

// rows below the cutoff -> keys[0] / vals[0] (their number in *counter and, read back, in *n_kept); every other row gets its q here
static int compact_pvalues(fhx_ctx* ctx, const double* d_p, int64_t n, unsigned long long* keys[2], unsigned int* vals[2], double* d_q,
                           unsigned long long* counter, const unsigned long long* d_cutoff, int64_t* n_kept_out) {
    FHX_HIP(hipMemsetAsync(counter, 0, sizeof(unsigned long long), ctx->stream));
    // one workgroup per tile, not a resident grid walking the column: 0.507 -> 0.451 ms on C3 (profiles/r03_x_k3_grid.txt); the
    // plain copy kernel of profiles/hbm_rate.hip shows the same (4.9 TB/s with 2048 grid-striding workgroups, 5.6 with one per
    // chunk).  FHX_K3_GRID caps the grid for measurements.
    static const int k3_cap = std::getenv("FHX_K3_GRID") ? std::atoi(std::getenv("FHX_K3_GRID")) : (1 << 30);
    hipLaunchKernelGGL(k3_compact, dim3(grid_for(n, SORT_TILE, k3_cap)), dim3(SORT_THREADS), 0, ctx->stream, d_p, n,
                       keys[0], vals[0], d_q, counter, d_cutoff);
    // how many keys survived decides the shape of the sort (one 8-byte read back: ~20 us against ~190 us of fixed cost saved)
    unsigned long long n_kept = 0;
    FHX_HIP(hipMemcpyAsync(&n_kept, counter, sizeof(n_kept), hipMemcpyDeviceToHost, ctx->stream));
    FHX_HIP(hipStreamSynchronize(ctx->stream));
    *n_kept_out = (int64_t)n_kept;
    return FHX_OK;
}

// the six radix passes over the n_kept compacted keys; the result is in buffer pair *sorted_buf
static int sort_kept(fhx_ctx* ctx, unsigned long long* keys[2], unsigned int* vals[2], const unsigned long long* counter, int64_t n_kept,
                     int* sorted_buf) {
    const int nblk = sort_blocks_for(n_kept);
    int src = 0;
    // p < 1 means the IEEE exponent field is <= 1022: bits 62 and 63 are always clear, 62 bits to sort
    for (int pass = 0; pass < SORT_PASSES; ++pass) {
        const int shift = pass * RADIX_BITS;
        hipLaunchKernelGGL(rs_count, dim3(nblk), dim3(SORT_THREADS), 0, ctx->stream, keys[src], counter, shift,
                           ctx->d_block_hist);
        hipLaunchKernelGGL(rs_scan, dim3(RADIX), dim3((nblk + 63) / 64 * 64), 0, ctx->stream, ctx->d_block_hist, ctx->d_digit_total, nblk);
        launch_rs_scatter(ctx, nblk, keys[src], vals[src], keys[1 - src], vals[1 - src], counter, shift);
        src = 1 - src;
    }
    FHX_HIP(hipGetLastError());
    *sorted_buf = src;
    return FHX_OK;
}

static int sort_pvalues(fhx_ctx* ctx, const double* d_p, int64_t n, unsigned long long* keys[2], unsigned int* vals[2],
                        double* d_q, unsigned long long* counter, const unsigned long long* d_cutoff, int* sorted_buf,
                        int64_t* n_sorted_out = nullptr) {
    int64_t n_kept = 0;
    const int rc = compact_pvalues(ctx, d_p, n, keys, vals, d_q, counter, d_cutoff, &n_kept);
    if (rc != FHX_OK) return rc;
    if (n_sorted_out) *n_sorted_out = n_kept;
    return sort_kept(ctx, keys, vals, counter, n_kept, sorted_buf);
}

static int bh_from_sorted(fhx_ctx* ctx, const unsigned long long* keys, const unsigned int* vals, int64_t n_rows,
                          const unsigned long long* counter, double n_total_tests, double* tile_max, double* d_q) {
    const int tiles = (int)std::max<int64_t>(1, (n_rows + BH_TILE - 1) / BH_TILE);
    hipLaunchKernelGGL(bh_tile_max, dim3(tiles), dim3(BH_THREADS), 0, ctx->stream, keys, counter, (int64_t)0, n_total_tests,
                       0.0, tile_max);
    hipLaunchKernelGGL(bh_scan_tiles, dim3(1), dim3(1024), 0, ctx->stream, tile_max, counter, (int64_t)0, 0.0,
                       (double*)nullptr);
    hipLaunchKernelGGL(bh_apply, dim3(tiles), dim3(BH_THREADS), 0, ctx->stream, keys, vals, counter, (int64_t)0,
                       n_total_tests, 0.0, tile_max, (const double*)nullptr, d_q);
    FHX_HIP(hipGetLastError());
    return FHX_OK;
}

// compaction, sort and BH of one p column -> q in row order
static int rank_and_adjust(fhx_ctx* ctx, const double* d_p, int64_t n, unsigned long long* keys[2], unsigned int* vals[2], double* d_q,
                           unsigned long long* counter, const unsigned long long* d_cutoff, double n_total_tests, double* tile_max,
                           int* sorted_buf, int64_t* n_sorted_out) {
    int64_t n_kept = 0;
    int rc = compact_pvalues(ctx, d_p, n, keys, vals, d_q, counter, d_cutoff, &n_kept);
    if (rc != FHX_OK) return rc;
    if (n_sorted_out) *n_sorted_out = n_kept;
    rc = sort_kept(ctx, keys, vals, counter, n_kept, sorted_buf);
    if (rc != FHX_OK) return rc;
    return bh_from_sorted(ctx, keys[*sorted_buf], vals[*sorted_buf], n, counter, n_total_tests, tile_max, d_q);
}

static int ensure_sort_scratch(fhx_ctx* ctx) {
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
                       ctx->d_misc + 6);
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

int fhx_bh_local_sort(fhx_ctx* ctx) {
    if (!ctx) return FHX_ERR_ARG;
    if (ctx->device < 0) return fail(ctx, FHX_ERR_NO_DEVICE, "host-only context");
    if (!ctx->have_p) return fail(ctx, FHX_ERR_ARG, "fhx_pvalues must run first");
    FHX_HIP(hipSetDevice(ctx->device));
    int64_t kept = 0;
    const int rc = sort_pvalues(ctx, ctx->d_p, ctx->n_rows, ctx->d_keys, ctx->d_vals, ctx->d_q, ctx->d_misc, ctx->d_misc + 6,
                                &ctx->sorted_buf, &kept);
    if (rc != FHX_OK) return rc;
    ctx->n_sorted = kept;
    return FHX_OK;
}

int fhx_bdtrc_array(fhx_ctx* ctx, double n_total, const int32_t* count, const double* prior, int64_t n, double* out) {
    if (!ctx || !count || !prior || !out || n < 0) return FHX_ERR_ARG;
    if (ctx->device < 0) return fail(ctx, FHX_ERR_NO_DEVICE, "host-only context");
    if (n == 0) return FHX_OK;
    FHX_HIP(hipSetDevice(ctx->device));
    int64_t mc = 1;
    for (int64_t i = 0; i < n; ++i) mc = std::max<int64_t>(mc, count[i]);
    std::vector<double> lb, ib;
    build_lbeta_table(n_total, mc, lb, ib);
    double *d_lb = nullptr, *d_ib = nullptr, *d_prior = nullptr, *d_out = nullptr;
    int32_t* d_count = nullptr;
    FHX_HIP(hipMalloc(&d_lb, lb.size() * sizeof(double)));
    FHX_HIP(hipMalloc(&d_ib, ib.size() * sizeof(double)));
    FHX_HIP(hipMalloc(&d_prior, (size_t)n * sizeof(double)));
    FHX_HIP(hipMalloc(&d_out, (size_t)n * sizeof(double)));
    FHX_HIP(hipMalloc(&d_count, (size_t)n * sizeof(int32_t)));
    FHX_HIP(hipMemcpyAsync(d_lb, lb.data(), lb.size() * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    FHX_HIP(hipMemcpyAsync(d_ib, ib.data(), ib.size() * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    FHX_HIP(hipMemcpyAsync(d_prior, prior, (size_t)n * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    FHX_HIP(hipMemcpyAsync(d_count, count, (size_t)n * sizeof(int32_t), hipMemcpyHostToDevice, ctx->stream));
    const dev::BinomTables T{d_lb, d_ib, n_total, (n_total + 1.0) < dev::kMaxGam};
    hipLaunchKernelGGL(k_bdtrc_array, dim3(grid_for(n, 256)), dim3(256), 0, ctx->stream, T, d_count, d_prior, n, d_out);
    FHX_HIP(hipGetLastError());
    FHX_HIP(hipMemcpyAsync(out, d_out, (size_t)n * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    FHX_HIP(hipStreamSynchronize(ctx->stream));
    dev_free(d_lb);
    dev_free(d_ib);
    dev_free(d_prior);
    dev_free(d_out);
    dev_free(d_count);
    return FHX_OK;
}

int fhx_debug_contfrac(fhx_ctx* ctx, int kind, int lazy, const double* a, const double* b, const double* x, int64_t n,
                       double* out) {
    if (!ctx || !a || !b || !x || !out || n < 0 || kind < 0 || kind > 1) return FHX_ERR_ARG;
    if (ctx->device < 0) return fail(ctx, FHX_ERR_NO_DEVICE, "host-only context");
    if (n == 0) return FHX_OK;
    FHX_HIP(hipSetDevice(ctx->device));
    double* d[4] = {nullptr, nullptr, nullptr, nullptr};
    const double* h[3] = {a, b, x};
    const size_t bytes = (size_t)n * sizeof(double);
    for (int k = 0; k < 4; ++k) FHX_HIP(hipMalloc(&d[k], bytes));
    for (int k = 0; k < 3; ++k) FHX_HIP(hipMemcpyAsync(d[k], h[k], bytes, hipMemcpyHostToDevice, ctx->stream));
    const dim3 g(grid_for(n, 256)), t(256);
    if (kind == 0 && !lazy) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_debug_contfrac<0, 0>), g, t, 0, ctx->stream, d[0], d[1], d[2], n, d[3]);
    if (kind == 0 && lazy) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_debug_contfrac<0, 1>), g, t, 0, ctx->stream, d[0], d[1], d[2], n, d[3]);
    if (kind == 1 && !lazy) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_debug_contfrac<1, 0>), g, t, 0, ctx->stream, d[0], d[1], d[2], n, d[3]);
    if (kind == 1 && lazy) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_debug_contfrac<1, 1>), g, t, 0, ctx->stream, d[0], d[1], d[2], n, d[3]);
    FHX_HIP(hipGetLastError());
    FHX_HIP(hipMemcpyAsync(out, d[3], bytes, hipMemcpyDeviceToHost, ctx->stream));
    FHX_HIP(hipStreamSynchronize(ctx->stream));
    for (int k = 0; k < 4; ++k) dev_free(d[k]);
    return FHX_OK;
}

int fhx_debug_classify(fhx_ctx* ctx, double n_total, const int32_t* count, const double* prior, int64_t n, int32_t* by_table,
                       int32_t* by_arith, double* thr5) {
    if (!ctx || n < 0 || (n > 0 && (!count || !prior || !by_table || !by_arith))) return FHX_ERR_ARG;
    if (ctx->device < 0) return fail(ctx, FHX_ERR_NO_DEVICE, "host-only context");
    if (n == 0) return FHX_OK;
    FHX_HIP(hipSetDevice(ctx->device));
    DeviceScratch G;
    int32_t *d_c = nullptr, *d_t = nullptr, *d_a = nullptr;
    double *d_p = nullptr, *d_thr = nullptr;
    FHX_HIP(G.get(&d_c, (size_t)n * 4));
    FHX_HIP(G.get(&d_t, (size_t)n * 4));
    FHX_HIP(G.get(&d_a, (size_t)n * 4));
    FHX_HIP(G.get(&d_p, (size_t)n * 8));
    if (thr5) FHX_HIP(G.get(&d_thr, (size_t)n * 40));
    FHX_HIP(hipMemcpyAsync(d_c, count, (size_t)n * 4, hipMemcpyHostToDevice, ctx->stream));
    FHX_HIP(hipMemcpyAsync(d_p, prior, (size_t)n * 8, hipMemcpyHostToDevice, ctx->stream));
    hipLaunchKernelGGL(k_debug_classify, dim3(grid_for(n, 128)), dim3(128), 0, ctx->stream, n_total, (const int32_t*)d_c, (const double*)d_p, n,
                       d_t, d_a, d_thr);
    FHX_HIP(hipGetLastError());
    FHX_HIP(hipMemcpyAsync(by_table, d_t, (size_t)n * 4, hipMemcpyDeviceToHost, ctx->stream));
    FHX_HIP(hipMemcpyAsync(by_arith, d_a, (size_t)n * 4, hipMemcpyDeviceToHost, ctx->stream));
    if (thr5) FHX_HIP(hipMemcpyAsync(thr5, d_thr, (size_t)n * 40, hipMemcpyDeviceToHost, ctx->stream));
    FHX_HIP(hipStreamSynchronize(ctx->stream));
    return FHX_OK;
}

int fhx_debug_lean_div(fhx_ctx* ctx, const double* n, const double* d, int64_t len, double* out) {
    if (!ctx || !n || !d || !out || len < 0) return FHX_ERR_ARG;
    if (ctx->device < 0) return fail(ctx, FHX_ERR_NO_DEVICE, "host-only context");
    if (len == 0) return FHX_OK;
    FHX_HIP(hipSetDevice(ctx->device));
    double* dv[3] = {nullptr, nullptr, nullptr};
    const size_t bytes = (size_t)len * sizeof(double);
    for (int k = 0; k < 3; ++k) FHX_HIP(hipMalloc(&dv[k], bytes));
    FHX_HIP(hipMemcpyAsync(dv[0], n, bytes, hipMemcpyHostToDevice, ctx->stream));
    FHX_HIP(hipMemcpyAsync(dv[1], d, bytes, hipMemcpyHostToDevice, ctx->stream));
    hipLaunchKernelGGL(k_debug_lean_div, dim3(grid_for(len, 256)), dim3(256), 0, ctx->stream, dv[0], dv[1], len, dv[2]);
    FHX_HIP(hipGetLastError());
    FHX_HIP(hipMemcpyAsync(out, dv[2], bytes, hipMemcpyDeviceToHost, ctx->stream));
    FHX_HIP(hipStreamSynchronize(ctx->stream));
    for (int k = 0; k < 3; ++k) dev_free(dv[k]);
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
    FHX_HIP(hipMemcpyAsync(d_p, p, cap * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    int buf = 0;
    unsigned long long* counter = ctx->d_misc + 2;
    unsigned long long* cutoff = ctx->d_misc + 7;
    rc = auto_cutoff(ctx, d_p, n, n_total_tests, cutoff);
    if (rc == FHX_OK) rc = rank_and_adjust(ctx, d_p, n, keys, vals, d_q, counter, cutoff, n_total_tests, tile_max, &buf, nullptr);
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
    FHX_HIP(hipEventRecord(ctx->ev[4], ctx->stream));
    int rc = auto_cutoff(ctx, ctx->d_p, ctx->n_rows, n_total_tests, ctx->d_misc + 6);
    if (rc != FHX_OK) return rc;
    int64_t kept = 0;
    rc = rank_and_adjust(ctx, ctx->d_p, ctx->n_rows, ctx->d_keys, ctx->d_vals, ctx->d_q, ctx->d_misc, ctx->d_misc + 6, n_total_tests,
                         ctx->d_tile_max, &ctx->sorted_buf, &kept);
    if (rc != FHX_OK) return rc;
    ctx->n_sorted = kept;
    FHX_HIP(hipEventRecord(ctx->ev[5], ctx->stream));
    ctx->ev_valid[2] = true;
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
    FHX_HIP(hipMalloc(&keys[0], (size_t)n * sizeof(unsigned long long)));
    FHX_HIP(hipMalloc(&vals[0], (size_t)n * sizeof(unsigned int)));
    unsigned long long* counter = ctx->d_misc + 3;
    const unsigned long long n_host = (unsigned long long)n;
    FHX_HIP(hipMemcpyAsync(counter, &n_host, sizeof(n_host), hipMemcpyHostToDevice, ctx->stream));
    // an even number of ping-pong passes: start in the caller's output pair so that the result lands there
    FHX_HIP(hipMemcpyAsync(keys[1], d_keys_in, (size_t)n * sizeof(unsigned long long), hipMemcpyDeviceToDevice, ctx->stream));
    hipLaunchKernelGGL(k_iota_u32, dim3(grid_for(n, 256)), dim3(256), 0, ctx->stream, vals[1], n);
    const int nblk = sort_blocks_for(n);
    int src = 1;
    for (int pass = 0; pass < SORT_PASSES; ++pass) {
        const int shift = pass * RADIX_BITS;
        hipLaunchKernelGGL(rs_count, dim3(nblk), dim3(SORT_THREADS), 0, ctx->stream, keys[src], counter, shift,
                           ctx->d_block_hist);
        hipLaunchKernelGGL(rs_scan, dim3(RADIX), dim3((nblk + 63) / 64 * 64), 0, ctx->stream, ctx->d_block_hist, ctx->d_digit_total, nblk);
        launch_rs_scatter(ctx, nblk, keys[src], vals[src], keys[1 - src], vals[1 - src], counter, shift);
        src = 1 - src;
    }
    FHX_HIP(hipGetLastError());
    FHX_HIP(hipStreamSynchronize(ctx->stream));
    dev_free(keys[0]);
    dev_free(vals[0]);
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

int fhx_memcpy_d2d(fhx_ctx* ctx, void* dst, const void* src, int64_t bytes) {
    if (!ctx || bytes < 0 || (bytes > 0 && (!dst || !src))) return FHX_ERR_ARG;
    if (ctx->device < 0) return fail(ctx, FHX_ERR_NO_DEVICE, "host-only context");
    FHX_HIP(hipSetDevice(ctx->device));
    if (bytes) FHX_HIP(hipMemcpyAsync(dst, src, (size_t)bytes, hipMemcpyDeviceToDevice, ctx->stream));
    FHX_HIP(hipStreamSynchronize(ctx->stream));
    return FHX_OK;
}

int fhx_sync(fhx_ctx* ctx) {
    if (!ctx) return FHX_ERR_ARG;
    if (ctx->device < 0) return FHX_OK;
    FHX_HIP(hipSetDevice(ctx->device));
    FHX_HIP(hipStreamSynchronize(ctx->stream));
    return FHX_OK;
}

int fhx_set_global_rows(fhx_ctx* ctx, const int64_t* rows, int64_t n) {
    if (!ctx || n < 0 || (n > 0 && !rows)) return FHX_ERR_ARG;
    if (ctx->device < 0) return fail(ctx, FHX_ERR_NO_DEVICE, "host-only context");
    if (n != ctx->n_rows) return fail(ctx, FHX_ERR_ARG, "one file position per loaded contact row is required");
    FHX_HIP(hipSetDevice(ctx->device));
    dev_free(ctx->d_grow);
    const size_t cap = std::max<size_t>(4, ((size_t)n + 3) / 4 * 4);
    FHX_HIP(hipMalloc(&ctx->d_grow, cap * sizeof(long long)));
    FHX_HIP(hipMemsetAsync(ctx->d_grow, 0, cap * sizeof(long long), ctx->stream));
    if (n) FHX_HIP(hipMemcpyAsync(ctx->d_grow, rows, (size_t)n * sizeof(long long), hipMemcpyHostToDevice, ctx->stream));
    FHX_HIP(hipStreamSynchronize(ctx->stream));
    return FHX_OK;
}

int64_t fhx_get_skip_limit(fhx_ctx* ctx) { return ctx ? ctx->skip_limit : INT64_MAX; }

int fhx_set_skip_limit(fhx_ctx* ctx, int64_t limit) {
    if (!ctx) return FHX_ERR_ARG;
    ctx->skip_limit = limit;
    return FHX_OK;
}

int fhx_next_pass(fhx_ctx* ctx, int64_t* n_outliers_total) {
    if (!ctx) return FHX_ERR_ARG;
    if (ctx->device < 0) return fail(ctx, FHX_ERR_NO_DEVICE, "host-only context");
    if (!ctx->have_p) return fail(ctx, FHX_ERR_ARG, "fhx_pvalues must run first");
    FHX_HIP(hipSetDevice(ctx->device));
    unsigned long long* n_out = ctx->d_misc + 1;
    unsigned long long* first_dup = ctx->d_misc + 4;
    FHX_HIP(hipMemsetAsync(n_out, 0, sizeof(unsigned long long), ctx->stream));
    FHX_HIP(hipMemsetAsync(first_dup, 0xFF, sizeof(unsigned long long), ctx->stream));
    if (ctx->nonfixed) {
        // the distances of this pass's outliers go to a list (reusing the sort workspace), then into the sorted multiset
        hipLaunchKernelGGL(nf_fold_outliers, dim3(grid_for(ctx->n_rows, 256)), dim3(256), 0, ctx->stream, ctx->d_loc1, ctx->d_loc2,
                           ctx->d_p, 1.0 / ctx->fit.bh_total_tests, ctx->d_skip, ctx->d_seen_twice, ctx->n_rows,
                           (const int32_t*)ctx->d_slot_mid, ctx->d_keys[0], n_out, first_dup, (const long long*)ctx->d_grow);
        FHX_HIP(hipGetLastError());
        unsigned long long added = 0, dup = ~0ull;
        FHX_HIP(hipMemcpyAsync(&added, n_out, sizeof(added), hipMemcpyDeviceToHost, ctx->stream));
        FHX_HIP(hipMemcpyAsync(&dup, first_dup, sizeof(dup), hipMemcpyDeviceToHost, ctx->stream));
        FHX_HIP(hipStreamSynchronize(ctx->stream));
        std::vector<int64_t> fresh((size_t)added);
        if (added) {
            FHX_HIP(hipMemcpyAsync(fresh.data(), ctx->d_keys[0], (size_t)added * sizeof(int64_t), hipMemcpyDeviceToHost, ctx->stream));
            FHX_HIP(hipStreamSynchronize(ctx->stream));
        }
        ctx->h_outlier_dists.insert(ctx->h_outlier_dists.end(), fresh.begin(), fresh.end());
        std::sort(ctx->h_outlier_dists.begin(), ctx->h_outlier_dists.end());
        ctx->n_outliers_total += (int64_t)added;
        if (dup != ~0ull) ctx->skip_limit = std::min<int64_t>(ctx->skip_limit, (int64_t)dup);
        ctx->skip_active = true;
        ctx->pass_no += 1;
        if (n_outliers_total) *n_outliers_total = ctx->n_outliers_total;
        return FHX_OK;
    }
    hipLaunchKernelGGL(k_fold_outliers, dim3(grid_for(ctx->n_rows, 256)), dim3(256), 0, ctx->stream, ctx->d_loc1, ctx->d_loc2,
                       ctx->d_p, 1.0 / ctx->fit.bh_total_tests, ctx->d_skip, ctx->d_seen_twice, ctx->n_rows, (int)ctx->prm.resolution, (int)ctx->n_dist,
                       ctx->d_slot_chr, ctx->d_grid, ctx->d_out_hist, n_out, first_dup, (const long long*)ctx->d_grow);
    FHX_HIP(hipGetLastError());
    unsigned long long added = 0, dup = ~0ull;
    FHX_HIP(hipMemcpyAsync(&dup, first_dup, sizeof(dup), hipMemcpyDeviceToHost, ctx->stream));
    ctx->h_out_hist.assign((size_t)ctx->n_dist, 0);
    FHX_HIP(hipMemcpyAsync(&added, n_out, sizeof(added), hipMemcpyDeviceToHost, ctx->stream));
    FHX_HIP(hipMemcpyAsync(ctx->h_out_hist.data(), ctx->d_out_hist, ctx->n_dist * sizeof(int64_t), hipMemcpyDeviceToHost,
                           ctx->stream));
    FHX_HIP(hipStreamSynchronize(ctx->stream));
    ctx->n_outliers_total += (int64_t)added;
    if (dup != ~0ull) ctx->skip_limit = std::min<int64_t>(ctx->skip_limit, (int64_t)dup);
    ctx->skip_active = true;
    ctx->pass_no += 1;
    if (n_outliers_total) *n_outliers_total = ctx->n_outliers_total;
    return FHX_OK;
}

int fhx_reset_passes(fhx_ctx* ctx) {
    if (!ctx) return FHX_ERR_ARG;
    if (ctx->device < 0) return fail(ctx, FHX_ERR_NO_DEVICE, "host-only context");
    if (!ctx->d_loc1) return fail(ctx, FHX_ERR_ARG, "no contact rows loaded");
    FHX_HIP(hipSetDevice(ctx->device));
    const size_t cap = std::max<size_t>(4, ((size_t)ctx->n_rows + 3) / 4 * 4);
    FHX_HIP(hipMemsetAsync(ctx->d_skip, 0, cap, ctx->stream));
    FHX_HIP(hipMemsetAsync(ctx->d_outlier, 0, cap, ctx->stream));
    FHX_HIP(hipMemsetAsync(ctx->d_seen_twice, 0, cap, ctx->stream));
    const size_t hist_len = ctx->nonfixed ? cap : (size_t)ctx->n_dist;
    FHX_HIP(hipMemsetAsync(ctx->d_out_hist, 0, hist_len * sizeof(unsigned long long), ctx->stream));
    FHX_HIP(hipStreamSynchronize(ctx->stream));
    ctx->pass_no = 0;
    ctx->skip_active = false;
    ctx->have_stats = ctx->have_fit = ctx->have_bins = ctx->have_p = ctx->have_q = false;
    ctx->n_outliers_total = 0;
    ctx->skip_limit = INT64_MAX;
    ctx->outlier_hist_nonempty = false;
    ctx->h_out_hist.assign((size_t)ctx->n_dist, 0);
    ctx->h_outlier_dists.clear();
    ctx->h_outlier_dists_global.clear();
    ctx->outlier_dists_are_global = false;
    ctx->n_sorted = -1;
    return FHX_OK;
}

int fhx_fetch(fhx_ctx* ctx, double* p, double* q, double* expcc, double* bias1, double* bias2) {
    if (!ctx) return FHX_ERR_ARG;
    if (ctx->device < 0) return fail(ctx, FHX_ERR_NO_DEVICE, "host-only context");
    if (!ctx->have_p) return fail(ctx, FHX_ERR_ARG, "no p-values yet");
    if (q && !ctx->have_q) return fail(ctx, FHX_ERR_ARG, "no q-values yet");
    FHX_HIP(hipSetDevice(ctx->device));
    const size_t bytes = (size_t)ctx->n_rows * sizeof(double);
    if (p) FHX_HIP(hipMemcpyAsync(p, ctx->d_p, bytes, hipMemcpyDeviceToHost, ctx->stream));
    if (q) FHX_HIP(hipMemcpyAsync(q, ctx->d_q, bytes, hipMemcpyDeviceToHost, ctx->stream));
    double* d_tmp[3] = {nullptr, nullptr, nullptr};
    double* host[3] = {expcc, bias1, bias2};
    DeviceScratch tmp;
    if (expcc || bias1 || bias2) {
        for (int k = 0; k < 3; ++k)
            if (host[k]) FHX_HIP(tmp.get(&d_tmp[k], bytes));
        const K2Params P = make_k2_params(ctx);
        hipLaunchKernelGGL(k2_extras, dim3(grid_for(ctx->n_rows, 256)), dim3(256), 0, ctx->stream, P, ctx->prm.bias_low,
                           ctx->prm.bias_up, d_tmp[0], d_tmp[1], d_tmp[2]);
        FHX_HIP(hipGetLastError());
        for (int k = 0; k < 3; ++k)
            if (host[k]) FHX_HIP(hipMemcpyAsync(host[k], d_tmp[k], bytes, hipMemcpyDeviceToHost, ctx->stream));
    }
    FHX_HIP(hipStreamSynchronize(ctx->stream));
    return FHX_OK;
}

int fhx_fetch_flags(fhx_ctx* ctx, uint8_t* outlier, uint8_t* skip) {
    if (!ctx) return FHX_ERR_ARG;
    if (ctx->device < 0) return fail(ctx, FHX_ERR_NO_DEVICE, "host-only context");
    if (!ctx->d_loc1) return fail(ctx, FHX_ERR_ARG, "no contact rows loaded");
    if (outlier && !ctx->have_p) return fail(ctx, FHX_ERR_ARG, "no p-values yet");
    FHX_HIP(hipSetDevice(ctx->device));
    if (outlier) {
        hipLaunchKernelGGL(k_outlier_flags, dim3(grid_for(ctx->n_rows, 256)), dim3(256), 0, ctx->stream, ctx->d_p,
                           1.0 / ctx->fit.bh_total_tests, ctx->n_rows, ctx->d_outlier);
        FHX_HIP(hipGetLastError());
    }
    if (outlier) FHX_HIP(hipMemcpyAsync(outlier, ctx->d_outlier, (size_t)ctx->n_rows, hipMemcpyDeviceToHost, ctx->stream));
    if (skip) FHX_HIP(hipMemcpyAsync(skip, ctx->d_skip, (size_t)ctx->n_rows, hipMemcpyDeviceToHost, ctx->stream));
    FHX_HIP(hipStreamSynchronize(ctx->stream));
    return FHX_OK;
}

// The row numbers of the outlier lines, ascending, compacted on the device (a flag byte per row back to the host and a
// flatnonzero over 1.5e8 bytes cost the command line 0.2 s for 3e4 outliers).  rows == NULL or cap < *n_out: only the count.
int fhx_fetch_outlier_rows(fhx_ctx* ctx, int64_t* rows, int64_t cap, int64_t* n_out) {
    if (!ctx || !n_out || cap < 0 || (cap > 0 && !rows)) return FHX_ERR_ARG;
    if (ctx->device < 0) return fail(ctx, FHX_ERR_NO_DEVICE, "host-only context");
    if (!ctx->d_loc1) return fail(ctx, FHX_ERR_ARG, "no contact rows loaded");
    if (!ctx->have_p) return fail(ctx, FHX_ERR_ARG, "no p-values yet");
    FHX_HIP(hipSetDevice(ctx->device));
    const int64_t n = ctx->n_rows;
    *n_out = 0;
    if (n == 0) return FHX_OK;
    hipLaunchKernelGGL(k_outlier_flags, dim3(grid_for(n, 256)), dim3(256), 0, ctx->stream, ctx->d_p, 1.0 / ctx->fit.bh_total_tests, n,
                       ctx->d_outlier);
    const int64_t n_tiles = (n + fhxscan::TILE - 1) / fhxscan::TILE;
    DeviceScratch tmp;
    unsigned int* d_counts = nullptr;
    unsigned long long *d_offsets = nullptr, *d_total = nullptr;
    int64_t* d_rows = nullptr;
    FHX_HIP(tmp.get(&d_counts, (size_t)n_tiles * sizeof(unsigned int)));
    FHX_HIP(tmp.get(&d_offsets, (size_t)n_tiles * sizeof(unsigned long long)));
    FHX_HIP(tmp.get(&d_total, sizeof(unsigned long long)));
    hipLaunchKernelGGL(k_flag_count, dim3((unsigned)n_tiles), dim3(fhxscan::THREADS), 0, ctx->stream, (const unsigned char*)ctx->d_outlier, n,
                       d_counts);
    hipLaunchKernelGGL(fhxscan::scan_tiles, dim3(1), dim3(fhxscan::THREADS), 0, ctx->stream, (const unsigned int*)d_counts, n_tiles, d_offsets,
                       d_total);
    unsigned long long total = 0;
    FHX_HIP(hipMemcpyAsync(&total, d_total, sizeof(total), hipMemcpyDeviceToHost, ctx->stream));
    FHX_HIP(hipStreamSynchronize(ctx->stream));
    *n_out = (int64_t)total;
    if (!rows || (int64_t)total > cap || total == 0) return FHX_OK;
    FHX_HIP(tmp.get(&d_rows, (size_t)total * sizeof(int64_t)));
    hipLaunchKernelGGL(k_flag_rows, dim3((unsigned)n_tiles), dim3(fhxscan::THREADS), 0, ctx->stream, (const unsigned char*)ctx->d_outlier, n,
                       (const unsigned long long*)d_offsets, d_rows);
    FHX_HIP(hipMemcpyAsync(rows, d_rows, (size_t)total * sizeof(int64_t), hipMemcpyDeviceToHost, ctx->stream));
    FHX_HIP(hipStreamSynchronize(ctx->stream));
    return FHX_OK;
}

int fhx_get_array(fhx_ctx* ctx, int which, void* dst, int64_t cap, int64_t* n_out) {
    if (!ctx) return FHX_ERR_ARG;
    const PassFit& f = ctx->fit;
    auto put = [&](const void* src, size_t n, size_t elem) -> int {
        if (n_out) *n_out = (int64_t)n;
        if (!dst) return FHX_OK;
        if ((int64_t)n > cap) return fail(ctx, FHX_ERR_ARG, "destination too small");
        if (n) std::memcpy(dst, src, n * elem);
        return FHX_OK;
    };
    auto bin_i64 = [&](int64_t Bin::*m) -> int {
        std::vector<int64_t> v;
        for (const auto& b : f.bins) v.push_back(b.*m);
        return put(v.data(), v.size(), sizeof(int64_t));
    };
    switch (which) {
        case FHX_A_HIST_SUMCC: return put(ctx->h_hist_cc.data(), ctx->h_hist_cc.size(), sizeof(int64_t));
        case FHX_A_HIST_NPAIRS: return put(ctx->h_hist_np.data(), ctx->h_hist_np.size(), sizeof(int64_t));
        case FHX_A_OUTLIER_DIST_HIST: return put(ctx->h_out_hist.data(), ctx->h_out_hist.size(), sizeof(int64_t));
        case FHX_A_DIST_KEYS: return put(ctx->h_dist_keys.data(), ctx->h_dist_keys.size(), sizeof(int64_t));
        case FHX_A_OUTLIER_DISTS: {
            const std::vector<int64_t>& od = ctx->outlier_dists_are_global ? ctx->h_outlier_dists_global : ctx->h_outlier_dists;
            return put(od.data(), od.size(), sizeof(int64_t));
        }
        default: break;
    }
    if (which == FHX_A_FDR_COUNTS) {
        if (ctx->device < 0) return fail(ctx, FHX_ERR_NO_DEVICE, "host-only context");
        if (!ctx->have_q) return fail(ctx, FHX_ERR_ARG, "no q-values yet");
        FHX_HIP(hipSetDevice(ctx->device));
        unsigned long long* buckets = ctx->d_misc + 8;
        FHX_HIP(hipMemsetAsync(buckets, 0, 51 * sizeof(unsigned long long), ctx->stream));
        hipLaunchKernelGGL(k_fdr_hist, dim3(grid_for(ctx->n_rows, 256)), dim3(256), 0, ctx->stream, ctx->d_q, ctx->n_rows, buckets);
        std::vector<int64_t> c(51, 0);
        FHX_HIP(hipMemcpyAsync(c.data(), buckets, 51 * sizeof(int64_t), hipMemcpyDeviceToHost, ctx->stream));
        FHX_HIP(hipStreamSynchronize(ctx->stream));
        for (int i = 1; i < 51; ++i) c[i] += c[i - 1];       // cumulative ...
        for (int i = 50; i >= 1; --i) c[i] = c[i - 1];       // ... shifted by one (fithic.py:1249-1254)
        c[0] = 0;
        return put(c.data(), c.size(), sizeof(int64_t));
    }
    if (!ctx->have_fit && !(ctx->have_bins && (which == FHX_A_BIN_LB || which == FHX_A_BIN_UB || which == FHX_A_BIN_SUMCC ||
                                               which == FHX_A_BIN_POSS0)))
        return fail(ctx, FHX_ERR_ARG, "fhx_fit must run first");
    switch (which) {
        case FHX_A_BIN_LB: return bin_i64(&Bin::lb);
        case FHX_A_BIN_UB: return bin_i64(&Bin::ub);
        case FHX_A_BIN_POSS: return bin_i64(&Bin::poss);
        case FHX_A_BIN_POSS0: return bin_i64(&Bin::poss0);
        case FHX_A_BIN_SUMCC: return bin_i64(&Bin::sumcc);
        case FHX_A_BIN_POSS7: return bin_i64(&Bin::poss7);
        case FHX_A_BIN_SUMDIST: {
            std::vector<double> v;
            for (const auto& b : f.bins) v.push_back(b.sumdist);
            return put(v.data(), v.size(), sizeof(double));
        }
        case FHX_A_X: return put(f.x.data(), f.x.size(), sizeof(double));
        case FHX_A_Y: return put(f.y.data(), f.y.size(), sizeof(double));
        case FHX_A_KNOTS: return put(f.spline.t.data(), f.spline.t.size(), sizeof(double));
        case FHX_A_COEFFS: return put(f.spline.c.data(), f.spline.c.size(), sizeof(double));
        case FHX_A_TABLE_X: return put(f.table_x.data(), f.table_x.size(), sizeof(int64_t));
        case FHX_A_TABLE_Y0: return put(f.table_y0.data(), f.table_y0.size(), sizeof(double));
        case FHX_A_TABLE_Y: return put(f.table_y.data(), f.table_y.size(), sizeof(double));
        default: return fail(ctx, FHX_ERR_ARG, "unknown array id");
    }
}

void* fhx_device_ptr(fhx_ctx* ctx, int which) {
    if (!ctx || ctx->device < 0) return nullptr;
    switch (which) {
        case 0: return ctx->d_p;
        case 1: return ctx->d_q;
        case 2: return ctx->d_keys[ctx->sorted_buf];
        case 3: return ctx->d_vals[ctx->sorted_buf];
        case 4: return ctx->d_top_hist;                  // 8192 x u64, valid after fhx_bh_top_hist(_device)
        default: return nullptr;
    }
}

int64_t fhx_n_sorted(fhx_ctx* ctx) {
    if (!ctx || ctx->device < 0 || ctx->n_sorted == -1) return -1;
    if (ctx->n_sorted == -2) {
        unsigned long long n = 0;
        if (hipSetDevice(ctx->device) != hipSuccess) return -1;
        if (hipMemcpyAsync(&n, ctx->d_misc, sizeof(n), hipMemcpyDeviceToHost, ctx->stream) != hipSuccess) return -1;
        if (hipStreamSynchronize(ctx->stream) != hipSuccess) return -1;
        ctx->n_sorted = (int64_t)n;
    }
    return ctx->n_sorted;
}

int fhx_k2_heavy_launch(fhx_ctx* ctx, double* seconds, int64_t* rows) {
    if (!ctx) return FHX_ERR_ARG;
    if (ctx->device < 0) return fail(ctx, FHX_ERR_NO_DEVICE, "host-only context");
    if (!ctx->ev_valid[1]) return fail(ctx, FHX_ERR_ARG, "fhx_pvalues has not run");
    FHX_HIP(hipSetDevice(ctx->device));
    FHX_HIP(hipStreamSynchronize(ctx->stream));
    float ms = 0.f;
    FHX_HIP(hipEventElapsedTime(&ms, ctx->ev[6], ctx->ev[7]));
    unsigned long long n = 0;
    if (!ctx->d_k2_counts || ctx->k2_shards <= 0) return fail(ctx, FHX_ERR_ARG, "fhx_pvalues has not run");
    std::vector<unsigned long long> part((size_t)ctx->k2_shards);
    FHX_HIP(hipMemcpy(part.data(), ctx->d_k2_counts + (size_t)(dev::BC_CF_SWAPPED - 1) * K2_MAX_SHARDS, part.size() * sizeof(unsigned long long),
                      hipMemcpyDeviceToHost));
    for (unsigned long long v : part) n += v;
    if (std::getenv("FHX_DEBUG_HEAVY")) {                  // how many rows the uniform kernel handed back to the per-lane loop
        unsigned long long redo = 0;
        FHX_HIP(hipMemcpy(&redo, ctx->d_misc + 11, sizeof(redo), hipMemcpyDeviceToHost));
        std::fprintf(stderr, "k2h_heavy: %.3f ms, %llu rows in the class, %llu handed back\n", ms, n, redo);
    }
    if (seconds) *seconds = ms * 1e-3;
    if (rows) *rows = (int64_t)n;
    return FHX_OK;
}

int fhx_k2_class_rows(fhx_ctx* ctx, int64_t* out5) {
    if (!ctx || !out5) return FHX_ERR_ARG;
    if (ctx->device < 0) return fail(ctx, FHX_ERR_NO_DEVICE, "host-only context");
    if (!ctx->d_k2_counts || ctx->k2_shards <= 0 || !ctx->ev_valid[1]) return fail(ctx, FHX_ERR_ARG, "fhx_pvalues has not run");
    FHX_HIP(hipSetDevice(ctx->device));
    FHX_HIP(hipStreamSynchronize(ctx->stream));
    std::vector<unsigned long long> part((size_t)(K2_QUEUES + 1) * K2_MAX_SHARDS);
    FHX_HIP(hipMemcpy(part.data(), ctx->d_k2_counts, part.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    for (int k = 0; k <= K2_QUEUES; ++k) {
        unsigned long long n = 0;
        for (int sh = 0; sh < ctx->k2_shards; ++sh) n += part[(size_t)k * K2_MAX_SHARDS + sh];
        out5[k] = (int64_t)n;
    }
    return FHX_OK;
}

int fhx_kernel_seconds(fhx_ctx* ctx, double* k1, double* k2, double* k3) {
    if (!ctx) return FHX_ERR_ARG;
    if (ctx->device < 0) return fail(ctx, FHX_ERR_NO_DEVICE, "host-only context");
    FHX_HIP(hipSetDevice(ctx->device));
    FHX_HIP(hipStreamSynchronize(ctx->stream));
    double* out[3] = {k1, k2, k3};
    for (int k = 0; k < 3; ++k) {
        if (!out[k]) continue;
        *out[k] = 0.0;
        if (!ctx->ev_valid[k]) continue;
        float ms = 0.f;
        FHX_HIP(hipEventElapsedTime(&ms, ctx->ev[2 * k], ctx->ev[2 * k + 1]));
        *out[k] = ms * 1e-3;
    }
    return FHX_OK;
}

#include "fhx_dist.inc"
#include "fhx_emit.inc"
#include "fhx_inflate.inc"
#include "fhx_ingest.inc"

// ---- host numerics exported for tests / host-only callers ----------------------------------------------
int fhx_host_spline_fit(const double* x, const double* y, int32_t m, double s, double* t, double* c, int32_t* n_knots,
                        double* fp, int32_t* ier, int32_t* restarted) {
    if (!x || !y || !t || !c || !n_knots) return FHX_ERR_ARG;
    Spline sp;
    const int rc = spline_fit(x, y, m, s, sp);
    if (rc != FHX_OK) return rc;
    *n_knots = (int32_t)sp.t.size();
    std::memcpy(t, sp.t.data(), sp.t.size() * sizeof(double));
    std::memcpy(c, sp.c.data(), sp.c.size() * sizeof(double));
    if (fp) *fp = sp.fp;
    if (ier) *ier = sp.ier;
    if (restarted) *restarted = sp.restarted ? 1 : 0;
    return FHX_OK;
}

int fhx_host_spline_eval(const double* t, const double* c, int32_t n_knots, const double* xs, int64_t nx, double* out) {
    if (!t || !c || !xs || !out || n_knots < 8) return FHX_ERR_ARG;
    Spline sp;
    sp.t.assign(t, t + n_knots);
    sp.c.assign(c, c + n_knots - 4);
    spline_eval(sp, xs, nx, out);
    return FHX_OK;
}

int fhx_host_pava_decreasing(const double* y, int64_t n, double* out) {
    if (!y || !out || n < 0) return FHX_ERR_ARG;
    pava_decreasing(y, n, out);
    return FHX_OK;
}

int fhx_host_lbeta_table(double n_total, int64_t max_count, double* lbeta_out, double* inv_beta_out) {
    if (!lbeta_out || max_count < 0) return FHX_ERR_ARG;
    std::vector<double> lb, ib;
    build_lbeta_table(n_total, max_count, lb, ib);
    std::memcpy(lbeta_out, lb.data(), lb.size() * sizeof(double));
    if (inv_beta_out) std::memcpy(inv_beta_out, ib.data(), ib.size() * sizeof(double));
    return FHX_OK;
}

}  // extern "C"
