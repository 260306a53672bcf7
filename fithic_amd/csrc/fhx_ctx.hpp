// fhx_ctx.hpp - what the device translation units of libfithic_mi355x.so share: the context behind the C ABI, the structs and
// constants that cross kernel groups, small wave helpers, and the host-side helpers every entry point uses.
//
//   fhx_k1.hip      K0 ingest, K1 classify + histogram (fithic.read_Interactions), -r 0 slotting, row arrays
//   fhx_k2.hip      K2 per-pair prior + bdtrc (fithic.fit_Spline's pair loop), outlier bookkeeping, fetch
//   fhx_k3.hip      K3 Benjamini-Hochberg: cutoff, compaction, radix sort, scan (myStats.benjamini_hochberg_correction)
//   fhx_device.hip  context life cycle, parameters, tables, the host fit, sharded runs (fhx_dist.inc), file I/O on the device
//                   (fhx_inflate.inc, fhx_ingest.inc, fhx_emit.inc)
// The whole library is compiled with -ffp-contract=off (fhx_bdtrc.hpp says why).
#pragma once
#include <hip/hip_runtime.h>

#include <atomic>
#include <functional>
#include <chrono>
#include <thread>
#include <memory>
#include <sys/stat.h>
#include <unistd.h>
#include <deque>
#include <mutex>
#include <map>
#include <condition_variable>
#include <algorithm>
#include <climits>
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

// ---- structs and constants shared by the kernel groups --------------------------------------------------------------------
struct ChrGrid {          // per chromosome id, device copy
    int32_t base;         // first slot
    int32_t off;          // mid % res shared by the chromosome's loci (-1: chromosome unseen)
    int32_t nslots;
    int32_t pad;
};

struct K1Sums {           // device accumulator block (int64 each)
    long long inter_count, inter_sum, intra_all_count, intra_all_sum, in_range_count, in_range_sum, n_skipped;
    int max_count, pad;
};

// everything a K2 kernel needs about the pass (built by make_k2_params)
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
    // the true totals (observedIntraInRangeSum, observedInterAllSum) for ExpCC = total * prior (fithic.py:1076, 1106): the n
    // of intra / inter above is what bdtrc is given, which differs from these once a total reaches 2^31 (bdtrc_total)
    double total_intra, total_inter;
    int lean_closed;              // experiment (FHX_LEAN_CLOSED=1): count == 1 rows with prior < 0.01 through a division-free log1p
};

constexpr int K2_THREADS = 256;

// K2 runs as one classification launch plus one launch per branch class so that waves are branch-homogeneous
// (SURVEY appendix C / F): the iteration count of Cephes' incbet is multi-modal - none for the closed form, ~15 for the
// power series, ~9 for the converging continued fractions and (practically always) the full 300 for the swapped
// continued fraction ("observed < expected").  k2_classify finishes the loop-free class in place and appends every
// other row to the queue of its class (wave-aggregated: one atomic per wave and class); k2_queue then runs one
// class at a time with every lane on the same code path and nearly the same trip count.
constexpr int K2_QUEUES = dev::BC_COUNT - 1;        // classes 1..4

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
// bound by neither, profiles/history/r02_z_pmc.txt.)  A shard's tiles are known in advance (tile t belongs to workgroup t % grid), so a
// region of ceil(tiles / grid) tiles per shard can never overflow; two classes share a buffer, growing towards each other
// inside every shard's region.  The class kernels read a queue as one dense list over its shards (QDense, fhx_k2.hip: a prefix
// of the shard counts in LDS); k2h_scatter and k2_closed take whole shards.
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

// words of fhx_ctx::d_misc that K2's class kernels use, zeroed together before they are launched
constexpr int MISC_K2_REDO = 64;                         // rows k2h_heavy handed back
constexpr int MISC_K2_NEXT = 65;                         // + 0: k2h_heavy's next task; + class (1..K2_QUEUES): that kernel's next piece
constexpr int MISC_K2_WORDS = 8;
constexpr int MISC_K3_DENSE = 80;                        // k3_cutoff's decision: this pass takes q through the dense array (fhx_k3.hip: DenseQ)
constexpr int K3_DENSE_PERCENT = 35;                     // ... when at least this share of the rows survives the BH cutoff

constexpr int K2_CL_ITEMS = 4;
constexpr int K2_CL_TILE = K2_THREADS * K2_CL_ITEMS;     // 1024 rows per workgroup step: four waves of 256 consecutive rows

constexpr int K2_CLOSED = K2_QUEUES + 1;                 // count == 1 rows with prior >= 0.01 (Cephes takes pow there): queued, k2_closed
constexpr int K2_CLOSED_LOCAL = K2_QUEUES + 2;           // count == 1 rows with prior < 0.01: wave-local, evaluated densely from LDS
constexpr int K2_CLASSES = K2_QUEUES + 2;

// the 300-iteration class, bucketed by (binomial, contact count): see fhx_k2.hip
constexpr int K2H_KCAP = 1023;
constexpr int K2H_GENERIC = 2 * K2H_KCAP;               // 2046
constexpr int K2H_THREADS = 256;
constexpr int K2H_MAX_ROWS = 4;

// K3's radix sort and BH scan: see fhx_k3.hip
constexpr int SORT_THREADS = 256;
constexpr int SORT_WAVES = SORT_THREADS / 64;
constexpr int SORT_ITEMS = 16;                        // per thread
constexpr int SORT_TILE = SORT_THREADS * SORT_ITEMS;  // 4096 keys per tile
// k3_compact / k3_fill_q: tiles of 16 384 rows (one workgroup of 16 waves).  A tile costs ONE returning atomic on the survivors'
// counter, and returning atomics on one address retire at 11 ns each: with 4096-row tiles the 36 000 tiles of C3 were 0.40 of the
// kernel's 0.42 ms whatever it read or wrote (profiles/r06/k3_compact_tiles.txt)
constexpr int CP_THREADS = 1024;
constexpr int CP_WAVES = CP_THREADS / 64;
constexpr int CP_ITEMS = 16;
constexpr int CP_TILE = CP_THREADS * CP_ITEMS;
constexpr int CP_STRIP = 2048;                        // survivors a workgroup of k3_compact holds in LDS across its tiles (24 KB)
constexpr int SORT_BLOCKS = 1024;                     // persistent: 4 workgroups per CU
constexpr int RADIX_BITS = 11;                       // 6 passes cover 66 >= 64 key bits
constexpr int RADIX = 1 << RADIX_BITS;
constexpr int SORT_PASSES = 6;                        // even: the result lands in the buffer pair it started in
// Digit width of the BH sort's passes over large survivor sets (fhx_k3.hip: sort_kept): EIGHT passes of 8 bits, not six of 11.
// With 2048 digits a 4096-key tile leaves runs of two keys per digit: every 16-byte (keys) / 8-byte (payloads) run dirties a
// 32-byte sector of its own and a pass wrote 463 MB for 181 MB of data (PMC, profiles/r04_z_od1_pmc.txt); with 256 digits the
// runs are 16 keys = whole lines, and the two extra passes cost less than that: K3 over 1.5e7 survivors 2.70 -> 2.12 ms
// (11 / 10 / 9 / 8 bits: 2.70 / 3.12 / 2.15-2.25 / 2.12 ms, same digest of all p and q; profiles/history/r04_n_rs_bits_ab.txt).
constexpr int SORT_BITS_LARGE = 8;

constexpr int TOP_SHIFT = 50;
constexpr int TOP_BINS = 8192;                         // keys of p < 1 are < 2^62, so key >> 50 < 4096 (kept at 8192 for slack)

__device__ __forceinline__ unsigned long long pvalue_key(double v) {
    unsigned long long bits = (unsigned long long)__double_as_longlong(v);
    if (bits == 0x8000000000000000ull) bits = 0ull;    // -0.0 sorts with +0.0
    return bits;
}

constexpr unsigned long long KEY_ONE = 0x3FF0000000000000ull;          // bits of 1.0
constexpr unsigned long long KEY_KEEP_ALL = 0x7FF0000000000001ull;     // above +inf: no value is cut

constexpr int SCAT_ITEMS = 8;                                   // keys per thread; a tile = SCAT_THREADS x 8 keys
constexpr int BH_THREADS = 256;
constexpr int BH_ITEMS = 8;
constexpr int BH_TILE = BH_THREADS * BH_ITEMS;
// run detection of the -r 0 path
constexpr int SEG_THREADS = 256;
constexpr int SEG_ITEMS = 16;
constexpr int SEG_TILE = SEG_THREADS * SEG_ITEMS;

}  // namespace fhx

// =====================================================================================================
// Context
// =====================================================================================================
using namespace fhx;

namespace fhx {
struct DistState;
}

struct FhxPinnedPair;                            // fhx_emit.inc: two pinned 64 MB buffers + events, kept for the context's life
void fhx_pinned_pair_free(FhxPinnedPair* p);

// Two parked host threads that build the per-count tables beside the fit (fhx_fit): started with the first pass that needs them,
// woken by a condition variable (starting two std::threads in every pass cost the fitting thread ~60 us of the 210 they saved).
struct SideWorkers {
    std::thread th[2];
    std::mutex mu;
    std::condition_variable cv;
    std::function<void()> job[2];
    bool busy[2] = {false, false}, quit = false, started = false;
    void start() {
        if (started) return;
        started = true;
        for (int k = 0; k < 2; ++k)
            th[k] = std::thread([this, k] {
                std::unique_lock<std::mutex> g(mu);
                for (;;) {
                    cv.wait(g, [&] { return quit || busy[k]; });
                    if (quit) return;
                    std::function<void()> f = std::move(job[k]);
                    g.unlock();
                    f();
                    g.lock();
                    busy[k] = false;
                    cv.notify_all();
                }
            });
    }
    void run(int k, std::function<void()> f) {
        {
            std::lock_guard<std::mutex> g(mu);
            job[k] = std::move(f);
            busy[k] = true;
        }
        cv.notify_all();
    }
    void wait() {
        std::unique_lock<std::mutex> g(mu);
        cv.wait(g, [&] { return !busy[0] && !busy[1]; });
    }
    ~SideWorkers() {
        if (!started) return;
        {
            std::lock_guard<std::mutex> g(mu);
            quit = true;
        }
        cv.notify_all();
        for (int k = 0; k < 2; ++k)
            if (th[k].joinable()) th[k].join();
    }
};

struct fhx_ctx {
    SideWorkers side;
    FhxPinnedPair* pinned = nullptr;
    struct TextIngest;                           // fhx_ingest.inc: a parsed contacts text waiting for its chromosome ids
    TextIngest* text_ingest = nullptr;
    int device = -1;
    hipStream_t stream = nullptr;
    hipEvent_t ev[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};   // [6],[7]: around the heavy K2 launch
    long long n_heavy_last = 0;
    bool ev_valid[3] = {false, false, false};
    // sums over the passes since the last reset: K1, K2, K3 and the heavy K2 launch.  A pair is added once, at a point where the
    // stream is known to have passed it (fold_kernel_events), so that a caller timing many passes does not have to stop the
    // stream after each one to read its events
    double ev_sum[4] = {0.0, 0.0, 0.0, 0.0};
    long long ev_count[4] = {0, 0, 0, 0};
    bool ev_folded[4] = {true, true, true, true};
    long long ev_dropped[4] = {0, 0, 0, 0};      // pairs re-recorded before they could be read (see before_rerecord)
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
    size_t work_bytes = 0;                            // its size (engine_sort_ctrl carves the one-sweep scratch behind the K3 view)
    QEntry* d_queue[2] = {nullptr, nullptr};          // K2's per-class row queues (sharded, see QSpan)
    int64_t queue_cap = 0;                            // entries per queue buffer
    unsigned long long* d_k2_counts = nullptr;        // (K2_QUEUES + 1) x K2_MAX_SHARDS queue counters
    QEntry* d_queue_sorted = nullptr;                 // the 300-iteration class, bucketed by (binomial, count), 64-aligned buckets
    dev::CfRow* d_cf_tab = nullptr;                   // K2H_GENERIC x 300 rows of iteration constants
    long long *d_stats_stage = nullptr, *h_stats_stage = nullptr;   // K1's sums + histogram window: device block, pinned host copy
    size_t stats_stage_cap = 0;
    // Results the host waits for in the middle of a pass leave the device by the kernel's own stores into coherent pinned memory,
    // followed by a ticket in h_flags (system-scope release); the host spins on the ticket (wait_ticket) instead of sleeping in
    // hipStreamSynchronize: no copy dispatch behind the kernel, no interrupt + wake-up in front of the host fit.
    //   h_flags[0]  k1_pack_window's ticket (fhx_pass_stats)      h_flags[8]  k3_cutoff's ticket (auto_cutoff)
    //   h_flags[16] K3's fault word (check_fault)
    volatile unsigned long long* h_flags = nullptr;
    unsigned int* d_done = nullptr;                   // [0]: workgroups of k1_pack_window that have stored their part
    unsigned long long ticket = 0;                    // last ticket handed to a kernel
    bool q_prefilled = false;                         // ... and filled the q column with 1.0: k3_compact stores only what differs
    bool k2_prezeroed = false;                        // fhx_pass_stats has zeroed K2's two histograms behind K1 (while the host fits)
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
    unsigned int* h_k3 = nullptr;                     // pinned: [0..1] survivors by the histogram, [8..15] the sort repair's verdict
    hipEvent_t ev_k3 = nullptr;                       // the copy into h_k3
    bool k3_n_is_bound = false;                       // the survivors' number compact_pvalues returned is an upper bound
    int64_t k3_last_kept = -1, k3_last_rows = -1;     // survivors and rows of the last fhx_bh: whether the dense-q launches are worth enqueueing
    hipEvent_t k3_wait_ev = nullptr;                  // the event recorded in front of k3_cutoff (K2's end) the host sleeps on
    unsigned long long k3_ticket = 0;                 // the ticket k3_cutoff publishes behind the survivors' number (h_flags[FLAG_K3 + 1])
    bool k3_counter_zeroed = false;                   // k3_cutoff zeroes the compaction's counter (auto_cutoff), no fill in front of k3_compact
    bool k3_kept_by_hist = false;                     // auto_cutoff has put the survivors' number on its way into h_k3[0..1]
    int64_t sort_stats[8] = {0, 0, 0, 0, 0, 0, 0, 0};   // the last large sort of K3 (fhx_bh_sort_stats)
    std::vector<int64_t> fdr_counts;
    fhx::DistState* dist = nullptr;                 // communicator + exchange buffers of sharded runs (fhx_dist.inc)
    bool dist_ndist_agreed = false;                   // sharded runs: the histogram length was made equal on all ranks
    long long dist_ndist_global = -1;                 // ... the all-reduced answer (max length | non-fixed bit), -1 = not asked yet
    bool dist_any_nonfixed = false;                   // ... and some rank holds off-grid / -r 0 rows (agreed in the same all-reduce)
};

namespace fhx {

// The n scipy.special.bdtrc is given for a total of counts (fithic.py:1070, 1101).  The reference passes a Python int and scipy's
// Cephes core takes `int n`: the value is narrowed to 32 bits - 2^31 becomes -2^31 (n < k: every p-value NaN), 2^32 + 10^6 becomes
// 10^6.  FHX_TOTALS_REFERENCE (default) narrows the same way, so the output is fithic.py's bit for bit; FHX_TOTALS_WIDE keeps the
// true total.  Below 2^31 the two are the same number.  Pinned by tests/golden/f15_*.
inline double bdtrc_total(const fhx_params& prm, long long sum) {
    if (prm.totals == FHX_TOTALS_WIDE) return (double)sum;
    return (double)(int32_t)(uint32_t)(unsigned long long)sum;
}

inline int fail(fhx_ctx* c, int code, const std::string& msg) {
    if (c) c->err = msg;
    return code;
}

#define FHX_HIP(call)                                                                                     \
    do {                                                                                                  \
        hipError_t e_ = (call);                                                                           \
        if (e_ != hipSuccess)                                                                             \
            return fail(ctx, FHX_ERR_HIP, std::string(#call) + ": " + hipGetErrorString(e_));            \
    } while (0)

// ---- tickets: a kernel's results read by the host without a stream synchronisation (fhx_ctx::h_flags) -------------------
constexpr int FLAG_K1 = 0, FLAG_K3 = 8, FLAG_FAULT = 16;   // words of h_flags, a cache line apart
inline int ensure_flags(fhx_ctx* ctx) {
    if (ctx->h_flags) return FHX_OK;
    void* h = nullptr;
    FHX_HIP(hipHostMalloc(&h, 64 * sizeof(unsigned long long), hipHostMallocCoherent | hipHostMallocMapped));
    std::memset(h, 0, 64 * sizeof(unsigned long long));
    ctx->h_flags = reinterpret_cast<volatile unsigned long long*>(h);
    FHX_HIP(hipMalloc(&ctx->d_done, 16 * sizeof(unsigned int)));
    FHX_HIP(hipMemsetAsync(ctx->d_done, 0, 16 * sizeof(unsigned int), ctx->stream));
    return FHX_OK;
}
// the last workgroup of a launch publishes `ticket`: every workgroup calls this after its stores to host memory
__device__ __forceinline__ void publish_ticket(unsigned int* done, volatile unsigned long long* flag, unsigned long long ticket) {
    __threadfence_system();                           // this thread's stores have reached the host
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned int total = gridDim.x * gridDim.y;
        const unsigned int prev = total > 1 ? atomicAdd(done, 1u) : 0u;
        if (prev == total - 1) {
            if (total > 1) atomicExch(done, 0u);     // the next launch on the stream starts from zero
            __hip_atomic_store(const_cast<unsigned long long*>(flag), ticket, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}
// Host side: spin until h_flags[word] shows `ticket`.  The stream is asked now and then: an error (or a stream that has drained
// without the ticket: cannot happen) ends the wait with that answer instead of hanging.  FHX_NO_SPIN=1: hipStreamSynchronize.
inline hipError_t wait_ticket(fhx_ctx* ctx, int word, unsigned long long ticket) {
    static const bool no_spin = std::getenv("FHX_NO_SPIN") != nullptr;
    if (no_spin) return hipStreamSynchronize(ctx->stream);
    volatile unsigned long long* f = ctx->h_flags + word;
    for (unsigned int spins = 1;; ++spins) {
        if (*f == ticket) {
            std::atomic_thread_fence(std::memory_order_acquire);
            return hipSuccess;
        }
        __builtin_ia32_pause();
        if ((spins & 0x3FFFu) == 0) {                 // every ~16 k polls (a few hundred microseconds)
            const hipError_t q = hipStreamQuery(ctx->stream);
            if (q == hipSuccess) return *f == ticket ? hipSuccess : hipStreamSynchronize(ctx->stream);
            if (q != hipErrorNotReady) return q;
        }
    }
}

// K3 sizes its sort by the histogram's survivor count without waiting for the device counter; a count ABOVE that bound (a stale
// histogram - p rewritten behind K2's back -, a kernel that stored p without counting it) would have dropped keys.  bh_scan_tiles
// leaves the counter there; every entry point that hands p / q to the host asks here after its stream wait.  Sticky until
// fhx_reset_passes / a new load of rows.
inline int check_fault(fhx_ctx* ctx) {
    if (!ctx->h_flags) return FHX_OK;
    const unsigned long long seen = ctx->h_flags[FLAG_FAULT];
    if (seen == 0) return FHX_OK;
    return fail(ctx, FHX_ERR_INTERNAL, "K3: " + std::to_string(seen) + " p-values survived the cutoff, more than the key histogram announced - the "
                                       "ranking would be truncated (was p rewritten after fhx_pvalues?)");
}

template <typename T>
inline void dev_free(T*& p) {
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

inline int grid_for(int64_t n, int threads, int max_blocks = 256 * 8) {
    const int64_t b = (n + threads - 1) / threads;
    return (int)std::max<int64_t>(1, std::min<int64_t>(b, max_blocks));
}

// k2_classify over n rows: workgroup b of `grid` takes tiles b, b + grid, ... and queues into shard b, so a shard receives at
// most ceil(tiles / grid) tiles of rows, whatever their classes
inline int k2_classify_grid(int64_t n) {
    static const int cap = [] {                      // FHX_CL_SHARDS: measurements (a smaller grid = fewer, longer queue shards)
        const char* e = std::getenv("FHX_CL_SHARDS");
        const int v = e ? std::atoi(e) : 0;
        return v >= 1 && v <= K2_MAX_SHARDS ? v : K2_MAX_SHARDS;
    }();
    return grid_for(n, K2_CL_TILE, cap);
}
inline long long k2_shard_capacity(int64_t n) {              // entries per shard region: the rows one workgroup of k2_classify can meet
    const long long tiles = std::max<long long>(1, (n + K2_CL_TILE - 1) / K2_CL_TILE), grid = k2_classify_grid(n);
    return ((tiles + grid - 1) / grid) * (long long)K2_CL_TILE;
}


// ---- host functions that cross translation units ----------------------------------------------------------------------------
// fhx_k1.hip
int build_slot_tables(fhx_ctx* ctx);
int run_ids(fhx_ctx* ctx, const unsigned long long* keys, int64_t n, unsigned int* ids, unsigned int* tile_scratch, int64_t* n_runs);
int alloc_row_arrays(fhx_ctx* ctx, int64_t n, int64_t n_dist);
int ingest_device_rows_nonfixed(fhx_ctx* ctx, const int32_t* c1, const int32_t* m1, const int32_t* c2, const int32_t* m2,
                                const int32_t* cnt, int64_t n);
int ingest_device_rows(fhx_ctx* ctx, const int32_t* c1, const int32_t* m1, const int32_t* c2, const int32_t* m2, const int32_t* cnt,
                       int64_t n);
int pass_stats_nonfixed(fhx_ctx* ctx, fhx_stats* out);
int launch_k1(fhx_ctx* ctx);
void fold_kernel_events(fhx_ctx* ctx);        // after a stream synchronisation only
void before_rerecord(fhx_ctx* ctx, int group);   // in front of the hipEventRecord that starts group 0 (K1), 1 (K2 + heavy), 2 (K3)
// fhx_k2.hip
K2Params make_k2_params(fhx_ctx* c);
void launch_k2_extras(fhx_ctx* ctx, const K2Params& P, int64_t n_rows, double* d_expcc, double* d_b1, double* d_b2);
// fhx_k3.hip
int ensure_sort_scratch(fhx_ctx* ctx);
int sort_blocks_for(int64_t n_hint);
int radix_sort_pairs(fhx_ctx* ctx, unsigned long long* keys[2], unsigned int* vals[2], const unsigned long long* counter, int passes,
                     int* result_buf, int64_t n_hint = -1);
void launch_rs_scan(fhx_ctx* ctx, int nblk);                     // exclusive scan of d_block_hist along the workgroup axis + digit totals
int fill_top_hist(fhx_ctx* ctx);
void launch_k3_cutoff(fhx_ctx* ctx, double n_tests, unsigned long long* d_cutoff);
void launch_bh_tile_max(fhx_ctx* ctx, int tiles, const unsigned long long* keys, const unsigned long long* n_ptr, int64_t n_fixed,
                        double n_tests, double rank0, double* tile_max);
void launch_bh_scan_tiles(fhx_ctx* ctx, double* tile_max, const unsigned long long* n_ptr, int64_t n_fixed, double carry_in,
                          double* total_max);
void launch_bh_apply(fhx_ctx* ctx, int tiles, const unsigned long long* keys, const unsigned int* vals, const unsigned long long* n_ptr,
                     int64_t n_fixed, double n_tests, double rank0, const double* tile_carry, const double* extra_carry, double* q_out);
void launch_scatter_q(fhx_ctx* ctx, int64_t n_rows, const unsigned int* rows, const double* q_sorted, const unsigned long long* n_ptr,
                      double* q);
void launch_fdr_hist(fhx_ctx* ctx, const double* q, int64_t n, unsigned long long* buckets);

}  // namespace fhx
