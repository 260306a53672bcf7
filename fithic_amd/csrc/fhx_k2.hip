// fhx_k2.hip - K2: per-pair prior + binomial survival p-value (fit_Spline's pair loop, fithic/fithic.py:1017-1124, with Cephes' bdtrc in
// fhx_bdtrc.hpp), the outlier bookkeeping between passes, what the writer fetches
// (one of the device translation units of libfithic_mi355x.so; shared declarations: fhx_ctx.hpp)
#include "fhx_ctx.hpp"

namespace fhx {

// ===================================================================================================
// K2: per-pair prior + binomial survival p-value
// ===================================================================================================
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

// p of a queued row goes to p[row]: an 8-byte store into a line nobody reads again before K3.  In the queue-order kernels the
// rows of a wave are neighbours and the L2 merges their stores into whole lines; the bucket-sorted heavy class scatters them over
// the whole column, and a plain store then makes the L2 FETCH every line it partially writes (PMC, k2h_heavy: 1639 MB read per
// launch for 427 MB of entries; 713 MB with nontemporal stores, which write through without allocating - and 0.5 ms less per
// pass, profiles/history/r02_y_*).  The queue-order kernels keep plain stores (nontemporal ones cost them 10-28 % more write traffic).
template <bool SCATTERED>
__device__ __forceinline__ void store_p(double* dst, double v) {
    if (SCATTERED)
        __builtin_nontemporal_store(v, dst);
    else
        *dst = v;
}

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
            const double pr = cf_prior[wave][j];
            const double pv = P.lean_closed ? -dev::cephes_expm1(n_total * dev::lean_log1p_neg(pr))
                                            : -dev::cephes_expm1(n_total * dev::cephes_log1p(-pr));             // bdtrc_closed_form, prior < 0.01
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
// (also zeroes the counters of the class kernels that follow it on the stream - `zero_words`: instead of a memset of their own)
// (... and puts "keep every p" into K3's cutoff word - what holds until a cutoff is computed: instead of a copy from the host at the end of K2)
__global__ __launch_bounds__(K2_THREADS) void k2_closed(K2Params P, QSpan q, unsigned long long* __restrict__ zero_words,
                                                        unsigned long long* __restrict__ cutoff_word) {
    __shared__ unsigned int hist_lds[K2_HIST_BINS];
    if (blockIdx.x == 0 && threadIdx.x < MISC_K2_WORDS) zero_words[threadIdx.x] = 0ull;
    if (blockIdx.x == 0 && threadIdx.x == MISC_K2_WORDS) *cutoff_word = KEY_KEEP_ALL;
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

// A class queue seen as ONE dense list across its shards (round 4).  Until then a consumer workgroup took whole shards - 2048
// workgroups where the chip holds 1024 (converging classes) to 1536 (power series) at a time, each with whatever its shard held
// and a partial tile at every shard's end; the timeline of a 1/8 shard of C3 (profiles/history/r04_tl_shard8.txt) had these kernels a
// third over their share of the full-size run.  Now every consumer workgroup builds the exclusive prefix of the shard counts in
// LDS (one DPP scan) and takes dense pieces of that list, and the launches are sized to what the chip holds at a time
// (`resident_grid`).  How the pieces are handed out follows from what a returning atomic on ONE address costs here - 11 ns,
// whoever asks (profiles/history/r02_s_classify_variants.txt): a counter for every 256 entries of the power-series class took twice the
// time of the kernel it fed (0.67 against 0.35 ms, profiles/history/r04_p_kernel_stats.txt), so that class is cut into one contiguous
// range per wave; the 1024-entry tiles of the converging classes and the 300-iteration tasks of k2h_heavy are few enough
// for a counter, and each taker's FIRST piece is its own number, so that nobody queues for the counter at the start.
// C3: converging classes 1.11 + 0.96 -> 1.02 + 0.89 ms, k2h_heavy 6.47 -> 5.69 ms (profiles/r04_z_kernel_stats.txt).
struct QDense {
    // LDS is what limits the residency of the class kernels (16 KB of tile + 16 KB of fused histogram: four workgroups per CU),
    // so the table holds one word per PAIR of shards and the second shard of a pair is told from the first one's count
    static constexpr int PAIRS = K2_MAX_SHARDS / 2;
    unsigned int* prefix;              // LDS, PAIRS + 1 words: entries in front of shard 2 i; [PAIRS] = all entries
    const unsigned long long* count;
    // all K2_THREADS threads of the workgroup; ends with a barrier
    __device__ __forceinline__ void build(const QSpan& q, unsigned int* prefix_lds, unsigned int* wave_tot) {
        static_assert(PAIRS % K2_THREADS == 0, "whole pairs per thread");
        constexpr int PER = PAIRS / K2_THREADS;
        prefix = prefix_lds;
        count = q.count;
        unsigned int c[PER], s = 0;
#pragma unroll
        for (int k = 0; k < PER; ++k) {
            const int sh = 2 * ((int)threadIdx.x * PER + k);
            c[k] = (sh < q.n_shards ? (unsigned int)q.count[sh] : 0u) + (sh + 1 < q.n_shards ? (unsigned int)q.count[sh + 1] : 0u);
            s += c[k];
        }
        const unsigned int incl = wave_incl_sum_u32(s);
        if ((threadIdx.x & 63) == 63) wave_tot[threadIdx.x >> 6] = incl;
        __syncthreads();
        unsigned int run = incl - s;
        for (int w = 0; w < (int)(threadIdx.x >> 6); ++w) run += wave_tot[w];
#pragma unroll
        for (int k = 0; k < PER; ++k) {
            prefix_lds[threadIdx.x * PER + k] = run;
            run += c[k];
        }
        if (threadIdx.x == K2_THREADS - 1) prefix_lds[PAIRS] = run;
        __syncthreads();
    }
    __device__ __forceinline__ unsigned int total() const { return prefix[PAIRS]; }
    // the pair that holds entry v < total(): the last i with prefix[i] <= v (empty pairs share their successor's start)
    __device__ __forceinline__ int pair_of(unsigned int v) const {
        int lo = 0, hi = PAIRS;                        // prefix[lo] <= v < prefix[hi]
#pragma unroll 1
        while (hi - lo > 1) {
            const int mid = (lo + hi) >> 1;
            if (prefix[mid] <= v)
                lo = mid;
            else
                hi = mid;
        }
        return lo;
    }
    // a reader's place in the list: pair `pr` holds entries [base, end), the first `c0` of them in its first shard
    struct Cursor {
        int pr;
        unsigned int base, end, c0;
    };
    __device__ __forceinline__ void load(Cursor& c) const {
        c.base = prefix[c.pr];
        c.end = prefix[c.pr + 1];
        c.c0 = (unsigned int)count[2 * c.pr];
    }
    __device__ __forceinline__ Cursor open(unsigned int v) const {
        Cursor c;
        c.pr = pair_of(v);
        load(c);
        return c;
    }
    // entry v of the dense list, v at or behind the cursor's pair (consecutive pieces of a thread: the same pair, or a step on)
    __device__ __forceinline__ const QEntry* at(const QSpan& q, Cursor& c, unsigned int v) const {
        if (v >= c.end) {
            do ++c.pr;
            while (v >= prefix[c.pr + 1]);
            load(c);
        }
        const unsigned int u = v - c.base;
        return u < c.c0 ? qentry(q, 2 * c.pr, (long long)u) : qentry(q, 2 * c.pr + 1, (long long)(u - c.c0));
    }
};
constexpr int K2_WAVES = K2_THREADS / 64;

// one class, one lane per row: every wave takes one contiguous range of the dense list (whole multiples of 64 entries)
template <int CLS, bool SMALL_N>
__global__ __launch_bounds__(K2_THREADS) void k2_queue(K2Params P, QSpan q) {
    __shared__ unsigned int hist_lds[K2_HIST_BINS];
    __shared__ unsigned int prefix_lds[QDense::PAIRS + 1], wave_tot[K2_WAVES];
    FusedHist H;
    H.init(hist_lds, P.top_hist);
    QDense D;
    D.build(q, prefix_lds, wave_tot);
    const unsigned long long total = D.total();
    const unsigned int lane = threadIdx.x & 63;
    const unsigned long long n_waves = (unsigned long long)gridDim.x * K2_WAVES;
    const unsigned long long share = ((total + n_waves - 1) / n_waves + 63ull) & ~63ull;
    const unsigned long long begin = ((unsigned long long)blockIdx.x * K2_WAVES + (threadIdx.x >> 6)) * share;
    const unsigned long long end = min(total, begin + share);
    QDense::Cursor cur{0, 0u, 0u, 0u};
    if (begin + lane < end) cur = D.open((unsigned int)(begin + lane));
#pragma unroll 1
    for (unsigned long long v = begin + lane; v < end; v += 64) {
        const QEntry e = *D.at(q, cur, (unsigned int)v);
        const bool is_inter = e.count < 0;
        const int c = is_inter ? -e.count : e.count;
        const double pv = dev::bdtrc_count_class<CLS, SMALL_N>(c, is_inter ? P.inter : P.intra, e.prior);
        store_p<false>(P.p + e.row, pv);
        H.add(pv);
    }
    H.flush(P.top_hist);
}

// The two converging continued-fraction classes need ~5..17 iterations, growing with the contact count: in queue order a
// wave waits for its slowest lane (measured: mean 9.3 iterations, mean of the per-wave maximum 20.3).  Each workgroup
// therefore takes a tile of 1024 entries, counting-sorts it by min(count, 31) in LDS (one LDS atomic per entry) and hands
// every wave 64 neighbours of that order (per-wave maximum 11.4).  Results go to P.p[row], so the order is free.
// Tiles are pieces of the dense list (QDense): a workgroup's first tile is its own number, the others come from the launch's
// counter (`quarters` 1..4 fixes the tile at that many times 256 entries: measurements).
constexpr int K2_SORT_TILE = 1024;
constexpr int K2_SORT_BUCKETS = 32;
template <int CLS, bool SMALL_N, int WPE>
__global__ __launch_bounds__(K2_THREADS) __attribute__((amdgpu_waves_per_eu(WPE))) void k2_queue_by_count(K2Params P, QSpan q,
                                                                                                         unsigned int* next_tile,
                                                                                                         int quarters) {
    static_assert(K2_SORT_TILE == 4 * K2_THREADS, "four entries per thread");
    __shared__ QEntry tile[K2_SORT_TILE];
    __shared__ unsigned int bucket_cnt[K2_SORT_BUCKETS], bucket_off[K2_SORT_BUCKETS];
    __shared__ unsigned int hist_lds[K2_HIST_BINS];
    __shared__ unsigned int prefix_lds[QDense::PAIRS + 1], wave_tot[K2_WAVES], tile_lds;
    FusedHist H;
    H.init(hist_lds, P.top_hist);
    QDense D;
    D.build(q, prefix_lds, wave_tot);
    const unsigned int total = D.total();
    // tile: at most 1024 entries, and such that the workgroups' shares are whole tiles - a 1/8 shard of C3 has 2.6 tiles of 1024
    // per workgroup, i.e. three rounds of which the last is 60 % full; three tiles of 896 each fill them all
    unsigned int tile_n = (unsigned int)quarters * K2_THREADS;
    if (quarters < 1 || quarters > 4) {
        const unsigned int share = (total + gridDim.x - 1) / gridDim.x;
        const unsigned int rounds = max(1u, (share + K2_SORT_TILE - 1) / K2_SORT_TILE);
        tile_n = min((unsigned int)K2_SORT_TILE, ((share + rounds - 1) / rounds + 63u) & ~63u);
        tile_n = max(tile_n, 64u);
    }
    const unsigned int tiles = (total + tile_n - 1) / tile_n;
    unsigned int mine = blockIdx.x;
    for (;;) {
        if (threadIdx.x < K2_SORT_BUCKETS) bucket_cnt[threadIdx.x] = 0;
        if (threadIdx.x == 0) tile_lds = mine;
        __syncthreads();
        const unsigned int t = tile_lds;
        if (t >= tiles) break;
        if (threadIdx.x == 0) mine = gridDim.x + atomicAdd(next_tile, 1u);      // the next tile's number arrives while this one is worked on
        const unsigned int v0 = t * tile_n;
        const int m = (int)min(tile_n, total - v0);
        QEntry e[4];
        int bucket[4];
        unsigned int slot[4];
        QDense::Cursor cur{0, 0u, 0u, 0u};
        if ((int)threadIdx.x < m) cur = D.open(v0 + threadIdx.x);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int idx = r * K2_THREADS + (int)threadIdx.x;
            bucket[r] = -1;
            if (idx < m) {
                e[r] = *D.at(q, cur, v0 + idx);
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
    H.flush(P.top_hist);
}

// Measured and dropped (round 4, profiles/history/r04_v_cf_split_ab.txt, profiles/ab_cf_split.sh): the same two classes as a LOOP kernel
// (count-sorted tile, no histogram, continued fraction only: 72-86 VGPRs, 7 workgroups per CU, result through 8 B per entry) and a
// FINISH kernel (queue order, Cephes' three logs and exp, histogram: 7 per CU).  Bit-identical, and slower: loop 0.80 + 0.65 ms,
// finish 0.44 + 0.44 ms against 1.03 + 0.90 ms fused - compiled for 5 or 7 waves per SIMD the loop kernel takes the same time
// (it waits on its own dependent chains and on the slowest lane of a wave, not on registers), and the finish on its own is bound
// by the scattered 8-byte p stores that the fused kernel hides under the loop.

// ---- the 300-iteration class in count-homogeneous waves -----------------------------------------------------------
// The swapped-continued-fraction queue is counting-sorted by (binomial, contact count) so that every wave of k2h_heavy
// holds 64 rows of ONE count: all per-iteration constants of Cephes' loop then come from a table row per iteration
// through scalar loads (cf_swapped_uniform, fhx_bdtrc.hpp).  Bucket = count for intra rows, K2H_KCAP + count for
// rows of the inter-chromosomal binomial, one last bucket for counts >= K2H_KCAP (evaluated by the per-lane k2_queue
// kernel).  Every bucket starts at a multiple of 64 entries in the sorted queue, so a wave never straddles two counts.

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

// the two consumers of the bucket totals in ONE launch (they do not depend on each other) - a launch less between k2_classify and
// k2h_heavy.  Workgroups 0..K2H_GENERIC-1: one per non-empty (binomial, count) bucket, one thread per iteration - the 300 rows of
// iteration constants.  The last workgroup: bucket starts, each rounded up to a multiple of `granule` entries (64 x the rows a lane
// of k2h_heavy takes): off[b] for b = 0..K2H_BUCKETS (the last one = padded total).
static_assert(dev::kCfIters <= 1024, "one thread per table row");
__device__ __forceinline__ void k2h_table_row(const unsigned int* __restrict__ digit_total, double n_intra, double n_inter,
                                              dev::CfRow* __restrict__ tab) {
    const int b = blockIdx.x;
    if (digit_total[b] == 0 || (int)threadIdx.x >= dev::kCfIters) return;
    const bool inter = b >= K2H_KCAP;
    tab[(size_t)b * dev::kCfIters + threadIdx.x] = dev::cf_swapped_row(inter ? n_inter : n_intra, inter ? b - K2H_KCAP : b, (int)threadIdx.x);
}
__global__ __launch_bounds__(1024) void k2h_offsets_and_tables(const unsigned int* __restrict__ digit_total, unsigned int* __restrict__ off,
                                                               unsigned int granule, double n_intra, double n_inter, dev::CfRow* __restrict__ tab) {
    if (blockIdx.x < K2H_GENERIC) {
        k2h_table_row(digit_total, n_intra, n_inter, tab);
        return;
    }
    __shared__ unsigned int part[1024];
    const unsigned int g1 = granule - 1u;
    const unsigned int a = (digit_total[2 * threadIdx.x] + g1) / granule * granule, b = (digit_total[2 * threadIdx.x + 1] + g1) / granule * granule;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const unsigned int incl = wave_incl_sum_u32(a + b);
    if (lane == 63) part[wave] = incl;
    __syncthreads();
    unsigned int excl = incl - (a + b);
    for (int w = 0; w < wave; ++w) excl += part[w];
    if (threadIdx.x == 1023) off[K2H_BUCKETS] = excl + a + b;
    off[2 * threadIdx.x] = excl;
    off[2 * threadIdx.x + 1] = excl + a;
}


// Lanes cf_swapped_uniform cannot take (unusual inputs or states, see fhx_bdtrc.hpp) are appended to `redo` - the space the
// unsorted queue occupied, free once k2h_scatter has run - and k2h_generic evaluates them with the per-lane loop; keeping that
// loop out of this kernel keeps it at 8 waves per SIMD (38 VGPRs instead of 102).
struct K2HeavyParams {            // the few fields of K2Params this kernel reads: its SGPR count decides how many waves a CU admits
    dev::BinomTables intra, inter;
    double* p;
    unsigned long long* top_hist;
};
// shader-cycle and constant-rate counters at the start and at the end of the first wave of k2h_heavy's workgroup 0 (a symbol, not a
// kernel argument: the kernel sits at its SGPR limit and a pointer held across its loop was four more spills)
__device__ unsigned long long g_k2h_clk[4];

// R rows per lane (a task = 64 R consecutive entries of one bucket: every bucket starts at a multiple of that), WPE waves per
// SIMD: see cf_swapped_uniform for why more rows per lane beat more waves.
template <int R, int WPE, bool SMALL_N>
__global__ __launch_bounds__(K2H_THREADS) __attribute__((amdgpu_waves_per_eu(WPE, WPE))) void k2h_heavy(
    K2HeavyParams P, const QEntry* __restrict__ sorted, const unsigned int* __restrict__ off,
    const unsigned int* __restrict__ digit_total, const dev::CfRow* __restrict__ tab, QEntry* __restrict__ redo,
    unsigned long long* __restrict__ n_redo, unsigned int* __restrict__ next_task) {
    static_assert(R >= 1 && R <= K2H_MAX_ROWS, "the sorted queue is padded for at most K2H_MAX_ROWS rows per lane");
    __shared__ unsigned int hist_lds[K2_HIST_BINS];
    FusedHist H;
    H.init(hist_lds, P.top_hist);
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int lane = threadIdx.x & 63;
    constexpr unsigned int TASK = 64u * R;
    const unsigned int n_tasks = off[K2H_GENERIC] / TASK;        // tasks in front of the generic bucket
    // (Round 6 tried two launches for small inputs - whole rounds of <4 rows, 4 waves>, the remainder as <1 row, 8 waves> - against
    // the one <2, 8> launch: 918 against 824 us on a 1/8 shard of C3, 768 against 681 on C2, profiles/r06/heavy_split.txt.  A round
    // of four-row tasks at four waves per SIMD takes 257 us where the full-size run's rate would make it 222, and the remainder
    // 170 us: dropped.)
    // tasks are handed out by a counter (round 4): with a fixed stride a 1/8 shard of C3 gave 27 % of the waves four tasks and
    // the others three - the launch took 4/3.3 of its share of the full-size one (profiles/history/r04_tl_shard8.txt).  One returning
    // atomic per 300 x 92 instructions of work.
    const unsigned int n_waves = gridDim.x * (K2H_THREADS / 64);
    // the clock this launch runs at (boxes differ by 4 %: what "the VALU issue floor" is in milliseconds depends on it): one wave
    // leaves s_memtime (shader cycles) and s_memrealtime (constant rate) at its start and its end - stored at once, nothing kept
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        g_k2h_clk[0] = __builtin_amdgcn_s_memtime();
        g_k2h_clk[1] = __builtin_amdgcn_s_memrealtime();
    }
    for (unsigned int task = blockIdx.x * (K2H_THREADS / 64) + wave; task < n_tasks;) {
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
        unsigned int next = 0;
        if (lane == 0) next = atomicAdd(next_task, 1u);
        task = n_waves + (unsigned int)__builtin_amdgcn_readfirstlane((int)next);
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        g_k2h_clk[2] = __builtin_amdgcn_s_memtime();
        g_k2h_clk[3] = __builtin_amdgcn_s_memrealtime();
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

// (Measured and dropped in round 4, profiles/history/r04_h_cfu_ab.txt: the two CONVERGING classes through the heavy class's machinery -
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
            if (within) e = (is_inter ? P.total_inter : P.total_intra) * prior;
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


// ---- host side ----------------------------------------------------------------------------------------------------------
K2Params make_k2_params(fhx_ctx* c) {
    K2Params P{};
    P.loc1 = c->d_loc1;
    P.loc2 = c->d_loc2;
    P.count = c->d_count;
    P.slot_bias = c->d_slot_bias;
    P.no_bias = !c->have_bias;
    static const int lean = std::getenv("FHX_LEAN_CLOSED") ? std::atoi(std::getenv("FHX_LEAN_CLOSED")) : 0;
    P.lean_closed = lean;
    P.prior_lut = c->d_lut;
    P.lut_len = (int)std::min<size_t>(std::max<size_t>(c->fit.prior_lut.size(), 1), (size_t)INT32_MAX);
    // the n of the two binomials: narrowed to a C int as scipy does, unless the caller asked for wide totals (bdtrc_total)
    const double n_intra = bdtrc_total(c->prm, c->stats.in_range_sum), n_inter = bdtrc_total(c->prm, c->stats.inter_sum);
    P.intra = dev::BinomTables{c->d_lbeta_intra, c->d_invb_intra, n_intra, (n_intra + 1.0) < dev::kMaxGam};
    P.inter = dev::BinomTables{c->d_lbeta_inter, c->d_invb_inter, n_inter, (n_inter + 1.0) < dev::kMaxGam};
    P.total_intra = (double)c->stats.in_range_sum;
    P.total_inter = (double)c->stats.inter_sum;
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


void launch_k2_extras(fhx_ctx* ctx, const K2Params& P, int64_t n_rows, double* d_expcc, double* d_b1, double* d_b2) {
    hipLaunchKernelGGL(k2_extras, dim3(grid_for(n_rows, 256)), dim3(256), 0, ctx->stream, P, ctx->prm.bias_low, ctx->prm.bias_up, d_expcc, d_b1,
                       d_b2);
}

}  // namespace fhx

// ---- C ABI -----------------------------------------------------------------------------------------------------------------
// workgroups of `kernel` (K2_THREADS threads, its static LDS) that the device holds at a time: the class kernels cut their queue into
// that many shares, or take pieces from a counter - a larger grid would run its surplus on a part-empty chip
template <typename K>
static int resident_grid(fhx_ctx* ctx, K kernel) {
    static std::mutex mu;
    static std::map<std::pair<int, const void*>, int> cache;
    std::lock_guard<std::mutex> lock(mu);
    const std::pair<int, const void*> key(ctx->device, (const void*)kernel);
    auto it = cache.find(key);
    if (it != cache.end()) return it->second;
    int per_cu = 0, cus = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, K2_THREADS, 0) != hipSuccess || per_cu < 1) per_cu = 4;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, ctx->device) != hipSuccess || cus < 1) cus = 256;
    if (std::getenv("FHX_DEBUG_GRID")) std::fprintf(stderr, "resident_grid: %d workgroups per CU x %d CUs\n", per_cu, cus);
    return cache[key] = per_cu * cus;
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
    before_rerecord(ctx, 1);
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
    // fhx_pass_stats zeroes the two histograms of this call behind K1, while the host fits (k1_prezero); any other caller - and
    // anything that used the count matrix in between (radix_passes clears the mark) - gets them zeroed here
    const bool prezeroed = ctx->k2_prezeroed;
    ctx->k2_prezeroed = false;
    // K3's key histogram rides on K2's stores of p - except on the table path, whose class kernels store table entries
    ctx->k2_hist_valid = false;
    if (memo_cap < 0 && !getenv("FHX_NO_FUSED_HIST")) {
        if (!ctx->d_k2_hist) FHX_HIP(hipMalloc(&ctx->d_k2_hist, TOP_BINS * sizeof(unsigned long long)));
        if (!prezeroed) FHX_HIP(hipMemsetAsync(ctx->d_k2_hist, 0, TOP_BINS * sizeof(unsigned long long), ctx->stream));
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
        if (!prezeroed)
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
    hipLaunchKernelGGL(k2_closed, qgrid, qblock, 0, ctx->stream, P, Q.q[K2_CLOSED - 1], ctx->d_misc + MISC_K2_REDO, ctx->d_misc + 6);
    // totals below 171: the kernels that carry Cephes' pow branch (a binomial without a single contact - no inter-chromosomal
    // rows - classifies every row as trivial and reaches no class kernel: it does not count)
    const bool small_n = (P.intra.small_n && P.intra.n >= 1.0) || (P.inter.small_n && P.inter.n >= 1.0);
    // one range per wave of TWICE the resident workgroups: the ranges are cut by entries, not by work, and the second half
    // evens the first one out (C3: 376 us at 1 x, 358 at 2.3 x, 362 at 4.7 x, profiles/history/r04_t_ps.txt); FHX_PS_GRID: measurements
    static const int ps_grid = std::getenv("FHX_PS_GRID") ? std::atoi(std::getenv("FHX_PS_GRID")) : 0;
#define FHX_LAUNCH_QUEUE(CLS)                                                                                         \
    do {                                                                                                              \
        if (small_n)                                                                                                  \
            hipLaunchKernelGGL(HIP_KERNEL_NAME(k2_queue<CLS, true>), dim3(2 * resident_grid(ctx, k2_queue<CLS, true>)), qblock, 0,      \
                               ctx->stream, P, Q.q[(CLS) - 1]);                                                       \
        else                                                                                                          \
            hipLaunchKernelGGL(HIP_KERNEL_NAME(k2_queue<CLS, false>), dim3(ps_grid > 0 ? ps_grid : 2 * resident_grid(ctx, k2_queue<CLS, false>)), qblock, 0,    \
                               ctx->stream, P, Q.q[(CLS) - 1]);                                                       \
    } while (0)
    const bool legacy_heavy = getenv("FHX_K2_LEGACY") != nullptr;      // A/B and tests: the per-lane kernel of round 1
    // the handed-back rows' counter and the work counters of the class kernels (k2h_heavy's tasks, one per class queue)
    unsigned long long* n_redo = ctx->d_misc + MISC_K2_REDO;
    unsigned long long* k2_next = ctx->d_misc + MISC_K2_NEXT;                 // both zeroed by k2_closed above
    if (legacy_heavy) {
        FHX_HIP(hipEventRecord(ctx->ev[6], ctx->stream));
        FHX_LAUNCH_QUEUE(dev::BC_CF_SWAPPED);            // longest-running class first
        FHX_HIP(hipEventRecord(ctx->ev[7], ctx->stream));
        ctx->ev_folded[3] = false;
    } else {
        static_assert(K2H_BUCKETS == RADIX && K2H_BLOCKS == SORT_BLOCKS, "the radix sort's count matrix and scan are reused");
        const QSpan hs = Q.q[dev::BC_CF_SWAPPED - 1];
        QEntry* hq = ctx->d_queue[0];                    // the handed-back rows: this buffer is dead once it is scattered and the
                                                         // power-series class (its other tenant) has run
        if (!Q.heavy_hist)              // otherwise k2_classify has counted while it queued
            hipLaunchKernelGGL(k2h_count, dim3(K2H_BLOCKS), dim3(K2H_THREADS), 0, ctx->stream, hs, ctx->d_block_hist);
        launch_rs_scan(ctx, (int)SORT_BLOCKS);
        // rows per lane: 4 at 4 waves/SIMD (7.43 -> 6.66 ms per 2.7e7 rows against one row per lane at 8 waves/SIMD; 2 x 8, 2 x 6,
        // 3 x 5, 4 x 3 are within 3 % of each other, profiles/history/r03_c_heavy_variants.txt); FHX_K2H_ROWS / FHX_K2H_WAVES: measurements
        // rows per lane: 4 at 4 waves/SIMD - C3 (2.7e7 rows in the class) 7.43 -> 6.66 ms, a 1/18 shard (1.5e6 rows) 0.87 -> 0.78 ms of
        // K2 against one row per lane at 8 waves/SIMD; 2 x 8, 3 x 5 and 4 x 3 are within 3 % (profiles/history/r03_c_*heavy_variants.txt).
        // FHX_K2H_ROWS (1, 2) / FHX_K2H_WAVES (3) select the instantiations kept for measurements.
        const char* heavy_rows_env = std::getenv("FHX_K2H_ROWS");                 // read per call: the tests run every instantiation
        const int heavy_rows = heavy_rows_env ? std::atoi(heavy_rows_env) : 0;
        static const int heavy_wpe = std::getenv("FHX_K2H_WAVES") ? std::atoi(std::getenv("FHX_K2H_WAVES")) : 0;
        // a small input (a shard of a strong-scaling run) has a handful of 256-entry tasks per wave and ends with most waves idle:
        // two rows per lane at eight waves per SIMD halves the task (a 1/8 shard of C3: 876 -> 829 us, profiles/history/r04_t_rows.txt)
        const int hr = (heavy_rows == 1 || heavy_rows == 2) ? heavy_rows       // the instantiations below: 1, 2 or 4 rows per lane - the
                       : (heavy_rows == 0 && k2_n < 32000000) ? 2 : 4;         // bucket granule must be the launched kernel's task size
        hipLaunchKernelGGL(k2h_offsets_and_tables, dim3(K2H_GENERIC + 1), dim3(1024), 0, ctx->stream, (const unsigned int*)ctx->d_digit_total,
                           ctx->d_k2h_off, 64u * (unsigned int)hr, P.intra.n, P.inter.n, ctx->d_cf_tab);
        hipLaunchKernelGGL(k2h_scatter, dim3(K2H_BLOCKS), dim3(K2H_THREADS), 0, ctx->stream, hs,
                           (const unsigned int*)ctx->d_block_hist, (const unsigned int*)ctx->d_k2h_off, ctx->d_queue_sorted);
        FHX_LAUNCH_QUEUE(dev::BC_PSERIES);               // before the redo list reuses the buffer it shares with the heavy queue
        FHX_HIP(hipEventRecord(ctx->ev[6], ctx->stream));
        const K2HeavyParams HP{P.intra, P.inter, P.p, P.top_hist};
#define FHX_HEAVY_N(R, W, S)                                                                                                        \
    hipLaunchKernelGGL(HIP_KERNEL_NAME(k2h_heavy<R, W, S>), dim3(256 * W), dim3(K2H_THREADS), 0, ctx->stream, HP,                     \
                       (const QEntry*)ctx->d_queue_sorted, (const unsigned int*)ctx->d_k2h_off, (const unsigned int*)ctx->d_digit_total, \
                       (const dev::CfRow*)ctx->d_cf_tab, hq, n_redo, (unsigned int*)k2_next)
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
        ctx->ev_folded[3] = false;
        hipLaunchKernelGGL(k2h_generic, dim3(256 * 4), dim3(K2_THREADS), 0, ctx->stream, P, (const QEntry*)ctx->d_queue_sorted,
                           (const unsigned int*)ctx->d_k2h_off, (const unsigned int*)ctx->d_digit_total, (const QEntry*)hq,
                           (const unsigned long long*)n_redo);
    }
    static const int cf_wpe = std::getenv("FHX_CF_WAVES") ? std::atoi(std::getenv("FHX_CF_WAVES")) : 0;              // measurements
    static const int cf_quarters = std::getenv("FHX_CF_QUARTERS") ? std::atoi(std::getenv("FHX_CF_QUARTERS")) : 0;   // 1..4: measurements
#define FHX_LAUNCH_QUEUE_BY_COUNT(CLS)                                                                                                \
    do {                                                                                                                              \
        if (small_n)                                                                                                                  \
            hipLaunchKernelGGL(HIP_KERNEL_NAME(k2_queue_by_count<CLS, true, 4>), dim3(resident_grid(ctx, k2_queue_by_count<CLS, true, 4>)), \
                               qblock, 0, ctx->stream, P, Q.q[(CLS) - 1],          \
                               (unsigned int*)(k2_next + (CLS)), cf_quarters);     \
        else if (cf_wpe == 4)                                                                                                         \
            hipLaunchKernelGGL(HIP_KERNEL_NAME(k2_queue_by_count<CLS, false, 4>), dim3(resident_grid(ctx, k2_queue_by_count<CLS, false, 4>)), \
                               qblock, 0, ctx->stream, P, Q.q[(CLS) - 1],          \
                               (unsigned int*)(k2_next + (CLS)), cf_quarters);     \
        else                                                                                                                          \
            hipLaunchKernelGGL(HIP_KERNEL_NAME(k2_queue_by_count<CLS, false, 5>), dim3(resident_grid(ctx, k2_queue_by_count<CLS, false, 5>)), \
                               qblock, 0, ctx->stream, P, Q.q[(CLS) - 1],          \
                               (unsigned int*)(k2_next + (CLS)), cf_quarters);     \
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
    ctx->ev_folded[1] = false;
    ctx->have_p = true;
    ctx->have_q = false;
    ctx->n_sorted = -1;
    return FHX_OK;                                                 // (K3's cutoff word says "keep every p": k2_closed)
}

int fhx_bdtrc_array(fhx_ctx* ctx, double n_total, const int32_t* count, const double* prior, int64_t n, double* out) {
    if (!ctx || !count || !prior || !out || n < 0) return FHX_ERR_ARG;
    if (ctx->device < 0) return fail(ctx, FHX_ERR_NO_DEVICE, "host-only context");
    if (n == 0) return FHX_OK;
    FHX_HIP(hipSetDevice(ctx->device));
    // an integral total is what the reference hands over (a Python int): scipy narrows it to a C int, and so does this entry
    // unless the context was set to FHX_TOTALS_WIDE (bdtrc_total; a context without parameters is in reference mode)
    if (std::fabs(n_total) < 9.2e18 && n_total == std::floor(n_total)) n_total = bdtrc_total(ctx->prm, (long long)n_total);
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
    return check_fault(ctx);
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
    return check_fault(ctx);
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

// the shader clock the last heavy launch ran at, GHz: cycles a wave of it counted (s_memtime) over the constant-rate ticks of the same
// stretch (s_memrealtime, hipDeviceAttributeWallClockRate); 0 when the launch held no task
int fhx_k2_heavy_clock(fhx_ctx* ctx, double* ghz) {
    if (!ctx || !ghz) return FHX_ERR_ARG;
    if (ctx->device < 0) return fail(ctx, FHX_ERR_NO_DEVICE, "host-only context");
    if (!ctx->ev_valid[1]) return fail(ctx, FHX_ERR_ARG, "fhx_pvalues has not run");
    FHX_HIP(hipSetDevice(ctx->device));
    FHX_HIP(hipStreamSynchronize(ctx->stream));
    unsigned long long w[4] = {0, 0, 0, 0};
    FHX_HIP(hipMemcpyFromSymbol(w, HIP_SYMBOL(g_k2h_clk), sizeof(w), 0, hipMemcpyDeviceToHost));
    int khz = 0;
    FHX_HIP(hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, ctx->device));
    *ghz = (w[3] > w[1] && w[2] > w[0] && khz > 0) ? (double)(w[2] - w[0]) / ((double)(w[3] - w[1]) / ((double)khz * 1e3)) / 1e9 : 0.0;
    return FHX_OK;
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
        FHX_HIP(hipMemcpy(&redo, ctx->d_misc + MISC_K2_REDO, sizeof(redo), hipMemcpyDeviceToHost));
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

