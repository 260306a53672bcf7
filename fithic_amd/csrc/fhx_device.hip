// fhx_device.hip - the context behind the C ABI of libfithic_mi355x.so: life cycle, parameters, fragment / bias tables, the
// host fit (fhx_host.cpp) and its tables on the device, plain accessors - and, included below, the sharded schedule
// (fhx_dist.inc) and the device-side file I/O (fhx_inflate.inc, fhx_ingest.inc, fhx_emit.inc).  The kernels live in fhx_k1.hip
// (K0 / K1), fhx_k2.hip (K2) and fhx_k3.hip (K3); what the units share is fhx_ctx.hpp.
//
// CDNA4 notes: wave64 everywhere (ballots are 64-bit); pair arrays are streamed with 16-byte-per-lane coalesced loads; the
// distance histogram is privatised in LDS and flushed with one global atomic per touched bin per workgroup; there is no dense
// contraction on this path, so no MFMA; every unit is compiled with -ffp-contract=off (see fhx_bdtrc.hpp for why).
#include <thread>

#include "fhx_ctx.hpp"

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
        if (ctx->ev_k3) (void)hipEventDestroy(ctx->ev_k3);
        if (ctx->h_k3) (void)hipHostFree(ctx->h_k3);
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
        if (ctx->h_flags) (void)hipHostFree((void*)ctx->h_flags);
        dev_free(ctx->d_done);
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
    if (p->totals != FHX_TOTALS_REFERENCE && p->totals != FHX_TOTALS_WIDE) return fail(ctx, FHX_ERR_ARG, "totals must be FHX_TOTALS_REFERENCE or FHX_TOTALS_WIDE");
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

#include "fhx_nfpairs.inc"

int fhx_fit(fhx_ctx* ctx, fhx_fit_info* out) {
    if (!ctx) return FHX_ERR_ARG;
    if (!ctx->have_params || !ctx->have_frags) return fail(ctx, FHX_ERR_ARG, "parameters and fragments must be loaded");
    if (!ctx->have_stats) return fail(ctx, FHX_ERR_ARG, "fhx_pass_stats (or fhx_set_global_stats) must run first");
    static const bool fit_times = std::getenv("FHX_FIT_TIMES") != nullptr;      // measurements: where the host part of a pass goes
    const auto t0 = std::chrono::steady_clock::now();
    // The per-count tables (log Beta and 1 / Beta of (count, n - count + 1): two lgamma each) depend on K1's statistics alone, not
    // on the fit: with large counts (40 kb bins: 16 000 entries, 210 us - as long as the whole fit) they are built by the context's two
    // parked side threads while this one bins and fits; small tables (5 kb bins: 700 entries, 10 us) stay inline.
    const int64_t mc = std::max<int64_t>(ctx->stats.max_count, 1);
    std::vector<double> lb_a, ib_a, lb_e, ib_e;
    const bool tables_aside = ctx->device >= 0 && mc >= 2048 && !std::getenv("FHX_FIT_SERIAL");
    if (tables_aside) {
        const double n_a = bdtrc_total(ctx->prm, ctx->stats.in_range_sum), n_e = bdtrc_total(ctx->prm, ctx->stats.inter_sum);
        lb_a.assign((size_t)mc + 1, 0.0);
        ib_a.assign((size_t)mc + 1, 0.0);
        lb_e.assign((size_t)mc + 1, 0.0);
        ib_e.assign((size_t)mc + 1, 0.0);
        const int64_t mid = mc / 2;
        double *la = lb_a.data(), *ia = ib_a.data(), *le = lb_e.data(), *ie = ib_e.data();
        ctx->side.start();
        ctx->side.run(0, [=] {
            fill_lbeta_table(n_a, 1, mid, la, ia);
            fill_lbeta_table(n_e, 1, mid, le, ie);
        });
        ctx->side.run(1, [=] {
            fill_lbeta_table(n_a, mid + 1, mc, la, ia);
            fill_lbeta_table(n_e, mid + 1, mc, le, ie);
        });
    }
    struct Join {                                        // (the tables' vectors must outlive the workers on every return path)
        SideWorkers* w;
        bool on;
        ~Join() {
            if (on) w->wait();
        }
    } join_side{&ctx->side, tables_aside};
    PassInputs in;
    fill_pass_inputs(ctx, in);
    if (ctx->device >= 0 && in.resolution == 0)          // -r 0: the walk over all possible pairs runs on the GPU (fhx_nfpairs.inc)
        in.nf_pairs = [ctx, &in](const std::vector<Bin>& bins, NfPairSums& sums) {
            return nf_pairs_on_device(ctx, ctx->frags, in.dist_low, in.dist_up, bins, sums);
        };
    std::string err;
    const int rc = run_host_pass(in, ctx->frags, ctx->fit, err);
    if (rc != FHX_OK) return fail(ctx, rc, err);
    const auto t1 = std::chrono::steady_clock::now();
    auto t2 = t1;
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
        if (tables_aside) {
            ctx->side.wait();
        } else {
            build_lbeta_table(bdtrc_total(ctx->prm, ctx->stats.in_range_sum), mc, lb_a, ib_a);
            build_lbeta_table(bdtrc_total(ctx->prm, ctx->stats.inter_sum), mc, lb_e, ib_e);
        }
        t2 = std::chrono::steady_clock::now();
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
    if (fit_times) {
        const auto t3 = std::chrono::steady_clock::now();
        auto us = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) {
            return std::chrono::duration<double, std::micro>(b - a).count();
        };
        std::fprintf(stderr, "fhx_fit: bins + spline + isotonic %.1f us, per-count tables (max count %lld) %.1f us, staging + copy %.1f us\n",
                     us(t0, t1), (long long)ctx->stats.max_count, us(t1, t2), us(t2, t3));
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
        out->totals = ctx->prm.totals;
        out->totals_narrowed = (ctx->stats.in_range_sum >= (1ll << 31) ? 1 : 0) | (ctx->stats.inter_sum >= (1ll << 31) ? 2 : 0);
        out->bdtrc_n_intra = (int64_t)bdtrc_total(ctx->prm, ctx->stats.in_range_sum);
        out->bdtrc_n_inter = (int64_t)bdtrc_total(ctx->prm, ctx->stats.inter_sum);
    }
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

// One pass in one call: fhx_pass_stats -> fhx_fit -> fhx_pvalues -> fhx_bh, nothing of the caller's language in between (the four
// ctypes round trips of the Python driver were ~40 us of a small shard's pass, between K1 and k2_classify with the GPU idle).
int fhx_run_pass(fhx_ctx* ctx, fhx_stats* stats, fhx_fit_info* info) {
    if (!ctx) return FHX_ERR_ARG;
    static const bool pass_times = std::getenv("FHX_PASS_TIMES") != nullptr;    // measurements: host time inside each of the four calls
    const auto t0 = std::chrono::steady_clock::now();
    int rc = fhx_pass_stats(ctx, stats);
    if (rc != FHX_OK) return rc;
    const auto t1 = std::chrono::steady_clock::now();
    fhx_fit_info mine{};
    rc = fhx_fit(ctx, &mine);
    if (rc != FHX_OK) return rc;
    if (info) *info = mine;
    const auto t2 = std::chrono::steady_clock::now();
    rc = fhx_pvalues(ctx);
    if (rc != FHX_OK) return rc;
    const auto t3 = std::chrono::steady_clock::now();
    rc = fhx_bh(ctx, mine.bh_total_tests);
    if (pass_times) {
        const auto t4 = std::chrono::steady_clock::now();
        auto us = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) {
            return std::chrono::duration<double, std::micro>(b - a).count();
        };
        std::fprintf(stderr, "fhx_run_pass: pass_stats %.1f us (launch + wait for K1), fit %.1f, pvalues %.1f (enqueue), bh %.1f (enqueue + wait for K2)\n",
                     us(t0, t1), us(t1, t2), us(t2, t3), us(t3, t4));
    }
    return rc;
}

int fhx_sync(fhx_ctx* ctx) {
    if (!ctx) return FHX_ERR_ARG;
    if (ctx->device < 0) return FHX_OK;
    FHX_HIP(hipSetDevice(ctx->device));
    FHX_HIP(hipStreamSynchronize(ctx->stream));
    return check_fault(ctx);
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
    if (ctx->h_flags) ctx->h_flags[FLAG_FAULT] = 0;
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
        launch_fdr_hist(ctx, ctx->d_q, ctx->n_rows, buckets);
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
        case 0: ctx->k2_hist_valid = false; return ctx->d_p;   // the caller may write p: K3 counts its keys itself next time
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

int fhx_kernel_seconds_total(fhx_ctx* ctx, double* sums4, int64_t* counts4, int reset) {
    if (!ctx) return FHX_ERR_ARG;
    if (ctx->device < 0) return fail(ctx, FHX_ERR_NO_DEVICE, "host-only context");
    FHX_HIP(hipSetDevice(ctx->device));
    FHX_HIP(hipStreamSynchronize(ctx->stream));
    fold_kernel_events(ctx);
    for (int k = 0; k < 4; ++k) {
        if (sums4) sums4[k] = ctx->ev_sum[k];
        if (counts4) counts4[k] = ctx->ev_count[k];
        if (reset) {
            ctx->ev_sum[k] = 0.0;
            ctx->ev_count[k] = 0;
            ctx->ev_dropped[k] = 0;
        }
    }
    return FHX_OK;
}

int fhx_kernel_events_dropped(fhx_ctx* ctx, int64_t* dropped4) {
    if (!ctx || !dropped4) return FHX_ERR_ARG;
    for (int k = 0; k < 4; ++k) dropped4[k] = ctx->ev_dropped[k];
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
