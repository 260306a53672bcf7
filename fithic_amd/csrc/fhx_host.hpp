// fhx_host.hpp - host-side (O(#bins), O(#distances)) stages of the MI355X Fit-Hi-C engine.
//
// Everything here is order-sensitive double arithmetic that the reference does in Python / Cephes /
// FITPACK; it is compiled with -ffp-contract=off so that no a*b+c is fused.
// The O(#pairs) work lives in fhx_kernels.hip.
#pragma once
#include <cstdint>
#include <functional>
#include <string>
#include <vector>

namespace fhx {

// ---------------------------------------------------------------------------------------------------
// Cephes log-gamma / log-beta / beta on the host: feeds the per-count tables K2 reads.
// scipy.special.bdtrc -> incbet -> lbeta(a,b) / beta(a,b)   (call sites fithic/fithic.py:1070,1101)
double cephes_lgam(double x);
double cephes_lbeta(double a, double b);
double cephes_beta(double a, double b);
// table[c] = lbeta(c, n-c+1), inv_beta[c] = 1/beta(c, n-c+1) (only meaningful when n+1 < MAXGAM), c = 0..max_count
void build_lbeta_table(double n_total, int64_t max_count, std::vector<double>& lbeta, std::vector<double>& inv_beta);
void fill_lbeta_table(double n_total, int64_t c_lo, int64_t c_hi, double* lbeta, double* inv_beta);   // entries [c_lo, c_hi] of zero-filled tables

// ---------------------------------------------------------------------------------------------------
// Cubic smoothing spline exactly as scipy.interpolate.UnivariateSpline(x, y, s=s) builds it
// (fithic/fithic.py:951): FITPACK fpcurf with nest = max(m/2, 8) and the iopt=1 continuation when
// FITPACK reports "nest too small".
struct Spline {
    std::vector<double> t;     // knots (n)
    std::vector<double> c;     // B-spline coefficients (n-4)
    double fp = 0.0;
    int ier = 0;
    bool restarted = false;
};
int spline_fit(const double* x, const double* y, int m, double s, Spline& out);
void spline_eval(const Spline& sp, const double* xs, int64_t nx, double* out);   // FITPACK splev, ext = 0

// sklearn IsotonicRegression(increasing=False).fit_transform on distinct X (fithic/fithic.py:965-966)
void pava_decreasing(const double* y, int64_t n, double* out);

// ---------------------------------------------------------------------------------------------------
// Stage logic.
struct Bin {
    int64_t lb = 0, ub = 0;    // binStats[b][0]
    int64_t poss = 0;          // [1]
    int64_t sumcc = 0;         // [2]
    double sumdist = 0.0;      // [3]
    int64_t poss7 = 0;         // [7]
    int64_t poss0 = 0;         // [1] right after makeBinsFromInteractions (outlier decrements only)
};

struct FragTable {             // fragments file reduced to what generate_FragPairs needs
    // one entry per chromosome that occurs in the file, in Python sorted() order of the names
    std::vector<int64_t> n_mappable;      // len(allFragsDic[ch])
    std::vector<int64_t> max_mid;         // max mappable mid (only valid when n_mappable > 0)
    std::vector<int32_t> chr_id;
    // non-fixed-size mode (-r 0) only: the mappable mids of every chromosome, ascending (fithic.py:699)
    std::vector<std::vector<int32_t>> mids;
};

// generate_FragPairs' non-fixed-size branch (fithic.py:691-778) as its caller may supply it: the per-bin sums over every in-range
// pair of mappable fragments.  fhx_fit computes them on the GPU (csrc/fhx_nfpairs.inc) - the bins must exist first, so
// run_host_pass asks for them between makeBinsFromInteractions and the fit; a host-only context walks the pairs on its threads.
struct NfPairSums {
    std::vector<int64_t> poss, poss7;            // binStats[b][1], [7] increments
    std::vector<double> sumdist;                 // binStats[b][3]: the sequential double sum in (chromosome, x, y) order
};

struct PassInputs {
    // (bins after makeBinsFromInteractions) -> sums; false: not available, walk the pairs here
    std::function<bool(const std::vector<Bin>&, NfPairSums&)> nf_pairs;
    int64_t resolution = 0, dist_low = 0, dist_up = INT64_MAX;
    int32_t n_bins = 100;
    int32_t mode = 0;
    // genome-wide distance histogram: index i <-> distance i*resolution
    const int64_t* hist_sumcc = nullptr;
    const int64_t* hist_npairs = nullptr;
    int64_t n_dist = 0;
    int64_t in_range_sum = 0, inter_count = 0, inter_sum = 0;
    // outlier-distance multiset of earlier passes (nullptr in pass 1): count per distance index
    const int64_t* outlier_dist_hist = nullptr;
    // non-fixed-size mode (resolution == 0): entry i of the histogram arrays belongs to distance dist_keys[i] (ascending,
    // distinct) instead of i*resolution, and the outlier multiset is an ascending list of distances
    const int64_t* dist_keys = nullptr;
    const int64_t* outlier_dists = nullptr;
    int64_t n_outlier_dists = 0;
};

struct PassFit {
    std::vector<Bin> bins;
    std::vector<double> x, y;                    // calculateProbabilities, bin order
    Spline spline;
    double spline_s = 0.0, residual = 0.0;
    std::vector<int64_t> table_x;                // splineX
    std::vector<double> table_y0, table_y;       // splineY, newSplineY
    double min_x = 0.0, max_x = 0.0;             // clamp limits of the per-pair lookup
    // generate_FragPairs scalars
    int64_t n_frags = 0, poss_intra_in_range = 0;
    double poss_inter_all = 0.0, poss_intra_all = 0.0, max_possible_dist = 0.0;
    double inter_chr_prob = 0.0, baseline_intra_prob = 0.0;
    double bh_total_tests = 0.0;
    // dense per-distance-index prior LUT for the kernel: lut[i] = newSplineY[min(bisect_left(splineX,
    // clamp(i*res, min x, max x)), len-1)]   (fithic/fithic.py:1066-1069)
    std::vector<double> prior_lut;               // fixed-size mode only; -r 0 searches (table_x, table_y) on the device
};

// makeBinsFromInteractions alone (fithic.py:463-553): fills out.bins (lb, ub, sumcc, outlier decrements)
void make_bins_stage(const PassInputs& in, PassFit& out);
// returns 0, or FHX_ERR_REFERENCE_EXIT with `err` set where the reference would exit / raise
int run_host_pass(const PassInputs& in, const FragTable& frags, PassFit& out, std::string& err);

}  // namespace fhx
