/*
 * fithic_mi355x.h - C ABI of libfithic_mi355x.so, the MI355X-native Fit-Hi-C significance engine.
 *
 * The reference (ay-lab/fithic 2.0.7) is pure Python and has no FFI / plugin interface
 * (SURVEY.md section 8b); the boundary this library honours is the set of Python call signatures on its
 * hot path.  Each entry point below names the reference lines it replaces:
 *
 *   fhx_load_pairs / fhx_load_fragments / fhx_load_bias
 *        the three tables the reference re-reads from gzip text in every stage
 *        (fithic/fithic.py:406-417, :581-590, :805-832), handed over once as host SoA arrays
 *   fhx_pass_stats      fithic.read_Interactions                      (fithic/fithic.py:389-454)      kernel K1
 *   fhx_fit             fithic.makeBinsFromInteractions               (fithic/fithic.py:463-553)
 *                       fithic.generate_FragPairs (fixed-size branch) (fithic/fithic.py:561-689)
 *                       fithic.calculateProbabilities                 (fithic/fithic.py:843-918)
 *                       fithic.fit_Spline, fit + table part           (fithic/fithic.py:936-968)      host C++
 *   fhx_pvalues         fithic.fit_Spline, per-pair loop              (fithic/fithic.py:1017-1124)    kernel K2
 *                       incl. scipy.special.bdtrc                     (call sites :1070, :1101)
 *   fhx_bh              myStats.benjamini_hochberg_correction         (fithic/myStats.py:24-48)       kernels K3
 *                       and its dispatch                              (fithic/fithic.py:1126-1164)
 *   fhx_fetch           the value lists fit_Spline hands to its writer (fithic/fithic.py:1188-1194)
 *   fhx_next_pass       outlier collection for pass >= 2              (fithic/fithic.py:1215-1217, :408-412, :528-548)
 *
 * Conventions: plain C, no exceptions cross the boundary; every function returns 0 (FHX_OK) or a negative
 * error code; fhx_last_error(ctx) returns a message owned by the context.  One context per GPU (or per
 * host thread for device = -1, host-only: everything except the kernels works).  A context is not
 * thread-safe; different contexts are independent.  Host pointers passed in are copied before return.
 */
#ifndef FITHIC_MI355X_H
#define FITHIC_MI355X_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FHX_OK 0
#define FHX_ERR_ARG (-1)           /* bad argument / call order */
#define FHX_ERR_NO_DEVICE (-2)     /* kernel entry point called on a host-only context, or no GPU */
#define FHX_ERR_HIP (-3)           /* a HIP runtime call failed */
#define FHX_ERR_UNSUPPORTED (-4)   /* input outside the accelerated path (e.g. loci off the fixed-size grid) */
#define FHX_ERR_REFERENCE_EXIT (-5)/* the reference would sys.exit(2) / raise here (message says where) */
#define FHX_ERR_NOMEM (-6)
#define FHX_ERR_INTERNAL (-7)      /* a device-side consistency check failed (results would be wrong): message says which */

#define FHX_MODE_INTRA_ONLY 0      /* -x intraOnly (default) */
#define FHX_MODE_INTER_ONLY 1      /* -x interOnly */
#define FHX_MODE_ALL 2             /* -x All */

/* How the binomial of fithic/fithic.py:1070,1101 sees a total (observedIntraInRangeSum / observedInterAllSum) that does not fit
 * a C int.  The reference hands Python ints to scipy.special.bdtrc, whose Cephes core takes `int n`: the total is narrowed to
 * 32 bits (2^31 arrives as -2^31 and every such p-value is NaN; 2^32 + 10^6 arrives as 10^6).
 *   FHX_TOTALS_REFERENCE (default)  the same narrowing: the output equals fithic.py's bit for bit, NaNs included
 *   FHX_TOTALS_WIDE                 bdtrc's arithmetic on the true total (no reference computes this; identical below 2^31)
 * Everything else (expCC, the bin probabilities, BH) uses the true totals in both modes, as the reference does. */
#define FHX_TOTALS_REFERENCE 0
#define FHX_TOTALS_WIDE 1

typedef struct fhx_ctx fhx_ctx;

/* The reference's module globals (fithic/fithic.py:203-260) after its "zero means unset" rule. */
typedef struct fhx_params {
    int64_t resolution;            /* -r; 0 = non-fixed-size data (arbitrary midpoints, fithic/fithic.py:691-778) */
    int64_t dist_low;              /* -L, default 0 */
    int64_t dist_up;               /* -U, INT64_MAX for +inf */
    int32_t n_bins;                /* -b, default 100 */
    int32_t mapp_thres;            /* -m, default 1 */
    int32_t mode;                  /* FHX_MODE_* */
    int32_t totals;                /* FHX_TOTALS_* (this field was `reserved`, always 0 = FHX_TOTALS_REFERENCE) */
    double bias_low;               /* -tL, default 0.5 */
    double bias_up;                /* -tU, default 2 */
} fhx_params;

/* What read_Interactions returns (fithic/fithic.py:454) plus the counters it logs (:447-449). */
typedef struct fhx_stats {
    int64_t n_rows;                /* rows on this context (shard) */
    int64_t inter_count;           /* observedInterAllCount */
    int64_t inter_sum;             /* observedInterAllSum */
    int64_t intra_all_count;       /* observedIntraAllCount */
    int64_t intra_all_sum;         /* observedIntraAllSum */
    int64_t in_range_count;        /* observedIntraInRangeCount */
    int64_t in_range_sum;          /* observedIntraInRangeSum */
    int64_t max_count;             /* largest contact count seen (sizes the lbeta tables) */
    int64_t n_dist;                /* length of the distance histogram: index i <-> distance i*resolution */
    int64_t n_skipped;             /* rows skipped as outliers of earlier passes */
} fhx_stats;

/* Scalars of one fitted pass (generate_FragPairs' return tuple, fit_Spline's spline diagnostics). */
typedef struct fhx_fit_info {
    int32_t n_bins_made;           /* len(binStats) */
    int32_t n_knots;               /* FITPACK n */
    int32_t spline_ier;            /* FITPACK ier */
    int32_t spline_restarted;      /* 1 if the nest-too-small continuation call ran */
    int64_t n_table;               /* len(splineX) */
    int64_t n_frags;               /* noOfFrags */
    int64_t possible_intra_in_range; /* possibleIntraInRangeCount (double-counted as the reference does) */
    double possible_inter_all;     /* possibleInterAllCount (float after /= 2) */
    double possible_intra_all;     /* possibleIntraAllCount */
    double max_possible_dist;      /* maxPossibleGenomicDist */
    double inter_chr_prob;         /* interChrProb */
    double baseline_intra_prob;    /* baselineIntraChrProb */
    double spline_s;               /* min(y)^2 */
    double spline_fp;              /* FITPACK fp */
    double residual;               /* sum((y - ius(x))^2), numpy pairwise order */
    double bh_total_tests;         /* N handed to benjamini_hochberg_correction for this mode */
    double outlier_thres;          /* 1/N */
    int32_t totals;                /* FHX_TOTALS_* this pass ran with */
    int32_t totals_narrowed;       /* bit 0: observedIntraInRangeSum >= 2^31, bit 1: observedInterAllSum >= 2^31 - the two modes differ */
    int64_t bdtrc_n_intra;         /* the n bdtrc was given for in-range cis rows (the narrowed value in reference mode) */
    int64_t bdtrc_n_inter;         /* ... and for inter-chromosomal rows */
} fhx_fit_info;

/* Arrays a caller can copy out with fhx_get_array (all caller-allocated). */
enum fhx_array {
    FHX_A_HIST_SUMCC = 0,          /* int64[n_dist]   sum of counts per distance index (mainDic[d][1]) */
    FHX_A_HIST_NPAIRS = 1,         /* int64[n_dist]   rows per distance index (key present iff > 0) */
    FHX_A_BIN_LB = 2,              /* int64[n_bins_made] */
    FHX_A_BIN_UB = 3,              /* int64[n_bins_made] */
    FHX_A_BIN_POSS = 4,            /* int64   binStats[b][1] */
    FHX_A_BIN_SUMCC = 5,           /* int64   binStats[b][2] */
    FHX_A_BIN_SUMDIST = 6,         /* double  binStats[b][3] */
    FHX_A_BIN_POSS7 = 7,           /* int64   binStats[b][7] */
    FHX_A_X = 8,                   /* double[n_bins_made]  calculateProbabilities x (unsorted) */
    FHX_A_Y = 9,                   /* double[n_bins_made] */
    FHX_A_KNOTS = 10,              /* double[n_knots] */
    FHX_A_COEFFS = 11,             /* double[n_knots-4] */
    FHX_A_TABLE_X = 12,            /* int64[n_table]  splineX */
    FHX_A_TABLE_Y0 = 13,           /* double[n_table] splineY (before isotonic regression) */
    FHX_A_TABLE_Y = 14,            /* double[n_table] newSplineY */
    FHX_A_OUTLIER_DIST_HIST = 15,  /* int64[n_dist]   multiset of outlier distances accumulated so far */
    FHX_A_FDR_COUNTS = 16,         /* int64[51]       plot_qvalues' shifted cumulative counts */
    FHX_A_BIN_POSS0 = 17,          /* int64   binStats[b][1] right after makeBinsFromInteractions */
    FHX_A_DIST_KEYS = 18,          /* int64[n_dist]   -r 0 only: the distinct in-range distances the histogram arrays belong to */
    FHX_A_OUTLIER_DISTS = 19       /* int64[...]      -r 0 only: ascending multiset of outlier distances accumulated so far */
};

/* ---- life cycle --------------------------------------------------------------------------------- */
int fhx_create(int device, fhx_ctx** out);      /* device >= 0: HIP device ordinal; -1: host-only */
void fhx_destroy(fhx_ctx* ctx);
const char* fhx_last_error(fhx_ctx* ctx);
const char* fhx_version(void);
/* The HIP runtime and the device's primary context brought up ahead of fhx_create (0.15-0.25 s the first time in a process);
 * a launcher calls it on a thread while it does other start-up work.  Optional: fhx_create does the same when needed. */
int fhx_warmup(int device);
int fhx_set_params(fhx_ctx* ctx, const fhx_params* p);

/* ---- tables (host SoA in, copied) --------------------------------------------------------------- */
/* Chromosome ids are small non-negative ints chosen by the caller, shared by all three tables.
 * chr_sort_rank[id] = position of that chromosome NAME in Python's sorted() order: the reference walks
 * chromosomes in that order (fithic.py:606) and its double accumulators depend on it. */
int fhx_load_fragments(fhx_ctx* ctx, const int32_t* chr, const int32_t* mid, const int32_t* hits, int64_t n,
                       const int32_t* chr_sort_rank, int32_t n_chr);
int fhx_load_bias(fhx_ctx* ctx, const int32_t* chr, const int32_t* mid, const double* bias, int64_t n);
int fhx_load_pairs(fhx_ctx* ctx, const int32_t* chr1, const int32_t* mid1, const int32_t* chr2,
                   const int32_t* mid2, const int32_t* count, int64_t n);
/* Same, but the five arrays already live in this GPU's memory (e.g. written by a generator kernel or
 * received over xGMI).  `stream` = the hipStream_t the arrays were produced on, NULL = the legacy default stream
 * (PyTorch's, unless the caller chose another): the call waits for it before reading the arrays. */
int fhx_load_pairs_device(fhx_ctx* ctx, const void* d_chr1, const void* d_mid1, const void* d_chr2,
                          const void* d_mid2, const void* d_count, int64_t n, void* stream);

/* The contacts file parsed BY THE GPU (csrc/fhx_ingest.inc): the inflated text (fhx_host_inflate) goes to HBM as it is and
 * kernels find the lines, read the five fields (fithic/fithic.py:404-417: ch1, int(mid1), ch2, int(mid2),
 * int(float(contactCount))) and collect the chromosome names in the reference's order of first appearance.  Two calls, because
 * the caller owns the id space the names map into (it is shared with the fragments and bias files):
 *   fhx_ingest_contacts_text    parse; n_rows, n_names out, fhx_ingest_contacts_name(ctx, i) = name i (valid until the commit)
 *   fhx_ingest_contacts_commit  ids[i] = the caller's id of name i; the rows enter the context as fhx_load_pairs would have
 *                               loaded them (fhx_set_params must have been called, as for fhx_load_pairs)
 * Only the regular file is taken - ASCII, five tokens per line, names of at most 63 bytes, [+-]digits midpoints and a
 * digits[.digits] count (15 digits at most) within int32.  For every other file (an exponent, an underscore in a number, a malformed line whose
 * error message must name it, ...) fhx_ingest_contacts_text returns FHX_ERR_UNSUPPORTED with nothing loaded, and the caller
 * gives the same fhx_text to fhx_host_parse_text, whose grammar is Python's.  fhx_ingest_contacts_discard drops a parsed text
 * that will not be committed. */
struct fhx_text;
int fhx_ingest_contacts_text(fhx_ctx* ctx, const struct fhx_text* text, int32_t n_threads, int64_t* n_rows, int32_t* n_names);
const char* fhx_ingest_contacts_name(const fhx_ctx* ctx, int32_t i);
/* The same straight from the FILE, for a file of size-tagged gzip members ("FH" of this library's writers, "BC" of bgzip):
 * the compressed bytes are uploaded and every member is inflated by the GPU (csrc/fhx_inflate.inc: one wave per member, CRC-32
 * and ISIZE checked), so the text exists in HBM only.  FHX_ERR_UNSUPPORTED with *refused = 1: not such a container, or a
 * stream the device decoder does not accept - fhx_host_inflate (zlib; it reports what the reference's gzip module would) and
 * fhx_ingest_contacts_text may still take it; *refused = 2: the text is outside the device parser's grammar -
 * fhx_host_read_table is the path. */
int fhx_ingest_contacts_file(fhx_ctx* ctx, const char* path, int32_t n_threads, int64_t* n_rows, int32_t* n_names, int32_t* refused);
/* test hook: the text of such a file as the device decoder produces it (n_out = its size; cap = room in out) */
int fhx_debug_inflate_file(fhx_ctx* ctx, const char* path, void* out, int64_t cap, int64_t* n_out);
int fhx_ingest_contacts_commit(fhx_ctx* ctx, const int32_t* ids, int32_t n_ids);
/* `fithic --gpus N` without a funnel (every rank parses the file on its own GPU, fithic/fithic.py:404-417 once per rank):
 *   fhx_ingest_contacts_chr_counts    rows of the parsed text per name, counted by the FIRST locus's chromosome (what decides which
 *                                     rank a row belongs to: the ranks agree on the owners because they count the same file)
 *   fhx_ingest_contacts_commit_shard  as fhx_ingest_contacts_commit, but only the rows whose first chromosome has mine[i] != 0
 *                                     enter the context, in file order, with their file positions (fhx_set_global_rows)
 *   fhx_shard_segments                the maximal stretches of consecutive file positions among the loaded rows: local_start,
 *                                     file_start, length of each, ordered; *n_out > cap: too many (nothing filled in)          */
/* The cheaper split (round 5, last step): the ranks cut the FILE, not the genome.  fhx_ingest_contacts_file_slice inflates and
 * parses part `part` of `n_parts` (whole gzip members, cut where the compressed bytes divide evenly; the members must carry their
 * sizes, else *refused = 1; *ends_with_newline: the part's text ends a row - required of every part but the last);
 * fhx_ingest_contacts_commit takes its rows; fhx_set_global_rows_range(first) numbers them first, first + 1, ... (first = the rows of
 * the parts before it).  Nothing in the engine needs a chromosome's rows on one rank. */
int fhx_ingest_contacts_file_slice(fhx_ctx* ctx, const char* path, int32_t n_threads, int32_t part, int32_t n_parts, int64_t* n_rows,
                                   int32_t* n_names, int32_t* refused, int32_t* ends_with_newline);
int fhx_set_global_rows_range(fhx_ctx* ctx, int64_t first);
/* The same for a file the device does not inflate (plain gzip): every rank inflates it (fhx_host_inflate), then uploads and parses
 * only the rows that start in its N-th of the text's bytes - every row in exactly one part. */
int fhx_ingest_contacts_text_slice(fhx_ctx* ctx, const struct fhx_text* text, int32_t n_threads, int32_t part, int32_t n_parts, int64_t* n_rows,
                                   int32_t* n_names);
/* bytes [*lo, *hi) of the text that part takes (host only; parts can be empty) */
int fhx_text_part_bounds(const struct fhx_text* text, int32_t part, int32_t n_parts, int64_t* lo, int64_t* hi);
/* Round 6: the ranks of a sharded run inflate ONE plain gzip stream together - rank r its N-th of the COMPRESSED bytes (1/N of the
 * inflate work and of the text in host memory per rank; with fhx_host_inflate every rank did all of it).  Replaces the one
 * gzip.open of fithic/fithic.py:404 for `fithic --gpus N`.  Protocol (fithic_amd/sharded.py: _ingest_stream_parts):
 *   fhx_host_inflate_part      this rank's chunks decoded without the 32 KB before them (csrc/fhx_gunzip.cpp); FHX_ERR_UNSUPPORTED:
 *                              not one plain stream / too small / a stream the decoder refuses - the caller takes the old route
 *   fhx_text_part_tail         the part's last 32768 symbols in terms of the window before the part (< 256: a byte, 256 + j: byte j
 *                              of that window): the caller chains them rank after rank into every part's window
 *   fhx_text_part_resolve      window -> the part's text as a new fhx_text object - the part object is left empty - and its CRC-32; the caller
 *                              checks fhx_crc32_combine of all parts and the summed lengths against the file's trailer
 *   fhx_text_first_row_end     length of the text's first (partial) row, newline included; -1: no newline in the text
 *   fhx_ingest_contacts_text_own   the text from byte `skip` on + `extra` (the rest of its last row: the head of the next rank's text)
 *                              uploaded and parsed: a row belongs to the rank whose text holds its first byte */
typedef struct fhx_text_part fhx_text_part;
int fhx_host_inflate_part(const char* path, int32_t n_threads, int32_t part, int32_t n_parts, fhx_text_part** out);
int64_t fhx_text_part_bytes(const fhx_text_part* p);
int32_t fhx_text_part_is_last(const fhx_text_part* p);
int fhx_text_part_tail(const fhx_text_part* p, uint16_t* tail, int64_t cap);                   /* cap >= 32768 */
int fhx_text_part_resolve(fhx_text_part* p, const uint8_t* window, int64_t window_bytes, struct fhx_text** text_out, uint32_t* crc32_out);
const char* fhx_text_part_error(const fhx_text_part* p);
void fhx_text_part_free(fhx_text_part* p);
uint32_t fhx_crc32_combine(uint32_t crc_a, uint32_t crc_b, int64_t len_b);
int64_t fhx_text_first_row_end(const struct fhx_text* text, char* row, int64_t cap);            /* row may be NULL */
int32_t fhx_text_ends_with_newline(const struct fhx_text* text);
int fhx_ingest_contacts_text_own(fhx_ctx* ctx, const struct fhx_text* text, int32_t n_threads, int64_t skip, const char* extra,
                                 int64_t extra_bytes, int64_t* n_rows, int32_t* n_names);
int fhx_ingest_contacts_chr_counts(fhx_ctx* ctx, int64_t* counts, int32_t n_names);
int fhx_ingest_contacts_commit_shard(fhx_ctx* ctx, const int32_t* ids, const uint8_t* mine, int32_t n_ids, int64_t* n_kept);
int fhx_shard_segments(fhx_ctx* ctx, int64_t* local_start, int64_t* file_start, int64_t* length, int64_t cap, int64_t* n_out);
void fhx_ingest_contacts_discard(fhx_ctx* ctx);
/* The identity columns of loaded rows, rebuilt from the resident rows (slot -> chromosome id, midpoint): rows = n row
 * numbers, or NULL for all rows in order (n must then be the loaded row count).  What a caller that ingested on the device
 * uses to look at a few rows (the outlier lines) without ever holding the columns on the host. */
int fhx_fetch_pairs(fhx_ctx* ctx, const int64_t* rows, int64_t n, int32_t* chr1, int32_t* mid1, int32_t* chr2, int32_t* mid2,
                    int32_t* count);

/* ---- one spline pass ---------------------------------------------------------------------------- */
int fhx_pass_stats(fhx_ctx* ctx, fhx_stats* out);                       /* K1, then waits for the sums */
int fhx_get_stats(fhx_ctx* ctx, fhx_stats* out);                        /* the statistics the current fit was made from */
/* Distributed runs: replace the local histogram / sums by the all-reduced ones before fhx_fit. */
int fhx_set_global_stats(fhx_ctx* ctx, const fhx_stats* global_stats, const int64_t* hist_sumcc,
                         const int64_t* hist_npairs, int64_t n_dist);
/* -r 0 only (host-side testing / callers that bring their own histogram): the distinct distances of the histogram arrays
 * handed to fhx_set_global_stats, and the ascending multiset of outlier distances of earlier passes. */
int fhx_set_dist_keys(fhx_ctx* ctx, const int64_t* keys, int64_t n);
int fhx_set_outlier_dists(fhx_ctx* ctx, const int64_t* dists, int64_t n);
/* Distributed runs, pass >= 2: replace the local multiset of outlier distances (count per distance index,
 * accumulated over all earlier passes) by the all-reduced one. */
int fhx_set_outlier_dist_hist(fhx_ctx* ctx, const int64_t* hist, int64_t n_dist);
/* makeBinsFromInteractions alone (host): bins are readable through fhx_get_array(FHX_A_BIN_LB/UB/SUMCC/POSS0). */
int fhx_make_bins(fhx_ctx* ctx, int32_t* n_bins_made);
int fhx_fit(fhx_ctx* ctx, fhx_fit_info* out);                           /* host; uploads the tables */
int fhx_pvalues(fhx_ctx* ctx);                                          /* K2 (asynchronous) */
int fhx_bh(fhx_ctx* ctx, double n_total_tests);                         /* K3 (asynchronous); any finite N: N <= 0 gives
                                                                          * q = 0 like the reference's loop (running max from 0) */
/* The four calls above as one: K1 -> host fit -> K2 -> K3 of the current pass (asynchronous like fhx_bh; stats and info may be
 * NULL).  What a driver's loop over passes calls: main() of fithic/fithic.py:317-370 between read_Interactions and the writer. */
int fhx_run_pass(fhx_ctx* ctx, fhx_stats* stats, fhx_fit_info* info);
int fhx_sync(fhx_ctx* ctx);
/* Sharded runs with -p >= 3 only.  The reference stops skipping outlier lines after the first line number that
 * is an outlier in two passes (fithic/fithic.py:408-412 on a SortedList with duplicates); that is a
 * statement about positions in the ONE input file, so a shard must know its rows' file positions
 * (fhx_set_global_rows, ascending not required) and the ranks must agree on the limit (min over ranks of
 * fhx_get_skip_limit, handed back with fhx_set_skip_limit; INT64_MAX = no duplicate yet). */
int fhx_set_global_rows(fhx_ctx* ctx, const int64_t* file_row_of_local_row, int64_t n);
int64_t fhx_get_skip_limit(fhx_ctx* ctx);
int fhx_set_skip_limit(fhx_ctx* ctx, int64_t limit);
int fhx_next_pass(fhx_ctx* ctx, int64_t* n_outliers_total);             /* fold this pass's outliers in */
/* Forget every pass: the outlier line set and distance multiset are emptied, the next fhx_pass_stats is pass 1 again.
 * The reference equivalent is starting main() over on the same input (fithic/fithic.py:317-331 builds fresh
 * SortedLists); the loaded tables stay on the device. */
int fhx_reset_passes(fhx_ctx* ctx);

/* ---- results ------------------------------------------------------------------------------------ */
/* Any pointer may be NULL.  Arrays have n_rows entries in input row order. */
int fhx_fetch(fhx_ctx* ctx, double* p, double* q, double* expcc, double* bias1, double* bias2);
/* Per-row byte flags: outlier[i] = (p_i < 1/N) of the last fhx_pvalues; skip[i] = row is skipped by fhx_pass_stats
 * (outlier of an earlier pass, fithic/fithic.py:408-412).  Either pointer may be NULL. */
int fhx_fetch_flags(fhx_ctx* ctx, uint8_t* outlier, uint8_t* skip);
/* The row numbers (file order, ascending) of the rows fhx_fetch_flags would flag as outliers, compacted on the device.
 * rows == NULL or cap < *n_out: only the count is returned. */
int fhx_fetch_outlier_rows(fhx_ctx* ctx, int64_t* rows, int64_t cap, int64_t* n_out);
int fhx_get_array(fhx_ctx* ctx, int which, void* dst, int64_t capacity_elems, int64_t* n_out);
/* Raw device pointers for plumbing (torch / RCCL exchange): 0 = p, 1 = q, 2 = sorted keys, 3 = sorted idx */
void* fhx_device_ptr(fhx_ctx* ctx, int which);
int64_t fhx_n_sorted(fhx_ctx* ctx);
/* How the last Benjamini-Hochberg call sorted its surviving p-values (diagnostics, tests): out8 = radix passes of the first
 * attempt (0: the small in-LDS sort ran), lowest key bit they covered, what the repair of the lower bits could not list (0 =
 * nothing; bit 0: a run whose ends were out of reach, bit 1: a list overflowed, bit 2: forced by FHX_OS_FORCE_FALLBACK, bit 3:
 * every bit of every key was sorted after all), inversions met, runs sorted by one thread each, inversions found in runs too
 * long for that, such runs sorted as segments of their own, keys in them. */
int fhx_bh_sort_stats(fhx_ctx* ctx, int64_t* out8);
/* Seconds the kernels of the last pass took on the context's stream (HIP events): k1, k2, k3. */
int fhx_kernel_seconds(fhx_ctx* ctx, double* k1, double* k2, double* k3);
/* The same summed over the passes since the last reset, without stopping the stream after each of them: sums4 = seconds of K1, K2,
 * K3 and of the heavy K2 launch (fhx_k2_heavy_launch), counts4 = how many passes each sum holds (either may be NULL).  The library
 * reads a pass's events at the next point where it waits for the stream anyway (the statistics of the following pass), this call
 * waits for the stream and reads the rest; reset != 0 clears the sums afterwards.  A timing harness calls it once with reset
 * before and once after its timed passes. */
int fhx_kernel_seconds_total(fhx_ctx* ctx, double* sums4, int64_t* counts4, int reset);
/* Passes whose event pair was recorded again before it could be read (a kernel group run twice with no statistics call and no
 * fhx_kernel_seconds_total in between, while the stream had not yet passed the first pair): dropped4 = K1, K2, K3, heavy launch,
 * since the last reset.  Zero means the sums above cover every pass that ran. */
int fhx_kernel_events_dropped(fhx_ctx* ctx, int64_t* dropped4);
/* Duration (HIP events on the context's stream) and row count of the dominant launch of the last fhx_pvalues: the queue
 * of rows whose continued fraction runs to Cephes' 300-iteration cap (k2_queue<BC_CF_SWAPPED>). */
int fhx_k2_heavy_launch(fhx_ctx* ctx, double* seconds, int64_t* rows);
/* The shader clock (GHz) that launch ran at, sampled by one of its waves (cycle counter over the constant-rate counter): what turns
 * a count of VALU wave-instructions into the milliseconds of an issue-bound launch on THIS box (bench.py: valu_issue_floor_ms). */
int fhx_k2_heavy_clock(fhx_ctx* ctx, double* ghz);
/* Rows the last fhx_pvalues queued per branch class of incbet (the per-pair branch table of fithic/fithic.py:1057-1116 followed
 * into Cephes): out5 = power series, converging incbcf, incbd, swapped incbcf (the 300-iteration class), closed form with
 * prior >= 0.01.  Every other row was finished by the classification launch itself (constants, closed form with a small prior).
 * Measurement aid: per-kernel bytes and rows of bench.py / profiles. */
int fhx_k2_class_rows(fhx_ctx* ctx, int64_t* out5);

/* myStats.benjamini_hochberg_correction(p_values, num_total_tests) on an arbitrary host array (fithic/myStats.py:24-48):
 * copies p to the GPU, runs the K3 kernels, copies q back (input order). */
int fhx_bh_array(fhx_ctx* ctx, const double* p, int64_t n, double n_total_tests, double* q);

/* scipy.special.bdtrc(count - 1, n_total, prior) element-wise on the GPU for integer counts (the only form the
 * reference uses, fithic/fithic.py:1070,1101); host arrays in and out.  Known-answer testing of K2's arithmetic.  An integral
 * n_total is narrowed to a C int as scipy does (FHX_TOTALS_REFERENCE, also for a context without parameters) unless the
 * context's parameters say FHX_TOTALS_WIDE. */
int fhx_bdtrc_array(fhx_ctx* ctx, double n_total, const int32_t* count, const double* prior, int64_t n, double* out);

/* Test hook: the raw Cephes continued fraction (kind 0 = incbcf, 1 = incbd) of K2 evaluated element-wise on the GPU,
 * either with Cephes' literal convergence test (lazy = 0) or with the division-free test K2 uses (lazy = 1); the two
 * must agree bit for bit.  Host arrays in and out. */
int fhx_debug_contfrac(fhx_ctx* ctx, int kind, int lazy, const double* a, const double* b, const double* x, int64_t n,
                       double* out);

/* Test hook: the branch class K2 assigns to (count, prior) under the binomial total n_total - by the per-count threshold table
 * the classify kernel reads (by_table) and by the predicates of Cephes' incbet evaluated directly (by_arith); the two must agree
 * for every double.  thr5 (optional, n x 5): the thresholds tA, tB, tC, tD, tE of each count. */
int fhx_debug_classify(fhx_ctx* ctx, double n_total, const int32_t* count, const double* prior, int64_t n, int32_t* by_table,
                       int32_t* by_arith, double* thr5);
/* Test hook: out[i] = K2's lean division of n[i] / d[i] (must equal IEEE n/d inside the operand window it is used in). */
int fhx_debug_lean_div(fhx_ctx* ctx, const double* n, const double* d, int64_t len, double* out);
/* Test hook: the device writer's number formatting on arbitrary doubles - kind 0 "%e", 1 "%f"; text32 receives 32 bytes per value
 * (zero padded), len the character count, or -1 where the device formatter hands the row to the host (|v| >= 2^64 / 2^63). */
int fhx_debug_format(fhx_ctx* ctx, const double* values, int64_t n, int32_t kind, char* text32, int32_t* len);

/* ---- distributed BH building blocks (section 8e): local sort, then rank/scan over a global segment -- */
/* Early cutoff (exact): the reference's q is a FORWARD running max of min(p*N/rank, 1), so every p at or above the first
 * key whose value*N/rank bound reaches 1 has q = 1 and need not be sorted.  fhx_bh computes that key from a coarse
 * histogram of its own p-values; sharded runs all-reduce the 8192-bin histograms (fhx_bh_top_hist) and hand the sum
 * back (fhx_bh_set_cutoff) before fhx_bh_local_sort.  Without either call every p < 1 is sorted. */
int fhx_bh_top_hist(fhx_ctx* ctx, int64_t* hist_out, int64_t capacity);
int fhx_bh_set_cutoff(fhx_ctx* ctx, const int64_t* global_hist, int64_t n_bins, double n_total_tests);
/* The same two steps with the histogram left in HBM: fhx_device_ptr(ctx, 4) is the 8192 x uint64 table; the caller
 * all-reduces it in place between the two calls (no host round trip). */
int fhx_bh_top_hist_device(fhx_ctx* ctx);
int fhx_bh_set_cutoff_device(fhx_ctx* ctx, double n_total_tests);
int fhx_bh_local_sort(fhx_ctx* ctx);                 /* compact p below the cutoff, radix sort (key, row) on this GPU */
int fhx_bh_apply_sorted(fhx_ctx* ctx, const void* d_sorted_keys, int64_t n, int64_t global_rank0,
                        double carry_in, double n_total_tests, void* d_q_sorted, double* block_max_out);
/* sort n 64-bit keys that live on this GPU (ascending, stable); d_perm_out[i] = original position of sorted element i */
int fhx_sort_u64(fhx_ctx* ctx, const void* d_keys_in, int64_t n, void* d_keys_out, void* d_perm_out);
/* q[row of local sorted element i] = d_q_sorted_local[i] for the fhx_n_sorted() elements of fhx_bh_local_sort */
int fhx_bh_scatter(fhx_ctx* ctx, const void* d_q_sorted_local);
/* device-to-device copy on the context's stream, then wait (plumbing between the library's buffers and torch's) */
int fhx_memcpy_d2d(fhx_ctx* ctx, void* dst, const void* src, int64_t bytes);

/* ---- multi-GPU runs behind the boundary (SURVEY 8b last row, 8e) -----------------------------------------------------
 * One process (or thread) per GPU, one fhx_ctx each, the contact rows sharded over the contexts (by chromosome, or any
 * other split: every exchange below is row-order free).  The reference is one process, so what must be global is what its
 * data structures make global: mainDic and the sums of read_Interactions (fithic/fithic.py:434-440) -> ONE all-reduce of
 * [sums | in-range histogram windows] in HBM; the global ascending order of benjamini_hochberg_correction
 * (fithic/myStats.py:27-46) -> sample sort: all-reduced 8192-bin key histogram (exact early cutoff), all-gather of
 * regular samples, splitters, all-to-all of the keys, local sort of the received slice, all-gather of the slice maxima
 * (carry of the running max), all-to-all of q back; the outlier multiset of pass >= 2 (fithic.py:528-548) -> all-reduce.
 * All collectives are enqueued on the context's stream between the kernels; the host waits twice per pass (for the summed
 * statistics the host fit needs, and for the send/receive counts of the key exchange).
 *
 * Transport 1 - RCCL over xGMI (production): rank 0 calls fhx_comm_unique_id, hands the bytes to the other ranks by
 * whatever means started them (environment, file, socket, MPI ...), every rank calls fhx_comm_init.  librccl is loaded at
 * run time (the copy already in the process, e.g. PyTorch's, else the system's), so single-GPU users need no RCCL.
 * Transport 2 - caller-provided collectives on device pointers (tests on a one-GPU box, other fabrics): fhx_comm_init_custom.
 * The library synchronises its stream before each callback; a callback returns when its result is in place. */
#define FHX_UNIQUE_ID_BYTES 128
int fhx_comm_unique_id(void* id_out, int64_t capacity);                 /* ncclGetUniqueId */
int fhx_comm_init(fhx_ctx* ctx, const void* rccl_unique_id, int rank, int nranks);   /* ncclCommInitRank on ctx's device */
typedef struct fhx_transport {
    void* user;
    /* in-place reduction of n int64 values at device address d_buf; op 0 = sum, 1 = max, 2 = min */
    int (*all_reduce_i64)(void* user, void* d_buf, int64_t n, int op);
    /* d_recv[r * bytes .. (r + 1) * bytes) = rank r's d_send[0 .. bytes) */
    int (*all_gather)(void* user, const void* d_send, void* d_recv, int64_t bytes);
    /* counts and offsets in elements of elem_bytes, one entry per rank */
    int (*all_to_all_v)(void* user, const void* d_send, const int64_t* send_counts, const int64_t* send_offsets, void* d_recv,
                        const int64_t* recv_counts, const int64_t* recv_offsets, int elem_bytes);
} fhx_transport;
int fhx_comm_init_custom(fhx_ctx* ctx, const fhx_transport* t, int rank, int nranks);
int fhx_comm_destroy(fhx_ctx* ctx);
/* rank, world size and the RCCL version code (0 for a custom transport) of an initialised communicator */
int fhx_comm_info(fhx_ctx* ctx, int* rank, int* nranks, int* rccl_version);
/* One spline pass over all ranks' rows: K1 -> all-reduce -> host fit (identical on every rank) -> K2 -> global BH; leaves
 * p and q of the LOCAL rows in row order, as fhx_pass_stats + fhx_fit + fhx_pvalues + fhx_bh do on one GPU.
 * Fixed-size loci only (-r 0 runs go through the building blocks above). */
int fhx_run_pass_distributed(fhx_ctx* ctx, fhx_fit_info* out);
/* The same pass stage by stage, for callers that keep the reference's stage order (read_Interactions ... fit_Spline):
 * fhx_pass_stats_distributed (K1 + the one all-reduce; afterwards stats and histograms are the genome-wide ones on every rank)
 * -> fhx_make_bins / fhx_fit / fhx_pvalues on every rank as on one GPU -> fhx_bh_distributed (global ranking). */
int fhx_pass_stats_distributed(fhx_ctx* ctx, fhx_stats* out);
int fhx_bh_distributed(fhx_ctx* ctx, double n_total_tests);
/* fhx_next_pass on every rank + the genome-wide outlier multiset, outlier count and first duplicated line */
int fhx_next_pass_distributed(fhx_ctx* ctx, int64_t* n_outliers_total);
/* host wall seconds per stage of the last fhx_run_pass_distributed: k1 + stats exchange, host fit, K2 launch, cutoff +
 * local sort + splitters (up to the count exchange), key exchange + slice sort + scan + q return */
int fhx_dist_stage_seconds(fhx_ctx* ctx, double* out5);
/* The collectives this context issued since the communicator was made (or since the last call with clear != 0), one text line
 * each: "<step> <kind> <size>" with the step ids, kinds and size units of fithic_amd/csrc/fhx_dist_schedule.def - the one list
 * of what a sharded pass exchanges.  Recorded only when FHX_DIST_TRACE=1 was set at fhx_comm_init(_custom) time.  *n_bytes
 * receives the length of the text; nothing is copied (or cleared) when capacity is smaller. */
int fhx_dist_trace(fhx_ctx* ctx, char* buf, int64_t capacity, int64_t* n_bytes, int clear);
/* plain copies between host and this context's GPU (plumbing for custom transports): kind 0 = host to device,
 * 1 = device to host, 2 = device to device; waits for completion */
int fhx_copy(fhx_ctx* ctx, void* dst, const void* src, int64_t bytes, int kind);

/* ---- host numerics, exported for tests and for callers that only need the host side ------------------ */
int fhx_host_spline_fit(const double* x, const double* y, int32_t m, double s, double* t, double* c,
                        int32_t* n_knots, double* fp, int32_t* ier, int32_t* restarted);
int fhx_host_spline_eval(const double* t, const double* c, int32_t n_knots, const double* xs, int64_t nx, double* out);
int fhx_host_pava_decreasing(const double* y, int64_t n, double* out);
int fhx_host_lbeta_table(double n_total, int64_t max_count, double* lbeta_out, double* inv_beta_out);

/* ---- native text I/O (host only) -------------------------------------------------------------------
 * Reader for the three gzip text tables (fithic/fithic.py:406-417 contacts, :581-590 fragments, :805-808 bias):
 * kind 0 = contacts (chr1 mid1 chr2 mid2 count), 1 = fragments (uses columns 0, 2, 3), 2 = bias (chr mid bias).
 * Chromosome names are interned in order of first appearance.  Columns for fhx_table_copy: 0 chr1, 1 mid1, 2 chr2, 3 mid2,
 * 4 count = int(float(text)) or hits (int32 each); 5 = the float itself / the bias (double).  A malformed line returns
 * FHX_ERR_REFERENCE_EXIT (the reference raises ValueError there); fhx_table_error gives the line number. */
typedef struct fhx_table fhx_table;
#define FHX_TABLE_NO_FLOAT 0x100   /* kind 0 | FHX_TABLE_NO_FLOAT: a contacts table without column 5 (8 B/row less to parse and keep) */
int fhx_host_read_table(const char* path, int32_t kind, int32_t n_threads, fhx_table** out);
int64_t fhx_table_rows(const fhx_table* t);
int32_t fhx_table_n_names(const fhx_table* t);
const char* fhx_table_name(const fhx_table* t, int32_t i);
const char* fhx_table_error(const fhx_table* t);
int fhx_table_copy(const fhx_table* t, int32_t column, void* dst);
/* ids[i] = the caller's id of fhx_table_name(t, i): columns 0 and 2 of later fhx_table_copy calls are in that id space. */
int fhx_table_map_names(fhx_table* t, const int32_t* ids, int32_t n_ids);
void fhx_table_free(fhx_table* t);
/* The two stages of fhx_host_read_table on their own.  fhx_host_inflate: the file read and inflated on the host cores - a file
 * of size-tagged members by one zlib per core, ONE plain gzip stream (what `gzip` writes) by csrc/fhx_gunzip.cpp: block starts
 * found by their headers, chunks decoded in parallel with the unknown 32 KB window as 16-bit symbols, windows resolved down the
 * chain, CRC-32 and ISIZE checked, zlib on one thread whenever any of that does not work out (an
 * object is returned even on failure, for fhx_text_error); fhx_host_parse_text: the table of that text.  The split exists for
 * fhx_ingest_contacts_text below, which parses the text on the GPU and leaves fhx_host_parse_text as the path for the files
 * it does not take. */
typedef struct fhx_text fhx_text;
int fhx_host_inflate(const char* path, int32_t n_threads, fhx_text** out);
int64_t fhx_text_bytes(const fhx_text* x);
int fhx_text_copy(const fhx_text* x, void* dst, int64_t cap);      /* the inflated bytes (cap >= fhx_text_bytes) */
const char* fhx_text_error(const fhx_text* x);
int fhx_host_parse_text(const fhx_text* text, int32_t kind, int32_t n_threads, fhx_table** out);
void fhx_text_free(fhx_text* x);
/* The contacts table written as the reference reads it ("%s\t%d\t%s\t%d\t%d\n", fithic/fithic.py:413-417) on all cores;
 * tooling for synthetic workloads.  Like every file this library writes it is a concatenation of gzip members that carry
 * their compressed size in an "FH" extra subfield - plain gzip to every other reader, inflated in parallel by
 * fhx_host_read_table (which does the same for bgzip's "BC" blocks). */
int fhx_host_write_contacts(const char* path, const char* const* chr_names, int32_t n_names, const int32_t* chr1,
                            const int32_t* mid1, const int32_t* chr2, const int32_t* mid2, const int32_t* count, int64_t n_rows,
                            int32_t gzip_level, int32_t n_threads);
/* Writer of <lib>.spline_passN.resR.significances.txt.gz (fithic/fithic.py:1167-1213): header + one
 * "%s\t%d\t%s\t%d\t%d\t%e\t%e\t%e\t%e\t%f" row per emitted contact (inter rows in All / interOnly mode, in-range intra
 * rows in All / intraOnly mode).  Rows are formatted and deflated in parallel, one gzip member per 65 536 rows; the
 * decompressed bytes equal the reference's.  n_threads <= 0: all host cores. */
int fhx_host_write_significances(const char* path, const char* const* chr_names, int32_t n_names, const int32_t* chr1,
                                 const int32_t* mid1, const int32_t* chr2, const int32_t* mid2, const int32_t* count,
                                 const double* p, const double* q, const double* bias1, const double* bias2,
                                 const double* expcc, int64_t n_rows, int32_t mode, int64_t dist_low, int64_t dist_up,
                                 int32_t gzip_level, int32_t n_threads, int64_t* rows_written);
/* The same file written BY THE GPU, for the p and q a context holds after fhx_pvalues + fhx_bh: rows are formatted in a
 * kernel (exact integer arithmetic: the characters C's printf gives), ExpCC and the biases recomputed there, and deflated in
 * kernels (tokens against the previous row + per-member dynamic Huffman codes built by the host from device histograms);
 * the host receives finished gzip members (same container as above) and only writes them.  The five identity columns are
 * the rows given to fhx_load_pairs (n_rows must equal the loaded row count); pass all five as NULL and they are rebuilt on
 * the device from the resident rows (slot -> chromosome, midpoint), so that nothing but the file leaves the GPU.  Returns FHX_ERR_UNSUPPORTED - nothing usable
 * written - when a row does not fit the device formatter (a chromosome name longer than 24 bytes, a value of 2^63 or more,
 * a row of 192 bytes or more): the caller then fetches the columns and uses fhx_host_write_significances. */
int fhx_write_significances_device(fhx_ctx* ctx, const char* path, const char* const* chr_names, int32_t n_names,
                                   const int32_t* chr1, const int32_t* mid1, const int32_t* chr2, const int32_t* mid2,
                                   const int32_t* count, int64_t n_rows, int64_t* rows_written, int64_t* bytes_written);
/* A stretch [row_begin, row_end) of the loaded rows written the same way (identity columns rebuilt on the device), as gzip members
 * of their own, with or without the column-header member in front.  The ranks of `fithic --gpus N` each write the stretches of the
 * file they hold; concatenated in file order the pieces are the reference's file (fithic/fithic.py:1167-1219). */
int fhx_write_significances_device_range(fhx_ctx* ctx, const char* path, const char* const* chr_names, int32_t n_names, int64_t row_begin,
                                         int64_t row_end, int32_t with_header, int64_t* rows_written, int64_t* bytes_written);

/* ---- Knight-Ruiz bias vectors (fithic/utils/HiCKRy.py; SURVEY 8f rank 4, the step before Fit-Hi-C) --------------------
 * One fhx_kr per GPU, independent of fhx_ctx.  Call order: load_loci -> load_pairs -> [remove_sparse] -> balance -> bias.
 * Summation orders are fixed by the kernels (csrc/fhx_kr.hip header) and restated by oracle/kr_oracle.c. */
typedef struct fhx_kr fhx_kr;
typedef struct fhx_kr_info {
    int64_t n, nnz;                 /* matrix that was balanced (after row removal) */
    int32_t outer_iterations;       /* `i` of knightRuizAlg's return value (31 = the 30-iteration cap was hit, HiCKRy.py:174) */
    int32_t inner_iterations;       /* `k` of the last outer iteration */
    int64_t matvecs;                /* A.dot calls */
    int64_t boundary_steps;         /* inner loops that ended on the cone boundary (HiCKRy.py:199-212) */
    double residual;                /* rout = |1 - x*(A x)|^2 at exit */
    double spmv_seconds;            /* HIP-event time of the timed SpMV launches */
    int64_t spmv_timed;
    int64_t value_bytes;            /* bytes per value streamed by the SpMV: 8 doubles; 4 binary32 (exact for integer counts), with 4-byte
                                     * columns either way; 2: the 4-byte cell (16-bit count + 16-bit column offset in its row) */
} fhx_kr_info;
int fhx_kr_create(int device, fhx_kr** out);
void fhx_kr_destroy(fhx_kr* kr);
const char* fhx_kr_last_error(const fhx_kr* kr);
/* fragments file, HiCKRy.py:21-33: locus index = line number; a repeated (chr, mid) maps to its last line. */
int fhx_kr_load_loci(fhx_kr* kr, const int32_t* chr, const int32_t* mid, int64_t n);
/* interactions file, HiCKRy.py:34-54: rawMatrix = coo((value,(x,y))) + its transpose, assembled as CSR in HBM.  A row whose
 * locus is not in the fragments file returns FHX_ERR_REFERENCE_EXIT (KeyError) and its row number. */
int fhx_kr_load_pairs(fhx_kr* kr, const int32_t* chr1, const int32_t* mid1, const int32_t* chr2, const int32_t* mid2,
                      const double* value, int64_t m, int64_t* first_unknown_row);
int fhx_kr_shape(const fhx_kr* kr, int64_t* n_full, int64_t* nnz_full, int64_t* n_reduced, int64_t* nnz_reduced);
/* which 0 = raw matrix, 1 = after remove_sparse; any output pointer may be NULL */
int fhx_kr_get_csr(fhx_kr* kr, int32_t which, int64_t* indptr, int32_t* col, double* val);
/* mtx.sum(axis=0), HiCKRy.py:78 */
int fhx_kr_row_sums(fhx_kr* kr, double* out);
/* removeZeroDiagonalCSR, HiCKRy.py:74-101: drops every row/column whose sum is <= the int(perc*n)-th smallest sum. */
int fhx_kr_remove_sparse(fhx_kr* kr, double perc, int64_t* n_removed, double* val_to_remove, int64_t* rem_rows);
int fhx_kr_get_removed(const fhx_kr* kr, int64_t* idx, int64_t capacity, int64_t* n_out);
/* knightRuizAlg, HiCKRy.py:139-243 (tol = 1e-6 in the reference's call) */
int fhx_kr_balance(fhx_kr* kr, double tol, fhx_kr_info* out);
int fhx_kr_get_x(const fhx_kr* kr, double* x);
/* computeBiasVector + addZeroBiases, HiCKRy.py:103-115: n_full values, -1 at the removed rows */
int fhx_kr_bias(fhx_kr* kr, double* bias);
/* test / measurement hooks: y = A x (host vectors), repeated `repeats` times with HIP-event timing; ordered dot product */
int fhx_kr_spmv(fhx_kr* kr, int32_t which, const double* x, double* y, int32_t repeats, double* seconds_per_call);
int fhx_kr_dot(fhx_kr* kr, const double* a, const double* b, int64_t n, double* out);

/* ---- merging of nearby significant contacts (fithic/utils/CombineNearbyInteraction.py; SURVEY 8f rank 4, the step after
 * Fit-Hi-C).  The caller parses the significances table, keeps the rows with chr1 == chr2 (:258-266), numbers the chromosomes
 * in the order they are to be written (byte-sorted names, :204-212) and passes n = int(float(mid) + res/2) (:301-303).
 * All numerators must share one offset modulo the resolution (true for every fixed-size Fit-Hi-C output); -p 0 is refused:
 * its result depends on CPython's set iteration order (:417-437). */
typedef struct fhx_cni fhx_cni;
typedef struct fhx_cni_info {
    int64_t rows, nodes, components, selected, pick_rounds, largest_component;
} fhx_cni_info;
typedef struct fhx_cni_record {        /* one output line (:589-600); bins are n / res */
    int32_t chr, reserved;
    int64_t n_lo, n_hi;                /* the picked cell */
    int64_t cc;
    double p, q;                       /* of the first row of that cell */
    int64_t box_min_lo, box_max_lo, box_min_hi, box_max_hi;   /* bounding box of its component (numerators) */
    int64_t sum_cc;                    /* over the component's cells */
    int64_t box_cells;                 /* cells of the chromosome inside the box (0 when bins are not whole numbers) */
    int64_t component_size;
    int64_t first_row;                 /* input row the cell's values come from */
} fhx_cni_record;
int fhx_cni_create(int device, fhx_cni** out);
void fhx_cni_destroy(fhx_cni* cn);
const char* fhx_cni_last_error(const fhx_cni* cn);
int fhx_cni_load(fhx_cni* cn, const int32_t* chr, const int64_t* n1, const int64_t* n2, const int64_t* cc, const double* p,
                 const double* q, int64_t rows, int64_t bin_size, int64_t* n_nodes);
/* connectivity 8|4 (-c), top_percent 1..100 (-p), neighborhood in bins (-n), sort_order 0|1 (-s) */
int fhx_cni_run(fhx_cni* cn, int32_t connectivity, int32_t top_percent, int32_t neighborhood, int32_t sort_order,
                fhx_cni_info* info);
int fhx_cni_get_records(const fhx_cni* cn, fhx_cni_record* out, int64_t capacity, int64_t* n_out);

#ifdef __cplusplus
}
#endif
#endif /* FITHIC_MI355X_H */
