"""The reference's hot-path functions, same names / arguments / return values, running on the MI355X engine.

Drop-in surface.  Like the reference, the functions communicate through module globals that
main() assigns (fithic/fithic.py:203-308): distLowThres, distUpThres, mappThres, interOnly, allReg,
biasLowerBound, biasUpperBound, logfile, visual - plus one global the reference does not need because it re-reads
its files in every stage: `resolution` (the engine needs the fixed-size grid before the contacts go to the GPU).

    read_Interactions(contactCountsFile, biasFile, outliers=None)              fithic/fithic.py:389
    makeBinsFromInteractions(mainDic, noOfBins, observedIntraInRangeSum, outliersdist=None)        :463
    generate_FragPairs(observedInterAllCount, observedInterAllSum, binStats, fragsfile, resolution) :561
    read_biases(infilename)                                                                         :798
    calculateProbabilities(mainDic, binStats, resolution, outfilename, observedIntraInRangeSum)     :843
    fit_Spline(mainDic, x, y, yerr, infilename, outfilename, biasDic, outliersline, outliersdist, ...)  :925
    benjamini_hochberg_correction(p_values, num_total_tests)                   fithic/myStats.py:24

The per-pair and per-distance work runs in the HIP kernels behind the C ABI (no CPU implementation exists here);
these wrappers only move tables in, shape results like the reference's Python objects, and write the two output
files with the reference's exact formats.
"""
import gzip
import os
import sys
import time

import numpy as np

from . import _capi, tables
from .engine import Engine, MODES, totals_notice

# ---- the reference's module globals (defaults as in fithic/fithic.py:203-304) ------------------------
mappThres = 1
distLowThres = 0
distUpThres = float("inf")
visual = False
interOnly = False
allReg = False
biasLowerBound = 0.5
biasUpperBound = 2
interChrProb = 0
baselineIntraChrProb = 0
distScaling = 1000000.0
toKb, toMb, toProb = 10 ** -3, 10 ** -6, 10 ** 5
logfile = None
resolution = None          # engine-only global, see the module docstring
device = 0                 # GPU ordinal the engine uses
gpus = 1                   # > 1: contact rows sharded over that many GPUs (fithic_amd.sharded)
# What bdtrc is given when observedIntraInRangeSum / observedInterAllSum reach 2^31 (:1070, :1101).  "reference": narrowed to a C
# int as scipy does with the reference's Python ints, so the files equal fithic.py's (nan p and q for 2^31 <= total < 2^32);
# "wide": the true total.  A line on stderr says which ran whenever it matters (engine.totals_notice).
totals = "reference"


class _Session:
    """What the reference keeps implicitly on disk and in globals between its stages."""

    def __init__(self):
        self.engine = None
        self.chroms = tables.ChromIndex()
        self.contacts = None
        self.contacts_path = None
        self.frags_path = None
        self.bias_path = None
        self.bias_loaded = False
        self.n_bins = 100
        self.stats = None
        self.info = None
        self.fit_done = False
        self.arrays = {}
        self.values = None
        self.pass_started = 0
        self.main_dic = None          # the dict read_Interactions handed out, and what it held then (_main_dic_sig)
        self.main_dic_sig = None
        self.outlier_refs = None      # (outliersline, len, outliersdist, len) as fit_Spline left them: the engine's own lists

    def mode(self):
        return "All" if allReg else ("interOnly" if interOnly else "intraOnly")

    def ensure_engine(self):
        if self.engine is None:
            if gpus > 1:
                from .sharded import ShardedEngine
                self.engine = ShardedEngine(gpus)
            else:
                self.engine = Engine(device)
        return self.engine

    def check_resolution(self):
        if resolution is None or resolution < 0:
            raise RuntimeError("fithic_amd.fithic.resolution must be set before read_Interactions (the -r value; 0 = "
                               "non-fixed-size data): the engine builds its locus grid when the contacts go to the GPU")

    def configure(self):
        self.check_resolution()
        self.ensure_engine().configure(resolution, distLowThres, distUpThres, self.n_bins, mappThres, self.mode(),
                                       biasLowerBound, biasUpperBound, totals)

    def ensure_fit(self):
        if not self.fit_done:
            self.configure()
            info = self.engine.fit()
            self.info = info.as_dict()
            notice = totals_notice(self.info)
            if notice:
                print(notice, file=sys.stderr)
            A = _capi
            for k, w in dict(bin_lb=A.A_BIN_LB, bin_ub=A.A_BIN_UB, bin_poss=A.A_BIN_POSS, bin_poss0=A.A_BIN_POSS0,
                             bin_sumcc=A.A_BIN_SUMCC, bin_sumdist=A.A_BIN_SUMDIST, bin_poss7=A.A_BIN_POSS7, x=A.A_X, y=A.A_Y).items():
                self.arrays[k] = self.engine.ctx.get_array(w)
            if not interOnly:
                for k, w in dict(table_x=A.A_TABLE_X, table_y=A.A_TABLE_Y, knots=A.A_KNOTS, coeffs=A.A_COEFFS).items():
                    self.arrays[k] = self.engine.ctx.get_array(w)
            self.fit_done = True
        return self.info


_S = _Session()


def session_stuck():
    """True when a sharded engine was abandoned with this process's own rank still inside a collective (sharded._all)."""
    return bool(getattr(_S.engine, "local_stuck", False))


def reset_session():
    """Forget the loaded tables (a new run in the same interpreter)."""
    global _S
    if _S.engine is not None:
        try:
            _S.engine.close()
        except Exception:                      # noqa: BLE001 - a session that failed must not block the next one
            pass
    _S = _Session()


def _log(text, mode="a"):
    if logfile:
        with open(logfile, mode) as f:
            f.write(text)


# ======================================================================================================
def read_Interactions(contactCountsFile, biasFile, outliers=None):
    """fithic/fithic.py:389-454.  Returns (mainDic, observedInterAllCount, observedInterAllSum,
    observedIntraAllSum, observedIntraInRangeSum); mainDic = {distance: [0, sum of counts]}."""
    print("Reading the contact counts file to generate bins...")
    t0 = time.time()
    S = _S
    if S.contacts is None or S.contacts_path != contactCountsFile:
        S.check_resolution()
        # the HIP runtime and the context come up (a few tenths of a second) while the host cores inflate the file; the GPU
        # then parses the text (tables.load_contacts; the host parser takes the files the device grammar does not cover)
        import threading
        starter = threading.Thread(target=S.ensure_engine)
        starter.start()

        def engine_of():
            starter.join()
            S.configure()
            return S.engine
        engine_of.sharded = gpus > 1                       # (known before the engine is up: load_contacts plans by it)
        try:
            S.contacts = tables.load_contacts(contactCountsFile, S.chroms, engine_of)
        finally:
            starter.join()
        S.contacts_path = contactCountsFile
        if os.environ.get("FHX_TIMING"):
            print("stage: inflate + parse + ingest of %d rows took %.3f s (%s parser)"
                  % (len(S.contacts), time.time() - t0, "device" if hasattr(S.contacts, "rows") else "host"))
        S.pass_started = 0
    elif outliers is not None and S.pass_started >= 1 and S.values is not None:
        if S.outlier_refs is not None and not (outliers is S.outlier_refs[0] and len(outliers) == S.outlier_refs[1]):
            raise ValueError("fithic_amd: read_Interactions skips the lines the ENGINE flagged in the previous fit_Spline; an outlier "
                             "list that was edited or built elsewhere cannot be honoured (pass the list fit_Spline returned)")
        S.engine.next_pass()                 # fold the previous pass's outliers into the skip mask (K1 skips them)
    S.configure()
    st = S.engine.pass_stats()
    S.stats = st.as_dict()
    S.fit_done = False
    S.pass_started += 1
    keys, hist_cc = _dist_keys(S)
    mainDic = {int(k): [0, int(c)] for k, c in zip(keys, hist_cc)}
    S.main_dic, S.main_dic_sig = mainDic, _main_dic_sig(mainDic)
    print("Interactions file read. Time took %s" % (time.time() - t0))
    lo = int(keys[0]) if len(keys) else float("inf")
    hi = int(keys[-1]) if len(keys) else 0
    _log("\n\nInteractions file read successfully\n"
         "------------------------------------------------------------------------------------\n"
         "Observed, Intra-chr in range: pairs= %s\t totalCount= %s\n"
         "Observed, Intra-chr all: pairs= %s\t totalCount= %s\n"
         "Observed, Inter-chr all: pairs= %s\t totalCount= %s\n"
         "Range of observed genomic distances [%s %s]\n\n"
         % (st.in_range_count, st.in_range_sum, st.intra_all_count, st.intra_all_sum, st.inter_count, st.inter_sum, lo, hi), "w")
    return (mainDic, st.inter_count, st.inter_sum, st.intra_all_sum, st.in_range_sum)


def _main_dic_sig(mainDic):
    """what identifies the contents of a mainDic for the stages below: number of distances, their sum, the summed counts"""
    return (len(mainDic), sum(mainDic), sum(v[1] for v in mainDic.values()))


def _adopt_main_dic(S, mainDic, observedIntraInRangeSum):
    """The reference bins whatever mainDic it is handed (fithic/fithic.py:463-553).  The engine bins the histogram K1 left on
    the host; when the caller passes the dict read_Interactions returned, untouched, that IS this histogram and nothing is
    done.  A dict the caller built or edited (distances removed, counts changed) replaces the engine's histogram through
    fhx_set_global_stats, so the bins, the fit and the p-values follow the argument like the reference's do."""
    if mainDic is S.main_dic and _main_dic_sig(mainDic) == S.main_dic_sig and observedIntraInRangeSum == S.stats["in_range_sum"]:
        return
    if S.engine is None or S.stats is None:
        raise ValueError("fithic_amd: makeBinsFromInteractions needs read_Interactions first (the contact rows live on the GPU)")
    ctx = S.engine.ctx
    if not hasattr(ctx, "set_global_stats"):
        raise ValueError("fithic_amd: a mainDic other than the one read_Interactions returned is not supported with gpus > 1")
    keys = np.array(sorted(mainDic), np.int64)
    sums = np.array([mainDic[int(k)][1] for k in keys], np.int64)
    st = _capi.FhxStats()
    for name, _ in st._fields_:
        setattr(st, name, S.stats[name])
    st.in_range_sum = int(observedIntraInRangeSum)
    if resolution == 0 or len(ctx.get_array(_capi.A_DIST_KEYS)):           # explicit distances: -r 0 / loci off the grid
        ctx.set_dist_keys(keys)
        ctx.set_global_stats(st, sums, np.ones(len(keys), np.int64))
    else:
        if len(keys) and (keys % resolution).any():
            raise ValueError("fithic_amd: mainDic holds distances that are not multiples of the resolution %d" % resolution)
        idx = keys // resolution
        n = max(int(S.stats["n_dist"]), int(idx.max()) + 1 if len(idx) else 1)
        hist_cc, hist_np = np.zeros(n, np.int64), np.zeros(n, np.int64)
        hist_cc[idx] = sums
        hist_np[idx] = 1
        ctx.set_global_stats(st, hist_cc, hist_np)
    S.stats = dict(S.stats, in_range_sum=int(observedIntraInRangeSum))
    S.main_dic, S.main_dic_sig = mainDic, _main_dic_sig(mainDic)
    S.fit_done = False


def _adopt_outlier_dists(S, outliersdist):
    """makeBinsFromInteractions subtracts one possible pair per outlier distance of the earlier passes (fithic/fithic.py:533-551).
    The engine keeps that multiset itself; the list fit_Spline extended and returned, passed back untouched, is that multiset.
    Any other list replaces it for this call (fhx_set_outlier_dists / fhx_set_outlier_dist_hist), like the argument does in
    the reference."""
    refs = S.outlier_refs
    if outliersdist is None or (refs is not None and outliersdist is refs[2] and len(outliersdist) == refs[3]):
        return
    ctx = S.engine.ctx
    if not hasattr(ctx, "set_outlier_dist_hist"):
        raise ValueError("fithic_amd: an outliersdist list other than the one fit_Spline returned is not supported with gpus > 1")
    dists = np.sort(np.asarray(list(outliersdist), np.int64))
    if resolution == 0 or len(ctx.get_array(_capi.A_DIST_KEYS)):
        ctx.set_outlier_dists(dists)
        return
    if len(dists) and ((dists % resolution).any() or dists.min() < 0):
        raise ValueError("fithic_amd: outliersdist holds distances that are not multiples of the resolution %d" % resolution)
    n = int(S.stats["n_dist"])
    # distances beyond the histogram fall behind the last bin, where the reference's cursor stops (fithic.py:541-547): slot n - 1
    ctx.set_outlier_dist_hist(np.bincount(np.minimum(dists // resolution, n - 1), minlength=n).astype(np.int64))


def _check_session_values(what, given, own):
    """calculateProbabilities / fit_Spline take x, y and binStats as arguments in the reference; here they are views of the
    engine's fit.  Values the caller changed in between would be ignored silently - refuse instead."""
    def same(a, b):                            # NaN is how the host fit marks a value Python would have raised on: untouched, it is equal
        a, b = float(a), float(b)
        return a == b or (a != a and b != b)
    if len(given) != len(own) or not all(same(a, b) for a, b in zip(given, own)):
        raise ValueError("fithic_amd: %s differs from what the engine computed in the previous stage; the engine fits its own "
                         "bins (edit mainDic before makeBinsFromInteractions instead)" % what)


def _dist_keys(S):
    """(distinct in-range distances ascending, their summed counts) = the reference's mainDic"""
    hist_cc = S.engine.ctx.get_array(_capi.A_HIST_SUMCC)
    hist_np = S.engine.ctx.get_array(_capi.A_HIST_NPAIRS)
    keys = S.engine.ctx.get_array(_capi.A_DIST_KEYS)
    if resolution == 0 or len(keys):               # -r 0, or -r N on loci off the grid: explicit distinct distances
        return keys, hist_cc[:len(keys)]
    idx = np.flatnonzero(hist_np > 0)
    return idx * resolution, hist_cc[idx]


def _bin_stats(S, poss_key):
    a = S.arrays
    keys = _dist_keys(S)[0]
    out = {}
    for b in range(len(a["bin_lb"])):
        lb, ub = int(a["bin_lb"][b]), int(a["bin_ub"][b])
        first = 0 if b == 0 else lb           # bin 0 starts at 0 but its first distance is the first key
        dists = [int(d) for d in keys[(keys >= first) & (keys <= ub)]]
        out[b] = [(lb, ub), int(a[poss_key][b]), int(a["bin_sumcc"][b]), 0, 0, 0, dists, int(a[poss_key][b])]
    return out


def makeBinsFromInteractions(mainDic, noOfBins, observedIntraInRangeSum, outliersdist=None):
    """fithic/fithic.py:463-553.  Returns binStats = {bin: [(lb, ub), possPairs, sumCC, sumDist, avgCC, avgDist, [distances], possPairs]}."""
    S = _S
    S.n_bins = noOfBins
    noPerBin = observedIntraInRangeSum / noOfBins
    _log("Making equal occupancy bins\n"
         "------------------------------------------------------------------------------------\n"
         "Observed intra-chr read counts in range\t%r\nDesired number of contacts per bin\t%r,\nNumber of bins\t%r\n"
         % (observedIntraInRangeSum, noPerBin, noOfBins))
    S.fit_done = False
    S.configure()
    _adopt_main_dic(S, mainDic, observedIntraInRangeSum)
    _adopt_outlier_dists(S, outliersdist)
    S.engine.ctx.make_bins()
    A = _capi
    for k, w in dict(bin_lb=A.A_BIN_LB, bin_ub=A.A_BIN_UB, bin_poss0=A.A_BIN_POSS0, bin_sumcc=A.A_BIN_SUMCC).items():
        S.arrays[k] = S.engine.ctx.get_array(w)
    _log("Equal occupancy bins generated\n\n")
    return _bin_stats(S, "bin_poss0")


def _load_fragments(S, fragsfile):
    if S.frags_path != fragsfile:
        fc, fm, fh = tables.read_fragments(fragsfile, S.chroms)
        S.frag_table = (fc, fm, fh)
        S.frags_path = fragsfile
    S.configure()
    fc, fm, fh = S.frag_table
    S.engine.load_fragments(fc, fm, fh, S.chroms.sort_rank())


def generate_FragPairs(observedInterAllCount, observedInterAllSum, binStats, fragsfile, resolution):
    """fithic/fithic.py:561-793 (fixed-size branch).  Returns (binStats, noOfFrags, maxPossibleGenomicDist,
    possibleIntraInRangeCount, possibleInterAllCount, interChrProb, baselineIntraChrProb)."""
    global interChrProb, baselineIntraChrProb
    S = _S
    t0 = time.time()
    _log(("Looping through all possible fragment pairs in-range\n" if resolution else "Enumerating all possible fragment pairs in-range\n")
         + "------------------------------------------------------------------------------------\n")
    _load_fragments(S, fragsfile)
    info = S.ensure_fit()
    full = _bin_stats(S, "bin_poss")
    for b, rec in full.items():
        rec[3] = float(S.arrays["bin_sumdist"][b])
        rec[7] = int(S.arrays["bin_poss7"][b])
    binStats.clear()
    binStats.update(full)
    fc, fm, fh = S.frag_table
    names = S.chroms.names
    n_frags = info["n_frags"]
    min_poss = float("inf")
    for c in sorted(set(fc.tolist()), key=lambda i: names[i]):
        sel = (fc == c) & (fh >= mappThres)
        n = int(sel.sum())
        if n == 0:
            continue
        if resolution:
            stop = int(fm[sel].max() - resolution / 2 + 1)
            d = np.arange(0, max(stop, 0), resolution)
            npairs = n - np.arange(len(d))
            rng = (d >= distLowThres) & (d <= distUpThres)
            per = int(npairs[rng].sum()) * (2 if len(full) else 1)
            if rng.any():
                min_poss = min(min_poss, int(d[rng][0]))
        else:
            m = np.sort(fm[sel].astype(np.int64))
            hi_edge = np.searchsorted(m, m + (distUpThres if distUpThres != float("inf") else m[-1]), side="right")
            lo_edge = np.searchsorted(m, m + distLowThres, side="left")
            lo_edge = np.maximum(lo_edge, np.arange(n) + 1)
            per = int(np.maximum(hi_edge - lo_edge, 0).sum())
            ok = hi_edge > lo_edge
            if ok.any():
                min_poss = min(min_poss, float((m[np.minimum(lo_edge, n - 1)] - m)[ok].min()))
        _log("Chromosome %r,\t%s mappable fragments, \t%s possible intra-chr fragment pairs in range,\t%s possible inter-chr "
             "fragment pairs\n" % (names[c], n, per, (n_frags - n) * n))
    print("Fragments file read. Time took %s" % (time.time() - t0))
    interChrProb = info["inter_chr_prob"] if info["inter_chr_prob"] else 0
    baselineIntraChrProb = info["baseline_intra_prob"]
    _log("Number of all fragments= %s\nPossible, Intra-chr in range: pairs= %s \nPossible, Intra-chr all: pairs= %s \n"
         "Possible, Inter-chr all: pairs= %s \n" % (n_frags, info["possible_intra_in_range"],
                                                    info["possible_intra_all"] if resolution else int(info["possible_intra_all"]),
                                                    info["possible_inter_all"]))
    _log("Desired genomic distance range   [%d %s] \n" % (distLowThres, distUpThres))
    try:
        _log("Range of possible genomic distances  [%d  %d] \n" % (min_poss, info["max_possible_dist"]))
    except (OverflowError, ValueError):
        pass
    _log("Baseline intrachromosomal probability is %s \nInterchromosomal probability is %s \n" % (baselineIntraChrProb, interChrProb))
    return (binStats, n_frags, info["max_possible_dist"], info["possible_intra_in_range"], info["possible_inter_all"],
            interChrProb, baselineIntraChrProb)


class _BiasDic(dict):
    """{chrom: {mid: bias}} like the reference's nested dict (fithic.py:798-837), filled when somebody looks: the engine reads the
    biases from its own table, and building 6e5 Python entries costs the command line 0.07 s.  Truthiness, len(), iteration,
    indexing, membership, get/keys/values/items, equality and repr all see the filled dict."""

    def __init__(self, fill=None, n_loci=0):
        super().__init__()
        self._fill_from, self._n_loci = fill, n_loci

    def _fill(self):
        fill, self._fill_from = self._fill_from, None
        if fill is not None:
            names, bc, bm, vals = fill
            for c, m, v in zip(bc.tolist(), bm.tolist(), vals.tolist()):
                d = dict.setdefault(self, names[c], {})
                if m not in d:
                    d[m] = -1 if v == -1.0 else v
        return self

    def __bool__(self):
        return self._n_loci > 0 if self._fill_from is not None else dict.__len__(self) > 0

    def __len__(self):
        return dict.__len__(self._fill())

    def __iter__(self):
        return dict.__iter__(self._fill())

    def __contains__(self, key):
        return dict.__contains__(self._fill(), key)

    def __getitem__(self, key):
        return dict.__getitem__(self._fill(), key)

    def __eq__(self, other):
        return dict.__eq__(self._fill(), other)

    def __ne__(self, other):
        return dict.__ne__(self._fill(), other)

    __hash__ = None

    def __repr__(self):
        return dict.__repr__(self._fill())

    def get(self, key, default=None):
        return dict.get(self._fill(), key, default)

    def keys(self):
        return dict.keys(self._fill())

    def values(self):
        return dict.values(self._fill())

    def items(self):
        return dict.items(self._fill())

    def setdefault(self, key, default=None):
        return dict.setdefault(self._fill(), key, default)

    def copy(self):
        return dict(self.items())


def read_biases(infilename):
    """fithic/fithic.py:798-837.  Loads the bias table into the engine; returns {chrom: {mid: bias}} like the reference."""
    t0 = time.time()
    S = _S
    bc, bm, bv = tables.read_bias(infilename, S.chroms)
    S.configure()
    S.engine.load_bias(bc, bm, bv)
    S.bias_path, S.bias_loaded = infilename, True
    bot, med, top = tables.bias_quantiles(bv)
    _log("5th quantile of biases: %s\n50th quantile of biases: %s\n95th quantile of biases: %s\n" % (bot, med, top))
    vals = np.where((bv < biasLowerBound) | np.isnan(bv) | (bv > biasUpperBound), -1.0, bv)
    discard = int(((bv < biasLowerBound) | np.isnan(bv) | (bv > biasUpperBound)).sum())
    out = _BiasDic((list(S.chroms.names), bc, bm, vals), len(bv))
    _log("Out of %s loci %s were discarded with biases not in range [%s-%s]\n\n" % (len(bv), discard, biasLowerBound, biasUpperBound))
    print("Bias file read. Time took %s" % (time.time() - t0))
    return out


def calculateProbabilities(mainDic, binStats, resolution, outfilename, observedIntraInRangeSum):
    """fithic/fithic.py:843-918.  Returns [x, y, yerr] and writes <outfilename>.res<R>.txt."""
    S = _S
    _log("\nCalculating probability means and standard deviations of contact counts\n"
         "------------------------------------------------------------------------------------\n")
    S.ensure_fit()
    _check_session_values("binStats (bin bounds / summed counts)",
                          [v for b in sorted(binStats) for v in (binStats[b][0][0], binStats[b][0][1], binStats[b][2])],
                          [v for b in range(len(S.arrays["bin_lb"])) for v in (S.arrays["bin_lb"][b], S.arrays["bin_ub"][b], S.arrays["bin_sumcc"][b])])
    x = [float(v) for v in S.arrays["x"]]
    y = [float(v) for v in S.arrays["y"]]
    yerr = [0] * len(x)
    name = outfilename + (".res" + str(resolution) if resolution else "") + ".txt"
    print("Writing %s" % name)
    with open(name, "w") as f:
        f.write("avgGenomicDist\tcontactProbability\tstandardError\tnoOfLocusPairs\ttotalOfContactCounts\n")
        for i in range(len(x)):
            f.write("%d" % x[i] + "\t" + "%.2e" % y[i] + "\t" + "%.2e" % yerr[i] + "\t" + "%d" % S.arrays["bin_poss"][i] + "\t"
                    + "%d" % S.arrays["bin_sumcc"][i] + "\n")
            if i in binStats:
                binStats[i][4], binStats[i][5] = y[i], x[i]
    _log("Means and error written to %s\n\n" % name)
    return [x, y, yerr]


def fit_Spline(mainDic, x, y, yerr, infilename, outfilename, biasDic, outliersline, outliersdist, observedIntraInRangeSum,
               possibleIntraInRangeCount, possibleInterAllCount, observedInterAllCount, observedIntraAllSum,
               observedInterAllSum, biasLowerBound, biasUpperBound, resolution, passNo):
    """fithic/fithic.py:925-1233.  Returns [splineX, newSplineY, residual, outliersline, outliersdist, FDRx, FDRy] and
    writes <outfilename>.res<R>.significances.txt.gz."""
    S = _S
    _log("\nFitting a univariate spline to the probability means\n"
         "------------------------------------------------------------------------------------\n")
    info = S.ensure_fit()
    _check_session_values("x (avgGenomicDist)", x, S.arrays["x"])
    _check_session_values("y (contactProbability)", y, S.arrays["y"])
    splineX = newSplineY = residual = None
    if not interOnly:
        splineX = [int(v) for v in S.arrays["table_x"]]
        newSplineY = np.array(S.arrays["table_y"])
        residual = info["residual"]
        if visual:
            from . import plots
            plots.plot_spline_fit(outfilename, passNo, x, y, yerr, splineX, newSplineY, distLowThres, distUpThres)
    eng = S.engine
    t_stage = time.time()
    eng.ctx.pvalues()                                   # K2
    eng.ctx.bh(info["bh_total_tests"])                  # K3
    print("Outlier threshold is... %s" % (info["outlier_thres"]))
    con = S.contacts
    if os.environ.get("FHX_TIMING"):
        getattr(eng.ctx, "kernel_seconds", lambda: None)()      # synchronises
        print("stage: K2 + K3 took %.3f s" % (time.time() - t_stage))
        t_stage = time.time()
    name = outfilename + (".res" + str(resolution) if resolution else "") + ".significances.txt.gz"
    print("Writing p-values and q-values to file %s" % (outfilename + ".significances.txt"))
    mode_id = MODES["All" if allReg else ("interOnly" if interOnly else "intraOnly")]
    # The GPU formats and deflates the rows from the resident p and q (fhx_write_significances_device); the host writer takes
    # over for a sharded run (each rank holds a part of the rows), for rows the device formatter does not cover, or on request.
    on_device = hasattr(eng.ctx, "write_significances_device") and not os.environ.get("FHX_HOST_WRITER")
    if on_device:
        try:
            eng.ctx.write_significances_device(name, S.chroms.names)      # identity columns: the rows resident since ingest
            S.values = True
        except _capi.FhxError as e:
            if e.code != _capi.FHX_ERR_UNSUPPORTED:
                raise
            on_device = False
    if not on_device:
        v = eng.fetch(p=True, q=True, expcc=True, bias=True)
        S.values = v
        _capi.host_write_significances(name, S.chroms.names, con.chr1, con.mid1, con.chr2, con.mid2, con.count, v["p"], v["q"],
                                       v["b1"], v["b2"], v["expcc"], mode_id, distLowThres, distUpThres)
    if os.environ.get("FHX_TIMING"):
        print("stage: format + deflate + write took %.3f s" % (time.time() - t_stage))
    t_stage = time.time()
    if hasattr(eng.ctx, "fetch_outlier_rows"):
        rows = eng.ctx.fetch_outlier_rows()                  # compacted on the device
    else:                                                    # a sharded run: flags of all ranks in file order
        flags, _ = eng.ctx.fetch_flags(len(con), outlier=True, skip=False)
        rows = np.flatnonzero(flags)
    (outliersline.update if hasattr(outliersline, "add") else outliersline.extend)(rows.tolist())
    # abs(mid1 - mid2) of every outlier line, inter-chromosomal ones included (fithic.py:1217)
    if hasattr(con, "rows"):                              # rows resident on the GPU(s): DeviceContacts, sharded.ShardedContacts
        _, out_mid1, _, out_mid2, _ = con.rows(rows)
    else:
        out_mid1, out_mid2 = con.mid1[rows], con.mid2[rows]
    dists = np.abs(out_mid1.astype(np.int64) - out_mid2.astype(np.int64)).tolist()
    (outliersdist.update if hasattr(outliersdist, "add") else outliersdist.extend)(dists)
    if not hasattr(outliersline, "add"):
        outliersline.sort()
        outliersdist.sort()
    if os.environ.get("FHX_TIMING"):
        print("stage: %d outlier lines collected in %.3f s" % (len(rows), time.time() - t_stage))
    S.outlier_refs = (outliersline, len(outliersline), outliersdist, len(outliersdist))
    FDRx = np.arange(0.0, 0.05 + 0.001, 0.001)
    FDRy = [int(c) for c in eng.fdr_counts()]                # plot_qvalues' counts (fithic.py:1235-1254), device histogram
    if visual:
        from . import plots
        print("Plotting q-values to file %s" % outfilename + ".qplot.png")
        plots.plot_qvalues(FDRx, FDRy, outfilename + ".qplot")
    _log("Spline successfully fit\n\n\n")
    return [splineX, newSplineY, residual, outliersline, outliersdist, FDRx, FDRy]


def benjamini_hochberg_correction(p_values, num_total_tests):
    """fithic/myStats.py:24-48 on the GPU (sort + scan kernels); returns a Python list like the reference."""
    eng = _S.ensure_engine()
    return eng.ctx.bh_array(np.asarray(p_values, np.float64), float(num_total_tests)).tolist()
