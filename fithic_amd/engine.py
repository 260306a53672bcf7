"""Host driver of one Fit-Hi-C run on one GPU: the body of the reference's main() after argument parsing
(fithic/fithic.py:317-370) expressed over the C ABI of libfithic_mi355x.so.

    pass_stats  (K1)   <- read_Interactions                       fithic/fithic.py:389-454
    fit         (host) <- makeBinsFromInteractions, generate_FragPairs, calculateProbabilities,
                          fit_Spline's spline + table             fithic/fithic.py:463-689, 843-918, 936-968
    pvalues     (K2)   <- fit_Spline's per-pair loop + bdtrc      fithic/fithic.py:1017-1124
    bh          (K3)   <- benjamini_hochberg_correction           fithic/myStats.py:24-48
    next_pass          <- outlier collection                      fithic/fithic.py:1215-1217

There is no CPU implementation behind this class: without the built library or without a GPU it raises.
"""
import os
import time

import numpy as np

from . import _capi

MODES = {"intraOnly": _capi.MODE_INTRA_ONLY, "interOnly": _capi.MODE_INTER_ONLY, "All": _capi.MODE_ALL}
# what bdtrc is given for a total of counts at or above 2^31: "reference" = narrowed to a C int, as scipy does with the Python ints
# of fithic.py:1070 / :1101 (p = q = nan for 2^31 <= total < 2^32, a wrong small n above) - the drop-in answer and the default;
# "wide" = the true total (what the formula means; no reference computes it)
TOTALS = {"reference": _capi.TOTALS_REFERENCE, "wide": _capi.TOTALS_WIDE}


def totals_notice(info):
    """One line for stderr when a total of this pass reached 2^31 (info = fhx_fit_info as a dict), else None."""
    bits = info.get("totals_narrowed", 0)
    if not bits:
        return None
    which = " and ".join(w for b, w in ((1, "observedIntraInRangeSum"), (2, "observedInterAllSum")) if bits & b)
    if info["totals"] == _capi.TOTALS_WIDE:
        return ("fithic-mi355x: %s >= 2^31: p-values computed with the true total(s) (--totals wide). The reference (scipy's bdtrc "
                "takes a C int) would have used n = %d (intra) / %d (inter) there and written nan wherever that is below "
                "count - 1; --totals reference reproduces that." % (which, _narrow(info["bdtrc_n_intra"]), _narrow(info["bdtrc_n_inter"])))
    return ("fithic-mi355x: %s >= 2^31: scipy's bdtrc narrows n to a C int, so the reference computes these p-values with n = %d "
            "(intra) / %d (inter) - nan wherever n < count - 1 - and so does this run (--totals reference, the default: output "
            "identical to fithic.py). --totals wide uses the true totals instead." % (which, info["bdtrc_n_intra"], info["bdtrc_n_inter"]))


def _narrow(n):
    return ((int(n) + 2 ** 31) % 2 ** 32) - 2 ** 31


class PassOutput:
    """Everything one spline pass produced (host copies of the small arrays; p/q stay on the GPU until fetched)."""

    def __init__(self):
        self.stats = None       # dict of fhx_stats
        self.info = None        # dict of fhx_fit_info
        self.arrays = {}


class Engine:
    def __init__(self, device=0):
        self.ctx = _capi.Context(device)
        self.device = device
        self.n_rows = 0
        self.mode = "intraOnly"
        self.pass_no = 0
        # FHX_CALL_TIMES=1 (measurements): host seconds inside pass_stats / fit / pvalues / bh summed over run_pass calls, then the
        # number of calls - where a small pass's wall time goes that its kernels do not explain
        self.call_seconds = [0.0, 0.0, 0.0, 0.0, 0] if os.environ.get("FHX_CALL_TIMES") else None

    def close(self):
        self.ctx.close()

    def configure(self, resolution, dist_low=0, dist_up=float("inf"), n_bins=100, mapp_thres=1, mode="intraOnly",
                  bias_low=0.5, bias_up=2.0, totals="reference"):
        if mode not in MODES:
            raise ValueError("Invalid Option. Only options are 'All', 'interOnly', or 'intraOnly'")
        if totals not in TOTALS:
            raise ValueError("totals must be 'reference' or 'wide'")
        self.mode = mode
        self.resolution = int(resolution)
        self.dist_low, self.dist_up = dist_low, dist_up
        self.totals = totals
        self.ctx.set_params(resolution, dist_low, dist_up, n_bins, mapp_thres, MODES[mode], bias_low, bias_up, TOTALS[totals])

    def load_fragments(self, chr_ids, mids, hits, chr_sort_rank):
        self.ctx.load_fragments(chr_ids, mids, hits, chr_sort_rank)

    def load_bias(self, chr_ids, mids, bias):
        self.ctx.load_bias(chr_ids, mids, bias)

    def load_contacts(self, chr1, mid1, chr2, mid2, count):
        self.ctx.load_pairs(chr1, mid1, chr2, mid2, count)
        self.n_rows = len(mid1)
        self.pass_no = 0

    def commit_contacts_text(self, ids, n):
        """second half of the device-side ingest (tables.load_contacts): ids of the names ctx.ingest_contacts_text returned"""
        self.ctx.ingest_contacts_commit(ids)
        self.n_rows = int(n)
        self.pass_no = 0

    def load_contacts_device(self, ptrs, n, stream=None):
        self.ctx.load_pairs_device(ptrs, n, stream)
        self.n_rows = int(n)
        self.pass_no = 0

    # ---- the four steps of a pass; run_pass() strings them together ----
    def pass_stats(self):
        return self.ctx.pass_stats()

    def fit(self):
        return self.ctx.fit()

    def run_pass(self, collect=True):
        """K1 -> host fit -> K2 -> K3, all on this context's stream.  Returns a PassOutput."""
        out = PassOutput()
        if self.call_seconds is None:
            st, info = self.ctx.run_pass()      # the four calls below as one C call: no interpreter between K1 and K2
        else:                                   # measurements: host wall time of each of the four calls, summed
            t = [time.perf_counter()]
            st = self.ctx.pass_stats()
            t.append(time.perf_counter())
            info = self.ctx.fit()
            t.append(time.perf_counter())
            self.ctx.pvalues()
            t.append(time.perf_counter())
            self.ctx.bh(info.bh_total_tests)
            t.append(time.perf_counter())
            for k in range(4):
                self.call_seconds[k] += t[k + 1] - t[k]
            self.call_seconds[4] += 1
        out.stats, out.info = st.as_dict(), info.as_dict()
        if collect:
            self.ctx.sync()
            A = _capi
            names = dict(hist_sumcc=A.A_HIST_SUMCC, hist_npairs=A.A_HIST_NPAIRS, bin_lb=A.A_BIN_LB, bin_ub=A.A_BIN_UB,
                         bin_poss=A.A_BIN_POSS, bin_poss0=A.A_BIN_POSS0, bin_sumcc=A.A_BIN_SUMCC, bin_sumdist=A.A_BIN_SUMDIST,
                         bin_poss7=A.A_BIN_POSS7, x=A.A_X, y=A.A_Y)
            dist_keys = self.ctx.get_array(A.A_DIST_KEYS)
            if self.resolution == 0 or len(dist_keys):      # -r 0, or -r N on loci off the grid
                out.arrays["dist_keys"] = dist_keys
            if self.mode != "interOnly":
                names.update(knots=A.A_KNOTS, coeffs=A.A_COEFFS, table_x=A.A_TABLE_X, table_y0=A.A_TABLE_Y0, table_y=A.A_TABLE_Y)
            for k, w in names.items():
                out.arrays[k] = self.ctx.get_array(w)
        self.pass_no += 1
        return out

    def fetch(self, p=True, q=True, expcc=False, bias=False):
        return self.ctx.fetch(self.n_rows, p=p, q=q, expcc=expcc, bias=bias)

    def fdr_counts(self):
        return self.ctx.get_array(_capi.A_FDR_COUNTS)

    def next_pass(self):
        """Fold this pass's outliers (p < 1/N) into the skip mask and the outlier-distance multiset."""
        return self.ctx.next_pass()

    def reset_passes(self):
        """Start over at pass 1 on the loaded tables (the reference: a fresh main() on the same files)."""
        self.ctx.reset_passes()
        self.pass_no = 0

    def kernel_seconds(self):
        return self.ctx.kernel_seconds()
