#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ by running the REAL reference.

This script only runs in the build container: it imports the reference package from
/root/reference/fithic (read-only), runs its unmodified `main()` and library calls
(scipy.special.bdtrc, scipy.interpolate.UnivariateSpline, sklearn IsotonicRegression,
myStats.benjamini_hochberg_correction) and records inputs + outputs as DATA.  No reference
source text is stored; the reference never travels to the GPU box - these fixtures do.

Fixture sets:
  F1  bundled hESC chr1 40 kb (the reference's own tests/data files, copied as data):
        f1_nobias   -r 40000 -L 50000 -U 5000000 -b 50 -p 1 -x All            (run_tests-git.sh:34-36)
        f1_bias     ... -t bias -x intraOnly -p 2                               (run_tests-git.sh:40-42, +pass 2)
  F2  synthetic contacts over the bundled IMR90 1 Mb fragments/bias, 24 chromosomes
        f2_all (All, -p 2, bias) / f2_inter (interOnly, bias) / f2_intra (intraOnly, -p 2, bias, -L/-U)
        f2_all_nobias (All, no bias, -L/-U)
  F3  scipy.special.bdtrc / betaln / log known-answer vectors (bit patterns)
  F4  UnivariateSpline known-answer vectors (t, c, fp, ier; incl. nest restarts, s=0)
  F5  myStats.benjamini_hochberg_correction known-answer vectors
  F6  quirk probes (small hand-made inputs through the full reference main())
  F7  fixed-size 14-chromosome set; F8 non-fixed-size (-r 0) sets
  F9  utils/HiCKRy.py (Knight-Ruiz bias vectors): removed rows, KR vector, bias file, iteration counts
  F10 utils/CombineNearbyInteraction.py: merged loop lists (output files) in the modes its flags offer

Usage:  python tests/golden/make_golden.py [f1] [f2] [f3] [f4] [f5] [f6]     (default: all)
  F15 totals at and above 2^31: bdtrc known answers with n beyond a C int, and whole runs whose in-range / inter totals wrap
  F13 more than two passes on the F1 / F2 / F6 data (-p 3, -p 4): the outlier lists live for the whole run, duplicates included
"""
import sys
import os
import io
import gzip
import json
import copy
import time
import math
import hashlib
import shutil
import tempfile
import contextlib

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
DATA = os.path.join(HERE, "data")
REF_PKG = "/root/reference/fithic"

sys.path.insert(0, REF_PKG)
import fithic as F          # noqa: E402  (the reference module itself)
import myStats as REF_STATS  # noqa: E402
import scipy.special as scsp  # noqa: E402
from scipy.interpolate import UnivariateSpline  # noqa: E402

_STAGES = ("read_Interactions", "makeBinsFromInteractions", "generate_FragPairs",
           "calculateProbabilities", "fit_Spline")


def _bins_snapshot(binStats):
    """binStats: {idx: [(lb,ub), poss, sumCC, sumDist, avgCC, avgDist, [dists], poss7]} -> arrays"""
    n = len(binStats)
    out = dict(lb=np.zeros(n, np.int64), ub=np.zeros(n, np.int64), s1=np.zeros(n, np.int64),
               s2=np.zeros(n, np.int64), s3=np.zeros(n, np.float64), s7=np.zeros(n, np.int64),
               ndist=np.zeros(n, np.int64))
    for i in range(n):
        b = binStats[i]
        out["lb"][i], out["ub"][i] = b[0]
        out["s1"][i] = b[1]
        out["s2"][i] = b[2]
        out["s3"][i] = b[3]
        out["s7"][i] = b[7]
        out["ndist"][i] = len(b[6])
    return out


def run_reference(argv, quiet=True, observed=None):
    """Run the reference's main() on argv; capture every stage's outputs through a profile hook.
    observed = (mainDic, interCount, interSum, intraAllSum, inRangeSum): what read_Interactions would return for a contacts file
    too large to push through the reference's line loop (f14: 10^8..10^9 rows).  main() then runs unmodified on it - argument
    handling, makeBinsFromInteractions, generate_FragPairs, calculateProbabilities, fit_Spline - with `-i` naming a few-line
    contacts file for fit_Spline's own loop."""
    passes = []          # one dict per pass
    cur = {}
    ref_file = os.path.join(REF_PKG, "fithic.py")
    real_read = F.read_Interactions

    def given_observations(contactCountsFile, biasFile, outliers=None):
        nonlocal cur
        assert outliers is None, "one pass only"
        mainDic, interCnt, interSum, intraAllSum, inRangeSum = observed
        cur = {}
        passes.append(cur)
        keys = np.array(sorted(mainDic.keys()), np.int64)
        cur["dist_keys"] = keys
        cur["dist_sumcc"] = np.array([mainDic[int(k)][1] for k in keys], np.int64)
        cur["sums"] = np.array([interCnt, interSum, intraAllSum, inRangeSum], np.int64)
        open(F.logfile, "w").close()            # read_Interactions (re)creates the log the later stages append to
        return ({int(k): [0, int(v[1])] for k, v in mainDic.items()}, interCnt, interSum, intraAllSum, inRangeSum)

    def hook(frame, event, arg):
        if event != "return":
            return
        code = frame.f_code
        name = code.co_name
        if name not in _STAGES or code.co_filename != ref_file:
            return
        nonlocal cur
        if name == "read_Interactions":
            cur = {}
            passes.append(cur)
            mainDic, interCnt, interSum, intraAllSum, inRangeSum = arg
            keys = np.array(sorted(mainDic.keys()), np.int64)
            cur["dist_keys"] = keys
            cur["dist_sumcc"] = np.array([mainDic[int(k)][1] for k in keys], np.int64)
            cur["sums"] = np.array([interCnt, interSum, intraAllSum, inRangeSum], np.int64)
        elif name == "makeBinsFromInteractions":
            for k, v in _bins_snapshot(arg).items():
                cur["bins0_" + k] = v
        elif name == "generate_FragPairs":
            (binStats, noOfFrags, maxPossDist, possIntraInRange, possInterAll, interChrProb, baseProb) = arg
            for k, v in _bins_snapshot(binStats).items():
                cur["bins1_" + k] = v
            cur["frag_scalars"] = np.array([noOfFrags, maxPossDist, possIntraInRange, possInterAll,
                                            interChrProb, baseProb], np.float64)
            cur["possibleIntraInRangeCount"] = np.array([int(possIntraInRange)], np.int64)
        elif name == "calculateProbabilities":
            cur["x"] = np.array(arg[0], np.float64)
            cur["y"] = np.array(arg[1], np.float64)
        elif name == "fit_Spline":
            loc = frame.f_locals
            cur["p"] = np.array(loc["p_vals"], np.float64)
            cur["q"] = np.array(loc["q_vals"], np.float64)
            cur["expcc"] = np.array(loc["expCC_List"], np.float64)
            cur["b1"] = np.array(loc["biasl"], np.float64)
            cur["b2"] = np.array(loc["biasr"], np.float64)
            cur["outlierThres"] = np.array([loc["outlierThres"]], np.float64)
            cur["n_outlier_lines"] = np.array([len(loc["outliersline"])], np.int64)
            cur["outliersline"] = np.array(list(loc["outliersline"]), np.int64)
            cur["outliersdist"] = np.array(list(loc["outliersdist"]), np.int64)
            cur["fdr_y"] = np.array(loc["FDRy"], np.int64)
            if loc.get("splineX") is not None:
                ius = loc["ius"]
                t, c, k = ius._eval_args
                cur["spl_t"] = np.array(t, np.float64)
                cur["spl_c"] = np.array(c, np.float64)[:len(t) - 4]
                cur["spl_s_fp_ier"] = np.array([loc["splineError"], ius._data[10], ius._data[-1]], np.float64)
                cur["splineX"] = np.array(loc["splineX"], np.int64)
                cur["splineY"] = np.array(loc["splineY"], np.float64)
                cur["newSplineY"] = np.array(loc["newSplineY"], np.float64)
                cur["residual"] = np.array([loc["residual"]], np.float64)
                cur["x_sorted"] = np.array(loc["x"], np.float64)
                cur["y_sorted"] = np.array(loc["y"], np.float64)

    old_argv = sys.argv
    sys.argv = ["fithic"] + list(argv)
    sink = io.StringIO()
    t0 = time.time()
    try:
        if observed is not None:
            F.read_Interactions = given_observations
        sys.setprofile(hook)
        with (contextlib.redirect_stdout(sink) if quiet else contextlib.nullcontext()):
            F.main()
    finally:
        sys.setprofile(None)
        sys.argv = old_argv
        F.read_Interactions = real_read
    return passes, time.time() - t0


def _md5_decompressed(path):
    h = hashlib.md5()
    with gzip.open(path, "rb") as f:
        for chunk in iter(lambda: f.read(1 << 20), b""):
            h.update(chunk)
    return h.hexdigest()


def _collect_outputs(outdir, lib, res, npasses):
    meta = {}
    tag = (".res%d" % res) if res else ""
    for i in range(1, npasses + 1):
        sig = os.path.join(outdir, "%s.spline_pass%d%s.significances.txt.gz" % (lib, i, tag))
        if os.path.exists(sig):
            meta["sig_md5_pass%d" % i] = _md5_decompressed(sig)
            with gzip.open(sig, "rt") as f:
                lines = f.readlines()
            meta["sig_rows_pass%d" % i] = len(lines) - 1
            # a few verbatim output rows (head / middle / tail) as text known-answers
            idx = sorted(set([0, 1, 2, 3, len(lines) // 2, len(lines) - 2, len(lines) - 1]))
            meta["sig_sample_pass%d" % i] = {str(j): lines[j] for j in idx if 0 <= j < len(lines)}
        fp = os.path.join(outdir, "%s.fithic_pass%d%s.txt" % (lib, i, tag))
        if os.path.exists(fp):
            meta["fithic_pass%d_txt" % i] = open(fp).read()
    log = os.path.join(outdir, lib + ".fithic.log")
    if os.path.exists(log):
        meta["log_txt"] = open(log).read()
    return meta


def save_case(name, argv_tail, files, passes, elapsed, outdir, lib, res, subsample=1):
    """Write tests/golden/<name>.npz (+ .json).  Per-pair arrays are sub-sampled by `subsample`."""
    arrays = {}
    meta = {"name": name, "argv": argv_tail, "files": files, "n_passes": len(passes),
            "reference_seconds_1core": round(elapsed, 2), "subsample": subsample}
    for pi, P in enumerate(passes, start=1):
        n = len(P["p"])
        meta["n_rows"] = n
        sel = np.arange(0, n, subsample)
        for k, v in P.items():
            if k in ("p", "q", "expcc", "b1", "b2"):
                arrays["p%d_%s" % (pi, k)] = v[sel]
            else:
                arrays["p%d_%s" % (pi, k)] = v
        q = P["q"]
        meta["pass%d" % pi] = {
            "n_q_lt_0.01": int(np.sum(q < 0.01)), "n_q_lt_0.05": int(np.sum(q < 0.05)),
            "n_p_lt_1": int(np.sum(P["p"] < 1.0)), "n_nan_p": int(np.sum(np.isnan(P["p"]))),
            "n_outlier_lines": int(P["n_outlier_lines"][0]),
            "p_sha256": hashlib.sha256(P["p"].tobytes()).hexdigest(),
            "q_sha256": hashlib.sha256(q.tobytes()).hexdigest(),
        }
    meta.update(_collect_outputs(outdir, lib, res, len(passes)))
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **arrays)
    with open(os.path.join(HERE, name + ".json"), "w") as f:
        json.dump(meta, f, indent=1, sort_keys=True)
    print("  wrote %s: %d passes, %d rows, reference took %.1f s" % (name, len(passes), meta.get("n_rows", -1), elapsed))


def run_case(name, contacts, frags, bias, res, extra, subsample=1, npasses=1):
    tmp = tempfile.mkdtemp(prefix="golden_")
    lib = "G"
    argv = ["-i", os.path.join(DATA, contacts), "-f", os.path.join(DATA, frags), "-o", tmp,
            "-r", str(res), "-l", lib] + extra
    if bias:
        argv += ["-t", os.path.join(DATA, bias)]
    passes, dt = run_reference(argv)
    tail = ["-r", str(res)] + extra
    save_case(name, tail, {"contacts": contacts, "frags": frags, "bias": bias}, passes, dt, tmp, lib, res, subsample)
    shutil.rmtree(tmp)


# ------------------------------------------------------------------------------------------------ F1
def make_f1():
    print("F1: bundled hESC chr1 40 kb")
    base = ["-L", "50000", "-U", "5000000", "-b", "50"]
    run_case("f1_nobias", "hESC_chr1_w40000.contacts.gz", "hESC_chr1_w40000.frags.gz", None, 40000,
             base + ["-p", "1", "-x", "All"], subsample=29)
    run_case("f1_bias", "hESC_chr1_w40000.contacts.gz", "hESC_chr1_w40000.frags.gz",
             "hESC_chr1_w40000.bias.gz", 40000, base + ["-p", "2", "-x", "intraOnly"], subsample=29)


# ------------------------------------------------------------------------------------------------ F2
def _write_gz(path, text):
    with open(path, "wb") as raw:
        with gzip.GzipFile(fileobj=raw, mode="wb", mtime=0, filename="") as f:
            f.write(text.encode())


def synth_imr90_contacts(path, seed=20260928):
    """Deterministic synthetic contacts over the bundled IMR90 1 Mb loci (24 chromosomes)."""
    rng = np.random.default_rng(seed)
    chrom_mids = {}
    order = []
    with gzip.open(os.path.join(DATA, "IMR90_w1Mb.frags.gz"), "rt") as f:
        for line in f:
            w = line.split()
            if w[0] not in chrom_mids:
                chrom_mids[w[0]] = []
                order.append(w[0])
            chrom_mids[w[0]].append(int(w[2]))
    rows = []
    for ch in order:
        mids = chrom_mids[ch]
        n = len(mids)
        for i in range(n):
            for j in range(i, min(n, i + 1 + 70)):
                d = j - i
                lam = 9.0 if d == 0 else 14.0 * d ** -1.05
                if rng.random() < 0.55:
                    continue
                c = rng.poisson(lam * rng.lognormal(0.0, 0.35))
                if c < 1:
                    continue
                if rng.random() < 0.01:
                    c *= 5
                rows.append((ch, mids[i], ch, mids[j], c))
    ncis = len(rows)
    allloci = [(ch, m) for ch in order for m in chrom_mids[ch]]
    ntrans = 0
    while ntrans < 7000:
        a = allloci[rng.integers(len(allloci))]
        b = allloci[rng.integers(len(allloci))]
        if a[0] == b[0]:
            continue
        rows.append((a[0], a[1], b[0], b[1], 1 + rng.poisson(0.7)))
        ntrans += 1
    # interleave a little: shuffle blocks so that inter rows are not all at the end
    perm = rng.permutation(len(rows))
    rows = [rows[i] for i in perm]
    # counts are written as floats with a fractional part now and then (int(float()) truncation, A4)
    out = []
    for k, (c1, m1, c2, m2, c) in enumerate(rows):
        if k % 97 == 0:
            out.append("%s\t%d\t%s\t%d\t%.1f\n" % (c1, m1, c2, m2, c + 0.7))
        else:
            out.append("%s\t%d\t%s\t%d\t%d\n" % (c1, m1, c2, m2, c))
    _write_gz(path, "".join(out))
    return ncis, ntrans


def make_f2():
    print("F2: synthetic multi-chromosome over IMR90 1 Mb loci")
    contacts = "synth_IMR90_w1Mb.contacts.gz"
    ncis, ntrans = synth_imr90_contacts(os.path.join(DATA, contacts))
    print("  synthetic contacts: %d cis + %d trans rows" % (ncis, ntrans))
    fr, bi = "IMR90_w1Mb.frags.gz", "IMR90_w1Mb.bias.gz"
    run_case("f2_all", contacts, fr, bi, 1000000, ["-b", "20", "-p", "2", "-x", "All"])
    run_case("f2_inter", contacts, fr, bi, 1000000, ["-b", "20", "-p", "2", "-x", "interOnly"])
    run_case("f2_intra", contacts, fr, bi, 1000000,
             ["-b", "20", "-p", "2", "-x", "intraOnly", "-L", "2000000", "-U", "50000000"])
    run_case("f2_all_nobias", contacts, fr, None, 1000000,
             ["-b", "15", "-p", "1", "-x", "All", "-L", "1000000", "-U", "60000000"])


# ------------------------------------------------------------------------------------------------ F3
def make_f3():
    print("F3: bdtrc / betaln / log known answers")
    rng = np.random.default_rng(7)
    ks, ns, ps = [], [], []
    # realistic Hi-C regimes: n = total in-range contacts, count small, expected around count
    for n in (10 ** 5, 649_576, 6_495_767, 3_549_437, 10 ** 7, 123_456_789, 10 ** 9, 2_000_000_011):
        for _ in range(1400):
            count = int(min(1 + rng.geometric(0.08 if rng.random() < 0.8 else 0.004), 200000))
            ratio = math.exp(rng.normal(0.0, 1.2))          # expected / observed
            prior = min(max(count * ratio / n, 1e-12), 0.999)
            ks.append(count - 1)
            ns.append(n)
            ps.append(prior)
    # low-expected regime (pseries branch: bb*xx <= 1)
    for n in (10 ** 5, 6_495_767, 10 ** 9):
        for _ in range(400):
            count = int(rng.integers(1, 12))
            prior = rng.uniform(1e-3, 1.0) / n
            ks.append(count - 1)
            ns.append(n)
            ps.append(prior)
    # small n (a+b < MAXGAM: pow/beta path) and edge cases
    for n in (5, 20, 60, 120, 169, 170, 171, 172, 400):
        for _ in range(60):
            k = int(rng.integers(0, n + 1))
            ks.append(k - 1 if k > 0 else 0)
            ns.append(n)
            ps.append(float(rng.uniform(0, 1)))
    edge = [(-1, 10, .5), (10, 10, .5), (3, 10, 0.0), (3, 10, 1.0), (0, 10, 0.005), (0, 10, 0.5),
            (3, 10, -0.1), (3, 10, 1.1), (11, 10, .5), (0, 6495767, 1e-7), (0, 6495767, 0.02),
            (4, 6495767, float("nan")), (2.7, 50, 0.3), (5, 6495767, -1e-9), (5, 6495767, 0.96),
            (50, 1000, 0.97), (3, 100000, 0.999)]
    for k, n, p in edge:
        ks.append(k)
        ns.append(n)
        ps.append(p)
    ks = np.array(ks, np.float64)
    ns = np.array(ns, np.int64)
    ps = np.array(ps, np.float64)
    with np.errstate(all="ignore"):
        vals = scsp.bdtrc(ks, ns, ps)
    # lbeta / log at table arguments
    lb_n = np.array([649_576, 6_495_767, 3_549_437, 123_456_789, 2_000_000_011, 150, 169], np.int64)
    lb_c = np.array(list(range(1, 400)) + [500, 1000, 5000, 40000, 100000], np.int64)
    lb = np.array([[scsp.betaln(float(c), float(n - c + 1)) if c <= n else np.nan for c in lb_c] for n in lb_n])
    logs_x = np.concatenate([lb_n.astype(np.float64), lb_n + 1.0, np.arange(1, 300, dtype=np.float64),
                             rng.uniform(1e-9, 1e-2, 300), 1.0 - rng.uniform(1e-9, 1e-2, 300)])
    logs = np.array([math.log(v) for v in logs_x])
    np.savez_compressed(os.path.join(HERE, "f3_bdtrc.npz"), k=ks, n=ns, p=ps, val=vals,
                        lb_n=lb_n, lb_c=lb_c, lbeta=lb, log_x=logs_x, log_val=logs)
    print("  %d bdtrc vectors (%d NaN), %d lbeta, %d log" % (len(ks), int(np.isnan(vals).sum()), lb.size, logs.size))


# ------------------------------------------------------------------------------------------------ F4
def make_f4():
    print("F4: UnivariateSpline known answers")
    g = np.load(os.path.join(HERE, "f1_nobias.npz"))
    cases = []
    xb, yb = g["p1_x"], g["p1_y"]
    base = float(min(yb)) ** 2
    for f in (1, 0.3, 0.1, 0.03, 0.01, 0.001, 0.0):
        cases.append(("hESC_f%g" % f, xb, yb, base * f))
    g2 = np.load(os.path.join(HERE, "f1_bias.npz"))
    for pi in (1, 2):
        xx, yy = g2["p%d_x_sorted" % pi], g2["p%d_y_sorted" % pi]
        cases.append(("hESCbias_pass%d" % pi, xx, yy, float(min(yy)) ** 2))
    rng = np.random.default_rng(11)
    for m in (8, 9, 12, 20, 50, 100, 200):
        x = np.cumsum(rng.uniform(0.5, 1.5, m)) * 20000.0 + 20000.0
        y = 3e-3 * (x / 20000.0) ** -1.1 * np.exp(rng.normal(0, 0.05, m))
        for f in (1, 0.1, 0.001):
            cases.append(("synth_m%d_f%g" % (m, f), x, y, float(min(y)) ** 2 * f))
    out = {}
    names = []
    for name, x, y, s in cases:
        ius = UnivariateSpline(x, y, s=s)
        t, c, k = ius._eval_args
        n = len(t)
        names.append(name)
        out[name + "_x"] = np.asarray(x, np.float64)
        out[name + "_y"] = np.asarray(y, np.float64)
        out[name + "_t"] = np.asarray(t, np.float64)
        out[name + "_c"] = np.asarray(c, np.float64)[:n - 4]
        out[name + "_sfpier"] = np.array([s, ius._data[10], ius._data[-1]], np.float64)
        xe = np.linspace(x[0], x[-1], 301)
        out[name + "_xe"] = xe
        out[name + "_ye"] = ius(xe)
        nest0 = max(len(x) // 2, 8) if s > 0 else len(x) + 4
        print("   %-22s m=%3d s=%.3e -> n=%3d ier=%d restart=%s" % (name, len(x), s, n, ius._data[-1], n > nest0))
    out["names"] = np.array(names)
    np.savez_compressed(os.path.join(HERE, "f4_fitpack.npz"), **out)


# ------------------------------------------------------------------------------------------------ F5
def make_f5():
    print("F5: benjamini_hochberg_correction known answers")
    rng = np.random.default_rng(5)
    out = {}
    names = []

    def add(name, p, N):
        q = np.array(REF_STATS.benjamini_hochberg_correction(list(p), N), np.float64)
        out[name + "_p"] = np.asarray(p, np.float64)
        out[name + "_N"] = np.array([N], np.float64)
        out[name + "_q"] = q
        names.append(name)

    add("uniform", rng.uniform(0, 1, 5000), 5000)
    add("bigN", rng.uniform(0, 1, 5000) ** 6, 1529788)
    p = rng.choice(np.concatenate([rng.uniform(0, 1, 40) ** 3, [1.0, 0.0]]), 4000)
    add("ties", p, 9000)
    p = rng.uniform(0, 1, 3000) ** 8
    p[rng.integers(0, 3000, 300)] = 1.0
    p[rng.integers(0, 3000, 50)] = np.nan
    p[rng.integers(0, 3000, 20)] = 0.0
    p[rng.integers(0, 3000, 20)] = 5e-324
    p[rng.integers(0, 3000, 20)] = 1e-310
    add("mixed_nan_one_zero_denormal", p, 123457)
    add("all_ones", np.ones(17), 40)
    add("single", np.array([0.03]), 10)
    add("tiny", np.array([0.03, 0.4, 0.7, 0.01]), 10)
    add("smallN", rng.uniform(0, 1, 200), 3)       # N < len(p): values exceed... min(.,1) saturates
    out["names"] = np.array(names)
    np.savez_compressed(os.path.join(HERE, "f5_bh.npz"), **out)
    print("  %d BH vectors" % len(names))


def make_f12():
    """benjamini_hochberg_correction with a number of tests that is zero or NEGATIVE.  fit_Spline can hand it one: N is the
    possible-pair count, and `npairs = n - idx` goes negative with unmappable loci (fithic.py:613-615).  The loop
    starts its running maximum at the int 0 (myStats.py:30), so every negative bh value becomes 0."""
    print("F12: benjamini_hochberg_correction, N <= 0")
    rng = np.random.default_rng(12)
    out = {}
    names = []

    def add(name, p, N):
        q = np.array(REF_STATS.benjamini_hochberg_correction(list(p), N), np.float64)
        out[name + "_p"] = np.asarray(p, np.float64)
        out[name + "_N"] = np.array([N], np.float64)
        out[name + "_q"] = q
        names.append(name)

    add("neg_uniform", rng.uniform(0, 1, 3000), -5000)
    p = rng.uniform(0, 1, 2000) ** 6
    p[rng.integers(0, 2000, 200)] = 1.0
    p[rng.integers(0, 2000, 40)] = np.nan
    p[rng.integers(0, 2000, 15)] = 0.0
    add("neg_mixed_nan_one_zero", p, -37)
    add("neg_fraction", rng.uniform(0, 1, 500) ** 3, -0.25)
    add("neg_tiny", np.array([0.03, 0.4, 1.0, 0.01]), -10)
    add("zero_tests", np.array([0.03, 0.4, 1.0, 0.01, 0.0]), 0)
    add("neg_leading_zero_p", np.array([0.0, 0.2, 0.5, 1.0]), -3)
    out["names"] = np.array(names)
    np.savez_compressed(os.path.join(HERE, "f12_bh_nonpositive_N.npz"), **out)
    print("  %d BH vectors" % len(names))


# ------------------------------------------------------------------------------------------------ F6
def make_f6():
    print("F6: quirk probes")
    res = 10000
    # 3 chromosomes whose names sort differently as strings (chr1 < chr10 < chr2); chr10 has an
    # unmappable tail (hits 0) so that npairs goes negative (A7); duplicate + missing bias rows (A11)
    frag_lines, bias_lines = [], []
    nloci = {"chr1": 60, "chr10": 45, "chr2": 52}
    for ch in ("chr1", "chr2", "chr10"):
        for i in range(nloci[ch]):
            hits = 2 if i % 4 else 1
            if ch == "chr10" and i in (3, 4, 20):
                hits = 0
            frag_lines.append("%s\t0\t%d\t%d\t1\n" % (ch, i * res + res // 2, hits))
            b = 0.6 + ((i * 37) % 23) / 16.0
            if i % 17 == 5:
                b = 0.3
            if i % 19 == 7:
                b = 2.4
            if i == 11:
                bias_lines.append("%s\t%d\tnan\n" % (ch, i * res + res // 2))
                continue
            if ch == "chr2" and i in (30, 31):
                continue                        # missing locus
            bias_lines.append("%s\t%d\t%.6f\n" % (ch, i * res + res // 2, b))
            if i == 8:
                bias_lines.append("%s\t%d\t%.6f\n" % (ch, i * res + res // 2, 1.9))   # duplicate: first wins
    _write_gz(os.path.join(DATA, "quirk.frags.gz"), "".join(frag_lines))
    _write_gz(os.path.join(DATA, "quirk.bias.gz"), "".join(bias_lines))
    rng = np.random.default_rng(3)
    rows = []
    for ch in ("chr1", "chr2", "chr10"):
        n = nloci[ch]
        for i in range(n):
            for j in range(i, n):
                d = j - i
                if rng.random() < 0.5:
                    continue
                c = rng.poisson(30.0 / (1 + d) ** 0.9) + (1 if rng.random() < 0.3 else 0)
                if c < 1:
                    continue
                rows.append("%s\t%d\t%s\t%d\t%d\n" % (ch, i * res + res // 2, ch, j * res + res // 2, c))
    rows.append("chr1\t5000\tchr1\t45000\t0.4\n")          # count truncating to 0 (A4)
    rows.append("chr1\t15000\tchr1\t55000\t3.9\n")
    rows.append("chr3\t5000\tchr3\t45000\t7\n")            # chromosome absent from frags and bias
    for _ in range(150):
        a, b = rng.choice(["chr1", "chr2", "chr10"], 2, replace=False)
        rows.append("%s\t%d\t%s\t%d\t%d\n" % (a, int(rng.integers(nloci[a])) * res + res // 2,
                                             b, int(rng.integers(nloci[b])) * res + res // 2, 1 + rng.poisson(0.5)))
    _write_gz(os.path.join(DATA, "quirk.contacts.gz"), "".join(rows))
    c, f, b = "quirk.contacts.gz", "quirk.frags.gz", "quirk.bias.gz"
    run_case("f6_quirk_all", c, f, b, res, ["-b", "12", "-p", "2", "-x", "All", "-L", "20000", "-U", "400000"])
    run_case("f6_quirk_intra_nobounds", c, f, b, res, ["-b", "10", "-p", "1", "-x", "intraOnly"])
    run_case("f6_quirk_zero_flags", c, f, None, res, ["-b", "0", "-p", "0", "-L", "0", "-U", "0", "-m", "0"])
    run_case("f6_quirk_mapp2", c, f, b, res, ["-b", "8", "-p", "1", "-m", "2", "-x", "All", "-tL", "0.4", "-tU", "2.5"])


# ------------------------------------------------------------------------------------------------ F7 / F8
def make_f7():
    """Fixed-size, 14 chromosomes: synthetic contacts over the bundled P. falciparum 10 kb loci (run_tests-git.sh:52-54)."""
    print("F7: synthetic contacts over Ay_Rings_MboI_Pfal_w10000 loci, -x All")
    shutil.copy("/root/reference/fithic/tests/data/fragmentLists/Ay_Rings_MboI_Pfal_w10000.gz", os.path.join(DATA, "Pfal_w10000.frags.gz"))
    rng = np.random.default_rng(77)
    loci = {}
    order = []
    with gzip.open(os.path.join(DATA, "Pfal_w10000.frags.gz"), "rt") as f:
        for line in f:
            w = line.split()
            if w[0] not in loci:
                loci[w[0]] = []
                order.append(w[0])
            loci[w[0]].append(int(w[2]))
    rows = []
    for ch in order:
        m = loci[ch]
        for i in range(len(m)):
            for j in range(i, len(m)):
                lam = 60.0 / (1 + (j - i)) ** 1.1
                c = rng.poisson(lam * rng.lognormal(0, 0.4))
                if c >= 1 and rng.random() < 0.7:
                    rows.append("%s\t%d\t%s\t%d\t%d\n" % (ch, m[i], ch, m[j], c))
    allloci = [(ch, v) for ch in order for v in loci[ch]]
    for _ in range(9000):
        a, b = allloci[rng.integers(len(allloci))], allloci[rng.integers(len(allloci))]
        if a[0] != b[0]:
            rows.append("%s\t%d\t%s\t%d\t%d\n" % (a[0], a[1], b[0], b[1], 1 + rng.poisson(1.0)))
    perm = rng.permutation(len(rows))
    _write_gz(os.path.join(DATA, "synth_Pfal_w10000.contacts.gz"), "".join(rows[i] for i in perm))
    run_case("f7_pfal_all", "synth_Pfal_w10000.contacts.gz", "Pfal_w10000.frags.gz", None, 10000, ["-b", "200", "-p", "2", "-x", "All"])


def make_f8():
    """Non-fixed-size mode (-r 0): synthetic contacts over the bundled hESC combineFrags10 chr1 fragments
    (run_tests-git.sh:28-30) and over a small irregular 3-chromosome fragment set with bias and inter-chromosomal rows."""
    print("F8: -r 0 (non-fixed-size)")
    shutil.copy("/root/reference/fithic/tests/data/fragmentLists/Dixon_hESC_HindIII_hg18_combineFrags10_chr1.gz",
                os.path.join(DATA, "hESC_combineFrags10_chr1.frags.gz"))
    rng = np.random.default_rng(88)
    mids, hits = [], []
    with gzip.open(os.path.join(DATA, "hESC_combineFrags10_chr1.frags.gz"), "rt") as f:
        for line in f:
            w = line.split()
            mids.append(int(w[2]))
            hits.append(int(w[3]))
    mids = np.array(mids)
    rows = []
    n = len(mids)
    for i in range(n):
        js = np.arange(i + 1, min(n, i + 160))
        d = np.abs(mids[js] - mids[i]).astype(np.float64)
        lam = 2.5e6 / np.maximum(d, 2e4) ** 1.05
        c = rng.poisson(lam * rng.lognormal(0, 0.5, len(js)))
        keep = (c >= 1) & (rng.random(len(js)) < 0.6)
        for j, cc in zip(js[keep], c[keep]):
            rows.append("1\t%d\t1\t%d\t%d\n" % (mids[i], mids[j], cc))
    perm = rng.permutation(len(rows))
    _write_gz(os.path.join(DATA, "synth_hESC_combineFrags10_chr1.contacts.gz"), "".join(rows[i] for i in perm))
    print("  %d synthetic rows over %d fragments" % (len(rows), n))
    run_case("f8_nonfixed_hESC", "synth_hESC_combineFrags10_chr1.contacts.gz", "hESC_combineFrags10_chr1.frags.gz", None, 0,
             ["-L", "50000", "-U", "5000000", "-b", "200", "-p", "1", "-x", "intraOnly"], subsample=3)
    # small irregular multi-chromosome set: bias file, inter rows, two passes, All
    frag_lines, bias_lines, loci = [], [], {}
    for ch, nl in (("chrA", 90), ("chrB", 70), ("chrC", 40)):
        pos = np.cumsum(rng.integers(3000, 40000, nl))
        loci[ch] = pos
        for k, m in enumerate(pos):
            frag_lines.append("%s\t0\t%d\t%d\t1\n" % (ch, m, 0 if k % 23 == 7 else 1 + k % 3))
            if k % 29 != 3:
                bias_lines.append("%s\t%d\t%.5f\n" % (ch, m, float(np.exp(rng.normal(0, 0.35)))))
    _write_gz(os.path.join(DATA, "irregular.frags.gz"), "".join(frag_lines))
    _write_gz(os.path.join(DATA, "irregular.bias.gz"), "".join(bias_lines))
    rows = []
    for ch, pos in loci.items():
        for i in range(len(pos)):
            for j in range(i + 1, len(pos)):
                d = float(pos[j] - pos[i])
                c = rng.poisson(3e5 / d ** 0.95)
                if c >= 1 and rng.random() < 0.6:
                    rows.append("%s\t%d\t%s\t%d\t%d\n" % (ch, pos[i], ch, pos[j], c))
    names = list(loci)
    for _ in range(400):
        a, b = rng.choice(3, 2, replace=False)
        rows.append("%s\t%d\t%s\t%d\t%d\n" % (names[a], rng.choice(loci[names[a]]), names[b], rng.choice(loci[names[b]]), 1 + rng.poisson(0.6)))
    perm = rng.permutation(len(rows))
    _write_gz(os.path.join(DATA, "irregular.contacts.gz"), "".join(rows[i] for i in perm))
    run_case("f8_nonfixed_all", "irregular.contacts.gz", "irregular.frags.gz", "irregular.bias.gz", 0,
             ["-b", "15", "-p", "2", "-x", "All", "-L", "10000", "-U", "900000"])
    run_case("f8_nonfixed_nobounds", "irregular.contacts.gz", "irregular.frags.gz", "irregular.bias.gz", 0,
             ["-b", "10", "-p", "1", "-x", "intraOnly"])


def make_f9():
    """Knight-Ruiz bias vectors (fithic/utils/HiCKRy.py) for the bundled hESC map and the synthetic sets, through the
    reference's own main(); intermediate results through its functions."""
    print("F9: HiCKRy")
    sys.path.insert(0, os.path.join(REF_PKG, "utils"))
    import HiCKRy as K
    cases = [("k1_kr_hESC", "hESC_chr1_w40000.contacts.gz", "hESC_chr1_w40000.frags.gz", 0.12),
             ("k1_kr_hESC_default", "hESC_chr1_w40000.contacts.gz", "hESC_chr1_w40000.frags.gz", 0.05),   # hits the 30-iteration cap
             ("k2_kr_pfal", "synth_Pfal_w10000.contacts.gz", "Pfal_w10000.frags.gz", 0.05),
             ("k3_kr_imr90", "synth_IMR90_w1Mb.contacts.gz", "IMR90_w1Mb.frags.gz", 0.1),               # non-integer counts
             ("k4_kr_irregular", "irregular.contacts.gz", "irregular.frags.gz", 0.0),
             ("k5_kr_combine", "synth_hESC_combineFrags10_chr1.contacts.gz", "hESC_combineFrags10_chr1.frags.gz", 0.03)]
    for name, con, frg, perc in cases:
        cpath, fpath = os.path.join(DATA, con), os.path.join(DATA, frg)
        tmp = tempfile.mkdtemp(prefix="kr_golden_")
        out = os.path.join(tmp, "bias.gz")
        buf = io.StringIO()
        argv0 = sys.argv
        sys.argv = ["HiCKRy.py", "-i", cpath, "-f", fpath, "-o", out, "-x", str(perc)]
        try:
            with contextlib.redirect_stdout(buf):
                K.main()
        finally:
            sys.argv = argv0
        with gzip.open(out, "rb") as f:
            text = f.read()
        with contextlib.redirect_stdout(io.StringIO()):
            M, rev = K.loadfastfithicInteractions(cpath, fpath)
            sums = np.array(M.sum(axis=0)).reshape(-1)
            mtx, removed = K.removeZeroDiagonalCSR(M, perc)
            x, i, k = K.knightRuizAlg(mtx)
            bias = K.addZeroBiases(removed, K.computeBiasVector(x)).ravel()
        file_bias = np.array([float(ln.split(b"\t")[2]) for ln in text.splitlines()])
        assert np.array_equal(file_bias, bias)
        stdout = "\n".join(ln for ln in buf.getvalue().splitlines() if "took" not in ln)
        np.savez_compressed(os.path.join(HERE, name + ".npz"), removed=np.array(removed, np.int64), x=x.ravel(), bias=bias,
                            row_sums=sums, indptr_reduced=mtx.indptr.astype(np.int64),
                            data_checksum=np.array([M.data.sum(), mtx.data.sum()]))
        meta = dict(name=name, contacts=con, frags=frg, perc=perc, n=int(M.shape[0]), nnz=int(M.nnz), nnz_reduced=int(mtx.nnz),
                    outer=int(i), inner=int(k), out_md5=hashlib.md5(text).hexdigest(), out_lines=text.count(b"\n"),
                    stdout=stdout, first_lines=text.decode().splitlines()[:5])
        with open(os.path.join(HERE, name + ".json"), "w") as f:
            json.dump(meta, f, indent=1)
        shutil.rmtree(tmp)
        print("  %s: n=%d nnz=%d removed=%d iterations=(%d,%d)" % (name, M.shape[0], M.nnz, len(removed), i, k))


def make_f10():
    """utils/CombineNearbyInteraction.py: merged loop lists for a real significances file (the reference's own Fit-Hi-C run on
    the bundled hESC chr1 set, rows with q < 1e-20) and for a synthetic 3-chromosome cluster set, in the modes its flags offer."""
    import subprocess
    print("F10: CombineNearbyInteraction")
    tool = os.path.join(REF_PKG, "utils", "CombineNearbyInteraction.py")
    # (a) real: run the reference Fit-Hi-C, keep the significant rows as they were written
    tmp = tempfile.mkdtemp(prefix="cni_golden_")
    run_reference(["-i", os.path.join(DATA, "hESC_chr1_w40000.contacts.gz"), "-f", os.path.join(DATA, "hESC_chr1_w40000.frags.gz"),
                   "-t", os.path.join(DATA, "hESC_chr1_w40000.bias.gz"), "-o", tmp, "-l", "G", "-r", "40000", "-L", "50000",
                   "-U", "5000000", "-b", "50", "-p", "1", "-x", "intraOnly"])
    with gzip.open(os.path.join(tmp, "G.spline_pass1.res40000.significances.txt.gz"), "rt") as f:
        lines = f.read().splitlines(True)
    keep = [lines[0]] + [ln for ln in lines[1:] if float(ln.split()[6]) < 1e-20]     # 6.5 k rows: the reference pairs all nodes (O(n^2))
    _write_gz(os.path.join(DATA, "hESC_chr1_sig_q1e20.txt.gz"), "".join(keep))
    print("  real input: %d significant rows" % (len(keep) - 1))
    shutil.rmtree(tmp)
    # (b) synthetic: clusters on three chromosomes, duplicates, a diagonal cell, tied q-values, one inter-chromosomal row
    rng = np.random.default_rng(1010)
    res = 40000
    rows = []
    for ch in ("chr2", "chr10", "chr1"):
        for _ in range(14):
            a, d = int(rng.integers(5, 200)), int(rng.integers(0, 60))
            for _ in range(int(rng.integers(1, 45))):
                b1 = a + int(rng.integers(-3, 4))
                b2 = b1 + d + int(rng.integers(-3, 4))
                if b1 < 0 or b2 < b1:
                    continue
                q = float(np.round(10 ** rng.uniform(-12, -2), 13)) if rng.random() < 0.8 else 1e-6
                rows.append("%s\t%d\t%s\t%d\t%d\t%e\t%e\t1.0\t1.0\t%f\n" % (ch, b1 * res + res // 2, ch, b2 * res + res // 2,
                                                                              int(rng.integers(5, 200)), q / 50, q, 1.0))
    rows.append("chr1\t20000\tchr2\t60000\t5\t1.000000e-05\t1.000000e-04\t1.0\t1.0\t1.000000\n")
    rows += [rows[3], rows[40]]                                   # the same cell twice: the first row keeps its values
    perm = rng.permutation(len(rows))
    body = "".join(rows[i] for i in perm)
    head = "chr1\tfragmentMid1\tchr2\tfragmentMid2\tcontactCount\tp-value\tq-value\tbias1\tbias2\tExpCC\n"
    _write_gz(os.path.join(DATA, "synth_sig_clusters.txt.gz"), head + body)
    _write_gz(os.path.join(DATA, "synth_sig_clusters_nohdr.txt.gz"), body)
    cases = [("c1_combine_hESC_default", "hESC_chr1_sig_q1e20.txt.gz", []),
             ("c1_combine_hESC_c4_n1", "hESC_chr1_sig_q1e20.txt.gz", ["-c", "4", "-n", "1"]),
             ("c1_combine_hESC_p50", "hESC_chr1_sig_q1e20.txt.gz", ["-p", "50"]),
             ("c2_combine_synth_default", "synth_sig_clusters.txt.gz", []),
             ("c2_combine_synth_c4", "synth_sig_clusters.txt.gz", ["-c", "4"]),
             ("c2_combine_synth_p50", "synth_sig_clusters.txt.gz", ["-p", "50"]),
             ("c2_combine_synth_p10_n3", "synth_sig_clusters.txt.gz", ["-p", "10", "-n", "3"]),
             ("c2_combine_synth_s1", "synth_sig_clusters.txt.gz", ["-s", "1"]),
             ("c2_combine_synth_p30_s1", "synth_sig_clusters.txt.gz", ["-p", "30", "-s", "1"]),
             ("c2_combine_synth_nohdr", "synth_sig_clusters_nohdr.txt.gz", ["-H", "0"])]
    for name, inp, extra in cases:
        tmp = tempfile.mkdtemp(prefix="cni_golden_")
        out = os.path.join(tmp, "merged.gz")
        env = dict(os.environ, LC_ALL="C")
        subprocess.run([sys.executable, tool, "-i", os.path.join(DATA, inp), "-o", out, "-r", "40000"] + extra, check=True,
                       stdout=subprocess.DEVNULL, env=env)
        with gzip.open(out, "rb") as f:
            text = f.read()
        _write_gz(os.path.join(HERE, name + ".out.gz"), text.decode())
        with open(os.path.join(HERE, name + ".json"), "w") as f:
            json.dump(dict(name=name, input=inp, argv=["-r", "40000"] + extra, out_md5=hashlib.md5(text).hexdigest(),
                           out_lines=len(text.decode().split("\n"))), f, indent=1)
        shutil.rmtree(tmp)
        print("  %s: %d lines" % (name, len(text.decode().split("\n"))))


def make_f11():
    """Fixed-size mode (-r N > 0) on loci that are NOT on one grid: the reference takes abs(mid1 - mid2) of whatever
    midpoints the files hold (myUtils.py:112-124, fithic.py:425-440) and still enumerates possible pairs at multiples of the
    resolution (fithic.py:592-689).  Uses the irregular 3-chromosome set of f8 (make_f8 writes the data files)."""
    print("F11: -r N on off-grid loci")
    run_case("f11_offgrid_all", "irregular.contacts.gz", "irregular.frags.gz", "irregular.bias.gz", 10000,
             ["-b", "12", "-p", "2", "-x", "All", "-L", "10000", "-U", "900000"])
    run_case("f11_offgrid_intra", "irregular.contacts.gz", "irregular.frags.gz", None, 5000,
             ["-b", "8", "-p", "2", "-x", "intraOnly", "-L", "5000", "-U", "600000"])


# ------------------------------------------------------------------------------------------------ F13
def make_f13():
    """More than two passes: the reference keeps ONE outliersline / outliersdist list for the whole run (fithic.py:336-370, 1216-1217),
    so pass 3 skips the outlier lines of pass 1 and of pass 2, and a line that is an outlier twice is in the lists twice.
    Run after f2 and f6 (it reuses their data files)."""
    print("F13: -p 3 and -p 4 on the f2 / f6 / f1 data")
    run_case("f13_all_p3", "synth_IMR90_w1Mb.contacts.gz", "IMR90_w1Mb.frags.gz", "IMR90_w1Mb.bias.gz", 1000000,
             ["-b", "20", "-p", "3", "-x", "All"])
    run_case("f13_quirk_p4", "quirk.contacts.gz", "quirk.frags.gz", "quirk.bias.gz", 10000,
             ["-b", "12", "-p", "4", "-x", "All", "-L", "20000", "-U", "400000"])
    run_case("f13_hESC_p3", "hESC_chr1_w40000.contacts.gz", "hESC_chr1_w40000.frags.gz", "hESC_chr1_w40000.bias.gz", 40000,
             ["-L", "50000", "-U", "5000000", "-b", "50", "-p", "3", "-x", "intraOnly"], subsample=29)



# ------------------------------------------------------------------------------------------------ F15
def _scaled_contacts(src, dst, cis_mul, trans_mul):
    """Rows of data/<src> with every cis count times cis_mul and every trans count times trans_mul (counts stay < 2^31)."""
    out = []
    sums = [0, 0]
    with gzip.open(os.path.join(DATA, src), "rt") as f:
        for line in f:
            c1, m1, c2, m2, cc = line.split()
            mul = cis_mul if c1 == c2 else trans_mul
            v = int(float(cc) * mul)                      # e.g. 0.4 x 500 000 -> 200 000
            assert 0 <= v < 2 ** 31
            sums[c1 != c2] += v
            out.append("%s\t%s\t%s\t%s\t%d\n" % (c1, m1, c2, m2, v))
    _write_gz(os.path.join(DATA, dst), "".join(out))
    return sums


def make_f15():
    """Totals at and above 2^31.  fit_Spline hands observedIntraInRangeSum / observedInterAllSum - Python ints, accumulated at
    fithic.py:436-440 - to scipy.special.bdtrc as n (fithic.py:1070, 1101); the ufunc loop that takes them is `dld->d` and its
    Cephes core takes `int n`: the C long is narrowed to 32 bits.  n = 2^31 arrives as -2^31 (n < k: NaN), n = 2^32 + 10^6 as
    10^6.  These fixtures pin that behaviour: known-answer vectors of bdtrc itself and whole runs of the reference's main()."""
    print("F15: totals >= 2^31 (scipy's bdtrc narrows n to a C int)")
    rng = np.random.default_rng(15)
    ks, ns, ps = [], [], []
    totals = [2 ** 31 - 1, 2 ** 31, 2 ** 31 + 1, 3_215_733_208, 2 ** 32 - 1, 2 ** 32, 2 ** 32 + 1, 2 ** 32 + 5, 2 ** 32 + 10 ** 6,
              2 ** 32 + 6_495_767, 7_150_761_687, 2 ** 33 + 123_456_789, 3 * 2 ** 32 + 649_576, 2 ** 40 + 10 ** 7, 2 ** 52 + 150]
    for n in totals:
        w = ((n + 2 ** 31) % 2 ** 32) - 2 ** 31           # what the C int holds
        for _ in range(260):
            count = int(min(1 + rng.geometric(0.08 if rng.random() < 0.8 else 0.004), 200000))
            ratio = math.exp(rng.normal(0.0, 1.2))
            base = w if (w > 0 and rng.random() < 0.7) else n          # expected ~ observed under the narrowed or the true total
            prior = min(max(count * ratio / base, 1e-13), 0.999)
            ks.append(count - 1)
            ns.append(n)
            ps.append(prior)
        for k, p in ((-1, 0.5), (0, 1e-9), (0, 0.3), (1, 1e-9), (3, 1e-9), (3, 0.0), (3, 1.0), (4, 0.3), (5, 0.3), (7, 0.3),
                     (0, float("nan")), (2, -0.1), (2, 1.5), (-1, 2.0)):
            ks.append(k)
            ns.append(n)
            ps.append(p)
    ks = np.array(ks, np.float64)
    ns = np.array(ns, np.int64)
    ps = np.array(ps, np.float64)
    with np.errstate(all="ignore"):
        vals = scsp.bdtrc(ks, ns, ps)                     # float64, int64, float64 -> the `dld->d` loop, as the reference's call
    np.savez_compressed(os.path.join(HERE, "f15_bdtrc_int_n.npz"), k=ks, n=ns, p=ps, val=vals)
    print("  %d bdtrc vectors at %d totals (%d NaN)" % (len(ks), len(totals), int(np.isnan(vals).sum())))
    # whole runs: the quirk set with its counts scaled (cis and trans separately, to put either total where it is wanted)
    fr, bi = "quirk.frags.gz", "quirk.bias.gz"
    cases = [
        # name, cis x, trans x, argv tail
        ("f15_intra_2p31_all", 500000, 500000, ["-b", "12", "-p", "2", "-x", "All", "-L", "20000", "-U", "400000"]),
        ("f15_intra_2p32_intra", 450000, 1, ["-b", "10", "-p", "2", "-x", "intraOnly"]),
        ("f15_inter_2p31_inter", 1, 10000000, ["-b", "12", "-p", "1", "-x", "interOnly"]),
        ("f15_both_2p32_all", 800000, 20000000, ["-b", "12", "-p", "1", "-x", "All", "-L", "20000", "-U", "400000"]),
    ]
    for name, cm, tm, tail in cases:
        contacts = "quirk_x%d_x%d.contacts.gz" % (cm, tm)
        sums = _scaled_contacts("quirk.contacts.gz", contacts, cm, tm)
        print("  %s: sum of cis counts %d, of trans counts %d" % (contacts, sums[0], sums[1]))
        run_case(name, contacts, fr, bi, 10000, tail)
        meta = json.load(open(os.path.join(HERE, name + ".json")))
        g = np.load(os.path.join(HERE, name + ".npz"))
        s = g["p1_sums"]
        print("    in-range sum %d, inter sum %d, NaN p in pass 1: %d of %d rows" % (s[3], s[1], meta["pass1"]["n_nan_p"], meta["n_rows"]))

# ------------------------------------------------------------------------------------------------ F14
def make_f14():
    """The fit at the HEADLINE sizes (bench.py's C3, C3w, C5: 22 autosomes at 5 kb / 1 kb, 576 216 / 2 881 044 loci, 397 / 49 734 /
    1 999 distance values): the reference's own main() - makeBinsFromInteractions, generate_FragPairs, calculateProbabilities and
    the spline + isotonic table of fit_Spline (fithic.py:463-689, 843-918, 936-968) - on the fragments file of the synthetic genome
    and on the distance histogram of the synth-v1 rows.  The 10^8..10^9 contact rows themselves cannot go through the reference's
    Python line loop (25 k rows/s); their histogram is what read_Interactions would have returned, computed with plain torch
    tensor arithmetic by tests/golden/dump_synth_hist.py on the GPU box (gpurun_out/r03/synth_hist/) and stored in the fixture,
    so that the GPU tests and bench.py first check the engine's K1 output against it and then its fit against the reference's.
    Stores every stage's outputs (bins, possible-pair counts, x, y, s, knots, coefficients, table) - no per-row arrays."""
    print("F14: the reference's fit on the full-size synthetic workloads")
    sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
    import bench
    from fithic_amd import synth
    src = os.environ.get("FHX_SYNTH_HIST", os.path.join(os.path.dirname(os.path.dirname(HERE)), "gpurun_out", "r03", "synth_hist"))
    for name in ("C2", "C3", "C3w", "C5"):                 # C2: pass 1 of its two passes (the second depends on the first's outliers)
        cfg = bench.CONFIGS[name]
        H = np.load(os.path.join(src, "synth_hist_%s.npz" % name))
        res = cfg["res"]
        genome = synth.Genome(res, cfg["lengths"])
        tmp = tempfile.mkdtemp(prefix="golden_f14_")
        frags = os.path.join(tmp, "frags.gz")
        with gzip.open(frags, "wt", compresslevel=1) as f:
            for c, n in enumerate(genome.n_loci):
                nm = genome.names[c]
                f.write("".join("%s\t0\t%d\t1\t0\n" % (nm, i * res + res // 2) for i in range(n)))
        contacts = os.path.join(tmp, "few.gz")           # fit_Spline's own loop: three in-range rows, nothing is kept of them
        lo = int(H["dist_idx"][0])
        _write_gz(contacts, "".join("chr1\t%d\tchr1\t%d\t%d\n" % (res // 2, (lo + k) * res + res // 2, k + 1) for k in range(3)))
        in_range_sum = int(H["sumcc"].sum())
        main_dic = {int(i) * res: [0, int(v)] for i, v in zip(H["dist_idx"], H["sumcc"])}
        inter_count, inter_sum = (int(v) for v in H["inter"])
        argv = ["-i", contacts, "-f", frags, "-o", tmp, "-r", str(res), "-l", "G", "-b", "100", "-p", "1", "-x", cfg["mode"]]
        if cfg["L"]:
            argv += ["-L", str(cfg["L"])]
        if cfg["U"] != float("inf"):
            argv += ["-U", str(cfg["U"])]
        passes, dt = run_reference(argv, observed=(main_dic, inter_count, inter_sum, in_range_sum, in_range_sum))
        P = passes[0]
        keep = {k: v for k, v in P.items() if k not in ("p", "q", "expcc", "b1", "b2", "outliersline", "outliersdist", "fdr_y",
                                                         "n_outlier_lines")}
        keep["hist_dist_idx"] = H["dist_idx"].astype(np.int64)
        keep["hist_sumcc"] = H["sumcc"].astype(np.int64)
        keep["hist_nrows"] = H["nrows"].astype(np.int64)
        keep["rows_per_chr"] = H["rows_per_chr"].astype(np.int64)
        keep["inter"] = H["inter"].astype(np.int64)
        np.savez_compressed(os.path.join(HERE, "f14_%s_fit.npz" % name), **keep)
        meta = dict(name="f14_%s_fit" % name, config=name, argv=[a for a in argv[6:]], n_rows=int(H["nrows"].sum()) + inter_count,
                    n_dist=len(H["dist_idx"]), in_range_sum=in_range_sum, inter_count=inter_count, inter_sum=inter_sum,
                    n_loci=int(sum(genome.n_loci)), n_bins=int(len(P["x"])), n_knots=int(len(P["spl_t"])),
                    n_table=int(len(P["splineX"])), possibleIntraInRangeCount=int(P["possibleIntraInRangeCount"][0]),
                    outlierThres=float(P["outlierThres"][0]), reference_seconds_1core=round(dt, 1),
                    histogram_from="tests/golden/dump_synth_hist.py on MI355X (torch %s)" % str(H["torch_version"][0]),
                    fithic_pass1_txt=open(os.path.join(tmp, "G.fithic_pass1.res%d.txt" % res)).read(),
                    log_txt=open(os.path.join(tmp, "G.fithic.log")).read() if name == "C3" else None)
        with open(os.path.join(HERE, "f14_%s_fit.json" % name), "w") as f:
            json.dump(meta, f, indent=1, sort_keys=True)
        shutil.rmtree(tmp)
        print("  f14_%s_fit: %d bins, %d knots, table of %d, N = %d, reference took %.1f s" %
              (name, meta["n_bins"], meta["n_knots"], meta["n_table"], meta["possibleIntraInRangeCount"], dt))


# ------------------------------------------------------------------------------------------------ calibration
def make_calib():
    """tests/golden/calibration.json: the REAL fithic.py (whole main(): text I/O included, and its fit_Spline alone) timed next to
    the oracle (arrays in memory) on the bundled hESC chr1 40 kb set, one core, in this container.  bench.py scales its CPU leg
    (the oracle timed on the GPU box) by reference/port to say what the reference itself would do there."""
    print("calibration: the reference and the oracle on the bundled hESC set")
    sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
    from oracle import fithic_oracle as fo
    contacts, frags, bias = (os.path.join(DATA, "hESC_chr1_w40000.%s.gz" % k) for k in ("contacts", "frags", "bias"))
    tmp = tempfile.mkdtemp(prefix="golden_calib_")
    argv = ["-i", contacts, "-f", frags, "-t", bias, "-o", tmp, "-r", "40000", "-l", "G", "-L", "50000", "-U", "5000000", "-b", "50",
            "-p", "1", "-x", "intraOnly"]
    spent = {"fit_Spline": 0.0}
    real_fit = F.fit_Spline

    def timed_fit(*a, **k):
        t = time.perf_counter()
        try:
            return real_fit(*a, **k)
        finally:
            spent["fit_Spline"] += time.perf_counter() - t
    old_argv = sys.argv
    sys.argv = ["fithic"] + argv
    F.fit_Spline = timed_fit
    try:
        t0 = time.perf_counter()
        with contextlib.redirect_stdout(io.StringIO()):
            F.main()
        t_ref = time.perf_counter() - t0
    finally:
        F.fit_Spline = real_fit
        sys.argv = old_argv
    shutil.rmtree(tmp)
    fo.build()
    pairs = fo.read_contacts_file(contacts)                 # the oracle's own readers: arrays in memory before the clock starts
    n = len(pairs)
    frag_rows = fo.read_fragments_file(frags)
    bias_dic = fo.read_biases(bias, 0.5, 2)
    t0 = time.perf_counter()
    fo.run(pairs, frag_rows, None, 40000, n_bins=50, passes=1, mode="intraOnly", L=50000, U=5000000, bias_dic=bias_dic)
    t_port = time.perf_counter() - t0
    cpu = None
    for line in open("/proc/cpuinfo"):
        if line.startswith("model name"):
            cpu = line.split(":", 1)[1].strip()
            break
    out = {"reference_rows_per_s": n / t_ref, "reference_fit_spline_only_rows_per_s": n / spent["fit_Spline"], "port_rows_per_s": n / t_port,
           "rows": n, "reference_seconds": round(t_ref, 2), "reference_fit_spline_seconds": round(spent["fit_Spline"], 2),
           "port_seconds": round(t_port, 3),
           "input": "bundled Dixon hESC chr1 40 kb, %d rows, -L 50000 -U 5000000 -b 50 -t bias, 1 pass, intraOnly" % n,
           "measured_in": "build container, 1 core (%s): reference = whole fithic.py main() incl. text I/O; port = oracle on arrays in memory" % cpu,
           "made_by": "tests/golden/make_golden.py calib"}
    with open(os.path.join(HERE, "calibration.json"), "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)
    print("  reference %.1f s (fit_Spline %.1f s), oracle %.2f s on %d rows" % (t_ref, spent["fit_Spline"], t_port, n))


if __name__ == "__main__":
    which = [a.lower() for a in sys.argv[1:]] or ["f1", "f2", "f3", "f4", "f5", "f6", "f7", "f8", "f9", "f10", "f11", "f12", "f13", "f15"]
    jobs = dict(f1=make_f1, f2=make_f2, f3=make_f3, f4=make_f4, f5=make_f5, f6=make_f6, f7=make_f7, f8=make_f8, f9=make_f9,
                f10=make_f10, f11=make_f11, f12=make_f12, f13=make_f13, f14=make_f14, f15=make_f15, calib=make_calib)
    for w in which:
        jobs[w]()
