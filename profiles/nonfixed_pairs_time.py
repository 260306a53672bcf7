#!/usr/bin/env python3
"""-r 0 possible-pair enumeration (fithic.py:691-778) at restriction-fragment scale: the host fit of libfithic_mi355x.so on a
synthetic HindIII-like genome (22 hg19 autosomes cut at irregular sites, mean fragment ~3.5 kb -> ~8e5 fragments), -L 10000
-U 2000000, 100 bins - with one host thread (the reference's visiting order, pair by pair) and with all of them (one
sequential chain per bin, integer slots in closed form).  Both must give the same bits.

    python profiles/nonfixed_pairs_time.py [--frag-bp 3500] [--upper 2000000]
"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frag-bp", type=float, default=3500.0)
    ap.add_argument("--upper", type=int, default=2000000)
    ap.add_argument("--lower", type=int, default=10000)
    args = ap.parse_args()
    from fithic_amd import _capi, synth
    rng = np.random.default_rng(5)
    chr_ids, mids = [], []
    for c, length in enumerate(synth.HG19_AUTOSOMES):
        cuts = np.unique(rng.integers(0, length, int(length / args.frag_bp)))
        m = ((cuts[:-1] + cuts[1:]) // 2).astype(np.int32)
        chr_ids.append(np.full(len(m), c, np.int32))
        mids.append(m)
    chr_ids, mids = np.concatenate(chr_ids), np.concatenate(mids)
    names = ["chr%d" % (i + 1) for i in range(22)]
    rank = np.argsort(np.argsort(names)).astype(np.int32)
    # observed distances: power-law contact counts over a grid of distinct distances in range (what K1 would hand over)
    keys = np.unique(rng.integers(args.lower, args.upper, 200000)).astype(np.int64)
    sumcc = np.maximum(1, (3e6 * (keys / 1e4) ** -1.08 * rng.random(len(keys))).astype(np.int64))
    st = _capi.FhxStats()
    st.in_range_sum = int(sumcc.sum())
    out = {}
    for threads in ("1", "0"):
        if threads == "0":
            os.environ.pop("FHX_THREADS", None)
        else:
            os.environ["FHX_THREADS"] = threads
        ctx = _capi.Context(-1)
        ctx.set_params(0, args.lower, args.upper, 100, 1, _capi.MODE_INTRA_ONLY)
        ctx.load_fragments(chr_ids, mids, np.ones(len(mids), np.int32), rank)
        ctx.set_dist_keys(keys)
        ctx.set_global_stats(st, sumcc, np.ones(len(keys), np.int64))
        t0 = time.perf_counter()
        info = ctx.fit()
        dt = time.perf_counter() - t0
        out[threads] = (dt, info.possible_intra_in_range, ctx.get_array(_capi.A_BIN_SUMDIST).tobytes(), ctx.get_array(_capi.A_BIN_POSS7).tobytes())
        print("%d fragments, %d possible in-range pairs, %s host thread(s): fit %.2f s = %.0f e6 pairs/s" %
              (len(mids), info.possible_intra_in_range, "1" if threads == "1" else "all (%d)" % (os.cpu_count() or 0), dt,
               info.possible_intra_in_range / dt / 1e6), flush=True)
        ctx.close()
    assert out["1"][1:] == out["0"][1:], "threaded enumeration differs from the single-thread walk"
    print("bin sums of distances and possible-pair counts identical bit for bit; speed-up %.1fx" % (out["1"][0] / out["0"][0]))
    # round 6: the same fit on a context with a GPU - the walk runs there (csrc/fhx_nfpairs.inc: chains as scans of parity maps)
    try:
        for rep in range(2):                                   # (the second call: buffers and code are warm)
            ctx = _capi.Context(0)
            ctx.set_params(0, args.lower, args.upper, 100, 1, _capi.MODE_INTRA_ONLY)
            ctx.load_fragments(chr_ids, mids, np.ones(len(mids), np.int32), rank)
            ctx.set_dist_keys(keys)
            ctx.set_global_stats(st, sumcc, np.ones(len(keys), np.int64))
            t0 = time.perf_counter()
            info = ctx.fit()
            dt = time.perf_counter() - t0
            got = (info.possible_intra_in_range, ctx.get_array(_capi.A_BIN_SUMDIST).tobytes(), ctx.get_array(_capi.A_BIN_POSS7).tobytes())
            print("GPU context, call %d: fit %.3f s = %.0f e6 pairs/s; %s the host walk bit for bit" %
                  (rep + 1, dt, info.possible_intra_in_range / dt / 1e6, "EQUALS" if got == out["1"][1:] else "DIFFERS FROM"), flush=True)
            ctx.close()
    except _capi.FhxError as e:
        print("no GPU context: %s" % e)


if __name__ == "__main__":
    main()
