#!/usr/bin/env python3
"""Wall time of every C-ABI call of one pass (each followed by a stream synchronisation), for a small and a large workload:
where the time between the kernels goes (launch gaps, host fit, small copies) - what limits strong scaling when a shard is small.

    python profiles/stage_times.py [--config C2|C3] [--max-chroms K] [--repeat 20]
"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="C3")
    ap.add_argument("--max-chroms", type=int, default=0)
    ap.add_argument("--repeat", type=int, default=20)
    args = ap.parse_args()
    import numpy as np
    import torch
    import bench
    from fithic_amd import synth
    from fithic_amd.engine import Engine
    cfg = dict(bench.CONFIGS[args.config])
    lengths = cfg["lengths"] if cfg["lengths"] is not None else synth.HG19_AUTOSOMES
    if args.max_chroms:
        lengths = lengths[-args.max_chroms:]            # the SMALLEST chromosomes: a small shard
    genome = synth.Genome(cfg["res"], lengths)
    dev = torch.device("cuda", 0)
    cols, n, _, _ = bench.build_rows(synth, torch, cfg, genome, list(range(len(genome))), 0, 1, dev)
    eng = Engine(0)
    eng.configure(cfg["res"], cfg["L"], cfg["U"], n_bins=100, mapp_thres=1, mode=cfg["mode"])
    eng.load_fragments(*genome.fragments(), genome.sort_rank())
    eng.load_bias(*genome.bias_table())
    eng.load_contacts_device([t.data_ptr() for t in cols], n)
    ctx = eng.ctx
    names = ["pass_stats", "fit", "pvalues", "bh", "whole pass, no sync inside"]
    acc = {k: [] for k in names}
    for it in range(args.repeat + 2):
        ctx.reset_passes()
        ctx.sync()
        t = [time.perf_counter()]
        ctx.pass_stats(); ctx.sync(); t.append(time.perf_counter())
        info = ctx.fit(); ctx.sync(); t.append(time.perf_counter())
        ctx.pvalues(); ctx.sync(); t.append(time.perf_counter())
        ctx.bh(info.bh_total_tests); ctx.sync(); t.append(time.perf_counter())
        ctx.reset_passes()
        ctx.sync()
        t0 = time.perf_counter()
        ctx.pass_stats(); info = ctx.fit(); ctx.pvalues(); ctx.bh(info.bh_total_tests); ctx.sync()
        whole = time.perf_counter() - t0
        if it >= 2:
            for k in range(4):
                acc[names[k]].append(t[k + 1] - t[k])
            acc[names[4]].append(whole)
    k1, k2, k3 = ctx.kernel_seconds()
    print("%s, %d chromosomes, %d rows: kernel stages (events) K1 %.3f K2 %.3f K3 %.3f ms" % (args.config, len(genome), n, k1 * 1e3, k2 * 1e3, k3 * 1e3))
    for k in names:
        v = np.array(acc[k]) * 1e3
        print("  %-28s median %.3f ms   min %.3f ms" % (k, np.median(v), v.min()))
    eng.close()


if __name__ == "__main__":
    main()
