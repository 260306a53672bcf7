#!/usr/bin/env python3
"""Distance histograms of the synth-v1 workloads at their FULL sizes (bench.py's C3, C3w, C5), computed with torch tensor
arithmetic only - the engine is not involved - chromosome by chromosome on whatever device is given (the GPU box: the
10^8..10^9 candidate pairs take minutes there).  The output (`synth_hist_<config>.npz`: per distance index the summed
contact count and the number of rows, the row total, and for C5 the trans row count / sum) is the `mainDic` the REAL
reference is then fed in the build container: tests/golden/make_golden.py f14 runs the reference's own
makeBinsFromInteractions -> generate_FragPairs -> calculateProbabilities -> fit_Spline on it and stores what they return.

    gpurun -- 'python tests/golden/dump_synth_hist.py gpurun_out/synth_hist C3 C3w C5'
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def main():
    import torch
    import bench
    from fithic_amd import synth
    out_dir = sys.argv[1]
    names = sys.argv[2:] or ["C3", "C3w", "C5"]
    os.makedirs(out_dir, exist_ok=True)
    dev = torch.device("cuda", 0) if torch.cuda.is_available() else torch.device("cpu")
    for name in names:
        cfg = dict(bench.CONFIGS[name])
        t0 = time.time()
        genome = synth.Genome(cfg["res"], cfg["lengths"])
        res, hi = cfg["res"], cfg["hi"]
        amp = synth.solve_amplitude(cfg["keep"], cfg["amp_lo"], hi if hi is not None else genome.n_loci[0] - 1)
        n_idx = max(genome.n_loci) + 1
        sumcc = torch.zeros(n_idx, dtype=torch.int64, device=dev)
        nrows = torch.zeros(n_idx, dtype=torch.int64, device=dev)
        rows_per_chr = []
        for c in range(len(genome)):
            part = synth.cis_contacts(genome, c, cfg["lo"], hi if hi is not None else genome.n_loci[c] - 1, amp, device=dev)
            d = ((part[3] - part[1]) // res).to(torch.int64)
            # float64 weights are exact here: per-chromosome sums stay far below 2^53
            sumcc += torch.bincount(d, weights=part[4].to(torch.float64), minlength=n_idx).to(torch.int64)
            nrows += torch.bincount(d, minlength=n_idx)
            rows_per_chr.append(int(d.numel()))
            del part, d
        n_trans = int(round(cfg["trans_per_locus"] * sum(genome.n_loci))) if len(genome) > 1 else 0
        inter_count = inter_sum = 0
        step = 1 << 26
        for a in range(0, n_trans, step):
            part = synth.trans_contacts(genome, n_trans, a, min(n_trans, a + step), device=dev)
            assert bool((part[0] != part[2]).all())
            inter_count += int(part[0].numel())
            inter_sum += int(part[4].to(torch.int64).sum())
            del part
        sumcc, nrows = sumcc.cpu().numpy(), nrows.cpu().numpy()
        keys = np.flatnonzero(nrows)
        path = os.path.join(out_dir, "synth_hist_%s.npz" % name)
        np.savez_compressed(path, dist_idx=keys.astype(np.int64), sumcc=sumcc[keys], nrows=nrows[keys],
                            rows_per_chr=np.array(rows_per_chr, np.int64), inter=np.array([inter_count, inter_sum], np.int64),
                            amplitude=np.array([amp]), torch_version=np.array([torch.__version__]), device=np.array([str(dev)]))
        print("%s: %d cis rows, %d distance values, sumCC %d, %d trans rows (sum %d), %.0f s -> %s" %
              (name, int(nrows.sum()), len(keys), int(sumcc.sum()), inter_count, inter_sum, time.time() - t0, path), flush=True)


if __name__ == "__main__":
    main()
