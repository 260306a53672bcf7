#!/usr/bin/env python3
"""Measurement of the Knight-Ruiz path (fhx_kr_*) on the C3-synth contact map (22 hg19 autosomes at
5 kb, the same rows bench.py uses): assembly, row removal, balance; roofline of the dominant kernel kr_spmv from the
HIP-event time of the SpMV launches inside fhx_kr_balance.  One JSON line on stdout.

    python profiles/kr_bench.py [--max-chroms K]          (GPU numbers only)
    python bench.py --path kr                              (the same + cpu_baseline)
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
HBM_PEAK_GBS = 8000.0


def measure(max_chroms=0, perc=0.05):
    """-> (result dict, genome, host columns); the CPU baseline is bench.py's business (oracle/ is imported there only)."""
    import numpy as np
    import torch
    from fithic_amd import synth, _capi
    res, lo_idx, hi_idx = 5000, 4, 400
    device = torch.device("cuda", 0)
    lengths = synth.HG19_AUTOSOMES[:max_chroms] if max_chroms else None
    genome = synth.Genome(res, lengths)
    amp = synth.solve_amplitude(0.66, lo_idx, hi_idx)
    parts = [synth.cis_contacts(genome, c, lo_idx, hi_idx, amp, device=device) for c in range(len(genome))]
    cols = [torch.cat([p[k] for p in parts]).cpu().numpy() for k in range(5)]
    del parts
    torch.cuda.empty_cache()
    m = len(cols[0])
    f_chr, f_mid, _ = genome.fragments()
    kr = _capi.KrContext(0)
    kr.load_loci(f_chr, f_mid)
    t0 = time.perf_counter()
    kr.load_pairs(cols[0], cols[1], cols[2], cols[3], cols[4].astype(np.float64))
    t_asm = time.perf_counter() - t0
    n_full, nnz_full, _, _ = kr.shape()
    t0 = time.perf_counter()
    removed, val, _ = kr.remove_sparse(perc)
    t_rem = time.perf_counter() - t0
    t0 = time.perf_counter()
    x, info = kr.balance(1e-6)
    t_bal = time.perf_counter() - t0
    bias = kr.bias()
    n, nnz = info.n, info.nnz
    spmv = info.spmv_seconds / max(info.spmv_timed, 1)
    vb = info.value_bytes                               # 4 when the counts are exact in binary32 (they are: integers), else 8
    cell = 4.0 if vb == 2 else vb + 4.0                 # the 4-byte cell (16-bit count + 16-bit column offset), else value + 4-byte column
    algo = cell * nnz + (28.0 if vb == 2 else 24.0) * n + 8.0 * (n + 1)   # cells; in/out/epilogue vectors (+ the row's first column); indptr
    # isolated SpMV (plain epilogue), many repeats
    _, spmv_iso = kr.spmv(np.ones(n), which=1, repeats=50)          # uses the same value array as the balance did
    traffic = None
    tp = os.path.join(ROOT, "profiles", "kr_pmc_traffic.json")
    if os.path.exists(tp):
        traffic = json.load(open(tp))["hbm_bytes_per_algorithmic_byte"] * algo
    out = {
        "metric": "Knight-Ruiz balancing of the C3-synth 5 kb contact map (HiCKRy path)", "n_gpus": 1, "dtype": "f64",
        "config": {"workload": "C3-synth: %d chromosomes @%d bp, %d contact rows -> %d loci, %d stored cells (symmetric CSR)"
                               % (len(genome), res, m, n_full, nnz_full), "perc": perc},
        "removed_rows": int(len(removed)), "balanced": {"n": n, "nnz": nnz}, "outer_iterations": info.outer_iterations,
        "inner_iterations_last": info.inner_iterations, "matvecs": info.matvecs, "residual": info.residual,
        "seconds": {"assemble_incl_h2d": t_asm, "remove_sparse": t_rem, "balance": t_bal},
        "bias_mean": float(np.mean(bias[bias > 0])),
        "roofline": {"bound": "hbm", "kernel": "kr_spmv", "achieved": algo / spmv / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": algo / spmv / 1e9 / HBM_PEAK_GBS, "traffic": traffic, "launch_seconds": spmv,
                     "isolated_launch_seconds": spmv_iso, "isolated_frac": algo / spmv_iso / 1e9 / HBM_PEAK_GBS,
                     "algorithmic_bytes_per_launch": algo,
                     "value_bytes": vb,
                     "cell_bytes": cell,
                     "note": "cell_bytes per stored cell (4 = 16-bit count + 16-bit column offset in its row; else value_bytes + 4) + 32-36 B per row "
                             "(indptr, gathered input, output, epilogue, first column)"},
    }
    kr.close()
    return out, genome, cols


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--max-chroms", type=int, default=0)
    ap.add_argument("--perc", type=float, default=0.05)
    args = ap.parse_args()
    out, _, _ = measure(args.max_chroms, args.perc)
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
