#!/usr/bin/env python3
"""End-to-end wall time of the drop-in CLI at scale: a synthetic 5 kb map of a few chromosomes written as the
reference's gz text files (contacts, fragments, bias), then `python -m fithic_amd` on it.  Prints stage times.

    python profiles/time_cli_scale.py [--chroms 3]
"""
import argparse
import gzip
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--chroms", type=int, default=2)
    args = ap.parse_args()
    import numpy as np
    import torch
    from fithic_amd import synth
    res = 5000
    genome = synth.Genome(res, synth.HG19_AUTOSOMES[16:16 + args.chroms])
    amp = synth.solve_amplitude(0.66, 4, 400)
    dev = torch.device("cuda", 0)
    parts = [synth.cis_contacts(genome, c, 4, 400, amp, device=dev) for c in range(len(genome))]
    cols = [torch.cat([p[k] for p in parts]).cpu().numpy() for k in range(5)]
    n = len(cols[0])
    out = "/tmp/cli_scale"
    os.makedirs(out, exist_ok=True)
    t0 = time.time()
    names = np.array(genome.names)
    import pandas as pd
    pd.DataFrame({0: names[cols[0]], 1: cols[1], 2: names[cols[2]], 3: cols[3], 4: cols[4]}).to_csv(
        out + "/contacts.gz", sep="\t", header=False, index=False, compression={"method": "gzip", "compresslevel": 1})
    f_chr, f_mid, f_hits = genome.fragments()
    pd.DataFrame({0: names[f_chr], 1: 0, 2: f_mid, 3: f_hits, 4: 1}).to_csv(out + "/frags.gz", sep="\t", header=False, index=False,
                                                                          compression="gzip")
    b_chr, b_mid, b_val = genome.bias_table()
    pd.DataFrame({0: names[b_chr], 1: b_mid, 2: b_val}).to_csv(out + "/bias.gz", sep="\t", header=False, index=False, compression="gzip")
    print("wrote %d contact rows (%.1f MB gz) in %.1f s" % (n, os.path.getsize(out + "/contacts.gz") / 1e6, time.time() - t0))
    for extra in (["-p", "1"], ["-p", "2"]):
        t0 = time.time()
        r = subprocess.run([sys.executable, "-m", "fithic_amd", "-i", out + "/contacts.gz", "-f", out + "/frags.gz", "-t", out + "/bias.gz",
                            "-o", out + "/run", "-r", str(res), "-L", "20000", "-U", "2000000"] + extra, cwd=ROOT, capture_output=True, text=True)
        dt = time.time() - t0
        tail = [ln for ln in r.stdout.splitlines() if "took" in ln or "Time" in ln]
        sig = out + "/run/FitHiC.spline_pass1.res%d.significances.txt.gz" % res
        print("fithic %s: wall %.2f s for %d rows (%.2f M rows/s end to end), output %.1f MB gz, cores %d, rc %d" %
              (" ".join(extra), dt, n, n / dt / 1e6, os.path.getsize(sig) / 1e6, os.cpu_count(), r.returncode))
        for ln in tail:
            print("    " + ln)


if __name__ == "__main__":
    main()
