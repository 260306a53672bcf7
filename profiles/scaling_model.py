#!/usr/bin/env python3
"""Strong-scaling model of the headline workload from ONE GPU: for N = 1, 2, 4, 8 the largest shard of the N-way chromosome
sharding (what the slowest rank of an N-GPU run holds) is run alone - through the plain single-GPU path and through the
sharded schedule with one rank over real RCCL (FHX_FORCE_DIST=1: every collective is issued, none has a peer) - and a line
pass_ms = fixed_ms + ns_per_row * rows is fitted.  What one GPU cannot measure - the latency of the seven small collectives
with real peers and the all-to-all of the survivors' keys - enters as a stated allowance (ALLOWANCE_MS per pass).

    python profiles/scaling_model.py [--config C3] [--steps 40]   -> profiles/scaling_model.json + a table on stdout

bench.py reads the JSON and prints `predicted_ms` next to the measured ms_per_step of an N > 1 run."""
import argparse
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ALLOWANCE_MS = 0.20        # 7 small collectives with peers at ~25 us each + the all-to-all of < 1e5 keys: not measurable on one GPU


def run(config, n, steps, forced):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    if forced:
        env["FHX_FORCE_DIST"] = "1"
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--config", config, "--steps", str(steps), "--warmup", "5", "--no-cpu-baseline",
           "--no-parity-check", "--no-weak", "--no-k3-stress", "--shard-of", str(n)]
    r = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=1200)
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    if not line:
        raise RuntimeError(r.stderr[-2000:])
    return json.loads(line[-1])


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="C3")
    ap.add_argument("--steps", type=int, default=40)
    args = ap.parse_args()
    rows, plain, forced, kern = [], [], [], []
    for n in (1, 2, 4, 8):
        a = run(args.config, n, args.steps, False)
        b = run(args.config, n, args.steps, True)
        rows.append(a["config"]["pairs"])
        plain.append(a["ms_per_step"])
        forced.append(b["ms_per_step"])
        kern.append(a["kernels_ms"])
        print("largest shard of %d: %11d rows  plain %.3f ms  sharded schedule (1 rank, RCCL) %.3f ms  kernels K1 %.3f K2 %.3f K3 %.3f" %
              (n, rows[-1], plain[-1], forced[-1], kern[-1]["k1_classify_hist"], kern[-1]["k2_pvalue"], kern[-1]["k3_bh_sort_scan"]), flush=True)
    # least squares over the four points (forced-dist times: that is the code path of an N-GPU run)
    import numpy as np
    A = np.stack([np.ones(4), np.array(rows, float)], axis=1)
    (fixed, per_row), *_ = np.linalg.lstsq(A, np.array(forced), rcond=None)
    (fixed_p, per_row_p), *_ = np.linalg.lstsq(A, np.array(plain), rcond=None)
    model = {"fixed_ms": float(fixed_p), "dist_fixed_ms": float(fixed - fixed_p) + ALLOWANCE_MS, "ns_per_row": float(per_row) * 1e6,
             "allowance_ms": ALLOWANCE_MS, "points": {"shard_of": [1, 2, 4, 8], "rows": rows, "plain_ms": plain, "forced_dist_ms": forced},
             "source": "profiles/scaling_model.py on one MI355X: largest shard of an N-way sharding run alone, plain and under FHX_FORCE_DIST=1"}
    print("\npredicted strong scaling (pass ms = forced-dist time of the largest shard + %.2f ms allowance for collectives with peers):" % ALLOWANCE_MS)
    one = plain[0]
    for n, r_, f in zip((1, 2, 4, 8), rows, forced):
        ms = one if n == 1 else f + ALLOWANCE_MS
        print("  N = %d: %.3f ms  speed-up %.2f  efficiency %.2f" % (n, ms, one / ms, one / ms / n))
    path = os.path.join(ROOT, "profiles", "scaling_model.json")
    allm = json.load(open(path)) if os.path.exists(path) else {}
    allm[args.config] = model
    with open(path, "w") as f:
        json.dump(allm, f, indent=1)
    print("wrote", path)
    out = os.path.join(ROOT, "gpurun_out")                       # what travels back from the GPU box
    os.makedirs(out, exist_ok=True)
    with open(os.path.join(out, "scaling_model.json"), "w") as f:
        json.dump(allm, f, indent=1)


if __name__ == "__main__":
    main()
