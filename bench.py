#!/usr/bin/env python3
"""bench.py - contact-pairs/sec through one full spline pass (K1 classify+histogram -> host fit -> K2 p-values ->
K3 Benjamini-Hochberg) on synthetic 5 kb human cis contacts, with the inputs resident in HBM.

    python bench.py --gpus 1 --steps 5 --warmup 1
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
           bench.py --gpus N --steps K --warmup W

One JSON line on stdout (rank 0) with the driver's contract fields plus `roofline` (dominant kernel K2, HIP-event
timed) and `cpu_baseline` (the oracle timed on this box's host cores on a bounded sample of the same rows).

Workload (BASELINE.json configs[2], "C3"): 22 hg19 autosomes at 5 kb (576 216 loci), -L 20000 -U 2000000 (397
distance values), ~1.5e8 observed cis pairs, ICE-like bias table, -b 100, 1 pass, intraOnly.  With N GPUs the default
is weak scaling: the genome is replicated N times (chr1..chr22, chr1_r1.., N x 22 chromosomes), chromosomes are
sharded over the ranks by size, the distance histogram is all-reduced and the BH ranking is global over all N x 1.5e8
p-values (RCCL).  `--strong` keeps the single 22-chromosome genome and shards it instead (BASELINE configs[3]).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

ALGO_BYTES_K1, ALGO_BYTES_K2, ALGO_BYTES_K3 = 12, 20, 16          # per pair, SURVEY.md 8d (48 B in total)
HBM_PEAK_GBS = 8000.0                                              # MI355X_MICROARCH.md: 8.0 TB/s spec
FP64_VALU_PEAK_TFLOPS = 78.6


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--resolution", type=int, default=5000)
    ap.add_argument("--keep", type=float, default=0.66, help="fraction of candidate cis pairs observed (sets the depth)")
    ap.add_argument("--strong", action="store_true", help="shard one genome instead of replicating it per GPU")
    ap.add_argument("--overdispersion", type=float, default=0.0,
                    help="0 = synth-v1 (Poisson around the model); s > 0 adds lognormal rate noise: heavier small-p tail, like real maps")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--max-chroms", type=int, default=0, help="debug: use only the first k chromosomes")
    ap.add_argument("--path", choices=["fithic", "kr", "cni"], default="fithic",
                    help="fithic (default): the headline pass.  kr / cni: the neighbouring steps (Knight-Ruiz bias vectors, merging of "
                         "nearby contacts) measured by profiles/kr_bench.py / profiles/cni_bench.py, plus their cpu_baseline")
    ap.add_argument("--no-bias", action="store_true", help="variant: no bias file (p depends on (distance, count) only: table path)")
    ap.add_argument("--replicas", type=int, default=0, help="debug: replicate the genome R times per run regardless of --gpus (size test)")
    args = ap.parse_args()

    # stdout carries exactly ONE line (the JSON): libraries that print banners there (RCCL prints its version block to
    # stdout when the first communicator is created) are diverted to stderr for the whole run
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    if args.path != "fithic":
        sys.path.insert(0, os.path.join(ROOT, "profiles"))
        if args.path == "kr":
            import kr_bench
            out, genome, cols = kr_bench.measure(args.max_chroms)
            if not args.no_cpu_baseline:
                out["cpu_baseline"] = cpu_baseline_kr(genome, cols, 0.05)
        else:
            import cni_bench
            out, table = cni_bench.measure()
            if not args.no_cpu_baseline:
                out["cpu_baseline"] = cpu_baseline_cni(*table)
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(out) + "\n").encode())
        return

    import numpy as np
    import torch
    from fithic_amd import synth, dist
    from fithic_amd.engine import Engine

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node %d for --gpus %d" % (args.gpus, args.gpus))
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    comm = None
    if world > 1 or os.environ.get("FHX_FORCE_DIST"):      # FHX_FORCE_DIST=1: run the RCCL path with a single rank
        import torch.distributed as td
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        td.init_process_group("nccl", device_id=device)          # nccl == RCCL on ROCm
        comm = dist.Comm(td, device)

    res = args.resolution
    L, U = 4 * res, 400 * res
    lo_idx, hi_idx = 4, 400
    lengths = synth.HG19_AUTOSOMES[:args.max_chroms] if args.max_chroms else None
    replicas = args.replicas if args.replicas > 0 else (1 if (args.strong or world == 1) else world)
    genome = synth.Genome(res, lengths, replicas=replicas)
    amp = synth.solve_amplitude(args.keep, lo_idx, hi_idx)
    owner = synth.assign_chromosomes(genome, world)
    mine = [c for c in range(len(genome)) if owner[c] == rank]

    t_gen = time.time()
    parts = [synth.cis_contacts(genome, c, lo_idx, hi_idx, amp, device=device, overdispersion=args.overdispersion) for c in mine]
    cols = [torch.cat([p[k] for p in parts]).contiguous() for k in range(5)]
    n_local = int(cols[0].numel())
    torch.cuda.synchronize()
    log("[rank %d] generated %d rows on %d chromosomes in %.1f s" % (rank, n_local, len(mine), time.time() - t_gen))

    eng = Engine(local_rank)
    eng.configure(res, L, U, n_bins=100, mapp_thres=1, mode="intraOnly")
    eng.load_fragments(*genome.fragments(), genome.sort_rank())
    if not args.no_bias:
        eng.load_bias(*genome.bias_table())
    eng.load_contacts_device([t.data_ptr() for t in cols], n_local)
    sample_cols = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:      # the CPU leg runs at N = 1 only
        # bounded sample for the CPU oracle: whole chromosomes, smallest first, up to ~3e7 rows (10-15 s of one core)
        by_size = sorted(range(len(mine)), key=lambda j: parts[j][0].numel())
        sample_idx, rows = [], 0
        for j in by_size:
            if sample_idx and rows + parts[j][0].numel() > 3.0e7:
                break
            sample_idx.append(j)
            rows += parts[j][0].numel()
        sample_cols = [[parts[j][k].cpu().numpy() for k in range(5)] for j in sorted(sample_idx)]
    del parts, cols
    torch.cuda.empty_cache()

    runner = dist.DistributedPass(eng, comm) if comm else None

    def one_step():
        if runner:
            return runner.run()
        eng.run_pass(collect=False)
        return None

    def barrier():
        if comm:
            comm.barrier()

    for _ in range(args.warmup):
        one_step()
    if runner:
        runner.timings.clear()
    kt = np.zeros(3)
    heavy = np.zeros(2)                                # seconds, rows of the dominant launch (k2_queue<swapped CF>)
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        one_step()
        kt += np.array(eng.kernel_seconds())          # HIP events on the engine's stream (syncs it)
        heavy += np.array(eng.ctx.k2_heavy_launch(), dtype=np.float64)
    torch.cuda.synchronize()
    barrier()
    elapsed = time.perf_counter() - t0
    if comm:
        elapsed = comm.max_float(elapsed)
        n_total = comm.sum_int(n_local)
    else:
        n_total = n_local
    if runner and rank == 0:
        log("[rank 0] host wall per pass of the distributed stages (ms): " +
            ", ".join("%s %.2f" % (k, 1e3 * v / max(args.steps, 1)) for k, v in runner.timings.items()))
    kt /= max(args.steps, 1)
    heavy /= max(args.steps, 1)
    mine_row = list(kt) + [float(n_local)] + list(heavy)
    k_all = comm.gather_floats(mine_row) if comm else [mine_row]

    result = None
    if rank == 0:
        ms = 1000.0 * elapsed / args.steps
        value = n_total * args.steps / elapsed
        # dominant kernel = the K2 launch over the rows whose continued fraction runs to Cephes' 300-iteration cap
        worst = max(k_all, key=lambda r: r[4])
        hv_s, hv_rows = worst[4], worst[5]
        achieved = ALGO_BYTES_K2 * hv_rows / hv_s / 1e9 if hv_s > 0 else 0.0
        traffic = None
        prof = os.path.join(ROOT, "profiles", "k2_pmc_traffic.json")
        if os.path.exists(prof):
            try:
                traffic = json.load(open(prof)).get("hbm_bytes_per_heavy_row") * hv_rows
            except Exception:
                traffic = None
        # fp64 view of the same launch: 300 iterations x 46 fp64 VALU instructions per row (ISA count of the hot path of the
        # one-Newton-step loop: 2 x 7 division, 6 numerator/denominator products, 8 recurrence, 11 tests, 7 counters, 4 masked)
        fp64_instr = hv_rows * 300.0 * 46.0
        fp64_issue_peak = 256 * 4 * 16 * 2.4e9          # CUs x SIMDs x fp64 lanes/clk x Hz  (= 78.6 TFLOP/s / 2)
        result = {
            "metric": "contact-pairs/sec through spline+p-value+BH pass (5 kb cis, whole node)",
            "value": value, "unit": "contact-pairs/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms, "higher_is_better": True, "scaling": "strong" if (args.strong and world > 1) else "weak",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "C3-synth: %d x hg19 22 autosomes @%d bp, -L %d -U %d, %d cis pairs, ICE-like bias, -b 100, "
                                   "1 pass, intraOnly" % (replicas, res, L, U, n_total),
                       "pairs": n_total, "resolution": res, "generator": "synth-v1" if args.overdispersion == 0 else
                       "synth-v1 + lognormal rate noise s=%g" % args.overdispersion, "parallelism": "chromosome-sharded x%d" % world,
                       "passes": 1},
            "roofline": {"bound": "hbm", "kernel": "k2_queue<BC_CF_SWAPPED>", "achieved": achieved, "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "note": "dominant launch = the rows whose Cephes continued fraction runs all 300 iterations; it is "
                                 "fp64-VALU-issue bound, not HBM bound: algorithmic bytes = 20 B/row (12 read + 8 written)",
                         "launch_seconds": hv_s, "rows_per_launch": hv_rows,
                         "fp64_valu_issue_frac": (fp64_instr / hv_s) / fp64_issue_peak if hv_s > 0 else None},
            "kernels_ms": {"k1_classify_hist": 1e3 * worst[0], "k2_pvalue": 1e3 * worst[1], "k3_bh_sort_scan": 1e3 * worst[2]},
            "whole_pass_hbm_frac": (ALGO_BYTES_K1 + ALGO_BYTES_K2 + ALGO_BYTES_K3) * value / (world * HBM_PEAK_GBS * 1e9),
        }
        if sample_cols:
            try:
                result["cpu_baseline"] = cpu_baseline(genome, sample_cols, res, L, U)
            except Exception as e:                           # the GPU line must not be lost to a problem of the CPU leg
                log("cpu_baseline failed: %r" % (e,))
                result["cpu_baseline"] = {"value": None, "unit": "contact-pairs/s", "cores": 1, "kind": "port", "sample": "failed: %r" % (e,)}
    if comm:
        comm.barrier()
    if rank == 0:
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(result) + "\n").encode())
    eng.close()
    if comm:
        import torch.distributed as td
        td.destroy_process_group()


def cpu_baseline(genome, sample_cols, res, L, U):
    """The oracle (plain C Cephes + numpy stage logic, 1 thread) on a bounded sample of the same rows."""
    import numpy as np
    from oracle import fithic_oracle as fo
    chr_ids = sorted({int(c[0][0]) for c in sample_cols if len(c[0])})
    names = {c: genome.names[c] for c in chr_ids}
    local = {c: i for i, c in enumerate(chr_ids)}
    cat = [np.concatenate([c[k] for c in sample_cols]) for k in range(5)]
    remap = np.vectorize(local.get)(cat[0]).astype(np.int32)
    pairs = fo.Pairs(remap, cat[1], remap, cat[3], cat[4], [names[c] for c in chr_ids])
    frags, bias_dic = [], {}
    for c in chr_ids:
        mids = np.arange(genome.n_loci[c], dtype=np.int64) * res + res // 2
        frags += [(names[c], int(m), 1) for m in mids]
        b = genome.bias(c)
        b = np.where((b < 0.5) | (b > 2.0), -1.0, b)
        bias_dic[names[c]] = dict(zip(mids.tolist(), b.tolist()))
    fo.build()
    t0 = time.perf_counter()
    fo.run(pairs, frags, None, res, n_bins=100, passes=1, mode="intraOnly", L=L, U=U, bias_dic=bias_dic)
    dt = time.perf_counter() - t0
    return {"value": len(pairs) / dt, "unit": "contact-pairs/s", "cores": 1, "kind": "port",
            "sample": "%d rows of %s (same synthetic rows, own genome-wide fit on the sample), %.1f s" %
                      (len(pairs), ",".join(names[c] for c in chr_ids), dt)}


def cpu_baseline_kr(genome, cols, perc):
    """Knight-Ruiz oracle (numpy + plain-C SpMV, 1 thread) on the rows of the four smallest chromosomes."""
    import numpy as np
    from oracle import hickry_oracle as ho
    small = np.argsort(np.array(genome.n_loci))[:4]
    sel = np.isin(cols[0], small)
    offs, n = {}, 0
    for c in sorted(small.tolist()):
        offs[c] = n
        n += int(genome.n_loci[c])
    res = genome.res
    base = np.vectorize(offs.get)(cols[0][sel]).astype(np.int64)
    x = base + (cols[1][sel].astype(np.int64) - res // 2) // res
    y = base + (cols[3][sel].astype(np.int64) - res // 2) // res
    z = cols[4][sel].astype(np.float64)
    ho.build()
    t0 = time.perf_counter()
    A = ho.assemble(x, y, z, n)
    removed, _, _ = ho.sparse_rows(A, perc)
    R = ho.drop(A, removed)
    xv, i, k = ho.knight_ruiz(R)
    dt = time.perf_counter() - t0
    return {"value": dt, "unit": "s (assemble + remove + balance)", "cores": 1, "kind": "port",
            "sample": "%d rows of the 4 smallest chromosomes -> %d loci, %d cells, %d outer iterations" % (int(sel.sum()), n, A.nnz, i)}


def cpu_baseline_cni(c, n1, n2, cc, p, q, res):
    """CombineNearbyInteraction oracle (pure Python, 1 thread) on the first 2e5 rows, text parse included."""
    import tempfile
    from oracle import combine_oracle as co
    k = min(200_000, len(c))
    path = os.path.join(tempfile.mkdtemp(), "s.txt")
    with open(path, "w") as f:
        f.write("h\n")
        for i in range(k):
            f.write("chr%d\t%d\tchr%d\t%d\t%d\t%e\t%e\n" % (c[i], n1[i] - res // 2, c[i], n2[i] - res // 2, cc[i], p[i], q[i]))
    t0 = time.perf_counter()
    lines = co.combine_lines(path, res)
    dt = time.perf_counter() - t0
    return {"value": k / dt, "unit": "rows/s", "cores": 1, "kind": "port",
            "sample": "the first %d rows (text parse included), %d lines out, %.1f s; the reference itself pairs all nodes of a "
                      "chromosome in Python (O(n^2))" % (k, len(lines), dt)}


if __name__ == "__main__":
    main()
