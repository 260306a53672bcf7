#!/usr/bin/env python3
"""bench.py - contact-pairs/sec through one full spline pass (K1 classify+histogram -> host fit -> K2 p-values ->
K3 Benjamini-Hochberg) on synthetic human contacts, with the inputs resident in HBM.

    python bench.py --gpus 1 --steps 5 --warmup 1
    python bench.py --gpus N --steps K --warmup W          (no launcher: the script starts its N ranks itself; fewer than N visible
                                                            GPUs -> one JSON diagnostic line with "value": null, exit code 2)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
           bench.py --gpus N --steps K --warmup W

One JSON line on stdout (rank 0) with the driver's contract fields plus `roofline` (dominant kernel, HIP-event timed),
`cpu_baseline` (the oracle timed on this box's host cores on a bounded sample of the same rows, N = 1 only) and
`parity_check` - the run that is timed is the run that is checked, at every N:
  N = 1: the engine's K1 histogram and its fit (bins, possible pairs, x, y, s, knots, table, N) against what the REAL reference's
         stage functions returned on this workload (fixtures tests/golden/f14_*), the p-values of the CPU sample recomputed by the
         oracle's Cephes from that table, and q of EVERY row against the oracle's Benjamini-Hochberg of the engine's p;
  N > 1: (and FHX_FORCE_DIST=1 on one GPU) after the timed steps rank 0 runs the whole genome alone on its GPU through the plain
         single-GPU path - checked as above - and a 64-bit order-free hash of (row, p) and of (row, q) per chromosome, summed over
         the ranks of the sharded run, must equal that run's: every p and q of the sharded pass is bit-identical to one GPU's.

Workloads (`--config`, BASELINE.json configs):
  C3 (default, the configuration the metric is quoted on): 22 hg19 autosomes at 5 kb (576 216 loci), -L 20000 -U 2000000
      (397 distance values), ~1.5e8 observed cis pairs, ICE-like bias table, -b 100, 1 pass, intraOnly.
  C2: one chromosome of 249 250 621 bp at 40 kb, no distance bounds, ~1e7 pairs, bias, -b 100, 2 passes (a step = both).
  C5: 22 autosomes at 1 kb (2 881 044 loci), -L 2000 -U 2000000, ~1.9e9 cis + 1e8 trans pairs, -x All, 1 pass
      (`--max-chroms k` takes the first k chromosomes; the trans rows scale with the loci).
With N > 1 GPUs the headline is STRONG scaling (BASELINE configs[3]: the one genome sharded by chromosome over the ranks,
distance histogram all-reduced, BH ranking global over RCCL); the weak-scaling figure (genome replicated N times, N x the
rows) is measured afterwards and reported under `weak_scaling` (`--no-weak` skips it, `--weak` makes it the headline).  The N > 1
line also carries `per_rank` (rows and kernel times of every rank), `single_gpu_ms_per_step` / `strong_speedup` / `strong_efficiency`
(against the one-GPU pass rank 0 times in the same run while it verifies) and `predicted_ms` (profiles/scaling_model.json).
At N = 1 on C3 a second, labelled workload follows the headline: `k3_stress` - the same genome with lognormal rate noise, where
12 % of the rows fall below the BH cutoff and K3's sort has work to do (`--no-k3-stress` skips it; it never enters `value`).
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
HEAVY_FP64_INSTR_PER_ITER = 23.0                                   # ISA count of cf_swapped_step's hot path (fhx_bdtrc.hpp)

CONFIGS = {
    "C2": dict(res=40000, lengths=[249250621], L=0, U=float("inf"), lo=0, hi=None, amp_lo=1, keep=0.52, passes=2,
               mode="intraOnly", trans_per_locus=0.0),
    "C3": dict(res=5000, lengths=None, L=20000, U=2000000, lo=4, hi=400, amp_lo=4, keep=0.66, passes=1, mode="intraOnly",
               trans_per_locus=0.0),
    # SURVEY 8(d)'s second C3 variant: no -U, 49 849 distance values (8x the LDS window of K1), ~1.0e9 rows - histogram + sort stress
    "C3w": dict(res=5000, lengths=None, L=20000, U=float("inf"), lo=4, hi=None, amp_lo=4, keep=0.05, passes=1, mode="intraOnly",
                trans_per_locus=0.0),
    "C5": dict(res=1000, lengths=None, L=2000, U=2000000, lo=2, hi=2000, amp_lo=2, keep=0.33, passes=1, mode="All",
               trans_per_locus=1.0e8 / 2881044),
}


def calibration():
    """reference / restatement calibration on the bundled hESC chr1 40 kb set (778 363 rows, 1 core), measured in the build
    container by tests/golden/make_golden.py calib (the real fithic.py next to the oracle on the same rows) -> dict."""
    with open(os.path.join(ROOT, "tests", "golden", "calibration.json")) as f:
        return json.load(f)


_T0 = time.time()


def log(*a):
    print("[%6.1f s]" % (time.time() - _T0), *a, file=sys.stderr, flush=True)


def build_rows(synth, torch, cfg, genome, mine, rank, world, device, overdispersion=0.0, hotspots=None):
    """This rank's contact rows as five int32 device columns (one allocation per column, filled chromosome by chromosome)."""
    n_chr = len(genome)
    hi = cfg["hi"]
    amp = synth.solve_amplitude(cfg["keep"], cfg["amp_lo"], hi if hi is not None else genome.n_loci[0] - 1)
    n_trans = int(round(cfg["trans_per_locus"] * sum(genome.n_loci))) if n_chr > 1 else 0
    t0, t1 = (n_trans * rank) // world, (n_trans * (rank + 1)) // world
    est = 0
    for c in mine:
        h = min(hi if hi is not None else genome.n_loci[c] - 1, genome.n_loci[c] - 1)
        width = max(h - cfg["lo"] + 1, 0)
        cand = genome.n_loci[c] * width - (width * (width + 1)) // 2 if hi is None else genome.n_loci[c] * width   # window cut by the end
        est += int(max(cand, 0) * min(1.0, cfg["keep"] * 1.08)) + 1024
    est += t1 - t0
    cols = [torch.empty(est, dtype=torch.int32, device=device) for _ in range(5)]
    n = 0
    for c in mine:
        part = synth.cis_contacts(genome, c, cfg["lo"], hi if hi is not None else genome.n_loci[c] - 1, amp, device=device,
                                  overdispersion=overdispersion, hotspots=hotspots)
        m = int(part[0].numel())
        if n + m > est:                                       # estimate too small: grow (rare)
            grow = [torch.empty(int((n + m) * 1.2), dtype=torch.int32, device=device) for _ in range(5)]
            for k in range(5):
                grow[k][:n] = cols[k][:n]
            cols, est = grow, int((n + m) * 1.2)
        for k in range(5):
            cols[k][n:n + m] = part[k]
        n += m
        del part
    n_cis = n
    if t1 > t0:
        part = synth.trans_contacts(genome, n_trans, t0, t1, device=device)
        m = int(part[0].numel())
        if n + m > est:
            grow = [torch.empty(n + m, dtype=torch.int32, device=device) for _ in range(5)]
            for k in range(5):
                grow[k][:n] = cols[k][:n]
            cols = grow
        for k in range(5):
            cols[k][n:n + m] = part[k]
        n += m
        del part
    return cols, n, n_cis, n_trans


def build_sample(torch, cols, n_local, n_cis_local, genome, mine, cfg, res, device, budget=3.0e7, chunk=1 << 27):
    """Bounded sample for the CPU legs: whole chromosomes, smallest first, up to ~3e7 cis rows (10-15 s of one core), plus every
    trans row between two sampled chromosomes; a chromosome that alone exceeds the budget (1 kb loci) is cut to its first loci -
    rows with both ends below the cut - so that the sample is a complete small genome.  The columns are walked in chunks of
    2^27 rows: at C5 (2e9 rows next to the engine's 158 GB) whole-column temporaries would not fit."""
    import numpy as np
    n_chr = len(genome)
    counts = torch.zeros(n_chr, dtype=torch.int64, device=device)
    for lo in range(0, n_cis_local, chunk):
        hi = min(n_cis_local, lo + chunk)
        counts += torch.bincount(cols[0][lo:hi].to(torch.int64), minlength=n_chr)
    counts = counts.cpu().numpy()
    order = sorted(mine, key=lambda c: counts[c])
    chosen, rows, cut_loci = [], 0, {}
    for c in order:
        if chosen and rows + counts[c] > budget:
            break
        chosen.append(c)
        rows += int(counts[c])
    sel_chr = torch.zeros(n_chr, dtype=torch.bool, device=device)
    sel_chr[torch.tensor(sorted(chosen), device=device)] = True
    cut = None
    if rows > 1.3 * budget:                           # one oversized chromosome: keep its first loci only
        c = chosen[0]
        cut = max(int(genome.n_loci[c] * budget / rows), min(genome.n_loci[c], (cfg["hi"] or 0) + 64))
        cut_loci[c] = cut
    idx_parts, col_parts = [], [[] for _ in range(5)]
    for lo in range(0, n_local, chunk):
        hi = min(n_local, lo + chunk)
        keep = sel_chr[cols[0][lo:hi].to(torch.int64)] & sel_chr[cols[2][lo:hi].to(torch.int64)]
        if cut is not None:
            keep &= (cols[1][lo:hi] < cut * res) & (cols[3][lo:hi] < cut * res)
        idx = torch.nonzero(keep).squeeze(1)
        if idx.numel():
            idx_parts.append((idx + lo).cpu().numpy())
            for k in range(5):
                col_parts[k].append(cols[k][lo:hi][idx].cpu().numpy())
        del keep, idx
    cat = lambda parts, dt: np.concatenate(parts) if parts else np.empty(0, dt)
    return {"rows": cat(idx_parts, np.int64), "cols": [cat(col_parts[k], np.int32) for k in range(5)], "chroms": sorted(chosen),
            "cut_loci": cut_loci}


def row_keys(torch, synth, cols, n):
    """64-bit identity of every row, independent of where the row sits: splitmix64 chained over (chr1, mid1, chr2, mid2, count)."""
    k = cols[0][:n].to(torch.int64)
    for c in cols[1:]:
        k = synth._splitmix64(torch, (k * 0x100000001B3) ^ c[:n].to(torch.int64))
    return k


def hash_table(torch, synth, keys, chr1, values, n_chr):
    """[n_chr] int64: per chromosome (of the row's first locus) the wrapping sum of splitmix64(key ^ splitmix64(bits of value))
    over the rows.  A sum is order-free and additive over shards: the table of a sharded run is the sum of its ranks' tables."""
    out = torch.zeros(n_chr, dtype=torch.int64, device=keys.device)
    if keys.numel():
        h = synth._splitmix64(torch, keys ^ synth._splitmix64(torch, values.view(torch.int64)))
        out.index_add_(0, chr1.to(torch.int64), h)
    return out


def result_hashes(torch, synth, eng, keys, chr1, n, n_chr):
    """[2, n_chr] int64: hash_table of the engine's p (row 0) and q (row 1), read from its device buffers."""
    out = torch.zeros((2, n_chr), dtype=torch.int64, device=keys.device)
    if n == 0:
        return out
    buf = torch.empty(n, dtype=torch.float64, device=keys.device)
    for which in (0, 1):
        eng.ctx.memcpy_d2d(buf.data_ptr(), eng.ctx.device_ptr(which), 8 * n)
        out[which] = hash_table(torch, synth, keys, chr1[:n], buf, n_chr)
    return out


class TorchComm:
    """The handful of torch.distributed calls bench.py itself needs (the pass's own collectives run inside the library)."""

    def __init__(self, td, torch, device):
        self.td, self.torch, self.device = td, torch, device
        self.rank, self.world = td.get_rank(), td.get_world_size()

    def barrier(self):
        self.td.barrier()

    def gather_rows(self, t):
        """all-gather of equally shaped tensors -> list, one per rank"""
        t = t.to(self.device).contiguous()
        out = [self.torch.empty_like(t) for _ in range(self.world)]
        self.td.all_gather(out, t)
        return out

    def max_float(self, v):
        t = self.torch.tensor([float(v)], dtype=self.torch.float64, device=self.device)
        return max(float(o.item()) for o in self.gather_rows(t))

    def sum_int(self, v):
        t = self.torch.tensor([int(v)], dtype=self.torch.int64, device=self.device)
        return sum(int(o.item()) for o in self.gather_rows(t))

    def gather_floats(self, vals):
        t = self.torch.tensor([float(v) for v in vals], dtype=self.torch.float64, device=self.device)
        return [o.cpu().tolist() for o in self.gather_rows(t)]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", choices=sorted(CONFIGS), default="C3", help="workload (BASELINE.json configs); C3 is the headline")
    ap.add_argument("--keep", type=float, default=0.0, help="fraction of candidate cis pairs observed (0 = the config's depth)")
    ap.add_argument("--strong", action="store_true", help="N > 1: shard one genome (the default headline)")
    ap.add_argument("--weak", action="store_true", help="N > 1: make the N-times replicated genome the headline instead")
    ap.add_argument("--no-weak", action="store_true", help="N > 1: skip the additional weak-scaling measurement")
    ap.add_argument("--overdispersion", type=float, default=0.0,
                    help="0 = synth-v1 (Poisson around the model); s > 0 adds lognormal rate noise: heavier small-p tail, like real maps")
    ap.add_argument("--hotspots", default="", metavar="PHI:M",
                    help="variant: a random fraction PHI of the pairs gets M times the model's rate, the rest less (mean kept) - the "
                         "contrast that puts a third to a half of a real map's rows below the BH cutoff")
    ap.add_argument("--no-k3-stress", action="store_true",
                    help="N = 1: skip the second workload (lognormal rate noise s = 1.0: a heavy small-p tail, where K3's sort works)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-parity-check", action="store_true")
    ap.add_argument("--max-chroms", type=int, default=0, help="use only the first k chromosomes")
    ap.add_argument("--shard-of", type=int, default=0,
                    help="scaling model: run, alone on this GPU, the LARGEST shard an N-way chromosome sharding of the workload gives "
                         "(the chromosomes synth.assign_chromosomes hands the fullest rank); profiles/scaling_model.py")
    ap.add_argument("--path", choices=["fithic", "kr", "cni"], default="fithic",
                    help="fithic (default): the headline pass.  kr / cni: the neighbouring steps (Knight-Ruiz bias vectors, merging of "
                         "nearby contacts) measured by profiles/kr_bench.py / profiles/cni_bench.py, plus their cpu_baseline")
    ap.add_argument("--totals", choices=["auto", "reference", "wide"], default="auto",
                    help="what bdtrc is given for a total of counts >= 2^31 (include/fithic_mi355x.h FHX_TOTALS_*): 'reference' narrows it to a "
                         "C int as scipy does under fithic.py (every in-range p-value of C3w and C5 is then nan - nothing to time); 'wide' uses "
                         "the true total.  auto = wide for C3w / C5, reference elsewhere (C2 / C3 totals are below 2^31: both are the same "
                         "function).  The line says which ran: totals_semantics")
    ap.add_argument("--no-bias", action="store_true", help="variant: no bias file (p depends on (distance, count) only: table path)")
    ap.add_argument("--replicas", type=int, default=0, help="debug: replicate the genome R times per run regardless of --gpus (size test)")
    args = ap.parse_args()

    # `python bench.py --gpus N` outside a launcher: this process starts the N ranks itself (torch.distributed.run on 127.0.0.1)
    # and passes their one JSON line through; with fewer than N visible GPUs the one line is a diagnostic instead
    if "RANK" not in os.environ and args.path == "fithic" and (args.gpus > 1 or os.environ.get("FHX_FORCE_DIST")):
        raise SystemExit(self_launch(args))

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
    from fithic_amd import synth
    from fithic_amd.engine import Engine

    world = world_all = int(os.environ.get("WORLD_SIZE", "1"))
    rank = rank_all = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and rank == 0:                     # the launcher's world is what runs; the line reports it as n_gpus
        log("[bench] --gpus %d but the launcher started %d rank(s): running with %d" % (args.gpus, world, world))
    # FHX_BENCH_TRANSPORT=pipes (tests only): the ranks share the GPUs that exist (rank r on device r % visible) and the library's
    # collectives go through fhx_comm_init_custom + sharded.PipeTransport (host-staged, what `fithic --gpus N` uses in its tests)
    # instead of RCCL, which refuses two ranks on one GPU; torch.distributed runs on gloo.  Everything else in the N > 1 branch
    # is the code an RCCL run executes.  The line of such a run carries "value": null: its timings say nothing about xGMI.
    pipes = transport_kind() == "pipes"
    n_vis = torch.cuda.device_count()
    if n_vis <= (0 if pipes else local_rank):
        diagnostic(json_fd, args, "rank %d finds %d visible GPU(s)" % (rank, n_vis), visible_devices=n_vis)
        raise SystemExit(2)
    dev_index = local_rank % n_vis if pipes else local_rank
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)
    comm = None
    rccl = None
    mesh = None
    if world > 1 or os.environ.get("FHX_FORCE_DIST"):      # FHX_FORCE_DIST=1: run the RCCL path with a single rank
        import datetime
        import torch.distributed as td
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        # the long timeout covers rank 0's single-GPU verification run, during which the others wait
        if pipes:
            from fithic_amd import sharded
            td.init_process_group("gloo", timeout=datetime.timedelta(minutes=45))
            comm = TorchComm(td, torch, torch.device("cpu"))
            mesh = sharded.socket_mesh(rank, world, os.environ.get("MASTER_PORT", "0"), comm.barrier)
            rccl = {"world": td.get_world_size(), "backend": td.get_backend(), "torch_rccl_version": None,
                    "transport": "pipes", "devices_shared": n_vis < world,
                    "driver": "library communicator: fhx_comm_init_custom (sharded.PipeTransport: collectives staged through host "
                              "memory over a socket mesh) + fhx_run_pass_distributed - NOT RCCL, timings are not xGMI timings"}
        else:
            td.init_process_group("nccl", device_id=device, timeout=datetime.timedelta(minutes=45))      # nccl == RCCL on ROCm
            comm = TorchComm(td, torch, device)
            try:
                ver = ".".join(str(v) for v in torch.cuda.nccl.version())
            except Exception:
                ver = None
            rccl = {"world": td.get_world_size(), "backend": td.get_backend(), "torch_rccl_version": ver, "transport": "rccl",
                    "driver": "library communicator: fhx_comm_init + fhx_run_pass_distributed (collectives on the engine's stream)"}

    comm_all = comm
    cfg = dict(CONFIGS[args.config])
    cfg["totals"] = ("wide" if args.config in ("C3w", "C5") else "reference") if args.totals == "auto" else args.totals
    if args.keep > 0:
        cfg["keep"] = args.keep
    res, L, U = cfg["res"], cfg["L"], cfg["U"]
    base_lengths = cfg["lengths"] if cfg["lengths"] is not None else synth.HG19_AUTOSOMES
    if args.max_chroms:
        base_lengths = base_lengths[:args.max_chroms]
    if args.shard_of > 1:
        g_all = synth.Genome(res, base_lengths)
        owner_all = synth.assign_chromosomes(g_all, args.shard_of)
        fullest = max(range(args.shard_of), key=lambda r: sum(g_all.n_loci[c] for c in range(len(g_all)) if owner_all[c] == r))
        base_lengths = [base_lengths[c] for c in range(len(g_all)) if owner_all[c] == fullest]
        del g_all

    def parse_hotspots(text):
        if not text:
            return None
        phi, m = (float(v) for v in text.split(":"))
        return phi, m

    def measure(replicas, with_cpu_leg, solo=False, want_hashes=False, steps=None, warmup=None, overdispersion=None, hotspots=None,
                sample_budget=3.0e7):
        """Generate, load, warm up, time `steps` steps.  Returns the result pieces of this workload.
        solo: this process alone takes the whole genome through the plain single-GPU path (the verification run of rank 0)."""
        world, rank, comm = (1, 0, None) if solo else (world_all, rank_all, comm_all)
        steps = args.steps if steps is None else steps
        warmup = args.warmup if warmup is None else warmup
        genome = synth.Genome(res, base_lengths, replicas=replicas)
        owner = synth.assign_chromosomes(genome, world)
        mine = [c for c in range(len(genome)) if owner[c] == rank]
        t_gen = time.time()
        cols, n_local, n_cis_local, n_trans = build_rows(synth, torch, cfg, genome, mine, rank, world, device,
                                                         args.overdispersion if overdispersion is None else overdispersion,
                                                         parse_hotspots(args.hotspots) if hotspots is None else (hotspots or None))
        torch.cuda.synchronize()
        log("[rank %d] generated %d rows (%d cis on %d chromosomes, %d of %d trans) in %.1f s" %
            (rank, n_local, n_cis_local, len(mine), n_local - n_cis_local, n_trans, time.time() - t_gen))
        eng = Engine(dev_index)
        eng.configure(res, L, U, n_bins=100, mapp_thres=1, mode=cfg["mode"], totals=cfg["totals"])
        eng.load_fragments(*genome.fragments(), genome.sort_rank())
        if not args.no_bias:
            eng.load_bias(*genome.bias_table())
        eng.load_contacts_device([t.data_ptr() for t in cols], n_local)
        sample = None
        if with_cpu_leg:
            # bounded sample for the CPU legs: whole chromosomes, smallest first, up to ~3e7 cis rows (10-15 s of one core),
            # plus every trans row between two sampled chromosomes; a chromosome that alone exceeds the budget (1 kb loci)
            # is cut to its first loci - rows with both ends below the cut - so that the sample is a complete small genome
            sample = build_sample(torch, cols, n_local, n_cis_local, genome, mine, cfg, res, device, budget=sample_budget)
        keys = chr1 = None
        if want_hashes:                                          # row identities for the sharded-vs-single comparison
            keys = row_keys(torch, synth, cols, n_local)
            chr1 = cols[0][:n_local].clone()
        del cols
        torch.cuda.empty_cache()

        runner = None
        if comm and pipes:                                       # tests: the caller-provided transport of `fithic --gpus N`
            from fithic_amd import sharded
            pt = sharded.PipeTransport(eng.ctx, rank, world, mesh)
            eng.ctx.comm_init_custom(pt.struct, rank, world)
            runner = NativeRunner(eng)
            runner.transport = pt                                # the callbacks live as long as the runner
        elif comm:                                               # the library's own RCCL communicator on the engine's stream
            import torch.distributed as td
            uid = [_capi_mod().comm_unique_id() if rank == 0 else None]
            td.broadcast_object_list(uid, src=0)                 # torch.distributed only carries the 128-byte id
            eng.ctx.comm_init(uid[0], rank, world)               # fails loudly: there is no second implementation to fall back to
            runner = NativeRunner(eng)
        passes = cfg["passes"]
        pass_ms = np.zeros(passes)

        def one_step(timed=False):
            info = None
            for k in range(passes):
                t_p = time.perf_counter()
                if runner:
                    info = (runner.run().as_dict(), runner.stats.as_dict())
                else:
                    out = eng.run_pass(collect=False)
                    info = (out.info, out.stats)
                if k + 1 < passes:
                    (runner.next_pass if runner else eng.next_pass)()
                if timed and passes > 1:
                    eng.ctx.sync()
                    pass_ms[k] += 1e3 * (time.perf_counter() - t_p)
            if passes > 1:
                eng.reset_passes()
                if runner:
                    runner.reset()
            return info

        def barrier():
            if comm:
                comm.barrier()

        for _ in range(warmup):
            one_step()
        if runner:
            runner.timings.clear()
        kt = np.zeros(3)
        heavy = np.zeros(2)                                # seconds, rows of the dominant launch (the 300-iteration class)
        eng.ctx.kernel_seconds_total(reset=True)           # the library sums its HIP events per pass from here on
        barrier()
        torch.cuda.synchronize()
        # The interpreter's cyclic garbage collector stays out of the timed region: one full collection over this process's objects
        # (torch, numpy, the synthetic genome's tables) is a 50 ms pause in whatever Python statement triggers it - it hit one step in
        # thirty of a 2 ms pass (profiles/r06/forced_dist_step_trace.txt) and is no part of the library's pass.
        import gc
        gc.collect()
        gc.disable()
        t0 = time.perf_counter()
        info = None
        try:
            for _ in range(steps):
                info = one_step(timed=True)                # nothing but the K passes in the timed region: the events are read by the
            torch.cuda.synchronize()                       # library where it waits for its stream anyway (fhx_kernel_seconds_total)
            barrier()
            elapsed = time.perf_counter() - t0
        finally:
            gc.enable()
        if eng.call_seconds is not None and eng.call_seconds[4]:
            log("host seconds per pass inside pass_stats / fit / pvalues / bh (FHX_CALL_TIMES): " +
                " / ".join("%.1f us" % (1e6 * v / eng.call_seconds[4]) for v in eng.call_seconds[:4]))
        ev_s, ev_n = eng.ctx.kernel_seconds_total()        # seconds of K1, K2, K3 and the heavy launch over ev_n passes of the region
        per_pass = [ev_s[k] / ev_n[k] if ev_n[k] else 0.0 for k in range(4)]
        kt = np.array(per_pass[:3])
        heavy = np.array([per_pass[3], float(eng.ctx.k2_heavy_launch()[1])])      # rows of the launch: the last pass's
        try:
            heavy_ghz = float(eng.ctx.k2_heavy_clock())          # the shader clock that launch ran at on THIS box (a wave's own counters)
        except Exception:                                        # noqa: BLE001 - informational only
            heavy_ghz = 0.0
        if comm:
            elapsed = comm.max_float(elapsed)
            n_total = comm.sum_int(n_local)
        else:
            n_total = n_local
        stage_ms = None
        if runner:
            stage_ms = {k: 1e3 * v / max(steps * passes, 1) for k, v in runner.timings.items()}
            if rank == 0:
                log("[rank 0] host wall per pass of the distributed stages (ms): " + ", ".join("%s %.2f" % kv for kv in stage_ms.items()))
        mine_row = list(kt) + [float(n_local)] + list(heavy)
        k_all = comm.gather_floats(mine_row) if comm else [mine_row]
        hashes = None
        if want_hashes:
            if passes > 1:                                       # the timed steps end with reset_passes(): hash an (untimed) pass 1
                info = (runner.run().as_dict(), runner.stats.as_dict()) if runner else (lambda o: (o.info, o.stats))(eng.run_pass(collect=False))
            eng.ctx.sync()
            hashes = result_hashes(torch, synth, eng, keys, chr1, n_local, len(genome))
            del keys, chr1
        try:
            bh_sorted = int(eng.ctx.n_sorted())                  # rows below the exact BH cutoff = what K3 had to sort (last pass)
            if bh_sorted < 0:                                    # multi-pass steps end with reset_passes()
                bh_sorted = None
        except Exception:                                        # noqa: BLE001 - informational only
            bh_sorted = None
        try:
            sort_stats = eng.ctx.bh_sort_stats() if bh_sorted else None     # how K3 sorted them (passes, repair, fallback)
        except Exception:                                        # noqa: BLE001 - informational only
            sort_stats = None
        try:
            class_rows = eng.ctx.k2_class_rows()                 # rows per branch class of the last K2 (what each class kernel worked on)
        except Exception:                                        # noqa: BLE001 - informational only
            class_rows = None
        return dict(heavy_ghz=heavy_ghz, sort_stats=sort_stats, class_rows=class_rows, steps=steps, genome=genome, eng=eng, sample=sample, elapsed=elapsed, n_total=n_total, n_local=n_local, k_all=k_all, bh_sorted=bh_sorted,
                    stage_ms=stage_ms, pass_ms=pass_ms / max(steps, 1), info=info, n_trans=n_trans, replicas=replicas,
                    hashes=hashes, hashed_pass1=want_hashes and passes > 1)

    weak_headline = args.weak and world > 1
    replicas = args.replicas if args.replicas > 0 else (world if weak_headline else 1)
    # the f14 fixture of this workload (the real reference's fit on it) applies when the run IS that workload
    canonical = (args.keep == 0 and args.max_chroms == 0 and args.shard_of <= 1 and args.overdispersion == 0 and not args.hotspots and replicas == 1)
    fixture_name = args.config if canonical and os.path.exists(os.path.join(ROOT, "tests", "golden", "f14_%s_fit.npz" % args.config)) else None
    verify_sharded = comm is not None and not args.no_parity_check         # N > 1 (or FHX_FORCE_DIST): compare with one GPU
    want_digest = bool(os.environ.get("FHX_BENCH_HASH"))          # A/B of kernel variants: a digest of every p and q in the result line
    M = measure(replicas, with_cpu_leg=(rank == 0 and comm is None and not (args.no_cpu_baseline and args.no_parity_check)),
                want_hashes=verify_sharded or want_digest)
    eng, genome = M["eng"], M["genome"]
    passes = cfg["passes"]

    def checked(eng_, genome_, sample_, info_, fixture):
        """run_check on a finished pass; a failure of the checking leg is reported, never raised (the GPU line must not be lost)"""
        try:
            from oracle import run_check
            return run_check.check_engine_run(eng_, genome_, sample_, cfg, info_, not args.no_bias, p_stride=4, fit_fixture_name=fixture,
                                              torch=torch)            # p and q streamed in 1.3e8-row chunks: C5's 2e9 rows as well
        except Exception as e:
            log("parity_check failed: %r" % (e,))
            return {"ok": False, "error": repr(e)}

    result = None
    if rank == 0:
        elapsed, n_total = M["elapsed"], M["n_total"]
        ms = 1000.0 * elapsed / args.steps
        value = n_total * passes * args.steps / elapsed
        # dominant kernel = the K2 launch over the rows whose continued fraction runs to Cephes' 300-iteration cap
        worst = max(M["k_all"], key=lambda r: r[4])
        hv_s, hv_rows = worst[4], worst[5]
        achieved = ALGO_BYTES_K2 * hv_rows / hv_s / 1e9 if hv_s > 0 else 0.0
        # HBM bytes of the dominant launch: PMC counters cannot be read from inside this process (rocprofv3 wraps the command), so
        # the per-row figure of the last counter pass over THIS command is scaled by this run's rows; the source is named beside it
        traffic = traffic_source = None
        prof = os.path.join(ROOT, "profiles", "k2_pmc_traffic.json")
        if os.path.exists(prof):
            try:
                tj = json.load(open(prof))
                traffic = tj.get("hbm_bytes_per_heavy_row") * hv_rows
                traffic_source = tj.get("source")
            except Exception:
                traffic = None
        # the bound that matters: every VALU wave-instruction of the pass at one per four cycles and SIMD (counters of the same
        # command on this tree, profiles/valu_floor.json made by profiles/update_valu_floor.py; scaled by this run's rows)
        # The file holds the COUNT (wave-instructions per pass); the clock is this run's own - k2h_heavy samples it (fhx_k2_heavy_clock) -
        # so that pass_over_floor means the same thing on a 2.15 GHz and on a 2.24 GHz box.
        floor_ms = floor_src = None
        clock_ghz = M.get("heavy_ghz") or None
        try:
            vf = json.load(open(os.path.join(ROOT, "profiles", "valu_floor.json"))).get(args.config)
            if vf and world == 1:
                ghz = clock_ghz or vf["clock_ghz"]
                floor_ms = (vf["valu_wave_instructions_per_pass"] * vf["cycles_per_wave_instruction"] / (vf["simds"] * ghz * 1e9) * 1e3
                            * n_total / vf["pairs"])
                floor_src = vf["source"] + ("; clock of this run's heavy launch" if clock_ghz else "; clock of the counter run (none measured here)")
        except Exception:                                        # noqa: BLE001 - informational
            floor_ms = None
        fp64_instr = hv_rows * 300.0 * HEAVY_FP64_INSTR_PER_ITER
        fp64_issue_peak = 256 * 4 * 16 * 2.4e9          # CUs x SIMDs x fp64 lanes/clk x Hz  (= 78.6 TFLOP/s / 2)
        n_chr = len(base_lengths)
        genome_loci = [-(-int(v) // res) for v in base_lengths]
        desc = {"C2": "C2-synth: one %d bp chromosome @%d bp, no distance bounds, %d cis pairs, bias, -b 100, 2 passes, intraOnly",
                "C3": "C3-synth: %d x hg19 %d autosomes @%d bp, -L %d -U %d, %d cis pairs, ICE-like bias, -b 100, 1 pass, intraOnly",
                "C5": "C5-synth: %d x hg19 %d autosomes @%d bp, -L %d -U %d, %d cis + %d trans pairs, ICE-like bias, -b 100, 1 pass, -x All"}
        if args.config == "C2":
            workload = desc["C2"] % (base_lengths[0], res, n_total)
        elif args.config == "C3":
            workload = desc["C3"] % (replicas, n_chr, res, L, U, n_total)
        elif args.config == "C3w":
            workload = ("C3w-synth: %d x hg19 %d autosomes @%d bp, -L %d and no -U (%d distance values), %d cis pairs, ICE-like bias, "
                        "-b 100, 1 pass, intraOnly" % (replicas, n_chr, res, L, max(genome_loci) - 1 - L // res + 1, n_total))
        else:
            workload = desc["C5"] % (replicas, n_chr, res, L, U, n_total - M["n_trans"], M["n_trans"])
        result = {
            "metric": METRIC,
            "value": value, "unit": "contact-pairs/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms, "higher_is_better": True,
            # the headline shards ONE genome over the ranks: total work is fixed as N grows (at N = 1 it is the whole genome)
            "scaling": "weak" if weak_headline else "strong",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": workload, "name": args.config, "pairs": n_total, "resolution": res,
                       "generator": "synth-v1" + ("" if args.overdispersion == 0 else " + lognormal rate noise s=%g" % args.overdispersion) +
                                    ("" if not args.hotspots else " + hotspots phi:m=%s" % args.hotspots),
                       "parallelism": "chromosome-sharded x%d" % world, "passes": passes, "mode": cfg["mode"],
                       "bias": not args.no_bias},
            "roofline": {"bound": "hbm", "binding_resource": "fp64_valu_issue", "kernel": "k2h_heavy (swapped incbcf, 300 iterations)",
                         "achieved": achieved, "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_source,
                         "note": "HBM is the designated roofline of the path; the dominant launch (rows whose Cephes continued "
                                 "fraction runs all 300 iterations) is bound by fp64 VALU issue, see fp64_valu_issue_frac: "
                                 "algorithmic bytes = 20 B/row (12 read + 8 written), %g fp64 instructions per row-iteration" % HEAVY_FP64_INSTR_PER_ITER,
                         "valu_issue_floor_ms": floor_ms, "pass_over_floor": (ms / floor_ms) if floor_ms else None,
                         "valu_issue_floor_source": floor_src, "shader_clock_ghz": clock_ghz,
                         "launch_seconds": hv_s, "rows_per_launch": hv_rows,
                         "fp64_valu_issue_frac": (fp64_instr / hv_s) / fp64_issue_peak if hv_s > 0 else None,
                         "fp64_note": "loop instructions only, against the nominal 2.4 GHz; by the SQ counters (profiles/r06/z_counters.txt: 3.176e9 VALU "
                                      "wave-instructions per launch = 25.4 per row-iteration all told, GRBM_GUI_ACTIVE 1.026e8 / 8 XCDs = 1.28e7 cycles) the "
                                      "launch fills 97 % of the VALU issue slots (one wave instruction per 4 cycles and SIMD) at the clock it runs at "
                                      "(shader_clock_ghz: this run's)"},
            "kernels_ms": {"k1_classify_hist": 1e3 * worst[0], "k2_pvalue": 1e3 * worst[1], "k3_bh_sort_scan": 1e3 * worst[2]},
            "bh_rows_sorted_rank0": M.get("bh_sorted"),
            "bh_sort_rank0": M.get("sort_stats"),
            "k2_class_rows_rank0": M.get("class_rows"),
            "whole_pass_hbm_frac": (ALGO_BYTES_K1 + ALGO_BYTES_K2 + ALGO_BYTES_K3) * value / (world * HBM_PEAK_GBS * 1e9),
            "totals_semantics": totals_semantics(cfg["totals"], M["info"][0], M["info"][1]),
        }
        if passes > 1:
            result["ms_per_pass"] = [float(v) for v in M["pass_ms"]]
        if pipes and rccl:
            result["value_over_pipes"], result["value"] = result["value"], None
            result["value_note"] = ("FHX_BENCH_TRANSPORT=pipes: %d rank(s) on %d GPU(s), collectives staged through host memory - a test of "
                                    "bench.py's N > 1 branch, not a measurement; every time in this line is labelled by rccl.transport" % (world, n_vis))
        if rccl:
            r_, w_, v_ = eng.ctx.comm_info()
            rccl.update(world_in_library=w_, library_rccl_version_code=v_)
            result["rccl"] = rccl
            result["stage_ms"] = M["stage_ms"]
            # what every rank held and how long its three kernel groups ran (HIP events on its own stream, mean per step)
            result["per_rank"] = [{"rank": i, "rows": int(r[3]), "k1_ms": 1e3 * r[0], "k2_ms": 1e3 * r[1], "k3_ms": 1e3 * r[2],
                                   "heavy_ms": 1e3 * r[4], "heavy_rows": int(r[5])} for i, r in enumerate(M["k_all"])]
            if not weak_headline:
                result["predicted_ms"] = predicted_pass_ms(world, max(int(r[3]) for r in M["k_all"]), args.config)
        if args.no_parity_check:
            result["parity_check"] = {"ok": None, "skipped": "--no-parity-check"}
        if want_digest and M["hashes"] is not None:
            import hashlib
            result["result_digest"] = hashlib.sha256(M["hashes"].cpu().numpy().tobytes()).hexdigest()[:16]
        if M["sample"] is not None:
            if not args.no_parity_check:
                if passes > 1:                              # the timed steps end with reset_passes(): check an (untimed) pass 1
                    out1 = eng.run_pass(collect=False)
                    M["info"] = (out1.info, out1.stats)
                result["parity_check"] = checked(eng, genome, M["sample"], M["info"], fixture_name)
                log("headline checked (%.1f s)" % result["parity_check"].get("seconds", -1))
            if not args.no_cpu_baseline:
                try:
                    result["cpu_baseline"] = cpu_baseline(genome, M["sample"], cfg, not args.no_bias)
                    log("cpu_baseline (one core) done")
                    try:
                        result["cpu_baseline"]["all_cores"] = cpu_baseline_all_cores(genome, M["sample"], cfg, not args.no_bias)
                    except Exception as e:                       # noqa: BLE001 - informational leg
                        log("cpu_baseline all_cores failed: %r" % (e,))
                        result["cpu_baseline"]["all_cores"] = {"value": None, "error": repr(e)}
                except Exception as e:                           # the GPU line must not be lost to a problem of the CPU leg
                    log("cpu_baseline failed: %r" % (e,))
                    result["cpu_baseline"] = {"value": None, "unit": "contact-pairs/s", "cores": 1, "kind": "port", "sample": "failed: %r" % (e,)}
    sharded_hashes, sharded_info = M["hashes"], M["info"]
    eng.close()
    del M
    torch.cuda.empty_cache()
    if (rank == 0 and comm is None and args.path == "fithic" and args.overdispersion == 0 and not args.hotspots and not args.no_k3_stress
            and args.config == "C3"):
        # K3 where it works: on synth-v1 the exact cutoff leaves < 0.1 % of the rows to sort.  Three more, labelled workloads on the
        # same genome put 12 %, 39 % and 55 % of the rows below the BH cutoff - real (over-dispersed, domain-structured) maps sit
        # at 27-53 % - each measured AND checked like the headline (p of sampled rows against the oracle's Cephes, q of
        # every row against the oracle's BH of the engine's p).  Never the headline.
        result["k3_stress"] = []
        for label, od, hot in (("lognormal rate noise s = 1.0", 1.0, None), ("hotspots phi:m = 0.2:4.5", 0.0, (0.2, 4.5)),
                               ("hotspots phi:m = 0.25:3.9", 0.0, (0.25, 3.9))):
            try:
                S = measure(replicas, with_cpu_leg=not args.no_parity_check, steps=max(2, min(args.steps, 10)), warmup=min(max(args.warmup, 1), 3), overdispersion=od,
                            hotspots=hot if hot else (), sample_budget=1.0e7)
                k3_s = max(r[2] for r in S["k_all"])
                entry = {
                    "workload": "the same genome and depth, synth-v1 + " + label,
                    "pairs": int(S["n_total"]), "rows_sorted": S["bh_sorted"],
                    "survivor_fraction": (S["bh_sorted"] / S["n_total"]) if S["bh_sorted"] else None, "k3_ms": 1e3 * k3_s,
                    "sorted_keys_per_s": (S["bh_sorted"] / k3_s) if (S["bh_sorted"] and k3_s > 0) else None,
                    "sort": S["sort_stats"],
                    "ms_per_step": 1e3 * S["elapsed"] / S["steps"], "pairs_per_s": S["n_total"] * passes * S["steps"] / S["elapsed"],
                    "kernels_ms": {"k1_classify_hist": 1e3 * max(r[0] for r in S["k_all"]), "k2_pvalue": 1e3 * max(r[1] for r in S["k_all"]),
                                   "k3_bh_sort_scan": 1e3 * k3_s},
                    "k3_hbm_frac": (ALGO_BYTES_K3 * S["n_total"] / k3_s) / (HBM_PEAK_GBS * 1e9) if k3_s > 0 else None}
                if S["sample"] is not None:
                    entry["parity_check"] = checked(S["eng"], S["genome"], S["sample"], S["info"], None)
                else:
                    entry["parity_check"] = {"ok": None, "skipped": "--no-parity-check"}
                result["k3_stress"].append(entry)
                log("k3_stress: %s measured%s" % (label, " and checked (%.1f s)" % entry["parity_check"].get("seconds", -1) if S["sample"] is not None else ""))
                S["eng"].close()
                del S
            except Exception as e:                               # noqa: BLE001 - the headline line must not be lost to the extra workloads
                log("k3_stress (%s) failed: %r" % (label, e))
                result["k3_stress"].append({"workload": label, "error": repr(e)})
            torch.cuda.empty_cache()
    if verify_sharded:
        # Every p and q of the sharded pass against ONE GPU: the ranks' hash tables are summed (a chromosome's cis rows live on
        # one rank, the trans rows are spread: the sum is what one GPU holding everything computes); rank 0 then takes the whole
        # genome through the plain single-GPU path (no communicator), which is checked against the fixtures and the oracle.
        gathered = comm.gather_rows(sharded_hashes)
        if rank == 0:
            t_v = time.perf_counter()
            try:
                total = gathered[0].clone()
                for g_ in gathered[1:]:
                    total += g_.to(total.device)
                V = measure(replicas, with_cpu_leg=True, solo=True, want_hashes=True, steps=max(2, min(args.steps, 5)), warmup=1)
                same = (V["hashes"] == total.to(V["hashes"].device))     # (gloo gathers on the host)
                chk = checked(V["eng"], V["genome"], V["sample"], V["info"], fixture_name)
                n_bad_p, n_bad_q = int((~same[0]).sum()), int((~same[1]).sum())
                names = V["genome"].names
                st_s, st_1 = sharded_info[1], V["info"][1]
                stats_equal = all(int(st_s[k]) == int(st_1[k]) for k in ("inter_count", "inter_sum", "intra_all_sum", "in_range_sum", "max_count"))
                fit_equal = all(sharded_info[0][k] == V["info"][0][k] for k in ("bh_total_tests", "spline_s", "spline_fp", "residual", "n_table", "inter_chr_prob"))
                result["parity_check"] = dict(
                    chk, ranks=world, rows=int(V["n_total"]), sharded_equals_single_gpu=bool(n_bad_p == 0 and n_bad_q == 0 and stats_equal and fit_equal),
                    chromosomes_hashed=len(names), chromosomes_p_differ=[names[i] for i in torch.nonzero(~same[0]).flatten().tolist()],
                    chromosomes_q_differ=[names[i] for i in torch.nonzero(~same[1]).flatten().tolist()],
                    global_stats_equal=stats_equal, fit_scalars_equal=fit_equal,
                    single_gpu_ms_per_pass=1e3 * V["elapsed"] / max(passes * V["steps"], 1), seconds_all=time.perf_counter() - t_v,
                    sharded_how="64-bit order-free hash of (row identity, p bits) and (row identity, q bits) per chromosome: sum over the %d "
                                "rank(s) of the timed sharded run == the same table of ONE GPU holding all rows (plain single-GPU path, no "
                                "communicator); that single-GPU pass is the one checked against the reference fixtures and the oracle" % world)
                result["parity_check"]["ok"] = bool(chk.get("ok") and result["parity_check"]["sharded_equals_single_gpu"])
                if not weak_headline:
                    # strong scaling against ONE GPU timed in this very run: rank 0's verification pass over the whole genome
                    one_ms = result["parity_check"]["single_gpu_ms_per_pass"] * passes
                    result["single_gpu_ms_per_step"] = one_ms
                    result["strong_speedup"] = one_ms / result["ms_per_step"]
                    result["strong_efficiency"] = one_ms / (world * result["ms_per_step"])
                V["eng"].close()
                del V
            except Exception as e:
                log("sharded verification failed: %r" % (e,))
                result["parity_check"] = {"ok": False, "error": repr(e)}
            torch.cuda.empty_cache()
        comm.barrier()
    if world > 1 and not weak_headline and not args.no_weak and args.replicas == 0:
        W = measure(world, with_cpu_leg=False)                  # the weak-scaling figure: genome replicated per GPU
        if rank == 0:
            result["weak_scaling"] = {"value": W["n_total"] * passes * args.steps / W["elapsed"], "ms_per_step": 1e3 * W["elapsed"] / args.steps,
                                      "pairs": W["n_total"], "replicas": world, "stage_ms": W["stage_ms"]}
        W["eng"].close()
    if comm:
        comm.barrier()
    if rank == 0:
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(result) + "\n").encode())
    if comm:
        import torch.distributed as td
        td.destroy_process_group()


METRIC = "contact-pairs/sec through spline+p-value+BH pass (5 kb cis, whole node)"


def diagnostic(fd, args, message, **extra):
    """The one JSON line of a run that cannot produce a value: same head as a result line, `value` null, the reason in `error`."""
    line = dict({"metric": METRIC, "value": None, "unit": "contact-pairs/s", "n_gpus": args.gpus, "steps": args.steps,
                 "warmup": args.warmup, "higher_is_better": True, "error": message}, **extra)
    os.write(fd, (json.dumps(line) + "\n").encode())


def transport_kind():
    """"rccl" (what a measurement uses) or "pipes" (FHX_BENCH_TRANSPORT=pipes: tests of the N > 1 branch on a box with one GPU)"""
    t = os.environ.get("FHX_BENCH_TRANSPORT", "rccl")
    if t not in ("rccl", "pipes"):
        raise SystemExit("FHX_BENCH_TRANSPORT must be rccl or pipes")
    return t


def visible_gpus():
    """GPUs this process could use, without creating a HIP context here (the ranks are separate processes)."""
    try:
        import torch
        return int(torch.cuda.device_count())
    except Exception:                                            # noqa: BLE001 - no torch / no driver: nothing visible
        return 0


def self_launch(args):
    """Start `--gpus N` ranks of this script (one per GPU, RCCL rendezvous on 127.0.0.1) and hand their single JSON line on.
    Returns the exit code.  Whatever happens, stdout carries exactly one JSON line: the ranks' result, or a diagnostic."""
    import socket
    import subprocess
    n_vis = visible_gpus()
    if n_vis < (1 if transport_kind() == "pipes" else args.gpus):        # pipes: the ranks share the GPUs there are
        diagnostic(1, args, "--gpus %d asked for, %d GPU(s) visible on this node: nothing was run" % (args.gpus, n_vis), visible_devices=n_vis)
        return 2
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"), FHX_BENCH_SELF_LAUNCHED="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    log("[bench] no launcher environment: starting %d rank(s) myself: %s" % (args.gpus, " ".join(cmd[1:8])))
    r = subprocess.run(cmd, stdout=subprocess.PIPE, env=env)
    lines = [ln for ln in r.stdout.decode(errors="replace").splitlines() if ln.strip().startswith("{")]
    if lines:
        os.write(1, (lines[-1] + "\n").encode())
        return r.returncode
    diagnostic(1, args, "the %d rank(s) started by bench.py ended with exit code %d without a result line (see stderr)" % (args.gpus, r.returncode),
               visible_devices=n_vis)
    return r.returncode or 1


def predicted_pass_ms(world, rows_largest_rank, config):
    """Pass time the one-GPU measurements predict for this sharding (profiles/scaling_model.json, made by profiles/stage_times.py:
    fixed per-pass cost + per-row cost of the largest rank + the BH exchange over xGMI); None when no model is committed for the
    workload.  Printed next to the measured ms_per_step so that the driver's scaling curve can be held against it."""
    path = os.path.join(ROOT, "profiles", "scaling_model.json")
    try:
        m = json.load(open(path)).get(config)
        if not m:
            return None
        ms = m["fixed_ms"] + m["dist_fixed_ms"] * (world > 1) + m["ns_per_row"] * 1e-6 * rows_largest_rank
        return {"ms": ms, "model": m, "rows_largest_rank": rows_largest_rank, "source": "profiles/scaling_model.json"}
    except Exception:                                            # noqa: BLE001 - informational
        return None


def _capi_mod():
    from fithic_amd import _capi
    return _capi


class NativeRunner:
    """fhx_run_pass_distributed / fhx_next_pass_distributed with the interface bench.py's step loop expects."""

    def __init__(self, eng):
        self.eng, self.timings, self.stats = eng, {}, None

    def run(self):
        t0 = time.perf_counter()
        info = self.eng.ctx.run_pass_distributed()
        t1 = time.perf_counter()
        self.stats = self.eng.ctx.stats()
        for k, v in self.eng.ctx.dist_stage_seconds().items():
            self.timings[k] = self.timings.get(k, 0.0) + v
        # (the wall time of the one C call next to the sum of its stages, and of the bookkeeping behind it: where a pass's host time goes)
        self.timings["whole_call"] = self.timings.get("whole_call", 0.0) + (t1 - t0)
        self.timings["bookkeeping"] = self.timings.get("bookkeeping", 0.0) + (time.perf_counter() - t1)
        if os.environ.get("FHX_STEP_TRACE"):
            sys.stderr.write("step trace: call %.3f ms, behind it %.3f ms\n" % (1e3 * (t1 - t0), 1e3 * (time.perf_counter() - t1)))
        return info

    def next_pass(self):
        return self.eng.ctx.next_pass_distributed()

    def reset(self):
        pass


def _oracle_tables(genome, chroms, res, with_bias, cut_loci=None):
    """Fragment rows and the bias dictionary of the sampled chromosomes, in the oracle's input form."""
    import numpy as np
    frags, bias_dic = [], {}
    for c in chroms:
        n_loci = (cut_loci or {}).get(c, genome.n_loci[c])
        mids = np.arange(n_loci, dtype=np.int64) * res + res // 2
        frags += [(genome.names[c], int(m), 1) for m in mids]
        if with_bias:
            b = genome.bias(c)[:n_loci]
            b = np.where((b < 0.5) | (b > 2.0), -1.0, b)
            bias_dic[genome.names[c]] = dict(zip(mids.tolist(), b.tolist()))
    return frags, (bias_dic if with_bias else 0)


def totals_semantics(mode, info, stats):
    """What bdtrc was given for the two totals of the timed pass (fhx_fit_info) and whether that is the reference's answer."""
    at = [w for b, w in ((1, "observedIntraInRangeSum"), (2, "observedInterAllSum")) if info.get("totals_narrowed", 0) & b]
    out = {"mode": mode, "observedIntraInRangeSum": int(stats["in_range_sum"]), "observedInterAllSum": int(stats["inter_sum"]),
           "bdtrc_n_intra": int(info.get("bdtrc_n_intra", stats["in_range_sum"])), "bdtrc_n_inter": int(info.get("bdtrc_n_inter", stats["inter_sum"])),
           "totals_at_or_above_2p31": at}
    if not at:
        out["note"] = "both totals fit a C int: reference and wide semantics are the same function, the output is the reference's"
    elif mode == "wide":
        out["note"] = ("%s beyond a C int: this run evaluates bdtrc on the TRUE total (FHX_TOTALS_WIDE). fithic.py itself would write nan / "
                       "wrong-n values there (scipy narrows n to 32 bits; fixtures tests/golden/f15_*); --totals reference reproduces that "
                       "bit for bit. parity_check uses the oracle's wide mode: HIP == oracle restatement, not HIP == reference" % " and ".join(at))
    else:
        out["note"] = ("%s beyond a C int: bdtrc is given the total narrowed to 32 bits exactly as scipy does under fithic.py "
                       "(FHX_TOTALS_REFERENCE) - the output is the reference's, nan where the narrowed n is below count - 1" % " and ".join(at))
    return out


def cpu_baseline(genome, sample, cfg, with_bias):
    """The oracle (plain C Cephes + numpy stage logic, 1 thread) on a bounded sample of the same rows, own fit on the sample."""
    import numpy as np
    from oracle import fithic_oracle as fo
    chr_ids = sample["chroms"]
    local = np.full(len(genome), -1, np.int32)
    local[chr_ids] = np.arange(len(chr_ids), dtype=np.int32)
    c1, m1, c2, m2, cnt = sample["cols"]
    pairs = fo.Pairs(local[c1], m1, local[c2], m2, cnt, [genome.names[c] for c in chr_ids])
    frags, bias_dic = _oracle_tables(genome, chr_ids, cfg["res"], with_bias, sample.get("cut_loci"))
    fo.build()
    t0 = time.perf_counter()
    fo.run(pairs, frags, None, cfg["res"], n_bins=100, passes=cfg["passes"], mode=cfg["mode"], L=cfg["L"], U=cfg["U"],
           bias_dic=bias_dic, totals=cfg.get("totals", "reference"))
    dt = time.perf_counter() - t0
    cal = calibration()
    cpu_model = None
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                cpu_model = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    return {"value": len(pairs) * cfg["passes"] / dt, "unit": "contact-pairs/s", "cores": 1, "kind": "port",
            "sample": "%d rows of %s%s (same synthetic rows, own genome-wide fit on the sample, %d pass(es)), %.1f s" %
                      (len(pairs), ",".join(genome.names[c] for c in chr_ids),
                       "".join(" [first %d loci of %s]" % (v, genome.names[k]) for k, v in (sample.get("cut_loci") or {}).items()),
                       cfg["passes"], dt),
            "cpu_model": cpu_model, "host_cores": os.cpu_count(), "implementation": "oracle (numpy stage logic + C Cephes bdtrc, 1 thread)",
            "calibration": dict(cal, reference_over_port=cal["reference_rows_per_s"] / cal["port_rows_per_s"],
                                estimated_reference_pairs_per_s_here=len(pairs) * cfg["passes"] / dt *
                                cal["reference_rows_per_s"] / cal["port_rows_per_s"])}


def _oracle_one_chromosome(job):
    """worker of cpu_baseline_all_cores: the oracle on the rows of one sampled chromosome (own fit on it) -> (rows, seconds)"""
    sys.path.insert(0, ROOT)
    from oracle import fithic_oracle as fo
    name, cols, frags, bias_dic, kw = job
    c1, m1, c2, m2, cnt = cols
    pairs = fo.Pairs(c1, m1, c2, m2, cnt, [name])
    fo.build()
    t0 = time.perf_counter()
    fo.run(pairs, frags, None, kw["res"], n_bins=100, passes=kw["passes"], mode=kw["mode"], L=kw["L"], U=kw["U"], bias_dic=bias_dic)
    return len(pairs), time.perf_counter() - t0


def cpu_baseline_all_cores(genome, sample, cfg, with_bias):
    """SURVEY 8(d)(ii), "for information": the oracle sharded by chromosome over the host cores - one process per sampled
    chromosome (its cis rows, a fit of its own), all at once.  -> dict for cpu_baseline["all_cores"]."""
    import multiprocessing as mp
    import numpy as np
    c1, m1, c2, m2, cnt = sample["cols"]
    jobs = []
    kw = dict(res=cfg["res"], passes=cfg["passes"], mode="intraOnly", L=cfg["L"], U=cfg["U"])
    for c in sample["chroms"]:
        sel = (c1 == c) & (c2 == c)
        if not sel.any():
            continue
        frags, bias_dic = _oracle_tables(genome, [c], cfg["res"], with_bias, sample.get("cut_loci"))
        z = np.zeros(int(sel.sum()), np.int32)
        jobs.append((genome.names[c], (z, m1[sel], z, m2[sel], cnt[sel]), frags, bias_dic, kw))
    if not jobs:
        return None
    t0 = time.perf_counter()
    with mp.get_context("spawn").Pool(min(len(jobs), os.cpu_count() or 1)) as pool:
        done = pool.map(_oracle_one_chromosome, jobs, chunksize=1)
    wall = time.perf_counter() - t0
    rows = sum(r for r, _ in done)
    busy = max(t for _, t in done)
    one_core = sum(r for r, _ in done) / sum(t for _, t in done)
    return {"value": rows * cfg["passes"] / busy, "unit": "contact-pairs/s", "cores": min(len(jobs), os.cpu_count() or 1),
            "host_cores": os.cpu_count(), "rows": rows, "seconds_slowest_worker": busy, "seconds_wall_with_process_start": wall,
            "per_core_in_this_leg": one_core * cfg["passes"],
            "projected_on_all_host_cores": one_core * cfg["passes"] * (os.cpu_count() or 1),
            "how": "one process per sampled chromosome (cis rows, own fit), all at once; value = rows / the slowest worker's oracle time; "
                   "the projection multiplies the per-core rate of this leg by the host's cores (the genome has 22 chromosomes: more "
                   "cores than that would need rows split within a chromosome, which the per-pair loop allows)"}


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
