"""world_size-2 test of the multi-GPU exchange logic (tests/dist_model.py, the model of the library's schedule in csrc/fhx_dist.inc) on CPU tensors over gloo.

The per-GPU compute of a distributed pass goes through `LocalOps`; here it is replaced by a checker-backed
implementation (oracle + numpy) so that what is exercised is exactly the product's exchange code: the packed
all-reduce of sums + distance histograms, the replicated host fit (the product's C++ host pass through a host-only
context), the sample-sort all-to-all of the BH keys, the rank offsets, the running-max carry and the way back.
Result: every rank's q-values equal the single-process reference values (golden fixture) bit for bit.
"""
import os
import sys
import socket

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, case, passes, result_dir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch
    import torch.distributed as td
    from conftest import load_case, case_args
    import dist_model as dist
    from fithic_amd import _capi, tables
    from fithic_amd.engine import MODES
    from oracle import fithic_oracle as fo

    td.init_process_group("gloo", rank=rank, world_size=world)
    comm = dist.Comm(td, torch.device("cpu"))
    meta, g = load_case(case)
    kw = case_args(meta)
    kw["passes"] = passes
    res = kw["resolution"]
    pairs = fo.read_contacts_file(kw["contacts"])
    ref = fo.run(pairs, kw["frags"], kw["bias_path"], res, kw["n_bins"], passes, kw["mode"], kw["L"], kw["U"], kw["mapp_thres"],
                 kw["tL"], kw["tU"])
    # shard rows by the chromosome of the first locus (round-robin over chromosome ids)
    mine = np.flatnonzero(pairs.chr1 % world == rank)
    local = fo.Pairs(pairs.chr1[mine], pairs.mid1[mine], pairs.chr2[mine], pairs.mid2[mine], pairs.count[mine], pairs.names)
    chroms = tables.ChromIndex()
    frag_tab = tables.read_fragments(kw["frags"], chroms)
    host = _capi.Context(-1)                  # the product's host stages, no GPU needed
    host.set_params(res, kw["L"], kw["U"], kw["n_bins"], kw["mapp_thres"], MODES[kw["mode"]], kw["tL"], kw["tU"])
    host.load_fragments(*frag_tab, chroms.sort_rank())

    class CheckerOps:
        """LocalOps with the compute done by the oracle / numpy (test double)."""

        def __init__(self):
            self.pass_idx = 0
            self.skip = np.zeros(len(local), bool)
            self.q = None
            self.out_hist = None
            self.n_out = 0

        nonfixed = res == 0
        out_list = None
        out_list_local = np.zeros(0, np.int64)

        def local_dist_keys(self):
            return self._keys

        def set_dist_keys(self, keys):
            host.set_dist_keys(keys)

        def set_outlier_dists(self, dists):
            self.out_list = np.asarray(dists, np.int64)

        def local_stats(self):
            keys, sumcc, icnt, isum, intra_all, rng_sum = fo.read_interactions(local, kw["L"], kw["U"], self.skip if self.pass_idx else None)
            if res == 0:                                    # -r 0: histograms aligned to this rank's distinct distances
                d = np.abs(local.mid1 - local.mid2)
                keep = ~self.skip if self.pass_idx else np.ones(len(local), bool)
                m = keep & (local.chr1 == local.chr2) & (d >= kw["L"]) & (d <= kw["U"])
                self._keys = np.asarray(keys, np.int64)
                hnp = np.zeros(len(self._keys), np.int64)
                np.add.at(hnp, np.searchsorted(self._keys, d[m]), 1)
                st = _capi.FhxStats()
                st.n_rows, st.inter_count, st.inter_sum, st.intra_all_sum, st.in_range_sum = len(local), icnt, isum, intra_all, rng_sum
                st.in_range_count = int(m.sum())
                st.max_count = int(local.count.max()) if len(local) else 0
                return st, np.asarray(sumcc, np.int64), hnp
            n_dist = int(max(np.abs(local.mid1 - local.mid2).max() // res + 2, 2))
            hcc = np.zeros(n_dist, np.int64)
            hnp = np.zeros(n_dist, np.int64)
            d = np.abs(local.mid1 - local.mid2)
            keep = ~self.skip if self.pass_idx else np.ones(len(local), bool)
            m = keep & (local.chr1 == local.chr2) & (d >= kw["L"]) & (d <= kw["U"])
            np.add.at(hcc, d[m] // res, local.count[m])
            np.add.at(hnp, d[m] // res, 1)
            st = _capi.FhxStats()
            st.n_rows, st.inter_count, st.inter_sum, st.intra_all_sum, st.in_range_sum = len(local), icnt, isum, intra_all, rng_sum
            st.in_range_count = int(m.sum())
            st.max_count = int(local.count.max())
            return st, hcc, hnp

        def set_global_and_fit(self, st, hist_cc, hist_np):
            self.global_sums = (st.inter_count, st.inter_sum, st.intra_all_sum, st.in_range_sum)
            host.set_global_stats(st, hist_cc, hist_np)
            if self.out_hist is not None:
                host.set_outlier_dist_hist(self.out_hist)
            if self.out_list is not None:
                host.set_outlier_dists(self.out_list)
            return host.fit()

        def pvalues(self):
            self.p = ref[self.pass_idx].p[mine]          # reference p of my rows (K2 itself is covered by the GPU tests)

        cutoff_bits = None

        def top_hist(self):
            h = np.zeros(8192, np.int64)
            keep = self.p < 1.0
            np.add.at(h, (np.abs(self.p[keep]).view(np.int64) >> 50), 1)
            return h

        def set_cutoff(self, global_hist, n_tests):
            cum = np.cumsum(global_hist)
            self.cutoff_bits = np.int64(0x3FF0000000000000)
            for b in np.flatnonzero(global_hist > 0):
                edge = np.array([int(b) << 50], np.int64).view(np.float64)[0]
                if edge * float(n_tests) / float(cum[b]) >= 1.0:
                    self.cutoff_bits = np.int64(int(b) << 50)
                    break

        def local_sorted_keys(self):
            keep = (self.p < 1.0) & (np.abs(np.nan_to_num(self.p, nan=2.0)).view(np.int64) < self.cutoff_bits)
            self.rows = np.flatnonzero(keep)
            bits = self.p[keep].view(np.int64)
            order = np.argsort(bits, kind="stable")
            self.rows = self.rows[order]
            self.q = np.where(np.isnan(self.p), np.nan, 1.0)
            return torch.from_numpy(bits[order].copy())

        def sort_keys(self, keys):
            k = keys.numpy()
            order = np.argsort(k, kind="stable")
            return torch.from_numpy(k[order].copy()), torch.from_numpy(order.astype(np.int32))

        def bh_segment(self, sorted_keys, rank0, carry, n_tests, want_q):
            pv = sorted_keys.numpy().view(np.float64)
            rank_ = rank0 + np.arange(1, len(pv) + 1, dtype=np.float64)
            bh = pv * float(n_tests) / rank_
            bh = np.where(bh > 1.0, 1.0, bh)
            run = np.maximum.accumulate(np.concatenate([[carry], bh]))[1:] if len(bh) else bh
            mx = float(run[-1]) if len(run) else float(carry)
            return (torch.from_numpy(run.copy()) if want_q else None), mx

        def scatter_q(self, q_sorted_local):
            self.q[self.rows] = q_sorted_local.numpy()

        def next_pass_local(self):
            r = ref[self.pass_idx]
            thr = r.outlier_thres
            with np.errstate(invalid="ignore"):
                out = self.p < thr
            self.skip |= out
            d = np.abs(local.mid1 - local.mid2)[out]
            if res == 0:
                self.out_list_local = np.sort(np.concatenate([self.out_list_local, d.astype(np.int64)]))
                self.n_out += int(out.sum())
                self.pass_idx += 1
                return self.n_out, self.out_list_local
            n_dist = int(max(np.abs(local.mid1 - local.mid2).max() // res + 2, 2))
            if self.out_hist_local is None:
                self.out_hist_local = np.zeros(n_dist, np.int64)
            np.add.at(self.out_hist_local, -(-d // res), 1)
            self.n_out += int(out.sum())
            self.pass_idx += 1
            return self.n_out, self.out_hist_local

        out_hist_local = None

        def set_outlier_hist(self, hist):
            self.out_hist = np.asarray(hist, np.int64)

        def get_skip_limit(self):
            return (1 << 63) - 1

        def set_skip_limit(self, limit):
            pass

    ops = CheckerOps()
    runner = dist.DistributedPass(None, comm, ops=ops)
    ok = True
    msgs = []
    for pi in range(passes):
        info = runner.run()
        r = ref[pi]
        if tuple(ops.global_sums) != tuple(r.sums):
            ok = False
            msgs.append("sums differ pass %d: %s vs %s" % (pi, ops.global_sums, r.sums))
        if info.bh_total_tests != r.N:
            ok = False
            msgs.append("N differs")
        want = r.q[mine]
        same = (ops.q.view(np.int64) == want.view(np.int64)) | (np.isnan(ops.q) & np.isnan(want))
        if not same.all():
            ok = False
            msgs.append("q differs on %d of %d rows in pass %d" % ((~same).sum(), len(same), pi))
        if pi + 1 < passes:
            total = runner.next_pass()
            if total != r.n_outlier_lines_total:
                ok = False
                msgs.append("outlier count %d vs %d" % (total, r.n_outlier_lines_total))
    # the collectives the model issued == the rows of csrc/fhx_dist_schedule.def, in order, with the table's sizes (the same table
    # the library's trace is held against in tests/test_gpu_native_dist.py): the model cannot drift from fhx_dist.inc unnoticed
    import dist_schedule
    try:
        s_per = dist.samples_per_rank(world)
        if res == 0:
            phases = ["LOAD", "STATS_NF", "BH"] + ["NEXT_NF", "STATS_NF", "BH"] * (passes - 1)
            envs = [dict(world=world, s=s_per, width=(1 if ph == "NEXT_NF" else 3)) for ph in phases]
        else:
            nd = runner.n_dist_global
            a, b = ops.hist_span(nd) if hasattr(ops, "hist_span") else (0, nd)
            phases = ["LOAD", "STATS", "BH"] + ["NEXT", "STATS", "BH"] * (passes - 1)
            envs = [dict(world=world, s=s_per, w=b - a, nd=nd) for _ in phases]
        dist_schedule.check(comm.trace, phases, envs)
    except AssertionError as e:
        ok = False
        msgs.append("schedule: %s" % e)
    with open(os.path.join(result_dir, "rank%d.txt" % rank), "w") as f:
        f.write("OK" if ok else "FAIL: " + "; ".join(msgs))
    td.destroy_process_group()


@pytest.mark.parametrize("case,passes", [("f2_all", 2), ("f6_quirk_all", 2), ("f2_inter", 1), ("f8_nonfixed_all", 2)])
def test_distributed_pass_world2_gloo(case, passes, tmp_path):
    torch = pytest.importorskip("torch")
    import torch.multiprocessing as mp
    port = _free_port()
    mp.spawn(_worker, args=(2, port, case, passes, str(tmp_path)), nprocs=2, join=True)
    for r in range(2):
        with open(os.path.join(str(tmp_path), "rank%d.txt" % r)) as f:
            assert f.read() == "OK"


def test_splitters_and_chromosome_assignment():
    torch = pytest.importorskip("torch")
    import dist_model as dist
    from fithic_amd import synth
    s = torch.arange(0, 1000, dtype=torch.int64)
    sp = dist.choose_splitters(torch, s, 4)
    assert sp.tolist() == [250, 500, 750]
    assert dist.choose_splitters(torch, s[:0], 4).numel() == 0
    g = synth.Genome(5000)
    owner = synth.assign_chromosomes(g, 8)
    load = [sum(g.n_loci[c] for c in range(len(g)) if owner[c] == r) for r in range(8)]
    assert len(set(owner)) == 8 and max(load) / (sum(load) / 8) < 1.15          # greedy balance within 15 %


def _bench_hash_worker(rank, world, port, result_dir):
    """bench.py's N > 1 verification plumbing on CPU tensors over gloo: per-chromosome hash tables of (row identity, value) are
    additive over ranks whatever the sharding - cis rows by chromosome, trans rows dealt in blocks - and TorchComm's helpers
    return the same numbers on every rank."""
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch
    import torch.distributed as td
    import bench
    from fithic_amd import synth
    td.init_process_group("gloo", rank=rank, world_size=world)
    comm = bench.TorchComm(td, torch, torch.device("cpu"))
    cfg = dict(bench.CONFIGS["C5"], keep=0.2)
    genome = synth.Genome(cfg["res"], [300_000, 250_000, 200_000])
    dev = torch.device("cpu")

    def rows_of(r, w):
        owner = synth.assign_chromosomes(genome, w)
        mine = [c for c in range(len(genome)) if owner[c] == r]
        cols, n, n_cis, n_trans = bench.build_rows(synth, torch, cfg, genome, mine, r, w, dev)
        return [c[:n] for c in cols], n, n_trans

    def table(cols, n):
        keys = bench.row_keys(torch, synth, cols, n)
        value = (cols[4].to(torch.float64) + 0.25) / (1.0 + cols[1].to(torch.float64))      # any function of the row
        return bench.hash_table(torch, synth, keys, cols[0], value, len(genome))

    mine, n_mine, n_trans = rows_of(rank, world)
    parts = comm.gather_rows(table(mine, n_mine))
    total = parts[0].clone()
    for t in parts[1:]:
        total += t
    whole, n_whole, _ = rows_of(0, 1)
    msgs = []
    if not torch.equal(total, table(whole, n_whole)):
        msgs.append("summed hash tables differ from the one-process table")
    if comm.sum_int(n_mine) != n_whole or n_trans < 100:
        msgs.append("row counts")
    if comm.max_float(1.5 + rank) != 1.5 + (world - 1) or comm.gather_floats([float(rank), 2.0]) != [[float(r), 2.0] for r in range(world)]:
        msgs.append("TorchComm helpers")
    # a single flipped bit in one rank's values must show in exactly that row's chromosome
    if rank == 0:
        keys = bench.row_keys(torch, synth, whole, n_whole)
        v = (whole[4].to(torch.float64) + 0.25) / (1.0 + whole[1].to(torch.float64))
        good = bench.hash_table(torch, synth, keys, whole[0], v, len(genome))
        v2 = v.clone()
        v2.view(torch.int64)[n_whole // 2] ^= 1
        bad = bench.hash_table(torch, synth, keys, whole[0], v2, len(genome))
        if int((good != bad).sum()) != 1:
            msgs.append("a one-ulp change is not seen (or seen in the wrong chromosome)")
    with open(os.path.join(result_dir, "rank%d.txt" % rank), "w") as f:
        f.write("OK" if not msgs else "FAIL: " + "; ".join(msgs))
    td.destroy_process_group()


def test_bench_sharded_verification_hashes_over_gloo(tmp_path):
    torch = pytest.importorskip("torch")
    import torch.multiprocessing as mp
    mp.spawn(_bench_hash_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    for r in range(2):
        with open(os.path.join(str(tmp_path), "rank%d.txt" % r)) as f:
            assert f.read() == "OK"


def test_schedule_check_catches_drift():
    """tests/dist_schedule.py against csrc/fhx_dist_schedule.def: the right trace passes; a collective left out, an extra one, a
    wrong kind or a wrong size is named."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import dist_schedule
    ids = [r[0] for r in dist_schedule.steps()]
    assert ids[:2] == ["NDIST_AGREE", "STATS_PACK"] and "KEYS" in ids and len(set(ids)) == len(ids)
    env = dict(world=2, w=10, s=128, nd=50, n_local=7, m=9)
    good = ([("NDIST_AGREE", "ALL_REDUCE_MAX", 1), ("STATS_PACK", "ALL_REDUCE_SUM", 30), ("TOP_HIST", "ALL_REDUCE_SUM", 8192),
             ("SAMPLES", "ALL_GATHER", 1024), ("COUNT_MATRIX", "ALL_GATHER", 16), ("KEYS", "ALL_TO_ALL_V", 7),
             ("SLICE_MAX", "ALL_GATHER", 8), ("Q_BACK", "ALL_TO_ALL_V", 9), ("OUTLIER_HIST", "ALL_REDUCE_SUM", 51),
             ("SKIP_LIMIT", "ALL_REDUCE_MIN", 1)])
    phases = ["LOAD", "STATS", "BH", "NEXT"]
    dist_schedule.check(good, phases, [env] * 4)
    for bad in (good[:4] + good[5:], good[:3] + [("SLICE_MAX", "ALL_GATHER", 8)] + good[3:],
                [("STATS_PACK", "ALL_REDUCE_MAX", 30) if r[0] == "STATS_PACK" else r for r in good],
                [("SAMPLES", "ALL_GATHER", 8192) if r[0] == "SAMPLES" else r for r in good]):
        with pytest.raises(AssertionError):
            dist_schedule.check(bad, phases, [env] * 4)
