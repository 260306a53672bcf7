"""TEST INFRASTRUCTURE - a model of the library's sharded schedule (fithic_amd/csrc/fhx_dist.inc) written over torch.distributed.

The product runs the sharded pass inside libfithic_mi355x.so (fhx_run_pass_distributed; RCCL or a caller-provided transport).
This file restates that schedule in Python so that it can be exercised WITHOUT a GPU: tests/test_dist_gloo.py runs it at world
size 2 over gloo with the oracle standing in for the kernels (the exchange logic - what is reduced, gathered, cut and sent
where - is then checked against a one-process run), and tests/test_gpu_dist.py runs it over the C ABI's building blocks
(fhx_bh_top_hist, fhx_bh_local_sort, fhx_bh_apply_sorted ...) with two ranks sharing GPU 0.  Nothing under fithic_amd/ imports it.

Multi-GPU pass: one process per GPU, contacts sharded by chromosome, torch.distributed for the exchanges
(backend "nccl" = RCCL over xGMI on the GPU box; "gloo" on CPU tensors in the tests).

The reference is single-process; what has to be global is exactly what its data structures
make global (section 8e):

  exchange 1  mainDic and the five sums are genome-wide (fithic/fithic.py:434-440)
              -> one all-reduce(SUM) of [sums | sumCC histogram | row-count histogram] (<= 2 x 50 k int64) and one
                 all-reduce(MAX) of the largest count; every rank then runs the same deterministic host fit
  exchange 2  benjamini_hochberg_correction ranks ALL p-values (fithic/myStats.py:27-46)
              -> local radix sort, regular samples all-gathered, common splitters, all-to-all(v) of the keys so that
                 rank r owns the r-th slice of the global order, local sort of the received runs, all-gather of the
                 slice sizes (rank offsets) and of the slice maxima (carry of the running MAX), local BH + max-scan,
                 all-to-all(v) of q back, scatter to row order
  pass >= 2   the multiset of outlier distances is genome-wide (fithic/fithic.py:528-548) -> all-reduce(SUM)

All per-GPU compute goes through `LocalOps` (the C ABI); the exchange logic itself is backend-agnostic so that the
world_size-2 gloo tests exercise the same code with a checker-backed LocalOps.

ONE list of collectives for both statements of the schedule: fithic_amd/csrc/fhx_dist_schedule.def.  Every collective below
names the row of that table it is (`step=`) and `Comm.trace` records (step, kind, size) in the table's units, exactly as the
library does under FHX_DIST_TRACE=1 (fhx_dist_trace); tests/dist_schedule.py holds both traces against the table
(tests/test_dist_gloo.py here on CPU, tests/test_gpu_native_dist.py for the library) - a change on one side only fails a test.
"""
import numpy as np

DIST_MAX_SAMPLES = 8192                  # fhx_dist.inc: world x samples per rank, sorted by one workgroup


def samples_per_rank(world):
    """fhx_bh_distributed: s = max(1, min(128, DIST_MAX_SAMPLES / world))"""
    return max(1, min(128, DIST_MAX_SAMPLES // world))


class Comm:
    """Thin wrapper over torch.distributed for tensors on `device`."""

    def __init__(self, td, device, compute_device=None):
        """`device`: where the collectives run (cuda for RCCL, cpu for gloo); `compute_device`: where LocalOps keeps
        its tensors (defaults to `device`; a cuda compute device with a cpu/gloo transport stages through host memory,
        which lets two ranks share one GPU in the tests)."""
        import torch
        self.td, self.torch, self.device = td, torch, device
        self.compute_device = compute_device if compute_device is not None else device
        self.rank, self.world = td.get_rank(), td.get_world_size()
        self.trace = []                  # (step id, kind, size) of every schedule collective, in fhx_dist_schedule.def's units

    def note(self, step, kind, size):
        if step is not None:
            self.trace.append((step, kind, int(size)))

    def barrier(self):
        self.td.barrier()

    def _done(self):
        """torch.distributed ops only order against torch's CURRENT stream; the engine launches on its own stream, so a
        tensor handed to the C ABI right after a collective must be complete on the host's view first."""
        if self.device.type == "cuda":
            self.torch.cuda.current_stream(self.device).synchronize()

    def all_reduce_i64(self, arr, op="sum", step=None):
        t = self.torch.as_tensor(np.ascontiguousarray(arr, np.int64)).to(self.device)
        self.note(step, {"sum": "ALL_REDUCE_SUM", "max": "ALL_REDUCE_MAX", "min": "ALL_REDUCE_MIN"}[op], t.numel())
        self.td.all_reduce(t, op={"sum": self.td.ReduceOp.SUM, "max": self.td.ReduceOp.MAX, "min": self.td.ReduceOp.MIN}[op])
        return t.cpu().numpy()

    def all_gather_i64(self, t, step=None):
        t = t.to(self.device)
        self.note(step, "ALL_GATHER", 8 * t.numel())
        out = [self.torch.empty_like(t) for _ in range(self.world)]
        self.td.all_gather(out, t)
        self._done()
        return [o.to(self.compute_device) for o in out]

    def gather_lists_i64(self, arr, width):
        """fhx_dist.inc dist_gather_lists: one variable-length list of `width`-tuples per rank -> list of numpy arrays.  Lengths
        first (NF_LENGTHS), then every list padded to the longest (NF_LISTS)."""
        arr = np.ascontiguousarray(arr, np.int64)
        n_mine = len(arr) // width
        lens = [int(t.item()) for t in self.all_gather_i64(self.torch.tensor([n_mine], dtype=self.torch.int64), step="NF_LENGTHS")]
        longest = max(lens)
        if longest == 0:
            return [np.zeros(0, np.int64) for _ in range(self.world)]
        pad = np.zeros(longest * width, np.int64)
        pad[:len(arr)] = arr
        out = self.all_gather_i64(self.torch.as_tensor(pad), step="NF_LISTS")
        return [o.cpu().numpy()[:n * width] for o, n in zip(out, lens)]

    def all_gather_f64_scalar(self, v, step=None):
        t = self.torch.tensor([float(v)], dtype=self.torch.float64, device=self.device)
        self.note(step, "ALL_GATHER", 8)
        out = [self.torch.empty_like(t) for _ in range(self.world)]
        self.td.all_gather(out, t)
        return [float(o.item()) for o in out]

    def all_reduce_device_i64(self, ptr, n, step=None):
        """In-place SUM all-reduce of n int64 values living at device address `ptr` (the engine's own buffer): wrapped as a
        torch tensor without a copy.  With a host transport (gloo in the tests) it is staged through the host."""
        torch = self.torch
        self.note(step, "ALL_REDUCE_SUM", n)

        class _View:                                    # minimal __cuda_array_interface__ carrier
            __cuda_array_interface__ = {"shape": (int(n),), "typestr": "<i8", "data": (int(ptr), False), "version": 2}

        t = torch.as_tensor(_View(), device=self.compute_device)
        if self.device.type == "cuda":
            self.td.all_reduce(t, op=self.td.ReduceOp.SUM)
            self._done()
        else:
            h = t.cpu()
            self.td.all_reduce(h, op=self.td.ReduceOp.SUM)
            t.copy_(h)
            torch.cuda.current_stream(self.compute_device).synchronize()

    def all_to_all_v(self, send, send_counts, recv_counts, step=None):
        """send: 1-D tensor of 8-byte elements laid out rank-major; both count vectors are known to the caller (they come out of
        the all-gathered count matrix, as in the library); returns (recv tensor, recv_counts)."""
        torch = self.torch
        recv_counts = [int(v) for v in recv_counts]
        self.note(step, "ALL_TO_ALL_V", sum(int(v) for v in send_counts))
        send = send.to(self.device).contiguous()
        recv = torch.empty(sum(recv_counts), dtype=send.dtype, device=self.device)
        self.td.all_to_all_single(recv, send, output_split_sizes=recv_counts, input_split_sizes=[int(v) for v in send_counts])
        self._done()
        return recv.to(self.compute_device), recv_counts

    def max_float(self, v):
        return max(self.all_gather_f64_scalar(v))

    def sum_int(self, v):
        return int(self.all_reduce_i64(np.array([int(v)]))[0])

    def gather_floats(self, vals):
        t = self.torch.tensor([float(v) for v in vals], dtype=self.torch.float64, device=self.device)
        out = [self.torch.empty_like(t) for _ in range(self.world)]
        self.td.all_gather(out, t)
        return [o.cpu().tolist() for o in out]


class LocalOps:
    """Per-GPU compute of a distributed pass, through the C ABI (fithic_amd._capi.Context)."""

    def __init__(self, engine, torch, device):
        self.eng, self.ctx, self.torch, self.device = engine, engine.ctx, torch, device

    def local_stats(self):
        from fithic_amd import _capi
        st = self.ctx.pass_stats()
        return st, self.ctx.get_array(_capi.A_HIST_SUMCC), self.ctx.get_array(_capi.A_HIST_NPAIRS)

    @property
    def nonfixed(self):
        return getattr(self.eng, "resolution", 1) == 0

    def hist_span(self, n_dist):
        """Index window of the distance histogram that can be non-zero (only in-range rows are counted): the exchange
        carries this window instead of one entry per locus of the longest chromosome."""
        res = getattr(self.eng, "resolution", 0)
        up = getattr(self.eng, "dist_up", float("inf"))
        lo = getattr(self.eng, "dist_low", 0)
        if not res:
            return 0, n_dist
        first = max(0, int(lo) // res)
        last = n_dist if up == float("inf") else min(n_dist, int(up) // res + 2)
        return min(first, last), last

    def local_dist_keys(self):
        """-r 0: the distinct in-range distances of this rank's rows (its histograms are aligned to them)."""
        from fithic_amd import _capi
        return self.ctx.get_array(_capi.A_DIST_KEYS)

    def set_dist_keys(self, keys):
        self.ctx.set_dist_keys(keys)

    def set_global_and_fit(self, st, hist_cc, hist_np):
        self.ctx.set_global_stats(st, hist_cc, hist_np)
        return self.ctx.fit()

    def pvalues(self):
        self.ctx.pvalues()

    def top_hist(self):
        return self.ctx.bh_top_hist()

    def set_cutoff(self, global_hist, n_tests):
        self.ctx.bh_set_cutoff(global_hist, n_tests)

    def top_hist_device(self):
        """(device address, length) of the key histogram, left in HBM for an in-place all-reduce."""
        return self.ctx.bh_top_hist_device(), 8192

    def set_cutoff_device(self, n_tests):
        self.ctx.bh_set_cutoff_device(n_tests)

    def local_sorted_keys(self):
        """int64 tensor (all keys < 2^62, so signed order = unsigned order) of this rank's sorted p < 1 bit patterns."""
        self.ctx.bh_local_sort()
        n = self.ctx.n_sorted()
        t = self.torch.empty(n, dtype=self.torch.int64, device=self.device)
        self.ctx.memcpy_d2d(t.data_ptr(), self.ctx.device_ptr(2), 8 * n)
        return t

    def sort_keys(self, keys):
        out = self.torch.empty_like(keys)
        perm = self.torch.empty(keys.numel(), dtype=self.torch.int32, device=self.device)
        self.ctx.sort_u64(keys.data_ptr(), keys.numel(), out.data_ptr(), perm.data_ptr())
        return out, perm

    def bh_segment(self, sorted_keys, rank0, carry, n_tests, want_q):
        q = self.torch.empty(sorted_keys.numel(), dtype=self.torch.float64, device=self.device) if want_q else None
        mx = self.ctx.bh_apply_sorted(sorted_keys.data_ptr() if sorted_keys.numel() else 0, sorted_keys.numel(), rank0, carry,
                                      n_tests, q.data_ptr() if (want_q and q.numel()) else 0)
        return q, mx

    def scatter_q(self, q_sorted_local):
        self.ctx.bh_scatter(q_sorted_local.data_ptr() if q_sorted_local.numel() else 0)

    def next_pass_local(self):
        from fithic_amd import _capi
        n = self.ctx.next_pass()
        if self.nonfixed:
            # the context's list = what was set last (the genome-wide list of the previous pass) + this rank's fresh ones
            after = self.ctx.get_array(_capi.A_OUTLIER_DISTS)
            fresh = _multiset_minus(after, self._outlier_global)
            self._outlier_local = np.sort(np.concatenate([self._outlier_local, fresh]))
            return n, self._outlier_local
        return n, self.ctx.get_array(_capi.A_OUTLIER_DIST_HIST)

    _outlier_global = np.zeros(0, np.int64)
    _outlier_local = np.zeros(0, np.int64)

    def set_outlier_dists(self, dists):
        self._outlier_global = np.sort(np.asarray(dists, np.int64))
        self.ctx.set_outlier_dists(self._outlier_global)

    def set_outlier_hist(self, hist):
        self.ctx.set_outlier_dist_hist(hist)

    def get_skip_limit(self):
        return self.ctx.get_skip_limit()

    def set_skip_limit(self, limit):
        self.ctx.set_skip_limit(limit)


def _multiset_minus(a, b):
    """Sorted int64 multiset a minus multiset b (b is contained in a)."""
    a, b = np.asarray(a, np.int64), np.asarray(b, np.int64)
    if len(b) == 0:
        return a.copy()
    va, ca = np.unique(a, return_counts=True)
    vb, cb = np.unique(b, return_counts=True)
    ca = ca.copy()
    ca[np.searchsorted(va, vb)] -= cb
    return np.repeat(va, ca)


def merge_keyed_histograms(keys_per_rank, cc_per_rank, np_per_rank):
    """-r 0: union of the ranks' distinct distances and the sums of their histograms on it."""
    keys = np.unique(np.concatenate(keys_per_rank)) if keys_per_rank else np.zeros(0, np.int64)
    cc, npairs = np.zeros(len(keys), np.int64), np.zeros(len(keys), np.int64)
    for k, c, m in zip(keys_per_rank, cc_per_rank, np_per_rank):
        at = np.searchsorted(keys, k)
        np.add.at(cc, at, c[:len(k)])
        np.add.at(npairs, at, m[:len(k)])
    return keys, cc, npairs


def choose_splitters(torch, samples_sorted, world):
    """world-1 splitters at regular positions of the sorted, gathered samples."""
    n = samples_sorted.numel()
    if n == 0 or world == 1:
        return samples_sorted[:0]
    pos = [min(n - 1, (r * n) // world) for r in range(1, world)]
    return samples_sorted[torch.tensor(pos, dtype=torch.int64, device=samples_sorted.device)]


def distributed_bh(comm, ops, n_tests, timings=None):
    """Global BH over the p-values of all ranks; leaves q in row order on every rank (fhx_bh_distributed)."""
    import time
    torch = comm.torch
    t_last = [time.perf_counter()]

    def lap(name):
        if timings is not None:
            now = time.perf_counter()
            timings[name] = timings.get(name, 0.0) + (now - t_last[0])
            t_last[0] = now

    # exact early cutoff: rows whose q is provably 1 are neither sorted nor exchanged (needs the GLOBAL key histogram)
    if hasattr(ops, "top_hist_device") and comm.compute_device.type == "cuda":
        ptr, length = ops.top_hist_device()              # histogram stays in HBM: all-reduce in place, cutoff on the device
        comm.all_reduce_device_i64(ptr, length, step="TOP_HIST")
        ops.set_cutoff_device(n_tests)
    else:
        ops.set_cutoff(comm.all_reduce_i64(ops.top_hist(), step="TOP_HIST"), n_tests)
    lap("bh_cutoff")
    keys = ops.local_sorted_keys()
    n = keys.numel()
    lap("bh_local_sort")
    # regular samples -> common splitters
    s = samples_per_rank(comm.world)
    if n:
        idx = torch.clamp(((torch.arange(1, s + 1, device=keys.device, dtype=torch.int64) * n) // (s + 1)), max=n - 1)
        samples = keys[idx]
    else:
        samples = torch.full((s,), (1 << 62), dtype=torch.int64, device=keys.device)      # sorts after every real key
    gathered = torch.cat(comm.all_gather_i64(samples, step="SAMPLES"))
    gathered = gathered[gathered < (1 << 62)]
    splitters = choose_splitters(torch, torch.sort(gathered).values, comm.world)
    # cut the local run at the splitters; the count matrix (who sends how much to whom) goes to everybody
    if splitters.numel():
        cuts = torch.searchsorted(keys, splitters, right=False).cpu().tolist()
    else:
        cuts = [n] * (comm.world - 1)
    bounds = [0] + [int(c) for c in cuts] + [n]
    send_counts = [bounds[r + 1] - bounds[r] for r in range(comm.world)]
    matrix = [t.cpu().tolist() for t in comm.all_gather_i64(torch.tensor(send_counts, dtype=torch.int64), step="COUNT_MATRIX")]
    recv_counts = [int(matrix[src][comm.rank]) for src in range(comm.world)]
    rank0 = sum(int(matrix[src][dst]) for dst in range(comm.rank) for src in range(comm.world))
    lap("bh_splitters")
    recv, _ = comm.all_to_all_v(keys, send_counts, recv_counts, step="KEYS")
    lap("bh_exchange_keys")
    # my slice of the global order: merge the received runs (ties keep sender order), remember where each element came from
    mine_sorted, perm = ops.sort_keys(recv)
    m = mine_sorted.numel()
    _, seg_max = ops.bh_segment(mine_sorted, rank0, 0.0, n_tests, want_q=False)
    maxima = comm.all_gather_f64_scalar(seg_max, step="SLICE_MAX")
    carry = max([0.0] + maxima[:comm.rank])
    q_sorted, _ = ops.bh_segment(mine_sorted, rank0, carry, n_tests, want_q=True)
    lap("bh_slice")
    # back to arrival order, then back to the owners (reverse all-to-all), then to row order
    q_arrival = torch.empty_like(q_sorted)
    if m:
        q_arrival[perm.to(torch.int64)] = q_sorted
    q_back, _ = comm.all_to_all_v(q_arrival, recv_counts, send_counts, step="Q_BACK")
    ops.scatter_q(q_back)
    lap("bh_return")
    return n, m


class DistributedPass:
    """K1 -> all-reduce -> fit -> K2 -> distributed BH, one call per spline pass."""

    def __init__(self, engine, comm, ops=None):
        self.comm = comm
        self.ops = ops if ops is not None else LocalOps(engine, comm.torch, comm.compute_device)
        self.n_dist_global = None
        self.info = None
        self.stats = None
        self.timings = {}                # host wall seconds per stage, summed over the passes run so far

    def run(self):
        import time
        comm, ops = self.comm, self.ops
        t0 = time.perf_counter()
        st, hist_cc, hist_np = ops.local_stats()
        self.timings["k1"] = self.timings.get("k1", 0.0) + time.perf_counter() - t0
        if getattr(ops, "nonfixed", False):
            return self._run_nonfixed(st, hist_cc, hist_np)
        if self.n_dist_global is None:                   # once per loaded row set (dist_agree_n_dist)
            self.n_dist_global = int(comm.all_reduce_i64(np.array([len(hist_cc)]), op="max", step="NDIST_AGREE")[0])
        nd = self.n_dist_global
        a, b = ops.hist_span(nd) if hasattr(ops, "hist_span") else (0, nd)      # the window that can be non-zero
        w = b - a
        full_cc, full_np = np.zeros(nd, np.int64), np.zeros(nd, np.int64)
        full_cc[:len(hist_cc)] = hist_cc
        full_np[:len(hist_np)] = hist_np
        # one SUM all-reduce: 7 sums, one slot per rank for its largest count (a MAX in disguise), the two windows
        pack = np.zeros(8 + comm.world + 2 * w, np.int64)
        pack[:8] = [st.inter_count, st.inter_sum, st.intra_all_count, st.intra_all_sum, st.in_range_count, st.in_range_sum,
                    st.n_skipped, 0]
        pack[8 + comm.rank] = st.max_count
        pack[8 + comm.world:8 + comm.world + w] = full_cc[a:b]
        pack[8 + comm.world + w:] = full_np[a:b]
        pack = comm.all_reduce_i64(pack, step="STATS_PACK")
        (st.inter_count, st.inter_sum, st.intra_all_count, st.intra_all_sum, st.in_range_count, st.in_range_sum,
         st.n_skipped) = [int(v) for v in pack[:7]]
        st.max_count = int(pack[8:8 + comm.world].max())
        full_cc[:] = 0
        full_np[:] = 0
        full_cc[a:b] = pack[8 + comm.world:8 + comm.world + w]
        full_np[a:b] = pack[8 + comm.world + w:]
        t1 = time.perf_counter()
        info = ops.set_global_and_fit(st, full_cc, full_np)
        t2 = time.perf_counter()
        ops.pvalues()
        t3 = time.perf_counter()
        distributed_bh(comm, ops, info.bh_total_tests, self.timings)
        for name, dt in (("allreduce_stats", t1 - t0 - 0.0), ("fit", t2 - t1), ("k2_launch", t3 - t2)):
            self.timings[name] = self.timings.get(name, 0.0) + dt
        self.info, self.stats = info, st
        return info

    def _run_nonfixed(self, st, hist_cc, hist_np):
        """-r 0: the histogram is keyed by the distinct distances, which differ between ranks: every rank's (distance, sum of
        counts, rows) triples travel as one padded list with the seven sums and the largest count in front
        (dist_pass_stats_nonfixed), are merged by distance on every host, then the same fit / K2 / distributed BH as fixed-size runs."""
        comm, ops = self.comm, self.ops
        if self.n_dist_global is None:                   # the agreement of dist_agree_n_dist also carries "explicit distances here"
            comm.all_reduce_i64(np.array([1 << 62]), op="max", step="NDIST_AGREE")
            self.n_dist_global = 0
        keys = np.asarray(ops.local_dist_keys(), np.int64)
        head = [st.inter_count, st.inter_sum, st.intra_all_count, st.intra_all_sum, st.in_range_count, st.in_range_sum, st.n_skipped,
                st.max_count, 0]
        body = np.stack([keys, np.asarray(hist_cc, np.int64)[:len(keys)], np.asarray(hist_np, np.int64)[:len(keys)]], axis=1).reshape(-1)
        lists = comm.gather_lists_i64(np.concatenate([np.array(head, np.int64), body]), width=3)
        sums = np.zeros(7, np.int64)
        max_count = 0
        k_all, cc_all, np_all = [], [], []
        for L in lists:
            sums += L[:7]
            max_count = max(max_count, int(L[7]))
            t = L[9:].reshape(-1, 3)
            k_all.append(t[:, 0])
            cc_all.append(t[:, 1])
            np_all.append(t[:, 2])
        gkeys, gcc, gnp = merge_keyed_histograms(k_all, cc_all, np_all)
        (st.inter_count, st.inter_sum, st.intra_all_count, st.intra_all_sum, st.in_range_count, st.in_range_sum,
         st.n_skipped) = [int(v) for v in sums]
        st.max_count = max_count
        if len(gkeys) == 0:                                # no in-range row anywhere: keep one key so that the C ABI accepts it
            gkeys, gcc, gnp = np.zeros(1, np.int64), np.zeros(1, np.int64), np.zeros(1, np.int64)
        ops.set_dist_keys(gkeys)
        info = ops.set_global_and_fit(st, gcc, gnp)
        ops.pvalues()
        distributed_bh(comm, ops, info.bh_total_tests)
        self.info, self.stats = info, st
        return info

    def reset(self):
        """After Engine.reset_passes(): forget the outlier state kept on this side as well."""
        self.ops._outlier_global = np.zeros(0, np.int64)
        self.ops._outlier_local = np.zeros(0, np.int64)

    def next_pass(self):
        """Fold outliers locally, then make the outlier-distance multiset genome-wide (fhx_next_pass_distributed)."""
        n_local, hist = self.ops.next_pass_local()
        cap = 1 << 62
        if getattr(self.ops, "nonfixed", False):           # `hist` is this rank's cumulative list of outlier distances
            send = np.concatenate([np.array([n_local, min(self.ops.get_skip_limit(), cap)], np.int64), np.asarray(hist, np.int64)])
            lists = self.comm.gather_lists_i64(send, width=1)
            total = sum(int(L[0]) for L in lists)
            limit = min(int(L[1]) for L in lists)
            self.ops.set_outlier_dists(np.sort(np.concatenate([L[2:] for L in lists])))
            self.ops.set_skip_limit(limit if limit < cap else (1 << 63) - 1)
            return total
        nd = int(self.n_dist_global)
        buf = np.zeros(nd + 1, np.int64)                   # the multiset of outlier distances + this rank's outlier total
        buf[:min(len(hist), nd)] = hist[:nd]
        buf[nd] = n_local
        buf = self.comm.all_reduce_i64(buf, step="OUTLIER_HIST")
        self.ops.set_outlier_hist(buf[:nd])
        # first duplicated outlier LINE of the whole file (-p >= 3 semantics): min over ranks, in file positions
        limit = int(self.comm.all_reduce_i64(np.array([min(self.ops.get_skip_limit(), cap)]), op="min", step="SKIP_LIMIT")[0])
        self.ops.set_skip_limit(limit if limit < cap else (1 << 63) - 1)
        return int(buf[nd])
