"""The sharded pass BEHIND the C ABI (fhx_comm_init* / fhx_run_pass_distributed / fhx_next_pass_distributed), driven with
bare ctypes + numpy - no torch in these processes.

  * world 2 and 3 on ONE GPU: RCCL refuses two ranks on one device, so the ranks are separate processes whose collectives
    are the library's caller-provided transport (fhx_comm_init_custom) with fithic_amd.sharded.PipeTransport (host-staged
    exchange over multiprocessing pipes) - the schedule, the kernels and the buffers are exactly the ones the RCCL transport runs.  Each rank's p and q must equal,
    bit for bit, what a single-GPU run over all rows gives for its rows (1-3 passes, an empty shard, inter rows).
  * world 1 through REAL RCCL (ncclCommInitRank, all-reduce / all-gather / send-recv to self on the engine's stream): the
    same bits as fhx_pass_stats + fhx_fit + fhx_pvalues + fhx_bh.
The 8-GPU RCCL run itself is the driver's (bench.py --gpus N)."""
import ctypes
import multiprocessing as mp
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _load_case(case):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from conftest import load_case, case_args
    from fithic_amd import tables
    if case.startswith("{"):                            # a random case of the fuzz generator, written by the parent: its kw as JSON
        import json
        kw = json.loads(case)
        kw["U"] = float("inf") if kw["U"] is None else kw["U"]
    else:
        meta, _ = load_case(case)
        kw = case_args(meta)
    chroms = tables.ChromIndex()
    con = tables.read_contacts(kw["contacts"], chroms)
    frag = tables.read_fragments(kw["frags"], chroms)
    bias = tables.read_bias(kw["bias_path"], chroms) if kw["bias_path"] else None
    return kw, chroms, con, frag, bias


def _make_ctx(kw, chroms, frag, bias, con, rows):
    from fithic_amd import _capi
    from fithic_amd.engine import MODES
    c = _capi.Context(0)
    c.set_params(kw["resolution"], kw["L"], kw["U"], kw["n_bins"], kw["mapp_thres"], MODES[kw["mode"]], kw["tL"], kw["tU"])
    c.load_fragments(*frag, chroms.sort_rank())
    if bias:
        c.load_bias(*bias)
    t = con.take(rows)
    c.load_pairs(t.chr1, t.mid1, t.chr2, t.mid2, t.count)
    return c


def _same(a, b):
    return bool(((a.view(np.int64) == b.view(np.int64)) | (np.isnan(a) & np.isnan(b))).all())


def _single_gpu_passes(ctx, n_rows, passes):
    out = []
    for pi in range(passes):
        ctx.pass_stats()
        info = ctx.fit()
        ctx.pvalues()
        ctx.bh(info.bh_total_tests)
        v = ctx.fetch(n_rows)
        out.append((v["p"], v["q"], ctx.next_pass() if pi + 1 < passes else None))
    return out


def _rank_main(rank, world, conns, case, passes, split, result_path):
    msgs = []
    try:
        kw, chroms, con, frag, bias = _load_case(case)
        n = len(con)
        if split == "by_chr":
            mine = np.flatnonzero(con.chr1 % world == rank)
        elif split.startswith("random:"):             # rows dealt at random (the same deal in every rank process)
            owner = np.random.default_rng(int(split[7:])).integers(0, world, n)
            mine = np.flatnonzero(owner == rank)
        elif split == "empty_last":                   # the last rank holds no row at all
            mine = np.flatnonzero(con.chr1 % (world - 1) == rank) if rank < world - 1 else np.zeros(0, np.int64)
        else:                                         # contiguous blocks of the file
            mine = np.arange(n * rank // world, n * (rank + 1) // world)
        single = _make_ctx(kw, chroms, frag, bias, con, np.arange(n))
        want = _single_gpu_passes(single, n, passes)
        single.close()
        local = _make_ctx(kw, chroms, frag, bias, con, mine)
        local.set_global_rows(mine)                   # file positions (the -p >= 3 semantics)
        from fithic_amd.sharded import PipeTransport
        tr = PipeTransport(local, rank, world, conns)
        os.environ["FHX_DIST_TRACE"] = "1"             # the library records every collective it issues (fhx_dist_trace)
        local.comm_init_custom(tr.struct, rank, world)
        assert local.comm_info()[:2] == (rank, world)
        n_sorted = []
        for pi in range(passes):
            info = local.run_pass_distributed()
            n_sorted.append(local.n_sorted())
            got = local.fetch(len(mine))
            for key, ref in (("p", want[pi][0]), ("q", want[pi][1])):
                if not _same(got[key], ref[mine]):
                    bad = ~((got[key].view(np.int64) == ref[mine].view(np.int64)) | (np.isnan(got[key]) & np.isnan(ref[mine])))
                    msgs.append("pass %d: %s differs on %d of %d rows" % (pi + 1, key, int(bad.sum()), len(mine)))
            st = local.stats()
            if pi == 0 and st.n_skipped != 0:
                msgs.append("skipped rows in pass 1")
            if pi + 1 < passes:
                tot = local.next_pass_distributed()
                if tot != want[pi][2]:
                    msgs.append("outliers %d vs %d" % (tot, want[pi][2]))
        msgs += _schedule_errors(local, kw, world, passes, n_sorted)
        local.close()
    except Exception as e:
        import traceback
        msgs.append("exception: %r\n%s" % (e, traceback.format_exc()))
    with open(result_path, "w") as f:
        f.write("OK" if not msgs else "FAIL: " + "; ".join(msgs))


def _schedule_errors(ctx, kw, world, passes, n_sorted):
    """The collectives the library issued (FHX_DIST_TRACE=1) against csrc/fhx_dist_schedule.def - the list the Python model of the
    schedule is held to as well (tests/test_dist_gloo.py): ids, order, kinds and every size the test can name."""
    import dist_schedule
    trace = ctx.dist_trace()
    explicit = any(step == "NF_LENGTHS" for step, _, _ in trace)        # -r 0 / off-grid loci anywhere: distances travel as lists
    s_per = max(1, min(128, 8192 // world))
    phases, envs = ["LOAD"], [dict(world=world)]
    res = kw["resolution"]
    nd = int(ctx.stats().n_dist)
    for pi in range(passes):
        if pi:
            phases.append("NEXT_NF" if explicit else "NEXT")
            envs.append(dict(world=world, nd=nd, width=1))
        phases.append("STATS_NF" if explicit else "STATS")
        if explicit:
            envs.append(dict(world=world, width=3))
        else:
            a = min(max(0, int(kw["L"]) // res), nd)
            b = nd if kw["U"] == float("inf") else max(a, min(nd, int(kw["U"]) // res + 2))
            envs.append(dict(world=world, w=b - a))
        phases.append("BH")
        envs.append(dict(world=world, s=s_per, n_local=n_sorted[pi]))
    try:
        dist_schedule.check(trace, phases, envs)
    except AssertionError as e:
        return ["schedule: %s" % e]
    return []


def _run_world(world, case, passes, split, tmp_path):
    ctxmp = mp.get_context("spawn")
    pipes = {}
    for a in range(world):
        for b in range(a + 1, world):
            pipes[(a, b)] = ctxmp.Pipe(duplex=True)
    procs = []
    for r in range(world):
        conns = {}
        for (a, b), (ca, cb) in pipes.items():
            if a == r:
                conns[b] = ca
            elif b == r:
                conns[a] = cb
        out = os.path.join(str(tmp_path), "rank%d.txt" % r)
        p = ctxmp.Process(target=_rank_main, args=(r, world, conns, case, passes, split, out))
        p.start()
        procs.append((p, out))
    for p, out in procs:
        p.join(240)
        if p.is_alive():
            for q, _ in procs:
                q.kill()
            pytest.fail("a rank hangs")
        assert p.exitcode == 0, "rank process died (exit code %r)" % (p.exitcode,)
    for _, out in procs:
        with open(out) as f:
            assert f.read() == "OK"


@pytest.mark.parametrize("world,case,passes,split", [
    (2, "f2_all", 2, "by_chr"), (2, "f1_bias", 2, "blocks"), (2, "f6_quirk_all", 3, "by_chr"),
    (3, "f2_all", 2, "empty_last"), (3, "f2_intra", 2, "blocks"),
    # explicit distances (-r 0, and -r N on loci that are not on one grid): mainDic travels as (distance, sum, rows) triples
    (2, "f8_nonfixed_all", 3, "by_chr"), (3, "f8_nonfixed_hESC", 1, "blocks"), (3, "f8_nonfixed_nobounds", 2, "random:5"),
    (2, "f11_offgrid_all", 2, "by_chr"), (3, "f11_offgrid_intra", 2, "empty_last")])
def test_native_sharded_pass_equals_single_gpu(world, case, passes, split, tmp_path):
    _run_world(world, case, passes, split, tmp_path)


def test_ranks_on_a_grid_follow_the_ranks_that_are_not(tmp_path):
    """-r 10000 on a genome where only SOME chromosomes have irregular midpoints, sharded by chromosome: the ranks whose rows sit
    on a grid learn (in the all-reduce that equalises the histogram lengths) that another rank's do not, re-slot their rows the
    same way, and the pass runs on explicit distances everywhere - the same bits as one GPU holding all rows, 2 passes."""
    import gzip
    import json
    data = os.path.join(ROOT, "tests", "golden", "data")
    paths = {}
    for kind in ("contacts", "frags", "bias"):
        text = b"".join(gzip.open(os.path.join(data, "%s.%s.gz" % (name, kind)), "rb").read() for name in ("quirk", "irregular"))
        paths[kind] = os.path.join(str(tmp_path), "mixed.%s.gz" % kind)
        with gzip.open(paths[kind], "wb") as f:
            f.write(text)
    kw = dict(contacts=paths["contacts"], frags=paths["frags"], bias_path=paths["bias"], resolution=10000, n_bins=12, passes=2,
              mode="All", L=10000, U=900000, mapp_thres=1, tL=0.5, tU=2)
    for world, split in ((2, "by_chr"), (3, "by_chr")):
        _run_world(world, json.dumps(kw), 2, split, tmp_path)


# FHX_FUZZ_SEEDS="lo:hi": random cases of tests/test_gpu_fuzz.py's generator, rows dealt to 2 or 3 ranks at random (one rank may
# get nothing): every rank's p and q must equal the single-GPU run's bits.  Default: six seeds (two of each kind of loci).
_LO, _HI = (int(v) for v in os.environ.get("FHX_FUZZ_SEEDS", "0:6").split(":"))


@pytest.mark.parametrize("seed", range(_LO, _HI))
def test_native_sharded_pass_on_random_cases(seed, tmp_path):
    import json
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import test_gpu_fuzz as tf
    rng = np.random.default_rng(77000 + seed)
    # every third case with irregular midpoints under -r 0, every third with irregular midpoints under -r N (explicit distances)
    paths, kw, n_rows, _ = tf._make_case(rng, str(tmp_path), seed % 3 == 1, offgrid=seed % 3 == 2)
    if kw["mode"] == "interOnly" or n_rows < 8:
        pytest.skip("no spline pass / too few rows")
    from oracle import fithic_oracle as fo
    try:                                                # cases the reference refuses are the single-GPU fuzz test's business
        fo.run(paths["contacts"], paths["frags"], kw["bias_path"], kw["resolution"], kw["n_bins"], min(kw["passes"], 2), kw["mode"],
               kw["L"], kw["U"], kw["mapp_thres"], kw["tL"], kw["tU"])
    except (SystemExit, ZeroDivisionError, TypeError, ValueError, IndexError, KeyError):
        pytest.skip("the reference refuses this case")
    case = dict(kw, contacts=paths["contacts"], frags=paths["frags"], U=None if kw["U"] == float("inf") else kw["U"])
    world = 2 + seed % 2
    _run_world(world, json.dumps(case), min(kw["passes"], 2), "random:%d" % seed, tmp_path)


def _rccl_single_rank(case, passes, result_path):
    msgs = []
    try:
        from fithic_amd import _capi
        kw, chroms, con, frag, bias = _load_case(case)
        n = len(con)
        single = _make_ctx(kw, chroms, frag, bias, con, np.arange(n))
        want = _single_gpu_passes(single, n, passes)
        single.close()
        local = _make_ctx(kw, chroms, frag, bias, con, np.arange(n))
        os.environ["FHX_DIST_TRACE"] = "1"
        local.comm_init(_capi.comm_unique_id(), 0, 1)
        n_sorted = []
        rank, world, version = local.comm_info()
        if (rank, world) != (0, 1) or version <= 0:
            msgs.append("comm_info %r" % ((rank, world, version),))
        for pi in range(passes):
            local.run_pass_distributed()
            n_sorted.append(local.n_sorted())
            got = local.fetch(n)
            for key, ref in (("p", want[pi][0]), ("q", want[pi][1])):
                if not _same(got[key], ref):
                    msgs.append("pass %d: %s differs" % (pi + 1, key))
            if pi + 1 < passes and local.next_pass_distributed() != want[pi][2]:
                msgs.append("outlier totals differ")
        stages = local.dist_stage_seconds()
        if not all(v >= 0 for v in stages.values()):
            msgs.append("stage clocks %r" % (stages,))
        msgs += _schedule_errors(local, kw, 1, passes, n_sorted)
        local.close()
    except Exception as e:
        import traceback
        msgs.append("exception: %r\n%s" % (e, traceback.format_exc()))
    with open(result_path, "w") as f:
        f.write("OK" if not msgs else "FAIL: " + "; ".join(msgs))


@pytest.mark.parametrize("case,passes", [("f2_all", 2), ("f1_bias", 1), ("f8_nonfixed_all", 2)])
def test_rccl_transport_with_one_rank(case, passes, tmp_path):
    """Real RCCL, world size 1, in a fresh process without torch (the library loads the system's librccl itself)."""
    out = os.path.join(str(tmp_path), "rccl.txt")
    p = mp.get_context("spawn").Process(target=_rccl_single_rank, args=(case, passes, out))
    p.start()
    p.join(240)
    if p.is_alive():
        p.kill()
        pytest.fail("the rank hangs")
    assert p.exitcode == 0
    with open(out) as f:
        assert f.read() == "OK"
