"""bench.py itself on the GPU: every configuration that prints a `value` carries a parity verdict.  Small workloads (the first
chromosomes of C3) so that the two runs take seconds; the full-size checks live in tests/test_gpu_scale.py."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(extra_env, *flags):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
               HSA_ENABLE_IPC_MODE_LEGACY="0", **extra_env)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1"] + list(flags),
                       capture_output=True, text=True, env=env, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout[-2000:]               # exactly one JSON line on stdout
    return json.loads(lines[0])


def test_single_gpu_line_is_checked_against_the_oracle():
    out = _run({}, "--max-chroms", "2", "--no-cpu-baseline")
    pc = out["parity_check"]
    assert pc["ok"] and pc["max_dp"] <= 1e-10 and pc["max_dq"] == 0.0 and pc["rows_q"] == out["config"]["pairs"]
    assert "fit_vs_reference" not in pc                     # two chromosomes are not the workload the f14 fixture was made on
    assert out["n_gpus"] == 1 and out["roofline"]["frac"] > 0 and out["value"] > 1e8
    ks = out["k3_stress"]                                   # three more, labelled workloads: 12 %, 39 %, 55 % of the rows below the cutoff
    assert len(ks) == 3 and [k["survivor_fraction"] > f for k, f in zip(ks, (0.08, 0.3, 0.45))] == [True] * 3
    for k in ks:                                            # each with a verdict of its own
        pc = k["parity_check"]
        assert pc["ok"] and pc["max_dp"] <= 1e-10 and pc["max_dq"] == 0.0 and pc["rows_q"] == k["pairs"], k["workload"]
        assert k["rows_sorted"] > out["bh_rows_sorted_rank0"] and k["k3_ms"] > 0 and k["sort"]["passes"] == 5
    assert sum(out["k2_class_rows_rank0"].values()) < out["config"]["pairs"]


def test_sharded_schedule_over_rccl_is_verified_against_one_gpu():
    """FHX_FORCE_DIST=1: the library's communicator (real RCCL, one rank) and fhx_run_pass_distributed carry the timed pass; the
    same code then verifies it as an N > 1 run is verified - per-chromosome hashes of (row, p) and (row, q) of the sharded run
    against a plain single-GPU pass over all rows, which in turn is checked against the oracle."""
    out = _run({"FHX_FORCE_DIST": "1"}, "--max-chroms", "3", "--no-weak")
    pc = out["parity_check"]
    assert pc["ok"] and pc["sharded_equals_single_gpu"] and pc["ranks"] == 1 and pc["chromosomes_hashed"] == 3
    assert pc["chromosomes_p_differ"] == [] and pc["chromosomes_q_differ"] == [] and pc["global_stats_equal"] and pc["fit_scalars_equal"]
    assert pc["rows"] == out["config"]["pairs"] and pc["max_dp"] <= 1e-10 and pc["max_dq"] == 0.0
    assert out["rccl"]["world_in_library"] == 1 and "library communicator" in out["rccl"]["driver"]


def _plain(extra_env, *flags, expect_rc=0):
    """bench.py as the driver starts it for one GPU: no launcher environment at all"""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env.update(extra_env)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1"] + list(flags),
                       capture_output=True, text=True, env=env, timeout=900)
    assert r.returncode == expect_rc, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout[-2000:]
    return json.loads(lines[0])


def test_plain_invocation_starts_its_own_ranks():
    """`python bench.py --gpus N` without torch.distributed.run: the script launches the ranks itself (here N = 1 with the
    sharded schedule forced) and the one line carries everything an N > 1 line must carry."""
    out = _plain({"FHX_FORCE_DIST": "1"}, "--gpus", "1", "--max-chroms", "3", "--no-weak")
    assert out["value"] > 1e8 and out["scaling"] == "strong" and out["n_gpus"] == 1
    assert out["rccl"]["world_in_library"] == 1
    assert len(out["per_rank"]) == 1 and out["per_rank"][0]["rows"] == out["config"]["pairs"]
    assert all(out["per_rank"][0][k] > 0 for k in ("k1_ms", "k2_ms", "k3_ms"))
    assert len(out["stage_ms"]) == 7           # the five stages of the C call + the call as a whole + the bookkeeping behind it
    assert out["parity_check"]["ok"] and out["parity_check"]["sharded_equals_single_gpu"]
    assert 0.3 < out["strong_efficiency"] < 1.5 and out["single_gpu_ms_per_step"] > 0


def test_more_gpus_than_the_node_has_yields_one_diagnostic_line():
    import torch
    n = torch.cuda.device_count()
    out = _plain({}, "--gpus", str(n + 1), expect_rc=2)
    assert out["value"] is None and out["visible_devices"] == n and out["n_gpus"] == n + 1 and "visible" in out["error"]


def test_single_gpu_line_says_strong_scaling():
    out = _plain({}, "--max-chroms", "2", "--no-cpu-baseline", "--no-parity-check", "--no-k3-stress")
    assert out["scaling"] == "strong" and out["n_gpus"] == 1 and out["value"] > 1e8 and "k3_stress" not in out


def test_shard_of_runs_the_fullest_rank_of_an_n_way_sharding():
    """--shard-of N (profiles/scaling_model.py): the chromosomes the fullest of N ranks would hold, alone on this GPU."""
    import bench
    from fithic_amd import synth
    g = synth.Genome(5000, synth.HG19_AUTOSOMES[:6])
    owner = synth.assign_chromosomes(g, 3)
    loads = [sum(g.n_loci[c] for c in range(6) if owner[c] == r) for r in range(3)]
    out = _plain({}, "--max-chroms", "6", "--shard-of", "3", "--no-cpu-baseline", "--no-k3-stress")
    assert out["parity_check"]["ok"] and "fit_vs_reference" not in out["parity_check"]
    whole = _plain({}, "--max-chroms", "6", "--no-cpu-baseline", "--no-parity-check", "--no-k3-stress")
    frac = out["config"]["pairs"] / whole["config"]["pairs"]
    assert abs(frac - max(loads) / sum(loads)) < 0.03


@pytest.mark.parametrize("world, chroms", [(2, 3), (4, 5), (8, 8)])
def test_n_ranks_branch_of_bench_runs_end_to_end_over_the_pipe_transport(world, chroms):
    """bench.py's whole N > 1 branch with N > 1: `python bench.py --gpus N` starts N ranks that share this
    box's GPU; torch.distributed runs on gloo and the library's collectives go through fhx_comm_init_custom + the pipe transport
    of `fithic --gpus N` (RCCL refuses two ranks on one device).  Everything else is what an RCCL run executes: NativeRunner
    under a world > 1 communicator, the per-rank hash tables gathered and summed, rank 0's solo verification while the others
    wait, per_rank, predicted_ms, strong_efficiency, the weak-scaling leg - and one JSON line, whose `value` is null because
    nothing in it measures xGMI."""
    # (eight ranks: the weak-scaling leg - eight genomes on this one GPU - is left to the 2- and 4-rank cases)
    extra = ("--no-weak",) if world == 8 else ()
    out = _plain({"FHX_BENCH_TRANSPORT": "pipes"}, "--gpus", str(world), "--max-chroms", str(chroms), *extra)
    assert out["n_gpus"] == world and out["scaling"] == "strong"
    assert out["value"] is None and out["value_over_pipes"] > 1e6 and "pipes" in out["value_note"]
    assert out["rccl"]["transport"] == "pipes" and out["rccl"]["backend"] == "gloo" and out["rccl"]["world"] == world
    assert out["rccl"]["world_in_library"] == world
    pc = out["parity_check"]
    assert pc["ok"] and pc["sharded_equals_single_gpu"] and pc["ranks"] == world and pc["chromosomes_hashed"] == chroms
    assert pc["chromosomes_p_differ"] == [] and pc["chromosomes_q_differ"] == [] and pc["global_stats_equal"] and pc["fit_scalars_equal"]
    assert pc["rows"] == out["config"]["pairs"] and pc["max_dp"] <= 1e-10 and pc["max_dq"] == 0.0
    ranks = out["per_rank"]
    assert len(ranks) == world and [r["rank"] for r in ranks] == list(range(world))
    assert sum(r["rows"] for r in ranks) == out["config"]["pairs"]
    assert sum(1 for r in ranks if r["rows"] > 0) == min(world, chroms)        # a chromosome's rows live on one rank
    assert all(r["k1_ms"] > 0 and r["k2_ms"] > 0 for r in ranks if r["rows"] > 0)
    assert len(out["stage_ms"]) == 7 and out["single_gpu_ms_per_step"] > 0 and out["strong_efficiency"] > 0
    assert out["predicted_ms"] is None or out["predicted_ms"]["ms"] > 0
    if world == 8:
        assert "weak_scaling" not in out
        return
    w = out["weak_scaling"]
    # (every replica of the genome draws its own rows: N x the pairs up to sampling noise)
    assert w["replicas"] == world and abs(w["pairs"] / (world * out["config"]["pairs"]) - 1.0) < 0.01 and w["value"] > 1e6
