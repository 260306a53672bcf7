"""GPU test of the distributed pass with the REAL per-GPU compute (LocalOps over the C ABI): two ranks share GPU 0,
transport = gloo staged through host memory (RCCL refuses two ranks on one device; the 8-GPU RCCL run is the
driver's).  Each rank's p and q must equal, bit for bit, what a single-GPU run over all rows gives for its rows."""
import os
import sys
import socket

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
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
    from fithic_amd import tables
    from fithic_amd.engine import Engine

    td.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    comm = dist.Comm(td, torch.device("cpu"), compute_device=torch.device("cuda", 0))
    meta, g = load_case(case)
    kw = case_args(meta)
    chroms = tables.ChromIndex()
    con = tables.read_contacts(kw["contacts"], chroms)
    frag = tables.read_fragments(kw["frags"], chroms)
    bias = tables.read_bias(kw["bias_path"], chroms) if kw["bias_path"] else None

    def make_engine(rows):
        e = Engine(0)
        e.configure(kw["resolution"], kw["L"], kw["U"], kw["n_bins"], kw["mapp_thres"], kw["mode"], kw["tL"], kw["tU"])
        e.load_fragments(*frag, chroms.sort_rank())
        if bias:
            e.load_bias(*bias)
        c = con.take(rows)
        e.load_contacts(c.chr1, c.mid1, c.chr2, c.mid2, c.count)
        return e

    mine = np.flatnonzero(con.chr1 % world == rank)
    single = make_engine(np.arange(len(con)))
    local = make_engine(mine)
    local.ctx.set_global_rows(mine)           # file positions of my rows (needed for the -p >= 3 semantics only)
    runner = dist.DistributedPass(local, comm)
    msgs = []
    for pi in range(passes):
        single.run_pass(collect=False)
        want = single.fetch()
        info = runner.run()
        got = local.fetch()
        for key in ("p", "q"):
            a, b = got[key], want[key][mine]
            same = (a.view(np.int64) == b.view(np.int64)) | (np.isnan(a) & np.isnan(b))
            if not same.all():
                msgs.append("pass %d: %s differs on %d of %d rows" % (pi + 1, key, (~same).sum(), len(same)))
        if pi + 1 < passes:
            t_single = single.next_pass()
            t_dist = runner.next_pass()
            if t_single != t_dist:
                msgs.append("outliers %d vs %d" % (t_dist, t_single))
    with open(os.path.join(result_dir, "rank%d.txt" % rank), "w") as f:
        f.write("OK" if not msgs else "FAIL: " + "; ".join(msgs))
    single.close()
    local.close()
    td.destroy_process_group()


@pytest.mark.parametrize("case,passes", [("f2_all", 2), ("f1_bias", 2), ("f6_quirk_all", 3), ("f8_nonfixed_all", 3), ("f8_nonfixed_hESC", 1)])
def test_two_ranks_one_gpu_bit_identical_to_single_gpu(case, passes, tmp_path):
    torch = pytest.importorskip("torch")
    import torch.multiprocessing as mp
    mp.spawn(_worker, args=(2, _free_port(), case, passes, str(tmp_path)), nprocs=2, join=True)
    for r in range(2):
        with open(os.path.join(str(tmp_path), "rank%d.txt" % r)) as f:
            assert f.read() == "OK"
