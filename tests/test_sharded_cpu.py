"""CPU checks of the multi-rank plumbing of `fithic --gpus N` (fithic_amd/sharded.py): the pipe transport's exchange must not
deadlock on messages far larger than a pipe buffer (both ranks of a pair sending first would), for any world size."""
import multiprocessing as mp

import numpy as np
import pytest


def _rank(rank, world, conns, nbytes, q):
    from fithic_amd import sharded

    class Bare(sharded.PipeTransport):                # the exchange needs no GPU context
        def __init__(self):
            self.rank, self.world, self.conns = rank, world, conns

    data = [np.full(nbytes + r, (rank * 16 + r) % 251, np.uint8) for r in range(world)]
    got = Bare()._exchange(data)
    q.put((rank, [(int(g[0]), len(g)) for g in got]))


@pytest.mark.parametrize("world", [2, 3, 4])
def test_pipe_exchange_with_large_messages(world):
    from fithic_amd import sharded
    ctx = mp.get_context("spawn")
    mesh = sharded.make_mesh(ctx, world)
    q = ctx.Queue()
    nbytes = 3_000_000
    procs = [ctx.Process(target=_rank, args=(r, world, mesh[r], nbytes, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(30)
    for me in range(world):
        assert res[me] == [((src * 16 + me) % 251, nbytes + me) for src in range(world)]


def test_engine_that_cannot_start_leaves_no_worker_behind():
    """No GPU here: rank 0's own engine (and every worker's) fails to start.  The constructor must stop and join the worker
    processes it had already spawned before the error travels on - nobody will ever hold the object to close it."""
    from fithic_amd import sharded
    before = set(p.pid for p in mp.active_children())
    with pytest.raises(Exception):
        sharded.ShardedEngine(3, devices=[0, 1, 2], transport="pipes")
    left = [p for p in mp.active_children() if p.pid not in before]
    assert left == []


def _scripted_worker(conn, script):
    """a stand-in rank: answers every command as `script` says - ("ok", value), ("error", text), "hang" or "die" """
    import os
    import time
    while True:
        name, args = conn.recv()
        what = script.get(name, ("ok", None))
        if what == "hang":
            time.sleep(3600)
        if what == "die":
            os._exit(7)
        conn.send(what)
        if name == "close":
            return


def _scripted_engine(scripts, local):
    """A ShardedEngine around scripted ranks (no GPU, no library calls): scripts[r-1] drives worker r, `local` is rank 0."""
    from fithic_amd import sharded
    ctx = mp.get_context("fork")
    eng = object.__new__(sharded.ShardedEngine)
    eng.world, eng.transport, eng.broken, eng.local_stuck, eng.local = len(scripts) + 1, "pipes", None, False, local
    eng.workers = []
    for sc in scripts:
        parent, child = ctx.Pipe(duplex=True)
        p = ctx.Process(target=_scripted_worker, args=(child, sc), daemon=True)
        p.start()
        eng.workers.append((p, parent))
    eng.GRACE_S = 1.5
    return eng


class _Local:
    def __init__(self, **behaviour):
        self.behaviour = behaviour

    def __getattr__(self, name):
        def call(*args):
            what = self.behaviour.get(name, None)
            if what == "hang":
                import time
                time.sleep(3600)
            if isinstance(what, Exception):
                raise what
            return what
        return call


def test_a_rank_that_fails_before_a_collective_does_not_hang_the_others():
    """one rank returns an error before the first collective of a command while the others (rank 0 among
    them, whose part runs in this process) sit in it.  The engine must give up after the grace period - not wait for ever -
    name the failed rank, and say that this process's own rank is stuck (the CLI then leaves through os._exit)."""
    import time
    eng = _scripted_engine([{"pass_stats": ("error", "FhxError(-3, 'out of memory')")}, {"pass_stats": "hang"}], _Local(pass_stats="hang"))
    t0 = time.monotonic()
    with pytest.raises(RuntimeError) as e:
        eng._all("pass_stats")
    assert time.monotonic() - t0 < 20
    assert "rank 1 failed in pass_stats" in str(e.value) and "[0, 2]" in str(e.value)
    assert eng.local_stuck and eng.broken
    for p, _ in eng.workers:
        p.join(10)
        assert not p.is_alive()
    with pytest.raises(RuntimeError):                      # and it stays unusable
        eng._all("fit")
    eng.close()                                            # never raises


def test_a_failure_on_every_rank_keeps_the_protocol_in_step():
    """A refusal all ranks share (agreed over the communicator, or the same host-side error everywhere) is an ordinary
    exception: every answer is collected first, and the next command runs."""
    err = ("error", "FhxError(-4, 'a rank of this run holds loci that are not on one grid')")
    eng = _scripted_engine([{"pass_stats": err, "fit": ("ok", {"a": 1})}], _Local(pass_stats=ValueError("same refusal on rank 0"), fit={"a": 0}))
    with pytest.raises(ValueError):
        eng._all("pass_stats")
    assert not eng.broken and not eng.local_stuck
    assert eng._all("fit") == [{"a": 0}, {"a": 1}]
    eng.close()


def test_a_rank_that_dies_is_reported():
    eng = _scripted_engine([{"pvalues": "die"}], _Local(pvalues=None))
    with pytest.raises(RuntimeError) as e:
        eng._all("pvalues")
    assert "rank 1" in str(e.value) and eng.broken
    eng.close()


_MESH_RANK = r"""
import os, sys
import numpy as np
import torch.distributed as td
sys.path.insert(0, %r)
from fithic_amd import sharded
td.init_process_group("gloo")
rank, world = td.get_rank(), td.get_world_size()
mesh = sharded.socket_mesh(rank, world, os.environ["MASTER_PORT"], td.barrier)

class Bare(sharded.PipeTransport):
    def __init__(self):
        self.rank, self.world, self.conns = rank, world, mesh

got = Bare()._exchange([np.full(2_000_000 + r, (rank * 16 + r) %% 251, np.uint8) for r in range(world)])
assert [(int(g[0]), len(g)) for g in got] == [((src * 16 + rank) %% 251, 2_000_000 + rank) for src in range(world)], rank
td.barrier()
sys.stdout.write("mesh-ok-%%d\n" %% rank)
"""


@pytest.mark.parametrize("world", [2, 4])
def test_socket_mesh_between_ranks_of_a_launcher(world, tmp_path):
    """bench.py under FHX_BENCH_TRANSPORT=pipes: its ranks are started by torch.distributed.run and share no parent that could hand
    them pipes; the mesh is made of AF_UNIX sockets named after the rendezvous port, and PipeTransport runs over it unchanged."""
    import os
    import socket
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = tmp_path / "mesh_rank.py"
    script.write_text(_MESH_RANK % root)
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr",
                        "127.0.0.1", "--master-port", str(port), str(script)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-3000:]
    for k in range(world):                                   # (the ranks share one stdout: lines may interleave)
        assert r.stdout.count("mesh-ok-%d" % k) == 1, r.stdout
