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
