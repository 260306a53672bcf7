"""CPU checks of the multi-rank plumbing of `fithic --gpus N` (fithic_amd/sharded.py): the pipe transport's exchange must not
deadlock on messages far larger than a pipe buffer (both ranks of a pair sending first would), for any world size."""
import multiprocessing as mp
import os

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


class _ScriptedRanks:
    """stands in for a ShardedEngine's ranks: answers every command from a script and records what was asked"""

    def __init__(self, world, answers):
        self.world, self.answers, self.asked = world, answers, []
        self.segments = self._rows_of = self.split = None
        self.n_rows = 0

    def _ingest_stream_parts(self, path, threads):
        from fithic_amd import sharded
        return sharded.ShardedEngine._ingest_stream_parts(self, path, threads)

    def _whole_text_per_rank_fits(self, path):
        return getattr(self, "host_has_room", True)

    def _all(self, command, *args, per_rank=None):
        self.asked.append(command)
        if command == "ingest_text_own":
            self.own = per_rank
        if command == "commit_slice":
            self.committed = per_rank
            return [(self._n[r], [(0, per_rank[r][1], self._n[r])] if self._n[r] else []) for r in range(self.world)]
        if command in ("ingest_discard", "inflate_part_drop"):
            return [None] * self.world
        if command == "inflate_part" and command not in self.answers:
            return [("unsupported", "not one plain stream")] * self.world
        out = self.answers[command]
        self._n = [res[1] if res[0] == "ok" else 0 for res in out]
        return out


@pytest.mark.parametrize("case", ["tagged members", "plain gzip", "members end inside rows", "a part the parser refuses",
                                  "text outside the grammar"])
def test_how_the_ranks_of_the_cli_split_a_contacts_file(case):
    """ShardedEngine._ingest_slices, the decisions only (the ranks are scripted): parts of the FILE when every part parsed and all
    but the last end a row; else parts of the TEXT; None - on to the split by chromosome and the host parser - when that fails too.
    Names are interned in rank order (the file's order of first appearance), file positions are the prefix of the parts' rows."""
    from fithic_amd import sharded, tables
    ok_file = [("ok", 5, ["chr2", "chr1"], True), ("ok", 0, [], True), ("ok", 7, ["chr1", "chr3"], False)]
    ok_text = [("ok", 4, ["chr2"], True), ("ok", 6, ["chr1", "chr2"], True), ("ok", 2, ["chr3"], True)]
    refused = [("unsupported", "line 3 has four fields")] * 3
    script = {
        "tagged members": {"ingest_slice": ok_file},
        "plain gzip": {"ingest_slice": [("container", "no member sizes")] * 3, "ingest_text_slice": ok_text},
        "members end inside rows": {"ingest_slice": [("ok", 5, ["chr2"], False), ("ok", 3, ["chr1"], True), ("ok", 4, ["chr3"], True)],
                                    "ingest_text_slice": ok_text},
        "a part the parser refuses": {"ingest_slice": [ok_file[0], refused[0], ok_file[2]], "ingest_text_slice": ok_text},
        "text outside the grammar": {"ingest_slice": refused, "ingest_text_slice": refused},
    }[case]
    ranks = _ScriptedRanks(3, script)
    chroms = tables.ChromIndex()
    chroms.intern("chr1")                                     # (the fragments file was read first)
    got = sharded.ShardedEngine._ingest_slices(ranks, "contacts.gz", chroms, 4)
    if case == "tagged members":
        assert ranks.asked == ["ingest_slice", "commit_slice"] and ranks.split == "file"
        assert chroms.names == ["chr1", "chr2", "chr3"]
        assert [list(ids) for ids, _ in ranks.committed] == [[1, 0], [], [0, 2]]
        assert [first for _, first in ranks.committed] == [0, 5, 5]
        assert len(got) == 12 and ranks.segments == [[(0, 0, 5)], [], [(0, 5, 7)]]
    elif case == "text outside the grammar":
        assert got is None and ranks.asked == ["ingest_slice", "ingest_discard", "inflate_part", "inflate_part_drop", "ingest_text_slice",
                                               "ingest_discard"]
        assert chroms.names == ["chr1"]                       # nothing interned by the attempts
    else:
        assert ranks.asked == ["ingest_slice", "ingest_discard", "inflate_part", "inflate_part_drop", "ingest_text_slice",
                               "commit_slice"] and ranks.split == "text"
        assert chroms.names == ["chr1", "chr2", "chr3"]
        assert [list(ids) for ids, _ in ranks.committed] == [[1], [0, 1], [2]]
        assert [first for _, first in ranks.committed] == [0, 4, 10]
        assert len(got) == 12 and ranks.segments == [[(0, 0, 4)], [(0, 4, 6)], [(0, 10, 2)]]


class _FileRanks:
    """ranks that hold ready-made parts of an output file; place / drop are the real _Rank methods"""

    def __init__(self, tmp_path, name, fail_place_on=None):
        self.world, self.name, self.fail = 3, name, fail_place_on
        self.segments = [[(0, 0, 2)], [(0, 2, 1), (1, 5, 1)], [(0, 3, 2)]]
        self.local = self
        self.eng = self
        self.ctx = self

    def write_significances_range(self, path, chr_names, a, b, header):
        with open(path, "wb") as f:
            f.write(b"HEAD;")

    def _all(self, command, *args, per_rank=None):
        from fithic_amd import sharded
        out = []
        for r in range(self.world):
            a = per_rank[r] if per_rank is not None else args
            if command == "write_parts":
                parts = []
                for _, file_start, length in self.segments[r]:
                    path = "%s.part-%015d" % (self.name, file_start)
                    with open(path, "wb") as f:
                        f.write(b"<%d+%d>" % (file_start, length))
                    parts.append((file_start, path, os.path.getsize(path)))
                out.append(("ok", parts))
            elif command == "place_parts" and r == self.fail:
                raise OSError(28, "No space left on device")
            else:
                out.append(getattr(sharded._Rank, command)(None, *a))
        return out


@pytest.mark.parametrize("fail", [None, 1])
def test_parts_of_the_output_are_dropped_only_after_every_rank_placed_its_own(tmp_path, fail):
    """_CtxFacade.write_significances_device: header + the ranks' parts in file order; on success and on a failure half way nothing
    but the output (or nothing at all) stays next to it."""
    from fithic_amd import sharded
    name = str(tmp_path / "out.significances.txt.gz")
    ranks = _FileRanks(tmp_path, name, fail)
    facade = sharded._CtxFacade(ranks)
    if fail is None:
        facade.write_significances_device(name, ["chr1"])
        assert open(name, "rb").read() == b"HEAD;<0+2><2+1><3+2><5+1>"
        assert os.listdir(tmp_path) == ["out.significances.txt.gz"]
    else:
        with pytest.raises(OSError):
            facade.write_significances_device(name, ["chr1"])
        assert os.listdir(tmp_path) == []


def test_one_plain_gzip_stream_is_inflated_by_the_ranks_together(tmp_path):
    """ShardedEngine._ingest_stream_parts, the decisions only (scripted ranks, a real trailer): the tails are chained into windows,
    the parts' CRC-32s and lengths must give the file's trailer, a rank's text starts after the rest of the row the rank before it
    owns and ends with the head of the next rank's text; anything that does not add up -> the whole-text route, and that only when
    N copies of the text fit the host."""
    import gzip
    import zlib
    from fithic_amd import sharded, tables
    parts = [b"chr1\t5\tchr1\t15\t2\nchr1\t5\tch", b"r1\t25\t1\nchr2\t5\tchr2\t15\t7\n", b"chr2\t5\tchr2\t35\t1\nchr3\t5\tchr3\t15\t1\n"]
    path = str(tmp_path / "c.gz")
    with open(path, "wb") as f:
        f.write(gzip.compress(b"".join(parts)))
    tail = np.arange(256, 256 + 32768, dtype=np.uint16).tobytes()

    def head(b):
        k = b.find(b"\n")
        return (k + 1, b[:k + 1]) if k >= 0 else (-1, b"")

    first = [("ok", tail, len(b), r == 2) for r, b in enumerate(parts)]
    second = [("ok", zlib.crc32(b), len(b)) + head(b) + (b.endswith(b"\n"),) for b in parts]
    own = [("ok", 1, ["chr1"], True), ("ok", 1, ["chr2"], True), ("ok", 2, ["chr2", "chr3"], True)]
    ranks = _ScriptedRanks(3, {"ingest_slice": [("container", "no member sizes")] * 3, "inflate_part": first,
                               "inflate_part_resolve": second, "ingest_text_own": own})
    chroms = tables.ChromIndex()
    got = sharded.ShardedEngine._ingest_slices(ranks, path, chroms, 4)
    assert ranks.asked == ["ingest_slice", "ingest_discard", "inflate_part", "inflate_part_resolve", "ingest_text_own", "commit_slice"]
    assert ranks.split == "stream" and len(got) == 4 and chroms.names == ["chr1", "chr2", "chr3"]
    # rank 0 takes its text from the start and the rest of its last row from rank 1; rank 1 starts behind that rest; rank 2 at 0
    assert [(sk, ex) for sk, ex, _ in ranks.own] == [(0, b"r1\t25\t1\n"), (len(b"r1\t25\t1\n"), b""), (0, b"")]
    # a part whose CRC-32 is another one: nothing of the stream route is kept, every rank inflates the whole file
    wrong = list(second)
    wrong[1] = ("ok", second[1][1] ^ 1) + second[1][2:]
    text3 = [("ok", 1, ["chr1"], True), ("ok", 1, ["chr2"], True), ("ok", 2, ["chr2", "chr3"], True)]
    ranks = _ScriptedRanks(3, {"ingest_slice": [("container", "no member sizes")] * 3, "inflate_part": first,
                               "inflate_part_resolve": wrong, "ingest_text_slice": text3})
    got = sharded.ShardedEngine._ingest_slices(ranks, path, tables.ChromIndex(), 4)
    assert ranks.asked == ["ingest_slice", "ingest_discard", "inflate_part", "inflate_part_resolve", "inflate_part_drop", "ingest_text_slice",
                           "commit_slice"] and ranks.split == "text" and len(got) == 4
    # ... unless N copies of the text would not fit the host: then not at all (the caller's funnel holds the text once)
    ranks = _ScriptedRanks(3, {"ingest_slice": [("container", "no member sizes")] * 3, "inflate_part": first, "inflate_part_resolve": wrong})
    ranks.host_has_room = False
    assert sharded.ShardedEngine._ingest_slices(ranks, path, tables.ChromIndex(), 4) is None
    assert ranks.asked[-1] == "inflate_part_drop"


def test_a_row_belongs_to_the_part_that_holds_its_first_byte():
    """sharded.rows_owned_by_parts on texts cut at random bytes: skipping / appending as it says and splitting every part into rows
    gives back the rows of the whole text, each once, in order"""
    from fithic_amd import sharded
    rng = np.random.default_rng(11)
    for trial in range(300):
        rows = [b"r%d\t%d\n" % (k, rng.integers(0, 10 ** int(rng.integers(1, 9)))) for k in range(int(rng.integers(3, 60)))]
        text = b"".join(rows)
        if trial % 3 == 0:
            text = text[:-1]
        cuts = sorted(set(int(c) for c in rng.integers(1, len(text) - 1, int(rng.integers(1, 6)))))
        parts = [text[a:b] for a, b in zip([0] + cuts, cuts + [len(text)])]
        if any(b"\n" not in p for p in parts):
            continue
        firsts = [(p.index(b"\n") + 1, p[:p.index(b"\n") + 1]) for p in parts]
        own = sharded.rows_owned_by_parts(firsts, [p.endswith(b"\n") for p in parts])
        got = []
        for p, (skip, extra) in zip(parts, own):
            mine = p[skip:] + extra
            got += [ln + b"\n" for ln in mine.split(b"\n") if ln] if mine else []
        want = [ln + b"\n" for ln in text.split(b"\n") if ln]
        assert got == want, (trial, cuts)
