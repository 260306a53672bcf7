"""`fithic --gpus N`: the reference's one-process run (fithic/fithic.py:317-370) over N MI355X of one node.

Rank 0 is the process the user started; N-1 worker processes (one per GPU) take commands from it.  `ShardedEngine` offers the
subset of `Engine` the stage functions of fithic_amd.fithic use, so the same functions write the same files: every stage is
forwarded to all ranks, the genome-wide steps go through the library's communicator (fhx_pass_stats_distributed: K1 + one
all-reduce; fhx_bh_distributed: global ranking; fhx_next_pass_distributed: outlier multiset).

Where the contact rows come from and where the significances go (fithic.py:404-417, 1167-1219 - half of the reference's time):
  * default: EVERY RANK reads the contacts file itself on its own GPU and later formats + deflates its own stretches of the
    output file, which the ranks copy into place side by side.  No row passes through rank 0 (ShardedEngine.ingest_file,
    _CtxFacade.write_significances_device).  A file of gzip members that carry their sizes is cut into N parts of whole
    members - rank r inflates and parses part r only, its rows are one stretch of the file (_ingest_slices); ONE plain gzip
    stream (what `gzip` writes) is inflated by the ranks together, rank r its N-th of the compressed bytes (_ingest_stream_parts);
    any other gzip file (bgzip, several plain members) is inflated by every rank on the host - when N copies of the text fit its
    memory - and rank r uploads and parses the rows that start in the r-th N-th of the text.
    (FHX_CLI_SPLIT=chromosome: every rank parses the whole file and keeps the rows whose first chromosome is its own - greedy
    owner map by row count, the same on every rank because they count the same file);
  * a file the device parser does not take, or one split by chromosome without being sorted by chromosome (a rank's rows would
    be more than a thousand separate stretches of the output), or FHX_CLI_FUNNEL=1: rank 0 parses on the host cores, hands the columns out
    over the pipes (load_contacts) and gathers p, q, ExpCC and the biases back for the one host writer (fetch).

Transports: "rccl" (default; fhx_comm_init, RCCL over xGMI) or "pipes" (fhx_comm_init_custom with the collectives
below, staged through host memory - for boxes where several ranks must share one GPU, which RCCL refuses; used by the
tests).  No compute happens here: this module moves tables and commands.
"""
import multiprocessing as mp
from multiprocessing.connection import wait as mp_wait
import os
import sys
import threading
import time

import numpy as np

from . import _capi
from .engine import Engine, MODES, TOTALS


class PipeTransport:
    """all_reduce / all_gather / all_to_all_v on device buffers over a full mesh of pipes (fhx_transport callbacks)."""

    def __init__(self, ctx, rank, world, conns):
        self.ctx, self.rank, self.world, self.conns = ctx, rank, world, conns
        T = _capi.FhxTransport
        self.struct = T(None, T.ALL_REDUCE(self.all_reduce), T.ALL_GATHER(self.all_gather), T.ALL_TO_ALL_V(self.all_to_all_v))

    def _d2h(self, ptr, nbytes):
        a = np.empty(nbytes, np.uint8)
        if nbytes:
            self.ctx.copy(a.ctypes.data, ptr, nbytes, 1)
        return a

    def _h2d(self, ptr, a):
        if a.nbytes:
            a = np.ascontiguousarray(a)
            self.ctx.copy(ptr, a.ctypes.data, a.nbytes, 0)

    def _exchange(self, per_peer):
        """per_peer[r] = bytes for rank r; returns what every rank sent to this one (its own entry passes through).
        Pairwise in rank order - the lower rank of a pair sends first, the higher receives first - so two ranks never sit
        in send() on full pipe buffers at the same time (messages here reach hundreds of megabytes)."""
        got = [None] * self.world
        got[self.rank] = per_peer[self.rank]
        for r in range(self.world):
            if r == self.rank:
                continue
            if self.rank < r:
                self.conns[r].send_bytes(per_peer[r].tobytes())
                got[r] = np.frombuffer(self.conns[r].recv_bytes(), np.uint8)
            else:
                got[r] = np.frombuffer(self.conns[r].recv_bytes(), np.uint8)
                self.conns[r].send_bytes(per_peer[r].tobytes())
        return got

    def all_reduce(self, user, d_buf, n, op):
        try:                                          # an exception must not unwind through the C caller
            mine = self._d2h(d_buf, 8 * n)
            parts = [g.view(np.int64) for g in self._exchange([mine] * self.world)]
            red = {0: np.sum, 1: np.max, 2: np.min}[op](np.stack(parts), axis=0).astype(np.int64)
            self._h2d(d_buf, red)
            return 0
        except Exception as e:
            sys.stderr.write("transport all_reduce: %r\n" % (e,))
            return 1

    def all_gather(self, user, d_send, d_recv, nbytes):
        try:
            mine = self._d2h(d_send, nbytes)
            self._h2d(d_recv, np.concatenate(self._exchange([mine] * self.world)))
            return 0
        except Exception as e:
            sys.stderr.write("transport all_gather: %r\n" % (e,))
            return 1

    def all_to_all_v(self, user, d_send, sc, so, d_recv, rc, ro, elem):
        try:
            w = self.world
            sc, so, rc, ro = ([int(v[r]) for r in range(w)] for v in (sc, so, rc, ro))
            total = max((so[r] + sc[r] for r in range(w)), default=0)
            send = self._d2h(d_send, total * elem)
            got = self._exchange([send[so[r] * elem:(so[r] + sc[r]) * elem] for r in range(w)])
            for r in range(w):
                if len(got[r]) != rc[r] * elem:
                    raise RuntimeError("rank %d sent %d bytes, expected %d" % (r, len(got[r]), rc[r] * elem))
                if rc[r]:
                    self._h2d(d_recv + ro[r] * elem, got[r])
            return 0
        except Exception as e:
            sys.stderr.write("transport all_to_all_v: %r\n" % (e,))
            return 1


def make_mesh(ctxmp, world):
    """conns[r][peer] = this rank's end of the pipe to `peer`."""
    conns = [dict() for _ in range(world)]
    for a in range(world):
        for b in range(a + 1, world):
            ca, cb = ctxmp.Pipe(duplex=True)
            conns[a][b] = ca
            conns[b][a] = cb
    return conns


def socket_mesh(rank, world, tag, barrier, directory=None):
    """The same full mesh between processes that do NOT share a parent (bench.py's ranks are started by torch.distributed.run):
    every rank listens on an AF_UNIX socket named after `tag` (the rendezvous port: unique per launch), `barrier()` (the
    launcher's own process group) separates "all listen" from "all connect", the higher rank of a pair connects to the lower
    and says who it is.  Returns {peer: Connection} - what PipeTransport takes."""
    import tempfile
    from multiprocessing.connection import Client, Listener
    directory = directory or tempfile.gettempdir()
    addr = lambda r: os.path.join(directory, "fhx_mesh_%s_%d.sock" % (tag, r))
    if os.path.exists(addr(rank)):
        os.unlink(addr(rank))
    listener = Listener(addr(rank), family="AF_UNIX", backlog=max(world, 1))
    conns = {}
    try:
        barrier()
        for peer in range(rank):                           # lower ranks listen for me
            c = Client(addr(peer), family="AF_UNIX")
            c.send(rank)
            conns[peer] = c
        for _ in range(rank + 1, world):                   # higher ranks come to me
            c = listener.accept()
            conns[int(c.recv())] = c
        barrier()
    finally:
        listener.close()                                   # unlinks the socket file
    return conns


def rows_owned_by_parts(first_rows, ends_with_newline):
    """A text cut into consecutive parts anywhere: a row belongs to the part that holds its first byte.  first_rows[r] = (length of
    part r's first row with its newline, that row) - every part must hold a newline -, ends_with_newline[r]: part r ends a row.
    -> per part (bytes to skip at its start: the rest of a row the part before owns; bytes to append: the rest of its last row,
    which is the head of the next part)."""
    out = []
    last = len(first_rows) - 1
    for r, (n, row) in enumerate(first_rows):
        skip = 0 if r == 0 or ends_with_newline[r - 1] else n
        extra = b"" if r == last or ends_with_newline[r] else first_rows[r + 1][1]
        out.append((skip, extra))
    return out


class _Rank:
    """One rank's engine + communicator; the same object serves rank 0 (in process) and the workers' command loop."""

    def __init__(self, rank, world, device, transport, unique_id, mesh):
        self.rank, self.world = rank, world
        self.eng = Engine(device)
        self.transport, self.unique_id, self.mesh = transport, unique_id, mesh
        self.comm_ready = False
        self.n_rows = 0
        self.segments = None

    def _ensure_comm(self):
        if self.comm_ready:
            return
        if self.transport == "pipes":
            self._pt = PipeTransport(self.eng.ctx, self.rank, self.world, self.mesh)
            self.eng.ctx.comm_init_custom(self._pt.struct, self.rank, self.world)
        else:
            self.eng.ctx.comm_init(self.unique_id, self.rank, self.world)
        self.comm_ready = True

    # every method below is a command; the return value travels back to rank 0
    def configure(self, *a):
        self.eng.configure(*a)

    def load_fragments(self, *a):
        self.eng.load_fragments(*a)

    def load_bias(self, *a):
        self.eng.load_bias(*a)

    def load_contacts(self, c1, m1, c2, m2, cnt, rows):
        self.eng.load_contacts(c1, m1, c2, m2, cnt)
        self.eng.ctx.set_global_rows(rows)               # file positions: the -p >= 3 semantics
        self.n_rows = len(rows)
        self._ensure_comm()

    # ---- the contacts file read by every rank itself (no rows through rank 0) ----
    def ingest_file(self, path, threads, host_text_allowed=True):
        """inflate + parse the whole file on this rank's GPU -> ("ok", rows, names, rows per name) or ("unsupported", why)"""
        ctx = self.eng.ctx
        text = None
        try:
            try:
                n, names = ctx.ingest_contacts_file(path, threads)
            except _capi.FhxError as e:
                if e.code != _capi.FHX_ERR_UNSUPPORTED:
                    raise
                if getattr(e, "refused", 1) == 2:
                    return ("unsupported", str(e))
                if not host_text_allowed:
                    return ("unsupported", "every rank would hold the whole inflated text in host memory")
                text = _capi.HostText(path, threads)
                try:
                    n, names = ctx.ingest_contacts_text(text, threads)
                except _capi.FhxError as e2:
                    if e2.code != _capi.FHX_ERR_UNSUPPORTED:
                        raise
                    return ("unsupported", str(e2))
            return ("ok", n, names, ctx.ingest_contacts_chr_counts(len(names)))
        finally:
            if text is not None:
                text.close()

    def ingest_discard(self):
        self.eng.ctx.ingest_contacts_discard()

    def ingest_slice(self, path, threads):
        """inflate + parse this rank's part of the file (whole gzip members) -> ("ok", rows, names, ends with a newline) or
        ("unsupported", why)"""
        try:
            n, names, newline = self.eng.ctx.ingest_contacts_file_slice(path, self.rank, self.world, threads)
        except _capi.FhxError as e:
            if e.code != _capi.FHX_ERR_UNSUPPORTED:
                raise
            return ("container" if getattr(e, "refused", 2) == 1 else "unsupported", str(e))
        self._parsed = int(n)
        return ("ok", n, names, newline)

    def ingest_text_slice(self, path, threads):
        """a file the device does not inflate (plain gzip): inflated whole on this rank's share of the host cores, then the rows
        that start in this rank's N-th of the text uploaded and parsed -> as ingest_slice"""
        self._note_memory()
        text = _capi.HostText(path, threads)
        try:
            n, names = self.eng.ctx.ingest_contacts_text_slice(text, self.rank, self.world, threads)
        except _capi.FhxError as e:
            if e.code != _capi.FHX_ERR_UNSUPPORTED:
                raise
            return ("unsupported", str(e))
        finally:
            text.close()
            self._report_memory("whole text")
        self._parsed = int(n)
        return ("ok", n, names, True)

    # ---- ONE plain gzip stream inflated by all ranks together: rank r decodes its N-th of the compressed bytes (TextPart) ----
    def inflate_part(self, path, threads):
        """first call: this rank's chunks decoded without the window before them -> ("ok", tail symbols, bytes of text, holds the
        end of the stream) or ("unsupported", why)"""
        self.inflate_part_drop()
        self._note_memory()
        try:
            self._part = _capi.TextPart(path, self.rank, self.world, threads)
        except _capi.FhxError as e:
            if e.code != _capi.FHX_ERR_UNSUPPORTED:
                raise
            return ("unsupported", str(e))
        return ("ok", self._part.tail().tobytes(), len(self._part), self._part.is_last())

    def inflate_part_resolve(self, window):
        """second call, with the 32 KB of text before this part -> ("ok", CRC-32, bytes, length of the first (partial) row or -1,
        that row, the text ends with a newline) or ("unsupported", why)"""
        try:
            self._text, crc = self._part.resolve(np.frombuffer(window, np.uint8))
        except _capi.FhxError as e:
            if e.code != _capi.FHX_ERR_UNSUPPORTED:
                raise
            return ("unsupported", str(e))
        finally:
            self._part.close()
            self._part = None
        n, row = self._text.first_row_end()
        if n > (1 << 20):
            n, row = -1, b""                                   # (no contact row is a megabyte long: not a table of rows)
        return ("ok", crc, len(self._text), n, row, self._text.ends_with_newline())

    def inflate_part_drop(self):
        for name in ("_part", "_text"):
            obj = getattr(self, name, None)
            if obj is not None:
                obj.close()
            setattr(self, name, None)

    def ingest_text_own(self, skip, extra, threads):
        """the rows whose first byte lies in this rank's part of the text (from byte `skip` on, the last one completed by `extra`)
        uploaded and parsed -> as ingest_slice"""
        try:
            n, names = self.eng.ctx.ingest_contacts_text_own(self._text, skip, extra, threads)
        except _capi.FhxError as e:
            if e.code != _capi.FHX_ERR_UNSUPPORTED:
                raise
            return ("unsupported", str(e))
        finally:
            self.inflate_part_drop()
            self._report_memory("part of the stream")
        self._parsed = int(n)
        return ("ok", n, names, True)

    def _report_memory(self, what):
        if os.environ.get("FHX_TIMING"):
            import resource
            sys.stderr.write("rank %d of %d (%s): peak host RSS %.0f MB (%.0f MB before the file was opened)\n" %
                             (self.rank, self.world, what, resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 1024.0, getattr(self, "_rss_before", 0.0)))

    def _note_memory(self):
        if os.environ.get("FHX_TIMING") and not hasattr(self, "_rss_before"):
            import resource
            self._rss_before = resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 1024.0

    def commit_slice(self, ids, first):
        """the parsed part becomes this rank's rows, at file positions first, first + 1, ..."""
        ctx = self.eng.ctx
        ctx.ingest_contacts_commit(ids)
        self.eng.n_rows = self.n_rows = self._parsed
        self.eng.pass_no = 0
        ctx.set_global_rows_range(first)
        self.segments = [(0, int(first), self.n_rows)] if self.n_rows else []
        self._ensure_comm()
        return self.n_rows, self.segments

    def commit_shard(self, ids, mine):
        """keep the rows whose first chromosome is this rank's -> (rows kept, [(local start, file position, length)] or None)"""
        n = self.eng.ctx.ingest_contacts_commit_shard(ids, mine)
        self.eng.n_rows = int(n)
        self.eng.pass_no = 0
        self.n_rows = int(n)
        self.segments = self.eng.ctx.shard_segments()
        self._ensure_comm()
        return self.n_rows, self.segments

    def write_parts(self, name, chr_names):
        """this rank's stretches of the significances file, one part file each -> ("ok", [(file position, path, bytes)]) or
        ("unsupported", why): then nothing is left behind"""
        parts = []
        try:
            for local_start, file_start, length in self.segments:
                path = "%s.part-%015d" % (name, file_start)
                self.eng.ctx.write_significances_range(path, chr_names, local_start, local_start + length, False)
                parts.append((file_start, path, os.path.getsize(path)))
        except BaseException as e:                             # nothing is left behind, whatever went wrong
            for _, path, _ in parts:
                if os.path.exists(path):
                    os.unlink(path)
            if isinstance(e, _capi.FhxError) and e.code == _capi.FHX_ERR_UNSUPPORTED:
                return ("unsupported", str(e))
            raise
        return ("ok", parts)

    def place_parts(self, target, placements):
        """copy this rank's parts into `target` at their offsets (all ranks at once: the file is put together in parallel).  The
        parts stay where they are: the caller drops them once EVERY rank has placed its own (or after a failure)."""
        fd = os.open(target, os.O_WRONLY)
        try:
            for path, offset in placements:
                with open(path, "rb") as src:
                    size = os.fstat(src.fileno()).st_size
                    done = 0
                    while done < size:
                        try:
                            k = os.copy_file_range(src.fileno(), fd, size - done, done, offset + done)
                        except OSError:                          # file systems without it
                            k = os.pwrite(fd, os.pread(src.fileno(), min(size - done, 64 << 20), done), offset + done)
                        if k <= 0:
                            raise OSError("short copy of %s" % path)
                        done += k
        finally:
            os.close(fd)

    def drop_parts(self, paths):
        for p in paths:
            if os.path.exists(p):
                os.unlink(p)

    def outlier_rows(self):
        return self.eng.ctx.fetch_outlier_rows()

    def fetch_pairs(self, local_rows):
        return self.eng.ctx.fetch_pairs(rows=local_rows)

    def pass_stats(self):
        return self.eng.ctx.pass_stats_distributed().as_dict()

    def make_bins(self):
        return self.eng.ctx.make_bins()

    def fit(self):
        return self.eng.fit().as_dict()

    def pvalues(self):
        self.eng.ctx.pvalues()

    def bh(self, n_tests):
        self.eng.ctx.bh_distributed(n_tests)

    def fetch(self, p, q, expcc, bias):
        return self.eng.fetch(p=p, q=q, expcc=expcc, bias=bias)

    def fetch_flags(self, outlier, skip):
        return self.eng.ctx.fetch_flags(self.n_rows, outlier=outlier, skip=skip)

    def fdr_counts(self):
        return self.eng.fdr_counts()

    def next_pass(self):
        return self.eng.ctx.next_pass_distributed()

    def close(self):
        self.eng.close()


def _worker_main(rank, world, device, transport, unique_id, mesh, cmd):
    try:
        me = _Rank(rank, world, device, transport, unique_id, mesh)
    except Exception as e:                                # no GPU / no library: report, then leave
        cmd.send(("error", repr(e)))
        return
    cmd.send(("ready", None))
    while True:
        name, args = cmd.recv()
        try:
            out = getattr(me, name)(*args)
            cmd.send(("ok", out))
        except Exception as e:
            import traceback
            cmd.send(("error", "%r\n%s" % (e, traceback.format_exc())))
        if name == "close":
            return


class _Info:
    """dict with attribute access and as_dict(), like the ctypes structs Engine returns"""

    def __init__(self, d):
        self.__dict__.update(d)
        self._d = dict(d)

    def as_dict(self):
        return dict(self._d)


class _CtxFacade:
    def __init__(self, owner):
        self._o = owner

    def get_array(self, which):
        if which == _capi.A_FDR_COUNTS:
            return self._o.fdr_counts()
        return self._o.local.eng.ctx.get_array(which)    # host-side arrays are genome-wide and equal on every rank

    def make_bins(self):
        return self._o._all("make_bins")[0]

    def pvalues(self):
        self._o._all("pvalues")

    def bh(self, n_total_tests):
        self._o._all("bh", float(n_total_tests))

    def fetch_flags(self, n_rows, outlier=True, skip=False):
        parts = self._o._all("fetch_flags", outlier, skip)
        out = []
        for k, want in enumerate((outlier, skip)):
            if not want:
                out.append(None)
                continue
            a = np.empty(self._o.n_rows, np.uint8)
            for r, part in enumerate(parts):
                a[self._o.rows_of[r]] = part[k]
            out.append(a)
        return out[0], out[1]

    def bh_array(self, p, n_total_tests):
        return self._o.local.eng.ctx.bh_array(p, n_total_tests)

    def fetch_outlier_rows(self):
        """file positions of the outlier rows of all ranks, ascending (each rank compacts its own on its GPU)"""
        o = self._o
        parts = [o.file_rows(r, np.asarray(local, np.int64)) for r, local in enumerate(o._all("outlier_rows"))]
        return np.sort(np.concatenate(parts)) if parts else np.zeros(0, np.int64)

    def write_significances_device(self, name, chr_names):
        """The significances file written by all ranks at once: every rank formats and deflates the stretches of the file it holds
        (gzip members of their own), the sizes give every part its offset, and the ranks copy their parts into place.  Raises
        FhxError(FHX_ERR_UNSUPPORTED) - nothing written - when the rows did not come from the file reader (the ranks hold no file
        positions), when a rank's rows are too scattered over the file, or when the device formatter refuses a row."""
        o = self._o
        if o.segments is None or any(sg is None for sg in o.segments):
            raise _capi.FhxError(_capi.FHX_ERR_UNSUPPORTED, "the ranks' rows are not stretches of the file")
        results = o._all("write_parts", name, list(chr_names))
        parts = [p for res in results if res[0] == "ok" for p in res[1]]
        if any(res[0] != "ok" for res in results):
            o._all("drop_parts", per_rank=[([p[1] for p in (res[1] if res[0] == "ok" else [])],) for res in results])
            raise _capi.FhxError(_capi.FHX_ERR_UNSUPPORTED, "; ".join(res[1] for res in results if res[0] != "ok"))
        tmp = "%s.fhx-tmp-%d" % (name, os.getpid())
        head = tmp + ".head"
        drop = [([p[1] for p in res[1]],) for res in results]
        try:
            o.local.eng.ctx.write_significances_range(head, list(chr_names), 0, 0, True)
            offsets, at = {}, os.path.getsize(head)
            for file_start, path, nbytes in sorted(parts):
                offsets[path] = at
                at += nbytes
            with open(tmp, "wb") as f, open(head, "rb") as h:
                f.write(h.read())
                f.truncate(at)
            per_rank = [(tmp, [(p[1], offsets[p[1]]) for p in (res[1])]) for res in results]
            o._all("place_parts", per_rank=per_rank)
            os.replace(tmp, name)
        except BaseException:                                  # a full disk, a rank that died: no litter next to the output
            for path in (tmp, head):
                if os.path.exists(path):
                    os.unlink(path)
            try:
                o._all("drop_parts", per_rank=drop)
            except Exception:
                pass
            raise
        os.unlink(head)
        o._all("drop_parts", per_rank=drop)                    # only now: every rank has placed its parts


class ShardedContacts:
    """Contact rows that every rank read from the file itself (ShardedEngine.ingest_file): the length is known, identity columns
    come back from the ranks that hold them - a few rows (the outlier lines) or, for the host writer's fallback, all."""

    raw_count = None

    def __init__(self, owner, n_rows):
        self._o, self._n, self._cols = owner, int(n_rows), None

    def __len__(self):
        return self._n

    def rows(self, rows):
        rows = np.asarray(rows, np.int64)
        out = [np.empty(len(rows), np.int32) for _ in range(5)]
        if len(rows) == 0:
            return out
        ranks, local = self._o.locate(rows)
        got = self._o._all("fetch_pairs", per_rank=[(local[ranks == r],) for r in range(self._o.world)])
        for r, cols in enumerate(got):
            sel = ranks == r
            for k in range(5):
                out[k][sel] = cols[k]
        return out

    def _all_cols(self):
        if self._cols is None:
            self._cols = self.rows(np.arange(self._n, dtype=np.int64))
        return self._cols

    chr1 = property(lambda self: self._all_cols()[0])
    mid1 = property(lambda self: self._all_cols()[1])
    chr2 = property(lambda self: self._all_cols()[2])
    mid2 = property(lambda self: self._all_cols()[3])
    count = property(lambda self: self._all_cols()[4])


class ShardedEngine:
    """The Engine methods fithic_amd.fithic uses, over `gpus` ranks (see the module docstring)."""

    def __init__(self, gpus, devices=None, transport=None):
        transport = transport or os.environ.get("FHX_CLI_TRANSPORT", "rccl")
        if devices is None:
            env = os.environ.get("FHX_CLI_DEVICES")
            devices = [int(v) for v in env.split(",")] if env else list(range(gpus))
        if len(devices) != gpus:
            raise ValueError("one device ordinal per rank is needed")
        self.world, self.transport = gpus, transport
        ctxmp = mp.get_context("spawn")
        mesh = make_mesh(ctxmp, gpus) if transport == "pipes" else [None] * gpus
        uid = _capi.comm_unique_id() if transport != "pipes" else None
        self.workers = []
        self.broken = None
        self.local = None
        self.local_stuck = False
        try:
            for r in range(1, gpus):
                parent, child = ctxmp.Pipe(duplex=True)
                p = ctxmp.Process(target=_worker_main, args=(r, gpus, devices[r], transport, uid, mesh[r], child), daemon=True)
                p.start()
                self.workers.append((p, parent))
            self.local = _Rank(0, gpus, devices[0], transport, uid, mesh[0])
            for r, (p, conn) in enumerate(self.workers, start=1):
                while not conn.poll(1.0):
                    if not p.is_alive():
                        raise RuntimeError("rank %d died while starting (exit code %r)" % (r, p.exitcode))
                status, msg = conn.recv()
                if status != "ready":
                    raise RuntimeError("rank %d could not start: %s" % (r, msg))
        except BaseException:
            # nobody will ever hold this object: stop the workers that did start (they keep a GPU and their pipes) and release
            # rank 0's engine before the error travels on
            for p, _ in self.workers:
                if p.is_alive():
                    p.terminate()
            for p, _ in self.workers:
                p.join(10)
            if self.local is not None:
                try:
                    self.local.close()
                except Exception:                              # noqa: BLE001
                    pass
            raise
        self.ctx = _CtxFacade(self)
        self.n_rows = 0
        self.segments = None                                 # file reader mode: per rank [(local start, file position, length)]
        self.split = None                                    # file reader mode: "file" / "text" (parts of the file / of its text) or "chromosome"
        self.rows_of = [np.zeros(0, np.int64) for _ in range(gpus)]
        self.resolution = None

    GRACE_S = 30.0          # how long the other ranks may stay inside a command after one rank has failed in it

    def _all(self, name, *args, per_rank=None):
        """Run one command on every rank -> results by rank.  The workers get it first and rank 0's own part runs on a helper
        thread, so that this thread can watch everybody: collectives need every rank inside the call, and a rank that fails
        BEFORE a collective (its own HIP error, out of memory) leaves the others waiting in it for ever.  Every rank's answer is
        collected before anything is raised, so that a failure all ranks share (a spline the reference would exit on, a refusal
        agreed over the communicator) leaves the command / answer protocol in step for the next call; when some rank has failed
        or died and another is still inside the command GRACE_S later, the engine is abandoned instead of waiting."""
        if self.broken:
            raise RuntimeError("the sharded engine lost a rank earlier (%s): create a new one" % self.broken)
        for r, (proc, conn) in enumerate(self.workers, start=1):
            try:
                conn.send((name, per_rank[r] if per_rank else args))
            except (BrokenPipeError, OSError) as e:
                self._abandon("rank %d is gone (%r)" % (r, e))
                raise RuntimeError("rank %d died before %s" % (r, name))
        box = {}
        wake_r, wake_w = os.pipe()                         # rank 0's thread says "done" through it: one wait covers everybody

        class _Wake:
            def fileno(self):
                return wake_r

        def run_local():
            try:
                box["out"] = getattr(self.local, name)(*(per_rank[0] if per_rank else args))
            except BaseException as e:                     # noqa: BLE001 - re-raised below, after the other ranks have answered
                box["exc"] = e
            finally:
                try:
                    os.write(wake_w, b"x")
                except OSError:                            # abandoned meanwhile: nobody listens any more
                    pass

        th = threading.Thread(target=run_local, name="fhx-rank0-%s" % name, daemon=True)
        th.start()
        out = [None] * self.world
        pending = set(range(1, self.world))
        failures = []
        first_failure_at = None
        local_done = False
        wake = _Wake()
        try:
            while pending or not local_done:
                ready = mp_wait([self.workers[r - 1][1] for r in sorted(pending)] + ([] if local_done else [wake]), timeout=1.0)
                if wake in ready:
                    th.join()
                    local_done = True
                for r in sorted(pending):
                    proc, conn = self.workers[r - 1]
                    if conn in ready:
                        try:
                            status, val = conn.recv()
                        except (EOFError, OSError) as e:
                            status, val = "error", "connection lost (%r)" % (e,)
                            self.broken = "rank %d is gone in %s" % (r, name)
                        pending.discard(r)
                        if status != "ok":
                            failures.append("rank %d failed in %s: %s" % (r, name, val))
                        out[r] = val
                    elif not proc.is_alive() and not conn.poll(0):
                        pending.discard(r)
                        failures.append("rank %d died in %s (exit code %r)" % (r, name, proc.exitcode))
                        self.broken = failures[-1]
                failed = bool(failures) or (local_done and "exc" in box)
                if failed and first_failure_at is None:
                    first_failure_at = time.monotonic()
                if failed and (pending or not local_done) and time.monotonic() - first_failure_at >= self.GRACE_S:
                    stuck = ([] if local_done else [0]) + sorted(pending)
                    self.local_stuck = not local_done      # rank 0's thread sits in a collective nobody will complete
                    why = "; ".join(failures) if failures else "rank 0 failed in %s: %r" % (name, box.get("exc"))
                    self._abandon("%s - rank(s) %s were still inside the command %.0f s later" % (why, stuck, self.GRACE_S))
                    if "exc" in box and not failures:
                        raise box["exc"]
                    raise RuntimeError(self.broken)
        finally:
            os.close(wake_r)
            if local_done:
                os.close(wake_w)                           # else the stuck thread still owns the write end
        if self.broken:                                    # a worker died: nothing further can run
            self._abandon(self.broken)
            raise RuntimeError(self.broken)
        if "exc" in box:
            raise box["exc"]
        if failures:
            raise RuntimeError("; ".join(failures))
        out[0] = box.get("out")
        return out

    def _abandon(self, why):
        """A rank is lost: no further command can complete.  Stop the remaining workers."""
        self.broken = why
        for p, _ in self.workers:
            if p.is_alive():
                p.terminate()

    # ---- Engine surface -------------------------------------------------------------------------------------------
    def configure(self, resolution, dist_low=0, dist_up=float("inf"), n_bins=100, mapp_thres=1, mode="intraOnly",
                  bias_low=0.5, bias_up=2.0, totals="reference"):
        if mode not in MODES:
            raise ValueError("Invalid Option. Only options are 'All', 'interOnly', or 'intraOnly'")
        if totals not in TOTALS:
            raise ValueError("totals must be 'reference' or 'wide'")
        self.resolution = int(resolution)
        self._all("configure", resolution, dist_low, dist_up, n_bins, mapp_thres, mode, bias_low, bias_up, totals)

    def load_fragments(self, chr_ids, mids, hits, chr_sort_rank):
        self._all("load_fragments", chr_ids, mids, hits, chr_sort_rank)

    def load_bias(self, chr_ids, mids, bias):
        self._all("load_bias", chr_ids, mids, bias)

    def load_contacts(self, chr1, mid1, chr2, mid2, count):
        chr1 = np.asarray(chr1)
        n = len(chr1)
        per_chr = np.bincount(chr1, minlength=int(chr1.max()) + 1 if n else 1)
        load = [0] * self.world
        owner = np.zeros(len(per_chr), np.int64)
        for c in np.argsort(-per_chr, kind="stable"):
            r = min(range(self.world), key=lambda k: load[k])
            owner[c] = r
            load[r] += int(per_chr[c])
        row_owner = owner[chr1] if n else np.zeros(0, np.int64)
        self.rows_of = [np.flatnonzero(row_owner == r) for r in range(self.world)]
        cols = [np.asarray(a) for a in (chr1, mid1, chr2, mid2, count)]
        per_rank = [tuple(a[rows] for a in cols) + (rows,) for rows in self.rows_of]
        self._all("load_contacts", per_rank=per_rank)
        self.segments = None
        self.n_rows = n

    def ingest_file(self, path, chroms, threads=0):
        """The contacts file read by EVERY rank on its own GPU (inflate + parse of the whole file, then each keeps the rows whose
        first chromosome is its own): no row travels through this process.  The ranks agree on the owners because they count the
        same file.  -> a ShardedContacts, or None when the device reader does not take the file (the caller then parses on the
        host and hands the columns out: load_contacts)."""
        from . import tables
        per = max(1, (os.cpu_count() or 1) // self.world) if not threads else threads
        if os.environ.get("FHX_CLI_THREADS_PER_RANK"):         # measurements: a fixed number of host threads per rank, whatever N
            per = max(1, int(os.environ["FHX_CLI_THREADS_PER_RANK"]))
        if os.environ.get("FHX_CLI_SPLIT", "file") != "chromosome":
            con = self._ingest_slices(path, chroms, per)
            if con is not None:
                return con
        # (a file the device does not inflate goes through the host on every rank: only if N copies of the text fit there)
        results = self._all("ingest_file", path, per, self.world == 1 or self._whole_text_per_rank_fits(path))
        if any(res[0] != "ok" for res in results):
            self._all("ingest_discard")
            return None
        _, n, names, counts = results[0]
        ids = tables._interner(chroms)(names)
        load = [0] * self.world
        owner = np.zeros(len(names), np.int64)
        for k in np.argsort(-np.asarray(counts), kind="stable"):          # greedy by row count, as load_contacts
            r = min(range(self.world), key=lambda q: load[q])
            owner[k] = r
            load[r] += int(counts[k])
        kept = self._all("commit_shard", per_rank=[(ids, (owner == r).astype(np.uint8)) for r in range(self.world)])
        if sum(k[0] for k in kept) != n:
            raise RuntimeError("the ranks kept %d of %d rows" % (sum(k[0] for k in kept), n))
        if any(k[1] is None for k in kept):
            # a file that is not sorted by chromosome: some rank's rows are more than a thousand separate stretches of it, each of
            # which would be a part file and a writer call of its own - the caller parses on the host and hands the columns out
            return None
        self.n_rows = int(n)
        self.segments = [k[1] for k in kept]
        self._rows_of = None
        self.split = "chromosome"
        return ShardedContacts(self, self.n_rows)

    def _ingest_slices(self, path, chroms, threads):
        """The cheaper split: rank r inflates and parses part r of the FILE (whole gzip members, cut where the compressed bytes
        divide evenly), so the file is read once in all, each rank's rows are ONE stretch of the output, and the rows - K2's work -
        are balanced whatever the chromosomes' sizes.  Nothing in the engine needs a chromosome's rows on one rank (the statistics
        are sums over rows; the off-grid test and the outlier mask are per row).  Needs members that carry their sizes (this
        library's writers) and parts that end on a row; a file without such members (plain gzip) is inflated by every rank on the
        host and its TEXT cut into N parts on row starts, and so is one whose members end inside rows (bgzip).  None (a text outside
        the device parser's grammar): the caller goes on to the split by chromosome and from there to the host parser."""
        from . import tables
        cut = "file"
        results = self._all("ingest_slice", path, threads)
        if any(res[0] != "ok" for res in results) or not all(res[3] for res in results[:-1]):
            # not a chain of size-tagged members (plain gzip), or members that end inside rows (bgzip; the pieces of a row at the
            # ends of a part may also have made the parser refuse it): every rank inflates the file on its share of the host cores
            # and takes the rows that start in its N-th of the TEXT - 1/N of the upload and of the parse
            self._all("ingest_discard")
            results = self._ingest_stream_parts(path, threads)
            cut = "stream"
            if results is None:
                if not self._whole_text_per_rank_fits(path):
                    return None
                results = self._all("ingest_text_slice", path, threads)
                cut = "text"
            if any(res[0] != "ok" for res in results):
                self._all("ingest_discard")
                return None
        intern = tables._interner(chroms)
        ids = [intern(res[2]) for res in results]              # in rank order = the file's order of first appearance
        first = np.concatenate([[0], np.cumsum([res[1] for res in results])])
        kept = self._all("commit_slice", per_rank=[(ids[r], int(first[r])) for r in range(self.world)])
        self.n_rows = int(first[-1])
        if sum(k[0] for k in kept) != self.n_rows:
            raise RuntimeError("the ranks kept %d of %d rows" % (sum(k[0] for k in kept), self.n_rows))
        self.segments = [k[1] for k in kept]
        self._rows_of = None
        self.split = cut
        return ShardedContacts(self, self.n_rows)

    def _ingest_stream_parts(self, path, threads):
        """ONE plain gzip stream (what `gzip` writes) inflated by the ranks together: rank r decodes the chunks of its N-th of the
        compressed bytes without the 32 KB before them and reports its last 32 K symbols in terms of that unknown window; chained
        here rank after rank they give every part its window, with which the rank decodes its chunks into text.  CRC-32 and
        length of the whole are checked against the file's trailer; a row belongs to the rank whose text holds its first byte
        (the head of a text that begins inside a row goes to the rank before).  1/N of the inflate and of the text per rank, where
        every rank inflating the file is N times both.  -> per rank ("ok", rows, names, True) as ingest_text_slice, or None: not
        such a file (then the whole-text route, if N copies of the text fit the host's memory)."""
        if os.environ.get("FHX_CLI_STREAM_PARTS", "1") == "0":
            return None
        first = self._all("inflate_part", path, threads)
        if any(res[0] != "ok" for res in first) or not first[-1][3] or any(res[3] for res in first[:-1]):
            self._all("inflate_part_drop")
            return None
        windows = _capi.chain_windows([np.frombuffer(res[1], np.uint16) for res in first])
        second = self._all("inflate_part_resolve", per_rank=[(w.tobytes(),) for w in windows])
        ok = all(res[0] == "ok" for res in second)
        if ok:
            with open(path, "rb") as f:
                f.seek(-8, os.SEEK_END)
                trailer = f.read(8)
            want_crc, want_isize = int.from_bytes(trailer[:4], "little"), int.from_bytes(trailer[4:], "little")
            crc, total = second[0][1], second[0][2]
            for res in second[1:]:
                crc = _capi.crc32_combine(crc, res[1], res[2])
                total += res[2]
            ok = crc == want_crc and total % (1 << 32) == want_isize and all(res[2] == f[2] for res, f in zip(second, first))
            ok = ok and all(res[3] >= 0 for res in second)             # every part holds the end of a row
        if not ok:
            self._all("inflate_part_drop")
            return None
        own = rows_owned_by_parts([(res[3], res[4]) for res in second], [res[5] for res in second])
        return self._all("ingest_text_own", per_rank=[(skip, extra, threads) for skip, extra in own])

    @staticmethod
    def _available_host_bytes():
        try:
            with open("/proc/meminfo") as f:
                for line in f:
                    if line.startswith("MemAvailable:"):
                        return int(line.split()[1]) * 1024
        except OSError:
            pass
        return None

    def _whole_text_per_rank_fits(self, path):
        """The routes on which EVERY rank inflates the whole file hold N copies of its text in host memory (63 GB each for a
        2e9-row file).  Texts of this kind compress 4-5 x; with 8 x the file's size per rank as the estimate the routes are taken
        only when that is under half of what the host has free - else rank 0 parses and hands the columns out (one copy)."""
        if os.environ.get("FHX_CLI_TEXT_ROUTE", "") == "force":
            return True
        free = self._available_host_bytes()
        try:
            need = 8 * os.path.getsize(path) * self.world
        except OSError:
            return True
        return free is None or need <= free // 2

    def file_rows(self, rank, local_rows):
        """file positions of some local rows of a rank"""
        if self.segments is None or self.segments[rank] is None:
            return self.rows_of[rank][local_rows]
        seg = self.segments[rank]
        if not seg:
            return np.zeros(0, np.int64)
        starts = np.array([sg[0] for sg in seg], np.int64)
        files = np.array([sg[1] for sg in seg], np.int64)
        k = np.searchsorted(starts, local_rows, side="right") - 1
        return files[k] + (local_rows - starts[k])

    def locate(self, file_rows):
        """(rank, local row) of some file positions (file reader mode)"""
        table = sorted((sg[1], sg[2], r, sg[0]) for r, seg in enumerate(self.segments) for sg in seg)
        fs = np.array([t[0] for t in table], np.int64)
        k = np.searchsorted(fs, file_rows, side="right") - 1
        ranks = np.array([t[2] for t in table], np.int64)[k]
        local = np.array([t[3] for t in table], np.int64)[k] + (file_rows - fs[k])
        return ranks, local

    @property
    def rows_of(self):
        if self._rows_of is None:                          # file reader mode: rebuilt from the stretches (host-writer fallback, tests)
            if self.segments is None or any(sg is None for sg in self.segments):
                raise RuntimeError("the ranks' rows are too scattered over the file to list them")
            self._rows_of = [np.concatenate([np.arange(f, f + ln, dtype=np.int64) for _, f, ln in seg]) if seg else np.zeros(0, np.int64)
                             for seg in self.segments]
        return self._rows_of

    @rows_of.setter
    def rows_of(self, value):
        self._rows_of = value

    def pass_stats(self):
        return _Info(self._all("pass_stats")[0])

    def fit(self):
        return _Info(self._all("fit")[0])

    def fetch(self, p=True, q=True, expcc=False, bias=False):
        parts = self._all("fetch", p, q, expcc, bias)
        out = {}
        for key in parts[0]:
            a = np.empty(self.n_rows, np.float64)
            for r, part in enumerate(parts):
                a[self.rows_of[r]] = part[key]
            out[key] = a
        return out

    def fdr_counts(self):
        return np.sum(self._all("fdr_counts"), axis=0)       # shifted cumulative counts are linear in the rows

    def next_pass(self):
        return self._all("next_pass")[0]

    def close(self):
        """Never raises: a session is closed on error paths too (and by reset_session before the next run)."""
        closed_local = False
        try:
            if not self.broken:
                self._all("close")
                closed_local = True
        except Exception:                                  # noqa: BLE001
            pass
        finally:
            for p, _ in self.workers:
                p.join(10)
                if p.is_alive():
                    p.terminate()
            if not closed_local and not self.local_stuck:  # a stuck rank-0 thread still uses the context: leave it to process exit
                try:
                    self.local.close()
                except Exception:                          # noqa: BLE001
                    pass
