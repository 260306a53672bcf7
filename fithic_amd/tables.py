"""Input tables of Fit-Hi-C as SoA numpy arrays (parsed by the native reader in libfithic_mi355x.so).

File formats are the reference's (README.md:139-208 of ay-lab/fithic):
  contacts   gz text, whitespace separated  chr1 mid1 chr2 mid2 count      (fithic/fithic.py:413-417)
  fragments  gz text, 5 columns, uses col 0 (chr), 2 (mid), 3 (hits)         (fithic/fithic.py:581-590)
  bias       gz text  chr mid bias                                            (fithic/fithic.py:805-808)
Chromosome names of all three files are interned into one id space (`ChromIndex`); the reference iterates
chromosomes in Python's sorted() string order (fithic/fithic.py:606), which `sort_rank()` hands to the engine.
"""
import numpy as np


class ChromIndex:
    """Chromosome name <-> small int id, shared by the three tables."""

    def __init__(self):
        self.names = []
        self._ids = {}

    def intern(self, name):
        j = self._ids.get(name)
        if j is None:
            j = self._ids[name] = len(self.names)
            self.names.append(name)
        return j

    def sort_rank(self):
        order = sorted(range(len(self.names)), key=lambda i: self.names[i])
        rank = np.empty(len(self.names), np.int32)
        for r, i in enumerate(order):
            rank[i] = r
        return rank

    def __len__(self):
        return len(self.names)


class Contacts:
    """chr1, mid1, chr2, mid2 (int32) + count = int(float(text)) and the float itself (the writer prints %d of it)."""

    def __init__(self, chr1, mid1, chr2, mid2, count, raw_count):
        self.chr1, self.mid1, self.chr2, self.mid2 = chr1, mid1, chr2, mid2
        self.count, self.raw_count = count, raw_count

    def __len__(self):
        return len(self.mid1)

    def take(self, sel):
        return Contacts(self.chr1[sel], self.mid1[sel], self.chr2[sel], self.mid2[sel], self.count[sel],
                        None if self.raw_count is None else self.raw_count[sel])


class DeviceContacts:
    """Contact rows that were parsed on the GPU and never existed as host columns (fhx_ingest_contacts_text): the length is
    known, the identity columns come back from the device on demand - all of them (the host writer's fallback) or a few rows
    (the outlier lines)."""

    raw_count = None

    def __init__(self, ctx, n_rows):
        self._ctx, self._n, self._cols = ctx, int(n_rows), None

    def __len__(self):
        return self._n

    def rows(self, rows):
        """(chr1, mid1, chr2, mid2, count) of some rows"""
        return self._ctx.fetch_pairs(rows=rows)

    def _all(self):
        if self._cols is None:
            self._cols = self._ctx.fetch_pairs(n=self._n)
        return self._cols

    chr1 = property(lambda self: self._all()[0])
    mid1 = property(lambda self: self._all()[1])
    chr2 = property(lambda self: self._all()[2])
    mid2 = property(lambda self: self._all()[3])
    count = property(lambda self: self._all()[4])


def _interner(chroms):
    """names of a file (order of first appearance) -> ids in the run's shared id space, new names appended in that order"""
    def ids_of(names):
        out = np.empty(len(names), np.int32)
        for k, name in enumerate(names):
            j = chroms._ids.get(name)
            if j is None:
                j = chroms._ids[name] = len(chroms.names)
                chroms.names.append(name)
            out[k] = j
        return out
    return ids_of


def read_contacts(path, chroms, threads=0, want_raw=False):
    """Native multi-threaded parser (libfithic_mi355x.so: fhx_host_read_table).  Same semantics as the reference's loop:
    whitespace split, exactly five fields, int(mid), count = int(float(text)); a malformed line raises like the reference.
    Both loci share the file's name table (interned in the reference's order of appearance: row by row, chr1 then chr2).
    want_raw: also return the float the count was parsed from (HiCKRy's matrix values); 8 B/row, so only on request."""
    from . import _capi
    names, cols, raw = _capi.host_read_table(path, 0, threads, name_ids=_interner(chroms), want_float=want_raw)
    return Contacts(cols[0], cols[1], cols[2], cols[3], cols[4], raw)


def load_contacts(path, chroms, engine_of, threads=0):
    """The contacts file into the engine, on the GPU as far as the file allows.  engine_of() -> the engine, configured.
      1. a file of size-tagged gzip members (this library's writers, bgzip): the compressed bytes are uploaded, the GPU inflates
         and parses them (fhx_ingest_contacts_file) - the text never exists on the host;
      2. any other gzip file: inflated on the host cores, the text parsed by the GPU (fhx_ingest_contacts_text);
      3. a text outside the device parser's grammar (it says FHX_ERR_UNSUPPORTED), a sharded engine (its ranks take host
         columns) or FHX_HOST_READER=1: the host parser, which reports a malformed line as the reference would.
    Returns a DeviceContacts (1, 2) or a Contacts (3)."""
    import os
    import time
    from . import _capi
    timing = os.environ.get("FHX_TIMING")
    marks = [("start", time.time())]

    def mark(what):
        if timing:
            marks.append((what, time.time()))

    def report():
        if timing:
            print("stage: contacts:" + "".join(" %s %.3f s;" % (w, t - marks[k][1]) for k, (w, t) in enumerate(marks[1:])))

    def commit(eng, ctx, names, n):
        try:
            eng.commit_contacts_text(_interner(chroms)(names), n)
            mark("rows into the engine")
        except BaseException:
            ctx.ingest_contacts_discard()
            raise
        return DeviceContacts(ctx, n)
    # `fithic --gpus N`: every rank reads the file itself on its own GPU (sharded.ShardedEngine.ingest_file); a file the device
    # reader does not take falls through to the host parser below, whose columns rank 0 hands out
    if getattr(engine_of, "sharded", False) and not os.environ.get("FHX_HOST_READER") and not os.environ.get("FHX_CLI_FUNNEL"):
        eng0 = engine_of()
        if hasattr(eng0, "ingest_file"):
            got = eng0.ingest_file(path, chroms, threads)
            mark({"file": "every rank: inflate + parse its part of the file",
                  "stream": "every rank: inflate its part of the stream on the host, parse its rows",
                  "text": "every rank: inflate on the host, parse its part of the text"}.get(getattr(eng0, "split", None),
                                                                                             "every rank: inflate + parse + keep its chromosomes"))
            if got is not None:
                report()
                return got
    # a file without member sizes (plain gzip) is inflated by the host cores whatever follows: do that before waiting for the
    # engine, whose start-up (HIP runtime, context) then hides behind it
    text = None
    tagged = False
    try:
        with open(path, "rb") as f:
            head = f.read(16)
        tagged = len(head) >= 14 and head[:3] == b"\x1f\x8b\x08" and bool(head[3] & 4) and head[12:14] in (b"FH", b"BC")
    except OSError:
        pass
    if not tagged:
        text = _capi.HostText(path, threads)
        mark("read + inflate")
    try:
        eng = engine_of()
    except BaseException:
        if text is not None:
            text.close()
        raise
    mark("engine ready")
    ctx = getattr(eng, "ctx", None)
    on_device = ctx is not None and hasattr(ctx, "ingest_contacts_text") and not os.environ.get("FHX_HOST_READER")
    try:
        if text is None and on_device and not os.environ.get("FHX_HOST_INFLATE"):
            try:
                n, names = ctx.ingest_contacts_file(path, threads)
                mark("device inflate + parse")
                return commit(eng, ctx, names, n)
            except _capi.FhxError as e:
                if e.code != _capi.FHX_ERR_UNSUPPORTED:
                    raise
                mark("device refused")
                if getattr(e, "refused", 1) == 2:
                    on_device = False
        if text is None:
            text = _capi.HostText(path, threads)
            mark("read + inflate")
        if on_device:
            try:
                n, names = ctx.ingest_contacts_text(text, threads)
                mark("device parse")
            except _capi.FhxError as e:
                if e.code != _capi.FHX_ERR_UNSUPPORTED:
                    raise
            else:
                text.close()                         # the text is in HBM and not needed again
                text = None
                mark("text freed")
                return commit(eng, ctx, names, n)
        names, cols, _ = _capi.host_parse_text(text, 0, threads, name_ids=_interner(chroms), want_float=False)
        mark("host parse")
    finally:
        if text is not None:
            text.close()
            mark("text freed")
        report()
    c = Contacts(cols[0], cols[1], cols[2], cols[3], cols[4], None)
    eng.load_contacts(c.chr1, c.mid1, c.chr2, c.mid2, c.count)
    return c


def read_fragments(path, chroms, threads=0):
    """-> (chr ids, mids, hits) as int32 arrays in file order."""
    from . import _capi
    names, cols, _ = _capi.host_read_table(path, 1, threads, name_ids=_interner(chroms))
    return cols[0], cols[1], cols[4]


def read_bias(path, chroms, threads=0):
    """-> (chr ids, mids, raw bias values); bounds / NaN / first-occurrence rules are applied by the engine."""
    from . import _capi
    names, cols, bias = _capi.host_read_table(path, 2, threads, name_ids=_interner(chroms))
    return cols[0], cols[1], bias


def bias_quantiles(bias):
    """The three numbers read_biases logs (fithic/fithic.py:805-815): scipy.stats.mstats.mquantiles defaults
    (alphap = betap = 0.4) over the biases that differ from 1.0."""
    v = np.sort(np.asarray(bias, np.float64)[np.asarray(bias) != 1.0])
    n = len(v)
    out = []
    for prob in (0.05, 0.5, 0.95):
        if n == 0:
            out.append(float("nan"))
            continue
        m = 0.4 + prob * (1.0 - 0.4 - 0.4)
        aleph = n * prob + m
        k = int(np.floor(min(max(aleph, 1), n - 1)))
        gamma = min(max(aleph - k, 0.0), 1.0)
        out.append((1.0 - gamma) * v[k - 1] + gamma * v[min(k, n - 1)])
    return out
