"""Native text I/O of libfithic_mi355x.so, CPU only: the reader against an independent pandas parse of
every fixture file, the writer against the md5 of the reference's own decompressed output."""
import gzip
import hashlib
import os

import numpy as np
import pytest

from conftest import load_case, case_args, DATA, ALL_CASES
from fithic_amd import _capi, tables


@pytest.mark.parametrize("fname,kind", [("hESC_chr1_w40000.contacts.gz", 0), ("synth_IMR90_w1Mb.contacts.gz", 0),
                                        ("quirk.contacts.gz", 0), ("IMR90_w1Mb.frags.gz", 1), ("quirk.frags.gz", 1),
                                        ("hESC_chr1_w40000.frags.gz", 1), ("IMR90_w1Mb.bias.gz", 2), ("quirk.bias.gz", 2)])
@pytest.mark.parametrize("threads", [1, 5])
def test_reader_matches_an_independent_parse(fname, kind, threads):
    path = os.path.join(DATA, fname)
    names, cols, dv = _capi.host_read_table(path, kind, threads)
    rows = [ln.split() for ln in gzip.open(path, "rt")]
    assert len(rows) == len(cols[0])
    mine_chr = [names[i] for i in cols[0]]
    assert mine_chr == [r[0] for r in rows]
    seen = []
    for r in rows:                                   # order of first appearance (chr1 then chr2 per row for contacts)
        for name in ([r[0], r[2]] if kind == 0 else [r[0]]):
            if name not in seen:
                seen.append(name)
    assert names == seen
    if kind == 0:
        assert np.array_equal(cols[1], [int(r[1]) for r in rows]) and np.array_equal(cols[3], [int(r[3]) for r in rows])
        assert [names[i] for i in cols[2]] == [r[2] for r in rows]
        assert np.array_equal(cols[4], [int(float(r[4])) for r in rows])
        assert np.array_equal(dv, [float(r[4]) for r in rows])
    elif kind == 1:
        assert np.array_equal(cols[1], [int(r[2]) for r in rows]) and np.array_equal(cols[4], [int(r[3]) for r in rows])
    else:
        want = np.array([float(r[2]) for r in rows])
        assert np.array_equal(cols[1], [int(r[1]) for r in rows])
        assert np.array_equal(np.isnan(dv), np.isnan(want)) and np.array_equal(dv[~np.isnan(dv)], want[~np.isnan(want)])


def test_reader_rejects_what_the_reference_rejects(tmp_path):
    bad = tmp_path / "bad.gz"
    with gzip.open(bad, "wt") as f:
        f.write("chr1\t5000\tchr1\t15000\t3\nchr1\t5000\tchr1\t25000\n")        # 4 fields: ValueError in the reference
    with pytest.raises(_capi.FhxError) as e:
        _capi.host_read_table(str(bad), 0)
    assert e.value.code == _capi.FHX_ERR_REFERENCE_EXIT and "line 2" in str(e.value)
    with gzip.open(bad, "wt") as f:
        f.write("chr1\t5000\tchr1\t1.5e4\t3\n")                                  # int('1.5e4') raises
    with pytest.raises(_capi.FhxError):
        _capi.host_read_table(str(bad), 0)
    with pytest.raises(_capi.FhxError):
        _capi.host_read_table(str(tmp_path / "missing.gz"), 0)


def test_reader_follows_python_number_and_split_grammar(tmp_path):
    """fithic.py:413-417 does `lines.split()`, `int(mid)`, `int(float(count))` on lines read in text mode.  Random lines built from
    ordinary and adversarial ASCII tokens (signs, underscores by PEP 515, exponents, nan / inf, hexadecimal forms that strtod
    takes and float() does not, values outside int32, wrong field counts, every ASCII separator str.split() knows) must be
    accepted or refused exactly as Python does, with the same values.  Non-ASCII digits and spaces (which Python would take)
    are refused: a documented deviation.  A number the reference takes but the int32 columns cannot hold is
    FHX_ERR_UNSUPPORTED (a limit of the library), everything the reference itself raises on is FHX_ERR_REFERENCE_EXIT."""
    rng = np.random.default_rng(3)
    int_tok = ["5", "+5", "-5", "0", "007", "1_000", "1__0", "_1", "1_", "2147483647", "2147483648", "-2147483648", "-2147483649", "12a", "1.0",
               "1e3", "0x10", "+", "-", "+-3"]
    flt_tok = ["3", "3.9", "-2.5", "1e3", "1E3", "1e-3", ".5", "5.", "nan", "NaN", "inf", "-inf", "Infinity", "infinity", "1_0.5", "1__0", "0x10",
               "0x1p3", "1e400", "-1e400", "4294967296", "2147483647.9", "2147483648", "abc", "1e", "e5", "+7", "--7", "1.2.3", "1d3", "1,5",
               "9" * 20, "0" * 30 + "7", "1" + "0" * 70, "nan(1)", "1e+3", "1e+", "-.5e-2", "1_0e1_0", "_1.0", ".", "in", "infin"]

    def python_parse(line):
        """-> (row, None), or (None, "range") when the reference takes the line but a number does not fit the int32 columns, or
        (None, "malformed") when the reference itself raises"""
        f = line.split()
        if len(f) != 5:
            return None, "malformed"
        try:
            m1, m2, c = int(f[1]), int(f[3]), int(float(f[4]))
        except (ValueError, OverflowError):
            return None, "malformed"
        if not all(-2 ** 31 <= v < 2 ** 31 for v in (m1, m2, c)):
            return None, "range"                                   # int32 columns: refusing is the documented behaviour
        return (f[0], m1, f[2], m2, c), None

    path = str(tmp_path / "c.gz")
    checked = accepted = ranged = 0
    for trial in range(1500):
        t1 = "123" if rng.random() < 0.5 else str(rng.choice(int_tok))
        t2 = str(rng.choice(int_tok[:4]))
        tf = "4" if rng.random() < 0.3 else str(rng.choice(flt_tok))
        sep = str(rng.choice(["\t", " ", "  ", " \t", "\x0b", "\x0c", "\x1c", "\x1f", "\r"]))
        fields = ["chrA", t1, "chrB", t2, tf]
        if rng.random() < 0.05:
            fields = fields[:4]
        if rng.random() < 0.05:
            fields.append("x")
        line = sep.join(fields) + str(rng.choice(["\n", "\r\n", " \n", "\r", " \r"]))
        with gzip.open(path, "wb") as f:
            f.write(("chr1\t5\tchr1\t9\t2\n" + line + "chr2\t1\tchr2\t2\t3\n").encode())
        # what the reference sees: the lines of text mode (universal newlines: \r and \r\n end a line too, fithic.py:406)
        want_rows, why = [], None
        for ln in gzip.open(path, "rt"):
            row, bad = python_parse(ln)
            if bad:
                why = bad
                break
            want_rows.append(row)
        try:
            names, cols, _ = _capi.host_read_table(path, 0, 1)
            got = [(names[cols[0][i]], int(cols[1][i]), names[cols[2][i]], int(cols[3][i]), int(cols[4][i])) for i in range(len(cols[1]))]
            assert why is None and got == want_rows, repr(line)
        except _capi.FhxError as e:
            assert why is not None, repr(line)
            assert e.code == (_capi.FHX_ERR_UNSUPPORTED if why == "range" else _capi.FHX_ERR_REFERENCE_EXIT), (repr(line), why, str(e))
        checked += 1
        accepted += why is None
        ranged += why == "range"
    assert checked > 1000 and 300 < accepted < checked - 300 and ranged > 20
    for line in ("chrA\t\u0663\tchrB\t5\t4\n", "chrA\t3\tchrB\t5\t\u0661\u0662\n"):        # Arabic-Indic digits: Python reads 3 and 12
        with gzip.open(path, "wt", encoding="utf-8") as f:
            f.write(line)
        with pytest.raises(_capi.FhxError):
            _capi.host_read_table(path, 0, 1)


def test_bytes_after_the_last_gzip_member_as_python_gzip_treats_them(tmp_path):
    """gzip.open(..., 'rt') (fithic.py:406): zero padding after a member is skipped, another member is read, anything else raises
    BadGzipFile once the reader gets there - the reference dies with a traceback; a truncated next member raises EOFError."""
    text = b"chr1\t5\tchr1\t9\t2\nchr1\t5\tchr1\t19\t1\n"
    member = gzip.compress(text)
    path = str(tmp_path / "t.gz")
    cases = [(member, True), (member + b"\0" * 7, True), (member + member, True), (member + b"\0\0\0" + member, True),
             (member + b"garbage", False), (member + b"\0\0x", False), (member + b"\x1f", False), (member + b"\x1f\x8c" + b"\0" * 20, False),
             (member + member[:12], False), (member + b"\x1f\x8b\x07" + b"\0" * 20, False)]
    for data, fine in cases:
        with open(path, "wb") as f:
            f.write(data)
        try:
            with gzip.open(path, "rt") as f:
                n_py = len(f.readlines())
            py_ok = True
        except (OSError, EOFError):
            py_ok = False
        assert py_ok == fine, data[len(member):]
        if fine:
            names, cols, _ = _capi.host_read_table(path, 0, 1)
            assert len(cols[1]) == n_py
        else:
            with pytest.raises(_capi.FhxError) as e:
                _capi.host_read_table(path, 0, 1)
            assert e.value.code == _capi.FHX_ERR_REFERENCE_EXIT


def test_lone_carriage_return_is_a_line_break(tmp_path):
    """Text mode's universal newlines: \\r, \\n and \\r\\n all end a line, so the line NUMBERS the outlier lists carry follow."""
    path = str(tmp_path / "cr.gz")
    body = "chr1\t5\tchr1\t9\t2\rchr1\t5\tchr1\t19\t1\r\nchr2\t1\tchr2\t2\t3\nchr2\t1\tchr2\t7\t4\r"
    with gzip.open(path, "wb") as f:
        f.write(body.encode())
    want = [ln.split() for ln in gzip.open(path, "rt")]
    assert len(want) == 4
    for threads in (1, 3):
        names, cols, _ = _capi.host_read_table(path, 0, threads)
        assert [int(v) for v in cols[3]] == [int(r[3]) for r in want] and [int(v) for v in cols[4]] == [int(r[4]) for r in want]
    with gzip.open(path, "wb") as f:
        f.write(b"chr1\t5\tchr1\t9\t2\r\rchr1\t5\tchr1\t19\t1\n")          # an empty line: split() gives [] -> ValueError
    with pytest.raises(_capi.FhxError) as e:
        _capi.host_read_table(path, 0, 1)
    assert e.value.code == _capi.FHX_ERR_REFERENCE_EXIT and "line 2" in str(e.value)


@pytest.mark.parametrize("name", ALL_CASES)
def test_writer_reproduces_the_reference_file(name, tmp_path):
    """Values from the checker (bit-identical to the reference, see test_oracle_golden) through the PRODUCT's writer:
    the decompressed bytes must hash to the md5 of the reference's own output file."""
    from oracle import fithic_oracle as fo
    from fithic_amd.engine import MODES
    meta, g = load_case(name)
    kw = case_args(meta)
    chroms = tables.ChromIndex()
    con = tables.read_contacts(kw["contacts"], chroms)
    ref = fo.run(**kw)
    for pi, r in enumerate(ref, 1):
        out = str(tmp_path / ("p%d.gz" % pi))
        n = _capi.host_write_significances(out, chroms.names, con.chr1, con.mid1, con.chr2, con.mid2, con.count, r.p, r.q,
                                           r.b1, r.b2, r.expcc, MODES[kw["mode"]], kw["L"], kw["U"], threads=3)
        with gzip.open(out, "rb") as f:
            text = f.read()
        assert n == meta["sig_rows_pass%d" % pi] == text.count(b"\n") - 1
        assert hashlib.md5(text).hexdigest() == meta["sig_md5_pass%d" % pi]


def test_inflate_and_parse_as_separate_stages(tmp_path):
    """fhx_host_inflate + fhx_host_parse_text (the fallback of the device parser) = fhx_host_read_table: rows, names, and the
    message of a malformed line; a file that is not gzip fails in the inflate stage with the reader's message"""
    path = os.path.join(DATA, "hESC_chr1_w40000.contacts.gz")
    text = _capi.HostText(path, 3)
    assert len(text) == len(gzip.open(path, "rb").read())
    for want_float in (True, False):
        a = _capi.host_parse_text(text, 0, 2, want_float=want_float)
        b = _capi.host_read_table(path, 0, 2, want_float=want_float)
        assert a[0] == b[0] and all(np.array_equal(a[1][k], b[1][k]) for k in b[1])
        assert (a[2] is None and b[2] is None) or np.array_equal(a[2], b[2])
    text.close()
    text.close()                                                            # idempotent
    bad = tmp_path / "bad.gz"
    with gzip.open(bad, "wt") as f:
        f.write("chr1\t5000\tchr1\t15000\t3\nchr1\t5000\tchr1\t25000\n")
    text = _capi.HostText(str(bad))
    with pytest.raises(_capi.FhxError) as e1:
        _capi.host_parse_text(text, 0)
    with pytest.raises(_capi.FhxError) as e2:
        _capi.host_read_table(str(bad), 0)
    assert str(e1.value) == str(e2.value) and "line 2" in str(e1.value)
    text.close()
    notgz = tmp_path / "plain.txt"
    notgz.write_text("chr1\t1\tchr1\t2\t3\n")
    with pytest.raises(_capi.FhxError) as e1:
        _capi.HostText(str(notgz))
    with pytest.raises(_capi.FhxError) as e2:
        _capi.host_read_table(str(notgz), 0)
    assert e1.value.code == e2.value.code == _capi.FHX_ERR_REFERENCE_EXIT and str(e1.value) == str(e2.value)
    empty = tmp_path / "empty.gz"
    empty.write_bytes(b"")
    with pytest.raises(_capi.FhxError):
        _capi.HostText(str(empty))


def _plain_gzip(data, level=6, strategy=None, name=None):
    import struct
    import zlib
    co = zlib.compressobj(level, zlib.DEFLATED, -15, 8, zlib.Z_DEFAULT_STRATEGY if strategy is None else strategy)
    body = co.compress(data) + co.flush()
    head = b"\x1f\x8b\x08" + (b"\x08" if name else b"\x00") + b"\0\0\0\0\x00\x03" + ((name + b"\0") if name else b"")
    return head + body + struct.pack("<II", zlib.crc32(data), len(data) & 0xffffffff)


def test_one_plain_gzip_stream_is_inflated_on_all_cores(tmp_path, monkeypatch):
    """csrc/fhx_gunzip.cpp: block starts found by their headers, chunks decoded with the unknown window as 16-bit symbols.  The
    text must be what Python's gzip gives, for every level and for streams that mix stored, fixed and dynamic blocks; what the
    scheme cannot take (two members, fixed codes only, damage) comes out of zlib on one thread as before - same text, same errors.
    FHX_PGUNZIP_MIN / _CHUNK shrink the sizes from which it is used so that small files go through it."""
    import random
    import zlib
    rng = random.Random(12)
    rows = []
    for _ in range(150_000):
        c = rng.randrange(1, 23)
        rows.append("chr%d\t%d\tchr%d\t%d\t%d\n" % (c, rng.randrange(1, 50000) * 5000 + 2500, c, rng.randrange(1, 50000) * 5000 + 2500, rng.randrange(1, 500)))
    text = "".join(rows).encode()
    noise = bytes(rng.getrandbits(8) for _ in range(300_000))
    monkeypatch.setenv("FHX_PGUNZIP_MIN", "0")
    monkeypatch.setenv("FHX_PGUNZIP_CHUNK", "200000")
    monkeypatch.setenv("FHX_TIMING", "1")
    cases = {
        "level1": _plain_gzip(text, 1), "level6": _plain_gzip(text, 6), "level9": _plain_gzip(text, 9),
        "named": _plain_gzip(text, 6, name=b"contacts.txt"),
        "mixed": _plain_gzip(text[:1_000_000] + noise + text[1_000_000:2_000_000] + b"\0" * 500_000 + noise[:70_000] + text[2_000_000:], 6),
        "huffman_only": _plain_gzip(text[:800_000], 6, zlib.Z_HUFFMAN_ONLY),
        "rle": _plain_gzip(text, 6, zlib.Z_RLE),
        "fixed_only": _plain_gzip(text[:500_000], 6, zlib.Z_FIXED),            # no dynamic block to find: zlib
        "stored_only": _plain_gzip(noise * 3, 0),
        "two_members": gzip.compress(text[:900_000]) + gzip.compress(text[900_000:1_800_000]),
        "python_gzip": gzip.compress(text, 6),
    }
    for label, blob in cases.items():
        path = str(tmp_path / (label + ".gz"))
        with open(path, "wb") as f:
            f.write(blob)
        want = gzip.decompress(blob)
        for threads in (2, 5):
            t = _capi.HostText(path, threads)
            assert t.bytes() == want, (label, threads)
            t.close()
    # the parallel path is really the one that ran on the regular streams (its stage line goes to stderr), and not on the others
    # damage: the same error as zlib on one thread
    blob = bytearray(cases["level6"])
    blob[len(blob) // 2] ^= 0x10
    bad = str(tmp_path / "bad.gz")
    with open(bad, "wb") as f:
        f.write(bytes(blob))
    with pytest.raises(_capi.FhxError) as e1:
        _capi.HostText(bad, 4)
    monkeypatch.setenv("FHX_SERIAL_GUNZIP", "1")
    with pytest.raises(_capi.FhxError) as e2:
        _capi.HostText(bad, 4)
    assert str(e1.value) == str(e2.value)
    cut = str(tmp_path / "cut.gz")
    with open(cut, "wb") as f:
        f.write(cases["level6"][:-5000])
    monkeypatch.delenv("FHX_SERIAL_GUNZIP")
    with pytest.raises(_capi.FhxError):
        _capi.HostText(cut, 4)


def test_parallel_gunzip_reports_its_stages(tmp_path, monkeypatch, capfd):
    import random
    rng = random.Random(2)
    text = "".join("chr1\t%d\tchr1\t%d\t%d\n" % (rng.randrange(10 ** 8), rng.randrange(10 ** 8), rng.randrange(99)) for _ in range(200_000)).encode()
    path = str(tmp_path / "c.gz")
    with gzip.open(path, "wb", compresslevel=6) as f:
        f.write(text)
    monkeypatch.setenv("FHX_PGUNZIP_MIN", "0")
    monkeypatch.setenv("FHX_PGUNZIP_CHUNK", "150000")
    monkeypatch.setenv("FHX_TIMING", "1")
    t = _capi.HostText(path, 4)
    assert t.bytes() == text
    t.close()
    assert "parallel gunzip:" in capfd.readouterr().err


@pytest.mark.parametrize("seed", range(int(os.environ.get("FHX_FUZZ_SEEDS", "0:3").split(":")[0]),
                                       int(os.environ.get("FHX_FUZZ_SEEDS", "0:3").split(":")[1])))
def test_parallel_gunzip_on_random_streams(seed, tmp_path, monkeypatch):
    """random mixtures of table text, noise, zero runs and repeats far apart, every level and strategy, chunk sizes from 1 KB (many
    guessed block starts, chunks of a block or two) to 1 MB, 2-7 threads: the text equals Python's, whichever path produced it"""
    import random
    import zlib
    rng = random.Random(4000 + seed)
    monkeypatch.setenv("FHX_PGUNZIP_MIN", "0")
    for trial in range(8):
        parts = []
        for _ in range(rng.randrange(1, 7)):
            kind = rng.randrange(5)
            if kind == 0:
                parts.append("".join("chr%d\t%d\tchr%d\t%d\t%d\n" % (rng.randrange(1, 23), rng.randrange(10 ** 8), rng.randrange(1, 23),
                                                                     rng.randrange(10 ** 8), rng.randrange(999))
                                     for _ in range(rng.randrange(100, 40000))).encode())
            elif kind == 1:
                parts.append(bytes(rng.getrandbits(8) for _ in range(rng.randrange(10, 100_000))))
            elif kind == 2:
                parts.append(bytes([rng.randrange(256)]) * rng.randrange(1, 400_000))
            elif kind == 3 and parts:
                parts.append(parts[rng.randrange(len(parts))][:50_000])          # a repeat, possibly beyond the window
            else:
                parts.append(("%d\n" % rng.randrange(10 ** 9)).encode() * rng.randrange(1, 30000))
        data = b"".join(parts)
        blob = _plain_gzip(data, rng.choice([1, 2, 4, 6, 9]), rng.choice([None, None, zlib.Z_FILTERED, zlib.Z_RLE, zlib.Z_HUFFMAN_ONLY]))
        path = str(tmp_path / ("r%d.gz" % trial))
        with open(path, "wb") as f:
            f.write(blob)
        monkeypatch.setenv("FHX_PGUNZIP_CHUNK", str(rng.choice([1024, 5000, 40_000, 300_000, 1_000_000])))
        t = _capi.HostText(path, rng.randrange(2, 8))
        assert t.bytes() == data, (seed, trial)
        t.close()


def _inflate_in_parts(path, n_parts, threads):
    """the protocol of sharded._ingest_stream_parts in one process -> (texts of the parts, combined CRC-32) or None when a part refuses"""
    parts = []
    try:
        for r in range(n_parts):
            try:
                parts.append(_capi.TextPart(path, r, n_parts, threads))
            except _capi.FhxError as e:
                assert e.code == _capi.FHX_ERR_UNSUPPORTED, e
                return None
        assert [p.is_last() for p in parts] == [False] * (n_parts - 1) + [True]
        windows = _capi.chain_windows([p.tail() for p in parts])
        texts, crc = [], 0
        for r, p in enumerate(parts):
            announced = len(p)
            t, c = p.resolve(windows[r])
            b = t.bytes()
            assert len(b) == announced
            n, row = t.first_row_end()
            k = b.find(b"\n")
            assert (n, row) == ((k + 1, b[:k + 1]) if k >= 0 else (-1, b"")) and t.ends_with_newline() == b.endswith(b"\n")
            t.close()
            crc = _capi.crc32_combine(crc, c, len(b)) if r else c
            texts.append(b)
        return texts, crc
    finally:
        for p in parts:
            p.close()


def test_one_plain_gzip_stream_is_inflated_in_parts(tmp_path, monkeypatch):
    """fhx_host_inflate_part / fhx_text_part_tail / fhx_text_part_resolve (csrc/fhx_gunzip.cpp): N parts of the compressed bytes, each
    decoded without the window before it, the windows chained from the parts' tails.  The concatenation is Python's text, the combined
    CRC-32 the file's, for every N, chunk size and level; what the scheme cannot take is refused (the caller has the other routes)."""
    import random
    import zlib
    rng = random.Random(31)
    text = "".join("chr%d\t%d\tchr%d\t%d\t%d\n" % (c, rng.randrange(1, 50000) * 5000 + 2500, c, rng.randrange(1, 50000) * 5000 + 2500, rng.randrange(1, 500))
                   for c in (rng.randrange(1, 23) for _ in range(120_000))).encode()
    noise = bytes(rng.getrandbits(8) for _ in range(200_000))
    monkeypatch.setenv("FHX_PGUNZIP_MIN", "0")
    cases = {"level1": _plain_gzip(text, 1), "level6": _plain_gzip(text, 6), "level9": _plain_gzip(text, 9),
             "mixed": _plain_gzip(text[:900_000] + noise + text[900_000:1_700_000] + b"\0" * 400_000 + text[1_700_000:], 6),
             "no_final_newline": _plain_gzip(text[:-1], 6)}
    for label, blob in cases.items():
        path = str(tmp_path / (label + ".gz"))
        with open(path, "wb") as f:
            f.write(blob)
        want = gzip.decompress(blob)
        for n_parts, chunk, threads in ((1, 100_000, 3), (2, 50_000, 2), (3, 200_000, 4), (5, 20_000, 2), (8, 2_000_000, 1)):
            monkeypatch.setenv("FHX_PGUNZIP_CHUNK", str(chunk))
            got = _inflate_in_parts(path, n_parts, threads)
            assert got is not None, (label, n_parts)
            assert b"".join(got[0]) == want and got[1] == zlib.crc32(want), (label, n_parts, chunk)
    # refused, not mis-inflated: several members, fixed codes only (no block start to find), too small a file
    for label, blob in (("two_members", gzip.compress(text[:900_000]) + gzip.compress(text[900_000:])),
                        ("fixed_only", _plain_gzip(text[:400_000], 6, zlib.Z_FIXED))):
        path = str(tmp_path / (label + ".gz"))
        with open(path, "wb") as f:
            f.write(blob)
        assert _inflate_in_parts(path, 3, 2) is None, label
    monkeypatch.delenv("FHX_PGUNZIP_MIN")
    assert _inflate_in_parts(str(tmp_path / "level6.gz"), 2, 2) is None           # 1 MB: under the size from which it pays


@pytest.mark.parametrize("seed", range(int(os.environ.get("FHX_FUZZ_SEEDS", "0:3").split(":")[0]),
                                       int(os.environ.get("FHX_FUZZ_SEEDS", "0:3").split(":")[1])))
def test_inflate_in_parts_on_random_streams(seed, tmp_path, monkeypatch):
    """as test_parallel_gunzip_on_random_streams, through the parts protocol: whenever every part accepts its share, the parts'
    texts are Python's text and the combined CRC-32 is the trailer's (a wrongly guessed block start ends in a refusal or in a
    CRC-32 that differs - which is what the caller checks)"""
    import random
    import zlib
    rng = random.Random(9000 + seed)
    monkeypatch.setenv("FHX_PGUNZIP_MIN", "0")
    taken = 0
    for trial in range(8):
        parts = []
        for _ in range(rng.randrange(1, 6)):
            kind = rng.randrange(4)
            if kind == 0:
                parts.append("".join("chr%d\t%d\tchr%d\t%d\t%d\n" % (rng.randrange(1, 23), rng.randrange(10 ** 8), rng.randrange(1, 23),
                                                                     rng.randrange(10 ** 8), rng.randrange(999))
                                     for _ in range(rng.randrange(100, 40000))).encode())
            elif kind == 1:
                parts.append(bytes(rng.getrandbits(8) for _ in range(rng.randrange(10, 60_000))))
            elif kind == 2:
                parts.append(bytes([rng.randrange(256)]) * rng.randrange(1, 200_000))
            else:
                parts.append(("%d\n" % rng.randrange(10 ** 9)).encode() * rng.randrange(1, 30000))
        data = b"".join(parts)
        blob = _plain_gzip(data, rng.choice([1, 2, 4, 6, 9]), rng.choice([None, None, zlib.Z_FILTERED, zlib.Z_RLE]))
        path = str(tmp_path / ("p%d.gz" % trial))
        with open(path, "wb") as f:
            f.write(blob)
        monkeypatch.setenv("FHX_PGUNZIP_CHUNK", str(rng.choice([1024, 5000, 40_000, 300_000])))
        got = _inflate_in_parts(path, rng.randrange(1, 7), rng.randrange(1, 5))
        if got is None:
            continue
        taken += 1
        same_crc = got[1] == zlib.crc32(data) and sum(len(b) for b in got[0]) == len(data)
        assert (b"".join(got[0]) == data) == same_crc, trial       # equal text <=> the check the caller makes passes
    assert taken >= 1
