"""fhx_ingest_contacts_text: the contacts file parsed by the GPU (csrc/fhx_ingest.inc).

The checker is the library's host parser (fhx_host_read_table), whose grammar test_native_io.py pins against Python's own
`line.split()` / int() / int(float()) - the reference's loop (fithic/fithic.py:404-417).  For every text, the device path must
either give exactly the host parser's rows and names (names in order of first appearance), or say FHX_ERR_UNSUPPORTED and
leave the text to fhx_host_parse_text; it may never give different rows."""
import gzip
import os
import random

import numpy as np
import pytest

from conftest import load_case, case_args

pytestmark = pytest.mark.gpu


def _write(tmp_path, name, data):
    path = str(tmp_path / name)
    with gzip.open(path, "wb", compresslevel=1) as f:
        f.write(data if isinstance(data, bytes) else data.encode())
    return path


def _engine(resolution=0):
    from fithic_amd.engine import Engine
    eng = Engine(0)
    eng.configure(resolution, 0, None, n_bins=10, mapp_thres=1, mode="All", bias_low=0.5, bias_up=2.0)
    return eng


def _host(path):
    """(names, 5 columns) of the host parser, or the FhxError it raises"""
    from fithic_amd import _capi
    try:
        names, cols, _ = _capi.host_read_table(path, 0, 2, want_float=False)
    except _capi.FhxError as e:
        return e
    return names, [cols[k] for k in range(5)]


def _device(eng, path, threads=3):
    """(names, 5 columns) through the device parser with identity ids, "unsupported", and the text for the fallback"""
    from fithic_amd import _capi
    text = _capi.HostText(path, threads)
    try:
        n, names = eng.ctx.ingest_contacts_text(text, threads)
    except _capi.FhxError as e:
        assert e.code == _capi.FHX_ERR_UNSUPPORTED, e
        return "unsupported", text
    eng.commit_contacts_text(np.arange(len(names), dtype=np.int32), n)
    return (names, eng.ctx.fetch_pairs(n=n)), text


def _same(got, want):
    assert got[0] == want[0]
    for g, w in zip(got[1], want[1]):
        assert g.dtype == np.int32 and np.array_equal(g, w)


REGULAR = [
    ("plain", "chr1\t100\tchr1\t300\t5\nchr1\t100\tchr2\t700\t1\nchr2\t500\tchr2\t700\t12\n"),
    ("no final newline", "a 1 b 2 3\nb 4 a 5 6"),
    ("one line", "x\t5\tx\t15\t2\n"),
    ("dos line ends", "a 1 b 2 3\r\nb 4 a 5 6\r\n"),
    ("runs of separators, leading and trailing blanks", "  a \t 1   b\x0b2\x0c3 \t\n\x1ca\x1d7\x1eb\x1f8 9  \n"),
    ("signs and leading zeros", "a +5 b +3 0007\na -0 b +0000000012 0\na 2147483647 b 0 2147483647\n"),
    ("names with punctuation", "chr1_KI270706v1_random 1 HLA-DRB1*15:01:01:01 2 3\n#c 1 \"q\" 2 3\n1 1 2 2 3\n"),
    ("63-byte name", "n" * 63 + " 1 b 2 3\n"),
    ("decimal counts", "a 1 b 2 3.0\na 1 b 2 1.7\na 1 b 2 5.\na 1 b 2 .9\na 1 b 2 0.99999999999999\na 1 b 2 2147483647.99999\n"
                       "a 1 b 2 99999.9999999999\na 1 b 2 000.5\n"),
]


@pytest.mark.parametrize("label,text", REGULAR, ids=[c[0] for c in REGULAR])
def test_regular_texts_give_the_host_parsers_rows(label, text, tmp_path):
    path = _write(tmp_path, "c.gz", text)
    eng = _engine()
    got, handle = _device(eng, path)
    handle.close()
    assert got != "unsupported"
    _same(got, _host(path))
    eng.close()


IRREGULAR = [
    ("sixteen digits in a count", "a 1 b 2 0.9999999999999999\n"),
    ("a point alone", "a 1 b 2 .\n"),
    ("two points", "a 1 b 2 1.2.3\n"),
    ("negative count", "a 1 b 2 -1.5\n"),
    ("exponent count", "a 1 b 2 1e3\n"),
    ("signed count", "a 1 b 2 +3\n"),
    ("underscore in a number", "a 1_000 b 2 3\n"),
    ("eleven digits", "a 00000000001 b 2 3\n"),
    ("midpoint beyond int32", "a 2147483648 b 2 3\n"),
    ("count beyond int32", "a 1 b 2 4294967296\n"),
    ("six tokens", "a 1 b 2 3 4\n"),
    ("four tokens", "a 1 b 2\n"),
    ("empty line in the middle", "a 1 b 2 3\n\na 1 b 2 3\n"),
    ("blank last line", "a 1 b 2 3\n   \n"),
    ("a lone newline", "\n"),
    ("lone carriage return", "a 1 b 2 3\ra 1 b 2 3\n"),
    ("carriage return at the end of the text", "a 1 b 2 3\r"),
    ("NUL in a name", "a\x00 1 b 2 3\n"),
    ("non-ASCII name", "chré 1 b 2 3\n"),
    ("64-byte name", "n" * 64 + " 1 b 2 3\n"),
    ("letters in a midpoint", "a 1x b 2 3\n"),
    ("hex", "a 0x10 b 2 3\n"),
    ("a very long line", "a 1 b 2 3" + " " * 5000 + "\n"),
]


@pytest.mark.parametrize("label,text", IRREGULAR, ids=[c[0] for c in IRREGULAR])
def test_other_texts_are_left_to_the_host_parser(label, text, tmp_path):
    """the device parser refuses; fhx_host_parse_text of the same inflated text then equals fhx_host_read_table of the file -
    rows, or the error the reference's ValueError maps to"""
    from fithic_amd import _capi
    path = _write(tmp_path, "c.gz", text)
    eng = _engine()
    got, handle = _device(eng, path)
    assert got == "unsupported"
    want = _host(path)
    if isinstance(want, Exception):
        with pytest.raises(_capi.FhxError) as e:
            _capi.host_parse_text(handle, 0, 2, want_float=False)
        assert e.value.code == want.code and str(e.value) == str(want)
    else:
        names, cols, _ = _capi.host_parse_text(handle, 0, 2, want_float=False)
        _same((names, [cols[k] for k in range(5)]), want)
    handle.close()
    # the context is as it was: a regular file can follow
    ok = _write(tmp_path, "ok.gz", "a 1 b 2 3\n")
    got, handle = _device(eng, ok)
    handle.close()
    _same(got, _host(ok))
    eng.close()


def test_negative_midpoints_parse_and_are_refused_like_host_rows(tmp_path):
    """int("-3") is a number to the reader; the engine refuses such rows from either parser with the same message"""
    from fithic_amd import _capi
    path = _write(tmp_path, "c.gz", "a 5 b -3 7\na -2147483648 b 1 1\n")
    eng = _engine()
    text = _capi.HostText(path, 1)
    n, names = eng.ctx.ingest_contacts_text(text)
    text.close()
    assert (n, names) == (2, ["a", "b"])
    with pytest.raises(_capi.FhxError) as dev:
        eng.commit_contacts_text(np.arange(2, dtype=np.int32), n)
    names, cols = _host(path)
    assert cols[3].tolist() == [-3, 1] and cols[1].tolist() == [5, -2147483648]
    with pytest.raises(_capi.FhxError) as host:
        eng.load_contacts(*cols)
    assert str(dev.value) == str(host.value)
    eng.close()


def test_empty_file(tmp_path):
    path = _write(tmp_path, "c.gz", "")
    eng = _engine()
    got, handle = _device(eng, path)
    handle.close()
    assert got[0] == [] and all(len(c) == 0 for c in got[1])
    eng.close()


def test_caller_ids_and_row_subsets(tmp_path):
    """the commit maps names into the caller's id space; fhx_fetch_pairs returns any rows, in the order asked for"""
    from fithic_amd import _capi
    rng = np.random.default_rng(5)
    names = ["chr%d" % k for k in range(1, 8)]
    n = 5000
    c1, c2 = rng.integers(0, 7, n), rng.integers(0, 7, n)
    m1, m2 = rng.integers(0, 500, n) * 1000 + 500, rng.integers(0, 500, n) * 1000 + 500
    cnt = rng.integers(1, 90, n)
    path = _write(tmp_path, "c.gz", "".join("%s\t%d\t%s\t%d\t%d\n" % (names[a], b, names[c], d, e) for a, b, c, d, e in zip(c1, m1, c2, m2, cnt)))
    for resolution in (1000, 0):
        eng = _engine(resolution)
        text = _capi.HostText(path, 2)
        k, seen = eng.ctx.ingest_contacts_text(text)
        text.close()
        assert k == n and sorted(seen) == sorted(names)
        ids = np.array([100 + names.index(s) for s in seen], np.int32)           # an id space of the caller's own
        eng.commit_contacts_text(ids, k)
        got = eng.ctx.fetch_pairs(n=n)
        for g, w in zip(got, (c1 + 100, m1, c2 + 100, m2, cnt)):
            assert np.array_equal(g, w)
        rows = rng.integers(0, n, 300)
        sub = eng.ctx.fetch_pairs(rows=rows)
        for g, w in zip(sub, got):
            assert np.array_equal(g, w[rows])
        with pytest.raises(_capi.FhxError):
            eng.ctx.fetch_pairs(rows=np.array([n]))
        eng.close()


def test_thousands_of_names_in_order_of_first_appearance(tmp_path):
    """more names than a wave has lanes, every block meeting new ones, hash-table probing under contention"""
    rng = random.Random(11)
    pool = ["scaffold_%d_%s" % (k, "x" * rng.randrange(0, 30)) for k in range(6000)]
    lines = []
    for _ in range(120_000):
        a, b = rng.choice(pool), rng.choice(pool)
        lines.append("%s %d %s %d %d\n" % (a, rng.randrange(1, 10 ** 6), b, rng.randrange(1, 10 ** 6), rng.randrange(1, 50)))
    path = _write(tmp_path, "c.gz", "".join(lines))
    eng = _engine()
    got, handle = _device(eng, path)
    handle.close()
    assert got != "unsupported" and len(got[0]) > 5900
    _same(got, _host(path))
    eng.close()


def test_more_names_than_the_table_takes_go_to_the_host(tmp_path):
    lines = ["n%d 1 n%d 2 3\n" % (2 * k, 2 * k + 1) for k in range(4200)]          # 8400 names > 8192
    path = _write(tmp_path, "c.gz", "".join(lines))
    eng = _engine()
    got, handle = _device(eng, path)
    handle.close()
    assert got == "unsupported"
    eng.close()


def test_several_members_and_pieces(tmp_path):
    """a file of many size-tagged members is inflated by several threads into several pieces; lines are found across the
    16 KB blocks of the device pass wherever the pieces end"""
    from fithic_amd import _capi
    rng = np.random.default_rng(3)
    n = 1_200_000
    names = ["chr%d" % k for k in range(1, 24)]
    c1 = np.sort(rng.integers(0, 23, n)).astype(np.int32)
    c2 = np.where(rng.random(n) < 0.9, c1, rng.integers(0, 23, n)).astype(np.int32)
    m1 = (rng.integers(0, 50_000, n) * 5000 + 2500).astype(np.int32)
    m2 = (rng.integers(0, 50_000, n) * 5000 + 2500).astype(np.int32)
    cnt = rng.integers(1, 2000, n).astype(np.int32)
    path = str(tmp_path / "c.gz")
    _capi.host_write_contacts(path, names, c1, m1, c2, m2, cnt, threads=4)
    eng = _engine(5000)
    got, handle = _device(eng, path, threads=5)
    assert len(handle) > 20 * n
    handle.close()
    assert got != "unsupported"
    _same(got, _host(path))
    eng.close()


@pytest.mark.parametrize("seed", range(int(os.environ.get("FHX_FUZZ_SEEDS", "0:6").split(":")[0]),
                                       int(os.environ.get("FHX_FUZZ_SEEDS", "0:6").split(":")[1])))
def test_random_texts_equal_the_host_parser_or_are_refused(seed, tmp_path):
    """random files from a grammar that mixes regular lines with everything the host reader's fuzz found interesting; most
    files are regular, some hold one irregular line somewhere"""
    rng = random.Random(7000 + seed)
    names = ["chr" + str(k) for k in range(rng.randrange(1, 30))] + ["X", "Y_random", "M" * rng.randrange(1, 63)]
    seps = [" ", "\t", "  ", " \t", "\x1c", "\x0b"]
    odd = ["1_0", "1.5", "1e2", "+7", "nan", "0x1f", "", "99999999999", "٣", "1\r2", "١"]
    for trial in range(12):
        lines = []
        n = rng.choice([1, 3, 700, 5000])
        poison = rng.randrange(n) if rng.random() < 0.4 else -1
        for i in range(n):
            tok = [rng.choice(names), str(rng.randrange(0, 3 * 10 ** 8)), rng.choice(names), str(rng.randrange(0, 3 * 10 ** 8)),
                   str(rng.randrange(0, 10 ** rng.randrange(1, 10))) + rng.choice(["", "", ".", ".0", ".5", ".%d" % rng.randrange(10 ** 6)])]
            if i == poison:
                k = rng.randrange(6)
                if k == 5:
                    tok = tok[:rng.randrange(0, 5)] if rng.random() < 0.5 else tok + ["extra"]
                else:
                    tok[k] = rng.choice(odd)
                    if tok[k] == "":
                        del tok[k]
            end = rng.choice(["\n", "\n", "\n", "\r\n"]) if i < n - 1 or rng.random() < 0.7 else ""
            lines.append((rng.choice(["", "", " "]) + rng.choice(seps).join(tok) + rng.choice(["", "", "\t"]) + end))
        path = _write(tmp_path, "c%d.gz" % trial, "".join(lines).encode("utf-8"))
        eng = _engine()
        got, handle = _device(eng, path)
        want = _host(path)
        if got != "unsupported":
            assert not isinstance(want, Exception), (seed, trial, str(want))
            _same(got, want)
        elif poison < 0:
            raise AssertionError("a regular file was refused (seed %d trial %d)" % (seed, trial))
        handle.close()
        eng.close()


@pytest.mark.parametrize("case", ["f1_bias", "f2_all", "f6_quirk_all", "f11_offgrid_all", "f8_nonfixed_all"])
def test_command_line_uses_the_device_parser_and_either_parser_gives_the_reference_files(case, tmp_path, monkeypatch, capsys):
    """the golden inputs through the CLI twice: the default run must have parsed on the GPU, FHX_HOST_READER=1 on the host; both
    write the reference's files (md5 of the decompressed significances, the pass tables)"""
    import hashlib
    from fithic_amd import cli
    meta, _ = load_case(case)
    kw = case_args(meta)
    monkeypatch.setenv("FHX_TIMING", "1")
    for which in ("device", "host"):
        out = tmp_path / which
        out.mkdir()
        if which == "host":
            monkeypatch.setenv("FHX_HOST_READER", "1")
        else:
            monkeypatch.delenv("FHX_HOST_READER", raising=False)
        argv = ["-i", kw["contacts"], "-f", kw["frags"], "-o", str(out), "-l", "G"] + meta["argv"]
        if kw["bias_path"]:
            argv += ["-t", kw["bias_path"]]
        cli.main(argv)
        assert "(%s parser)" % which in capsys.readouterr().out
        tag = (".res%d" % kw["resolution"]) if kw["resolution"] else ""
        for pi in range(1, meta["n_passes"] + 1):
            with gzip.open(os.path.join(str(out), "G.spline_pass%d%s.significances.txt.gz" % (pi, tag)), "rb") as f:
                assert hashlib.md5(f.read()).hexdigest() == meta["sig_md5_pass%d" % pi]
            with open(os.path.join(str(out), "G.fithic_pass%d%s.txt" % (pi, tag))) as f:
                assert f.read() == meta["fithic_pass%d_txt" % pi]


def test_host_writer_after_device_parser_fetches_the_columns_back(tmp_path, monkeypatch, capsys):
    """FHX_HOST_WRITER=1 (or a row the device formatter does not take) needs the identity columns on the host: they come back
    through fhx_fetch_pairs, and the file is the reference's.  A file with integer counts, so that the device parser takes it."""
    import hashlib
    from fithic_amd import cli
    meta, _ = load_case("f1_bias")
    kw = case_args(meta)
    # the golden contacts with their counts truncated as the reader would: the same rows for both runs below
    lines = []
    with gzip.open(kw["contacts"], "rt") as f:
        for ln in f:
            a, b, c, d, e = ln.split()
            lines.append("%s\t%s\t%s\t%s\t%d\n" % (a, b, c, d, int(float(e))))
    contacts = _write(tmp_path, "contacts.gz", "".join(lines))
    monkeypatch.setenv("FHX_TIMING", "1")
    digests = {}
    for which in ("device", "host"):
        out = tmp_path / which
        out.mkdir()
        if which == "host":
            monkeypatch.setenv("FHX_HOST_READER", "1")
        else:
            monkeypatch.delenv("FHX_HOST_READER", raising=False)
        monkeypatch.setenv("FHX_HOST_WRITER", "1")
        argv = ["-i", contacts, "-f", kw["frags"], "-o", str(out), "-l", "G", "-t", kw["bias_path"]] + meta["argv"]
        cli.main(argv)
        assert "(%s parser)" % which in capsys.readouterr().out
        tag = ".res%d" % kw["resolution"]
        digests[which] = []
        for pi in range(1, meta["n_passes"] + 1):
            with gzip.open(os.path.join(str(out), "G.spline_pass%d%s.significances.txt.gz" % (pi, tag)), "rb") as f:
                digests[which].append(hashlib.md5(f.read()).hexdigest())
    assert digests["device"] == digests["host"] == [meta["sig_md5_pass%d" % pi] for pi in range(1, meta["n_passes"] + 1)]


@pytest.mark.parametrize("seed", range(int(os.environ.get("FHX_FUZZ_SEEDS", "0:8").split(":")[0]),
                                       int(os.environ.get("FHX_FUZZ_SEEDS", "0:8").split(":")[1])))
def test_rows_of_a_text_cut_anywhere_are_each_parsed_by_one_part(seed, tmp_path):
    """fhx_ingest_contacts_text_own + sharded.rows_owned_by_parts (the ranks of `fithic --gpus N` inflating one gzip stream
    together): the text is cut at arbitrary bytes - inside a row, right behind a newline, right in front of one, inside a name -
    and every part parses the rows whose first byte it holds, the last one completed from the next part's head.  All parts' rows,
    in order, are the host parser's rows of the whole text; names in order of first appearance inside each part."""
    from fithic_amd import _capi, sharded
    rng = np.random.default_rng(700 + seed)
    names = ["chr%d" % k for k in range(1, 6)] + ["chrX", "scaffold_%d" % seed]
    n = int(rng.integers(50, 4000))
    rows = ["%s\t%d\t%s\t%d\t%d\n" % (names[rng.integers(len(names))], rng.integers(0, 10 ** int(rng.integers(1, 9))),
                                     names[rng.integers(len(names))], rng.integers(0, 10 ** 8), rng.integers(1, 500)) for _ in range(n)]
    text = "".join(rows).encode()
    if seed % 3 == 2:
        text = text[:-1]                                           # a last row without its newline
    whole = _write(tmp_path, "whole.gz", text)
    want_names, want_cols = _host(whole)
    n_parts = int(rng.integers(2, 7))
    # cuts: random bytes, plus on purpose the byte behind a newline and the newline itself
    nl = [i for i, b in enumerate(text) if b == 10]
    cuts = sorted(set([int(rng.integers(1, len(text) - 1)) for _ in range(n_parts - 1)]))
    if seed % 2 == 0 and len(cuts) >= 1:
        cuts[0] = nl[len(nl) // 3] + 1                             # the part before ends a row
    if seed % 4 == 1 and len(cuts) >= 2:
        cuts[1] = nl[len(nl) // 2]                                 # the part begins with the newline of the row before
    cuts = sorted(set(c for c in cuts if 0 < c < len(text)))
    parts = [text[a:b] for a, b in zip([0] + cuts, cuts + [len(text)])]
    if any(b"\n" not in p for p in parts):
        pytest.skip("a part without a newline: the driver takes another route")
    texts = [_capi.HostText(_write(tmp_path, "part%d.gz" % k, p), 2) for k, p in enumerate(parts)]
    firsts = [t.first_row_end() for t in texts]
    ends = [t.ends_with_newline() for t in texts]
    assert [f[1] for f in firsts] == [p[:p.index(b"\n") + 1] for p in parts] and ends == [p.endswith(b"\n") for p in parts]
    own = sharded.rows_owned_by_parts(firsts, ends)
    got = [[] for _ in range(5)]
    for t, (skip, extra) in zip(texts, own):
        eng = _engine()
        k, seen = eng.ctx.ingest_contacts_text_own(t, skip, extra)
        ids = np.array([want_names.index(s) for s in seen], np.int32)     # (the whole text's id space)
        eng.commit_contacts_text(ids, k)
        cols = eng.ctx.fetch_pairs(n=k)
        for c, g in zip(got, cols):
            c.append(g)
        eng.close()
        t.close()
    for k in range(5):
        assert np.array_equal(np.concatenate(got[k]), want_cols[k]), k
