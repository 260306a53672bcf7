"""The device gzip decoder (csrc/fhx_inflate.inc) against zlib.

fhx_debug_inflate_file inflates a file of size-tagged gzip members on the GPU and returns the text; the checker is Python's
zlib/gzip (what the reference reads its inputs with, fithic/fithic.py:404).  Every block type of RFC 1951, every zlib strategy,
the window limits (distance 32 768, length 258), members of all sizes, BGZF containers - and damaged streams, which must come
back as FHX_ERR_UNSUPPORTED (the caller then lets zlib on the host report the error) and never as different text."""
import gzip
import os
import random
import struct
import zlib

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def fh_member(data, level=6, strategy=zlib.Z_DEFAULT_STRATEGY, mem_level=8, name=None, flushes=()):
    """one gzip member whose "FH" extra subfield holds its own size (the container of this library's writers)"""
    co = zlib.compressobj(level, zlib.DEFLATED, -15, mem_level, strategy)
    body = b""
    at = 0
    for cut, mode in flushes:
        body += co.compress(data[at:cut]) + co.flush(mode)
        at = cut
    body += co.compress(data[at:]) + co.flush()
    flags = 4 | (8 if name is not None else 0)
    tail = (name + b"\0") if name is not None else b""
    size = 10 + 2 + 12 + len(tail) + len(body) + 8
    hdr = b"\x1f\x8b\x08" + bytes([flags]) + b"\0\0\0\0" + b"\0\xff" + struct.pack("<H", 12) + b"FH" + struct.pack("<HQ", 8, size) + tail
    return hdr + body + struct.pack("<II", zlib.crc32(data), len(data) & 0xffffffff)


def bgzf_member(data):
    body = zlib.compress(data, 6)[2:-4]
    size = 10 + 2 + 6 + len(body) + 8
    hdr = b"\x1f\x8b\x08\x04" + b"\0\0\0\0" + b"\0\xff" + struct.pack("<H", 6) + b"BC" + struct.pack("<HH", 2, size - 1)
    return hdr + body + struct.pack("<II", zlib.crc32(data), len(data))


def _ctx():
    from fithic_amd import _capi
    return _capi.Context(0)


def _device_text(ctx, tmp_path, blob, cap, name="m.gz"):
    path = str(tmp_path / name)
    with open(path, "wb") as f:
        f.write(blob)
    assert gzip.decompress(blob) is not None
    return ctx.debug_inflate_file(path, cap)


def _rows(rng, n):
    out = []
    for _ in range(n):
        c = rng.randrange(1, 23)
        out.append("chr%d\t%d\tchr%d\t%d\t%d\n" % (c, rng.randrange(1, 50000) * 5000 + 2500, c, rng.randrange(1, 50000) * 5000 + 2500, rng.randrange(1, 500)))
    return "".join(out).encode()


def test_every_block_type_strategy_and_level(tmp_path):
    rng = random.Random(1)
    text = _rows(rng, 60_000)                                # 1.8 MB
    noise = bytes(rng.getrandbits(8) for _ in range(200_000))
    members, want = [], []

    def add(data, **kw):
        members.append(fh_member(data, **kw))
        want.append(data)
    for level in (0, 1, 2, 4, 6, 9):
        add(text, level=level)
    for strategy in (zlib.Z_FIXED, zlib.Z_HUFFMAN_ONLY, zlib.Z_RLE, zlib.Z_FILTERED):
        add(text[:400_000], strategy=strategy)
    add(text[:300_000], mem_level=1)                         # many small dynamic blocks
    add(noise)                                               # zlib stores what does not compress
    add(noise, level=0)
    add(b"")                                                 # a member without text
    add(b"x")
    add(b"\0" * 1_000_000)                                   # distance 1, length 258, over and over
    block = bytes(rng.getrandbits(8) for _ in range(32768))
    add(block * 6, level=9)                                  # matches at the largest distance
    add((block[:32767] + b"!") * 5 + block[:100], level=9)
    add(text[:500_000], flushes=[(1000, zlib.Z_SYNC_FLUSH), (70_000, zlib.Z_FULL_FLUSH), (70_001, zlib.Z_SYNC_FLUSH)])
    add(text[:10_000], name=b"contacts.txt")                 # FNAME after the extra field
    add(bytes(range(256)) * 50, strategy=zlib.Z_FIXED)
    ctx = _ctx()
    blob = b"".join(members)
    got = _device_text(ctx, tmp_path, blob, sum(map(len, want)) + 16)
    assert got == b"".join(want)
    # and one at a time, so that a failure names its case
    for k, (m, w) in enumerate(zip(members, want)):
        assert _device_text(ctx, tmp_path, m, len(w) + 16, "one%d.gz" % k) == w, k
    ctx.close()


def test_bgzf_container_and_many_small_members(tmp_path):
    rng = random.Random(2)
    text = _rows(rng, 100_000)
    members, at = [], 0
    while at < len(text):
        n = rng.choice([1, 7, 300, 65280, 65280, 20000])
        members.append(bgzf_member(text[at:at + n]))
        at += n
    members.append(bgzf_member(b""))                          # bgzip's end-of-file block
    ctx = _ctx()
    assert _device_text(ctx, tmp_path, b"".join(members), len(text) + 16) == text
    ctx.close()


def test_the_library_writer_files_equal_zlib(tmp_path):
    from fithic_amd import _capi
    rng = np.random.default_rng(4)
    n = 700_000
    names = ["chr%d" % k for k in range(1, 23)]
    c1 = np.sort(rng.integers(0, 22, n)).astype(np.int32)
    m1 = (rng.integers(0, 50_000, n) * 5000 + 2500).astype(np.int32)
    m2 = (rng.integers(0, 50_000, n) * 5000 + 2500).astype(np.int32)
    cnt = rng.integers(1, 2000, n).astype(np.int32)
    ctx = _ctx()
    for level in (1, 6):
        path = str(tmp_path / ("c%d.gz" % level))
        _capi.host_write_contacts(path, names, c1, m1, c1, m2, cnt, gzip_level=level, threads=3)
        with gzip.open(path, "rb") as f:
            want = f.read()
        assert ctx.debug_inflate_file(path, len(want) + 16) == want
    ctx.close()


def test_plain_gzip_is_left_to_the_host(tmp_path):
    from fithic_amd import _capi
    path = str(tmp_path / "plain.gz")
    with gzip.open(path, "wb") as f:
        f.write(b"a 1 b 2 3\n")
    ctx = _ctx()
    with pytest.raises(_capi.FhxError) as e:
        ctx.debug_inflate_file(path, 100)
    assert e.value.code == _capi.FHX_ERR_UNSUPPORTED
    ctx.close()


@pytest.mark.parametrize("seed", range(int(os.environ.get("FHX_FUZZ_SEEDS", "0:4").split(":")[0]),
                                       int(os.environ.get("FHX_FUZZ_SEEDS", "0:4").split(":")[1])))
def test_damaged_members_are_refused_not_misread(seed, tmp_path):
    """one member among good ones is damaged - a flipped bit, a changed byte, a wrong CRC or ISIZE, a cut stream: the device
    must either return exactly the text zlib returns (damage that does not matter, e.g. the MTIME bytes) or refuse"""
    from fithic_amd import _capi
    rng = random.Random(900 + seed)
    ctx = _ctx()
    texts = [_rows(rng, rng.choice([50, 2000, 20000])) for _ in range(3)]
    kinds = [dict(level=1), dict(level=9), dict(strategy=zlib.Z_FIXED), dict(level=0), dict(mem_level=1)]
    refused = same = 0
    for trial in range(40):
        members = [bytearray(fh_member(t, **rng.choice(kinds))) for t in texts]
        k = rng.randrange(3)
        m = members[k]
        how = rng.randrange(5)
        if how == 0:
            at = rng.randrange(24, len(m) - 8)
            m[at] ^= 1 << rng.randrange(8)
        elif how == 1:
            at = rng.randrange(24, len(m))
            m[at] = rng.randrange(256)
        elif how == 2:
            m[-8 + rng.randrange(4)] ^= 0xff                                    # CRC-32
        elif how == 3:
            struct.pack_into("<I", m, len(m) - 4, max(0, len(texts[k]) + rng.choice([-1, 1, 1000, -1000])))   # ISIZE
        else:
            cut = rng.randrange(1, min(200, len(m) - 40))
            del m[24:24 + cut]                                                  # bytes missing from the stream
            struct.pack_into("<Q", m, 16, len(m))
        blob = b"".join(bytes(x) for x in members)
        path = str(tmp_path / ("d%d.gz" % trial))
        with open(path, "wb") as f:
            f.write(blob)
        try:
            want = gzip.decompress(blob)
        except Exception:                                                       # noqa: BLE001 - zlib.error, BadGzipFile, EOFError
            want = None
        try:
            got = ctx.debug_inflate_file(path, sum(map(len, texts)) + 4096)
        except _capi.FhxError as e:
            assert e.code in (_capi.FHX_ERR_UNSUPPORTED, _capi.FHX_ERR_ARG), e
            refused += 1
            continue
        assert want is not None and got == want, (seed, trial, how)
        same += 1
    assert refused > 20
    ctx.close()


def test_contacts_file_inflated_and_parsed_on_the_device(tmp_path):
    """fhx_ingest_contacts_file = device inflate + device parse: rows and names of the host reader"""
    from fithic_amd import _capi
    from fithic_amd.engine import Engine
    rng = np.random.default_rng(8)
    n = 900_000
    names = ["chr%d" % k for k in range(1, 23)] + ["chrX"]
    c1 = np.sort(rng.integers(0, 23, n)).astype(np.int32)
    c2 = np.where(rng.random(n) < 0.95, c1, rng.integers(0, 23, n)).astype(np.int32)
    m1 = (rng.integers(0, 50_000, n) * 5000 + 2500).astype(np.int32)
    m2 = (rng.integers(0, 50_000, n) * 5000 + 2500).astype(np.int32)
    cnt = rng.integers(1, 2000, n).astype(np.int32)
    path = str(tmp_path / "c.gz")
    _capi.host_write_contacts(path, names, c1, m1, c2, m2, cnt, threads=4)
    eng = Engine(0)
    eng.configure(5000, 0, None, n_bins=10, mapp_thres=1, mode="All", bias_low=0.5, bias_up=2.0)
    k, seen = eng.ctx.ingest_contacts_file(path)
    want_names, cols, _ = _capi.host_read_table(path, 0, 2, want_float=False)
    assert k == n and seen == want_names
    eng.commit_contacts_text(np.arange(len(seen), dtype=np.int32), k)
    for g, w in zip(eng.ctx.fetch_pairs(n=n), [cols[i] for i in range(5)]):
        assert np.array_equal(g, w)
    # a text outside the device grammar: refused = 2; a plain gzip file: refused = 1
    odd = str(tmp_path / "odd.gz")
    with open(odd, "wb") as f:
        f.write(fh_member(b"a 1 b 2 1e3\n") + fh_member(b"a 1 b 2 3\n"))
    with pytest.raises(_capi.FhxError) as e:
        eng.ctx.ingest_contacts_file(odd)
    assert e.value.code == _capi.FHX_ERR_UNSUPPORTED and e.value.refused == 2
    plain = str(tmp_path / "plain.gz")
    with gzip.open(plain, "wb") as f:
        f.write(b"a 1 b 2 3\n")
    with pytest.raises(_capi.FhxError) as e:
        eng.ctx.ingest_contacts_file(plain)
    assert e.value.code == _capi.FHX_ERR_UNSUPPORTED and e.value.refused == 1
    eng.close()


def test_command_line_on_a_bgzf_contacts_file_writes_the_reference_files(tmp_path, monkeypatch, capsys):
    """the golden hESC contacts recompressed as BGZF blocks (what bgzip writes): the CLI inflates and parses them on the GPU and
    writes the reference's files; the stage line says which path ran"""
    import hashlib
    from conftest import load_case, case_args
    from fithic_amd import cli
    meta, _ = load_case("f1_bias")
    kw = case_args(meta)
    with gzip.open(kw["contacts"], "rb") as f:
        text = f.read()
    rng = random.Random(3)
    members, at = [], 0
    while at < len(text):
        n = rng.choice([65280, 65280, 30000, 1000])
        members.append(bgzf_member(text[at:at + n]))                  # blocks end anywhere, also inside a line
        at += n
    members.append(bgzf_member(b""))
    contacts = str(tmp_path / "contacts.bgzf.gz")
    with open(contacts, "wb") as f:
        f.write(b"".join(members))
    monkeypatch.setenv("FHX_TIMING", "1")
    out = tmp_path / "out"
    cli.main(["-i", contacts, "-f", kw["frags"], "-o", str(out), "-l", "G", "-t", kw["bias_path"]] + meta["argv"])
    printed = capsys.readouterr().out
    assert "device inflate + parse" in printed and "(device parser)" in printed
    tag = ".res%d" % kw["resolution"]
    for pi in range(1, meta["n_passes"] + 1):
        with gzip.open(os.path.join(str(out), "G.spline_pass%d%s.significances.txt.gz" % (pi, tag)), "rb") as f:
            assert hashlib.md5(f.read()).hexdigest() == meta["sig_md5_pass%d" % pi]
        with open(os.path.join(str(out), "G.fithic_pass%d%s.txt" % (pi, tag))) as f:
            assert f.read() == meta["fithic_pass%d_txt" % pi]
