"""fhx_write_significances_device: the significances file formatted and deflated by the GPU.

The decompressed bytes must equal, byte for byte, the reference's rows ("%s\\t%d\\t%s\\t%d\\t%d\\t%e\\t%e\\t%e\\t%e\\t%f\\n" per emitted
row, fithic/fithic.py:1167-1213): checked against Python's own formatting of the fetched columns (what the reference runs),
against the library's host writer, and - through the CLI tests of test_gpu_parity.py, which now go through this writer - against
the reference-generated golden files.  The container must be what every gzip reader accepts (Python's gzip verifies CRC-32
and ISIZE of every member) and what the library's parallel reader expects (every member carries its size)."""
import gzip
import os
import struct

import numpy as np
import pytest

from conftest import load_case, case_args

pytestmark = pytest.mark.gpu

HEADER = "chr1\tfragmentMid1\tchr2\tfragmentMid2\tcontactCount\tp-value\tq-value\tbias1\tbias2\tExpCC\n"


def _members(path):
    """(offset, size) of every gzip member, walking the sizes of the "FH" extra subfields."""
    out = []
    with open(path, "rb") as f:
        data = f.read()
    at = 0
    while at < len(data):
        assert data[at:at + 4] == b"\x1f\x8b\x08\x04", "member %d at %d has no extra field" % (len(out), at)
        xlen, = struct.unpack_from("<H", data, at + 10)
        assert xlen == 12 and data[at + 12:at + 16] == b"FH\x08\x00"
        size, = struct.unpack_from("<Q", data, at + 16)
        assert size >= 24 + 8 and at + size <= len(data)
        out.append((at, size))
        at += size
    assert at == len(data)
    return out


def _python_text(names, cols, v, mode, L, U):
    names = np.asarray(names)
    inter = cols[0] != cols[2]
    d = np.abs(cols[1].astype(np.int64) - cols[3].astype(np.int64))
    emit = np.where(inter, mode in ("All", "interOnly"), (mode in ("All", "intraOnly")) & (d >= L) & (d <= U))
    rows = [HEADER]
    for i in np.flatnonzero(emit).tolist():
        rows.append("%s\t%d\t%s\t%d\t%d\t%e\t%e\t%e\t%e\t%f\n" % (names[cols[0][i]], cols[1][i], names[cols[2][i]], cols[3][i], cols[4][i],
                                                                 v["p"][i], v["q"][i], v["b1"][i], v["b2"][i], v["expcc"][i]))
    return "".join(rows).encode(), int(emit.sum())


def _engine_with_case(case):
    from fithic_amd import tables
    from fithic_amd.engine import Engine
    meta, _ = load_case(case)
    kw = case_args(meta)
    chroms = tables.ChromIndex()
    con = tables.read_contacts(kw["contacts"], chroms)
    frag = tables.read_fragments(kw["frags"], chroms)
    bias = tables.read_bias(kw["bias_path"], chroms) if kw["bias_path"] else None
    eng = Engine(0)
    eng.configure(kw["resolution"], kw["L"], kw["U"], n_bins=kw["n_bins"], mapp_thres=kw["mapp_thres"], mode=kw["mode"],
                  bias_low=kw["tL"], bias_up=kw["tU"])
    eng.load_fragments(*frag, chroms.sort_rank())
    if bias:
        eng.load_bias(*bias)
    eng.load_contacts(con.chr1, con.mid1, con.chr2, con.mid2, con.count)
    return eng, kw, chroms, con


def test_device_writer_writes_through_links_and_replaces_plain_files(tmp_path):
    """The writer publishes a plain file by renaming a finished temporary over it (an earlier output survives a failure), but a
    target that is a symlink or a device node is written THROUGH, as fopen(path, "wb") did - renaming over it would replace the node."""
    eng, kw, chroms, con = _engine_with_case("f2_all")
    eng.run_pass(collect=False)
    plain = str(tmp_path / "plain.gz")
    rows, nbytes = eng.ctx.write_significances_device(plain, chroms.names)
    with open(plain, "rb") as f:
        want = f.read()
    with open(plain, "wb") as f:
        f.write(b"an older, longer output" * 100)               # replaced as a whole
    assert eng.ctx.write_significances_device(plain, chroms.names) == (rows, nbytes)
    real, link = str(tmp_path / "real.gz"), str(tmp_path / "link.gz")
    with open(real, "wb") as f:
        f.write(b"x")
    os.symlink(real, link)
    assert eng.ctx.write_significances_device(link, chroms.names) == (rows, nbytes)
    assert os.path.islink(link) and os.readlink(link) == real
    for path in (plain, real):
        with open(path, "rb") as f:
            assert f.read() == want
    assert eng.ctx.write_significances_device("/dev/null", chroms.names) == (rows, nbytes)
    assert os.path.exists("/dev/null") and not os.path.isfile("/dev/null")
    assert sorted(os.listdir(str(tmp_path))) == ["link.gz", "plain.gz", "real.gz"]      # no temporary left behind
    eng.close()


@pytest.mark.parametrize("case", ["f1_bias", "f2_all", "f2_intra", "f6_quirk_all", "f11_offgrid_all", "f8_nonfixed_all"])
def test_device_writer_equals_python_formatting_on_the_golden_inputs(case, tmp_path):
    from fithic_amd import _capi
    eng, kw, chroms, con = _engine_with_case(case)
    eng.run_pass(collect=False)
    v = eng.fetch(p=True, q=True, expcc=True, bias=True)
    cols = [con.chr1, con.mid1, con.chr2, con.mid2, con.count]
    path = str(tmp_path / "dev.gz")
    rows, nbytes = eng.ctx.write_significances_device(path, chroms.names, *cols)
    U = kw["U"] if kw["U"] not in (None, float("inf")) else (1 << 62)
    want, n_emit = _python_text(chroms.names, cols, v, kw["mode"], kw["L"], U)
    with gzip.open(path, "rb") as f:
        got = f.read()
    assert rows == n_emit
    assert got == want
    # without the identity columns: rebuilt on the device from the rows ingest stored - the same file, byte for byte
    rebuilt = str(tmp_path / "rebuilt.gz")
    assert eng.ctx.write_significances_device(rebuilt, chroms.names) == (rows, nbytes)
    with open(rebuilt, "rb") as f, open(path, "rb") as g:
        assert f.read() == g.read()
    host = str(tmp_path / "host.gz")
    from fithic_amd.engine import MODES
    _capi.host_write_significances(host, chroms.names, *cols, v["p"], v["q"], v["b1"], v["b2"], v["expcc"], MODES[kw["mode"]], kw["L"], U)
    with gzip.open(host, "rb") as f:
        assert f.read() == got
    mem = _members(path)
    assert sum(s for _, s in mem) + 0 == os.path.getsize(path) and nbytes == sum(s for _, s in mem[1:])
    eng.close()


def test_many_members_with_skipped_rows_and_both_row_kinds(tmp_path):
    """cis + trans rows of the C5-shaped generator (-x All with a distance window, so intra rows outside it are NOT written):
    about 40 members, compared with the host writer's text; the members' size tags walk the file exactly."""
    import torch
    import bench
    from fithic_amd import _capi
    from test_gpu_scale import _bench_rows, _engine_for
    cfg = dict(bench.CONFIGS["C5"])
    genome, cols_t, n, n_cis, n_trans = _bench_rows(cfg, [3_000_000, 2_000_000], torch.device("cuda", 0))
    assert n > 1_500_000 and n_trans > 10_000
    cols = [t[:n].cpu().numpy() for t in cols_t]
    L, U = 20_000, 800_000
    eng = _engine_for(genome, cfg["res"], L, U, 100, mode="All")
    eng.load_contacts_device([t.data_ptr() for t in cols_t], n)
    eng.run_pass(collect=False)
    v = eng.fetch(p=True, q=True, expcc=True, bias=True)
    dev, host = str(tmp_path / "dev.gz"), str(tmp_path / "host.gz")
    rows, nbytes = eng.ctx.write_significances_device(dev, genome.names, *cols)
    assert eng.ctx.write_significances_device(str(tmp_path / "rebuilt.gz"), genome.names) == (rows, nbytes)
    with open(str(tmp_path / "rebuilt.gz"), "rb") as f, open(dev, "rb") as g:
        assert f.read() == g.read()
    # the writer works through the rows in batches of members (bounded device memory): one member per batch = the same file
    os.environ["FHX_EMIT_BATCH_MEMBERS"] = "1"
    try:
        assert eng.ctx.write_significances_device(str(tmp_path / "batched.gz"), genome.names) == (rows, nbytes)
        assert eng.ctx.write_significances_device(str(tmp_path / "batched_cols.gz"), genome.names, *cols) == (rows, nbytes)
    finally:
        del os.environ["FHX_EMIT_BATCH_MEMBERS"]
    for name in ("batched.gz", "batched_cols.gz"):
        with open(str(tmp_path / name), "rb") as f, open(dev, "rb") as g:
            assert f.read() == g.read()
    from fithic_amd.engine import MODES
    n_host = _capi.host_write_significances(host, genome.names, *cols, v["p"], v["q"], v["b1"], v["b2"], v["expcc"], MODES["All"], L, U)
    d = np.abs(cols[1].astype(np.int64) - cols[3].astype(np.int64))
    n_emit = int(((cols[0] != cols[2]) | ((d >= L) & (d <= U))).sum())
    assert rows == n_host == n_emit and 0 < n_emit < n
    with gzip.open(dev, "rb") as f:
        got = f.read()
    with gzip.open(host, "rb") as f:
        want = f.read()
    assert got == want
    mem = _members(dev)
    assert 20 < len(mem) <= 1 + (n + 65535) // 65536
    # a sample of the first rows against Python's formatting itself
    first, _ = _python_text(genome.names, [c[:20000] for c in cols], {k: a[:20000] for k, a in v.items()}, "All", L, U)
    assert got.startswith(first)
    ratio = os.path.getsize(dev) / len(got)
    assert ratio < 0.45, ratio
    eng.close()


def test_rows_outside_the_device_formatter_are_refused_not_mangled(tmp_path):
    """A chromosome name longer than the kernel's 24 bytes: FHX_ERR_UNSUPPORTED (the caller falls back to the host writer);
    and the drop-in fit path does exactly that and still writes the right file."""
    from fithic_amd import _capi
    eng, kw, chroms, con = _engine_with_case("f2_all")
    eng.run_pass(collect=False)
    long_names = [n + "_with_a_very_long_suffix_0123456789" for n in chroms.names]
    with pytest.raises(_capi.FhxError) as e:
        eng.ctx.write_significances_device(str(tmp_path / "x.gz"), long_names, con.chr1, con.mid1, con.chr2, con.mid2, con.count)
    assert e.value.code == _capi.FHX_ERR_UNSUPPORTED
    with pytest.raises(_capi.FhxError):          # one identity row per loaded row
        eng.ctx.write_significances_device(str(tmp_path / "y.gz"), chroms.names, con.chr1[:-1], con.mid1[:-1], con.chr2[:-1], con.mid2[:-1],
                                           con.count[:-1])
    eng.close()


def test_members_without_an_emitted_row_are_left_out(tmp_path):
    """-x interOnly on cis + trans rows (cis rows first): the 65 536-row blocks that hold cis rows only give no member at
    all, the trans rows are exactly the reference's."""
    import torch
    import bench
    from test_gpu_scale import _bench_rows, _engine_for
    cfg = dict(bench.CONFIGS["C5"])
    genome, cols_t, n, n_cis, n_trans = _bench_rows(cfg, [3_000_000, 2_000_000], torch.device("cuda", 0))
    cols = [t[:n].cpu().numpy() for t in cols_t]
    eng = _engine_for(genome, cfg["res"], cfg["L"], cfg["U"], 100, mode="interOnly")
    eng.load_contacts_device([t.data_ptr() for t in cols_t], n)
    eng.run_pass(collect=False)
    v = eng.fetch(p=True, q=True, expcc=True, bias=True)
    path = str(tmp_path / "inter.gz")
    rows, _ = eng.ctx.write_significances_device(path, genome.names, *cols)
    want, n_emit = _python_text(genome.names, cols, v, "interOnly", cfg["L"], cfg["U"])
    with gzip.open(path, "rb") as f:
        assert f.read() == want
    assert rows == n_emit == n_trans == int((cols[0] != cols[2]).sum()) > 0
    assert len(_members(path)) <= 1 + (n_trans + 65535) // 65536 + 1 < (n + 65535) // 65536
    eng.close()


def test_device_number_formatting_equals_python_on_adversarial_doubles():
    """fmt_e6 / fmt_f6 as the GPU compiles them (fhx_debug_format) against Python's % on 1.2e6 doubles: random bit patterns, p-value-
    like and ExpCC-like magnitudes, exact decimal ties ((2k+1)/2^j and their 1e-6 multiples), neighbours of every power of ten
    from 1e-330 to 1e25, denormals, zeros, infinities, NaN.  The CPU build of the same header is checked against libc in
    tests/test_fmt.py; this is the device build."""
    from fithic_amd import _capi
    rng = np.random.default_rng(99)
    vals = [rng.integers(0, 1 << 64, 400_000, dtype=np.uint64).view(np.float64),
            rng.random(200_000) ** (1.0 + 40.0 * rng.random(200_000)),
            rng.random(100_000) * 10.0 ** rng.integers(-3, 9, 100_000),
            np.exp(rng.normal(0, 0.4, 50_000))]
    ties = []
    for j in range(1, 40):
        k = 2.0 * np.arange(2000) + 1.0
        ties += [np.ldexp(k, -j), np.ldexp(k, -j) * 1e-6, k * 0.5e-6, np.arange(2000) + 0.5]
    vals += ties
    p10 = []
    for e in range(-330, 26):
        v = np.float64(10.0) ** e
        row = [v]
        for _ in range(3):
            row += [np.nextafter(row[-1], np.inf)]
        lo = v
        for _ in range(3):
            lo = np.nextafter(lo, 0.0)
            row.append(lo)
        row = np.array(row)
        p10 += [row, row * 9.9999995, row * 1.0000005]
    vals += p10
    vals.append(np.array([0.0, -0.0, 1.0, -1.0, 0.5, 9.9999995, 0.9999995, 5e-7, 1.5e-6, 2.5e-6, 123456.5, 1e15, 1e16, 9007199254740993.0, 4.9e-324,
                          2.2250738585072014e-308, 1.7976931348623157e308, 1e22, 1e23, 1.8446744073709552e19, 9.2233720368547758e18, np.nan, np.inf,
                          -np.inf, 999999.5, 9999999.5, 0.1 + 0.2, 1.0 / 3.0]))
    v = np.concatenate([np.asarray(a, np.float64).ravel() for a in vals])
    v = np.concatenate([v, -v])
    ctx = _capi.Context(0)
    checked = 0
    for kind, spec, limit in ((0, "%e", 2.0 ** 64), (1, "%f", 2.0 ** 63)):
        got = ctx.debug_format(v, kind)
        vl = v.tolist()
        for i, g in enumerate(got):
            x = vl[i]
            if g is None:
                assert abs(x) >= limit, (spec, x)           # only what the header says it does not cover
                continue
            want = (spec % x).encode()
            if want in (b"-nan",):
                want = b"nan"
            assert g == want, (spec, repr(x), g, want)
            checked += 1
    ctx.close()
    assert checked > 2_000_000
