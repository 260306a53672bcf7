"""Differential fuzzing of the whole pass on the GPU against the oracle (itself bit-identical to the real reference on all
golden cases): random small genomes, resolutions, bounds, bin counts, modes, bias tables with missing / out-of-range
entries, unmappable fragments, 1-3 passes, fixed-size and -r 0.  Integer results equal, p/q within 1e-10; where the
reference would exit or raise, both sides must refuse."""
import gzip
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
TOL = 1e-10


def _write(path, text):
    with gzip.open(path, "wt") as f:
        f.write(text)


def _make_case(rng, d, nonfixed, count_scale=1, offgrid=False):
    """count_scale multiplies every contact count: sums of 1e6..1e10 put the continued fractions into the regimes where K2 seeds
    its divisions from the previous reciprocal (a >= 1e6, a >= 2e8) and counts beyond the bucket table of the heavy class.
    offgrid: irregular midpoints (as for -r 0) but run with -r RES: the reference takes |mid1 - mid2| of whatever the files hold
    and still enumerates possible pairs at multiples of the resolution (the engine slots such input)."""
    res = int(rng.choice([1000, 5000, 40000]))
    n_chr = int(rng.integers(1, 5))
    names = ["chr%s" % s for s in rng.permutation(["1", "2", "10", "X", "M"])[:n_chr]]
    loci = {}
    frag_lines, bias_lines = [], []
    for ch in names:
        n = int(rng.integers(20, 110))
        if nonfixed or offgrid:
            mids = np.cumsum(rng.integers(res // 4 + 1, 3 * res, n))
        else:
            mids = np.arange(n) * res + res // 2
        loci[ch] = mids
        for k, m in enumerate(mids):
            hits = 0 if rng.random() < 0.04 else int(rng.integers(1, 4))
            frag_lines.append("%s\t0\t%d\t%d\t1\n" % (ch, m, hits))
            if rng.random() > 0.05:
                b = float(np.exp(rng.normal(0, 0.4)))
                if rng.random() < 0.03:
                    b = float(rng.choice([0.1, 3.5]))
                bias_lines.append("%s\t%d\t%.6f\n" % (ch, m, b))
    rows = []
    for ch in names:
        m = loci[ch]
        for i in range(len(m)):
            for j in range(i + (0 if rng.random() < 0.02 else 1), len(m)):
                dist = abs(int(m[j]) - int(m[i])) / res
                lam = 25.0 / (1.0 + dist) ** 1.1
                c = int(rng.poisson(lam * rng.lognormal(0, 0.5)))
                if c >= 1 and rng.random() < 0.7:
                    rows.append("%s\t%d\t%s\t%d\t%d\n" % (ch, m[i], ch, m[j], c * count_scale))
    if n_chr > 1:
        for _ in range(int(rng.integers(0, 300))):
            a, b = rng.choice(n_chr, 2, replace=False)
            rows.append("%s\t%d\t%s\t%d\t%d\n" % (names[a], rng.choice(loci[names[a]]), names[b], rng.choice(loci[names[b]]),
                                                 (1 + int(rng.poisson(0.8))) * count_scale))
    order = rng.permutation(len(rows))
    paths = dict(contacts=os.path.join(d, "c.gz"), frags=os.path.join(d, "f.gz"), bias=os.path.join(d, "b.gz"))
    _write(paths["contacts"], "".join(rows[i] for i in order))
    _write(paths["frags"], "".join(frag_lines))
    _write(paths["bias"], "".join(bias_lines))
    span = max(int(v.max()) for v in loci.values())
    kw = dict(resolution=0 if nonfixed else res, n_bins=int(rng.integers(4, 40)), passes=int(rng.integers(1, 4)),
              mode=str(rng.choice(["intraOnly", "All", "interOnly"] if n_chr > 1 else ["intraOnly", "All"])),
              mapp_thres=int(rng.choice([1, 1, 2])), tL=0.5, tU=2.0, L=0, U=float("inf"))
    if rng.random() < 0.6:
        kw["L"] = int(rng.integers(0, 4)) * res
        kw["U"] = int(rng.integers(8, 60)) * res if rng.random() < 0.8 else float("inf")
    kw["bias_path"] = paths["bias"] if rng.random() < 0.7 else None
    return paths, kw, len(rows), span


# FHX_FUZZ_SEEDS="lo:hi" runs another range of seeds (campaigns; the default 36 run in every GPU test session)
_LO, _HI = (int(v) for v in os.environ.get("FHX_FUZZ_SEEDS", "0:36").split(":"))


_FOUND = [1413, 2006, 2772, 4554,      # campaign finds, kept: a NEGATIVE number of tests (possible pairs < 0) reaches the BH step
          100024, 110087, 110772,         # counts scaled to 1e4..1e6: libm amplification / the oracle's fitpack squares
          *range(200000, 200010)]          # -r RES on loci that are not on one grid


@pytest.mark.parametrize("seed", list(range(_LO, _HI)) + ([] if "FHX_FUZZ_SEEDS" in os.environ else _FOUND))
def test_random_small_runs_match_the_oracle(seed, tmp_path):
    from fithic_amd import _capi, tables
    from fithic_amd.engine import Engine
    from oracle import fithic_oracle as fo
    rng = np.random.default_rng(9000 + seed)
    nonfixed = seed % 4 == 3
    # seeds from 100 000 on: counts scaled up (the cases of the lower seeds stay what they were)
    scale = 1 if seed < 100000 else int(np.random.default_rng(seed).choice([1, 37, 2500, 400000]))
    offgrid = seed >= 200000                       # seeds from 200 000 on: -r RES on loci that are not on one grid
    if offgrid:
        nonfixed, scale = False, 1
    paths, kw, n_rows, span = _make_case(rng, str(tmp_path), nonfixed, scale, offgrid)
    try:
        ref = fo.run(paths["contacts"], paths["frags"], kw["bias_path"], kw["resolution"], kw["n_bins"], kw["passes"], kw["mode"],
                     kw["L"], kw["U"], kw["mapp_thres"], kw["tL"], kw["tU"])
        ref_error = None
    except (SystemExit, ZeroDivisionError, TypeError, ValueError, IndexError, KeyError) as e:      # what the reference raises
        ref, ref_error = None, e
    chroms = tables.ChromIndex()
    con = tables.read_contacts(paths["contacts"], chroms)
    eng = Engine(0)
    try:
        eng.configure(kw["resolution"], kw["L"], kw["U"], kw["n_bins"], kw["mapp_thres"], kw["mode"], kw["tL"], kw["tU"])
        eng.load_fragments(*tables.read_fragments(paths["frags"], chroms), chroms.sort_rank())
        if kw["bias_path"]:
            eng.load_bias(*tables.read_bias(kw["bias_path"], chroms))
        eng.load_contacts(con.chr1, con.mid1, con.chr2, con.mid2, con.count)
        if ref_error is not None:
            with pytest.raises(_capi.FhxError):
                for _ in range(kw["passes"]):
                    eng.run_pass()
                    eng.next_pass()
            return
        # Cephes forms p = exp(a log x + b log(1 - x) - lbeta + ...) with a = count: the last bit of the libm's log (glibc on the
        # reference's side, ocml on the device: both < 1 ulp, not identical) is multiplied by the count.  At Hi-C counts (<= 1e4)
        # that stays below 1e-11; the scaled-up cases of the campaign seeds reach counts of 1e6 and ~1e-9.
        tol = max(TOL, 1e-15 * float(np.max(np.abs(con.count)))) if len(con) else TOL
        for pi, r in enumerate(ref):
            out = eng.run_pass()
            v = eng.fetch(p=True, q=True, expcc=True, bias=True)
            # the significances file as the GPU formats and deflates it: the rows the oracle emits, the characters Python's % gives
            # for the engine's own values (NaN, 0, 1, denormals, 1e-300 ... all occur here)
            sig = os.path.join(str(tmp_path), "sig%d.gz" % pi)
            try:
                n_written, _ = eng.ctx.write_significances_device(sig, chroms.names)
            except _capi.FhxError as e:                       # a row of 192 bytes or more: the host writer takes over, as in fit_Spline
                assert e.code == _capi.FHX_ERR_UNSUPPORTED
                from fithic_amd.engine import MODES
                n_written = _capi.host_write_significances(sig, chroms.names, con.chr1, con.mid1, con.chr2, con.mid2, con.count, v["p"], v["q"],
                                                           v["b1"], v["b2"], v["expcc"], MODES[kw["mode"]], kw["L"], kw["U"])
            emit = np.flatnonzero(r.emit)
            assert n_written == len(emit)
            names = chroms.names
            want = ["chr1\tfragmentMid1\tchr2\tfragmentMid2\tcontactCount\tp-value\tq-value\tbias1\tbias2\tExpCC\n"]
            for i in emit.tolist():
                want.append("%s\t%d\t%s\t%d\t%d\t%e\t%e\t%e\t%e\t%f\n" % (names[con.chr1[i]], con.mid1[i], names[con.chr2[i]], con.mid2[i], con.count[i],
                                                                         v["p"][i], v["q"][i], v["b1"][i], v["b2"][i], v["expcc"][i]))
            with gzip.open(sig, "rb") as f:
                assert f.read() == "".join(want).encode(), ("significances text", pi)
            assert [out.stats["inter_count"], out.stats["inter_sum"], out.stats["intra_all_sum"], out.stats["in_range_sum"]] == list(r.sums)
            assert out.info["bh_total_tests"] == r.N
            # the host fit, bit for bit: bins, their means, the spline table after the antitonic regression, BH's N above
            A = out.arrays
            for k, mine in (("lb", "bin_lb"), ("ub", "bin_ub"), ("s1", "bin_poss"), ("s2", "bin_sumcc"), ("s7", "bin_poss7")):
                assert np.array_equal(A[mine], np.array([b[k] for b in r.bins])), (k, pi)
            bits = lambda a, b: len(a) == len(b) and np.array_equal(np.asarray(a, np.float64).view(np.int64), np.asarray(b, np.float64).view(np.int64))
            assert bits(A["x"], r.x) and bits(A["y"], r.y), pi
            if r.newSplineY is not None:
                assert np.array_equal(A["table_x"], r.splineX) and bits(A["table_y"], r.newSplineY) and bits(A["knots"], r.t), pi
            assert bits(v["expcc"], r.expcc) and bits(v["b1"], r.b1) and bits(v["b2"], r.b2), pi
            for key, want in (("p", r.p), ("q", r.q)):
                got = v[key]
                assert np.array_equal(np.isnan(got), np.isnan(want)), (key, pi)
                ok = ~np.isnan(want)
                assert not ok.any() or np.max(np.abs(got[ok] - want[ok])) <= tol, (key, pi)
            assert eng.next_pass() == r.n_outlier_lines_total
    finally:
        eng.close()


@pytest.mark.parametrize("seed", [100003, 100011, 100021, 100034] if "FHX_FUZZ_SEEDS" not in os.environ else list(range(_LO, _HI)))
def test_table_fed_heavy_class_equals_the_per_lane_kernel_on_random_cases(seed, tmp_path, monkeypatch):
    """k2h_heavy (count-homogeneous waves, iteration constants from a table, uniform renormalisation, nontemporal stores) against
    round 1's per-lane kernel (FHX_K2_LEGACY=1) on the random cases of this file, counts scaled so that the totals range from
    1e3 to 1e10: p and q bit for bit, every pass."""
    from fithic_amd import tables
    from fithic_amd.engine import Engine
    rng = np.random.default_rng(9000 + seed)
    scale = 1 if seed < 100000 else int(np.random.default_rng(seed).choice([1, 37, 2500, 400000]))
    paths, kw, n_rows, _ = _make_case(rng, str(tmp_path), seed % 4 == 3, scale)
    chroms = tables.ChromIndex()
    con = tables.read_contacts(paths["contacts"], chroms)
    frags = tables.read_fragments(paths["frags"], chroms)
    bias = tables.read_bias(kw["bias_path"], chroms) if kw["bias_path"] else None
    got = {}
    for tag in ("table", "per-lane"):
        if tag == "per-lane":
            monkeypatch.setenv("FHX_K2_LEGACY", "1")
        else:
            monkeypatch.delenv("FHX_K2_LEGACY", raising=False)
        eng = Engine(0)
        res_passes = []
        try:
            eng.configure(kw["resolution"], kw["L"], kw["U"], kw["n_bins"], kw["mapp_thres"], kw["mode"], kw["tL"], kw["tU"])
            eng.load_fragments(*frags, chroms.sort_rank())
            if bias:
                eng.load_bias(*bias)
            eng.load_contacts(con.chr1, con.mid1, con.chr2, con.mid2, con.count)
            for _ in range(kw["passes"]):
                eng.run_pass(collect=False)
                res_passes.append(eng.fetch())
                eng.next_pass()
        except Exception as e:                                 # cases the reference refuses: both modes must refuse alike
            res_passes.append(type(e).__name__)
        finally:
            eng.close()
        got[tag] = res_passes
    assert len(got["table"]) == len(got["per-lane"])
    for a, b in zip(got["table"], got["per-lane"]):
        if isinstance(a, str) or isinstance(b, str):
            assert a == b
            continue
        for key in ("p", "q"):
            same = (a[key].view(np.int64) == b[key].view(np.int64)) | (np.isnan(a[key]) & np.isnan(b[key]))
            assert same.all(), (key, int((~same).sum()))


@pytest.mark.parametrize("seed", [3, 8, 14, 21, 27, 33] if "FHX_FUZZ_SEEDS" not in os.environ else list(range(_LO, _HI)))
def test_command_line_on_random_cases_writes_the_oracle_files(seed, tmp_path, capsys):
    _cli_case(seed, tmp_path, capsys, gpus=1)


@pytest.mark.parametrize("seed", [4, 9] if "FHX_FUZZ_SEEDS" not in os.environ else list(range(_LO, _HI)))
def test_sharded_command_line_on_random_cases(seed, tmp_path, capsys, monkeypatch):
    """The same with `--gpus 2` / `--gpus 3` (rows sharded by chromosome over worker processes, collectives over pipes on this
    one-GPU box; needs -r > 0)."""
    if seed % 4 == 3:
        pytest.skip("-r 0 runs on one GPU")
    monkeypatch.setenv("FHX_CLI_TRANSPORT", "pipes")
    gpus = 2 + seed % 2
    monkeypatch.setenv("FHX_CLI_DEVICES", ",".join(["0"] * gpus))
    if seed % 3 != 0:
        # the ranks inflate the (one-stream) contacts file together wherever the stream has a block start for every part - rows that
        # straddle parts, parts that end on a newline, a last row without one - and fall back to the whole text where it has not
        monkeypatch.setenv("FHX_PGUNZIP_MIN", "0")
        monkeypatch.setenv("FHX_PGUNZIP_CHUNK", str(1024 << (seed % 5)))
    _cli_case(seed, tmp_path, capsys, gpus=gpus)


def _cli_case(seed, tmp_path, capsys, gpus):
    """`python -m fithic_amd` end to end on the random cases: the .fithic_passN text equals the oracle's byte for byte; every row
    of the significances file has the oracle's identity columns, biases and ExpCC as text and p / q within tolerance (a p that
    differs in its 14th digit may print a different 7th once in ~1e7 values, so p and q are compared as numbers)."""
    from fithic_amd import cli
    from oracle import fithic_oracle as fo
    rng = np.random.default_rng(9000 + seed)
    nonfixed = seed % 4 == 3
    paths, kw, n_rows, _ = _make_case(rng, str(tmp_path), nonfixed)
    try:
        ref = fo.run(paths["contacts"], paths["frags"], kw["bias_path"], kw["resolution"], kw["n_bins"], kw["passes"], kw["mode"],
                     kw["L"], kw["U"], kw["mapp_thres"], kw["tL"], kw["tU"], keep_text=True)
    except (SystemExit, ZeroDivisionError, TypeError, ValueError, IndexError, KeyError):
        ref = None
    out = os.path.join(str(tmp_path), "out")
    argv = ["-i", paths["contacts"], "-f", paths["frags"], "-o", out, "-l", "Z", "-r", str(kw["resolution"]), "-b", str(kw["n_bins"]),
            "-p", str(kw["passes"]), "-x", kw["mode"], "-m", str(kw["mapp_thres"])]
    if kw["L"]:
        argv += ["-L", str(kw["L"])]
    if kw["U"] != float("inf"):
        argv += ["-U", str(kw["U"])]
    if kw["bias_path"]:
        argv += ["-t", kw["bias_path"]]
    if gpus > 1:
        argv += ["--gpus", str(gpus)]
    if ref is None:
        with pytest.raises((SystemExit, Exception)):
            cli.main(argv)
        return
    cli.main(argv)
    capsys.readouterr()
    tag = (".res%d" % kw["resolution"]) if kw["resolution"] else ""
    for pi, r in enumerate(ref, 1):
        with open(os.path.join(out, "Z.fithic_pass%d%s.txt" % (pi, tag))) as f:
            assert f.read() == r.pass_txt, pi
        with gzip.open(os.path.join(out, "Z.spline_pass%d%s.significances.txt.gz" % (pi, tag)), "rt") as f:
            mine = f.read().splitlines()
        want = r.sig_txt.splitlines()
        assert len(mine) == len(want) and mine[0] == want[0], pi
        for a, b in zip(mine[1:], want[1:]):
            fa, fb = a.split("\t"), b.split("\t")
            assert fa[:5] == fb[:5] and fa[7:] == fb[7:], (pi, a, b)
            for k in (5, 6):
                if fa[k] != fb[k]:
                    va, vb = float(fa[k]), float(fb[k])
                    assert (np.isnan(va) and np.isnan(vb)) or abs(va - vb) <= max(TOL, 2e-6 * abs(vb)), (pi, a, b)
