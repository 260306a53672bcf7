"""GPU parity of the nearby-contact merging (fithic_amd.combine / fhx_cni_*) through the C ABI: the output
file equals the real reference's byte for byte (tests/golden/c*_combine_*.out.gz) and the oracle's on larger inputs."""
import gzip
import os

import numpy as np
import pytest

from test_combine_oracle import COMBINE_CASES, DATA, combine_case

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name", COMBINE_CASES)
def test_merged_file_equals_the_reference(name, tmp_path, monkeypatch, capsys):
    from fithic_amd import combine
    meta, want, res, kw = combine_case(name)
    out = str(tmp_path / "merged.gz")
    monkeypatch.setattr("sys.argv", ["CombineNearbyInteraction.py", "-i", os.path.join(DATA, meta["input"]), "-o", out] + meta["argv"])
    combine.main()
    with gzip.open(out, "rt") as f:
        got = f.read().split("\n")
    assert got == want


def _random_table(path, rng, n_rows, n_chr, res, span):
    rows = ["chr1\tfragmentMid1\tchr2\tfragmentMid2\tcontactCount\tp-value\tq-value\tbias1\tbias2\tExpCC\n"]
    chroms = ["chr%d" % (c + 1) for c in range(n_chr)]
    c = rng.integers(0, n_chr, n_rows)
    b1 = rng.integers(0, span, n_rows)
    b2 = np.minimum(b1 + rng.geometric(0.05, n_rows) - 1, span - 1)
    q = np.round(10 ** rng.uniform(-9, -2, n_rows), 12)
    q[rng.random(n_rows) < 0.2] = 1e-5                                # ties
    cc = rng.integers(1, 300, n_rows)
    for i in range(n_rows):
        rows.append("%s\t%d\t%s\t%d\t%d\t%e\t%e\t1.0\t1.0\t1.0\n" % (chroms[c[i]], b1[i] * res + res // 2, chroms[c[i]],
                                                                  b2[i] * res + res // 2, cc[i], q[i] / 10, q[i]))
    with gzip.open(path, "wt") as f:
        f.write("".join(rows))


@pytest.mark.parametrize("kw", [dict(), dict(conn=4, neigh=1), dict(pct=40), dict(pct=70, order=1), dict(order=1, neigh=4)])
def test_large_random_tables_equal_the_oracle(kw, tmp_path):
    """2e5 rows on 5 chromosomes (dense enough for components of thousands of cells): same lines as the oracle."""
    from fithic_amd import combine
    from oracle import combine_oracle as co
    rng = np.random.default_rng(11)
    path = str(tmp_path / "sig.gz")
    res = 5000
    _random_table(path, rng, 200_000, 5, res, 1500)
    want = co.combine_lines(path, res, **kw)
    names, rec, info = combine.combine_records(combine.read_significances(path, 1), res, **kw)
    got = combine.format_lines(names, rec, res)
    assert info.nodes > 100_000 and info.largest_component > 500
    assert got == want


_CN_LO, _CN_HI = (int(v) for v in os.environ.get("FHX_FUZZ_SEEDS", "0:6").split(":"))      # campaigns: FHX_FUZZ_SEEDS="lo:hi"


@pytest.mark.parametrize("seed", range(_CN_LO, _CN_HI))
def test_random_tables_and_parameters_equal_the_oracle(seed, tmp_path):
    """Random table sizes, densities, resolutions and every parameter of the merge: the same lines as the oracle."""
    from fithic_amd import combine
    from oracle import combine_oracle as co
    rng = np.random.default_rng(5000 + seed)
    path = str(tmp_path / "sig.gz")
    res = int(rng.choice([1000, 5000, 40000]))
    _random_table(path, rng, int(rng.integers(50, 30000)), int(rng.integers(1, 6)), res, int(rng.integers(40, 1200)))
    kw = dict(conn=int(rng.choice([4, 8])), pct=int(rng.choice([10, 30, 50, 70, 100])), neigh=int(rng.integers(1, 5)), order=int(rng.integers(0, 2)))
    want = co.combine_lines(path, res, **kw)
    names, rec, info = combine.combine_records(combine.read_significances(path, 1), res, **kw)
    assert combine.format_lines(names, rec, res) == want, kw


def test_off_lattice_rows_and_mode_zero_are_refused():
    from fithic_amd import _capi
    cn = _capi.CniContext(0)
    with pytest.raises(_capi.FhxError):
        cn.load([0, 0], [40000, 60001], [80000, 120000], [5, 6], [1e-3, 1e-4], [1e-2, 1e-3], 40000)
    cn.load([0, 0], [40000, 80000], [80000, 120000], [5, 6], [1e-3, 1e-4], [1e-2, 1e-3], 40000)
    with pytest.raises(_capi.FhxError):
        cn.run(8, 0, 2, 0)
    rec, info = cn.run(8, 100, 2, 0)
    assert info.nodes == 2 and info.components == 1 and len(rec) == 1 and rec[0]["q"] == 1e-3
    cn.close()


def test_half_bin_offsets_keep_float_bins(tmp_path):
    """mids on multiples of the resolution give bins k + 0.5: components work, the box-cell count is 0 like the reference's
    integer lookups in a dict of float keys."""
    from fithic_amd import combine
    from oracle import combine_oracle as co
    path = str(tmp_path / "sig.txt")
    rows = ["h\n"] + ["chrA\t%d\tchrA\t%d\t%d\t1e-5\t%e\tx\n" % (a * 1000, b * 1000, 7 + a, 1e-4 * (1 + (a * b) % 5))
                      for a, b in [(3, 9), (4, 9), (4, 10), (20, 30), (21, 31), (3, 9)]]
    with open(path, "w") as f:
        f.write("".join(rows))
    want = co.combine_lines(path, 1000)
    names, rec, info = combine.combine_records(combine.read_significances(path, 1), 1000)
    assert combine.format_lines(names, rec, 1000) == want and len(want) == 2 and want[0].endswith("\t0.0")


def test_repeated_runs_are_identical():
    """The union-find is lock-free: 12 runs over the same 3e5 cells must give the same records every time."""
    from fithic_amd import _capi
    rng = np.random.default_rng(21)
    n = 400_000
    b1 = rng.integers(0, 3000, n)
    b2 = np.minimum(b1 + rng.geometric(0.03, n) - 1, 2999)
    q = np.round(10 ** rng.uniform(-9, -2, n), 10)
    cn = _capi.CniContext(0)
    cn.load(rng.integers(0, 3, n), (b1 + 1) * 5000, (b2 + 1) * 5000, rng.integers(1, 99, n), q / 7, q, 5000)
    first, info0 = cn.run(4, 100, 1, 0)
    assert info0.largest_component > 200
    for _ in range(11):
        rec, info = cn.run(4, 100, 1, 0)
        assert info.as_dict() == info0.as_dict() and rec.tobytes() == first.tobytes()
    cn.close()
