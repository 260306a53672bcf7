"""CPU: the Knight-Ruiz oracle (oracle/hickry_oracle.py + kr_oracle.c) pinned against the real reference's results
(tests/golden/k*_kr_*.npz from fithic/utils/HiCKRy.py via make_golden.py f9), plus the host-side helpers of
fithic_amd.hickry that need no GPU."""
import json
import os

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
GOLDEN = os.path.join(HERE, "golden")
DATA = os.path.join(GOLDEN, "data")
KR_CASES = ["k1_kr_hESC", "k1_kr_hESC_default", "k2_kr_pfal", "k3_kr_imr90", "k4_kr_irregular", "k5_kr_combine"]


@pytest.mark.parametrize("name", KR_CASES)
def test_oracle_reproduces_the_reference(name):
    from oracle import hickry_oracle as ho
    with open(os.path.join(GOLDEN, name + ".json")) as f:
        meta = json.load(f)
    g = np.load(os.path.join(GOLDEN, name + ".npz"))
    o = ho.run(os.path.join(DATA, meta["contacts"]), os.path.join(DATA, meta["frags"]), meta["perc"])
    assert o["A"].n == meta["n"] and o["A"].nnz == meta["nnz"] and o["R"].nnz == meta["nnz_reduced"]
    assert float(o["A"].data.sum()) == float(g["data_checksum"][0]) and float(o["R"].data.sum()) == float(g["data_checksum"][1])
    assert np.allclose(o["row_sums"], g["row_sums"], rtol=1e-13, atol=0)
    assert o["removed"].tolist() == g["removed"].tolist()
    assert np.array_equal(o["R"].indptr, g["indptr_reduced"])
    assert (o["outer"], o["inner"]) == (meta["outer"], meta["inner"])
    assert np.max(np.abs(o["x"] - g["x"]) / np.abs(g["x"])) <= 1e-12
    ok = g["bias"] != -1.0
    assert np.array_equal(o["bias"] == -1.0, ~ok)
    assert np.max(np.abs(o["bias"][ok] - g["bias"][ok]) / np.abs(g["bias"][ok])) <= 1e-12


def test_oracle_summation_orders():
    """The fixed orders the kernels use: per row 64 partials (chunks of 256 cells, partial l = cells 4l..4l+3 of each chunk)
    + tree; 1024-element tiles for dot products."""
    from oracle import hickry_oracle as ho
    rng = np.random.default_rng(3)
    n = 3000
    a, b = rng.normal(size=n), rng.normal(size=n)
    # tile order by hand
    total = None
    for t0 in range(0, n, 1024):
        th = np.zeros(256)
        for t in range(256):
            acc = 0.0
            for k in range(4):
                i = t0 + t + 256 * k
                if i < n:
                    acc = acc + a[i] * b[i]
            th[t] = acc
        tile = None
        for w in range(4):
            v = th[64 * w:64 * w + 64].copy()
            s = 32
            while s >= 1:
                v[:s] = v[:s] + v[s:2 * s]
                s //= 2
            tile = v[0] if tile is None else tile + v[0]
        total = tile if total is None else total + tile
    assert ho.dot(a, b) == total
    # a row of 700 cells (two full chunks and a partial one)
    A = ho.Csr(1, np.array([0, 700], np.int64), np.arange(700, dtype=np.int32), rng.normal(size=700))
    x = rng.normal(size=700)
    lane = np.zeros(64)
    for j in range(700):
        lane[(j % 256) // 4] = lane[(j % 256) // 4] + A.data[j] * x[j]
    s = 32
    while s >= 1:
        lane[:s] = lane[:s] + lane[s:2 * s]
        s //= 2
    assert A.dot(x)[0] == lane[0]


def test_host_helpers_keep_the_reference_shapes(tmp_path, capsys):
    import gzip
    from fithic_amd import hickry
    x = np.array([[0.5], [2.0], [4.0]])
    b = hickry.computeBiasVector(x)
    assert b.shape == (3, 1) and np.allclose(b.ravel(), np.array([2.0, 0.5, 0.25]) / (2.75 / 3))
    full = hickry.addZeroBiases([0, 3], b)
    assert full.shape == (5, 1) and full.ravel().tolist()[0] == -1.0 and full.ravel().tolist()[3] == -1.0
    hickry.checkBias(np.array([0.1, 0.2, 0.3]))
    assert "WARNING... Bias vector has a mean outside" in capsys.readouterr().out
    hickry.checkBias(np.array([1.0, 1.1, 0.9]))
    assert capsys.readouterr().out == ""
    out = str(tmp_path / "b.gz")
    hickry.outputBias(np.array([[-1.0], [0.1234567890123], [9.905585280664274e-11]]), [("chr1", 5), ("chr1", 15), ("chrX", 7)], out)
    with gzip.open(out, "rt") as f:
        assert f.read() == "chr1\t5\t-1.0\nchr1\t15\t0.1234567890123\nchrX\t7\t9.905585280664274e-11\n"
