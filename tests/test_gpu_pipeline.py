"""The FitHiC2 protocol end to end on the GPU - HiCKRy bias -> Fit-Hi-C -> merge of nearby significant contacts - through the
three command lines, each stage's file compared with the oracle chain run on the same inputs (bundled hESC chr1, 40 kb)."""
import gzip
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
DATA = os.path.join(HERE, "golden", "data")


def test_bias_then_significance_then_merge(tmp_path, monkeypatch, capsys):
    from fithic_amd import cli, combine, hickry
    from oracle import combine_oracle as co
    from oracle import fithic_oracle as fo
    from oracle import hickry_oracle as ho
    con, frg = os.path.join(DATA, "hESC_chr1_w40000.contacts.gz"), os.path.join(DATA, "hESC_chr1_w40000.frags.gz")
    # 1. bias file
    bias_path = str(tmp_path / "kr_bias.gz")
    monkeypatch.setattr("sys.argv", ["HiCKRy.py", "-i", con, "-f", frg, "-o", bias_path, "-x", "0.12"])
    hickry.main()
    o = ho.run(con, frg, 0.12)
    want_bias = "".join("%s\t%s\t%s\n" % (c, m, repr(float(v))) for (c, m), v in zip(o["revFrag"], o["bias"]))
    with gzip.open(bias_path, "rt") as f:
        assert f.read() == want_bias
    # 2. significances with that bias, two passes
    out = tmp_path / "fit"
    cli.main(["-i", con, "-f", frg, "-t", bias_path, "-o", str(out), "-l", "P", "-r", "40000", "-L", "50000", "-U", "5000000", "-b", "50",
              "-p", "2"])
    ref = fo.run(con, frg, bias_path, 40000, n_bins=50, passes=2, mode="intraOnly", L=50000, U=5000000, keep_text=True)
    sig = str(out / "P.spline_pass2.res40000.significances.txt.gz")
    with gzip.open(sig, "rt") as f:
        got_sig = f.read()
    assert got_sig == ref[1].sig_txt
    # 3. merge the strongest contacts
    lines = got_sig.splitlines(True)
    strong = str(tmp_path / "strong.txt.gz")
    with gzip.open(strong, "wt") as f:
        f.write(lines[0] + "".join(ln for ln in lines[1:] if float(ln.split()[6]) < 1e-12))
    merged = str(tmp_path / "merged.gz")
    monkeypatch.setattr("sys.argv", ["CombineNearbyInteraction.py", "-i", strong, "-o", merged, "-r", "40000", "-H", "1"])
    combine.main()
    with gzip.open(merged, "rt") as f:
        got = f.read().split("\n")
    want = [co.HEADER] + co.combine_lines(strong, 40000)
    assert got == want and len(got) > 500
    capsys.readouterr()
