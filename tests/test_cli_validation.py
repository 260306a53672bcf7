"""The drop-in command line refuses what the reference refuses, with its messages and exit status 2
(fithic/fithic.py:136-263).  Everything here exits before the engine is touched: no GPU needed."""
import gzip
import os

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
DATA = os.path.join(HERE, "golden", "data")
CON = os.path.join(DATA, "quirk.contacts.gz")
FRG = os.path.join(DATA, "quirk.frags.gz")


def _run(argv, capsys):
    from fithic_amd import cli
    with pytest.raises(SystemExit) as e:
        cli.main(argv)
    return e.value.code, capsys.readouterr().out


def test_missing_and_ungzipped_inputs(tmp_path, capsys):
    code, out = _run(["-i", CON, "-f", str(tmp_path / "nope.gz"), "-o", str(tmp_path), "-r", "10000"], capsys)
    assert code == 2 and "Fragment file not found" in out
    plain = tmp_path / "frags.txt"
    plain.write_text("chr1\t0\t5000\t1\t1\n")
    code, out = _run(["-i", CON, "-f", str(plain), "-o", str(tmp_path), "-r", "10000"], capsys)
    assert code == 2 and "Fragments file is not gzipped. Exiting now..." in out
    code, out = _run(["-i", str(tmp_path / "nope.gz"), "-f", FRG, "-o", str(tmp_path), "-r", "10000"], capsys)
    assert code == 2 and "Interaction file not found" in out
    code, out = _run(["-i", CON, "-f", FRG, "-o", str(tmp_path), "-r", "10000", "-t", str(tmp_path / "nobias.gz")], capsys)
    assert code == 2 and "Bias file not found" in out


def test_invalid_options(tmp_path, capsys):
    code, out = _run(["-i", CON, "-f", FRG, "-o", str(tmp_path), "-r", "-5"], capsys)
    assert code == 2 and "INVALID RESOLUTION ARGUMENT DETECTED" in out and "User-given resolution: -5" in out
    code, out = _run(["-i", CON, "-f", FRG, "-o", str(tmp_path), "-r", "10000", "-x", "everything"], capsys)
    assert code == 2 and "Invalid Option. Only options are 'All', 'interOnly', or 'intraOnly'" in out
    code, out = _run(["-i", CON, "-f", FRG, "-o", str(tmp_path), "-r", "10000", "-tL", "3", "-tU", "2"], capsys)
    assert code == 2 and "Bias lower bound is greater than bias upper bound" in out
    # argparse itself: a required flag is missing
    code, _ = _run(["-i", CON, "-f", FRG, "-r", "10000"], capsys)
    assert code == 2


def test_banner_reports_the_zero_means_unset_rule(tmp_path, capsys):
    """-p 0 -b 0 -m 0 -U 0 -L 0 print the defaults (fithic.py:194-221); stop the run at the resolution check that follows."""
    code, out = _run(["-i", CON, "-f", FRG, "-o", str(tmp_path / "new_dir"), "-r", "-1", "-p", "0", "-b", "0"], capsys)
    assert code == 2 and "Output path created" in out and os.path.isdir(str(tmp_path / "new_dir"))
    assert out.index("Reading fragments file from") < out.index("Reading interactions file from") < out.index("INVALID RESOLUTION")


def test_console_script_is_declared_like_the_reference():
    """ay-lab/fithic installs `fithic = fithic.fithic:main` (setup.py:14-16); pyproject.toml declares the same command."""
    import importlib
    import os
    import subprocess
    import sys
    import tomli
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with open(os.path.join(root, "pyproject.toml"), "rb") as f:
        meta = tomli.load(f)
    mod, fn = meta["project"]["scripts"]["fithic"].split(":")
    assert callable(getattr(importlib.import_module(mod), fn))
    out = subprocess.run([sys.executable, "-c", "import sys; sys.argv = ['fithic', '-V']; from %s import %s as m; m()" % (mod, fn)],
                         capture_output=True, text=True, cwd=root, timeout=120)
    assert out.returncode == 0 and "Fit-Hi-C" in out.stdout
