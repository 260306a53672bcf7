"""pytest configuration: registers the `gpu` marker and shared fixtures."""
import os
import sys
import json

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")
DATA = os.path.join(GOLDEN, "data")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_case(name):
    """(meta, npz) of a golden case written by tests/golden/make_golden.py."""
    with open(os.path.join(GOLDEN, name + ".json")) as f:
        meta = json.load(f)
    return meta, np.load(os.path.join(GOLDEN, name + ".npz"))


def case_args(meta):
    """Decode the reference CLI tail stored in the golden meta into keyword arguments, applying the
    reference's "zero means unset" rule (fithic.py:194-221, 257-260)."""
    a = meta["argv"]

    def opt(flag, conv):
        return conv(a[a.index(flag) + 1]) if flag in a else None

    f = meta["files"]
    return dict(
        contacts=os.path.join(DATA, f["contacts"]), frags=os.path.join(DATA, f["frags"]),
        bias_path=os.path.join(DATA, f["bias"]) if f["bias"] else None,
        resolution=int(a[a.index("-r") + 1]),
        n_bins=opt("-b", int) or 100, passes=opt("-p", int) or 1, mode=opt("-x", str) or "intraOnly",
        L=opt("-L", int) or 0, U=opt("-U", int) or float("inf"), mapp_thres=opt("-m", int) or 1,
        tL=opt("-tL", float) or 0.5, tU=opt("-tU", float) or 2)


FIXED_CASES = ["f1_nobias", "f1_bias", "f2_all", "f2_inter", "f2_intra", "f2_all_nobias",
               "f6_quirk_all", "f6_quirk_intra_nobounds", "f6_quirk_zero_flags", "f6_quirk_mapp2", "f7_pfal_all"]
NONFIXED_CASES = ["f8_nonfixed_hESC", "f8_nonfixed_all", "f8_nonfixed_nobounds"]        # -r 0
OFFGRID_CASES = ["f11_offgrid_all", "f11_offgrid_intra"]     # -r N on loci that are not on one grid (fixed-size possible pairs)
MULTIPASS_CASES = ["f13_all_p3", "f13_quirk_p4", "f13_hESC_p3"]   # -p 3 / -p 4: one outlier list for the whole run (fithic.py:336-370)
# in-range and / or inter-chromosomal totals at and above 2^31: scipy's bdtrc narrows n to a C int (fithic.py:1070, 1101)
WRAP_CASES = ["f15_intra_2p31_all", "f15_intra_2p32_intra", "f15_inter_2p31_inter", "f15_both_2p32_all"]
ALL_CASES = FIXED_CASES + NONFIXED_CASES + OFFGRID_CASES + MULTIPASS_CASES + WRAP_CASES
SMALL_CASES = [c for c in ALL_CASES if not c.startswith("f1_") and c != "f13_hESC_p3"]


def bits_equal(a, b):
    a = np.ascontiguousarray(a, np.float64)
    b = np.ascontiguousarray(b, np.float64)
    return a.shape == b.shape and np.array_equal(a.view(np.int64), b.view(np.int64))


def max_abs_diff(a, b):
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    both_nan = np.isnan(a) & np.isnan(b)
    assert np.array_equal(np.isnan(a), np.isnan(b)), "NaN pattern differs"
    d = np.abs(np.where(both_nan, 0.0, a) - np.where(both_nan, 0.0, b))
    return float(d.max()) if d.size else 0.0
