"""CPU tests of the small Python pieces around the C ABI: the numpy-free loader the command line starts first, and the lazily
built bias dictionary read_biases returns (fithic/fithic.py:798-837)."""
import threading

import numpy as np


def test_loader_is_numpy_free_and_warmup_never_raises():
    import subprocess
    import sys
    code = ("import sys; from fithic_amd import _loader; t = _loader.warm_in_background(0); t.join(60); "
            "print('numpy' in sys.modules, t.is_alive())")
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr
    assert out.stdout.split() == ["False", "False"]            # numpy not imported by the loader; the thread ended (GPU or not)


def test_loader_and_capi_share_one_library_object():
    from fithic_amd import _capi, _loader
    assert _capi.lib() is _loader.load() and _capi.LIB_PATH == _loader.LIB_PATH
    got = []
    ts = [threading.Thread(target=lambda: got.append(_loader.load())) for _ in range(8)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert all(g is got[0] for g in got)


def test_bias_dictionary_fills_itself_on_first_use():
    from fithic_amd.fithic import _BiasDic
    names = ["chr1", "chr2", "chrX"]
    bc = np.array([0, 0, 1, 1, 0, 2], np.int32)
    bm = np.array([100, 200, 100, 100, 100, 5], np.int32)          # (chr2, 100) and (chr1, 100) twice: the first occurrence wins
    vals = np.array([1.5, -1.0, 0.7, 0.9, 2.0, 1.0])
    want = {"chr1": {100: 1.5, 200: -1}, "chr2": {100: 0.7}, "chrX": {5: 1.0}}

    def fresh():
        return _BiasDic((names, bc, bm, vals), len(bc))
    d = fresh()
    assert bool(d) and dict.__len__(d) == 0
    assert d == want and dict.__len__(d) == 3
    assert isinstance(fresh()["chr1"][200], int) and fresh()["chr1"][200] == -1
    assert len(fresh()) == 3 and list(fresh()) == list(want) and "chr2" in fresh() and "chr9" not in fresh()
    assert fresh().get("chrX") == {5: 1.0} and fresh().get("nope", 7) == 7
    assert sorted(fresh().keys()) == sorted(want) and dict(fresh().items()) == want and list(fresh().values()) == list(want.values())
    assert fresh().copy() == want and repr(fresh()) == repr(want) and (fresh() != want) is False
    e = fresh()
    e.setdefault("chrY", {})[1] = 2.0
    assert e["chrY"] == {1: 2.0} and e["chr1"][100] == 1.5
    assert not _BiasDic((names, bc[:0], bm[:0], vals[:0]), 0) and not _BiasDic()


def test_myUtils_names_follow_the_reference_predicates():
    """fithic/myUtils.py:85-92: -1 = no bound, both ends inclusive; the oracle's vectorised in_range is the same predicate with
    0 / inf as 'no bound'."""
    from fithic_amd import myUtils
    from oracle import fithic_oracle as fo
    import numpy as np
    cases = [(d, lo, up) for d in (0, 1, 5000, 20000, 20001, 2000000, 2000001) for lo in (-1, 0, 20000) for up in (-1, 20000, 2000000)]
    for d, lo, up in cases:
        want = (lo == -1 or d >= lo) and (up == -1 or d <= up)
        assert myUtils.in_range_check(d, lo, up) is want, (d, lo, up)
        assert bool(fo.in_range(np.array([d]), max(lo, 0), float("inf") if up == -1 else up)[0]) is want
    assert myUtils.scale_a_list([1, 2], 0.5) == [0.5, 1.0]
    # the record class of fithic/myUtils.py:98-147, for scripts that built such objects themselves
    it = myUtils.Interaction(["chr1", "15000", "chr1", "45000"])
    assert (it.type, it.distance, it.getCount(), it.pval, it.qval) == ("intra", 30000, 0, -1.0, -1.0)
    assert it.getType(20000, 40000) == "intraInRange" and it.getType(30000, -1) == "intraInRange"
    assert it.getType(40000, -1) == "intraShort" and it.getType(-1, 20000) == "intraLong"
    it.setCount("7"); it.setPval("0.5"); it.setQval(1); it.setType(3)
    assert (it.hitCount, it.pval, it.qval, it.type) == (7, 0.5, 1.0, "3")
    tr = myUtils.Interaction(("chr1", 5, "chr2", 9))
    assert tr.type == "inter" and tr.getDistance() == -1 and tr.getType(-1, -1) == "inter"


def test_myStats_carries_the_one_name_of_the_path():
    """the module imports without a GPU (the BH name binds to the engine lazily)"""
    from fithic_amd import myStats
    assert callable(myStats.benjamini_hochberg_correction)
    assert myStats.meanAndVariance([1, 2, 3, 6]) == (3.0, 12.5 - 9.0)


def test_session_value_check_treats_untouched_nan_as_equal():
    """x / y handed back untouched must pass even where the host fit left NaN or inf (a 0-pair bin, where the reference's Python
    would have raised): NaN != NaN must not read as "the caller edited it"."""
    import pytest
    from fithic_amd.fithic import _check_session_values
    own = [1.0, float("nan"), float("inf"), 0.0]
    _check_session_values("x", list(own), own)
    _check_session_values("x", np.array(own), own)
    for edited in ([1.0, 2.0, float("inf"), 0.0], [1.0, float("nan"), float("inf")], [float("nan")] * 4):
        with pytest.raises(ValueError):
            _check_session_values("x", edited, own)


def test_the_ctypes_only_example_declares_the_structs_as_the_bindings_do():
    """examples/ctypes_minimal.py carries its own copies of fhx_params / fhx_stats / fhx_fit_info (it shows a binding without this
    package).  A field appended to the header must reach it too: the library writes the whole struct into the caller's memory."""
    import ctypes
    import importlib.util
    import os
    from fithic_amd import _capi
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "examples", "ctypes_minimal.py")
    spec = importlib.util.spec_from_file_location("ctypes_minimal", path)
    ex = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ex)
    for theirs, ours in ((ex.Params, _capi.FhxParams), (ex.Stats, _capi.FhxStats), (ex.FitInfo, _capi.FhxFitInfo)):
        assert ctypes.sizeof(theirs) == ctypes.sizeof(ours), theirs.__name__
        assert [(n, t) for n, t in theirs._fields_] == [(n, t) for n, t in ours._fields_], theirs.__name__
