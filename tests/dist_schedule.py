"""TEST INFRASTRUCTURE - reader of fithic_amd/csrc/fhx_dist_schedule.def, the one list of the collectives of a sharded run.

Both statements of the schedule are held against it: the library's (a trace of what fhx_dist.inc issued, tests/test_gpu_native_dist.py)
and the Python model's (tests/dist_model.py over gloo, tests/test_dist_gloo.py).  A change on one side only fails a test."""
import os
import re

DEF = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "fithic_amd", "csrc", "fhx_dist_schedule.def")


def steps():
    """[(id, kind, size expression, set of phases)] in file order"""
    out = []
    for line in open(DEF):
        m = re.match(r'\s*FHX_DIST_STEP\(\s*(\w+)\s*,\s*(\w+)\s*,\s*"([^"]*)"\s*,\s*([\w|]+)\s*\)', line)
        if m:
            out.append((m.group(1), m.group(2), m.group(3), set(m.group(4).split("|"))))
    assert out, "no FHX_DIST_STEP rows found"
    return out


def expected(phase, **env):
    """[(id, kind, size or None)] of one phase; a size is None when its expression names a value env does not hold"""
    rows = []
    for ident, kind, size, phases in steps():
        if phase in phases:
            try:
                val = int(eval(size, {"__builtins__": {}}, dict(env)))
            except NameError:
                val = None
            rows.append((ident, kind, val))
    assert rows, "no such phase: %s" % phase
    return rows


def check(trace, phases_in_order, envs):
    """trace: [(id, kind, size)] as recorded; phases_in_order: e.g. ["LOAD", "STATS", "BH", "NEXT", "STATS", "BH"]; envs: one dict
    per phase (world, w, s, nd, n_local, m, width, longest - whatever the phase's size expressions need; a missing name leaves that
    size unchecked).  An all-reduce / all-gather of size 0 is not issued (and NF_LISTS not when every list is empty).
    Raises AssertionError naming the first difference."""
    want = []
    for phase, env in zip(phases_in_order, envs):
        want += [r for r in expected(phase, **env) if r[2] is None or r[2] > 0 or r[1] == "ALL_TO_ALL_V"]
    got = [tuple(r) for r in trace]
    assert [g[:2] for g in got] == [w_[:2] for w_ in want], (
        "collectives issued differ from fhx_dist_schedule.def\n issued   %s\n schedule %s" % ([g[0] for g in got], [w_[0] for w_ in want]))
    for i, (g, w_) in enumerate(zip(got, want)):
        assert w_[2] is None or g[2] == w_[2], "collective %d (%s): size %d issued, the schedule gives %d" % (i, g[0], g[2], w_[2])
