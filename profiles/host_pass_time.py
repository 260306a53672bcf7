#!/usr/bin/env python3
"""Host fit of a pass (fhx_fit on a host-only context fed with the f14 fixture's histogram) timed on this machine's CPU:
what sits between K1 and k2_classify with the GPU idle.  FHX_FIT_TIMES=1 prints the stages.   python profiles/host_pass_time.py [C2 C3 ...]"""
import os
import sys
import time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench                                        # noqa: E402
from fithic_amd import _capi, synth                 # noqa: E402
from fithic_amd.engine import MODES                 # noqa: E402
from oracle import run_check                        # noqa: E402

for name in sys.argv[1:] or ["C2", "C3", "C3w", "C5"]:
    g = run_check.fit_fixture(name)
    cfg = bench.CONFIGS[name]
    genome = synth.Genome(cfg["res"], cfg["lengths"])
    ctx = _capi.Context(-1)
    ctx.set_params(cfg["res"], cfg["L"], cfg["U"], 100, 1, MODES[cfg["mode"]])
    ctx.load_fragments(*genome.fragments(), genome.sort_rank())
    st = _capi.FhxStats()
    st.inter_count, st.inter_sum, st.intra_all_sum, st.in_range_sum = [int(v) for v in g["sums"]]
    n_dist = int(g["hist_dist_idx"].max()) + 1
    hist_cc, hist_np = np.zeros(n_dist, np.int64), np.zeros(n_dist, np.int64)
    hist_cc[g["hist_dist_idx"]] = g["hist_sumcc"]
    hist_np[g["hist_dist_idx"]] = g["hist_nrows"]
    ctx.set_global_stats(st, hist_cc, hist_np)
    for _ in range(3):
        info = ctx.fit()
    reps = 100
    quiet = os.environ.pop("FHX_FIT_TIMES", None)
    t0 = time.perf_counter()
    for _ in range(reps):
        info = ctx.fit()
    dt = (time.perf_counter() - t0) / reps
    ok = run_check.compare_fit(ctx.get_array, info.as_dict(), g) == []
    print("%s: fhx_fit %.1f us per call, %d knots, table %d entries, bit-identical to the reference's fit: %s" %
          (name, dt * 1e6, info.n_knots, info.n_table, ok))
    ctx.close()
