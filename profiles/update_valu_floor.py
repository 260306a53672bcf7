#!/usr/bin/env python3
"""profiles/valu_floor.json (what bench.py reads for `roofline.valu_issue_floor_ms`) from a counters summary of THIS tree:

    python profiles/update_valu_floor.py <counters.txt of run_pmc_counters.sh> <bench.json of the same command> [config] [label]

floor = sum over the pass's kernels of SQ_INSTS_VALU (wave-instructions per launch) x 4 cycles / 1024 SIMDs / clock, the clock
taken from the run itself (GRBM_GUI_ACTIVE of the dominant launch / its HIP-event duration): the time the pass would take if
every SIMD issued one VALU instruction every four cycles without a gap.  K2's fp64 work binds this path, not HBM
(DESIGN.md 4, 7): the pass's distance to THIS floor is what is left to gain."""
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SIMDS, CYCLES_PER_WAVE_INSTR, XCDS = 1024, 4, 8           # (GRBM_GUI_ACTIVE is summed over the eight XCDs)


def main():
    counters, bench = sys.argv[1], sys.argv[2]
    config = sys.argv[3] if len(sys.argv) > 3 else "C3"
    label = sys.argv[4] if len(sys.argv) > 4 else os.path.basename(counters)
    valu, active = {}, {}
    for ln in open(counters):
        m = re.match(r"(\S+)\s+(.*?)\s+launches\s+(\d+) total (\S+) per-launch (\S+)", ln)
        if not m:
            continue
        (valu if m.group(1) == "SQ_INSTS_VALU" else active if m.group(1) == "GRBM_GUI_ACTIVE" else {})[m.group(2).strip()] = float(m.group(5))
    d = json.loads([ln for ln in open(bench) if ln.startswith("{")][-1])
    heavy = [k for k in valu if k.startswith("k2h_heavy")][0]
    clock_hz = active[heavy] / XCDS / d["roofline"]["launch_seconds"]
    total = sum(valu.values())
    floor_ms = 1e3 * total * CYCLES_PER_WAVE_INSTR / SIMDS / clock_hz
    path = os.path.join(ROOT, "profiles", "valu_floor.json")
    out = json.load(open(path)) if os.path.exists(path) else {}
    out[config] = {"source": "%s (rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU / GRBM_GUI_ACTIVE, separate passes)" % label,
                   "pairs": d["config"]["pairs"], "valu_wave_instructions_per_pass": total, "by_kernel": valu, "clock_ghz": clock_hz / 1e9,
                   "simds": SIMDS, "cycles_per_wave_instruction": CYCLES_PER_WAVE_INSTR, "floor_ms": floor_ms,
                   "pass_ms_of_that_run": d["ms_per_step"], "heavy_share_of_instructions": valu[heavy] / total}
    json.dump(out, open(path, "w"), indent=1)
    print(json.dumps(out[config], indent=1))


if __name__ == "__main__":
    main()
