#!/usr/bin/env python3
"""profiles/k2_pmc_traffic.json (what bench.py reads for `roofline.traffic`) from a PMC summary of THIS tree:

    python profiles/update_traffic_json.py gpurun_out/r04/z_pmc.txt gpurun_out/r04/z_bench.json [source label]

The summary is profiles/run_pmc.sh's (FETCH_SIZE / WRITE_SIZE per kernel, gfx950 correction applied); the bench line of the same
command gives the rows of the dominant launch."""
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    pmc, bench = sys.argv[1], sys.argv[2]
    label = sys.argv[3] if len(sys.argv) > 3 else os.path.basename(pmc)
    line = [ln for ln in open(pmc) if ln.startswith("fhx::k2h_heavy")]
    if not line:
        raise SystemExit("no k2h_heavy row in " + pmc)
    f = line[0].split()
    read_mb, write_mb = float(f[-2]), float(f[-1])
    d = json.loads([ln for ln in open(bench) if ln.startswith("{")][-1])
    rows = int(d["roofline"]["rows_per_launch"])
    prev = {}
    path = os.path.join(ROOT, "profiles", "k2_pmc_traffic.json")
    if os.path.exists(path):
        prev = json.load(open(path))
    out = {"source": "%s (rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE, two separate passes, same tree as the bench line)" % label,
           "kernel": re.sub(r"\s+", " ", " ".join(f[:-5])), "rows_per_launch": rows,
           "hbm_read_bytes_per_launch": read_mb * 1e6, "hbm_write_bytes_per_launch": write_mb * 1e6,
           "correction": prev.get("correction"), "history": prev.get("history"),
           "hbm_bytes_per_heavy_row": (read_mb + write_mb) * 1e6 / rows}
    with open(path, "w") as fh:
        json.dump(out, fh, indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
