#!/usr/bin/env python3
"""The drop-in CLI at C5 scale on ONE GPU: 22 autosomes at 1 kb, ~1.9e9 cis + 1e8 trans rows written as the
reference's gz text files by the library's writer, then `python -m fithic_amd -x All -p 1` on them, stage times from FHX_TIMING, and
the first and the last rows of the significances file held against Python's formatting ('%e' / '%f', fithic/fithic.py:1202-1212) of
the values an engine run over the same rows holds (p, q from the device; ExpCC and the biases recomputed as the reference does,
fithic/fithic.py:1066-1116).

    python profiles/cli_c5.py [--max-chroms k] [--dir /dev/shm/cli_c5] [--check-rows 200000]

Refuses (one line, exit 3) when the box has not the memory for it: ~45 GB of host columns + the input file + the output file in
--dir (a RAM disk by default) + the engine's 158 GB of HBM."""
import argparse
import os
import struct
import subprocess
import sys
import time
import zlib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def members(path):
    """(offset, size) of every gzip member of a file written by this library: the 'FH' extra subfield carries the member's size"""
    out = []
    with open(path, "rb") as f:
        size = os.fstat(f.fileno()).st_size
        at = 0
        while at < size:
            f.seek(at)
            h = f.read(24)
            if len(h) < 24 or h[:3] != b"\x1f\x8b\x08" or not (h[3] & 4) or h[12:14] != b"FH":
                raise ValueError("not a size-tagged member at %d" % at)
            total = struct.unpack("<Q", h[16:24])[0]
            out.append((at, total))
            at += total
    return out


def inflate_members(path, chain):
    data = []
    with open(path, "rb") as f:
        for at, total in chain:
            f.seek(at)
            data.append(zlib.decompress(f.read(total), 31))
    return b"".join(data)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--max-chroms", type=int, default=0)
    ap.add_argument("--dir", default="/dev/shm/cli_c5")
    ap.add_argument("--check-rows", type=int, default=200000)
    ap.add_argument("--again-with", action="append", default=[], metavar="K=V[,K=V]", help="run the CLI once more with these environment settings")
    args = ap.parse_args()
    import numpy as np
    import torch
    import bench
    from fithic_amd import synth, _capi
    from fithic_amd.engine import Engine
    cfg = dict(bench.CONFIGS["C5"])
    res, L, U = cfg["res"], cfg["L"], cfg["U"]
    lengths = synth.HG19_AUTOSOMES[:args.max_chroms] if args.max_chroms else synth.HG19_AUTOSOMES
    genome = synth.Genome(res, lengths)
    est_rows = 2.0e9 * sum(lengths) / sum(synth.HG19_AUTOSOMES)
    need_gb = est_rows * (20 + 6 + 27) / 1e9 + 40                   # host columns + input gz + output gz + slack
    avail_gb = [int(ln.split()[1]) for ln in open("/proc/meminfo") if ln.startswith("MemAvailable")][0] / 1e6
    os.makedirs(args.dir, exist_ok=True)
    st = os.statvfs(args.dir)
    disk_gb = st.f_bavail * st.f_frsize / 1e9
    print("host memory available %.0f GB, %s has %.0f GB free, this run needs about %.0f GB (host columns + files)" % (avail_gb, args.dir, disk_gb, need_gb))
    if avail_gb < need_gb * 1.3 or disk_gb < est_rows * 33 / 1e9 * 1.3:
        print("REFUSED: not enough memory on this box for the C5 files (see the budget above)")
        return 3
    dev = torch.device("cuda", 0)
    t0 = time.time()
    cols_t, n, n_cis, n_trans = bench.build_rows(synth, torch, cfg, genome, list(range(len(genome))), 0, 1, dev)
    cols = [t[:n].cpu().numpy() for t in cols_t]
    del cols_t
    torch.cuda.empty_cache()
    print("generated %d rows (%d cis + %d trans) in %.1f s" % (n, n_cis, n - n_cis, time.time() - t0))
    out = args.dir
    t0 = time.time()
    _capi.host_write_contacts(out + "/contacts.gz", genome.names, *cols, gzip_level=1)
    print("contacts.gz: %.1f GB written in %.1f s (library writer, size-tagged members)" % (os.path.getsize(out + "/contacts.gz") / 1e9, time.time() - t0))
    import pandas as pd
    names = np.array(genome.names)
    f_chr, f_mid, f_hits = genome.fragments()
    pd.DataFrame({0: names[f_chr], 1: 0, 2: f_mid, 3: f_hits, 4: 1}).to_csv(out + "/frags.gz", sep="\t", header=False, index=False, compression="gzip")
    b_chr, b_mid, b_val = genome.bias_table()
    pd.DataFrame({0: names[b_chr], 1: b_mid, 2: b_val}).to_csv(out + "/bias.gz", sep="\t", header=False, index=False, compression="gzip")
    cmd = [sys.executable, "-m", "fithic_amd", "-i", out + "/contacts.gz", "-f", out + "/frags.gz", "-t", out + "/bias.gz", "-o", out + "/run",
           "-r", str(res), "-L", str(L), "-U", str(int(U)), "-x", "All", "-p", "1"]
    sig = out + "/run/FitHiC.spline_pass1.res%d.significances.txt.gz" % res

    def run_cli(command, extra_env, label):
        t0 = time.time()
        r = subprocess.run(command, cwd=ROOT, capture_output=True, text=True, env=dict(os.environ, FHX_TIMING="1", **extra_env))
        wall = time.time() - t0
        made = command[command.index("-o") + 1] + "/FitHiC.spline_pass1.res%d.significances.txt.gz" % res
        print("fithic -x All -p 1%s: wall %.1f s for %d rows (%.1f M rows/s end to end), output %.1f GB gz, rc %d" %
              (label, wall, n, n / wall / 1e6, os.path.getsize(made) / 1e9 if os.path.exists(made) else -1, r.returncode))
        for ln in r.stdout.splitlines() + r.stderr.splitlines():
            if ("took" in ln or "stage" in ln or ln.startswith("contacts on the device") or (ln.startswith("fhx_write_significances_device:") and "bytes in" in ln)
                    or ln.startswith("fhx_load_pairs_device") or ln.startswith("fhx_ingest_contacts_commit")):
                print("    " + ln[:300])
        if r.returncode != 0:
            print(r.stderr[-3000:])
        return r.returncode, made

    rc, _ = run_cli(cmd, {}, "")
    if rc != 0:
        return 1
    for spec in args.again_with:                                  # the same command once more under other settings (A/B on the same files)
        env2 = dict(kv.split("=", 1) for kv in spec.split(","))
        cmd2 = [c if c != out + "/run" else out + "/run2" for c in cmd]
        rc2, made2 = run_cli(cmd2, env2, " [" + spec + "]")
        if rc2 == 0:
            same = subprocess.run(["cmp", "-s", sig, made2]).returncode == 0
            print("    output file %s the first run's, byte for byte" % ("EQUALS" if same else "DIFFERS FROM"))
        subprocess.run(["rm", "-rf", out + "/run2"])
    # ---- the first and the last rows against Python's formatting of an engine run's values ----
    k = min(args.check_rows, n_cis, n - n_cis if n > n_cis else n_cis)
    t0 = time.time()
    eng = Engine(0)
    eng.configure(res, L, U, n_bins=100, mapp_thres=1, mode="All")
    eng.load_fragments(f_chr, f_mid, f_hits, genome.sort_rank())
    eng.load_bias(b_chr, b_mid, b_val)
    eng.load_contacts(*cols)
    o = eng.run_pass(collect=False)
    info, stats = o.info, o.stats
    table_x = eng.ctx.get_array(_capi.A_TABLE_X).astype(np.float64)
    table_y = eng.ctx.get_array(_capi.A_TABLE_Y)
    xs = eng.ctx.get_array(_capi.A_X)
    ok_all = True
    chain = members(sig)
    for label, lo in (("first", 0), ("last", n - k)):
        rows = np.arange(lo, lo + k)
        pq = []
        for which in (0, 1):
            buf = np.empty(k, np.float64)
            eng.ctx.copy(buf.ctypes.data, eng.ctx.device_ptr(which) + 8 * lo, 8 * k, 1)
            pq.append(buf)
        c1, m1, c2, m2, cnt = [a[rows] for a in cols]
        bias = {c: np.where((genome.bias(c) < 0.5) | (genome.bias(c) > 2.0), -1.0, genome.bias(c)) for c in np.unique(np.concatenate([c1, c2]))}
        b1 = np.array([bias[c][m // res] for c, m in zip(c1, m1)])
        b2 = np.array([bias[c][m // res] for c, m in zip(c2, m2)])
        inter = c1 != c2
        d = np.abs(m1.astype(np.int64) - m2.astype(np.int64))
        look = np.minimum(np.maximum(d.astype(np.float64), xs.min()), xs.max())
        idx = np.minimum(np.searchsorted(table_x, look, side="left"), len(table_x) - 1)
        in_range = ~inter & (d >= L) & (d <= U)
        prior = np.where(inter, info["inter_chr_prob"] * (b1 * b2), table_y[idx] * (b1 * b2))
        total = np.where(inter, float(stats["inter_sum"]), float(stats["in_range_sum"]))
        valid = (b1 >= 0.5) & (b1 <= 2.0) & (b2 >= 0.5) & (b2 <= 2.0)      # fithic.py:1075-1078, 1105-1108: else ExpCC = 0
        expcc = np.where(valid, total * prior, 0.0)
        want = []
        for i in range(k):
            if inter[i] or in_range[i]:                          # the rows the writer emits with -x All (fithic.py:1195-1212)
                want.append((genome.names[c1[i]], m1[i], genome.names[c2[i]], m2[i], cnt[i], pq[0][i], pq[1][i], b1[i], b2[i], expcc[i]))
        # the file's rows: the first members / the last members
        if label == "first":
            text = inflate_members(sig, chain[:2 + k // 65536 + 2])
            lines = text.split(b"\n")[1:]                        # (drop the header)
            got = lines[:len(want)]
        else:
            text = inflate_members(sig, chain[-(k // 65536 + 3):])
            lines = [ln for ln in text.split(b"\n") if ln]
            got = lines[-len(want):]
        bad = 0
        for w, g in zip(want, got):
            f = g.decode().split("\t")
            exp = "%s\t%d\t%s\t%d\t%d\t%e\t%e\t%e\t%e" % w[:9]
            if "\t".join(f) != exp + "\t%f" % w[9]:
                bad += 1
                if bad <= 3:
                    print("    differs: file %r, expected %r" % (g, exp + "\t%f" % w[9]))
        print("    %s %d input rows -> %d output rows: %s Python's formatting of the engine's values" % (label, k, len(want), "EQUAL" if bad == 0 and len(got) == len(want) else "%d DIFFER from" % bad))
        ok_all = ok_all and bad == 0 and len(got) == len(want)
    eng.close()
    print("checked in %.1f s; output members %d" % (time.time() - t0, len(chain)))
    for f in ("contacts.gz", "frags.gz", "bias.gz"):
        os.unlink(out + "/" + f)
    import shutil
    shutil.rmtree(out + "/run", ignore_errors=True)
    return 0 if ok_all else 1


if __name__ == "__main__":
    sys.exit(main())
