#!/usr/bin/env python3
"""End-to-end wall time of the drop-in CLI at scale: the C3-synth map (or its first chromosomes) written as the
reference's gz text files (contacts, fragments, bias), then `python -m fithic_amd` on it, with the decompressed output
checked against the oracle's text on a sample of rows.  Prints stage times.

    python profiles/time_cli_scale.py [--chroms 22] [--plain] [--gpus N]

The contacts file is written by the library's own parallel writer (size-tagged gzip members, which the reader inflates on
all cores); --plain rewrites it as ONE plain gzip member (what `gzip` produces: a single deflate stream, inflated by one
thread) to show that case too.
"""
import argparse
import gzip
import hashlib
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def plain_member(src, dst, level=1, piece=64 << 20):
    """src (any gzip file) rewritten as ONE plain gzip member - one deflate stream, no size fields: what `gzip` writes - the way
    pigz does it on many cores: pieces of the text deflated on their own, every one but the last ended by a sync flush (an empty
    stored block on a byte boundary), concatenated behind one header, one CRC-32 + ISIZE trailer."""
    import struct
    import zlib
    from concurrent.futures import ThreadPoolExecutor
    from fithic_amd import _capi
    t = _capi.HostText(src, 0)
    text = memoryview(t.bytes())
    t.close()
    cuts = list(range(0, len(text), piece)) or [0]

    def one(k):
        z = zlib.compressobj(level, zlib.DEFLATED, -15)
        last = k + 1 == len(cuts)
        body = z.compress(text[cuts[k]:cuts[k] + piece])
        return body + z.flush(zlib.Z_FINISH if last else zlib.Z_FULL_FLUSH), zlib.crc32(text[cuts[k]:cuts[k] + piece]), len(text[cuts[k]:cuts[k] + piece])

    crc = 0
    with open(dst, "wb") as f, ThreadPoolExecutor(max(1, (os.cpu_count() or 2) // 2)) as pool:
        f.write(b"\x1f\x8b\x08\x00\0\0\0\0\x04\xff")
        for k, (body, c, n) in enumerate(pool.map(one, range(len(cuts)))):
            f.write(body)
            crc = c if k == 0 else _capi.crc32_combine(crc, c, n)
        f.write(struct.pack("<II", crc, len(text) & 0xffffffff))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--chroms", type=int, default=22)
    ap.add_argument("--plain", action="store_true")
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--passes", type=int, nargs="*", default=[1])
    ap.add_argument("--check-rows", type=int, default=200000)
    ap.add_argument("--rocprof", default=None, help="directory: run the CLI once more under rocprofv3 --kernel-trace --stats into it")
    ap.add_argument("--compare-host-writer", action="store_true",
                    help="run once more with FHX_HOST_WRITER=1 and compare md5 + line count of the two decompressed files (all rows)")
    ap.add_argument("--compare-host-reader", action="store_true",
                    help="run once more with FHX_HOST_READER=1 (contacts parsed on the host cores) and compare the output files' md5")
    ap.add_argument("--dir", default="/tmp/cli_scale", help="where the input and output files go (/dev/shm/... takes the disk out)")
    ap.add_argument("--reuse", action="store_true", help="take the input files --dir already holds (a run after another with the same --chroms)")
    ap.add_argument("--md5", action="store_true", help="print the md5 and line count of the decompressed significances file")
    ap.add_argument("--tag", default="run", help="output sub-directory")
    args = ap.parse_args()
    import numpy as np
    import bench
    from fithic_amd import synth, _capi
    cfg = dict(bench.CONFIGS["C3"])
    res = cfg["res"]
    out = args.dir
    os.makedirs(out, exist_ok=True)
    if args.reuse and all(os.path.exists(out + "/" + f) for f in ("contacts.gz", "frags.gz", "bias.gz", "rows.txt")):
        n = int(open(out + "/rows.txt").read())
        print("reusing the input files of %s (%d contact rows)" % (out, n))
    else:
        import torch
        genome = synth.Genome(res, synth.HG19_AUTOSOMES[:args.chroms] if args.chroms < 22 else None)
        dev = torch.device("cuda", 0)
        cols_t, n, _, _ = bench.build_rows(synth, torch, cfg, genome, list(range(len(genome))), 0, 1, dev)
        cols = [t[:n].cpu().numpy() for t in cols_t]
        del cols_t
        torch.cuda.empty_cache()
        t0 = time.time()
        _capi.host_write_contacts(out + "/contacts.gz", genome.names, *cols, gzip_level=1)
        t_w = time.time() - t0
        if args.plain:
            t0 = time.time()
            plain_member(out + "/contacts.gz", out + "/contacts_plain.gz")
            os.replace(out + "/contacts_plain.gz", out + "/contacts.gz")
            print("rewritten as one plain gzip member in %.1f s" % (time.time() - t0))
        import pandas as pd
        names = np.array(genome.names)
        f_chr, f_mid, f_hits = genome.fragments()
        pd.DataFrame({0: names[f_chr], 1: 0, 2: f_mid, 3: f_hits, 4: 1}).to_csv(out + "/frags.gz", sep="\t", header=False, index=False,
                                                                              compression="gzip")
        b_chr, b_mid, b_val = genome.bias_table()
        pd.DataFrame({0: names[b_chr], 1: b_mid, 2: b_val}).to_csv(out + "/bias.gz", sep="\t", header=False, index=False, compression="gzip")
        open(out + "/rows.txt", "w").write(str(n))
        print("wrote %d contact rows (%.1f MB gz, %s) in %.1f s; host cores %d" %
              (n, os.path.getsize(out + "/contacts.gz") / 1e6, "one plain member" if args.plain else "size-tagged members", t_w, os.cpu_count()))
    for passes in args.passes:
        t0 = time.time()
        cmd = [sys.executable, "-m", "fithic_amd", "-i", out + "/contacts.gz", "-f", out + "/frags.gz", "-t", out + "/bias.gz",
               "-o", out + "/" + args.tag, "-r", str(res), "-L", str(cfg["L"]), "-U", str(cfg["U"]), "-p", str(passes)]
        if args.gpus > 1:
            cmd += ["--gpus", str(args.gpus)]
        r = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, env=dict(os.environ, FHX_TIMING="1"))
        dt = time.time() - t0
        sig = out + "/" + args.tag + "/FitHiC.spline_pass%d.res%d.significances.txt.gz" % (passes, res)
        print("fithic -p %d%s: wall %.2f s for %d rows (%.2f M rows/s end to end), output %.1f MB gz, rc %d" %
              (passes, " --gpus %d" % args.gpus if args.gpus > 1 else "", dt, n, n / dt / 1e6,
               os.path.getsize(sig) / 1e6 if os.path.exists(sig) else -1, r.returncode))
        for ln in r.stdout.splitlines() + r.stderr.splitlines():
            if "took" in ln or "Time" in ln or "stage" in ln or ln.startswith("fhx_") or ln.startswith("contacts on the device") or ln.startswith("parallel gunzip") or ln.startswith("rank "):
                print("    " + ln)
        if r.returncode != 0:
            print(r.stderr[-2000:])
            continue
        if args.md5:
            pr = subprocess.run("gzip -dc %s | tee >(wc -l >&2) | md5sum" % sig, shell=True, executable="/bin/bash", capture_output=True, text=True)
            print("    decompressed output: %s lines, md5 %s" % (pr.stderr.strip(), pr.stdout.split()[0]))
        if args.rocprof:
            env = dict(os.environ, TMPDIR="/tmp")
            rp = subprocess.run(["rocprofv3", "--kernel-trace", "--stats", "-d", args.rocprof, "-o", "cli", "--"] + cmd, cwd="/tmp",
                                capture_output=True, text=True, env=dict(env, PYTHONPATH=ROOT))
            print("rocprofv3 run of the same command: rc %d" % rp.returncode)
        if args.compare_host_writer:
            t0 = time.time()
            cmd_h = [c if c != out + "/run" else out + "/run_host" for c in cmd]
            rh = subprocess.run(cmd_h, cwd=ROOT, capture_output=True, text=True, env=dict(os.environ, FHX_TIMING="1", FHX_HOST_WRITER="1"))
            dt_h = time.time() - t0
            sig_h = sig.replace(out + "/run/", out + "/run_host/")
            print("same run with the host writer (FHX_HOST_WRITER=1): wall %.2f s, output %.1f MB gz, rc %d" %
                  (dt_h, os.path.getsize(sig_h) / 1e6 if os.path.exists(sig_h) else -1, rh.returncode))
            for ln in rh.stdout.splitlines():
                if "stage" in ln:
                    print("    " + ln)
            procs = [subprocess.Popen("gzip -dc %s | tee >(wc -l >&2) | md5sum" % f, shell=True, executable="/bin/bash",
                                      stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True) for f in (sig, sig_h)]
            res_ = [p.communicate() for p in procs]
            md5s = [o.split()[0] for o, _ in res_]
            lines = [int(e.strip()) for _, e in res_]
            print("    all rows: device writer %d lines md5 %s; host writer %d lines md5 %s -> %s (expected %d lines)" %
                  (lines[0], md5s[0], lines[1], md5s[1], "EQUAL" if md5s[0] == md5s[1] and lines[0] == lines[1] else "DIFFERENT", n + 1))
        if args.compare_host_reader:
            t0 = time.time()
            cmd_h = [c if c != out + "/run" else out + "/run_hostreader" for c in cmd]
            rh = subprocess.run(cmd_h, cwd=ROOT, capture_output=True, text=True, env=dict(os.environ, FHX_TIMING="1", FHX_HOST_READER="1"))
            dt_h = time.time() - t0
            sig_h = sig.replace(out + "/run/", out + "/run_hostreader/")
            print("same run with the host parser (FHX_HOST_READER=1): wall %.2f s, rc %d" % (dt_h, rh.returncode))
            for ln in rh.stdout.splitlines() + rh.stderr.splitlines():
                if "stage: inflate" in ln or ln.startswith("fhx_host_read_table") and "contacts" in ln:
                    print("    " + ln)
            md5s = []
            for f in (sig, sig_h):
                with open(f, "rb") as fh:
                    md5s.append(hashlib.md5(fh.read()).hexdigest())
            print("    significances file (compressed bytes) md5: device parser %s, host parser %s -> %s" %
                  (md5s[0], md5s[1], "EQUAL" if md5s[0] == md5s[1] else "DIFFERENT"))
        if passes == 1 and args.check_rows > 0 and "cols" in dir():       # (not with --reuse: the rows are not in memory)
            # the first rows of the output against the oracle's text for the same rows (p, q, biases, ExpCC from the engine's
            # fetch are formatted by the oracle's Python '%e' / '%f'; the row selection and the order are the reference's)
            from fithic_amd.engine import Engine
            t0 = time.time()
            eng = Engine(0)
            eng.configure(res, cfg["L"], cfg["U"], n_bins=100, mapp_thres=1, mode="intraOnly")
            eng.load_fragments(f_chr, f_mid, f_hits, genome.sort_rank())
            eng.load_bias(b_chr, b_mid, b_val)
            eng.load_contacts(*cols)
            eng.run_pass(collect=False)
            v = eng.fetch(p=True, q=True, expcc=True, bias=True)
            eng.close()
            k = min(args.check_rows, n)
            want = ["chr1\tfragmentMid1\tchr2\tfragmentMid2\tcontactCount\tp-value\tq-value\tbias1\tbias2\tExpCC\n"]
            for i in range(k):
                want.append("%s\t%d\t%s\t%d\t%d\t%e\t%e\t%e\t%e\t%f\n" % (names[cols[0][i]], cols[1][i], names[cols[2][i]], cols[3][i],
                                                                         cols[4][i], v["p"][i], v["q"][i], v["b1"][i], v["b2"][i], v["expcc"][i]))
            want = "".join(want).encode()
            with gzip.open(sig, "rb") as f:
                got = f.read(len(want))
            print("    first %d output rows %s Python's formatting of the same values (md5 %s), checked in %.1f s" %
                  (k, "EQUAL" if got == want else "DIFFER FROM", hashlib.md5(got).hexdigest(), time.time() - t0))


if __name__ == "__main__":
    main()
