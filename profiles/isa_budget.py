#!/usr/bin/env python3
"""Static instruction budget of one kernel by source region, from hipcc's assembly with line tables.

    hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-fast-math -gline-tables-only -S --cuda-device-only \
          -o /tmp/fhx_k2_g.s fithic_amd/csrc/fhx_k2.hip
    python profiles/isa_budget.py /tmp/fhx_k2_g.s k2_classify 'ILi0ELi4ELi0ELb1'

Every instruction carries the innermost source position it was generated from (.loc; inlined callees keep their own file and
line), so a range of lines of fhx_k2.hip / fhx_bdtrc.hpp names a piece of the algorithm.  The regions are found by the
marker comments / function names in the sources (REGIONS below), not by line numbers typed here.  Counts are STATIC wave
instructions; the kernel's tile loop executes each block once per step of 4 rows per lane unless noted, so static / 4 is the
per-row cost (a wave pays for every branch any of its lanes takes - on Hi-C rows that is all of them)."""
import re
import sys
from collections import defaultdict

ROOT = __file__.rsplit("/profiles/", 1)[0]


def line_ranges():
    """name -> (file, first line, last line), located by text anchors in the sources"""
    dev = open(ROOT + "/fithic_amd/csrc/fhx_k2.hip").read().split("\n")
    bd = open(ROOT + "/fithic_amd/csrc/fhx_bdtrc.hpp").read().split("\n")

    def find(lines, text, start=0):
        for i in range(start, len(lines)):
            if text in lines[i]:
                return i + 1
        raise KeyError(text)

    def func_end(lines, first):              # the closing brace in column 0 after `first`
        for i in range(first, len(lines)):
            if lines[i].startswith("}"):
                return i + 1
        return len(lines)
    R = {}
    a = find(dev, "void rows_prior_fixed(")
    R["prior: 12 gathers + branch table (rows_prior_fixed)"] = ("fhx_k2.hip", a, func_end(dev, a))
    a = find(dev, "bool row_prior(const K2Params& P")
    R["prior, row by row (row_prior)"] = ("fhx_k2.hip", a, func_end(dev, a))
    a = find(dev, "struct FusedHist {")
    R["K3's key histogram (FusedHist)"] = ("fhx_k2.hip", a, func_end(dev, a))
    k = find(dev, "void k2_classify(K2Params P, K2Queues Q)")
    R["column loads, tile loop, row bounds"] = ("fhx_k2.hip", k, find(dev, "int cls_of[ITEMS];", k) - 1)
    b = find(dev, "int cls_of[ITEMS];", k)
    c = find(dev, "// slot reservation for the whole wave at once", k)
    R["class dispatch, p of trivial rows (store)"] = ("fhx_k2.hip", b, c - 1)
    d = find(dev, "unsigned int wave_base[K2_QUEUES + 1];", k)
    R["slot reservation (ballots / prefix, LDS atomic)"] = ("fhx_k2.hip", c, d - 1)
    e = find(dev, "// the small-prior closed-form rows of this wave", k)
    R["queue entries + closed-form strip writes"] = ("fhx_k2.hip", d, e - 1)
    f = find(dev, "__syncthreads();", e)
    R["closed-form strip: loop control, loads, p store"] = ("fhx_k2.hip", e, f - 1)
    R["epilogue: counters, heavy histogram, flush"] = ("fhx_k2.hip", f, func_end(dev, f))
    a = find(bd, "int bdtrc_class(int count, double n_total, double p)")
    R["incbet's branch predicates (bdtrc_class)"] = ("fhx_bdtrc.hpp", a, func_end(bd, a))
    a = find(bd, "bool cls_is_trivial(int count, double n_total, double p)")
    b3 = find(bd, "int bdtrc_class_tb(int count, double n_total, double p, double tB)")
    R["incbet's branch predicates, orientation threshold given (bdtrc_class_tb)"] = ("fhx_bdtrc.hpp", a, func_end(bd, b3))
    a = find(bd, "bool bdtrc_is_closed_form(")
    b2 = find(bd, "double bdtrc_count_trivial_open(")
    R["trivial rows: constants (bdtrc_is_closed_form / _trivial_open)"] = ("fhx_bdtrc.hpp", a, func_end(bd, b2))
    for name in ("cephes_log1p", "cephes_expm1"):
        a = find(bd, "double %s(" % name)
        R["closed form: %s" % name] = ("fhx_bdtrc.hpp", a, func_end(bd, a))
    return R


def main():
    path, kernel, inst = sys.argv[1], sys.argv[2], (sys.argv[3] if len(sys.argv) > 3 else "")
    regions = line_ranges()
    files = {}
    counts = defaultdict(lambda: defaultdict(int))
    inside, cur = False, None
    for ln in open(path):
        if not inside:
            m = re.match(r"^(_Z\w+):", ln)
            if m and kernel in m.group(1) and inst in m.group(1) and not m.group(1).startswith("_ZN3fhx") is False:
                inside = True
            m = re.match(r'\s*\.file\s+(\d+)\s+"[^"]*"\s+"([^"]+)"', ln)
            if m:
                files[int(m.group(1))] = m.group(2).rsplit("/", 1)[-1]
            continue
        m = re.match(r'\s*\.file\s+(\d+)\s+"[^"]*"\s+"([^"]+)"', ln)
        if m:
            files[int(m.group(1))] = m.group(2).rsplit("/", 1)[-1]
            continue
        m = re.match(r"\s*\.loc\s+(\d+)\s+(\d+)", ln)
        if m:
            if int(m.group(2)) != 0:
                cur = (files.get(int(m.group(1)), "?"), int(m.group(2)))
            continue
        t = ln.strip()
        if t.startswith("s_endpgm"):
            break
        if not t or t[0] in ";." or t.endswith(":"):
            continue
        op = t.split()[0]
        kind = ("valu_f64" if op.startswith("v_") and "f64" in op else "valu" if op.startswith("v_") else
                "salu" if op.startswith("s_") else "lds" if op.startswith("ds_") else
                "vmem" if op.startswith(("global_", "buffer_", "flat_", "scratch_")) else "other")
        name = "other (%s)" % (cur[0] if cur else "?")
        if cur and cur[0] == "fhx_ctx.hpp":
            name = "wave helpers, queue addressing (fhx_ctx.hpp)"
        if cur:
            for rn, (f, a, b) in regions.items():
                if cur[0] == f and a <= cur[1] <= b:
                    name = rn
                    break
        counts[name][kind] += 1
    tot = defaultdict(int)
    print("%-66s %6s %6s %6s %5s %5s" % ("region", "VALU", "(f64)", "SALU", "LDS", "VMEM"))
    for name, c in sorted(counts.items(), key=lambda kv: -(kv[1]["valu"] + kv[1]["valu_f64"])):
        v = c["valu"] + c["valu_f64"]
        print("%-66s %6d %6d %6d %5d %5d" % (name[:66], v, c["valu_f64"], c["salu"], c["lds"], c["vmem"]))
        for k, n in c.items():
            tot[k] += n
    print("%-66s %6d %6d %6d %5d %5d" % ("total (static)", tot["valu"] + tot["valu_f64"], tot["valu_f64"], tot["salu"], tot["lds"], tot["vmem"]))


if __name__ == "__main__":
    main()
