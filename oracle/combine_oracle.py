"""TEST INFRASTRUCTURE - CPU oracle of fithic/utils/CombineNearbyInteraction.py (merging of nearby significant contacts by
connected-component labelling; SURVEY.md 8f rank 4, the step after Fit-Hi-C).  Pure Python, no networkx.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this; the product (fithic_amd.combine) never
does.  Pinned by tests/golden/c*_combine_*.json: output files of the real reference (make_golden.py f10).

What fixes the output (reference lines in brackets):
  * chromosomes: distinct values of column 1, byte-sorted (`sort -k1,1 | uniq`, C locale) [204-212]; rows with $1 == $3 == chr [258-266]
  * node = (bin_lo, bin_hi), bin = int(float(mid) + res/2) / res (true division) [301-309]; first row of a node keeps its
    CC (col 5), p (col 6), q (col 7) [312]
  * edges: |d bin_lo| <= 1 and |d bin_hi| <= 1 (8) or |d bin_lo| + |d bin_hi| <= 1 (4) [324-340]
  * components in order of their first node in the file, then stably sorted by size, largest first [354]
  * per component: bounding box, sumCC, fraction of box cells that are nodes of the chromosome [367-402]
  * -p 100 / 0 < -p < 100: nodes by (q, -CC, bin_lo, bin_hi) (a total order), greedy pick, a pick suppresses later nodes
    of the component within -n bins in both coordinates [470-600]; 0 < -p < 100 stops at the component's percentile
    (custom_percent [36-52]); with -s 1 the stop test compares -q with a positive bound and stops at once [508] (kept)
  * -p 0 depends on CPython's set iteration order (non-total comparisons over list(set)) [417-437]: not reproducible, raises.
"""
import gzip


def custom_percent(lst, K, order=1):
    s = sorted(lst) if order == 1 else sorted(lst, reverse=True)
    index = int((len(lst) * K) / 100)
    if index <= 1:
        return max(s) if order == 1 else min(s)
    return s[index]


def _open(path):
    return gzip.open(path, "rt") if path.endswith(".gz") else open(path, "rt")


HEADER = "\t".join(["chr1", "mid1", "chr2", "mid2", "CC", "p", "fdr", "bin1_low", "bin1_high", "bin2_low", "bin2_high", "sumCC", "StrongConn"])


def combine_lines(in_path, res, header=1, conn=8, pct=100, neigh=2, order=0):
    """-> list of output lines (without the header), in the reference's order."""
    if pct == 0:
        raise NotImplementedError("-p 0 depends on CPython set iteration order (CombineNearbyInteraction.py:417-437)")
    bin_size = int(res)
    thr = int(neigh) * bin_size
    with _open(in_path) as f:
        rows = [ln.split() for k, ln in enumerate(f) if not (header == 1 and k == 0)]
    rows = [w for w in rows if w]
    out = []
    for chrom in sorted({w[0] for w in rows}):
        nodes = {}
        for w in rows:
            if w[0] != chrom or w[2] != chrom:
                continue
            b1 = int(float(w[1]) + (bin_size / 2)) / bin_size
            b2 = int(float(w[3]) + (bin_size / 2)) / bin_size
            key = (b1, b2) if b1 < b2 else (b2, b1)
            nodes.setdefault(key, (int(w[4]), float(w[5]), float(w[6])))
        if not nodes:
            continue
        keys = list(nodes)
        index = {k: i for i, k in enumerate(keys)}
        # adjacency through unit buckets: a neighbour differs by <= 1 in both coordinates
        buckets = {}
        for k in keys:
            buckets.setdefault((int(k[0] // 1), int(k[1] // 1)), []).append(k)
        parent = list(range(len(keys)))

        def find(a):
            while parent[a] != a:
                parent[a] = parent[parent[a]]
                a = parent[a]
            return a

        for k in keys:
            g0, g1 = int(k[0] // 1), int(k[1] // 1)
            for d0 in (-1, 0, 1):
                for d1 in (-1, 0, 1):
                    for o in buckets.get((g0 + d0, g1 + d1), ()):
                        if o is k:
                            continue
                        a, b = abs(k[0] - o[0]), abs(k[1] - o[1])
                        if (conn == 8 and a <= 1 and b <= 1) or (conn == 4 and a + b <= 1):
                            ra, rb = find(index[k]), find(index[o])
                            if ra != rb:
                                parent[max(ra, rb)] = min(ra, rb)
        comps = {}
        for i, k in enumerate(keys):
            comps.setdefault(find(i), []).append(k)                  # roots are the first node of each component
        comp_list = sorted((comps[r] for r in sorted(comps)), key=len, reverse=True)
        for comp in comp_list:
            mn1, mx1 = int(min(x[0] for x in comp)), int(max(x[0] for x in comp))
            mn2, mx2 = int(min(x[1] for x in comp)), int(max(x[1] for x in comp))
            span = ((mn1 - 1) * bin_size, mx1 * bin_size, (mn2 - 1) * bin_size, mx2 * bin_size)
            sum_cc = sum(nodes[x][0] for x in comp)
            total = (mx1 - mn1 + 1) * (mx2 - mn2 + 1)
            have = sum(1 for a in range(mn1, mx1 + 1) for b in range(mn2, mx2 + 1) if (a, b) in nodes)
            strong = (have * 1.0) / total
            ranked = sorted(([nodes[x][2] if order == 0 else -nodes[x][2], -nodes[x][0], x[0], x[1]] for x in comp))
            bound = None
            if 0 < pct < 100:
                bound = custom_percent([nodes[x][2] for x in comp], pct, order + 1)
            picked = []
            for e in ranked:
                if bound is not None and ((order == 0 and e[0] > bound) or (order == 1 and e[0] < bound)):
                    break
                if any(abs(p[0] - e[2]) * bin_size <= thr and abs(p[1] - e[3]) * bin_size <= thr for p in picked):
                    continue
                picked.append((e[2], e[3]))
            for k in picked:
                lo1, hi1, lo2, hi2 = (k[0] - 1) * bin_size, k[0] * bin_size, (k[1] - 1) * bin_size, k[1] * bin_size
                cc, p, q = nodes[k]
                out.append("\t".join(str(v) for v in (chrom, (lo1 + hi1) / 2, chrom, (lo2 + hi2) / 2, cc, p, q) + span + (sum_cc, strong)))
    return out


def combine(in_path, out_path, res, **kw):
    lines = combine_lines(in_path, res, **kw)
    with gzip.open(out_path, "wt") as f:
        f.write(HEADER)
        for ln in lines:
            f.write("\n" + ln)
    return len(lines)
