"""CombineNearbyInteraction on the MI355X engine: merges nearby significant Fit-Hi-C contacts by connected-component
labelling (reference: fithic/utils/CombineNearbyInteraction.py), same flags and same output file.

    python -m fithic_amd.combine -i sig.txt.gz -H 1 -o merged.gz -r 5000 [-c 8] [-p 100] [-n 2] [-s 0]

Parsing and text formatting happen here (pandas / Python, like the reference's own str() of floats); every per-node
step - node table, neighbour search, components, statistics, ranking, greedy pick - runs in csrc/fhx_cni.hip.
Not reproduced: `-p 0` (its result depends on CPython's set iteration order, :417-437) and the reference's debug prints
(they list Python sets).  There is no CPU implementation here: without the library or a GPU the calls raise.
"""
import argparse
import gzip
import os
import sys

import numpy as np

from . import _capi

device = 0
HEADER = "\t".join(["chr1", "mid1", "chr2", "mid2", "CC", "p", "fdr", "bin1_low", "bin1_high", "bin2_low", "bin2_high", "sumCC", "StrongConn"])


def parse_args(args):
    """CombineNearbyInteraction.py:85-111 (like the reference, sys.argv is parsed)."""
    parser = argparse.ArgumentParser(description="Check the help flag")
    parser.add_argument("-i", "--InpFile", help="Input gzipped interaction Fit-Hi-C output file.", required=True)
    parser.add_argument("-H", "--headerInp", dest="headerInp", type=int, default=1,
                        help="If 1, indicates that input interaction file has a header line (such as field names). Default 1.")
    parser.add_argument("-o", "--OutFile", help="Output merged gzipped interaction file.", required=True)
    parser.add_argument("-r", "--resolution", help="Resolution of Fit-Hi-C run.", required=True)
    parser.add_argument("-c", "--conn", help="Rule of connectivity (8 or 4). Default is 8.", required=False, default=8, type=int,
                        dest="connectivity_rule")
    parser.add_argument("-p", "--percent", dest="TopPctElem", type=int, default=100,
                        help="Percentage of elements to be selected from each connected component. Default: 100.")
    parser.add_argument("-n", "--Neigh", dest="NeighborHoodBin", type=int, default=2,
                        help="Loops within this many bins of an included loop (both ends) are discarded. Default 2.")
    parser.add_argument("-s", "--order", dest="SortOrder", type=int, default=0,
                        help="0: significance values sorted ascending (default); 1: descending.")
    return parser.parse_args()


def read_significances(path, header=1):
    """Whitespace-split rows like the reference's awk / split(): columns 1-7 = chr1 mid1 chr2 mid2 CC p q (:147-149)."""
    import pandas as pd
    df = pd.read_csv(path, sep=r"\s+", header=None, skiprows=1 if header == 1 else 0, usecols=range(7), dtype=str, engine="c",
                     compression="gzip" if path.endswith(".gz") else None, keep_default_na=False)
    return df


def combine_records(df, bin_size, conn=8, pct=100, neigh=2, order=0):
    """-> (sorted chromosome names, record array in output order, info)."""
    intra = (df[0] == df[2]).to_numpy()
    names = sorted(set(df[0]))                                   # `sort -k1,1 | uniq` in the C locale (:204-212)
    ids = {name: i for i, name in enumerate(names)}
    sub = df[intra]
    chr_ids = sub[0].map(ids).to_numpy(np.int32)
    half = bin_size / 2
    n1 = np.trunc(sub[1].astype(np.float64).to_numpy() + half).astype(np.int64)      # int(float(mid) + bin_size/2)  (:301)
    n2 = np.trunc(sub[3].astype(np.float64).to_numpy() + half).astype(np.int64)
    cc = sub[4].to_numpy().astype(np.int64)                      # int(text): ValueError on "12.0", like the reference (:312)
    p = sub[5].to_numpy().astype(np.float64)                     # float(text), correctly rounded
    q = sub[6].to_numpy().astype(np.float64)
    cn = _capi.CniContext(device)
    try:
        cn.load(chr_ids, n1, n2, cc, p, q, bin_size)
        rec, info = cn.run(conn, pct, neigh, order)
    finally:
        cn.close()
    return names, rec, info


def format_lines(names, rec, bin_size):
    """The reference's str() arithmetic (:565-600): bins are floats n / bin_size, box indices are int() of them."""
    out = []
    for r in rec:
        b1, b2 = int(r["n_lo"]) / bin_size, int(r["n_hi"]) / bin_size
        lo1, hi1, lo2, hi2 = (b1 - 1) * bin_size, b1 * bin_size, (b2 - 1) * bin_size, b2 * bin_size
        mn1, mx1 = int(int(r["box_min_lo"]) / bin_size), int(int(r["box_max_lo"]) / bin_size)
        mn2, mx2 = int(int(r["box_min_hi"]) / bin_size), int(int(r["box_max_hi"]) / bin_size)
        total = (mx1 - mn1 + 1) * (mx2 - mn2 + 1)
        strong = (int(r["box_cells"]) * 1.0) / total
        chrom = names[int(r["chr"])]
        fields = (chrom, (lo1 + hi1) / 2, chrom, (lo2 + hi2) / 2, int(r["cc"]), float(r["p"]), float(r["q"]), (mn1 - 1) * bin_size,
                  mx1 * bin_size, (mn2 - 1) * bin_size, mx2 * bin_size, int(r["sum_cc"]), strong)
        out.append("\t".join(str(v) for v in fields))
    return out


def main():
    options = parse_args(sys.argv[1:])
    bin_size = int(options.resolution)
    conn, pct, neigh, order = int(options.connectivity_rule), int(options.TopPctElem), int(options.NeighborHoodBin), int(options.SortOrder)
    print("\n *** bin_size: ", bin_size)
    print("\n *** headerInp: ", int(options.headerInp))
    print("\n *** connectivity_rule: ", conn)
    print("\n *** TopPctElem: ", pct)
    print("\n *** NeighborHoodBinThr: ", neigh * bin_size)
    print("\n *** QValCol: ", 7)
    print("\n *** PValCol: ", 6)
    print("\n *** SortOrder: ", order)
    out_dir = os.path.dirname(os.path.realpath(options.OutFile))
    if not os.path.exists(out_dir):
        os.makedirs(out_dir)
    print("OutDir: ", str(out_dir))
    df = read_significances(options.InpFile, int(options.headerInp))
    names, rec, info = combine_records(df, bin_size, conn, pct, neigh, order)
    print("List of chromosomes considered: ", str(names))
    print("No of nodes: ", info.nodes, " connected components: ", info.components, " selected loops: ", info.selected)
    with gzip.open(options.OutFile, "wt") as f:
        f.write(HEADER)
        for ln in format_lines(names, rec, bin_size):
            f.write("\n" + ln)
    print("End of merging filtering loops !!! ")


if __name__ == "__main__":
    main()
