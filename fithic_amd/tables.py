"""Input tables of Fit-Hi-C as SoA numpy arrays + the output writer.

File formats are the reference's (README.md:139-208 of ay-lab/fithic):
  contacts   gz text, whitespace separated  chr1 mid1 chr2 mid2 count      (fithic/fithic.py:413-417)
  fragments  gz text, 5 columns, uses col 0 (chr), 2 (mid), 3 (hits)         (fithic/fithic.py:581-590)
  bias       gz text  chr mid bias                                            (fithic/fithic.py:805-808)
Chromosome names of all three files are interned into one id space (`ChromIndex`); the reference iterates
chromosomes in Python's sorted() string order (fithic/fithic.py:606), which `sort_rank()` hands to the engine.
"""
import gzip
import io

import numpy as np


class ChromIndex:
    """Chromosome name <-> small int id, shared by the three tables."""

    def __init__(self):
        self.names = []
        self._ids = {}

    def intern_column(self, col):
        """ids (int32) of an object / str column; new names get new ids in order of first appearance."""
        import pandas as pd
        codes, uniques = pd.factorize(col, sort=False)
        remap = np.empty(len(uniques), np.int32)
        for k, name in enumerate(uniques):
            name = str(name)
            j = self._ids.get(name)
            if j is None:
                j = self._ids[name] = len(self.names)
                self.names.append(name)
            remap[k] = j
        return remap[codes]

    def sort_rank(self):
        order = sorted(range(len(self.names)), key=lambda i: self.names[i])
        rank = np.empty(len(self.names), np.int32)
        for r, i in enumerate(order):
            rank[i] = r
        return rank

    def __len__(self):
        return len(self.names)


def _read_table(path, ncols_min):
    import pandas as pd
    opener = gzip.open
    with opener(path, "rb") as f:
        raw = f.read()
    return pd.read_csv(io.BytesIO(raw), sep=r"\s+", header=None, engine="c", dtype=str if ncols_min == 0 else None)


class Contacts:
    """chr1, mid1, chr2, mid2 (int32) + count = int(float(text)) and the float itself (the writer prints %d of it)."""

    def __init__(self, chr1, mid1, chr2, mid2, count, raw_count):
        self.chr1, self.mid1, self.chr2, self.mid2 = chr1, mid1, chr2, mid2
        self.count, self.raw_count = count, raw_count

    def __len__(self):
        return len(self.mid1)

    def take(self, sel):
        return Contacts(self.chr1[sel], self.mid1[sel], self.chr2[sel], self.mid2[sel], self.count[sel], self.raw_count[sel])


def read_contacts(path, chroms):
    import pandas as pd
    df = pd.read_csv(path, sep=r"\s+", header=None, names=["c1", "m1", "c2", "m2", "cc"], compression="gzip", engine="c",
                     dtype={"c1": str, "m1": np.int64, "c2": str, "m2": np.int64, "cc": np.float64})
    raw = df["cc"].values.astype(np.float64)
    both = chroms.intern_column(np.concatenate([df["c1"].values, df["c2"].values]))
    n = len(df)
    return Contacts(both[:n].copy(), df["m1"].values.astype(np.int32), both[n:].copy(), df["m2"].values.astype(np.int32),
                    np.trunc(raw).astype(np.int32), raw)


def read_fragments(path, chroms):
    """-> (chr ids, mids, hits) as int32 arrays in file order."""
    import pandas as pd
    df = pd.read_csv(path, sep=r"\s+", header=None, compression="gzip", engine="c", dtype={0: str})
    return chroms.intern_column(df[0].values), df[2].values.astype(np.int32), df[3].values.astype(np.int32)


def read_bias(path, chroms):
    """-> (chr ids, mids, raw bias values); bounds / NaN / first-occurrence rules are applied by the engine."""
    import pandas as pd
    df = pd.read_csv(path, sep=r"\s+", header=None, names=["c", "m", "b"], compression="gzip", engine="c",
                     dtype={"c": str, "m": np.int64, "b": np.float64})
    return chroms.intern_column(df["c"].values), df["m"].values.astype(np.int32), df["b"].values.astype(np.float64)


def bias_quantiles(bias):
    """The three numbers read_biases logs (fithic/fithic.py:805-815): scipy.stats.mstats.mquantiles defaults
    (alphap = betap = 0.4) over the biases that differ from 1.0."""
    v = np.sort(np.asarray(bias, np.float64)[np.asarray(bias) != 1.0])
    n = len(v)
    out = []
    for prob in (0.05, 0.5, 0.95):
        if n == 0:
            out.append(float("nan"))
            continue
        m = 0.4 + prob * (1.0 - 0.4 - 0.4)
        aleph = n * prob + m
        k = int(np.floor(min(max(aleph, 1), n - 1)))
        gamma = min(max(aleph - k, 0.0), 1.0)
        out.append((1.0 - gamma) * v[k - 1] + gamma * v[min(k, n - 1)])
    return out


def format_significance_rows(names, contacts, emit, p, q, b1, b2, expcc):
    """Text of the .significances.txt file (header + the rows the reference's writer emits,
    fithic/fithic.py:1178-1212): "%s\\t%d\\t%s\\t%d\\t%d\\t%e\\t%e\\t%e\\t%e\\t%f"."""
    out = ["chr1\tfragmentMid1\tchr2\tfragmentMid2\tcontactCount\tp-value\tq-value\tbias1\tbias2\tExpCC\n"]
    c1, m1, c2, m2, rc = contacts.chr1, contacts.mid1, contacts.chr2, contacts.mid2, contacts.raw_count
    fmt = "%s\t%d\t%s\t%d\t%d\t%e\t%e\t%e\t%e\t%f\n"
    for i in np.flatnonzero(emit).tolist():
        out.append(fmt % (names[c1[i]], m1[i], names[c2[i]], m2[i], rc[i], p[i], q[i], b1[i], b2[i], expcc[i]))
    return "".join(out)
