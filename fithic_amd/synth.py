"""synth-v1: deterministic synthetic Hi-C inputs shaped like the configurations of BASELINE.json.

  * chromosomes: the 22 hg19 autosomes (optionally replicated `replicas` times for weak scaling: chr1_r1, ...)
  * fragments: fixed-size bins [i*res, (i+1)*res), mid = i*res + res/2, last partial bin kept, hits = 1
    (the layout createFitHiCFragments-fixedsize.py writes, fithic/utils/createFitHiCFragments-fixedsize.py:53-78)
  * bias per locus: exp(0.25 z), z ~ N(0,1); then 5 % set to exactly 1.0, 2 % to 0.30 (discarded: < 0.5),
    1 % to 2.50 (discarded: > 2)
  * cis pair (i, i+delta), delta in the in-range index window: count ~ Poisson(A * delta^-1.08 * b_i * b_j), kept
    if count >= 1, 0.1 % of kept pairs multiplied by 4 ("loops"); rows in (chr, mid1, mid2) order
  * A is solved (bias-free expectation, bisection) so that the kept fraction hits the target pair count

Small tables (fragments, bias) are numpy on the host.  The O(#pairs) contact rows are drawn with torch on whatever
device is given (the GPU in bench.py: torch is plumbing here, the rows are handed to the engine as raw device
pointers); every chromosome has its own seed, so a shard does not depend on how many ranks share the genome.
"""
import math

import numpy as np

SEED = 20260928
HG19_AUTOSOMES = [249250621, 243199373, 198022430, 191154276, 180915260, 171115067, 159138663, 146364022, 141213431,
                  135534747, 135006516, 133851895, 115169878, 107349540, 102531392, 90354753, 81195210, 78077248,
                  59128983, 63025520, 48129895, 51304566]


class Genome:
    def __init__(self, resolution, lengths=None, replicas=1):
        self.res = int(resolution)
        base = list(HG19_AUTOSOMES if lengths is None else lengths)
        self.names, self.lengths = [], []
        for r in range(replicas):
            for i, ln in enumerate(base):
                self.names.append("chr%d" % (i + 1) if r == 0 else "chr%d_r%d" % (i + 1, r))
                self.lengths.append(ln)
        self.n_loci = [(ln + self.res - 1) // self.res for ln in self.lengths]

    def __len__(self):
        return len(self.names)

    def sort_rank(self):
        order = sorted(range(len(self.names)), key=lambda i: self.names[i])
        rank = np.empty(len(order), np.int32)
        for r, i in enumerate(order):
            rank[i] = r
        return rank

    def fragments(self):
        """(chr ids, mids, hits) for every locus, int32."""
        ch = np.concatenate([np.full(n, c, np.int32) for c, n in enumerate(self.n_loci)])
        mid = np.concatenate([np.arange(n, dtype=np.int64) * self.res + self.res // 2 for n in self.n_loci]).astype(np.int32)
        return ch, mid, np.ones(len(ch), np.int32)

    def bias(self, chrom):
        """Raw bias values of one chromosome's loci (float64), deterministic per chromosome."""
        rng = np.random.default_rng([SEED, 1, chrom])
        n = self.n_loci[chrom]
        b = np.exp(0.25 * rng.standard_normal(n))
        u = rng.random(n)
        b[u < 0.05] = 1.0
        b[(u >= 0.05) & (u < 0.07)] = 0.30
        b[(u >= 0.07) & (u < 0.08)] = 2.50
        return b

    def bias_table(self):
        ch, mid, _ = self.fragments()
        return ch, mid, np.concatenate([self.bias(c) for c in range(len(self))])


def solve_amplitude(keep_fraction, lo_idx, hi_idx):
    """A with mean_delta(1 - exp(-A delta^-1.08)) = keep_fraction over delta in [lo_idx, hi_idx] (delta >= 1)."""
    d = np.arange(max(lo_idx, 1), hi_idx + 1, dtype=np.float64) ** -1.08
    lo, hi = 1e-6, 1e9
    for _ in range(200):
        mid = math.sqrt(lo * hi)
        if np.mean(1.0 - np.exp(-mid * d)) < keep_fraction:
            lo = mid
        else:
            hi = mid
    return math.sqrt(lo * hi)


def cis_contacts(genome, chrom, lo_idx, hi_idx, amplitude, device="cpu", max_window=None, overdispersion=0.0):
    """Contact rows of one chromosome as torch int32 tensors (chr1, mid1, chr2, mid2, count) on `device`.
    overdispersion = 0 is synth-v1 (pure Poisson around the model); s > 0 multiplies every pair's rate by
    exp(s*z - s^2/2), z ~ N(0,1) (gamma-Poisson-like counts, the heavier small-p tail real Hi-C maps show)."""
    import torch
    n = genome.n_loci[chrom]
    hi = min(hi_idx, n - 1)
    if max_window is not None:
        hi = min(hi, lo_idx + max_window - 1)
    lo = max(lo_idx, 0)
    if hi < lo:
        z = torch.zeros(0, dtype=torch.int32, device=device)
        return z, z, z, z, z
    gen = torch.Generator(device=device)
    gen.manual_seed(SEED * 1000 + chrom)
    b = torch.from_numpy(genome.bias(chrom)).to(device)
    delta = torch.arange(lo, hi + 1, device=device, dtype=torch.float64)
    decay = amplitude * torch.where(delta > 0, delta, torch.ones_like(delta)) ** -1.08
    rows = []
    step = max(1, (1 << 24) // (hi - lo + 1))               # ~16 M candidates per chunk
    for i0 in range(0, n, step):
        i1 = min(n, i0 + step)
        i = torch.arange(i0, i1, device=device)
        j = i[:, None] + torch.arange(lo, hi + 1, device=device)[None, :]
        ok = j < n
        jj = torch.where(ok, j, torch.zeros_like(j))
        lam = (b[i][:, None] * b[jj]) * decay[None, :]
        if overdispersion > 0:
            z = torch.randn(lam.shape, device=device, generator=gen, dtype=torch.float32)
            lam = lam * torch.exp(overdispersion * z - 0.5 * overdispersion * overdispersion).to(lam.dtype)
        cnt = torch.poisson(lam.to(torch.float32), generator=gen).to(torch.int32)
        boost = torch.rand(cnt.shape, device=device, generator=gen) < 0.001
        cnt = torch.where(boost, cnt * 4, cnt)
        keep = ok & (cnt >= 1)
        ii = i[:, None].expand_as(j)[keep]
        rows.append((ii.to(torch.int32), j[keep].to(torch.int32), cnt[keep]))
    i_all = torch.cat([r[0] for r in rows])
    j_all = torch.cat([r[1] for r in rows])
    c_all = torch.cat([r[2] for r in rows])
    res = genome.res
    mid1 = i_all * res + res // 2
    mid2 = j_all * res + res // 2
    ch = torch.full_like(mid1, chrom)
    return ch, mid1, ch.clone(), mid2, c_all


def assign_chromosomes(genome, world_size):
    """Greedy size-balanced chromosome -> rank map (largest first), as SURVEY.md 8e describes."""
    load = [0] * world_size
    owner = [0] * len(genome)
    for c in sorted(range(len(genome)), key=lambda c: -genome.n_loci[c]):
        r = min(range(world_size), key=lambda k: load[k])
        owner[c] = r
        load[r] += genome.n_loci[c]
    return owner
