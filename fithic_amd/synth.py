"""synth-v1: deterministic synthetic Hi-C inputs shaped like the configurations of BASELINE.json.

  * chromosomes: the 22 hg19 autosomes (optionally replicated `replicas` times for weak scaling: chr1_r1, ...)
  * fragments: fixed-size bins [i*res, (i+1)*res), mid = i*res + res/2, last partial bin kept, hits = 1
    (the layout createFitHiCFragments-fixedsize.py writes, fithic/utils/createFitHiCFragments-fixedsize.py:53-78)
  * bias per locus: exp(0.25 z), z ~ N(0,1); then 5 % set to exactly 1.0, 2 % to 0.30 (discarded: < 0.5),
    1 % to 2.50 (discarded: > 2)
  * cis pair (i, i+delta), delta in the in-range index window: count ~ Poisson(A * delta^-1.08 * b_i * b_j), kept
    if count >= 1, 0.1 % of kept pairs multiplied by 4 ("loops"); rows in (chr, mid1, mid2) order
  * A is solved (bias-free expectation, bisection) so that the kept fraction hits the target pair count

Small tables (fragments, bias) are numpy on the host.  The O(#pairs) contact rows are computed with torch tensor
arithmetic on whatever device is given (the GPU in bench.py: torch is plumbing here, the rows are handed to the engine as
raw device pointers).  Randomness is counter-based as SURVEY.md 8d specifies - splitmix64(SEED ^ index), index unique per
(chromosome, locus, delta) - so a row does not depend on the device, on chunking or on how many ranks share the genome;
Poisson counts come from sequential CDF inversion of that uniform.
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


def _splitmix64(torch, x):
    """splitmix64 finaliser on int64 tensors (wrapping arithmetic; logical shifts emulated with masks)."""
    def lsr(v, k):
        return (v >> k) & ((1 << (64 - k)) - 1)
    z = x + (-7046029254386353131)                         # 0x9E3779B97F4A7C15 as a signed 64-bit value
    z = (z ^ lsr(z, 30)) * (-4658895280553007687)          # 0xBF58476D1CE4E5B9
    z = (z ^ lsr(z, 27)) * (-7723592293110705685)          # 0x94D049BB133111EB
    return z ^ lsr(z, 31)


def _uniform(torch, index):
    """u in [0, 1) from the counter: the top 53 bits of splitmix64(SEED ^ index)."""
    z = _splitmix64(torch, index ^ SEED)
    return ((z >> 11) & ((1 << 53) - 1)).to(torch.float64) * (1.0 / (1 << 53))


def _poisson_inverse(torch, lam, u):
    """Poisson(lam) by sequential CDF inversion of the uniform u (float64, vectorised; identical code on host and device)."""
    lam = lam.to(torch.float64)
    p = torch.exp(-lam)
    cdf = p.clone()
    k = torch.zeros(lam.shape, dtype=torch.int64, device=lam.device)
    active = u >= cdf
    steps = 0
    while bool(active.any()):
        k = k + active
        p = torch.where(active, p * lam / k.clamp(min=1).to(torch.float64), p)
        cdf = torch.where(active, cdf + p, cdf)
        active = active & (u >= cdf)
        steps += 1
        if steps > 4000:
            break
    return k


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


def cis_contacts(genome, chrom, lo_idx, hi_idx, amplitude, device="cpu", max_window=None, overdispersion=0.0, hotspots=None):
    """Contact rows of one chromosome as torch int32 tensors (chr1, mid1, chr2, mid2, count) on `device`.
    overdispersion = 0 is synth-v1 (pure Poisson around the model); s > 0 multiplies every pair's rate by
    exp(s*z - s^2/2), z ~ N(0,1) (gamma-Poisson-like counts, the heavier small-p tail real Hi-C maps show); hotspots = (phi, m)
    multiplies the rate of a random fraction phi of the pairs by m and of the rest by (1 - phi*m)/(1 - phi) (mean kept): the
    contrast of pairs inside and outside domains, which is what puts a third to a half of a real map's rows below the
    Benjamini-Hochberg cutoff."""
    import torch
    n = genome.n_loci[chrom]
    hi = min(hi_idx, n - 1)
    if max_window is not None:
        hi = min(hi, lo_idx + max_window - 1)
    lo = max(lo_idx, 0)
    if hi < lo:
        z = torch.zeros(0, dtype=torch.int32, device=device)
        return z, z, z, z, z
    gen = None
    if hotspots is not None and not (0.0 < hotspots[0] < 1.0 and 0.0 < hotspots[0] * hotspots[1] < 1.0):
        raise ValueError("hotspots = (phi, m) needs 0 < phi < 1 and phi * m < 1")
    if overdispersion > 0 or hotspots is not None:          # variant, not synth-v1: extra rate noise from torch's generator
        gen = torch.Generator(device=device)
        gen.manual_seed(SEED * 1000 + chrom)
    b = torch.from_numpy(genome.bias(chrom)).to(device)
    rows = []
    shift = 13 if hi < (1 << 13) else 17                    # delta field of the counter: 13 bits (synth-v1) unless the window is wider
    # delta bands: the Poisson inversion below runs as many steps as the largest rate of its band needs
    bands, d0 = [], lo
    for edge in (16, 64, hi + 1):
        if d0 < min(edge, hi + 1):
            bands.append((d0, min(edge, hi + 1)))
            d0 = min(edge, hi + 1)
    for (da, db) in bands:
        width = db - da
        delta = torch.arange(da, db, device=device, dtype=torch.float64)
        decay = amplitude * torch.where(delta > 0, delta, torch.ones_like(delta)) ** -1.08
        step = max(1, (1 << 23) // width)                   # ~8 M candidates per chunk
        for i0 in range(0, n, step):
            i1 = min(n, i0 + step)
            i = torch.arange(i0, i1, device=device)
            dd = torch.arange(da, db, device=device)
            j = i[:, None] + dd[None, :]
            ok = j < n
            jj = torch.where(ok, j, torch.zeros_like(j))
            lam = (b[i][:, None] * b[jj]) * decay[None, :]
            if overdispersion > 0:
                z = torch.randn(lam.shape, device=device, generator=gen, dtype=torch.float32)
                lam = lam * torch.exp(overdispersion * z - 0.5 * overdispersion * overdispersion).to(lam.dtype)
            if hotspots is not None:
                phi, m = hotspots
                hot = torch.rand(lam.shape, device=device, generator=gen, dtype=torch.float32) < phi
                lam = lam * torch.where(hot, torch.full_like(lam, m), torch.full_like(lam, (1.0 - phi * m) / (1.0 - phi)))
            # counter-based uniforms: splitmix64(seed ^ index), index unique per (chromosome, i, delta, stream)
            index = ((chrom * (1 << 22) + i[:, None]) * (1 << shift) + dd[None, :]) * 2
            cnt = _poisson_inverse(torch, lam, _uniform(torch, index))
            boost = _uniform(torch, index + 1) < 0.001
            cnt = torch.where(boost, cnt * 4, cnt)
            keep = ok & (cnt >= 1)
            ii = i[:, None].expand_as(j)[keep]
            rows.append((ii.to(torch.int32), j[keep].to(torch.int32), cnt[keep].to(torch.int32)))
    i_all = torch.cat([r[0] for r in rows])
    j_all = torch.cat([r[1] for r in rows])
    c_all = torch.cat([r[2] for r in rows])
    if len(bands) > 1:                                      # rows in (chr, mid1, mid2) order, as a sorted contact file has them
        order = torch.argsort(i_all.to(torch.int64) * (1 << 32) + j_all.to(torch.int64))
        i_all, j_all, c_all = i_all[order], j_all[order], c_all[order]
    res = genome.res
    mid1 = i_all * res + res // 2
    mid2 = j_all * res + res // 2
    ch = torch.full_like(mid1, chrom)
    return ch, mid1, ch.clone(), mid2, c_all


TRANS_STREAM = 1 << 60          # counter space of the trans rows (cis counters stay below 2^46)


def trans_contacts(genome, n_trans, t0=0, t1=None, device="cpu", chunk=1 << 24):
    """Inter-chromosomal rows t0 <= t < t1 of the genome-wide list of `n_trans` trans pairs: uniform
    random locus pairs on different chromosomes, count = 1 + Poisson(0.7).  Row t is a function of (SEED, t) alone -
    splitmix64(SEED ^ (TRANS_STREAM + 4 t + k)), k = 0..2 - so any slicing of [0, n_trans) over ranks, chunks or devices
    yields the same rows.  Locus a is uniform over all loci; locus b uniform over the loci of the OTHER chromosomes; the
    pair is written with the lower chromosome index first.  Returned sorted by (chr1, mid1, chr2, mid2) within the slice,
    as torch int32 tensors (chr1, mid1, chr2, mid2, count)."""
    import torch
    t1 = n_trans if t1 is None else t1
    n_loci = torch.tensor(genome.n_loci, dtype=torch.int64, device=device)
    if len(genome) < 2 or t1 <= t0:
        z = torch.zeros(0, dtype=torch.int32, device=device)
        return z, z, z, z, z
    start = torch.cumsum(n_loci, 0) - n_loci                                   # first genome-wide locus of each chromosome
    ends = torch.cumsum(n_loci, 0)
    total = int(n_loci.sum())
    lam = torch.tensor(0.7, dtype=torch.float64, device=device)
    out = []
    for a0 in range(t0, t1, chunk):
        t = torch.arange(a0, min(t1, a0 + chunk), device=device, dtype=torch.int64)
        base = TRANS_STREAM + t * 4
        la = torch.clamp((_uniform(torch, base) * total).to(torch.int64), max=total - 1)
        ca = torch.bucketize(la, ends, right=True)
        rest = total - n_loci[ca]
        r = torch.minimum((_uniform(torch, base + 1) * rest.to(torch.float64)).to(torch.int64), rest - 1)
        lb = torch.where(r < start[ca], r, r + n_loci[ca])                     # skip chromosome ca's range
        cb = torch.bucketize(lb, ends, right=True)
        cnt = 1 + _poisson_inverse(torch, lam.expand(t.shape), _uniform(torch, base + 2))
        ia, ib = la - start[ca], lb - start[cb]
        swap = cb < ca
        c1, i1 = torch.where(swap, cb, ca), torch.where(swap, ib, ia)
        c2, i2 = torch.where(swap, ca, cb), torch.where(swap, ia, ib)
        out.append((c1, i1, c2, i2, cnt))
    c1, i1, c2, i2, cnt = [torch.cat([o[k] for o in out]) for k in range(5)]
    order = torch.argsort(((c1 << 22 | i1) << 30) | (c2 << 22 | i2))          # chromosome ids < 2^8, loci < 2^22
    res = genome.res
    c1, i1, c2, i2, cnt = c1[order], i1[order], c2[order], i2[order], cnt[order]
    return (c1.to(torch.int32), (i1 * res + res // 2).to(torch.int32), c2.to(torch.int32), (i2 * res + res // 2).to(torch.int32),
            cnt.to(torch.int32))


def assign_chromosomes(genome, world_size):
    """Greedy size-balanced chromosome -> rank map (largest first), as SURVEY.md 8e describes."""
    load = [0] * world_size
    owner = [0] * len(genome)
    for c in sorted(range(len(genome)), key=lambda c: -genome.n_loci[c]):
        r = min(range(world_size), key=lambda k: load[k])
        owner[c] = r
        load[r] += genome.n_loci[c]
    return owner
