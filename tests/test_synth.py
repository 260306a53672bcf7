"""synth-v1: counter-based, so rows depend neither on the device nor on chunking / sharding."""
import numpy as np
import pytest

torch = pytest.importorskip("torch")


def test_splitmix64_known_answers():
    from fithic_amd import synth
    z = synth._splitmix64(torch, torch.tensor([0, 1234567], dtype=torch.int64))
    assert int(z[0]) & 0xFFFFFFFFFFFFFFFF == 0xE220A8397B1DCDAF          # first output of splitmix64 seeded with 0
    x = (1234567 + 0x9E3779B97F4A7C15) & 0xFFFFFFFFFFFFFFFF             # plain-Python restatement
    x = ((x ^ (x >> 30)) * 0xBF58476D1CE4E5B9) & 0xFFFFFFFFFFFFFFFF
    x = ((x ^ (x >> 27)) * 0x94D049BB133111EB) & 0xFFFFFFFFFFFFFFFF
    assert int(z[1]) & 0xFFFFFFFFFFFFFFFF == x ^ (x >> 31)


def test_rows_are_sorted_deterministic_and_poisson_like():
    from fithic_amd import synth
    g = synth.Genome(5000, [20_000_000, 9_000_000])
    amp = synth.solve_amplitude(0.66, 4, 400)
    a = synth.cis_contacts(g, 0, 4, 400, amp)
    b = synth.cis_contacts(g, 0, 4, 400, amp)
    assert all(torch.equal(x, y) for x, y in zip(a, b))
    m1, m2, c = a[1].numpy().astype(np.int64), a[3].numpy().astype(np.int64), a[4].numpy()
    assert np.all(np.diff(m1 * (1 << 32) + m2) > 0) and np.all(m2 > m1) and c.min() >= 1
    # another chromosome draws other rows; the kept fraction is near the target the amplitude was solved for
    other = synth.cis_contacts(g, 1, 4, 400, amp)
    assert len(other[0]) != len(a[0])
    frac = len(c) / (g.n_loci[0] * 397 - 397 * 401 / 2)
    assert 0.55 < frac < 0.75
    # inversion: P(count = 0) = exp(-lam); at delta = 400 the bias-free rate is A * 400^-1.08
    d = (m2 - m1) // 5000
    lam400 = amp * 400.0 ** -1.08
    kept400 = np.sum(d == 400) / (g.n_loci[0] - 400)
    assert abs(kept400 - (1 - np.exp(-lam400))) < 0.05
