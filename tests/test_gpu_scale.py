"""GPU tests at larger sizes: BASELINE configs[1] (C2: one chromosome, 40 kb, ~10 M cis rows, 2 passes) against the oracle
row by row, and size-independent properties at a C3-shaped size (histogram == numpy bincount, q == BH(p) recomputed
independently, q monotone in p, idempotent second run)."""
import numpy as np
import pytest

from conftest import bits_equal, max_abs_diff

pytestmark = pytest.mark.gpu
TOL = 1e-10


def _engine_for(genome, res, L, U, n_bins, mode="intraOnly", totals="reference"):
    from fithic_amd.engine import Engine
    eng = Engine(0)
    eng.configure(res, L, U, n_bins=n_bins, mapp_thres=1, mode=mode, totals=totals)
    eng.load_fragments(*genome.fragments(), genome.sort_rank())
    eng.load_bias(*genome.bias_table())
    return eng


def _oracle_inputs(genome, cols, res):
    from oracle import fithic_oracle as fo
    chr_ids = sorted(set(int(c) for c in np.unique(cols[0])))
    local = {c: i for i, c in enumerate(chr_ids)}
    remap = np.vectorize(local.get)(cols[0]).astype(np.int32)
    pairs = fo.Pairs(remap, cols[1], remap, cols[3], cols[4], [genome.names[c] for c in chr_ids])
    frags, bias_dic = [], {}
    for c in range(len(genome)):
        mids = np.arange(genome.n_loci[c], dtype=np.int64) * res + res // 2
        frags += [(genome.names[c], int(m), 1) for m in mids]
        b = genome.bias(c)
        bias_dic[genome.names[c]] = dict(zip(mids.tolist(), np.where((b < 0.5) | (b > 2.0), -1.0, b).tolist()))
    return pairs, frags, bias_dic


def test_c2_one_chromosome_40kb_two_passes_vs_oracle():
    """BASELINE configs[1]: synthetic single chromosome at 40 kb (6 232 loci), no distance bounds, ~10 M rows, 2 passes."""
    import torch
    from fithic_amd import synth
    from oracle import fithic_oracle as fo
    res = 40000
    genome = synth.Genome(res, lengths=[249250621])
    n = genome.n_loci[0]
    amp = synth.solve_amplitude(0.52, 1, n - 1)
    cols_t = synth.cis_contacts(genome, 0, 0, n - 1, amp, device=torch.device("cuda", 0))
    cols = [t.cpu().numpy() for t in cols_t]
    assert 8_000_000 < len(cols[0]) < 13_000_000
    eng = _engine_for(genome, res, 0, float("inf"), 100)
    eng.load_contacts_device([t.data_ptr() for t in cols_t], len(cols[0]))
    pairs, frags, bias_dic = _oracle_inputs(genome, cols, res)
    ref = fo.run(pairs, frags, None, res, n_bins=100, passes=2, mode="intraOnly", bias_dic=bias_dic)
    for r in ref:
        out = eng.run_pass()
        v = eng.fetch(p=True, q=True, expcc=True, bias=True)
        assert [out.stats["inter_count"], out.stats["inter_sum"], out.stats["intra_all_sum"], out.stats["in_range_sum"]] == list(r.sums)
        keys = np.flatnonzero(out.arrays["hist_npairs"] > 0) * res
        assert np.array_equal(keys, r.dist_keys) and np.array_equal(out.arrays["hist_sumcc"][keys // res], r.dist_sumcc)
        assert bits_equal(out.arrays["x"], np.array(r.x)) and bits_equal(out.arrays["table_y"], r.newSplineY)
        assert max_abs_diff(v["p"], r.p) <= TOL and max_abs_diff(v["q"], r.q) <= TOL
        assert bits_equal(v["expcc"], r.expcc) and bits_equal(v["b1"], r.b1)
        assert bits_equal(v["q"], fo.benjamini_hochberg(v["p"], out.info["bh_total_tests"]))
        assert eng.next_pass() == r.n_outlier_lines_total
    eng.close()


def test_histogram_beyond_the_lds_window_is_bit_exact():
    """5 kb, no upper bound: 49 851 distance bins, far more than the 6 144 LDS bins of K1 (global-atomic path)."""
    from fithic_amd import synth
    res = 5000
    genome = synth.Genome(res, lengths=[249250621])
    n = genome.n_loci[0]
    rng = np.random.default_rng(5)
    m = 3_000_001
    i = rng.integers(0, n, m)
    d = np.minimum((rng.pareto(0.7, m) * 30).astype(np.int64), n - 1)
    j = np.minimum(i + d, n - 1)
    cnt = 1 + rng.poisson(2.0, m)
    chrom = np.zeros(m, np.int32)
    eng = _engine_for(genome, res, 10000, float("inf"), 100)
    eng.load_contacts(chrom, i * res + res // 2, chrom, j * res + res // 2, cnt)
    st = eng.pass_stats()
    from fithic_amd import _capi
    hist_cc = eng.ctx.get_array(_capi.A_HIST_SUMCC)
    hist_np = eng.ctx.get_array(_capi.A_HIST_NPAIRS)
    dd = j - i
    rng_mask = dd * res >= 10000
    want_cc = np.bincount(dd[rng_mask], weights=None, minlength=len(hist_cc)) * 0
    np.add.at(want_cc, dd[rng_mask], cnt[rng_mask])
    want_np = np.bincount(dd[rng_mask], minlength=len(hist_np))
    assert dd.max() > 6144 + 2 and len(hist_cc) > 6144
    assert np.array_equal(hist_cc, want_cc[:len(hist_cc)].astype(np.int64)) and np.array_equal(hist_np, want_np[:len(hist_np)])
    assert st.in_range_sum == int(cnt[rng_mask].sum()) and st.intra_all_sum == int(cnt.sum()) and st.max_count == int(cnt.max())
    eng.close()


def test_c3_shaped_properties_at_scale():
    """Three hg19 autosomes at 5 kb, -L 20000 -U 2000000 (3.6e7 rows): properties that need no oracle pass."""
    import torch
    from fithic_amd import synth, _capi
    from oracle import fithic_oracle as fo
    res = 5000
    genome = synth.Genome(res, lengths=synth.HG19_AUTOSOMES[:3])
    amp = synth.solve_amplitude(0.66, 4, 400)
    dev = torch.device("cuda", 0)
    parts = [synth.cis_contacts(genome, c, 4, 400, amp, device=dev, overdispersion=0.5) for c in range(3)]
    cols_t = [torch.cat([p[k] for p in parts]).contiguous() for k in range(5)]
    cols = [t.cpu().numpy() for t in cols_t]
    nrows = len(cols[0])
    eng = _engine_for(genome, res, 4 * res, 400 * res, 100)
    eng.load_contacts_device([t.data_ptr() for t in cols_t], nrows)
    out = eng.run_pass()
    v = eng.fetch()
    # K1: histogram and sums equal numpy's on the same rows
    d = (cols[3].astype(np.int64) - cols[1]) // res
    want = np.zeros(len(out.arrays["hist_sumcc"]), np.int64)
    np.add.at(want, d, cols[4])
    assert np.array_equal(out.arrays["hist_sumcc"], want) and out.stats["in_range_sum"] == int(cols[4].sum())
    # K3: q is exactly the reference's BH of our p (independent numpy recomputation), and monotone in p
    q_ref = fo.benjamini_hochberg(v["p"], out.info["bh_total_tests"])
    assert bits_equal(v["q"], q_ref)
    order = np.argsort(v["p"], kind="stable")
    assert np.all(np.diff(v["q"][order]) >= 0)
    assert 0.0 < float(np.mean(v["q"] < 1.0)) < 0.9
    # K2 on a 1-in-97 sample of the rows against the oracle's bdtrc with the engine's own prior table
    sel = np.arange(0, nrows, 97)
    bias = np.concatenate([np.where((genome.bias(c) < 0.5) | (genome.bias(c) > 2.0), -1.0, genome.bias(c)) for c in range(3)])
    base = np.cumsum([0] + genome.n_loci[:2])
    l1 = base[cols[0][sel]] + cols[1][sel] // res
    l2 = base[cols[2][sel]] + cols[3][sel] // res
    ok = (bias[l1] > 0) & (bias[l2] > 0)
    table_x, table_y = out.arrays["table_x"], out.arrays["table_y"]
    xs = np.sort(out.arrays["x"])
    look = np.minimum(np.maximum((d[sel] * res).astype(np.float64), xs[0]), xs[-1])
    idx = np.minimum(np.searchsorted(table_x.astype(np.float64), look), len(table_x) - 1)
    prior = table_y[idx] * (bias[l1] * bias[l2])
    want_p = np.where(ok, fo.bdtrc(cols[4][sel].astype(np.float64) - 1, float(out.stats["in_range_sum"]), np.where(ok, prior, 0.5)), 1.0)
    assert max_abs_diff(v["p"][sel], want_p) <= TOL
    # idempotence: a second identical pass gives bit-identical results
    eng.run_pass(collect=False)
    v2 = eng.fetch()
    assert bits_equal(v2["p"], v["p"]) and bits_equal(v2["q"], v["q"])
    eng.close()


def test_no_bias_table_path_equals_per_row_evaluation(monkeypatch):
    """Without a bias file K2 evaluates a (distance, count) table and the rows gather (k2_memo_rows / k2_memo_gather):
    p and q must equal the per-row evaluation bit for bit (3.6e6 rows, intra + inter, counts beyond the table cap)."""
    import torch
    from fithic_amd import synth
    from fithic_amd.engine import Engine
    res = 5000
    genome = synth.Genome(res, lengths=synth.HG19_AUTOSOMES[19:22])
    amp = synth.solve_amplitude(0.66, 4, 400)
    dev = torch.device("cuda", 0)
    parts = [synth.cis_contacts(genome, c, 4, 400, amp, device=dev) for c in range(3)]
    cols = [torch.cat([p[k] for p in parts]).cpu().numpy() for k in range(5)]
    rng = np.random.default_rng(4)
    n = len(cols[0])
    cols[4][rng.integers(0, n, 500)] = rng.integers(2000, 90000, 500)            # counts far beyond the table
    m = 20000                                                                    # some inter-chromosomal rows
    mids = [np.arange(genome.n_loci[c]) * res + res // 2 for c in range(3)]
    c1 = rng.integers(0, 3, m)
    c2 = (c1 + rng.integers(1, 3, m)) % 3
    inter = [c1, np.array([rng.choice(mids[c]) for c in c1]), c2, np.array([rng.choice(mids[c]) for c in c2]), 1 + rng.poisson(0.7, m)]
    cols = [np.concatenate([a, b]).astype(np.int32) for a, b in zip(cols, inter)]
    out = {}
    for tag, env in (("table", None), ("rows", "1")):
        if env is None:
            monkeypatch.delenv("FHX_NO_MEMO", raising=False)
        else:
            monkeypatch.setenv("FHX_NO_MEMO", env)
        eng = Engine(0)
        eng.configure(res, 20000, 2000000, n_bins=100, mapp_thres=1, mode="All")
        eng.load_fragments(*genome.fragments(), genome.sort_rank())
        eng.load_contacts(*cols)
        res_passes = []
        for _ in range(2):
            eng.run_pass(collect=False)
            res_passes.append(eng.fetch())
            eng.next_pass()
        out[tag] = res_passes
        eng.close()
    for a, b in zip(out["table"], out["rows"]):
        for key in ("p", "q"):
            same = (a[key].view(np.int64) == b[key].view(np.int64)) | (np.isnan(a[key]) & np.isnan(b[key]))
            assert same.all(), key
    assert np.nanmin(out["table"][0]["p"]) < 1e-6


@pytest.mark.parametrize("n_rows", [5000, 6100, 66000, 2047 * 1024 + 13, 2048 * 1024 + 1, 2049 * 1024 + 5, 5000001])
def test_class_queues_read_as_dense_lists_do_not_depend_on_row_order(n_rows):
    """The class kernels read the queues k2_classify wrote - one shard per workgroup: 5, 6, 65, 2047 or 2048 shards here, the last
    ones ragged - as ONE dense list (QDense: prefix over pairs of shards, pieces from a counter or one range per wave).  p is a
    function of the row alone and q of the multiset of p, so the same rows in reverse order must give the same bits row for row:
    an entry dropped, read twice or taken from a neighbouring shard at any of those seams shows up as a difference; q of the whole
    column is checked against the oracle's BH as well (p against the oracle's Cephes: the parity files)."""
    import torch
    from fithic_amd import synth
    from fithic_amd.engine import Engine
    from oracle import fithic_oracle as fo
    res = 5000
    genome = synth.Genome(res, lengths=synth.HG19_AUTOSOMES[20:22])
    amp = synth.solve_amplitude(0.66, 4, 400)
    dev = torch.device("cuda", 0)
    parts = [synth.cis_contacts(genome, c, 4, 400, amp, device=dev) for c in range(2)]
    cols = [torch.cat([p[k] for p in parts]).cpu().numpy() for k in range(5)]
    assert len(cols[0]) >= n_rows
    pick = np.sort(np.random.default_rng(n_rows).choice(len(cols[0]), n_rows, replace=False))
    cols = [c[pick].astype(np.int32) for c in cols]
    out = []
    for order in (slice(None), slice(None, None, -1)):
        eng = Engine(0)
        eng.configure(res, 20000, 2000000, n_bins=100, mapp_thres=1, mode="intraOnly")
        eng.load_fragments(*genome.fragments(), genome.sort_rank())
        eng.load_bias(*genome.bias_table())
        eng.load_contacts(*[np.ascontiguousarray(c[order]) for c in cols])
        o = eng.run_pass()
        v = eng.fetch()
        classes = eng.ctx.k2_class_rows()
        out.append((o, {k: v[k][order] for k in ("p", "q")}, classes))
        eng.close()
    (o, a, ca), (_, b, cb) = out
    populated = sum(ca[k] > 0 for k in ("pseries", "cf_bcf", "cf_bd", "cf_swapped"))
    assert ca == cb and populated >= (4 if n_rows > 1000000 else 1)      # a thin sample of the map leaves some classes empty
    for key in ("p", "q"):
        same = (a[key].view(np.int64) == b[key].view(np.int64)) | (np.isnan(a[key]) & np.isnan(b[key]))
        assert same.all(), (key, int((~same).sum()))
    q_ref = fo.benjamini_hochberg(a["p"], o.info["bh_total_tests"])
    assert max_abs_diff(a["q"], q_ref) <= TOL


def test_count_homogeneous_waves_equal_the_per_lane_kernel(monkeypatch):
    """The 300-iteration class runs in waves of one (binomial, count) with table-fed iteration constants and a uniform
    renormalisation schedule (k2h_heavy / cf_swapped_uniform); FHX_K2_LEGACY=1 selects round 1's per-lane kernel, which is
    itself bit-exact against the oracle's Cephes.  Both must give the same bits: intra + inter binomials, counts beyond the
    table cap (generic bucket), 2 passes (different totals).  Every instantiation of the wave kernel is compared: 4 rows per lane
    (what a C3-sized input runs), 2 (what an input below 3.2e7 rows runs: this one, by default) and 1 (FHX_K2H_ROWS)."""
    import torch
    from fithic_amd import synth
    from fithic_amd.engine import Engine
    res = 5000
    genome = synth.Genome(res, lengths=synth.HG19_AUTOSOMES[19:22])
    amp = synth.solve_amplitude(0.66, 4, 400)
    dev = torch.device("cuda", 0)
    parts = [synth.cis_contacts(genome, c, 4, 400, amp, device=dev) for c in range(3)]
    cols = [torch.cat([p[k] for p in parts]).cpu().numpy() for k in range(5)]
    rng = np.random.default_rng(11)
    n = len(cols[0])
    cols[4][rng.integers(0, n, 300)] = rng.integers(1000, 5000, 300)            # around and beyond K2H_KCAP = 1023
    m = 200000                                                                   # inter-chromosomal rows: the second binomial
    c1 = rng.integers(0, 3, m)
    c2 = (c1 + rng.integers(1, 3, m)) % 3
    nl = np.array(genome.n_loci)
    inter = [c1, rng.integers(0, nl[c1]) * res + res // 2, c2, rng.integers(0, nl[c2]) * res + res // 2, 1 + rng.poisson(0.7, m)]
    cols = [np.concatenate([a, b]).astype(np.int32) for a, b in zip(cols, inter)]
    out = {}
    for tag in ("uniform", "rows4", "rows2", "rows1", "legacy"):
        monkeypatch.delenv("FHX_K2_LEGACY", raising=False)
        monkeypatch.delenv("FHX_K2H_ROWS", raising=False)
        if tag == "legacy":
            monkeypatch.setenv("FHX_K2_LEGACY", "1")
        elif tag.startswith("rows"):
            monkeypatch.setenv("FHX_K2H_ROWS", tag[4:])
        eng = Engine(0)
        eng.configure(res, 20000, 2000000, n_bins=100, mapp_thres=1, mode="All")
        eng.load_fragments(*genome.fragments(), genome.sort_rank())
        eng.load_bias(*genome.bias_table())
        eng.load_contacts(*cols)
        res_passes = []
        for _ in range(2):
            eng.run_pass(collect=False)
            res_passes.append(eng.fetch())
            _, heavy_rows = eng.ctx.k2_heavy_launch()
            assert heavy_rows > n // 20                                          # the class under test is populated
            by_class = eng.ctx.k2_class_rows()                                   # fhx_k2_class_rows: what each class kernel worked on
            assert by_class["cf_swapped"] == heavy_rows and all(v >= 0 for v in by_class.values())
            assert 0 < by_class["cf_bcf"] + by_class["cf_bd"] + by_class["pseries"] < len(cols[0])
            eng.next_pass()
        out[tag] = res_passes
        eng.close()
    for tag in ("uniform", "rows4", "rows2", "rows1"):
        for a, b in zip(out[tag], out["legacy"]):
            for key in ("p", "q"):
                same = (a[key].view(np.int64) == b[key].view(np.int64)) | (np.isnan(a[key]) & np.isnan(b[key]))
                assert same.all(), (tag, key, int((~same).sum()), float(np.nanmax(np.abs(a[key] - b[key]))))


def _bench_rows(cfg, lengths, device):
    import torch
    import bench
    from fithic_amd import synth
    genome = synth.Genome(cfg["res"], lengths)
    cols, n, n_cis, n_trans = bench.build_rows(synth, torch, cfg, genome, list(range(len(genome))), 0, 1, device)
    return genome, cols, n, n_cis, n_trans


def test_c5_shaped_cis_and_trans_all_mode_vs_oracle():
    """BASELINE configs[4] at reduced size: 1 kb loci, -L 2000 -U 2000000 (1 999 distance values), cis + trans rows from the
    synth-v1 generators, -x All (N = possibleIntraInRangeCount + observedInterAllCount, fithic.py:1138-1139; both binomials):
    every row against the oracle."""
    import torch
    import bench
    from oracle import fithic_oracle as fo
    cfg = dict(bench.CONFIGS["C5"])
    genome, cols_t, n, n_cis, n_trans = _bench_rows(cfg, [5_000_000, 4_000_000, 3_000_000], torch.device("cuda", 0))
    assert n_trans > 300_000 and 6_000_000 < n_cis < 10_000_000
    cols = [t[:n].cpu().numpy() for t in cols_t]
    res = cfg["res"]
    eng = _engine_for(genome, res, cfg["L"], cfg["U"], 100, mode="All")
    eng.load_contacts_device([t.data_ptr() for t in cols_t], n)
    out = eng.run_pass()
    v = eng.fetch(p=True, q=True, expcc=True, bias=True)
    frags, bias_dic = bench._oracle_tables(genome, list(range(len(genome))), res, True)
    pairs = fo.Pairs(cols[0], cols[1], cols[2], cols[3], cols[4], genome.names)
    r = fo.run(pairs, frags, None, res, n_bins=100, passes=1, mode="All", L=cfg["L"], U=cfg["U"], bias_dic=bias_dic)[0]
    assert [out.stats["inter_count"], out.stats["inter_sum"], out.stats["intra_all_sum"], out.stats["in_range_sum"]] == list(r.sums)
    assert out.stats["inter_count"] == n_trans
    assert out.info["bh_total_tests"] == r.N
    assert bits_equal(out.arrays["x"], np.array(r.x)) and bits_equal(out.arrays["table_y"], r.newSplineY)
    assert max_abs_diff(v["p"], r.p) <= TOL and max_abs_diff(v["q"], r.q) <= TOL
    assert bits_equal(v["expcc"], r.expcc)
    inter = cols[0] != cols[2]
    assert np.isnan(v["p"][inter]).sum() == np.isnan(r.p[inter]).sum() > 0          # one discarded bias -> NaN
    assert np.nanmin(v["p"][inter]) < 0.05 and float(np.mean(v["q"] < 1.0)) > 0
    eng.close()


def test_c3_at_full_size():
    """BASELINE configs[2] at its full size - 22 autosomes at 5 kb, 1.48e8 rows, exactly what bench.py times: K1 against
    numpy, p of a 1-in-16 sample of six chromosomes against the oracle's bdtrc with the engine's own fit, q of every row
    against the oracle's Benjamini-Hochberg of the engine's p."""
    import torch
    import bench
    from oracle import run_check
    cfg = dict(bench.CONFIGS["C3"])
    dev = torch.device("cuda", 0)
    genome, cols_t, n, n_cis, _ = _bench_rows(cfg, None, dev)
    assert 1.40e8 < n < 1.56e8 and n == n_cis
    res = cfg["res"]
    eng = _engine_for(genome, res, cfg["L"], cfg["U"], 100)
    eng.load_contacts_device([t.data_ptr() for t in cols_t], n)
    d_idx = ((cols_t[3][:n] - cols_t[1][:n]) // res).to(torch.int64)
    want_np = torch.bincount(d_idx, minlength=512).cpu().numpy()
    want_cc = torch.bincount(d_idx, weights=cols_t[4][:n].to(torch.float64), minlength=512).cpu().numpy().astype(np.int64)
    total_cc = int(cols_t[4][:n].to(torch.int64).sum())
    chroms = [16, 17, 18, 19, 20, 21]
    sel = torch.nonzero(cols_t[0][:n] >= 16).squeeze(1)
    sample = {"rows": sel.cpu().numpy(), "cols": [cols_t[k][:n][sel].cpu().numpy() for k in range(5)], "chroms": chroms}
    del cols_t, d_idx, sel
    torch.cuda.empty_cache()
    out = eng.run_pass()
    assert out.stats["in_range_sum"] == total_cc and out.stats["n_rows"] == n
    hist_np, hist_cc = out.arrays["hist_npairs"], out.arrays["hist_sumcc"]
    assert np.array_equal(hist_np[:512], want_np[:512]) and np.array_equal(hist_cc[:512], want_cc[:512]) and hist_np[512:].sum() == 0
    # the fit at this size against the REAL reference (fixture f14_C3_fit: the reference's own stage functions on this histogram)
    g = run_check.fit_fixture("C3")
    assert run_check.compare_histogram(hist_cc, hist_np, out.stats, g, res) == []
    assert run_check.compare_fit(eng.ctx.get_array, out.info, g) == []
    chk = run_check.check_engine_run(eng, genome, sample, cfg, (out.info, out.stats), True, p_stride=16, fit_fixture_name="C3")
    assert chk["rows_q"] == n and chk["rows_p"] > 1_000_000
    assert chk["nan_pattern_equal"] and chk["max_dp"] <= TOL and chk["max_dq"] == 0.0 and chk["fit_vs_reference"]["bit_identical"], chk
    # the streamed form of the same check (what bench.py and the C5 test below use: p and q read in chunks through device
    # tensors, survivors of the oracle's pruning threshold ranked by the oracle on the host) must give the same verdict
    st = run_check.check_engine_run(eng, genome, sample, cfg, (out.info, out.stats), True, p_stride=16, fit_fixture_name="C3",
                                    torch=torch, chunk_rows=40_000_000)
    assert st["streamed"]["chunk_rows"] == 40_000_000 and st["rows_q"] == n and st["q_rows_pruned_not_one"] == 0
    for k in ("max_dp", "max_dq", "rows_p", "nan_pattern_equal", "p_bit_identical_frac", "ok"):
        assert st[k] == chk[k], (k, st[k], chk[k])
    assert 0 < st["q_rows_ranked_by_oracle"] < n // 10
    eng.close()


def test_c5_at_full_size():
    """BASELINE configs[4] at its full size on one GPU - 22 autosomes at 1 kb, 1.9e9 cis + 1e8 trans rows, -x All, exactly what
    `bench.py --config C5` times: K1 + fit against the real reference's (fixture f14_C5_fit), p of a 1-in-8 sample of the
    smallest chromosome's ~3e7 rows (and the trans rows inside it) against the oracle's Cephes with that table, and q of ALL
    2.0e9 rows: rows at or above the oracle's pruning threshold must be exactly 1, the rest is ranked by the oracle.

    WHICH SEMANTICS.  observedIntraInRangeSum is 7 150 761 687 here, beyond a C int.  The first run is FHX_TOTALS_WIDE (what
    bench.py times: bdtrc on the true total) against the oracle's second mode fho_bdtrc_wide - HIP == the oracle's wide mode,
    NOT HIP == reference: no reference computes that.  The second run is FHX_TOTALS_REFERENCE, the default: scipy narrows the
    total to -1 439 172 905, every in-range cis p-value of the real reference is nan, and the engine's must be too (oracle
    mode fho_bdtrc, pinned at this very total by tests/golden/f15_bdtrc_int_n.npz); the trans rows (total 1.7e8) stay finite."""
    import torch
    import bench
    from fithic_amd import synth
    from oracle import run_check
    cfg = dict(bench.CONFIGS["C5"])
    dev = torch.device("cuda", 0)
    genome = synth.Genome(cfg["res"], synth.HG19_AUTOSOMES)
    mine = list(range(len(genome)))
    cols, n, n_cis, n_trans = bench.build_rows(synth, torch, cfg, genome, mine, 0, 1, dev)
    assert 1.9e9 < n < 2.1e9 and n_trans > 9e7
    eng = _engine_for(genome, cfg["res"], cfg["L"], cfg["U"], 100, mode="All", totals="wide")
    eng.load_contacts_device([t.data_ptr() for t in cols], n)
    sample = bench.build_sample(torch, cols, n, n_cis, genome, mine, cfg, cfg["res"], dev)
    del cols
    torch.cuda.empty_cache()
    out = eng.run_pass(collect=False)
    assert out.info["totals"] == 1 and out.info["totals_narrowed"] == 1 and out.info["bdtrc_n_intra"] == out.stats["in_range_sum"] > 2 ** 32
    chk = run_check.check_engine_run(eng, genome, sample, cfg, (out.info, out.stats), True, p_stride=8, fit_fixture_name="C5", torch=torch)
    assert chk["totals_semantics"] == "wide" and chk["totals_at_or_above_2p31"] == ["observedIntraInRangeSum"]
    assert chk["rows_q"] == n and chk["rows_p"] > 2_000_000 and chk["q_rows_pruned_not_one"] == 0
    assert chk["nan_pattern_equal"] and chk["max_dp"] <= TOL and chk["max_dq"] == 0.0 and chk["fit_vs_reference"]["bit_identical"], chk
    assert chk["ok"] and chk["p_nan_in_sample"] < chk["rows_p"] // 100
    # the reference's own semantics on the same rows: the narrowed total is negative, every in-range cis row is nan
    eng.configure(cfg["res"], cfg["L"], cfg["U"], n_bins=100, mapp_thres=1, mode="All", totals="reference")
    eng.reset_passes()
    out = eng.run_pass(collect=False)
    assert out.info["totals"] == 0 and out.info["bdtrc_n_intra"] == -1439172905 and out.info["bdtrc_n_inter"] == out.stats["inter_sum"]
    chk = run_check.check_engine_run(eng, genome, sample, cfg, (out.info, out.stats), True, p_stride=8, fit_fixture_name="C5", torch=torch)
    assert chk["totals_semantics"] == "reference" and chk["ok"] and chk["nan_pattern_equal"] and chk["max_dq"] == 0.0, chk
    assert chk["p_nan_in_sample"] > chk["rows_p"] // 2
    eng.close()


@pytest.mark.parametrize("name,n_shards", [("C3w", 2), ("C5", 4)])
def test_k1_and_fit_at_full_size_in_row_shards(name, n_shards):
    """C3w (1.06e9 rows, 49 734 distance values: K1's wide LDS window) and C5 (1.9e9 cis + 1e8 trans rows, 1 kb loci) at their
    FULL sizes: K1 over the rows in chromosome shards (its sums and histograms are additive - what the sharded run all-reduces),
    the summed histogram against the fixture's (torch bincount of the same rows, computed without the engine), then fhx_fit on
    it against what the real reference returned on that histogram (fixtures f14_C3w_fit / f14_C5_fit), bit for bit."""
    import torch
    import bench
    from fithic_amd import _capi, synth
    from oracle import run_check
    g = run_check.fit_fixture(name)
    cfg = dict(bench.CONFIGS[name])
    dev = torch.device("cuda", 0)
    genome = synth.Genome(cfg["res"], cfg["lengths"])
    owner = synth.assign_chromosomes(genome, n_shards)
    eng = _engine_for(genome, cfg["res"], cfg["L"], cfg["U"], 100, mode=cfg["mode"])
    total = None
    hist_cc = hist_np = None
    n_all = 0
    for r in range(n_shards):
        mine = [c for c in range(len(genome)) if owner[c] == r]
        cols, n, n_cis, n_trans = bench.build_rows(synth, torch, cfg, genome, mine, r, n_shards, dev)
        eng.load_contacts_device([t.data_ptr() for t in cols], n)
        del cols
        torch.cuda.empty_cache()
        st = eng.ctx.pass_stats().as_dict()
        cc, npairs = eng.ctx.get_array(_capi.A_HIST_SUMCC), eng.ctx.get_array(_capi.A_HIST_NPAIRS)
        if total is None:
            total = dict(st)
            hist_cc, hist_np = np.zeros(max(genome.n_loci) + 2, np.int64), np.zeros(max(genome.n_loci) + 2, np.int64)
        else:
            for k in ("inter_count", "inter_sum", "intra_all_count", "intra_all_sum", "in_range_count", "in_range_sum"):
                total[k] += st[k]
            total["max_count"] = max(total["max_count"], st["max_count"])
        hist_cc[:len(cc)] += cc                         # a shard's histogram is as long as its longest chromosome
        hist_np[:len(npairs)] += npairs
        n_all += n
    assert n_all == int(g["hist_nrows"].sum()) + int(g["inter"][0])
    assert run_check.compare_histogram(hist_cc, hist_np, total, g, cfg["res"]) == []
    stats = _capi.FhxStats()
    for k, v in total.items():
        setattr(stats, k, int(v))
    eng.ctx.set_global_stats(stats, hist_cc, hist_np)
    info = eng.ctx.fit().as_dict()
    assert run_check.compare_fit(eng.ctx.get_array, info, g) == []
    eng.close()


def test_wide_distance_histogram_counters_overflow_into_hbm():
    """More distance values than K1's 12-byte LDS window holds (no -U): the wide kernel keeps 24 576 bins as (u32 sum, 15-bit
    row count) per workgroup and moves overflow to HBM - here one distance takes 1.2e7 rows of count 200 000 (every workgroup's
    sum wraps 2^32 several times, its row count passes 2^15), others sit beyond the window (global atomics), one row has a
    negative count: sums and row counts must equal numpy's, exactly."""
    from fithic_amd import _capi
    from fithic_amd.engine import Engine
    res, n_loci = 1000, 30000
    rng = np.random.default_rng(11)
    f_chr = np.zeros(n_loci, np.int32)
    f_mid = (np.arange(n_loci, dtype=np.int32) * res + res // 2).astype(np.int32)
    f_hits = np.ones(n_loci, np.int32)
    n_big = 12_000_000
    i_small = rng.integers(0, n_loci - 1, 3_000_000)
    j_small = np.minimum(i_small + rng.integers(1, 29000, len(i_small)), n_loci - 1)
    l1 = np.concatenate([np.full(n_big, 10), i_small, [5]]).astype(np.int64)
    l2 = np.concatenate([np.full(n_big, 6510), j_small, [4005]]).astype(np.int64)
    cnt = np.concatenate([np.full(n_big, 200_000), rng.integers(1, 50, len(i_small)), [-3]]).astype(np.int32)
    perm = rng.permutation(len(l1))
    l1, l2, cnt = l1[perm], l2[perm], cnt[perm]
    eng = Engine(0)
    eng.configure(res, 0, float("inf"), n_bins=100, mapp_thres=1, mode="intraOnly")
    eng.load_fragments(f_chr, f_mid, f_hits, np.zeros(1, np.int32))
    z = np.zeros(len(l1), np.int32)
    eng.load_contacts(z, (l1 * res + res // 2).astype(np.int32), z, (l2 * res + res // 2).astype(np.int32), cnt)
    st = eng.ctx.pass_stats()
    got_cc = eng.ctx.get_array(_capi.A_HIST_SUMCC)
    got_np = eng.ctx.get_array(_capi.A_HIST_NPAIRS)
    d = np.abs(l1 - l2)
    want_np = np.bincount(d, minlength=len(got_np))
    want_cc = np.zeros(len(got_cc), np.int64)
    np.add.at(want_cc, d, cnt.astype(np.int64))
    assert len(got_np) > 24576 + 1000                       # bins beyond the wide window exist too
    assert np.array_equal(got_np, want_np[:len(got_np)]) and np.array_equal(got_cc, want_cc[:len(got_cc)])
    assert st.in_range_sum == int(cnt.astype(np.int64).sum()) and want_np[6500] >= n_big and want_cc[6500] > (1 << 41)
    eng.close()
