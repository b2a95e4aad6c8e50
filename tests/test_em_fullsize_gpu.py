"""EM paths that only large or skewed inputs reach (needs an MI355X).

* hot loci: a handful of loci that each span many tiles, so every locus has more than HEAVY_SLOTS
  (16) partial-sum slots and the column sums of AlignmentPropertyMatrix.sum(READ)
  (emase/AlignmentPropertyMatrix.py:288-298) go through the one-wavefront-per-locus gather
  (mstep_gather_kernel / gather_kernel) - checked against the oracle;
* BASELINE configs[1] at full size (40M reads x 8 haplotypes x 120k isoforms): size-independent
  properties - conservation of read mass, the tile layout against the plain CSC kernels (a
  different code path and summation order), merged rows against unmerged rows
  (emase/EMfactory.py:214-232).
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

RTOL = 1e-9


def close(a, b, rtol=RTOL):
    np.testing.assert_allclose(a, b, rtol=rtol, atol=1e-300)


@pytest.mark.parametrize("with_count", [False, True], ids=["ones", "count"])
@pytest.mark.parametrize("L", [4, 8])
def test_hot_loci_take_the_heavy_gather(L, with_count):
    from gbrs_amd import _lib, synth
    from gbrs_amd.engine import EmEngine
    from oracle.em_oracle import EMOracle
    inc = synth.make_em_problem(R=400_000, H=8, L=L, seed=77 + L, with_count=with_count, max_count=5)
    eff = inc.effective_length(100)
    o = EMOracle(inc.num_rows, L, 8, inc.indptr, inc.indices, inc.count)
    o.prepare(0.0, eff)
    theta0 = o.theta.copy()
    o.run(tol=0.0, max_iters=4)
    # single engine: E-step tiles -> fused gather + M-step (one wavefront per heavy locus)
    eng = EmEngine.from_host(inc.num_rows, L, 8, inc.indptr, inc.indices, inc.count, eff)
    inf = eng.info()
    assert inf.layout == 1 and inf.num_heavy_loci > 0, (inf.num_heavy_loci, inf.num_light_loci, inf.num_slots)
    assert inf.num_slots > 16 * inf.num_heavy_loci
    eng.prepare(0.0)
    close(eng.theta(), theta0)
    eng.run(model=4, tol=0.0, max_iters=4)
    close(eng.theta(), o.theta)
    close(eng.expected_counts(), o.expected_read_counts())
    total = inc.num_rows if inc.count is None else inc.count.sum()
    assert abs(eng.expected_counts().sum() - total) <= 1e-9 * total
    # sharded building blocks (A handed out, as for an all-reduce): the stand-alone gather kernel
    eng.prepare_partial()
    eng.finish_prepare(0.0)
    for _ in range(4):
        eng.estep_partial()
        eng.finish_step(want_err=True)
    close(eng.theta(), o.theta)
    eng.close()
    # the same rows merged into weighted rows, and through the plain CSC kernels
    for flags in (_lib.GBRS_EM_MERGE_IDENTICAL_ROWS, _lib.GBRS_EM_LAYOUT_CSC):
        e2 = EmEngine.from_host(inc.num_rows, L, 8, inc.indptr, inc.indices, inc.count, eff, flags=flags)
        e2.prepare(0.0)
        e2.run(model=4, tol=0.0, max_iters=4)
        close(e2.theta(), o.theta)
        e2.close()


@pytest.fixture(scope="module")
def c2_sample():
    """BASELINE configs[1] built in HBM by the bench generator (SURVEY 8d recipe)."""
    import torch
    from gbrs_amd import synth, synth_torch
    prob = synth_torch.make_em_problem_device(40_000_000, 8, 120_000, synth.SEED_BASE_EM + 1, "cuda:0")
    yield prob
    del prob
    torch.cuda.empty_cache()


def _engine(prob, flags=0):
    from gbrs_amd.engine import EmEngine
    return EmEngine.from_device(prob["R"], prob["L"], prob["H"], [t.data_ptr() for t in prob["indptr"]],
                                [t.data_ptr() for t in prob["indices"]], None, prob["eff_len"].data_ptr(),
                                device=0, flags=flags)


def test_c2_full_size_properties(c2_sample):
    from gbrs_amd import _lib
    prob = c2_sample
    R = prob["R"]
    eng = _engine(prob)
    inf = eng.info()
    assert inf.layout == 1 and inf.num_rows == R and inf.num_entries == prob["N"]
    assert (inf.num_heavy_loci > 0 or inf.num_locus_sets > 0) and inf.num_long_rows == 0
    eng.prepare(0.0)
    th0 = eng.theta()
    eff = prob["eff_len"].cpu().numpy()
    # prepare: every read spreads one unit of mass over its alignments
    assert abs((th0 * eff).sum() - R) <= 1e-9 * R
    eng.step(3)
    th3, cnt3 = eng.theta(), eng.expected_counts()
    assert np.isfinite(th3).all() and (th3 >= 0).all()
    assert abs(cnt3.sum() - R) <= 1e-9 * R                      # conservation of read mass
    assert abs((th3 * eff).sum() - R) <= 1e-9 * R               # theta' = counts / len
    # the reference's stopping rule on the full sample: a deterministic iteration count > 1
    eng.prepare(0.0)
    n_it, hist = eng.run(model=4, tol=1e-4, max_iters=999)
    assert 1 < n_it < 999 and hist[-1] <= 100.0 and (np.diff(hist) < 0).all()
    assert abs(eng.expected_counts().sum() - R) <= 1e-9 * R
    eng.close()

    # plain CSC kernels (two passes, global atomics) on the same arrays
    ec = _engine(prob, _lib.GBRS_EM_LAYOUT_CSC)
    assert ec.info().layout == 0
    ec.prepare(0.0)
    close(ec.theta(), th0)
    ec.step(3)
    close(ec.theta(), th3)
    close(ec.expected_counts(), cnt3)
    ec.close()

    # identical reads merged into weighted rows: same fixed-point iteration
    em = _engine(prob, _lib.GBRS_EM_MERGE_IDENTICAL_ROWS)
    assert em.info().num_device_rows < R // 4
    em.prepare(0.0)
    close(em.theta(), th0)
    em.step(3)
    close(em.theta(), th3)
    em.close()


def test_c2_full_size_genotype_mask_on_device(c2_sample):
    """`gbrs quantify -G` at BASELINE configs[1] size (gbrs/emase_utils.py:240-273): every gene keeps the two
    haplotypes of a called diplotype, gbrs_em_create_masked_device drops the other columns on the device.  Checked
    against the same restriction carried out with torch on the arrays and fed to the plain CSC kernels (a different
    masking code, layout and summation order), and through conservation of the surviving reads' mass."""
    import torch
    from gbrs_amd import _lib
    prob = c2_sample
    R, L, H = prob["R"], prob["L"], prob["H"]
    rng = np.random.default_rng(5)
    starts = prob["gene_starts"]
    sizes = np.diff(np.concatenate((starts, [L])))
    pair = rng.integers(0, H, size=(len(starts), 2))
    gene_bits = ((1 << pair[:, 0]) | (1 << pair[:, 1])).astype(np.uint32)
    allowed = np.repeat(gene_bits, sizes)
    assert allowed.shape == (L,)
    # the reference's order of operations, with torch: entries of dropped columns leave the arrays
    alive = torch.zeros(R, dtype=torch.bool, device="cuda:0")
    m_indptr, m_indices, n_kept = [], [], 0
    for h in range(H):
        keep_col = torch.from_numpy(((allowed >> h) & 1).astype(bool)).to("cuda:0")
        width = (prob["indptr"][h][1:] - prob["indptr"][h][:-1]).to(torch.int64)
        keep = torch.repeat_interleave(keep_col, width)
        idx = prob["indices"][h][keep].contiguous()
        alive[idx.to(torch.int64)] = True
        m_indices.append(idx)
        ptr = torch.zeros(L + 1, dtype=torch.int64, device="cuda:0")
        ptr[1:] = torch.cumsum(torch.where(keep_col, width, torch.zeros_like(width)), 0)
        m_indptr.append(ptr.to(torch.int32))
        n_kept += int(idx.numel())
        del keep, width
    survivors = int(alive.sum().item())
    assert 0 < survivors < R and n_kept < prob["N"] // 2
    from gbrs_amd.engine import EmEngine
    host_masked = EmEngine.from_device(R, L, H, [t.data_ptr() for t in m_indptr], [t.data_ptr() for t in m_indices],
                                       None, prob["eff_len"].data_ptr(), device=0, flags=_lib.GBRS_EM_LAYOUT_CSC)
    host_masked.prepare(0.0)
    ref0 = host_masked.theta()
    host_masked.step(3)
    ref3, refc = host_masked.theta(), host_masked.expected_counts()
    host_masked.close()
    del m_indptr, m_indices, alive
    torch.cuda.empty_cache()

    eng = EmEngine.from_device(R, L, H, [t.data_ptr() for t in prob["indptr"]], [t.data_ptr() for t in prob["indices"]],
                               None, prob["eff_len"].data_ptr(), device=0, allowed=allowed)
    inf = eng.info()
    assert inf.layout == 1 and inf.num_entries == n_kept and inf.num_device_rows == survivors
    assert inf.retained_build_bytes == 0
    eng.prepare(0.0)
    th0 = eng.theta()
    close(th0, ref0)
    hap_kept = ((allowed[None, :] >> np.arange(H, dtype=np.uint32)[:, None]) & 1).astype(bool)
    assert not th0[~hap_kept].any()                         # a masked (haplotype, locus) never gets abundance
    eng.step(3)
    close(eng.theta(), ref3)
    close(eng.expected_counts(), refc)
    assert abs(eng.expected_counts().sum() - survivors) <= 1e-9 * survivors
    eng.prepare(0.0)
    n_it, hist = eng.run(model=4, tol=1e-4, max_iters=999)
    assert 1 < n_it < 999 and hist[-1] <= 100.0
    assert abs(eng.expected_counts().sum() - survivors) <= 1e-9 * survivors
    eng.close()


def test_c2_full_size_deterministic_mode(c2_sample):
    """40M reads: two deterministic-mode runs are bit-identical, agree with the default path to 1e-9 and stop
    at the same iteration."""
    from gbrs_amd import _lib
    prob = c2_sample
    res = []
    for _ in range(2):
        eng = _engine(prob, _lib.GBRS_EM_DETERMINISTIC)
        eng.prepare(0.0)
        n, hist = eng.run(model=4, tol=1e-4, max_iters=999)
        res.append((n, hist, eng.theta()))
        eng.close()
    assert res[0][0] == res[1][0] and np.array_equal(res[0][1], res[1][1]) and np.array_equal(res[0][2], res[1][2])
    eng = _engine(prob)
    eng.prepare(0.0)
    n, hist = eng.run(model=4, tol=0.0, max_iters=res[0][0])
    close(eng.theta(), res[0][2])
    np.testing.assert_allclose(hist, res[0][1], rtol=1e-7)
    eng.close()


def test_c5_shard_full_size_properties():
    """One GPU's shard of BASELINE configs[4] (200M reads x 16 haplotypes x 200k isoforms over 8 GPUs =
    25M reads per GPU): the 16-haplotype E-step at size - conservation of read mass, the tile layout against
    the plain CSC kernels after 3 iterations, the sharded building blocks (A handed out as for the
    all-reduce) against the fused single-GPU step, and two handles fed the same arrays agreeing to 1e-9."""
    import torch
    from gbrs_amd import _lib, synth, synth_torch
    R, H, L = 25_000_000, 16, 200_000
    prob = synth_torch.make_em_problem_device(R, H, L, synth.SEED_BASE_EM + 4, "cuda:0")
    eng = _engine(prob)
    inf = eng.info()
    assert inf.layout == 1 and inf.num_haps == 16 and inf.num_entries == prob["N"] and inf.num_long_rows == 0
    eng.prepare(0.0)
    th0 = eng.theta()
    eng.step(3)
    th3, cnt3 = eng.theta(), eng.expected_counts()
    assert np.isfinite(th3).all() and (th3 >= 0).all()
    assert abs(cnt3.sum() - R) <= 1e-9 * R
    eff = prob["eff_len"].cpu().numpy()
    assert abs((th3 * eff).sum() - R) <= 1e-9 * R
    # the sharded building blocks on the same handle: E-step into the partial vector, M-step on it
    eng.prepare_partial()
    eng.finish_prepare(0.0)
    close(eng.theta(), th0)
    for _ in range(3):
        eng.estep_partial()
        eng.finish_step(want_err=True)
    close(eng.theta(), th3)
    eng.close()
    ec = _engine(prob, _lib.GBRS_EM_LAYOUT_CSC)
    ec.prepare(0.0)
    close(ec.theta(), th0)
    ec.step(3)
    close(ec.theta(), th3)
    ec.close()
    del prob
    torch.cuda.empty_cache()
