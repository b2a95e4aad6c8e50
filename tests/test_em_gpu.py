"""HIP EM path vs the reference goldens and vs the CPU oracle (needs an MI355X)."""
import numpy as np
import pytest

from conftest import em_case_inputs, em_case_values, golden_files, load_golden

pytestmark = pytest.mark.gpu

# float64 path; the only differences from the reference are summation order (atomics / trees)
# and theta*A/len vs sum(theta/den)/len rounding.  North-star tolerance is 1e-4 relative;
# we hold the kernels to 1e-9.
RTOL = 1e-9


# tiles = packed row tiles in the default stream order; the other row orders stay reachable through
# GBRS_EM_NO_STREAMS (16: sorted order, or interleaved when rows are distinct patterns)
LAYOUTS = {"tiles": dict(), "tiles_merged": dict(merge_identical_rows=True), "csc": dict(csc_layout=True),
           "tiles_sorted": dict(extra_flags=16), "tiles_merged_interleaved": dict(merge_identical_rows=True, extra_flags=16),
           "tiles_deterministic": dict(deterministic=True),
           "tiles_deterministic_merged": dict(deterministic=True, merge_identical_rows=True),
           "tiles_no_locus_sets": dict(extra_flags=512),
           # the layout the headline number runs on: the build's own rule (>= 100 words per id) declines the sets on
           # inputs of this size, GBRS_TUNING_LOCUS_SETS=1 forces them (see the fixture below)
           "tiles_locus_sets_forced": dict(),
           # round 4: locus sets per mask group (the loci of a read that share a mask; what the multi-isoform sample takes),
           # forced on with every candidate kept (the build's own rule wants a set carried by >= 64 reads)
           "tiles_group_sets_forced": dict(),
           # round 4: the E-step on persistent workgroups (GBRS_TUNING_PERSISTENT=1; built, measured, not the default).  A golden
           # has a handful of tiles, so every workgroup would get one (the path with nothing to prefetch); 128-word tiles on
           # two (three) workgroups make each walk many tiles, with the next tile's dictionary and theta fetched under the
           # batch loop - plain loci, and locus sets
           "tiles_persistent_walk": dict(),
           "tiles_persistent_walk_sets": dict(),
           "tiles_persistent_walk_merged": dict(merge_identical_rows=True),
           "tiles_persistent_one_tile_each": dict()}
LAYOUT_ENV = {"tiles_locus_sets_forced": {"GBRS_TUNING_LOCUS_SETS": "1"},
              "tiles_group_sets_forced": {"GBRS_TUNING_LOCUS_SETS": "0", "GBRS_TUNING_GROUP_SETS": "1", "GBRS_TUNING_SET_MIN_ROWS": "1"},
              "tiles_persistent_walk": {"GBRS_TUNING_PERSISTENT": "1", "GBRS_TUNING_PERSISTENT_GROUPS": "2", "GBRS_TUNING_TILE_WORDS": "128"},
              "tiles_persistent_walk_sets": {"GBRS_TUNING_PERSISTENT": "1", "GBRS_TUNING_PERSISTENT_GROUPS": "3",
                                             "GBRS_TUNING_TILE_WORDS": "128", "GBRS_TUNING_LOCUS_SETS": "1"},
              "tiles_persistent_walk_merged": {"GBRS_TUNING_PERSISTENT": "1", "GBRS_TUNING_PERSISTENT_GROUPS": "2",
                                               "GBRS_TUNING_TILE_WORDS": "64"},
              "tiles_persistent_one_tile_each": {"GBRS_TUNING_PERSISTENT": "1"}}


@pytest.fixture(autouse=True)
def _layout_environment(request, monkeypatch):
    """Tuning switches a layout name stands for (read by the library when a handle is created)."""
    cs = getattr(request.node, "callspec", None)
    if cs is not None:
        for k, v in LAYOUT_ENV.get(cs.params.get("layout"), {}).items():
            monkeypatch.setenv(k, v)
    yield


def pack_mask(gtmask):
    """(H x L) 0/1 mask -> uint32[L] with bit h set where gtmask[h, l] != 0 (the `allowed` argument of
    gbrs_em_create_masked)."""
    H = gtmask.shape[0]
    return ((gtmask != 0).astype(np.uint32) << np.arange(H, dtype=np.uint32)[:, None]).sum(axis=0).astype(np.uint32)


def make_factory(g, layout="tiles", mask_on="device"):
    """mask_on: where a golden's `-G` mask is carried out - "device" (gbrs_em_create_masked, what `gbrs quantify -G`
    does) or "host" (the container edits its arrays first, the reference's order of operations)."""
    from gbrs_amd.alignment import AlignmentPropertyMatrix
    from gbrs_amd.em import EMfactory
    R, L, H, indptr, indices, count, eff_len, groups, gtmask = em_case_inputs(g)
    apm = AlignmentPropertyMatrix(shape=(L, H, R), indptr=indptr, indices=indices, count=count,
                                  haplotype_names=[chr(65 + h) for h in range(H)],
                                  locus_names=[f"T{l:07d}" for l in range(L)], values=em_case_values(g))
    apm.groups = groups
    apm.gname = np.array([f"G{i:07d}" for i in range(len(groups))])
    apm.num_groups = len(groups)
    if gtmask is not None:
        if mask_on == "device":
            apm.set_haplotype_mask(pack_mask(gtmask))
        else:
            apm.mask_haplotype_loci(gtmask)
    em = EMfactory(apm, **LAYOUTS[layout])
    em.target_lengths = eff_len
    return em


def _rows_with_one_mask_over_several_loci(g):
    """Number of reads of an em_*.npz fixture that align to more than one locus with the same haplotype mask at each
    (after the fixture's `-G` mask): the reads a locus set replaces by one word."""
    R, L, H, indptr, indices, count, eff_len, groups, gtmask = em_case_inputs(g)
    m = np.zeros((R, L), dtype=np.int64)
    for h in range(H):
        col = np.repeat(np.arange(L), np.diff(indptr[h].astype(np.int64)))
        keep = np.ones(len(col), dtype=bool) if gtmask is None else gtmask[h, col] != 0
        m[indices[h].astype(np.int64)[keep], col[keep]] |= 1 << h
    nl = (m != 0).sum(axis=1)
    same = (np.where(m != 0, m, m.max(axis=1, keepdims=True)) == m.max(axis=1, keepdims=True)).all(axis=1)
    return int(((nl > 1) & same).sum())


def _rows_with_a_mask_group(g):
    """Number of reads of an em_*.npz fixture in which at least two loci carry the same haplotype mask (after the fixture's
    `-G` mask): the reads a mask-group locus set shortens."""
    R, L, H, indptr, indices, count, eff_len, groups, gtmask = em_case_inputs(g)
    m = np.zeros((R, L), dtype=np.int64)
    for h in range(H):
        col = np.repeat(np.arange(L), np.diff(indptr[h].astype(np.int64)))
        keep = np.ones(len(col), dtype=bool) if gtmask is None else gtmask[h, col] != 0
        m[indices[h].astype(np.int64)[keep], col[keep]] |= 1 << h
    n = 0
    for row in m:
        nz = row[row != 0]
        n += len(nz) != len(np.unique(nz))
    return n


def close(a, b, rtol=RTOL):
    np.testing.assert_allclose(a, b, rtol=rtol, atol=1e-300)


@pytest.mark.parametrize("layout", list(LAYOUTS))
@pytest.mark.parametrize("path", golden_files("em"), ids=lambda p: p.split("/")[-1][:-4])
def test_em_matches_reference_golden(path, layout):
    g = load_golden(path)
    pc = float(g["pseudocount"])
    em = make_factory(g, layout)
    expect_layout = 0 if (layout == "csc" or int(g["num_haps"]) > 16) else 1
    em.prepare(pseudocount=pc)
    assert em.info().layout == expect_layout
    if layout.startswith("tiles_persistent_walk") and expect_layout == 1:
        # every workgroup really walks several tiles (the one-haplotype golden's rows merge into a single tile)
        if int(g["num_haps"]) > 1:
            assert em.info().num_tiles >= (2 if "merged" in layout else 4), em.info().num_tiles
    if layout == "tiles_group_sets_forced" and expect_layout == 1 and not bool(g["has_count"]):
        assert (em.info().num_locus_sets > 0) == (_rows_with_a_mask_group(g) > 0)
    if layout in ("tiles_locus_sets_forced", "tiles_persistent_walk_sets") and expect_layout == 1 and not bool(g["has_count"]):
        # unweighted rows (the build never takes sets for weighted ones): the sets must really be there whenever some
        # read aligns to several loci under one mask
        multi = _rows_with_one_mask_over_several_loci(g)
        assert (em.info().num_locus_sets > 0) == (multi > 0), (em.info().num_locus_sets, multi)
    close(em.allelic_expression, g["theta0"])
    # fixed iteration counts: tol=0 never stops early
    done = 0
    for k in (1, 2, 5):
        if f"theta_iter{k}" not in g or k > int(g["num_iters"]):
            continue
        for _ in range(k - done):
            em.update_allelic_expression(model=4)
        done = k
        close(em.allelic_expression, g[f"theta_iter{k}"])
    # full run with the reference's stopping rule: identical iteration count and err sequence
    em.prepare(pseudocount=pc)
    em.run(model=4, tol=float(g["tol"]), max_iters=int(g["max_iters"]), verbose=False)
    assert em.num_iters == int(g["num_iters"])
    np.testing.assert_allclose(em.err_history, g["err_history"], rtol=1e-7)
    close(em.allelic_expression, g["theta_final"])
    close(em.expected_read_counts(), g["expected_counts"])
    close(em.get_allelic_expression(at_group_level=True), g["gene_theta"])
    close(em._group_sums(1), g["gene_counts"])
    em.close()


@pytest.mark.parametrize("layout", ["tiles", "csc", "tiles_deterministic_merged"])
@pytest.mark.parametrize("case", ["h8_mask", "h8_mask_count"])
def test_genotype_mask_on_host_and_on_device_agree(case, layout):
    """gbrs_em_create_masked drops the masked columns on the device; the container can also carry the mask out on
    its arrays before create (gbrs/emase_utils.py:271-273).  Same structure either way: same entry count, same
    reference numbers; bit-identical in the deterministic layout."""
    g = load_golden([p for p in golden_files("em") if p.endswith(f"em_{case}.npz")][0])
    res = []
    for where in ("device", "host"):
        em = make_factory(g, layout, mask_on=where)
        em.prepare(pseudocount=float(g["pseudocount"]))
        n_entries = em.info().num_entries
        em.run(model=4, tol=float(g["tol"]), max_iters=int(g["max_iters"]), verbose=False)
        assert em.num_iters == int(g["num_iters"])
        close(em.allelic_expression, g["theta_final"])
        res.append((n_entries, em.allelic_expression.copy(), em.expected_read_counts()))
        em.close()
    assert res[0][0] == res[1][0] == int((np.concatenate([g[f"indptr{h}"][1:] - g[f"indptr{h}"][:-1]
                                                           for h in range(int(g["num_haps"]))])
                                          * (g["gtmask"].ravel() != 0)).sum())
    if layout == "tiles_deterministic_merged":
        assert np.array_equal(res[0][1], res[1][1]) and np.array_equal(res[0][2], res[1][2])
    else:
        close(res[0][1], res[1][1])


def test_genotype_mask_with_stored_values():
    """A file that stores alignment values (incidence_only=False) under a `-G` mask: the values handed to
    gbrs_em_set_initial_values line up with the unmasked index arrays and move with the surviving columns."""
    from gbrs_amd.alignment import AlignmentPropertyMatrix
    from gbrs_amd.em import EMfactory
    from oracle.em_oracle import EMOracle
    path = [p for p in golden_files("em") if "values" in p][0]
    g = load_golden(path)
    R, L, H, indptr, indices, count, eff_len, groups, _ = em_case_inputs(g)
    rng = np.random.default_rng(11)
    gtmask = (rng.random((H, L)) < 0.6).astype(np.float64)
    out = []
    for where in ("device", "host"):
        apm = AlignmentPropertyMatrix(shape=(L, H, R), indptr=indptr, indices=indices, count=count,
                                      haplotype_names=[chr(65 + h) for h in range(H)],
                                      locus_names=[f"T{l:07d}" for l in range(L)], values=em_case_values(g))
        if where == "device":
            apm.set_haplotype_mask(pack_mask(gtmask))
        else:
            apm.mask_haplotype_loci(gtmask)
        em = EMfactory(apm)
        em.target_lengths = eff_len
        em.prepare(pseudocount=0.0)
        th0 = em.allelic_expression.copy()
        em.run(model=4, tol=0.0, max_iters=3, verbose=False)
        out.append((th0, em.allelic_expression.copy()))
        em.close()
    close(out[0][0], out[1][0])
    close(out[0][1], out[1][1])
    o = EMOracle(R, L, H, indptr, indices, count, values=em_case_values(g))
    o.apply_genotype_mask(gtmask)
    o.prepare(0.0, eff_len)
    close(out[0][0], o.theta)
    o.run(tol=0.0, max_iters=3)
    close(out[0][1], o.theta)


def test_masked_create_argument_checks(hip_lib):
    import ctypes as C
    from gbrs_amd import _lib
    from gbrs_amd.engine import EmEngine
    ip = [np.array([0, 1, 2], dtype=np.uint32)] * 2
    ix = [np.array([0, 1], dtype=np.uint32)] * 2
    with pytest.raises(_lib.GbrsHipError, match="names a haplotype"):
        EmEngine.from_host(2, 2, 2, ip, ix, allowed=np.array([1, 4], dtype=np.uint32))
    # everything masked away: a handle with no entries; prepare gives zeros
    e = EmEngine.from_host(2, 2, 2, ip, ix, allowed=np.zeros(2, dtype=np.uint32))
    assert e.info().num_entries == 0
    e.prepare(0.0)
    assert not e.theta().any()
    e.close()
    # NULL mask = gbrs_em_create
    h = C.c_void_p()
    _lib.check(hip_lib.gbrs_em_create_masked(2, 2, 2, _lib.ptr_table(ip), _lib.ptr_table(ix), None, None, None, 0, 0,
                                             C.byref(h)))
    inf = _lib.EmInfo()
    _lib.check(hip_lib.gbrs_em_info(h, C.byref(inf)))
    assert inf.num_entries == 4 and inf.retained_build_bytes == 0
    hip_lib.gbrs_em_destroy(h)


def test_single_step_after_a_run_that_stopped():
    """EMfactory.update_allelic_expression knows no stopping rule (EMfactory.py:214-232): called after run() has met
    its tolerance it still applies one more EM step (the device's stop flag of the finished run must not turn the
    step's kernels into no-ops)."""
    from oracle.em_oracle import EMOracle
    g = load_golden([p for p in golden_files("em") if p.endswith("em_h8_len.npz")][0])
    R, L, H, indptr, indices, count, eff_len, groups, gtmask = em_case_inputs(g)
    for layout in ("tiles", "csc", "tiles_deterministic"):
        em = make_factory(g, layout)
        em.prepare(pseudocount=0.0)
        em.run(model=4, tol=float(g["tol"]), max_iters=int(g["max_iters"]), verbose=False)
        assert em.num_iters == int(g["num_iters"])
        close(em.allelic_expression, g["theta_final"])
        em.update_allelic_expression(model=4)
        o = EMOracle(R, L, H, indptr, indices, count)
        o.prepare(0.0, eff_len)
        o.theta = np.array(g["theta_final"])
        o.em_step()
        close(em.allelic_expression, o.theta)
        assert not np.allclose(em.allelic_expression, g["theta_final"], rtol=1e-12, atol=0)
        em.close()


def test_em_c1_shape_vs_oracle(monkeypatch):
    """BASELINE config 1 (100k reads / 2 haplotypes / 5k isoforms) against the oracle (locus sets forced on: the build's
    own rule only takes them for sample-sized inputs)."""
    monkeypatch.setenv("GBRS_TUNING_LOCUS_SETS", "1")
    from gbrs_amd import synth
    from gbrs_amd.alignment import AlignmentPropertyMatrix
    from gbrs_amd.em import EMfactory
    from oracle.em_oracle import EMOracle
    inc = synth.make_em_problem(R=100_000, H=2, L=5_000, seed=synth.SEED_BASE_EM)
    eff = inc.effective_length(100)
    o = EMOracle(inc.num_rows, inc.num_loci, inc.num_haps, inc.indptr, inc.indices, None)
    o.prepare(0.0, eff)
    n = o.run(tol=1e-4, max_iters=999)
    apm = AlignmentPropertyMatrix(shape=(inc.num_loci, inc.num_haps, inc.num_rows), indptr=inc.indptr,
                                  indices=inc.indices, haplotype_names=inc.hap_names,
                                  locus_names=inc.locus_names)
    for kw in LAYOUTS.values():
        em = EMfactory(apm, **kw)
        em.target_lengths = eff
        em.prepare(0.0)
        em.run(model=4, tol=1e-4, max_iters=999, verbose=False)
        assert em.num_iters == n
        close(em.allelic_expression, o.theta)
        close(em.expected_read_counts(), o.expected_read_counts())
        em.close()


@pytest.mark.parametrize("tile_words", [64, 1000, 8128, 11008, 16320, -1000])
def test_em_tile_sizes_vs_oracle(tile_words, monkeypatch):
    """The tile size is a layout choice (2,048 to 16,320 words by sample size, em_layout.h): every size, from one batch
    per tile to the capacity of the dictionary sort, gives the oracle's iteration count and values."""
    from gbrs_amd import synth
    from gbrs_amd.alignment import AlignmentPropertyMatrix
    from gbrs_amd.em import EMfactory
    from oracle.em_oracle import EMOracle
    inc = synth.make_em_problem(R=60_000, H=8, L=900, seed=synth.SEED_BASE_EM + 7)
    eff = inc.effective_length(100)
    o = EMOracle(inc.num_rows, inc.num_loci, inc.num_haps, inc.indptr, inc.indices, None)
    o.prepare(0.0, eff)
    n = o.run(tol=1e-4, max_iters=999)
    apm = AlignmentPropertyMatrix(shape=(inc.num_loci, inc.num_haps, inc.num_rows), indptr=inc.indptr,
                                  indices=inc.indices, haplotype_names=inc.hap_names,
                                  locus_names=inc.locus_names)
    monkeypatch.setenv("GBRS_TUNING_TILE_WORDS", str(abs(tile_words)))
    monkeypatch.setenv("GBRS_TUNING_LOCUS_SETS", "1" if tile_words % 128 == 0 else "0")
    if tile_words < 0:                                  # the tiles in locus order instead of largest first
        monkeypatch.setenv("GBRS_TUNING_TILE_ORDER", "0")
    for kw in (LAYOUTS["tiles"], LAYOUTS["tiles_merged"], LAYOUTS["tiles_deterministic"]):
        em = EMfactory(apm, **kw)
        em.target_lengths = eff
        em.prepare(0.0)
        em.run(model=4, tol=1e-4, max_iters=999, verbose=False)
        assert em.num_iters == n
        close(em.allelic_expression, o.theta)
        close(em.expected_read_counts(), o.expected_read_counts())
        em.close()


def test_em_properties_h8():
    """Size-independent properties on a DO-shaped (H=8) problem: conservation of read mass,
    row-permutation invariance, and EC-compression invariance (duplicated rows == count)."""
    from gbrs_amd import synth
    from gbrs_amd.alignment import AlignmentPropertyMatrix
    from gbrs_amd.em import EMfactory
    inc = synth.make_em_problem(R=200_000, H=8, L=3_000, seed=5)
    eff = inc.effective_length(100)

    def solve(inc_, iters=6):
        apm = AlignmentPropertyMatrix(shape=(inc_.num_loci, inc_.num_haps, inc_.num_rows),
                                      indptr=inc_.indptr, indices=inc_.indices, count=inc_.count)
        em = EMfactory(apm)
        em.target_lengths = eff
        em.prepare(0.0)
        em.run(model=4, tol=0.0, max_iters=iters, verbose=False)
        th, cnt = em.allelic_expression.copy(), em.expected_read_counts()
        em.close()
        return th, cnt

    th, cnt = solve(inc)
    assert abs(cnt.sum() - inc.num_rows) < 1e-6 * inc.num_rows
    assert abs((th * eff).sum() - inc.num_rows) < 1e-6 * inc.num_rows
    # permute the rows
    rng = np.random.default_rng(0)
    perm = rng.permutation(inc.num_rows).astype(np.uint32)
    import dataclasses
    ind2 = []
    for h in range(inc.num_haps):
        new = perm[inc.indices[h]]
        # keep rows ascending inside each column as a canonical CSC would
        ptr = inc.indptr[h].astype(np.int64)
        col = np.repeat(np.arange(inc.num_loci), np.diff(ptr))
        order = np.lexsort((new, col))
        ind2.append(new[order])
    th2, _ = solve(dataclasses.replace(inc, indices=ind2))
    close(th2, th, rtol=1e-9)
    # EC compression
    ec = synth.compress_rows(dataclasses.replace(inc, num_rows=20_000,
                                                 indptr=[np.searchsorted(
                                                     np.repeat(np.arange(inc.num_loci), np.diff(inc.indptr[h].astype(np.int64)))[inc.indices[h] < 20_000],
                                                     np.arange(inc.num_loci + 1)).astype(np.uint32) for h in range(inc.num_haps)],
                                                 indices=[inc.indices[h][inc.indices[h] < 20_000] for h in range(inc.num_haps)]))
    sub = dataclasses.replace(inc, num_rows=20_000,
                              indptr=[np.searchsorted(
                                  np.repeat(np.arange(inc.num_loci), np.diff(inc.indptr[h].astype(np.int64)))[inc.indices[h] < 20_000],
                                  np.arange(inc.num_loci + 1)).astype(np.uint32) for h in range(inc.num_haps)],
                              indices=[inc.indices[h][inc.indices[h] < 20_000] for h in range(inc.num_haps)])
    th_sub, _ = solve(sub)
    th_ec, _ = solve(ec)
    assert ec.num_rows < sub.num_rows
    close(th_ec, th_sub, rtol=1e-9)


def test_em_errors():
    from gbrs_amd.alignment import AlignmentPropertyMatrix
    from gbrs_amd.em import EMfactory
    apm = AlignmentPropertyMatrix(shape=(3, 2, 4),
                                  indptr=[np.array([0, 2, 3, 3], dtype=np.uint32), np.array([0, 1, 1, 2], dtype=np.uint32)],
                                  indices=[np.array([0, 1, 2], dtype=np.uint32), np.array([0, 3], dtype=np.uint32)])
    em = EMfactory(apm)
    with pytest.raises(RuntimeError):
        em.run(model=4)                      # not prepared
    em.prepare()
    with pytest.raises(RuntimeError, match="should be 1, 2, 3, or 4"):
        em.run(model=7, verbose=False)
    with pytest.raises(RuntimeError, match="not implemented"):
        em.run(model=2, verbose=False)
    em.run(model=4, tol=0.0, max_iters=3, verbose=False)
    assert em.num_iters == 3
    # theta forced to zero everywhere -> the reference raises FloatingPointError (np.seterr raise)
    em.allelic_expression = np.zeros((2, 3))
    with pytest.raises(FloatingPointError):
        em.run(model=4, tol=0.0, max_iters=2, verbose=False)
    em.close()


def _parse_tsv(text):
    lines = [l.split("\t") for l in text.strip().split("\n")]
    return lines[0], {l[0]: l[1:] for l in lines[1:]}


@pytest.mark.parametrize("name,use_mask,fmt", [("h8_count_len", False, "npz"), ("h8_mask", True, "npz"),
                                               ("h8_count_len", False, "h5"), ("h8_mask", True, "h5"),
                                               ("h8_values", False, "h5"), ("h8_values", False, "npz")])
def test_quantify_cli_files(tmp_path, name, use_mask, fmt):
    """`gbrs quantify` end to end through the CLI: EMASE alignment file (PyTables-layout .h5 through
    libhdf5 on the GPU box, or the .npz mirror; with stored values for the *_values case), group / length /
    genotype files in, the reference's report files out (numbers within 1e-9 of the reference's text)."""
    from gbrs_amd import cli
    from gbrs_amd.alignment import AlignmentPropertyMatrix
    g = load_golden([p for p in golden_files("em") if p.endswith(f"em_{name}.npz")][0])
    R, L, H, indptr, indices, count, eff_len, groups, gtmask = em_case_inputs(g)
    hn = [chr(65 + h) for h in range(H)]
    ln = [f"T{l:07d}" for l in range(L)]
    apm = AlignmentPropertyMatrix(shape=(L, H, R), indptr=indptr, indices=indices, count=count,
                                  haplotype_names=hn, locus_names=ln, values=em_case_values(g))
    aln = tmp_path / f"aln.{fmt}"
    if fmt == "h5":
        from gbrs_amd import emase_h5
        emase_h5._load()                       # libhdf5 must be loadable on the GPU box
        apm.save(str(aln), incidence_only=apm.values is None)
    else:
        apm.save_npz(str(aln))
    grp = tmp_path / "g2t.tsv"
    with open(grp, "w") as fh:
        for i, mem in enumerate(groups):
            fh.write(f"G{i:07d}\t" + "\t".join(ln[m] for m in mem) + "\n")
    lens = tmp_path / "len.tsv"
    with open(lens, "w") as fh:
        for l in range(L):
            for h in hn:
                fh.write(f"{ln[l]}_{h}\t{int(g['raw_length'][l])}\n")
    argv = ["quantify", "-i", str(aln), "-g", str(grp), "-L", str(lens), "-o", str(tmp_path / "out"),
            "-p", str(float(g["pseudocount"])), "-a"]
    suffix = "multiway"
    if use_mask:
        gt = tmp_path / "gt.tsv"
        with open(gt, "w") as fh:
            fh.write("#Gene_ID\tDiplotype\n")
            calls = []
            for i, mem in enumerate(groups):
                hs = np.flatnonzero(gtmask[:, mem[0]])
                calls.append("".join(hn[h] for h in (hs if len(hs) == 2 else [hs[0], hs[0]])))
                fh.write(f"G{i:07d}\t{calls[-1]}\n")
        argv += ["-G", str(gt)]
        suffix = "diploid"
    assert cli.main(argv) == 0
    for key, fname in (("text_isoforms_tpm", "isoforms.tpm"), ("text_isoforms_counts", "isoforms.expected_read_counts"),
                       ("text_genes_tpm", "genes.tpm"), ("text_genes_counts", "genes.expected_read_counts")):
        got_h, got = _parse_tsv(open(tmp_path / f"out.{suffix}.{fname}").read())
        exp_h, exp = _parse_tsv(str(g[key]))
        assert got_h[:len(exp_h)] == exp_h and list(got) == list(exp)
        if use_mask:
            # notes column: the diplotype called for the row's gene (gbrs/emase_utils.py:265-268)
            assert got_h[-1] == "notes"
            call_of_gene = {f"G{i:07d}": calls[i] for i in range(len(groups))}
            call_of = call_of_gene if fname.startswith("genes") else \
                {ln[m]: calls[i] for i, mem in enumerate(groups) for m in mem}
            assert all(got[k][-1] == call_of[k] for k in got)
        for k in exp:
            np.testing.assert_allclose([float(x) for x in got[k][:len(exp[k])]], [float(x) for x in exp[k]],
                                       rtol=1e-9, atol=1e-300)
    # the alignment-count reports are those of the file as it is, whatever `-G` restricted the EM to (the reference
    # reloads the file for them, gbrs/emase_utils.py:318-331)
    from gbrs_amd.alignment import load_alignment
    from gbrs_amd.counts import report_alignment_counts
    fresh = load_alignment(str(aln), grpfile=str(grp))
    for level, grp_wise in (("isoform", False), ("gene", True)):
        report_alignment_counts(fresh, str(tmp_path / f"fresh.{level}"), grp_wise=grp_wise)
        assert open(tmp_path / f"out.{suffix}.{level}s.alignment_counts").read() == open(tmp_path / f"fresh.{level}").read()


def _sharded_worker(rank, world, port, path, out_dir):
    import os, sys
    from conftest import ROOT
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    from gbrs_amd.dist import ShardedEM, shard_rows
    from gbrs_amd.engine import EmEngine
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    g = load_golden(path)
    R, L, H, indptr, indices, count, eff_len, groups, gtmask = em_case_inputs(g)
    r0, r1, ip, ix, cnt = shard_rows(indptr, indices, count, R, rank, world)
    eng = EmEngine.from_host(r1 - r0, L, H, ip, ix, cnt, eff_len, device=0)

    class Dev:
        def __init__(self, ptr, n):
            self.__cuda_array_interface__ = dict(shape=(n,), typestr="<f8", data=(ptr, False), version=2)

    def allreduce(ptr, n):          # gloo on host copies: both ranks share the one GPU of the box
        eng.sync()
        t = torch.as_tensor(Dev(ptr, n), device="cuda:0")
        hcopy = t.cpu()
        dist.all_reduce(hcopy)
        t.copy_(hcopy)
        torch.cuda.synchronize()
    drv = ShardedEM(eng, allreduce)
    drv.prepare(float(g["pseudocount"]))
    n = drv.run(model=4, tol=float(g["tol"]), max_iters=int(g["max_iters"]))
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), theta=eng.theta(), n=n, err=np.array(drv.err_history))
    eng.close()
    dist.barrier()
    dist.destroy_process_group()


def _synth16_case(tmp_path):
    """A 16-haplotype sample without row weights in the goldens' file format (the reference fixtures hold 16 haplotypes
    only with `count`): configs[4]'s kernel instance (tile_estep_kernel<16, false, ...>) under the multi-rank drivers.
    Expected values from the oracle - itself pinned bit-for-bit to the reference on the goldens."""
    from gbrs_amd import synth
    from oracle.em_oracle import EMOracle
    inc = synth.make_em_problem(R=6000, H=16, L=150, seed=41)
    eff = inc.effective_length(100)
    o = EMOracle(inc.num_rows, inc.num_loci, 16, inc.indptr, inc.indices, None)
    o.prepare(0.0, eff)
    n = o.run(tol=1e-3, max_iters=40)
    gp = np.concatenate(([0], np.cumsum([len(m) for m in inc.groups])))
    d = dict(num_haps=16, num_loci=inc.num_loci, num_rows=inc.num_rows, has_count=False, has_len=True, has_mask=False,
             eff_len=eff, group_ptr=gp, group_members=np.concatenate([np.asarray(m) for m in inc.groups]),
             pseudocount=0.0, tol=1e-3, max_iters=40, num_iters=n, theta_final=o.theta, err_history=np.asarray(o.err_history))
    for h in range(16):
        d[f"indptr{h}"], d[f"indices{h}"] = inc.indptr[h], inc.indices[h]
    path = str(tmp_path / "em_synth16.npz")
    np.savez(path, **d)
    return path


def _two_rank_case(name, tmp_path):
    if name == "synth16_unweighted":
        return _synth16_case(tmp_path)
    return [p for p in golden_files("em") if p.endswith(f"em_{name}.npz")][0]


TWO_RANK_CASES = ["h8_count_len", "h8_len", "h16_len", "synth16_unweighted"]


@pytest.mark.parametrize("name", TWO_RANK_CASES)
def test_two_rank_sharded_hip_engines(tmp_path, name):
    """Rows sharded over two processes driving HIP engines (same GPU, gloo all-reduce through host
    copies): identical on both ranks and equal to the unsharded reference."""
    import socket
    import torch.multiprocessing as mp
    path = _two_rank_case(name, tmp_path)
    g = load_golden(path)
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    mp.spawn(_sharded_worker, args=(2, port, path, str(tmp_path)), nprocs=2, join=True)
    a, b = np.load(tmp_path / "rank0.npz"), np.load(tmp_path / "rank1.npz")
    np.testing.assert_array_equal(a["theta"], b["theta"])
    assert int(a["n"]) == int(b["n"]) == int(g["num_iters"])
    close(a["theta"], g["theta_final"])
    np.testing.assert_allclose(a["err"], g["err_history"], rtol=1e-7)


def _pipelined_worker(rank, world, port, path, out_dir):
    import os, sys
    from conftest import ROOT
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    from gbrs_amd.dist import (PipelinedShardedEM, balanced_gene_boundary, rows_are_disjoint, shard_rows,
                               split_at_locus)
    from gbrs_amd.engine import EmEngine
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    g = load_golden(path)
    R, L, H, indptr, indices, count, eff_len, groups, gtmask = em_case_inputs(g)
    l_split = balanced_gene_boundary(indptr, sorted(min(m) for m in groups))
    r0, r1, ip, ix, cnt = shard_rows(indptr, indices, count, R, rank, world)
    (a_ip, a_ix), (b_ip, b_ix) = split_at_locus(ip, ix, l_split)
    assert rows_are_disjoint(a_ix, b_ix, r1 - r0)
    lens = (None, None) if eff_len is None else (np.ascontiguousarray(eff_len[:, :l_split]),
                                                 np.ascontiguousarray(eff_len[:, l_split:]))
    engs = [EmEngine.from_host(r1 - r0, nl, H, p, x, cnt, el, device=0)
            for nl, p, x, el in ((l_split, a_ip, a_ix, lens[0]), (L - l_split, b_ip, b_ix, lens[1]))]
    stream = torch.cuda.current_stream().cuda_stream
    for e in engs:
        e.set_stream(stream)

    class Dev:
        def __init__(self, ptr, n):
            self.__cuda_array_interface__ = dict(shape=(n,), typestr="<f8", data=(ptr, False), version=2)

    def start_allreduce(ptr, n):
        return dist.all_reduce(torch.as_tensor(Dev(ptr, n), device="cuda:0"), async_op=True)
    drv = PipelinedShardedEM(engs[0], engs[1], start_allreduce)
    drv.prepare(0.0)
    drv.step(int(g["num_iters"]))
    theta_fixed = drv.theta()
    # the reference's stopping rule over both ranges, evaluated on the device after every iteration (gbrs_em_pair_check)
    # and read every 8: iterations enqueued past the stopping one must be no-ops
    drv.prepare(0.0)
    n = drv.run(model=4, tol=float(g["tol"]), max_iters=int(g["max_iters"]), check_every=8)
    theta_run = drv.theta()
    drv.step(1)          # by hand, after a run that stopped through the device flags: applied, not swallowed
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), theta=theta_fixed, l_split=l_split, n=n, theta_run=theta_run,
             err=np.array(drv.err_history), theta_plus1=drv.theta(), n_plus1=drv.num_iters)
    for e in engs:
        e.close()
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("name", TWO_RANK_CASES)
def test_two_rank_pipelined_hip_engines(tmp_path, name):
    """Rows sharded over two processes, loci cut at a gene boundary into two HIP engines per process whose
    (gloo) all-reduces are started asynchronously and interleaved with the other half's E-step: after
    the reference's number of iterations theta is the reference's; 8 and 16 haplotypes, with and without row weights."""
    import socket
    import torch.multiprocessing as mp
    path = _two_rank_case(name, tmp_path)
    g = load_golden(path)
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    mp.spawn(_pipelined_worker, args=(2, port, path, str(tmp_path)), nprocs=2, join=True)
    a, b = np.load(tmp_path / "rank0.npz"), np.load(tmp_path / "rank1.npz")
    np.testing.assert_array_equal(a["theta"], b["theta"])
    assert 0 < int(a["l_split"]) < int(g["num_loci"])
    close(a["theta"], g["theta_final"])
    assert int(a["n"]) == int(b["n"]) == int(g["num_iters"])
    np.testing.assert_allclose(a["err"], g["err_history"], rtol=1e-7)
    np.testing.assert_array_equal(a["theta_run"], b["theta_run"])
    close(a["theta_run"], g["theta_final"])
    # run() -> step(1): the oracle one iteration past the stopping one
    from oracle.em_oracle import EMOracle
    R, L, H, indptr, indices, count, eff_len, groups, gtmask = em_case_inputs(g)
    o = EMOracle(R, L, H, indptr, indices, count)
    o.prepare(0.0, eff_len)
    for _ in range(int(g["num_iters"]) + 1):
        o.em_step()
    assert int(a["n_plus1"]) == int(g["num_iters"]) + 1
    close(a["theta_plus1"], o.theta)
    assert not np.allclose(a["theta_plus1"], a["theta_run"], rtol=1e-12, atol=0)


def _random_rows_problem(R, H, L, seed, min_loci, max_loci, with_count):
    """Rows with many loci each and no repetition structure: exercises cold tiles (dictionary cuts),
    multi-word row sums and, above 32 loci (16 for H > 8), the long-row kernel."""
    rng = np.random.default_rng(seed)
    nl = rng.integers(min_loci, max_loci + 1, size=R)
    rows = np.repeat(np.arange(R, dtype=np.int64), nl)
    loci = np.concatenate([rng.choice(L, size=k, replace=False) for k in nl]).astype(np.int64)
    masks = rng.integers(1, 1 << H, size=len(rows), dtype=np.int64)
    indptr, indices = [], []
    for h in range(H):
        sel = (masks >> h) & 1 == 1
        order = np.lexsort((rows[sel], loci[sel]))
        indices.append(rows[sel][order].astype(np.uint32))
        indptr.append(np.searchsorted(loci[sel][order], np.arange(L + 1)).astype(np.uint32))
    count = rng.integers(1, 5, size=R).astype(np.float64) if with_count else None
    eff = np.tile(np.maximum(np.round(rng.lognormal(7.3, 0.6, size=L)) - 99.0, 1.0), (H, 1))
    return indptr, indices, count, np.ascontiguousarray(eff)


def _shared_mask_rows_problem(R, H, L, seed, max_loci, with_count, n_lists=None):
    """Rows whose alignments to several loci mostly share one haplotype mask (what a read that hits the same exon of a
    gene's isoforms looks like): the layout stores such a row as one word on a locus SET.  The locus lists are drawn from
    a pool so that sets repeat; the second half of the loci only ever occur inside multi-locus rows (no slot of their own)."""
    rng = np.random.default_rng(seed)
    n_lists = n_lists or max(R // 20, 4)
    pool = []
    for _ in range(n_lists):
        k = int(rng.integers(1, max_loci + 1))
        lo = rng.choice(L // 2, size=1) if k == 1 else rng.choice(L, size=k, replace=False)
        pool.append(np.sort(lo))
    which = rng.integers(0, n_lists, size=R)
    nl = np.array([len(pool[w]) for w in which])
    rows = np.repeat(np.arange(R, dtype=np.int64), nl)
    loci = np.concatenate([pool[w] for w in which]).astype(np.int64)
    row_mask = rng.integers(1, 1 << H, size=R, dtype=np.int64)
    masks = np.repeat(row_mask, nl)
    odd = rng.random(len(masks)) < 0.15                       # some rows keep a locus with a mask of its own
    masks[odd] = rng.integers(1, 1 << H, size=int(odd.sum()), dtype=np.int64)
    indptr, indices = [], []
    for h in range(H):
        sel = (masks >> h) & 1 == 1
        order = np.lexsort((rows[sel], loci[sel]))
        indices.append(rows[sel][order].astype(np.uint32))
        indptr.append(np.searchsorted(loci[sel][order], np.arange(L + 1)).astype(np.uint32))
    count = rng.integers(1, 5, size=R).astype(np.float64) if with_count else None
    eff = np.tile(np.maximum(np.round(rng.lognormal(7.3, 0.6, size=L)) - 99.0, 1.0), (H, 1))
    return indptr, indices, count, np.ascontiguousarray(eff)


def _mask_group_rows_problem(R, H, L, seed, max_loci, n_lists=None):
    """Reads over several loci whose haplotype masks differ from locus to locus, but in GROUPS: a read's locus list (drawn
    from a pool, so lists repeat) is cut into one to three pieces and every piece carries one random mask - what a read over
    several isoforms of a gene looks like when most isoforms share the read's variants."""
    rng = np.random.default_rng(seed)
    n_lists = n_lists or max(R // 40, 4)
    pool = [np.sort(rng.choice(L, size=int(rng.integers(1, max_loci + 1)), replace=False)) for _ in range(n_lists)]
    which = rng.integers(0, n_lists, size=R)
    rows, loci, masks = [], [], []
    for r in range(R):
        lo = pool[which[r]]
        grp = rng.integers(0, int(rng.integers(1, 4)), size=len(lo))            # piece of every locus
        gm = rng.integers(1, 1 << H, size=3)
        rows.append(np.full(len(lo), r))
        loci.append(lo)
        masks.append(gm[grp])
    rows, loci, masks = np.concatenate(rows), np.concatenate(loci).astype(np.int64), np.concatenate(masks)
    indptr, indices = [], []
    for h in range(H):
        sel = (masks >> h) & 1 == 1
        order = np.lexsort((rows[sel], loci[sel]))
        indices.append(rows[sel][order].astype(np.uint32))
        indptr.append(np.searchsorted(loci[sel][order], np.arange(L + 1)).astype(np.uint32))
    eff = np.tile(np.maximum(np.round(rng.lognormal(7.3, 0.6, size=L)) - 99.0, 1.0), (H, 1))
    return indptr, indices, np.ascontiguousarray(eff)


@pytest.mark.parametrize("max_loci,sets", [(1, "0"), (2, "1"), (14, "1")])
def test_em_sixteen_haplotypes_wide_dictionaries_and_long_rows(max_loci, sets, monkeypatch):
    """Round 4: a 16-haplotype word keeps 3 + 3 bits for the row position fields and 10 bits of dictionary index (before:
    4 + 4 and 8), so that a tile can reference the 300 loci its 78 KB of LDS hold.  Many thin loci, so that tiles really fill
    their dictionaries past 256 entries; rows of up to 14 loci, of which those above 8 now take the long-row path.  Prepare,
    four steps and the expected counts against the oracle, default and deterministic."""
    from gbrs_amd.engine import EmEngine
    from oracle.em_oracle import EMOracle
    monkeypatch.setenv("GBRS_TUNING_GROUP_SETS", sets)
    monkeypatch.setenv("GBRS_TUNING_SET_MIN_ROWS", "8")
    monkeypatch.setenv("GBRS_TUNING_TILE_WORDS", "16000")        # tiles end where their dictionaries are full, not their words
    R, H, L = 120_000, 16, 40_000
    indptr, indices, eff = _mask_group_rows_problem(R, H, L, 977 + max_loci, max_loci, n_lists=R // 2)
    o = EMOracle(R, L, H, indptr, indices, None)
    o.prepare(0.0, eff)
    theta0 = o.theta.copy()
    o.run(tol=0.0, max_iters=4)
    for flags in (0, 32):
        eng = EmEngine.from_host(R, L, H, indptr, indices, None, eff, flags=flags)
        inf = eng.info()
        assert inf.layout == 1
        if flags == 0 and max_loci == 1:
            assert inf.num_slots / inf.num_tiles > 256            # dictionaries wider than the old 8-bit index
        assert (inf.num_long_rows > 0) == (max_loci > 8)
        eng.prepare(0.0)
        close(eng.theta(), theta0)
        eng.step(4)
        close(eng.theta(), o.theta)
        close(eng.expected_counts(), o.expected_read_counts())
        eng.close()


@pytest.mark.parametrize("R,H,L,hi,min_rows", [(6000, 8, 300, 5, 1), (6000, 8, 300, 5, 16), (20000, 8, 3000, 7, 4),
                                               (5000, 16, 400, 6, 2), (3000, 4, 100, 9, 1), (4000, 2, 150, 4, 1)])
def test_em_mask_group_sets_vs_oracle(R, H, L, hi, min_rows, monkeypatch):
    """Round 4: the loci of a read that share a mask become one word on a locus set when enough reads carry that set
    (em_layout.hip, step 3c).  Forced on here with a small threshold; prepare, five steps, the stopping-rule sequence and
    the expected counts against the oracle, default and deterministic layouts; the persistent E-step on the same layout."""
    from gbrs_amd.engine import EmEngine
    from oracle.em_oracle import EMOracle
    monkeypatch.setenv("GBRS_TUNING_GROUP_SETS", "1")
    monkeypatch.setenv("GBRS_TUNING_SET_MIN_ROWS", str(min_rows))
    indptr, indices, eff = _mask_group_rows_problem(R, H, L, 31 + R + H, hi)
    o = EMOracle(R, L, H, indptr, indices, None)
    o.prepare(0.0, eff)
    theta0 = o.theta.copy()
    o.run(tol=0.0, max_iters=5)
    words = {}
    for name, flags, env in (("sets", 0, {}), ("deterministic", 32, {}), ("no_sets", 512, {}),
                             # 16 haplotypes as half-loci on the 8-haplotype kernels (built in round 4, not the default)
                             ("half_loci", 0, {"GBRS_TUNING_HALF_LOCI": "1"}),
                             ("persistent", 0, {"GBRS_TUNING_PERSISTENT": "1", "GBRS_TUNING_PERSISTENT_GROUPS": "2",
                                                "GBRS_TUNING_TILE_WORDS": "256"})):
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        eng = EmEngine.from_host(R, L, H, indptr, indices, None, eff, flags=flags)
        for k in env:
            monkeypatch.delenv(k)
        inf = eng.info()
        assert (inf.num_locus_sets > 0) == (flags != 512)
        words[name] = int(inf.num_device_words)
        eng.prepare(0.0)
        close(eng.theta(), theta0)
        n, hist = eng.run(model=4, tol=0.0, max_iters=5)
        assert n == 5
        np.testing.assert_allclose(hist, o.err_history, rtol=1e-7)
        close(eng.theta(), o.theta)
        close(eng.expected_counts(), o.expected_read_counts())
        eng.close()
    assert words["sets"] < words["no_sets"]                       # the sets really shortened the rows


@pytest.mark.parametrize("R,H,L,hi,cnt,pc", [(6000, 8, 400, 5, False, 0.0),       # sets of 2-5 loci, many rows per set
                                             (3000, 8, 300, 60, True, 0.0),       # sets larger than a row may have words (long rows collapse)
                                             (4000, 16, 500, 6, False, 0.5),      # 16 haplotypes, pseudocount
                                             (2500, 3, 200, 4, True, 0.0),        # generic haplotype count (thread-per-locus M-step)
                                             (30000, 8, 20000, 3, False, 0.0)])   # many sets: dictionary cuts
def test_em_locus_sets_vs_oracle(R, H, L, hi, cnt, pc, monkeypatch):
    """Rows with one mask over several loci become one word on a locus set (include/gbrs_hip.h, GBRS_EM_NO_LOCUS_SETS):
    prepare, steps, the stopping rule and the expected counts against the oracle, for the plain, merged, deterministic
    layouts and with the sets switched off; the building blocks of the sharded path (partial vector handed out and taken
    back) give the same numbers."""
    from gbrs_amd import _lib
    from gbrs_amd.engine import EmEngine
    from oracle.em_oracle import EMOracle
    monkeypatch.setenv("GBRS_TUNING_LOCUS_SETS", "1")            # (the build's own rule wants a sample-sized problem)
    indptr, indices, count, eff = _shared_mask_rows_problem(R, H, L, 77 + R, hi, cnt)
    o = EMOracle(R, L, H, indptr, indices, count)
    o.prepare(pc, eff)
    theta0 = o.theta.copy()
    o.run(tol=0.0, max_iters=5)
    for flags in (0, 1, 32, 512):                                # tiles, tiles + merge, deterministic tiles, no sets
        eng = EmEngine.from_host(R, L, H, indptr, indices, count, eff, flags=flags)
        inf = eng.info()
        # weighted rows (counts given, or identical rows merged) keep one word per (read, locus) pair
        assert (inf.num_locus_sets > 0) == (flags in (0, 32) and count is None)
        eng.prepare(pc)
        close(eng.theta(), theta0)
        n, hist = eng.run(model=4, tol=0.0, max_iters=5)
        assert n == 5
        np.testing.assert_allclose(hist, o.err_history, rtol=1e-7)
        close(eng.theta(), o.theta)
        close(eng.expected_counts(), o.expected_read_counts())
        eng.set_theta(theta0)                                     # theta handed in: the sets' theta are re-summed
        eng.step(5)
        close(eng.theta(), o.theta)
        eng.close()
    if pc == 0.0:
        eng = EmEngine.from_host(R, L, H, indptr, indices, count, eff)
        eng.prepare_partial()
        eng.finish_prepare(0.0)
        close(eng.theta(), theta0)
        for _ in range(5):
            eng.estep_partial()
            eng.finish_step(want_err=False)
        close(eng.theta(), o.theta)
        close(eng.expected_counts(), o.expected_read_counts())
        eng.close()


@pytest.mark.parametrize("R,H,L,lo,hi,cnt", [(4000, 8, 3000, 1, 12, False),      # cold tiles, rows up to 12 words
                                             (600, 8, 400, 20, 70, True),        # long rows (> 32 loci), weighted
                                             (500, 16, 300, 10, 40, False),      # H = 16: long above 16 loci
                                             (20000, 8, 20000, 1, 3, False),     # many distinct lists: dictionary cuts
                                             (3000, 3, 500, 1, 6, False),        # generic haplotype counts: locus-major theta,
                                             (2000, 5, 300, 1, 9, True),         #   integer-built 0/1 doubles
                                             (1500, 2, 200, 1, 4, False),
                                             (800, 1, 100, 1, 3, True)])
def test_em_unstructured_rows_vs_oracle(R, H, L, lo, hi, cnt):
    from gbrs_amd.engine import EmEngine
    from oracle.em_oracle import EMOracle
    indptr, indices, count, eff = _random_rows_problem(R, H, L, 123 + R, lo, hi, cnt)
    o = EMOracle(R, L, H, indptr, indices, count)
    o.prepare(0.0, eff)
    theta0 = o.theta.copy()
    o.run(tol=0.0, max_iters=4)
    for flags in (0, 1, 2, 32):                                  # tiles, tiles + merge, csc, deterministic tiles
        eng = EmEngine.from_host(R, L, H, indptr, indices, count, eff, flags=flags)
        eng.prepare(0.0)
        close(eng.theta(), theta0)
        eng.run(model=4, tol=0.0, max_iters=4)
        close(eng.theta(), o.theta)
        close(eng.expected_counts(), o.expected_read_counts())
        inf = eng.info()
        if flags != 2 and hi > (32 if H <= 8 else 16):
            assert inf.num_long_rows > 0
        eng.close()


def test_create_rejects_bad_inputs():
    """Malformed inputs must come back as errors, never reach a kernel (host and device inputs)."""
    import torch
    from gbrs_amd import _lib
    from gbrs_amd.engine import EmEngine
    ip = [np.array([0, 2, 3], dtype=np.uint32)]
    with pytest.raises(_lib.GbrsHipError, match="row id"):
        EmEngine.from_host(3, 2, 1, ip, [np.array([0, 7, 1], dtype=np.uint32)])
    with pytest.raises(_lib.GbrsHipError, match="non-decreasing"):
        EmEngine.from_host(3, 2, 1, [np.array([0, 3, 2], dtype=np.uint32)], [np.array([0, 1], dtype=np.uint32)])
    with pytest.raises(_lib.GbrsHipError, match="duplicate"):
        EmEngine.from_host(3, 2, 1, ip, [np.array([1, 1, 0], dtype=np.uint32)])
    d_ip = torch.tensor([0, 2, 3], dtype=torch.int32, device="cuda")
    d_ix = torch.tensor([0, 9, 1], dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    for flags in (0, _lib.GBRS_EM_LAYOUT_CSC):
        with pytest.raises(_lib.GbrsHipError, match="row id"):
            EmEngine.from_device(3, 2, 1, [d_ip.data_ptr()], [d_ix.data_ptr()], flags=flags)
    with pytest.raises(_lib.GbrsHipError):
        EmEngine.from_host(3, 2, 40, ip * 40, [np.array([0, 1, 2], dtype=np.uint32)] * 40)     # H > 32


@pytest.mark.parametrize("H,with_count,merge", [(8, False, False), (8, True, False), (8, False, True), (16, False, False),
                                                (2, True, False), (3, False, False)])
def test_deterministic_mode_is_bit_reproducible(H, with_count, merge):
    """GBRS_EM_DETERMINISTIC: fixed-order sums instead of LDS float atomics.  Two handles built from the
    same arrays, and two runs on one handle, give bit-identical theta / expected counts / err history and
    the same iteration count; the result agrees with the default (atomic) path and the oracle to 1e-9."""
    from gbrs_amd import _lib, synth
    from gbrs_amd.engine import EmEngine
    from oracle.em_oracle import EMOracle
    inc = synth.make_em_problem(R=150_000, H=H, L=1_500, seed=200 + H, with_count=with_count, max_count=4)
    eff = inc.effective_length(100)
    flags = _lib.GBRS_EM_DETERMINISTIC | (_lib.GBRS_EM_MERGE_IDENTICAL_ROWS if merge else 0)
    runs = []
    for _ in range(2):
        eng = EmEngine.from_host(inc.num_rows, inc.num_loci, H, inc.indptr, inc.indices, inc.count, eff, flags=flags)
        assert eng.info().layout == 1
        for _ in range(2):
            eng.prepare(0.0)
            n, hist = eng.run(model=4, tol=1e-4, max_iters=60)
            runs.append((n, hist, eng.theta(), eng.expected_counts()))
        eng.close()
    n0, h0, t0, c0 = runs[0]
    for n, hist, th, cn in runs[1:]:
        assert n == n0
        assert np.array_equal(hist, h0) and np.array_equal(th, t0) and np.array_equal(cn, c0)
    ref = EmEngine.from_host(inc.num_rows, inc.num_loci, H, inc.indptr, inc.indices, inc.count, eff,
                             flags=flags & ~_lib.GBRS_EM_DETERMINISTIC)
    # the same number of steps on the default (LDS atomics) path and on the oracle: a run-to-run last-bit
    # difference there must not be able to move the comparison by a whole iteration
    ref.prepare(0.0)
    nr, hr = ref.run(model=4, tol=0.0, max_iters=n0)
    close(ref.theta(), t0)
    np.testing.assert_allclose(hr, h0, rtol=1e-7)
    ref.close()
    o = EMOracle(inc.num_rows, inc.num_loci, H, inc.indptr, inc.indices, inc.count)
    o.prepare(0.0, eff)
    o.run(tol=0.0, max_iters=n0)
    close(t0, o.theta)
    assert h0[-1] <= 100.0 and (n0 == 1 or h0[-2] > 100.0)        # stopped exactly where the rule says


def test_deterministic_mode_long_rows_and_limits():
    """Rows beyond the word capacity go through the serial long-row kernel in deterministic mode; the CSC
    layout (global float atomics) refuses the flag."""
    from gbrs_amd import _lib
    from gbrs_amd.engine import EmEngine
    from oracle.em_oracle import EMOracle
    indptr, indices, count, eff = _random_rows_problem(600, 8, 400, 321, 20, 70, True)
    o = EMOracle(600, 400, 8, indptr, indices, count)
    o.prepare(0.0, eff)
    o.run(tol=0.0, max_iters=4)
    out = []
    for _ in range(2):
        eng = EmEngine.from_host(600, 400, 8, indptr, indices, count, eff, flags=_lib.GBRS_EM_DETERMINISTIC)
        assert eng.info().num_long_rows > 0
        eng.prepare(0.0)
        eng.run(model=4, tol=0.0, max_iters=4)
        out.append((eng.theta(), eng.expected_counts()))
        eng.close()
    assert np.array_equal(out[0][0], out[1][0]) and np.array_equal(out[0][1], out[1][1])
    close(out[0][0], o.theta)
    with pytest.raises(_lib.GbrsHipError, match="DETERMINISTIC"):
        EmEngine.from_host(600, 400, 8, indptr, indices, count, eff,
                           flags=_lib.GBRS_EM_DETERMINISTIC | _lib.GBRS_EM_LAYOUT_CSC)
