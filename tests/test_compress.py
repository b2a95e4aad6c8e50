"""`gbrs compress`: HIP equivalence classes vs the CPU restatement (exact: ids, structure, counts)."""
import numpy as np
import pytest

from conftest import golden_files, load_golden


def problem(R, H, L, seed, with_count, empty=0):
    from gbrs_amd import synth
    inc = synth.make_em_problem(R=R, H=H, L=L, seed=seed, with_count=with_count, max_count=4)
    if empty:
        rng = np.random.default_rng(seed + 1)
        dead = rng.choice(R, size=empty, replace=False)
        for h in range(H):
            keep = ~np.isin(inc.indices[h], dead)
            loc = np.repeat(np.arange(L), np.diff(inc.indptr[h].astype(np.int64)))[keep]
            inc.indices[h] = inc.indices[h][keep]
            inc.indptr[h] = np.searchsorted(loc, np.arange(L + 1)).astype(np.uint32)
    return inc


def test_compress_oracle_small_known_answer():
    """Hand-checkable case for the restatement itself: rows 0 and 2 identical, row 3 empty."""
    from oracle.compress_oracle import compress
    # H=2, L=3.  hap0: row0->{0,2}, row1->{1}, row2->{0,2};  hap1: row0->{0}, row2->{0}
    indptr = [np.array([0, 2, 3, 5]), np.array([0, 2, 2, 2])]
    indices = [np.array([0, 2, 1, 0, 2]), np.array([0, 2])]
    n, ip, ix, cnt = compress(4, 3, 2, indptr, indices, np.array([1.0, 2.0, 3.0, 5.0]))
    assert n == 3
    np.testing.assert_array_equal(cnt, [4.0, 2.0, 5.0])           # {row0,row2}, {row1}, {empty row3}
    np.testing.assert_array_equal(ip[0], [0, 1, 2, 3])
    np.testing.assert_array_equal(ix[0], [0, 1, 0])
    np.testing.assert_array_equal(ip[1], [0, 1, 1, 1])
    np.testing.assert_array_equal(ix[1], [0])


def _golden_inputs(g):
    H, L, R = int(g["num_haps"]), int(g["num_loci"]), int(g["num_rows"])
    ip = [g[f"indptr{h}"] for h in range(H)]
    ix = [g[f"indices{h}"] for h in range(H)]
    cnt = g["count"] if bool(g["has_count"]) else None
    return R, L, H, ip, ix, cnt


def _doubled(R, L, H, ip, ix, cnt):
    """The reads of two identical input files one after the other."""
    ip2, ix2 = [], []
    for h in range(H):
        cols = np.repeat(np.arange(L, dtype=np.int64), np.diff(ip[h].astype(np.int64)))
        rows = ix[h].astype(np.int64)
        cols, rows = np.concatenate((cols, cols)), np.concatenate((rows, rows + R))
        order = np.lexsort((rows, cols))
        ix2.append(rows[order].astype(np.uint32))
        ip2.append(np.searchsorted(cols[order], np.arange(L + 1)).astype(np.uint32))
    return 2 * R, L, H, ip2, ix2, None if cnt is None else np.concatenate((cnt, cnt))


@pytest.mark.parametrize("path", golden_files("compress"), ids=lambda p: p.split("/")[-1][:-4])
def test_compress_oracle_matches_reference(path):
    """The restatement against what the reference's own compress() produced (oracle/gen_golden.py)."""
    from oracle.compress_oracle import compress
    g = load_golden(path)
    args = _golden_inputs(g)
    if bool(g["two_files"]):
        args = _doubled(*args)
    n, ip, ix, counts = compress(*args)
    assert n == int(g["num_ecs"])
    np.testing.assert_array_equal(counts, g["ec_count"])
    for h in range(args[2]):
        np.testing.assert_array_equal(ip[h], g[f"ec_indptr{h}"])
        np.testing.assert_array_equal(ix[h], g[f"ec_indices{h}"])


@pytest.mark.gpu
@pytest.mark.parametrize("path", golden_files("compress"), ids=lambda p: p.split("/")[-1][:-4])
def test_compress_hip_matches_reference_golden(path, tmp_path):
    """HIP equivalence classes against the reference's: same class order, structure and counts (exact)."""
    from gbrs_amd.alignment import AlignmentPropertyMatrix, load_alignment
    from gbrs_amd.compress import compress, compress_matrix
    g = load_golden(path)
    R, L, H, ip, ix, cnt = _golden_inputs(g)
    apm = AlignmentPropertyMatrix(shape=(L, H, R), indptr=ip, indices=ix, count=cnt,
                                  haplotype_names=[chr(65 + h) for h in range(H)],
                                  locus_names=[f"T{l:05d}" for l in range(L)])
    if bool(g["two_files"]):
        a, b = tmp_path / "a.npz", tmp_path / "b.npz"
        apm.save_npz(str(a)); apm.save_npz(str(b))
        compress([str(a), str(b)], str(tmp_path / "ec.npz"))
        ec = load_alignment(str(tmp_path / "ec.npz"))
    else:
        ec = compress_matrix(apm)
    assert ec.num_reads == int(g["num_ecs"])
    np.testing.assert_array_equal(ec.count, g["ec_count"])
    for h in range(H):
        np.testing.assert_array_equal(ec.indptr[h], g[f"ec_indptr{h}"])
        np.testing.assert_array_equal(ec.indices[h], g[f"ec_indices{h}"])


@pytest.mark.gpu
@pytest.mark.parametrize("R,H,L,seed,cnt,empty", [(3000, 8, 60, 1, False, 0), (2000, 8, 40, 2, True, 50),
                                                  (1500, 2, 30, 3, False, 20), (800, 16, 25, 4, True, 0)])
def test_compress_hip_matches_restatement(R, H, L, seed, cnt, empty, tmp_path):
    from gbrs_amd.alignment import AlignmentPropertyMatrix, load_alignment
    from gbrs_amd.compress import compress, compress_matrix
    from oracle.compress_oracle import compress as ref_compress
    inc = problem(R, H, L, seed, cnt, empty)
    apm = AlignmentPropertyMatrix(shape=(L, H, R), indptr=inc.indptr, indices=inc.indices, count=inc.count,
                                  haplotype_names=inc.hap_names, locus_names=inc.locus_names)
    ec = compress_matrix(apm)
    n, ip, ix, counts = ref_compress(R, L, H, inc.indptr, inc.indices, inc.count)
    assert ec.num_reads == n and n < R
    np.testing.assert_array_equal(ec.count, counts)
    for h in range(H):
        np.testing.assert_array_equal(ec.indptr[h], ip[h])
        np.testing.assert_array_equal(ec.indices[h], ix[h])
    # file interface, two input files = their reads one after another
    a, b = tmp_path / "a.npz", tmp_path / "b.npz"
    apm.save_npz(str(a)); apm.save_npz(str(b))
    compress([str(a), str(b)], str(tmp_path / "ec.npz"))
    ec2 = load_alignment(str(tmp_path / "ec.npz"))
    assert ec2.num_reads == n
    np.testing.assert_array_equal(ec2.count, 2 * counts)
    # EM on the classes == EM on the reads
    from gbrs_amd.em import EMfactory
    eff = inc.effective_length(100)
    out = []
    for m in (apm, ec):
        em = EMfactory(m)
        em.target_lengths = eff
        em.prepare(0.0)
        em.run(model=4, tol=0.0, max_iters=5, verbose=False)
        out.append(em.allelic_expression.copy())
        em.close()
    np.testing.assert_allclose(out[1], out[0], rtol=1e-9, atol=1e-300)
