"""`--report-alignment-counts`: oracle vs reference goldens (CPU) and HIP vs goldens (GPU, exact)."""
import numpy as np
import pytest

from conftest import golden_files, load_golden


def inputs(g):
    H, L, R = int(g["num_haps"]), int(g["num_loci"]), int(g["num_rows"])
    indptr = [g[f"indptr{h}"] for h in range(H)]
    indices = [g[f"indices{h}"] for h in range(H)]
    count = g["count"] if bool(g["has_count"]) else None
    gp, gm = g["group_ptr"], g["group_members"]
    groups = [list(gm[gp[i]:gp[i + 1]]) for i in range(len(gp) - 1)]
    return R, L, H, indptr, indices, count, groups


@pytest.mark.parametrize("path", golden_files("counts"), ids=lambda p: p.split("/")[-1][:-4])
def test_counts_oracle_matches_reference(path):
    from oracle.counts_oracle import alignment_counts
    g = load_golden(path)
    R, L, H, indptr, indices, count, groups = inputs(g)
    a, u, lu = alignment_counts(R, L, H, indptr, indices, count)
    np.testing.assert_array_equal(a, g["isoforms_aln"])
    np.testing.assert_array_equal(u, g["isoforms_uniq"])
    np.testing.assert_array_equal(lu, g["isoforms_locus_uniq"])
    grp = np.full(L, -1, dtype=np.int64)
    for i, m in enumerate(groups):
        grp[m] = i
    a, u, lu = alignment_counts(R, L, H, indptr, indices, count, grp, len(groups))
    np.testing.assert_array_equal(a, g["genes_aln"])
    np.testing.assert_array_equal(u, g["genes_uniq"])
    np.testing.assert_array_equal(lu, g["genes_locus_uniq"])


@pytest.mark.gpu
@pytest.mark.parametrize("path", golden_files("counts"), ids=lambda p: p.split("/")[-1][:-4])
def test_counts_hip_bit_exact(path, tmp_path):
    from gbrs_amd.alignment import AlignmentPropertyMatrix
    from gbrs_amd.counts import alignment_counts, report_alignment_counts
    g = load_golden(path)
    R, L, H, indptr, indices, count, groups = inputs(g)
    apm = AlignmentPropertyMatrix(shape=(L, H, R), indptr=indptr, indices=indices, count=count,
                                  haplotype_names=[chr(65 + h) for h in range(H)],
                                  locus_names=[f"T{l:07d}" for l in range(L)])
    apm.groups = groups
    # gene names as the reference printed them (a fixture may leave genes out of the group file)
    apm.gname = np.array([ln.split("\t")[0] for ln in str(g["text_genes"]).strip().split("\n")[1:]])
    assert len(apm.gname) == len(groups)
    apm.num_groups = len(groups)
    for level, grp_wise in (("isoforms", False), ("genes", True)):
        a, u, lu, names = alignment_counts(apm, grp_wise=grp_wise)
        np.testing.assert_array_equal(a, g[f"{level}_aln"])          # read-count integers: bit-exact
        np.testing.assert_array_equal(u, g[f"{level}_uniq"])
        np.testing.assert_array_equal(lu, g[f"{level}_locus_uniq"])
        p = tmp_path / f"{level}.tsv"
        report_alignment_counts(apm, str(p), grp_wise=grp_wise)
        assert open(p).read() == str(g[f"text_{level}"])
    # one upload, both levels (gbrs_counts_create / _get), in either order and twice: the workspace is re-used
    from gbrs_amd.counts import AlignmentCounter
    with AlignmentCounter(apm) as counter:
        for level, grp_wise in (("genes", True), ("isoforms", False), ("genes", True)):
            a, u, lu, names = counter.counts(grp_wise)
            np.testing.assert_array_equal(a, g[f"{level}_aln"])
            np.testing.assert_array_equal(u, g[f"{level}_uniq"])
            np.testing.assert_array_equal(lu, g[f"{level}_locus_uniq"])
            p = tmp_path / f"{level}.counter.tsv"
            report_alignment_counts(apm, str(p), grp_wise=grp_wise, counter=counter)
            assert open(p).read() == str(g[f"text_{level}"])
    with pytest.raises(RuntimeError, match="closed"):
        counter.counts()


@pytest.mark.gpu
def test_counts_and_compress_reject_bad_row_ids():
    """A row id >= num_rows must come back as an error before any kernel indexes per-row storage."""
    from gbrs_amd import _lib
    from gbrs_amd.alignment import AlignmentPropertyMatrix
    from gbrs_amd.compress import compress_matrix
    from gbrs_amd.counts import alignment_counts
    apm = AlignmentPropertyMatrix(shape=(2, 1, 3), indptr=[np.array([0, 2, 3], dtype=np.uint32)],
                                  indices=[np.array([0, 4_000_000_000, 1], dtype=np.uint32)],
                                  haplotype_names=["A"], locus_names=["t0", "t1"])
    with pytest.raises(_lib.GbrsHipError, match="row id"):
        alignment_counts(apm)
    with pytest.raises(_lib.GbrsHipError, match="row id"):
        compress_matrix(apm)


@pytest.mark.gpu
def test_counts_locus_in_two_groups_is_rejected():
    from gbrs_amd.alignment import AlignmentPropertyMatrix
    from gbrs_amd.counts import alignment_counts
    apm = AlignmentPropertyMatrix(shape=(2, 1, 3), indptr=[np.array([0, 2, 3], dtype=np.uint32)],
                                  indices=[np.array([0, 2, 1], dtype=np.uint32)],
                                  haplotype_names=["A"], locus_names=["t0", "t1"])
    apm.groups, apm.gname, apm.num_groups = [[0, 1], [1]], np.array(["g0", "g1"]), 2
    with pytest.raises(RuntimeError, match="more than one group"):
        alignment_counts(apm, grp_wise=True)
