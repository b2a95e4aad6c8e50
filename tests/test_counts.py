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
    grp = np.zeros(L, dtype=np.int64)
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
    apm.gname = np.array([f"G{i:07d}" for i in range(len(groups))])
    apm.num_groups = len(groups)
    for level, grp_wise in (("isoforms", False), ("genes", True)):
        a, u, lu, names = alignment_counts(apm, grp_wise=grp_wise)
        np.testing.assert_array_equal(a, g[f"{level}_aln"])          # read-count integers: bit-exact
        np.testing.assert_array_equal(u, g[f"{level}_uniq"])
        np.testing.assert_array_equal(lu, g[f"{level}_locus_uniq"])
        p = tmp_path / f"{level}.tsv"
        report_alignment_counts(apm, str(p), grp_wise=grp_wise)
        assert open(p).read() == str(g[f"text_{level}"])
