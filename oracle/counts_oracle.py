"""CPU ORACLE (test infrastructure) for `--report-alignment-counts`.

numpy restatement of AlignmentPropertyMatrix.count_alignments / count_unique_reads
(/root/reference/src/gbrs/emase/AlignmentPropertyMatrix.py:389-440) and of the gene-level
collapse _bundle_inline(reset=True) (:155-188).  Pinned by tests/golden/counts_*.npz, which
oracle/gen_golden.py wrote from the reference's own methods.  All results are sums of EC counts
(integers in practice), so they are exact in float64 whatever the order.
"""
import numpy as np


def alignment_counts(R, L, H, indptr, indices, count=None, locus_group=None, num_out=None):
    """Returns (aln (H x Lo), allele_unique (H x Lo), locus_unique (Lo))."""
    w = np.ones(R) if count is None else np.asarray(count, dtype=np.float64)
    Lo = L if locus_group is None else int(num_out)
    rows, cols, haps = [], [], []
    for h in range(H):
        ptr = np.asarray(indptr[h], dtype=np.int64)
        col = np.repeat(np.arange(L, dtype=np.int64), np.diff(ptr))
        if locus_group is not None:
            col = np.asarray(locus_group, dtype=np.int64)[col]
        rows.append(np.asarray(indices[h], dtype=np.int64))
        cols.append(col)
        haps.append(np.full(len(col), h, dtype=np.int64))
    rows, cols, haps = np.concatenate(rows), np.concatenate(cols), np.concatenate(haps)
    # a locus in no group (locus_group == -1) has no column in grp_conv_mat: its entries drop out (:166-176)
    keep = cols >= 0
    rows, cols, haps = rows[keep], cols[keep], haps[keep]
    # bundling makes (row, gene, hap) a set: several isoforms of one gene collapse to one entry
    ent = np.unique(np.stack((rows, cols, haps), axis=1), axis=0)
    r, c, h = ent[:, 0], ent[:, 1], ent[:, 2]
    nnz_row = np.bincount(r, minlength=R)                       # sum(LOCUS).sum(HAPLOTYPE), :405-407
    pair = np.unique(ent[:, :2], axis=0)
    nloc_row = np.bincount(pair[:, 0], minlength=R)             # nnz per row of sum(HAPLOTYPE), :399-400
    aln = np.zeros((H, Lo)); uniq = np.zeros((H, Lo)); lu = np.zeros(Lo)
    np.add.at(aln, (h, c), w[r])                                # count_alignments, :436-438
    one = nnz_row[r] == 1
    np.add.at(uniq, (h[one], c[one]), w[r[one]])                # ignore_haplotype=False, :404-411, :430-432
    pone = nloc_row[pair[:, 0]] == 1
    np.add.at(lu, pair[pone, 1], w[pair[pone, 0]])              # ignore_haplotype=True, :398-403, :420-428
    return aln, uniq, lu
