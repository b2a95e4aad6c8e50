"""CPU ORACLE (test infrastructure, not product code) for the EMASE Model-4 EM.

This is a numpy restatement of the reference's operation sequence for the
`gbrs quantify -M 4` hot path, kept op-for-op (same primitives, same summation
order) so that it reproduces the reference's float64 bits on the same machine.
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import it;
the product path (gbrs_amd.*) never does and fails loudly without its HIP library.

Parity pin: tests/golden/em_*.npz were produced by oracle/gen_golden.py, which runs
the *imported reference* (EMfactory / AlignmentPropertyMatrix from /root/reference)
and this file on the same inputs and asserts bit-equality before writing.

Reference map (all paths relative to /root/reference/src/gbrs/emase/):
  row_totals()        AlignmentPropertyMatrix.py:277-281  sum(LOCUS): per-haplotype
                      csc.sum(axis=1) == scipy csc_matvec with ones == sequential
                      accumulation in column-major entry order -> np.bincount(weights)
                      then :338 dense (R x H).sum(axis=1)
  normalize_rows()    AlignmentPropertyMatrix.py:335-342  normalize_reads(READ)
  column_totals()     AlignmentPropertyMatrix.py:288-298  sum(READ): optional
                      `*= count[indices]`, then csc.sum(axis=0) == np.add.reduceat over
                      the non-empty columns (scipy _minor_reduce)
  prepare()           EMfactory.py:27-111
  em_step()           EMfactory.py:146-159,204-208 (Model 4) + :214-232
  run()               EMfactory.py:234-287
  expected_read_counts()  EMfactory.py:302
  group_sums()        EMfactory.py:140-142  (H x L) @ (L x G) 0/1 matrix
  apply_genotype_mask()   gbrs/emase_utils.py:240-273
"""
from __future__ import annotations

import numpy as np


class EMOracle:
    def __init__(self, num_rows, num_loci, num_haps, indptr, indices, count=None, values=None):
        self.R, self.L, self.H = int(num_rows), int(num_loci), int(num_haps)
        self.indptr = [np.asarray(p).astype(np.int64) for p in indptr]
        self.indices = [np.asarray(i).astype(np.int64) for i in indices]
        self.count = None if count is None else np.asarray(count, dtype=np.float64)
        # stored values of a file saved with incidence_only=False (Sparse3DMatrix.py:84-88); ones otherwise
        self.values = [np.ones(len(i), dtype=np.float64) for i in self.indices] if values is None else \
            [np.array(v, dtype=np.float64) for v in values]
        self.theta = None            # allelic_expression (H x L)
        self.eff_len = None          # target_lengths (H x L) or None
        self.err_history = []
        self.num_iters = 0

    # ---- the three sparse primitives --------------------------------------------------
    def row_totals(self):
        cols = [np.bincount(self.indices[h], weights=self.values[h], minlength=self.R)
                .reshape(self.R, 1) for h in range(self.H)]
        return np.hstack(cols).sum(axis=1)

    def normalize_rows(self):
        den = self.row_totals().ravel()
        for h in range(self.H):
            self.values[h] /= den[self.indices[h]]

    def column_totals(self):
        out = []
        for h in range(self.H):
            v = self.values[h]
            if self.count is not None:
                v = v.copy()
                v *= self.count[self.indices[h]]
            tot = np.zeros(self.L, dtype=np.float64)
            ptr = self.indptr[h]
            nonempty = np.flatnonzero(np.diff(ptr))
            if len(nonempty):
                tot[nonempty] = np.add.reduceat(v, ptr[nonempty])
            out.append(tot.reshape(1, self.L))
        return np.vstack(out)

    # ---- EM driver ---------------------------------------------------------------------
    def prepare(self, pseudocount=0.0, eff_len=None):
        if eff_len is not None:
            eff_len = np.asarray(eff_len, dtype=np.float64)
            if not np.all(eff_len > 0.0):
                raise RuntimeError("There exist transcripts missing length information.")
        self.eff_len = eff_len
        self.normalize_rows()
        self.theta = self.column_totals()
        if self.eff_len is not None:
            self.theta = np.divide(self.theta, self.eff_len)
        if pseudocount > 0.0:
            before = self.theta.sum()
            nzloci = np.nonzero(self.theta)[1]
            self.theta[:, nzloci] += pseudocount
            self.theta *= before / self.theta.sum()

    def e_step(self):
        for h in range(self.H):
            self.values[h] = np.ones(len(self.indices[h]), dtype=np.float64)
        for h in range(self.H):
            vec = self.theta[h, :].ravel()
            self.values[h] *= vec.repeat(np.diff(self.indptr[h]))
        self.normalize_rows()

    def em_step(self):
        self.e_step()
        self.theta = self.column_totals()
        if self.eff_len is not None:
            self.theta = np.divide(self.theta, self.eff_len)

    def run(self, tol=0.001, max_iters=999, on_iter=None):
        old = np.seterr(all="raise")
        np.seterr(under="ignore")
        try:
            self.num_iters = 0
            self.err_history = []
            err_sum = 1000000.0
            target = 1000000.0 * tol
            while err_sum > target and self.num_iters < max_iters:
                prev = self.theta.copy().sum(axis=0)
                prev *= 1000000.0 / prev.sum()
                self.em_step()
                curr = self.theta.copy().sum(axis=0)
                curr *= 1000000.0 / curr.sum()
                err_sum = np.abs(curr - prev).sum()
                self.num_iters += 1
                self.err_history.append(float(err_sum))
                if on_iter is not None:
                    on_iter(self.num_iters, self.theta, err_sum)
        finally:
            np.seterr(**old)
        return self.num_iters

    def expected_read_counts(self):
        return self.column_totals()

    @staticmethod
    def group_sums(mat, groups):
        """(H x L) times the L x G incidence matrix, as scipy csc right-multiplication does
        it: per output column, entries accumulated in ascending locus order."""
        H = mat.shape[0]
        out = np.zeros((H, len(groups)), dtype=np.float64)
        for g, members in enumerate(groups):
            for l in sorted(set(members)):
                out[:, g] += mat[:, l]
        return out

    def apply_genotype_mask(self, gtmask):
        """Zero every entry whose haplotype is not in the called diplotype of its locus
        and drop it from the structure (multiply axis=2 + eliminate_zeros)."""
        for h in range(self.H):
            keep_col = gtmask[h, :] != 0.0
            width = np.diff(self.indptr[h])
            per_entry = np.repeat(keep_col, width)
            kept_per_col = np.where(keep_col, width, 0)
            self.indices[h] = self.indices[h][per_entry]
            self.indptr[h] = np.concatenate(([0], np.cumsum(kept_per_col))).astype(np.int64)
            self.values[h] = self.values[h][per_entry]      # kept entries are multiplied by 1.0: stored values survive


def tpm_report_values(theta):
    """What report_depths(tpm=True) prints for the isoform level: EMfactory.py:352-364.
    Returns the (H+1) x L matrix (haplotypes then total) and the rescaled theta, because the
    reference rescales allelic_expression IN PLACE here (aliasing quirk)."""
    theta = theta * (1000000.0 / theta.sum())
    total = theta.sum(axis=0)
    return np.vstack((theta, total)), theta
