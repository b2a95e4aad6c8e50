"""numpy stand-in for gbrs_amd.engine.EmEngine used ONLY by the CPU tests of the sharded driver
(gbrs_amd.dist).  Same five-method surface; arithmetic follows oracle/em_oracle.py."""
import numpy as np


class NumpyEngine:
    def __init__(self, R, L, H, indptr, indices, count=None, eff_len=None):
        self.R, self.L, self.H = R, L, H
        self.indptr = [np.asarray(p, dtype=np.int64) for p in indptr]
        self.indices = [np.asarray(i, dtype=np.int64) for i in indices]
        self.count = np.ones(R) if count is None else np.asarray(count, dtype=np.float64)
        self.eff_len = eff_len
        self.theta = np.ones((H, L))
        self.acc = np.zeros((H, L))
        self.counts = np.zeros((H, L))
        self.stopped = False          # set by the pair's stopping rule (gbrs_amd.dist._HostPair): later steps are no-ops
        self.last_totals = None       # per-locus totals (before, after) of the last M-step

    def _estep(self, theta):
        den = np.zeros(self.R)
        for h in range(self.H):
            den += np.bincount(self.indices[h], weights=np.repeat(theta[h], np.diff(self.indptr[h])),
                               minlength=self.R)
        acc = np.zeros((self.H, self.L))
        with np.errstate(divide='ignore', invalid='ignore'):
            w = np.where(den > 0, self.count / den, 0.0)
        for h in range(self.H):
            ptr = self.indptr[h]
            ne = np.flatnonzero(np.diff(ptr))
            if len(ne):
                acc[h, ne] = np.add.reduceat(w[self.indices[h]], ptr[ne])
        self.acc = acc
        return self.acc, acc.size

    def prepare_partial(self):
        return self._estep(np.ones((self.H, self.L)))

    def finish_prepare(self, pseudocount=0.0):
        th = self.acc.copy()
        self.counts = th.copy()
        if self.eff_len is not None:
            th = th / self.eff_len
        if pseudocount > 0:
            before = th.sum()
            nz = np.nonzero(th)[1]
            th[:, nz] += pseudocount
            th *= before / th.sum()
        self.theta = th

    def estep_partial(self):
        return self._estep(self.theta)

    def finish_step(self, want_err=True):
        if self.stopped:
            return 0.0
        self.last_totals = (self.theta.sum(axis=0), None)
        prev = self.theta.sum(axis=0)
        prev = prev * (1e6 / prev.sum())
        self.counts = self.theta * self.acc
        new = self.counts / self.eff_len if self.eff_len is not None else self.counts.copy()
        self.theta = new
        self.last_totals = (self.last_totals[0], new.sum(axis=0))
        cur = new.sum(axis=0)
        cur = cur * (1e6 / cur.sum())
        return float(np.abs(cur - prev).sum())
