"""CPU ORACLE (test infrastructure) for `gbrs interpolate` / `gbrs export`.

numpy restatement of the numeric bodies of gbrs_utils.interpolate
(/root/reference/src/gbrs/gbrs/gbrs_utils.py:664-692; scipy interp1d(kind='linear') semantics:
searchsorted + slope * (x - x_lo) + y_lo) and gbrs_utils.export (:888-927).  Pinned by
tests/golden/postproc_*.npz, written by oracle/gen_golden.py from the reference's own functions.
"""
from itertools import combinations_with_replacement

import numpy as np


def interpolate(x_gene, gamma, x_grid):
    x = np.append([0.0], np.asarray(x_gene, dtype=float))
    x = np.append(x, [x_grid[-1] + 1.0])
    y = np.hstack((gamma[:, 0][:, np.newaxis], gamma))
    y = np.hstack((y, y[:, -1][:, np.newaxis]))
    xq = np.asarray(x_grid, dtype=float)
    if (xq < x[0]).any() or (xq > x[-1]).any():
        raise ValueError("A value in x_new is outside the interpolation range.")
    idx = np.searchsorted(x, xq).clip(1, len(x) - 1)
    lo, hi = idx - 1, idx
    slope = (y[:, hi] - y[:, lo]) / (x[hi] - x[lo])[None, :]
    return slope * (xq - x[lo])[None, :] + y[:, lo]


def dosage(gprob_rows, num_strains):
    geno = list(combinations_with_replacement(range(num_strains), 2))
    conv = np.zeros((len(geno), num_strains))
    for g, (a, b) in enumerate(geno):
        conv[g, a] += 1
        conv[g, b] += 1
    conv *= 0.5
    return np.dot(gprob_rows, conv)
