"""Degenerate EM inputs (single entry, empty rows, one aligned row, full masks) on an MI355X vs the oracle,
all row orders / layouts.  Test infrastructure; usage: python scripts/tiny_parity.py"""
import sys, os
import numpy as np
sys.path.insert(0, os.getcwd())
from gbrs_amd.engine import EmEngine
from oracle.em_oracle import EMOracle
def case(R, L, H, entries, count=None):
    # entries: list of (r, l, h)
    indptr, indices = [], []
    for h in range(H):
        cols = [[] for _ in range(L)]
        for (r, l, hh) in entries:
            if hh == h: cols[l].append(r)
        ptr = [0]; idx = []
        for l in range(L):
            idx += sorted(cols[l]); ptr.append(len(idx))
        indptr.append(np.array(ptr, dtype=np.uint32)); indices.append(np.array(idx, dtype=np.uint32))
    eff = np.ones((H, L))
    o = EMOracle(R, L, H, indptr, indices, count); o.prepare(0.0, eff); o.run(tol=0.0, max_iters=3)
    for flags in (0, 1, 2, 16):
        e = EmEngine.from_host(R, L, H, indptr, indices, count, eff, flags=flags)
        e.prepare(0.0); e.run(model=4, tol=0.0, max_iters=3)
        np.testing.assert_allclose(e.theta(), o.theta, rtol=1e-9, atol=1e-300)
        e.step(2); e.close()
case(1, 1, 1, [(0, 0, 0)])
case(3, 2, 2, [(0, 0, 0), (0, 1, 1), (2, 1, 0)])            # row 1 empty
case(5, 3, 2, [(4, 2, 1)])                                    # only the last row aligned
case(2, 4, 8, [(0, l, h) for l in range(4) for h in range(8)] + [(1, 3, 7)])
case(70, 3, 1, [(r, r % 3, 0) for r in range(70)], count=np.arange(1, 71, dtype=float))
print("tiny cases ok")
