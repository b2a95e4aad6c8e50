"""Million-read differential check on an MI355X: HIP EM (default, merged, sorted row order) vs the CPU
oracle with the reference's stopping rule.  Test infrastructure; usage: python scripts/big_parity.py"""
import sys, os, time
import numpy as np
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
from gbrs_amd.engine import EmEngine
from gbrs_amd import synth
from oracle.em_oracle import EMOracle
for (R, H, L, cnt) in [(3_000_000, 8, 60_000, False), (1_500_000, 16, 40_000, True), (4_000_000, 2, 20_000, False)]:
    p = synth.make_em_problem(R=R, H=H, L=L, seed=4242 + H, with_count=cnt, max_count=5)
    t0 = time.time()
    eff = np.ascontiguousarray(p.effective_length())
    o = EMOracle(p.num_rows, p.num_loci, p.num_haps, p.indptr, p.indices, p.count)
    o.prepare(0.0, eff)
    o.run(tol=1e-4, max_iters=12)
    t1 = time.time()
    for flags in (0, 1, 16):
        eng = EmEngine.from_host(p.num_rows, p.num_loci, p.num_haps, p.indptr, p.indices, p.count, eff, flags=flags)
        eng.prepare(0.0)
        n, hist = eng.run(model=4, tol=1e-4, max_iters=12)
        assert n == o.num_iters, (n, o.num_iters)
        np.testing.assert_allclose(hist, o.err_history, rtol=1e-7, atol=1e-6)
        np.testing.assert_allclose(eng.theta(), o.theta, rtol=1e-9, atol=1e-300)
        np.testing.assert_allclose(eng.expected_counts(), o.expected_read_counts(), rtol=1e-9, atol=1e-300)
        eng.close()
    print(f"ok R={R} H={H} L={L} count={cnt} iters={o.num_iters} oracle {t1 - t0:.1f}s", flush=True)
print("big cases ok")
