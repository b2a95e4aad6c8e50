#!/usr/bin/env python3
"""How long does the E-step loop take to reach its settled rate?  The C2 sample, then groups of 20 iterations timed back
to back (wall clock around eng.step(20) + synchronize) and the same after idle gaps.  Usage: python3 scripts/short_run_clock.py"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from gbrs_amd import synth, synth_torch  # noqa: E402
from gbrs_amd.engine import EmEngine  # noqa: E402

prob = synth_torch.make_em_problem_device(40_000_000, 8, 120_000, synth.SEED_BASE_EM + 1, "cuda:0", row_seed=synth.SEED_BASE_EM + 1)
eng = EmEngine.from_device(prob["R"], prob["L"], prob["H"], [t.data_ptr() for t in prob["indptr"]],
                           [t.data_ptr() for t in prob["indices"]], None, prob["eff_len"].data_ptr(), device=0)
eng.prepare(0.0)


def group(k=20):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    eng.step(k)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 1e3 / k, eng.info().last_estep_ms


eng.step(5)
print("back to back:", " ".join("%.4f/%.4f" % group() for _ in range(12)))
for gap in (0.001, 0.01, 0.1, 1.0):
    time.sleep(gap)
    print(f"after {gap} s idle:", " ".join("%.4f/%.4f" % group() for _ in range(4)))
print("groups of 1:", " ".join("%.4f" % group(1)[0] for _ in range(8)))
print("groups of 8:", " ".join("%.4f" % group(8)[0] for _ in range(8)))
print("groups of 100:", " ".join("%.4f/%.4f" % group(100) for _ in range(4)))
