#!/usr/bin/env python3
"""One DO-sized HMM workload (40k genes, 36 states), n samples, a few passes: for rocprofv3 kernel traces of the
reconstruct kernels without the EM benchmark around them.  Usage: python3 scripts/hmm_only.py [n_samples] [passes]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
from gbrs_amd import synth  # noqa: E402
from gbrs_amd.hmm import DiplotypeHMM  # noqa: E402

ns = int(sys.argv[1]) if len(sys.argv) > 1 else 1
passes = int(sys.argv[2]) if len(sys.argv) > 2 else 10
prob = synth.make_hmm_problem(H=8)
chroms = prob.chroms
hmm = DiplotypeHMM(8, chroms, [len(prob.gene_ids[c]) for c in chroms], [prob.tprob[c] for c in chroms])
rng = np.random.default_rng(1)
ex, av, ha = [], [], []
for c in chroms:
    ids = prob.gene_ids[c]
    e = np.array([prob.expr[g] for g in ids])
    if ns > 1:
        e = np.stack([e] + [rng.gamma(1.0, 5.0, size=e.shape) * (rng.random(e.shape) < 0.5) for _ in range(ns - 1)])
    ex.append(e)
    ha.append(np.array([g in prob.avecs for g in ids], dtype=np.uint8))
    av.append(np.array([prob.avecs.get(g, np.zeros((8, 8))) for g in ids]))
hmm.set_expression(ex, av, ha, 1.5, 0.12)
for _ in range(3):
    hmm.run()
tot = []
for _ in range(passes):
    ts = time.perf_counter()
    hmm.set_expression(ex, expr_threshold=1.5, sigma=0.12)
    t0 = time.perf_counter()
    hmm.run()
    inf = hmm.info()
    t1 = time.perf_counter()
    tot.append((inf.last_emission_ms + inf.last_run_ms, (t1 - t0) * 1e3, inf.last_forward_ms, inf.last_backward_ms, (t0 - ts) * 1e3))
m = np.median(np.array(tot), axis=0)
print(f"samples {ns}: device {m[0]:.4f} ms  run wall {m[1]:.3f} ms  forward {m[2]:.3f}  backward {m[3]:.3f}  "
      f"{prob.num_genes * ns / m[0] / 1e3:.1f} M genes/s   set_expression wall {m[4]:.3f} ms, set + run wall {m[4] + m[1]:.3f} ms")
inf = hmm.info()
if inf.last_delta_blocks or inf.last_delta_fallbacks:
    print(f"delta by rank convergence: {inf.last_delta_blocks} blocks fixed up, longest fix-up {inf.last_delta_longest_fixup} genes, "
          f"{inf.last_delta_fallbacks} (sample, chromosome) fallbacks")
