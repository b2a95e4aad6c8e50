#!/usr/bin/env python3
"""Device time of one reconstruct pass (40k genes, 36 states) against the number of samples in the batch, on every kernel path
that can take that batch: the blocked scan (GBRS_TUNING_HMM_BLOCKED), the per-sample chains, the MFMA sweeps
(GBRS_TUNING_HMM_MFMA) - to place the hand-over points.  Usage: python3 scripts/hmm_path_sweep.py [max_samples]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
from gbrs_amd import synth  # noqa: E402
from gbrs_amd.hmm import DiplotypeHMM  # noqa: E402

nmax = int(sys.argv[1]) if len(sys.argv) > 1 else 64
prob = synth.make_hmm_problem(H=8)
chroms = prob.chroms
rng = np.random.default_rng(1)
base, ha, av = [], [], []
for c in chroms:
    ids = prob.gene_ids[c]
    e = np.array([prob.expr[g] for g in ids])
    base.append(np.stack([e] + [rng.gamma(1.0, 5.0, size=e.shape) * (rng.random(e.shape) < 0.5) for _ in range(nmax - 1)]))
    ha.append(np.array([g in prob.avecs for g in ids], dtype=np.uint8))
    av.append(np.array([prob.avecs.get(g, np.zeros((8, 8))) for g in ids]))

PATHS = {"default": {},
         "blocked": {"GBRS_TUNING_HMM_BLOCKED": "1000", "GBRS_TUNING_HMM_MFMA": "0"},
         "chains": {"GBRS_TUNING_HMM_BLOCKED": "0", "GBRS_TUNING_HMM_MFMA": "0"},
         "mfma": {"GBRS_TUNING_HMM_BLOCKED": "0", "GBRS_TUNING_HMM_MFMA": "1"}}
KEYS = sorted({k for p in PATHS.values() for k in p})


def measure(ns, env):
    for k in KEYS:
        os.environ.pop(k, None)
    os.environ.update(env)
    hmm = DiplotypeHMM(8, chroms, [len(prob.gene_ids[c]) for c in chroms], [prob.tprob[c] for c in chroms])
    ex = [b[0] if ns == 1 else b[:ns] for b in base]
    hmm.set_expression(ex, av, ha, 1.5, 0.12)
    for _ in range(3):
        hmm.run()
    t = []
    for _ in range(7):
        hmm.set_expression(ex, expr_threshold=1.5, sigma=0.12)
        hmm.run()
        inf = hmm.info()
        t.append(inf.last_emission_ms + inf.last_run_ms)
    hmm.close()
    return float(np.median(t))


print("samples " + " ".join(f"{p:>9s}" for p in PATHS))
for ns in [n for n in (1, 2, 3, 4, 5, 6, 8, 12, 16, 24, 32, 48, 64, 96, 128) if n <= nmax]:
    row = []
    for name, env in PATHS.items():
        if name == "blocked" and ns > 16:
            row.append("        -")
            continue
        if name == "mfma" and ns < 8:
            row.append("        -")
            continue
        try:
            row.append(f"{measure(ns, env):9.3f}")
        except Exception as ex:                        # noqa: BLE001
            row.append(f"  {type(ex).__name__[:7]:>7s}")
    print(f"{ns:7d} " + " ".join(row), flush=True)
