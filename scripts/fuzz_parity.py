#!/usr/bin/env python3
"""Randomised differential campaign on an MI355X: HIP path vs the CPU oracle over many random
shapes (EM: row-length mixes, haplotype counts, weights, merge / row-order flags, tolerances with
the stopping rule; HMM: founder counts, chromosome lengths, both tprob conventions, sample batches).
Test infrastructure like tests/: not part of the product.  Usage: python scripts/fuzz_parity.py [seconds]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))


def em_case(rng):
    from gbrs_amd.engine import EmEngine
    from oracle.em_oracle import EMOracle
    from test_em_gpu import _random_rows_problem, _shared_mask_rows_problem
    H = int(rng.choice([1, 2, 3, 4, 8, 8, 8, 16]))
    L = int(rng.integers(5, 4000))
    R = int(rng.integers(1, 30000))
    lo = int(rng.integers(1, 4))
    hi = int(rng.choice([lo, lo + 1, 3, 6, 12, 40, 70]))
    hi = max(lo, min(hi, L))
    lo = min(lo, hi)
    cnt = bool(rng.integers(0, 2))
    seed = int(rng.integers(1, 1 << 30))
    # a third of the cases: rows that share one mask over their loci, with the locus sets of the layout forced on
    shared = rng.random() < 0.35
    os.environ["GBRS_TUNING_LOCUS_SETS"] = "1" if shared else "0"
    if shared:
        indptr, indices, count, eff = _shared_mask_rows_problem(R, H, max(L, 8), seed, max(hi, 2), cnt)
        L = max(L, 8)
    else:
        indptr, indices, count, eff = _random_rows_problem(R, H, L, seed, lo, hi, cnt)
    pc = float(rng.choice([0.0, 0.0, 0.5]))
    tol = float(rng.choice([0.0, 1e-2, 1e-4]))
    iters = int(rng.integers(1, 9))
    o = EMOracle(R, L, H, indptr, indices, count)
    o.prepare(pc, eff)
    theta0 = o.theta.copy()
    n_o = o.run(tol=tol, max_iters=iters)
    for flags in (0, 1, 2, 16, 17, 16 | 4, 8, 32):
        eng = EmEngine.from_host(R, L, H, indptr, indices, count, eff, flags=flags)
        eng.prepare(pc)
        np.testing.assert_allclose(eng.theta(), theta0, rtol=1e-9, atol=1e-300)
        n, hist = eng.run(model=4, tol=tol, max_iters=iters)
        # An err_sum at rounding-noise level (an exact fixed point: 0.0 in numpy, 1e-10 here because the
        # reciprocal is a Newton iteration) decides the stop on noise when tol = 0: compare the
        # iteration count and history only where the stopping quantity is above that level.
        noise = any(e < 1e-6 for e in o.err_history)
        if not noise:
            assert n == o.num_iters, (n, o.num_iters, flags)
            np.testing.assert_allclose(hist, o.err_history, rtol=1e-7, atol=1e-6)    # err_sum is on a 1e6 (TPM) scale
        np.testing.assert_allclose(eng.theta(), o.theta, rtol=1e-9, atol=1e-300)
        np.testing.assert_allclose(eng.expected_counts(), o.expected_read_counts(), rtol=1e-9, atol=1e-300)
        # stepping API gives the same as run for a fixed count
        eng.prepare(pc)
        eng.step(n)
        np.testing.assert_allclose(eng.theta(), o.theta, rtol=1e-9, atol=1e-300)
        eng.close()
    return f"EM R={R} H={H} L={L} rows {lo}-{hi} count={cnt} pc={pc} tol={tol} iters={o.num_iters} shared_masks={shared}"


def hmm_case(rng):
    from gbrs_amd import synth
    from gbrs_amd.hmm import DiplotypeHMM
    from oracle import hmm_oracle
    H = int(rng.choice([2, 3, 4, 5, 7, 8, 8, 8, 9, 16]))
    nch = int(rng.integers(1, 6))
    lens = [int(x) for x in rng.integers(1, 90 if H == 16 else 400, size=nch)]
    ns = int(rng.choice([1, 1, 2, 3, 4, 5, 7, 16, 25, 40]))   # from 16 on: the MFMA sweeps (GBRS_TUNING_HMM_MFMA below)
    # the blocked scan (1-4 samples, 36 states) with blocks short enough to cut these small chromosomes many times, and
    # the samples-on-lanes backpointers with partly filled wavefronts
    tuning = {"GBRS_TUNING_HMM_BLOCK_GENES": str(int(rng.choice([2, 3, 5, 9, 17, 40]))),
              "GBRS_TUNING_HMM_BLOCKS_MAX": str(int(rng.choice([2, 5, 64]))),
              "GBRS_TUNING_HMM_BLOCKED": str(int(rng.choice([0, 2, 2, 4]))),
              "GBRS_TUNING_HMM_BPLANES": str(int(rng.choice([5, 32]))),
              # round 4: Viterbi values of the blocked scan by rank convergence (default) / max-plus operators / a tolerance no
              # block meets (every chromosome through the fallback chain); the batch kernels' layout and grid switches
              "GBRS_TUNING_HMM_DELTA_SPEC": str(int(rng.choice([1, 1, 0]))),
              "GBRS_TUNING_HMM_DELTA_TOL": str(rng.choice(["1e-9", "1e-9", "-1"])),
              "GBRS_TUNING_HMM_DELTA_ROWS": str(int(rng.integers(0, 2))),
              "GBRS_TUNING_HMM_XCD": str(int(rng.choice([0, 0, 1, 2, 3])))}
    os.environ.update(tuning)
    style = str(rng.choice(["benign", "do"])) if H == 8 else "benign"
    minus_one = bool(rng.integers(0, 2))
    seed = int(rng.integers(1, 1 << 30))
    probs = [synth.make_hmm_problem(H=H, genes_per_chrom=lens, seed=seed + s, tprob_len_minus_one=minus_one, style=style)
             for s in range(ns)]
    p0 = probs[0]
    chroms = p0.chroms
    hmm = DiplotypeHMM(H, chroms, [len(p0.gene_ids[c]) for c in chroms], [p0.tprob[c] for c in chroms])
    ex, av, ha = [], [], []
    for c in chroms:
        ids = p0.gene_ids[c]
        ex.append(np.array([[p.expr[g] for g in ids] for p in probs]))
        ha.append(np.array([g in p0.avecs for g in ids], dtype=np.uint8))
        av.append(np.array([p0.avecs.get(g, np.zeros((H, H))) for g in ids]))
    hmm.set_expression(ex, av, ha, 1.5, 0.12)
    hmm.run()
    want = ("gamma", "states", "calls", "alpha", "beta", "delta", "scaler")
    for s, p in enumerate(probs):
        res = hmm_oracle.reconstruct_arrays(p0.hap_names, chroms, p0.gene_ids, p0.tprob, p.expr, p0.avecs)
        for ci, c in enumerate(chroms):
            r = hmm.get(ci, sample=s, want=want)
            np.testing.assert_array_equal(r["states"], res[c]["states"])
            np.testing.assert_array_equal(r["calls"], res[c]["calls"])
            for k in ("alpha", "beta", "delta", "scaler"):
                np.testing.assert_allclose(r[k], res[c][k], rtol=1e-9, atol=1e-9)
            np.testing.assert_allclose(r["gamma"], res[c]["gamma"], rtol=1e-8, atol=1e-300)
    hmm.close()
    return (f"HMM H={H} lens={lens} samples={ns} tprob_n-1={minus_one} tables={style} "
            + " ".join(f"{k[16:].lower()}={v}" for k, v in tuning.items()))


def main():
    os.environ.setdefault("GBRS_TUNING_HMM_MFMA", "16")
    os.environ.setdefault("GBRS_TUNING_HMM_DLANES", "16")
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
    rng = np.random.default_rng(int(os.environ.get("FUZZ_SEED", "20241008")))
    t0 = time.time()
    n = {"em": 0, "hmm": 0}
    while time.time() - t0 < budget:
        which = "em" if rng.random() < 0.6 else "hmm"
        desc = em_case(rng) if which == "em" else hmm_case(rng)
        n[which] += 1
        if (n["em"] + n["hmm"]) % 10 == 0:
            print(f"[{time.time() - t0:6.1f}s] {n} last: {desc}", flush=True)
    print(f"fuzz ok: {n} cases in {time.time() - t0:.1f} s", flush=True)


if __name__ == "__main__":
    main()
