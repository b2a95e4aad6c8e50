"""The CPU oracle against the committed reference outputs (no GPU, no /root/reference)."""
import numpy as np
import pytest

from conftest import em_case_inputs, em_case_values, golden_files, hmm_case_inputs, load_golden, viterbi_decision_margins
from oracle import hmm_oracle
from oracle.em_oracle import EMOracle, tpm_report_values

RTOL = 1e-12   # same numpy build: bit-identical here; other CPUs may differ in the last bits


@pytest.mark.parametrize("path", golden_files("em"), ids=lambda p: p.split("/")[-1][:-4])
def test_em_oracle_matches_reference_outputs(path):
    g = load_golden(path)
    R, L, H, indptr, indices, count, eff_len, groups, gtmask = em_case_inputs(g)
    o = EMOracle(R, L, H, indptr, indices, count, values=em_case_values(g))
    if gtmask is not None:
        o.apply_genotype_mask(gtmask)
    o.prepare(pseudocount=float(g["pseudocount"]), eff_len=eff_len)
    np.testing.assert_allclose(o.theta, g["theta0"], rtol=RTOL, atol=0)
    snaps = {}

    def on_iter(i, theta, err):
        if f"theta_iter{i}" in g:
            snaps[i] = theta.copy()
    n = o.run(tol=float(g["tol"]), max_iters=int(g["max_iters"]), on_iter=on_iter)
    assert n == int(g["num_iters"])
    for i, th in snaps.items():
        np.testing.assert_allclose(th, g[f"theta_iter{i}"], rtol=RTOL, atol=0)
    np.testing.assert_allclose(o.err_history, g["err_history"], rtol=1e-9)
    np.testing.assert_allclose(o.theta, g["theta_final"], rtol=RTOL, atol=0)
    np.testing.assert_allclose(o.expected_read_counts(), g["expected_counts"], rtol=RTOL, atol=0)
    np.testing.assert_allclose(EMOracle.group_sums(o.theta, groups), g["gene_theta"], rtol=RTOL, atol=0)
    np.testing.assert_allclose(EMOracle.group_sums(o.expected_read_counts(), groups), g["gene_counts"],
                               rtol=RTOL, atol=0)
    # conservation: sum of expected counts == number of (weighted) reads with an alignment
    w = np.ones(R) if count is None else count
    aligned = np.zeros(R, dtype=bool)
    for h in range(H):
        aligned[o.indices[h]] = True
    assert abs(o.expected_read_counts().sum() - w[aligned].sum()) < 1e-6 * max(1.0, w.sum())
    rep, _ = tpm_report_values(g["theta_final"])
    line1 = str(g["text_isoforms_tpm"]).split("\n")[1].split("\t")[1:]
    assert line1 == [str(x) for x in rep[:, 0]]


@pytest.mark.parametrize("path", golden_files("hmm"), ids=lambda p: p.split("/")[-1][:-4])
def test_hmm_oracle_matches_reference_outputs(path):
    g = load_golden(path)
    c = hmm_case_inputs(g)
    H = c["H"]
    iv = hmm_oracle.init_vector(H)
    np.testing.assert_array_equal(iv, g["init_vec"])
    for ch in c["chroms"]:
        n = len(c["genes"][ch])
        E = np.array([hmm_oracle.emission(c["expr"][ch][i],
                                          c["avecs"][ch][i] if c["has_avec"][ch][i] else None, iv,
                                          float(g["expr_threshold"]), float(g["sigma"]))
                      for i in range(n)])
        np.testing.assert_allclose(E, g[f"eprob_{ch}"], rtol=1e-12, atol=0)
        T = c["tprob"][ch]
        alpha, scaler = hmm_oracle.forward(T, E, iv)
        beta = hmm_oracle.backward(T, E, scaler)
        gamma = hmm_oracle.posterior(alpha, beta)
        delta, states, calls = hmm_oracle.viterbi(T, E, iv)
        np.testing.assert_allclose(gamma, g[f"gamma_{ch}"], rtol=1e-10, atol=1e-300)
        np.testing.assert_allclose(gamma.sum(axis=0), 1.0, rtol=1e-12)
        np.testing.assert_array_equal(states, g[f"states_{ch}"])
        np.testing.assert_array_equal(calls, g[f"calls_{ch}"])
        # both tprob-length conventions (gbrs_utils.py:589-596)
        if bool(g["len_minus_one"]):
            assert len(T) == n - 1 and calls[-1] == -1 and len(states) == n
        else:
            assert len(T) == n and (calls >= 0).all() and len(states) == n + 1


@pytest.mark.parametrize("path", golden_files("hmm"), ids=lambda p: p.split("/")[-1][:-4])
def test_viterbi_calls_are_decided_by_a_wide_margin(path):
    """The genotype calls are argmax decisions over delta + T.  On every golden the closest decision is
    won by more than 1e-3 log units, nine orders of magnitude above the 1e-12-level differences between a
    device-computed delta and the reference's (tests/test_hmm_gpu.py measures those): that is what
    "bit-exact calls" rests on, next to the tests that compare the calls themselves."""
    g = load_golden(path)
    c = hmm_case_inputs(g)
    gaps = np.concatenate([viterbi_decision_margins(c["tprob"][ch], g[f"delta_{ch}"]) for ch in c["chroms"]])
    assert gaps.min() > 1e-3, gaps.min()
