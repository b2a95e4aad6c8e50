"""CPU ORACLE (test infrastructure, not product code) for the `gbrs reconstruct` HMM.

numpy restatement of the numeric body of gbrs_utils.reconstruct
(/root/reference/src/gbrs/gbrs/gbrs_utils.py:463-599) working on in-memory inputs.
The mix of builtin ``sum`` (sequential) and ``ndarray.sum`` (pairwise) is kept exactly as
the reference has it so the float64 bits agree on the same machine.
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import it.

Parity pin: tests/golden/hmm_*.npz are written by oracle/gen_golden.py from the imported
reference's own `reconstruct()` outputs (genoprobs.npz, genotypes.tsv, genotypes.npz) and
checked bit-for-bit against this file at generation time.

Reference map (gbrs_utils.py):
  unit()                 :63-67     unit_vector
  genotype_probability() :80-98     get_genotype_probability
  init_vector()          :465-471
  emission()             :475-492
  forward()              :500-526
  backward()             :530-550
  posterior()            :554-560
  viterbi()              :567-598   including the tprob-length quirk at :589-590
"""
from __future__ import annotations

from itertools import combinations_with_replacement

import numpy as np

TINY = np.nextafter(0, 1)


def unit(v):
    if sum(v) > 1e-6:
        return v / np.linalg.norm(v)
    return v


def genotype_probability(profile, specificity, sigma=0.12):
    H = len(profile)
    u = unit(profile)
    d = []
    for i in range(H):
        vi = unit(specificity[i])
        for j in range(i, H):
            if j == i:
                d.append(sum(np.power(u - vi, 2)))
            else:
                vj = unit(specificity[j])
                g = unit(vi + vj)
                d.append(sum(np.power(u - g, 2)))
    p = np.exp(np.array(d) / (-2 * sigma * sigma))
    return np.array(p / sum(p))


def init_vector(H):
    out = []
    for a, b in combinations_with_replacement(range(H), 2):
        out.append(np.log((1.0 if a == b else 2.0) / (H * H)))
    return np.array(out)


def naive_avecs(H):
    return np.eye(H) + (np.ones((H, H)) - np.eye(H)) * 0.0001


def emission(expr_vec, avec, init_vec, expr_threshold=1.5, sigma=0.12):
    """avec is None when the gene has no alignment-specificity entry."""
    if sum(expr_vec) < expr_threshold:
        return init_vec
    if avec is None:
        return np.log(genotype_probability(expr_vec, naive_avecs(len(expr_vec)), sigma=0.450) + TINY)
    return np.log(genotype_probability(expr_vec, avec, sigma=sigma) + TINY)


def forward(T, E, init_vec):
    """T [n_t,S,S] log, E [n,S] log emission -> alpha [S,n], scaler [n]."""
    n, S = E.shape
    alpha = np.zeros((S, n))
    scaler = np.zeros(n)
    alpha[:, 0] = init_vec + E[0]
    z = np.log(sum(np.exp(alpha[:, 0])))
    alpha[:, 0] -= z
    scaler[0] = -z
    for i in range(1, n):
        alpha[:, i] = np.log(np.exp(alpha[:, i - 1] + T[i - 1]).sum(axis=1) + TINY) + E[i]
        z = np.log(sum(np.exp(alpha[:, i])))
        alpha[:, i] -= z
        scaler[i] = -z
    return alpha, scaler


def backward(T, E, scaler):
    n, S = E.shape
    beta = np.zeros((S, n))
    beta[:, -1] = scaler[-1]
    for i in range(n - 2, -1, -1):
        beta[:, i] = np.log(np.exp(T[i].transpose() + beta[:, i + 1] + E[i + 1] + scaler[i]).sum(axis=1))
    return beta


def posterior(alpha, beta):
    g = np.exp(alpha + beta)
    return g / g.sum(axis=0)


def viterbi(T, E, init_vec):
    """Returns delta [S,n], the ordered state list (as the reference stores it in
    genotypes.npz, length n'+1) and per-gene calls (-1 = gene gets no TSV entry)."""
    n, S = E.shape
    delta = np.zeros((S, n))
    delta[:, 0] = init_vec + E[0]
    for i in range(1, n):
        delta[:, i] = (delta[:, i - 1] + T[i - 1]).max(axis=1) + E[i]
    sid = int(delta[:, n - 1].argmax())
    states = [sid]
    calls = np.full(n, -1, dtype=np.int32)
    m = n
    if m > len(T):
        m = len(T)
    for i in reversed(range(m)):
        sid = int((delta[:, i] + T[i][sid]).argmax())
        states.append(sid)
        calls[i] = sid
    states.reverse()
    return delta, np.asarray(states, dtype=np.int32), calls


def reconstruct_arrays(hap_names, chroms, gene_ids, tprob, expr, avecs,
                       expr_threshold=1.5, sigma=0.12):
    """Whole numeric body on dict inputs shaped like the reference's locals.  Returns a dict
    of per-chromosome arrays."""
    H = len(hap_names)
    iv = init_vector(H)
    out = {}
    for c in chroms:
        if c not in tprob:
            continue
        ids = gene_ids[c]
        E = np.array([emission(expr[g], avecs.get(g), iv, expr_threshold, sigma) for g in ids])
        T = tprob[c]
        alpha, scaler = forward(T, E, iv)
        beta = backward(T, E, scaler)
        gamma = posterior(alpha, beta)
        delta, states, calls = viterbi(T, E, iv)
        out[c] = dict(eprob=E, alpha=alpha, scaler=scaler, beta=beta, gamma=gamma,
                      delta=delta, states=states, calls=calls)
    return out
