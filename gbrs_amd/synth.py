"""Synthetic DO-shaped inputs for the two hot paths.

Nothing here comes from the reference (it ships no data and no generator); the
recipes are the ones SURVEY.md §8(d) fixes so that every round measures the same
workload.  numpy generators are used by the tests and the CPU baseline; the
torch generator builds the full-size EM workload directly in HBM for bench.py.

Names follow the reference's domain: a *row* is a read (or an equivalence class
after ``gbrs compress``), a *locus* is an isoform, a *group* is a gene, and the
alignment incidence tensor is stored as one CSC (rows x loci) matrix per
haplotype, exactly the on-disk EMASE form (Sparse3DMatrix.py:80-92).
"""
from __future__ import annotations

import dataclasses
from itertools import combinations_with_replacement

import numpy as np

SEED_BASE_EM = 20241008
SEED_HMM = 20241108

# gene counts per chromosome, proportional to the mouse genome, sum = 40,000
MOUSE_CHROMS = [str(i) for i in range(1, 20)] + ["X"]
MOUSE_GENES = [2600, 3100, 2100, 2500, 2400, 2300, 3200, 2000, 2300, 1900,
               3000, 1500, 1600, 1500, 1500, 1300, 1900, 1000, 1200, 1100]
assert sum(MOUSE_GENES) == 40000 and len(MOUSE_GENES) == len(MOUSE_CHROMS)


@dataclasses.dataclass
class Incidence:
    """Alignment incidence tensor in the reference's CSC-per-haplotype form."""
    num_rows: int                 # R
    num_loci: int                 # L
    num_haps: int                 # H
    indptr: list                  # H arrays uint32 [L+1]
    indices: list                 # H arrays uint32 [nnz_h]  (row ids)
    count: np.ndarray | None      # float64 [R] EC multiplicity or None
    raw_length: np.ndarray        # float64 [L] transcript length (same for every haplotype)
    groups: list                  # G lists of locus ids
    hap_names: list
    locus_names: list
    group_names: list

    @property
    def nnz(self) -> int:
        return int(sum(len(x) for x in self.indices))

    def effective_length(self, read_length: int = 100) -> np.ndarray:
        """(H x L) max(len - read_length + 1, 1), EMfactory.py:75-77."""
        eff = np.maximum(self.raw_length - read_length + 1.0, 1.0)
        return np.ascontiguousarray(np.tile(eff, (self.num_haps, 1)))


def _gene_layout(rng, L):
    sizes = []
    tot = 0
    while tot < L:
        s = 1 + int(rng.poisson(1.5))
        s = min(s, L - tot)
        sizes.append(s)
        tot += s
    sizes = np.asarray(sizes, dtype=np.int64)
    starts = np.concatenate(([0], np.cumsum(sizes)[:-1]))
    gene_of = np.repeat(np.arange(len(sizes)), sizes)
    return sizes, starts, gene_of


def make_em_rows(R, H, L, seed):
    """Row-major description of the synthetic reads: primary locus, sibling locus
    (or -1) and the haplotype bitmask shared by both.  Also returns gene layout,
    abundances and raw transcript lengths."""
    rng = np.random.default_rng(seed)
    sizes, starts, gene_of = _gene_layout(rng, L)
    abundance = rng.lognormal(0.0, 2.0, size=L) * (rng.random(L) < 0.6)
    if abundance.sum() <= 0:
        abundance[0] = 1.0
    p = abundance / abundance.sum()
    raw_len = np.round(rng.lognormal(7.3, 0.6, size=L))
    cdf = np.cumsum(p)
    cdf[-1] = 1.0
    t = np.searchsorted(cdf, rng.random(R), side="right").astype(np.int64)
    t = np.minimum(t, L - 1)
    true_hap = rng.integers(0, H, size=R)
    mask = np.zeros(R, dtype=np.uint32)
    for h in range(H):
        hit = (rng.random(R) < 0.85) | (true_hap == h)
        mask |= hit.astype(np.uint32) << np.uint32(h)
    g = gene_of[t]
    sib = starts[g] + (rng.random(R) * sizes[g]).astype(np.int64)
    use_sib = (rng.random(R) < 0.5) & (sib != t)
    sib = np.where(use_sib, sib, -1)
    return dict(primary=t, sibling=sib, mask=mask, sizes=sizes, starts=starts,
                raw_len=raw_len)


def rows_to_csc(R, H, L, primary, sibling, mask):
    """Build the per-haplotype CSC arrays (row ids ascending inside a column)."""
    indptr, indices = [], []
    has_sib = sibling >= 0
    rows_all = np.arange(R, dtype=np.int64)
    for h in range(H):
        bit = ((mask >> np.uint32(h)) & np.uint32(1)).astype(bool)
        loc = np.concatenate((primary[bit], sibling[bit & has_sib]))
        row = np.concatenate((rows_all[bit], rows_all[bit & has_sib]))
        order = np.lexsort((row, loc))
        loc = loc[order]
        row = row[order]
        ptr = np.searchsorted(loc, np.arange(L + 1), side="left")
        indptr.append(ptr.astype(np.uint32))
        indices.append(row.astype(np.uint32))
    return indptr, indices


def make_em_problem(R=100_000, H=2, L=5_000, seed=SEED_BASE_EM, with_count=False,
                    max_count=1) -> Incidence:
    d = make_em_rows(R, H, L, seed)
    indptr, indices = rows_to_csc(R, H, L, d["primary"], d["sibling"], d["mask"])
    count = None
    if with_count:
        rng = np.random.default_rng(seed + 7919)
        count = rng.integers(1, max(2, max_count + 1), size=R).astype(np.float64)
    sizes, starts = d["sizes"], d["starts"]
    groups = [list(range(int(s), int(s + n))) for s, n in zip(starts, sizes)]
    hap_names = [chr(ord("A") + h) for h in range(H)]
    locus_names = [f"T{l:07d}" for l in range(L)]
    group_names = [f"G{g:07d}" for g in range(len(groups))]
    return Incidence(R, L, H, indptr, indices, count, d["raw_len"], groups,
                     hap_names, locus_names, group_names)


def compress_rows(inc: Incidence) -> Incidence:
    """Collapse identical rows into equivalence classes with multiplicities.  Same
    result as ``gbrs compress`` (gbrs/emase_utils.py:60-103) up to row order, which
    no downstream number depends on."""
    R, H, L = inc.num_rows, inc.num_haps, inc.num_loci
    rows = np.concatenate([np.asarray(ix, dtype=np.int64) for ix in inc.indices])
    cols = np.concatenate([
        np.repeat(np.arange(L, dtype=np.int64), np.diff(inc.indptr[h].astype(np.int64))) * H + h
        for h in range(H)])
    order = np.lexsort((cols, rows))
    rows, cols = rows[order], cols[order]
    starts = np.searchsorted(rows, np.arange(R + 1))
    keys = {}
    new_id = np.empty(R, dtype=np.int64)
    weights = []
    w_in = inc.count if inc.count is not None else np.ones(R)
    for r in range(R):
        k = cols[starts[r]:starts[r + 1]].tobytes()
        j = keys.get(k)
        if j is None:
            j = len(keys)
            keys[k] = j
            weights.append(0.0)
        weights[j] += w_in[r]
        new_id[r] = j
    first = np.full(len(keys), -1, dtype=np.int64)
    for r in range(R - 1, -1, -1):
        first[new_id[r]] = r
    indptr, indices = [], []
    for h in range(H):
        ptr = inc.indptr[h].astype(np.int64)
        loc = np.repeat(np.arange(L, dtype=np.int64), np.diff(ptr))
        row = np.asarray(inc.indices[h], dtype=np.int64)
        keep = first[new_id[row]] == row
        loc, row = loc[keep], new_id[row[keep]]
        order = np.lexsort((row, loc))
        loc, row = loc[order], row[order]
        indptr.append(np.searchsorted(loc, np.arange(L + 1)).astype(np.uint32))
        indices.append(row.astype(np.uint32))
    return dataclasses.replace(inc, num_rows=len(keys), indptr=indptr, indices=indices,
                               count=np.asarray(weights, dtype=np.float64))


# --------------------------------------------------------------------------- HMM

@dataclasses.dataclass
class HmmProblem:
    hap_names: list            # H names
    chroms: list               # chromosome order (ref.fa.fai order)
    gene_ids: dict             # chrom -> list of gene ids in genome order
    tprob: dict                # chrom -> float64 [n_t, S, S] log transition, T[i][to, from]
    expr: dict                 # gene id -> float64 [H]
    avecs: dict                # gene id -> float64 [H, H]  (missing for some genes)

    @property
    def num_states(self):
        H = len(self.hap_names)
        return H * (H + 1) // 2

    @property
    def num_genes(self):
        return sum(len(v) for v in self.gene_ids.values())


def diplotype_names(hap_names):
    return [a + b for a, b in combinations_with_replacement(hap_names, 2)]


def _recombination_tables(rng, H, nt, structural_zeros=True):
    """DO-shaped log transition tables [nt, S, S] (T[i][to, from]): between neighbouring genes each of the
    two chromosomes of a diplotype keeps its founder with probability 1 - r and switches to one of the
    other H - 1 founders with probability r, r log-uniform over 1e-15 .. 1e-2 per interval (gene-dense
    stretches are near-deterministic, a few intervals recombine freely).  So a table holds entries from
    ~1 down to r^2 / (H-1)^2 ~ 1e-32, and - when `structural_zeros` - every fourth interval forbids double
    switches outright (probability 0, log = -inf), as a table built by thresholding would."""
    S = H * (H + 1) // 2
    pairs = list(combinations_with_replacement(range(H), 2))
    # number of founder changes between unordered pairs: best matching of the two chromosomes
    change = np.zeros((S, S), dtype=np.int64)
    for j, (a, b) in enumerate(pairs):
        for k, (c, d) in enumerate(pairs):
            change[j, k] = min((a != c) + (b != d), (a != d) + (b != c))
    r = 10.0 ** rng.uniform(-15.0, -2.0, size=nt)
    T = np.empty((nt, S, S))
    for i in range(nt):
        q = r[i] / (H - 1)
        P = np.where(change == 0, (1.0 - r[i]) ** 2, np.where(change == 1, q * (1.0 - r[i]), q * q))
        P = P * (1.0 + 0.05 * rng.random((S, S)))          # break the exact symmetry of the model
        if structural_zeros and i % 4 == 1:
            P = np.where(change == 2, 0.0, P)
        T[i] = P / P.sum(axis=0, keepdims=True)            # column-stochastic: sum over `to`
    with np.errstate(divide="ignore"):
        return np.log(T)


def make_hmm_problem(H=8, genes_per_chrom=None, chroms=None, seed=SEED_HMM,
                     tprob_len_minus_one=False, style="benign", expressed_fraction=0.5) -> HmmProblem:
    """style "benign": SURVEY 8d's tables (eye + 0.01 U, log entries -5 .. 0).  style "do": recombination-
    shaped tables with entries down to ~1e-32 and structural zeros (_recombination_tables).
    expressed_fraction: probability that a haplotype of a gene is expressed at all (low values give many
    genes under the expression threshold, whose emission is the prior)."""
    rng = np.random.default_rng(seed)
    if genes_per_chrom is None:
        genes_per_chrom = MOUSE_GENES
    if chroms is None:
        chroms = MOUSE_CHROMS[:len(genes_per_chrom)]
    S = H * (H + 1) // 2
    hap_names = [chr(ord("A") + h) for h in range(H)]
    gene_ids, tprob, expr, avecs = {}, {}, {}, {}
    gno = 0
    for c, n in zip(chroms, genes_per_chrom):
        ids = [f"ENSMUSG{gno + i:011d}" for i in range(n)]
        gno += n
        gene_ids[c] = ids
        nt = n - 1 if tprob_len_minus_one else n
        if style == "do":
            tprob[c] = _recombination_tables(rng, H, nt)
        else:
            T = np.eye(S)[None, :, :] + 0.01 * rng.random((nt, S, S))
            T /= T.sum(axis=1, keepdims=True)          # column-stochastic: sum over `to`
            tprob[c] = np.log(T)
        e = rng.gamma(1.0, 5.0, size=(n, H)) * (rng.random((n, H)) < expressed_fraction)
        has_avec = rng.random(n) < 0.7
        for i, g in enumerate(ids):
            expr[g] = e[i]
            if has_avec[i]:
                a = np.eye(H) + 0.05 * rng.random((H, H))
                avecs[g] = a / a.sum(axis=1, keepdims=True)
    return HmmProblem(hap_names, list(chroms), gene_ids, tprob, expr, avecs)
