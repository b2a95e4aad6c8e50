"""Full-size synthetic EM workload built directly in HBM with torch (bench.py only).

Same recipe as gbrs_amd.synth.make_em_rows / rows_to_csc but with torch's
generator, so a 40M-read DO-shaped sample is ready in a couple of seconds instead of minutes.
PyTorch is plumbing here (device memory + sort); nothing in this file is on the measured path.
"""
from __future__ import annotations

import numpy as np
import torch

from . import synth


def make_em_problem_device(R, H, L, seed, device="cuda:0", row_seed=None, variant="survey"):
    """variant "survey" is SURVEY.md §8d's generator (<= 2 loci per read, one mask shared by a read's loci);
    "multi_isoform" draws reads the way isoforms that share exons collect them (make_multi_isoform_device).

    Returns dict(indptr=[H int32 tensors L+1], indices=[H int32 tensors], eff_len (H x L) float64
    tensor, N, groups info) with every array resident on `device`.  int32 tensors carry the
    uint32 bit patterns the C ABI expects (all values < 2^31 at these sizes).  `seed` fixes the sample
    model (genes, abundances, lengths); `row_seed` (default: seed) the reads drawn from it, so the ranks
    of a sharded run draw different reads of one sample."""
    assert R < 2**31 and L < 2**24
    if variant == "multi_isoform":
        return make_multi_isoform_device(R, H, L, seed, device, row_seed)
    assert variant == "survey", variant
    rng = np.random.default_rng(seed)
    sizes, starts, gene_of = synth._gene_layout(rng, L)
    abundance = rng.lognormal(0.0, 2.0, size=L) * (rng.random(L) < 0.6)
    p = abundance / abundance.sum()
    raw_len = np.round(rng.lognormal(7.3, 0.6, size=L))
    dev = torch.device(device)
    g = torch.Generator(device=dev)
    g.manual_seed(int(seed if row_seed is None else row_seed))
    cdf = torch.from_numpy(np.cumsum(p)).to(dev)
    cdf[-1] = 1.0
    t = torch.searchsorted(cdf, torch.rand(R, generator=g, device=dev, dtype=torch.float64), right=True)
    t.clamp_(max=L - 1)
    true_hap = torch.randint(0, H, (R,), generator=g, device=dev, dtype=torch.int32)
    mask = torch.zeros(R, dtype=torch.int32, device=dev)
    for h in range(H):
        hit = (torch.rand(R, generator=g, device=dev) < 0.85) | (true_hap == h)
        mask |= hit.to(torch.int32) << h
        del hit
    gene_of_t = torch.from_numpy(gene_of).to(dev)[t]
    sizes_t = torch.from_numpy(sizes).to(dev)[gene_of_t]
    starts_t = torch.from_numpy(starts).to(dev)[gene_of_t]
    sib = starts_t + (torch.rand(R, generator=g, device=dev, dtype=torch.float64) * sizes_t).to(torch.int64)
    use_sib = (torch.rand(R, generator=g, device=dev) < 0.5) & (sib != t)
    del gene_of_t, sizes_t, starts_t, true_hap
    rows_all = torch.arange(R, device=dev, dtype=torch.int64)
    indptr, indices = [], []
    n_total = 0
    bounds = torch.arange(L + 1, device=dev, dtype=torch.int64) * R
    for h in range(H):
        bit = ((mask >> h) & 1).bool()
        k1 = t[bit] * R + rows_all[bit]
        b2 = bit & use_sib
        k2 = sib[b2] * R + rows_all[b2]
        key, _ = torch.sort(torch.cat((k1, k2)))
        del k1, k2, b2, bit
        indices.append((key % R).to(torch.int32))
        indptr.append(torch.searchsorted(key, bounds).to(torch.int32))
        n_total += int(key.numel())
        del key
    eff = np.maximum(raw_len - 100 + 1.0, 1.0)
    eff_len = torch.from_numpy(np.ascontiguousarray(np.tile(eff, (H, 1)))).to(dev)
    torch.cuda.synchronize(dev)
    return dict(indptr=indptr, indices=indices, eff_len=eff_len, N=n_total, R=R, H=H, L=L,
                num_groups=len(sizes), gene_starts=np.asarray(starts, dtype=np.int64))


MULTI_ISOFORM_MAX_LOCI = 12


def make_multi_isoform_device(R, H, L, seed, device="cuda:0", row_seed=None):
    """A second EM workload with realistic multi-isoform reads (round-2 review): same sample model as the survey
    generator (genes, abundances, lengths), but a read aligns to 1 + Poisson(2) isoforms of its gene (capped at the
    gene's size and at MULTI_ISOFORM_MAX_LOCI; a cyclic window of the gene's isoforms that contains the true one), and
    the haplotype masks differ from locus to locus: every (locus, haplotype) alignment of the read's base mask is
    dropped with probability 0.1, except the true haplotype at the true isoform.  Same return value as
    make_em_problem_device."""
    rng = np.random.default_rng(seed)
    sizes, starts, gene_of = synth._gene_layout(rng, L)
    abundance = rng.lognormal(0.0, 2.0, size=L) * (rng.random(L) < 0.6)
    p = abundance / abundance.sum()
    raw_len = np.round(rng.lognormal(7.3, 0.6, size=L))
    dev = torch.device(device)
    g = torch.Generator(device=dev)
    g.manual_seed(int(seed if row_seed is None else row_seed) + 7919)
    cdf = torch.from_numpy(np.cumsum(p)).to(dev)
    cdf[-1] = 1.0
    t = torch.searchsorted(cdf, torch.rand(R, generator=g, device=dev, dtype=torch.float64), right=True)
    t.clamp_(max=L - 1)
    true_hap = torch.randint(0, H, (R,), generator=g, device=dev, dtype=torch.int32)
    base = torch.zeros(R, dtype=torch.int32, device=dev)
    for h in range(H):
        hit = (torch.rand(R, generator=g, device=dev) < 0.85) | (true_hap == h)
        base |= hit.to(torch.int32) << h
        del hit
    gene_of_t = torch.from_numpy(gene_of).to(dev)[t]
    size_t = torch.from_numpy(sizes).to(dev)[gene_of_t].to(torch.int64)
    start_t = torch.from_numpy(starts).to(dev)[gene_of_t].to(torch.int64)
    del gene_of_t
    want = 1 + torch.poisson(torch.full((R,), 2.0, device=dev), generator=g).to(torch.int64)
    k = torch.minimum(torch.minimum(want, size_t), torch.tensor(MULTI_ISOFORM_MAX_LOCI, device=dev))
    del want
    # window of k isoforms of the gene, cyclic, with the true isoform at position u of it
    u = (torch.rand(R, generator=g, device=dev, dtype=torch.float64) * k).to(torch.int64)
    first = (t - start_t - u) % size_t
    kmax = int(k.max().item())
    locs, masks = [], []
    true_bit = (torch.ones_like(true_hap) << true_hap)
    for j in range(kmax):
        loc = start_t + (first + j) % size_t
        m = base.clone()
        for h in range(H):
            drop = torch.rand(R, generator=g, device=dev) < 0.1
            m &= ~(drop.to(torch.int32) << h)
            del drop
        m |= torch.where(u == j, true_bit, torch.zeros_like(true_bit))
        m = torch.where(k > j, m, torch.zeros_like(m))
        locs.append(loc)
        masks.append(m)
    del base, true_bit, first, u, size_t, start_t, true_hap
    rows_all = torch.arange(R, device=dev, dtype=torch.int64)
    bounds = torch.arange(L + 1, device=dev, dtype=torch.int64) * R
    indptr, indices = [], []
    n_total = 0
    for h in range(H):
        keys = []
        for loc, m in zip(locs, masks):
            bit = ((m >> h) & 1).bool()
            keys.append(loc[bit] * R + rows_all[bit])
            del bit
        key, _ = torch.sort(torch.cat(keys))
        del keys
        indices.append((key % R).to(torch.int32))
        indptr.append(torch.searchsorted(key, bounds).to(torch.int32))
        n_total += int(key.numel())
        del key
    pairs = int(sum(int((m != 0).sum().item()) for m in masks))
    del locs, masks
    eff = np.maximum(raw_len - 100 + 1.0, 1.0)
    eff_len = torch.from_numpy(np.ascontiguousarray(np.tile(eff, (H, 1)))).to(dev)
    if dev.type == "cuda":
        torch.cuda.synchronize(dev)
    return dict(indptr=indptr, indices=indices, eff_len=eff_len, N=n_total, R=R, H=H, L=L,
                num_groups=len(sizes), gene_starts=np.asarray(starts, dtype=np.int64), read_locus_pairs=pairs,
                mean_loci_per_read=pairs / float(R))
