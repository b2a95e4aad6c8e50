#!/usr/bin/env python3
"""Generate tests/golden/*.npz by RUNNING THE IMPORTED REFERENCE in this container.

Run from the repo root (build container only; /root/reference does not exist on the GPU box):

    PYTHONDONTWRITEBYTECODE=1 python oracle/gen_golden.py

What it does, per case:
  * builds a small seeded synthetic input (gbrs_amd.synth),
  * runs the reference's own classes/functions on it
      - EM : gbrs.emase.EMfactory / AlignmentPropertyMatrix (needs an empty stand-in for the
             absent `tables` module; only the h5 load/save paths touch it, and those are not used),
      - HMM: gbrs.gbrs.gbrs_utils.reconstruct, unmodified, through temp files,
  * runs oracle/em_oracle.py / oracle/hmm_oracle.py on the same input and asserts the results
    are bit-identical to the reference's (this is what pins the oracle),
  * writes inputs + expected outputs as a small .npz fixture (data only, no reference text).
"""
import os
import shutil
import sys
import tempfile
import types

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_SRC = "/root/reference/src"
sys.dont_write_bytecode = True
sys.path.insert(0, REPO)
sys.path.insert(0, REF_SRC)

import numpy as np  # noqa: E402

WORK = tempfile.mkdtemp(prefix="gbrs_golden_")
os.environ["GBRS_DATA"] = WORK            # read at import time by gbrs_utils (gbrs_utils.py:20)
sys.modules.setdefault("tables", types.ModuleType("tables"))

from gbrs.emase.AlignmentPropertyMatrix import AlignmentPropertyMatrix as RefAPM  # noqa: E402
from gbrs.emase.EMfactory import EMfactory as RefEM  # noqa: E402
from gbrs.gbrs import gbrs_utils as ref_gbrs_utils  # noqa: E402
import scipy.sparse as sp  # noqa: E402

from gbrs_amd import synth  # noqa: E402
from oracle.em_oracle import EMOracle, tpm_report_values  # noqa: E402
from oracle import hmm_oracle  # noqa: E402

GOLD = os.path.join(REPO, "tests", "golden")
SNAP_ITERS = (1, 2, 5)


def beq(a, b, what):
    a = np.asarray(a)
    b = np.asarray(b)
    if a.shape != b.shape or not np.array_equal(a, b):
        raise AssertionError(f"oracle != reference for {what}: max abs diff "
                             f"{np.max(np.abs(a - b)) if a.shape == b.shape else 'shape'}")


# ----------------------------------------------------------------------------- EM

def ref_apm_from(inc, grpfile, values=None):
    apm = RefAPM(shape=(inc.num_loci, inc.num_haps, inc.num_rows),
                 haplotype_names=inc.hap_names, locus_names=inc.locus_names, grpfile=grpfile)
    for h in range(inc.num_haps):
        apm.data[h] = sp.csc_matrix(
            (np.ones(len(inc.indices[h])) if values is None else values[h].copy(), inc.indices[h].astype(np.int64),
             inc.indptr[h].astype(np.int64)), shape=(inc.num_rows, inc.num_loci))
    apm.finalized = True
    if inc.count is not None:
        apm.count = inc.count.copy()
    return apm


def em_case(name, R, H, L, seed, with_count, with_len, pseudocount, mask, tol, max_iters,
            drop_rows=0, with_values=False):
    inc = synth.make_em_problem(R=R, H=H, L=L, seed=seed, with_count=with_count, max_count=5)
    if drop_rows:
        # make some rows empty (no alignment at all) - reads that the aligner dropped
        rng = np.random.default_rng(seed + 1)
        dead = rng.choice(R, size=drop_rows, replace=False)
        for h in range(H):
            keep = ~np.isin(inc.indices[h], dead)
            loc = np.repeat(np.arange(L), np.diff(inc.indptr[h].astype(np.int64)))[keep]
            inc.indices[h] = inc.indices[h][keep]
            inc.indptr[h] = np.searchsorted(loc, np.arange(L + 1)).astype(np.uint32)
    case_dir = os.path.join(WORK, name)
    os.makedirs(case_dir)
    grpfile = os.path.join(case_dir, "g2t.tsv")
    with open(grpfile, "w") as fh:
        for g, members in zip(inc.group_names, inc.groups):
            fh.write(g + "\t" + "\t".join(inc.locus_names[l] for l in members) + "\n")
    lenfile = None
    if with_len:
        lenfile = os.path.join(case_dir, "lengths.info")
        with open(lenfile, "w") as fh:
            for l in range(L):
                if H > 1:
                    for hn in inc.hap_names:
                        fh.write(f"{inc.locus_names[l]}_{hn}\t{int(inc.raw_length[l])}\n")
                else:
                    fh.write(f"{inc.locus_names[l]}\t{int(inc.raw_length[l])}\n")

    values = None
    if with_values:
        # what a file saved with incidence_only=False (or a legacy COO file) carries: the matrices arrive in
        # EMfactory with values other than 1, and prepare() normalises them (EMfactory.py:95-98)
        rng = np.random.default_rng(seed + 3)
        values = [rng.random(len(ix)) + 0.25 for ix in inc.indices]
    apm = ref_apm_from(inc, grpfile, values)
    orc = EMOracle(R, L, H, inc.indptr, inc.indices, inc.count, values=values)
    gtmask = None
    if mask:
        # a called diplotype per gene: two haplotypes (possibly equal) kept, rest masked out
        rng = np.random.default_rng(seed + 2)
        gtmask = np.zeros((H, L))
        for members in inc.groups:
            a, b = rng.integers(0, H, size=2)
            hid2set = np.array([a, b])
            tid2set = np.array(members)
            gtmask[tuple(np.meshgrid(hid2set, tid2set))] = 1.0
        apm.multiply(gtmask, axis=2)
        for h in range(H):
            apm.data[h].eliminate_zeros()
        orc.apply_genotype_mask(gtmask)
        for h in range(H):
            beq(apm.data[h].indptr, orc.indptr[h], f"{name} masked indptr h{h}")
            beq(apm.data[h].indices, orc.indices[h], f"{name} masked indices h{h}")

    ref = RefEM(apm)
    ref.prepare(pseudocount=pseudocount, lenfile=lenfile)
    eff_len = inc.effective_length(100) if with_len else None
    if with_len:
        beq(ref.target_lengths, eff_len, f"{name} target_lengths")
    orc.prepare(pseudocount=pseudocount, eff_len=eff_len)
    beq(ref.allelic_expression, orc.theta, f"{name} theta0")
    out = dict(theta0=orc.theta.copy())

    # the reference's run() has no per-iteration hook: re-run it with max_iters = k for snapshots
    snaps = {}
    for k in SNAP_ITERS:
        apm_k = ref_apm_from(inc, grpfile, values)
        if mask:
            apm_k.multiply(gtmask, axis=2)
            for h in range(H):
                apm_k.data[h].eliminate_zeros()
        ref_k = RefEM(apm_k)
        ref_k.prepare(pseudocount=pseudocount, lenfile=lenfile)
        ref_k.run(model=4, tol=0.0, max_iters=k, verbose=False)
        np.seterr(all="warn")
        snaps[k] = ref_k.allelic_expression.copy()
    ref.run(model=4, tol=tol, max_iters=max_iters, verbose=False)
    np.seterr(all="warn")

    def on_iter(i, theta, err):
        if i in snaps:
            beq(snaps[i], theta, f"{name} theta after iter {i}")
    n_it = orc.run(tol=tol, max_iters=max_iters, on_iter=on_iter)
    beq(ref.allelic_expression, orc.theta, f"{name} final theta")
    ref_counts = ref.probability.sum(axis=RefAPM.Axis.READ)
    beq(ref_counts, orc.expected_read_counts(), f"{name} expected read counts")
    ref_gene = np.asarray(ref.get_allelic_expression(at_group_level=True))
    beq(ref_gene, EMOracle.group_sums(orc.theta, inc.groups), f"{name} gene-level theta")
    ref_gene_counts = np.asarray(ref_counts * ref.grp_conv_mat)
    beq(ref_gene_counts, EMOracle.group_sums(orc.expected_read_counts(), inc.groups),
        f"{name} gene-level counts")

    # report text (digit format parity of the host-side writer)
    texts = {}
    final_theta = ref.allelic_expression.copy()
    for key, fn in (("isoforms_tpm", lambda p: ref.report_depths(filename=p, tpm=True)),
                    ("isoforms_counts", lambda p: ref.report_read_counts(filename=p)),
                    ("genes_tpm", lambda p: ref.report_depths(filename=p, tpm=True, grp_wise=True)),
                    ("genes_counts", lambda p: ref.report_read_counts(filename=p, grp_wise=True))):
        path = os.path.join(case_dir, key)
        fn(path)
        texts[key] = open(path).read()
    rep, _ = tpm_report_values(final_theta)
    first = texts["isoforms_tpm"].split("\n")[1].split("\t")[1:]
    assert first == [str(x) for x in rep[:, 0]], "tpm report restatement"

    out.update(
        num_rows=R, num_loci=L, num_haps=H, pseudocount=pseudocount, tol=tol, max_iters=max_iters,
        has_count=with_count, has_len=with_len, has_mask=bool(mask),
        count=inc.count if with_count else np.zeros(0),
        eff_len=eff_len if with_len else np.zeros((0, 0)),
        raw_length=inc.raw_length,
        gtmask=gtmask if mask else np.zeros((0, 0)),
        group_ptr=np.concatenate(([0], np.cumsum([len(g) for g in inc.groups]))).astype(np.int64),
        group_members=np.concatenate([np.asarray(g, dtype=np.int64) for g in inc.groups]),
        theta_final=final_theta, expected_counts=np.asarray(ref_counts),
        gene_theta=ref_gene, gene_counts=ref_gene_counts,
        num_iters=n_it, err_history=np.asarray(orc.err_history),
        **{f"theta_iter{k}": v for k, v in snaps.items()},
        **{f"text_{k}": np.array(v) for k, v in texts.items()},
    )
    # inputs are stored BEFORE masking (the product applies the mask itself)
    for h in range(H):
        out[f"indptr{h}"] = inc.indptr[h]
        out[f"indices{h}"] = inc.indices[h]
        if values is not None:
            out[f"values{h}"] = values[h]
    assert n_it == len(orc.err_history)
    np.savez_compressed(os.path.join(GOLD, f"em_{name}.npz"), **out)
    print(f"em_{name}: R={R} H={H} L={L} nnz={inc.nnz} iters={n_it} "
          f"err_last={orc.err_history[-1]:.4g}")


# ----------------------------------------------------------------------------- HMM

def hmm_case(name, H, genes_per_chrom, seed, len_minus_one, extra_fai_chrom=True, style="benign",
             expressed_fraction=0.5):
    prob = synth.make_hmm_problem(H=H, genes_per_chrom=genes_per_chrom, seed=seed,
                                  tprob_len_minus_one=len_minus_one, style=style,
                                  expressed_fraction=expressed_fraction)
    case_dir = os.path.join(WORK, name)
    os.makedirs(case_dir)
    # ref.fa.fai fixes the chromosome order; one chromosome without tprob exercises the skip
    with open(os.path.join(WORK, "ref.fa.fai"), "w") as fh:
        for c in prob.chroms:
            fh.write(f"{c}\t1000000\t0\t60\t61\n")
        if extra_fai_chrom:
            fh.write("MT\t16299\t0\t60\t61\n")
    expr_file = os.path.join(case_dir, "genes.tpm")
    with open(expr_file, "w") as fh:
        fh.write("locus\t" + "\t".join(prob.hap_names) + "\ttotal\n")
        for c in prob.chroms:
            for g in prob.gene_ids[c]:
                v = prob.expr[g]
                fh.write(g + "\t" + "\t".join(repr(float(x)) for x in v) + "\t" + repr(float(v.sum())) + "\n")
    tprob_file = os.path.join(case_dir, "tprob.npz")
    np.savez(tprob_file, **prob.tprob)
    avec_file = os.path.join(case_dir, "avecs.npz")
    np.savez(avec_file, **prob.avecs)
    gpos_file = os.path.join(case_dir, "gene_pos.npz")
    gpos = {}
    for c in prob.chroms:
        arr = np.zeros(len(prob.gene_ids[c]), dtype=[("f0", "U24"), ("f1", "i8")])
        arr["f0"] = prob.gene_ids[c]
        arr["f1"] = np.arange(len(arr)) * 1000
        gpos[c] = arr
    np.savez(gpos_file, **gpos)
    outbase = os.path.join(case_dir, "out")
    ref_gbrs_utils.reconstruct(expr_file, tprob_file, avec_file=avec_file, gpos_file=gpos_file,
                               expr_threshold=1.5, sigma=0.12, outbase=outbase)
    ref_gamma = np.load(outbase + ".genoprobs.npz")
    ref_states = np.load(outbase + ".genotypes.npz")
    tsv_text = open(outbase + ".genotypes.tsv").read()

    names = synth.diplotype_names(prob.hap_names)
    res = hmm_oracle.reconstruct_arrays(prob.hap_names, prob.chroms, prob.gene_ids, prob.tprob,
                                        prob.expr, prob.avecs, 1.5, 0.12)
    calls = {}
    for c in prob.chroms:
        beq(ref_gamma[c], res[c]["gamma"], f"{name} gamma {c}")
        beq(ref_states[c], np.array([names[s] for s in res[c]["states"]]), f"{name} states {c}")
        for g, s in zip(prob.gene_ids[c], res[c]["calls"]):
            if s >= 0:
                calls[g] = names[s]
    mine = "#Gene_ID\tDiplotype\n" + "".join(f"{g}\t{calls[g]}\n" for g in sorted(calls))
    assert mine == tsv_text, f"{name} genotypes.tsv"

    out = dict(num_haps=H, chroms=np.array(prob.chroms), len_minus_one=len_minus_one,
               expr_threshold=1.5, sigma=0.12, tsv_text=np.array(tsv_text),
               init_vec=hmm_oracle.init_vector(H))
    for c in prob.chroms:
        ids = prob.gene_ids[c]
        out[f"genes_{c}"] = np.array(ids)
        out[f"tprob_{c}"] = prob.tprob[c]
        out[f"expr_{c}"] = np.array([prob.expr[g] for g in ids])
        out[f"has_avec_{c}"] = np.array([g in prob.avecs for g in ids])
        out[f"avecs_{c}"] = np.array([prob.avecs.get(g, np.zeros((H, H))) for g in ids])
        for k in ("eprob", "alpha", "scaler", "beta", "gamma", "delta", "states", "calls"):
            out[f"{k}_{c}"] = res[c][k]
    np.savez_compressed(os.path.join(GOLD, f"hmm_{name}.npz"), **out)
    print(f"hmm_{name}: H={H} genes={genes_per_chrom} len-1={len_minus_one}")

# ----------------------------------------------------------------------------- alignment counts

def counts_case(name, R, H, L, seed, with_count, drop_rows=0, drop_every_nth_group=0):
    """`--report-alignment-counts` (AlignmentPropertyMatrix.py:389-459) at isoform level and, after
    _bundle_inline(reset=True) (:155-188), at gene level, from the reference's own methods."""
    inc = synth.make_em_problem(R=R, H=H, L=L, seed=seed, with_count=with_count, max_count=7)
    if drop_rows:
        rng = np.random.default_rng(seed + 1)
        dead = rng.choice(R, size=drop_rows, replace=False)
        for h in range(H):
            keep = ~np.isin(inc.indices[h], dead)
            loc = np.repeat(np.arange(L), np.diff(inc.indptr[h].astype(np.int64)))[keep]
            inc.indices[h] = inc.indices[h][keep]
            inc.indptr[h] = np.searchsorted(loc, np.arange(L + 1)).astype(np.uint32)
    if drop_every_nth_group:
        # genes missing from the group file: their isoforms belong to no group and vanish from the
        # bundled matrix (the product with grp_conv_mat, AlignmentPropertyMatrix.py:166-176)
        keep_g = [i for i in range(len(inc.groups)) if i % drop_every_nth_group != 1]
        inc.groups = [inc.groups[i] for i in keep_g]
        inc.group_names = [inc.group_names[i] for i in keep_g]
    case_dir = os.path.join(WORK, name)
    os.makedirs(case_dir)
    grpfile = os.path.join(case_dir, "g2t.tsv")
    with open(grpfile, "w") as fh:
        for g, members in zip(inc.group_names, inc.groups):
            fh.write(g + "\t" + "\t".join(inc.locus_names[l] for l in members) + "\n")
    out = dict(num_rows=R, num_loci=L, num_haps=H, has_count=with_count,
               count=inc.count if with_count else np.zeros(0),
               group_ptr=np.concatenate(([0], np.cumsum([len(g) for g in inc.groups]))).astype(np.int64),
               group_members=np.concatenate([np.asarray(g, dtype=np.int64) for g in inc.groups]))
    for h in range(H):
        out[f"indptr{h}"] = inc.indptr[h]
        out[f"indices{h}"] = inc.indices[h]
    apm = ref_apm_from(inc, grpfile)
    for level in ("isoforms", "genes"):
        if level == "genes":
            apm._bundle_inline(reset=True)
        out[f"{level}_aln"] = np.asarray(apm.count_alignments())
        out[f"{level}_uniq"] = np.asarray(apm.count_unique_reads(ignore_haplotype=False))
        out[f"{level}_locus_uniq"] = np.asarray(apm.count_unique_reads(ignore_haplotype=True))
        path = os.path.join(case_dir, level)
        apm.report_alignment_counts(filename=path)
        out[f"text_{level}"] = np.array(open(path).read())
    np.savez_compressed(os.path.join(GOLD, f"counts_{name}.npz"), **out)
    print(f"counts_{name}: R={R} H={H} L={L} nnz={inc.nnz}")

# ----------------------------------------------------------------------------- compress

def compress_case(name, R, H, L, seed, with_count, drop_rows=0, two_files=False):
    """`gbrs compress` (gbrs/emase_utils.py:22-107): the reference's own function is run on in-memory
    matrices.  Its equivalence-class loop (:46-103) is untouched; only the two file-I/O boundaries it
    crosses - `AlignmentPropertyMatrix(h5file=...)` and `.save(h5file=...)`, which need PyTables - are
    served from / captured into memory by a subclass of the reference's AlignmentPropertyMatrix (its
    `other=` copy constructor does the loading).  No PyTables call is imitated."""
    from gbrs.gbrs import emase_utils as ref_eu
    from oracle.compress_oracle import compress as oracle_compress
    inc = synth.make_em_problem(R=R, H=H, L=L, seed=seed, with_count=with_count, max_count=6)
    if drop_rows:
        rng = np.random.default_rng(seed + 1)
        dead = rng.choice(R, size=drop_rows, replace=False)
        for h in range(H):
            keep = ~np.isin(inc.indices[h], dead)
            loc = np.repeat(np.arange(L), np.diff(inc.indptr[h].astype(np.int64)))[keep]
            inc.indices[h] = inc.indices[h][keep]
            inc.indptr[h] = np.searchsorted(loc, np.arange(L + 1)).astype(np.uint32)
    source = ref_apm_from(inc, None)

    class MemoryAPM(RefAPM):
        loaded, saved = {}, {}

        def __init__(self, h5file=None, **kw):
            if h5file is not None:
                RefAPM.__init__(self, other=MemoryAPM.loaded[h5file])
            else:
                RefAPM.__init__(self, **kw)

        def save(self, h5file, **kw):
            MemoryAPM.saved[h5file] = self

    files = ["mem://a.h5", "mem://b.h5"] if two_files else ["mem://a.h5"]
    for f in files:
        MemoryAPM.loaded[f] = source
    keep_cls = ref_eu.AlignmentPropertyMatrix
    ref_eu.AlignmentPropertyMatrix = MemoryAPM
    try:
        ref_eu.compress(files, "mem://out.h5", "zlib")
    finally:
        ref_eu.AlignmentPropertyMatrix = keep_cls
    res = MemoryAPM.saved["mem://out.h5"]
    n_ref = res.num_reads
    # the oracle on the same rows (two files = the same reads twice, one after the other)
    if two_files:
        ip2, ix2 = [], []
        for h in range(H):
            cols = np.repeat(np.arange(L, dtype=np.int64), np.diff(inc.indptr[h].astype(np.int64)))
            rows = inc.indices[h].astype(np.int64)
            cols, rows = np.concatenate((cols, cols)), np.concatenate((rows, rows + R))
            order = np.lexsort((rows, cols))
            ix2.append(rows[order].astype(np.uint32))
            ip2.append(np.searchsorted(cols[order], np.arange(L + 1)).astype(np.uint32))
        cnt2 = None if inc.count is None else np.concatenate((inc.count, inc.count))
        n, ip, ix, counts = oracle_compress(2 * R, L, H, ip2, ix2, cnt2)
    else:
        n, ip, ix, counts = oracle_compress(R, L, H, inc.indptr, inc.indices, inc.count)
    assert n == n_ref, (n, n_ref)
    beq(res.count, counts, f"{name} EC counts")
    out = dict(num_rows=R, num_loci=L, num_haps=H, has_count=with_count, two_files=two_files,
               count=inc.count if with_count else np.zeros(0), num_ecs=n_ref, ec_count=np.asarray(res.count))
    for h in range(H):
        m = res.data[h].tocsc()
        m.sort_indices()
        beq(m.indptr, ip[h], f"{name} EC indptr h{h}")
        beq(m.indices, ix[h], f"{name} EC indices h{h}")
        assert (m.data == 1).all()
        out[f"indptr{h}"] = inc.indptr[h]
        out[f"indices{h}"] = inc.indices[h]
        out[f"ec_indptr{h}"] = m.indptr.astype(np.uint32)
        out[f"ec_indices{h}"] = m.indices.astype(np.uint32)
    np.savez_compressed(os.path.join(GOLD, f"compress_{name}.npz"), **out)
    print(f"compress_{name}: R={R} H={H} L={L} -> {n_ref} ECs")

# ----------------------------------------------------------------------------- interpolate / export

def postproc_case(name, hmm_name, n_grid):
    """gbrs_utils.interpolate and gbrs_utils.export run unmodified on a genoprobs file taken from
    an HMM golden; the oracle restatement must reproduce them bit for bit."""
    from oracle import postproc_oracle
    g = np.load(os.path.join(GOLD, f"hmm_{hmm_name}.npz"))
    chroms = [str(c) for c in g["chroms"]]
    H = int(g["num_haps"])
    strains = [chr(65 + h) for h in range(H)]
    case_dir = os.path.join(WORK, name)
    os.makedirs(case_dir)
    with open(os.path.join(WORK, "ref.fa.fai"), "w") as fh:
        for c in chroms:
            fh.write(f"{c}\t1000000\t0\t60\t61\n")
    rng = np.random.default_rng(77)
    gpos, gamma, xg, grid = {}, {}, {}, {}
    for c in chroms:
        n = g[f"gamma_{c}"].shape[1]
        pos = np.sort(rng.uniform(0.5, 90.0, size=n))
        arr = np.zeros(n, dtype=[("f0", "U24"), ("f1", "f8")])
        arr["f0"] = g[f"genes_{c}"]
        arr["f1"] = pos
        gpos[c], gamma[c], xg[c] = arr, g[f"gamma_{c}"], pos
        grid[c] = np.sort(np.concatenate(([0.0, pos[0], pos[-1]], rng.uniform(0.0, pos[-1] + 5.0, size=n_grid - 3))))
    gpos_file = os.path.join(case_dir, "gpos.npz"); np.savez(gpos_file, **gpos)
    gp_file = os.path.join(case_dir, "genoprobs.npz"); np.savez(gp_file, **gamma)
    grid_file = os.path.join(case_dir, "grid.txt")
    with open(grid_file, "w") as fh:
        fh.write("marker\tchr\tbp\tcM\n")
        k = 0
        for c in chroms:
            for x in grid[c]:
                fh.write(f"m{k}\t{c}\t{int(x * 1e6)}\t{repr(float(x))}\n")
                k += 1
    out_i = os.path.join(case_dir, "interp.npz")
    ref_gbrs_utils.interpolate(gp_file, grid_file=grid_file, gpos_file=gpos_file, output_file=out_i)
    ref_i = np.load(out_i)
    out_e = os.path.join(case_dir, "export.tsv")
    ref_gbrs_utils.export(out_i, strains, grid_file=grid_file, output_file=out_e)
    out = dict(num_haps=H, chroms=np.array(chroms), export_text=np.array(open(out_e).read()))
    rows = []
    for c in chroms:
        mine = postproc_oracle.interpolate(xg[c], gamma[c], grid[c])
        beq(ref_i[c], mine, f"{name} interpolate {c}")
        out[f"xgene_{c}"] = xg[c]; out[f"gamma_{c}"] = gamma[c]; out[f"grid_{c}"] = grid[c]
        out[f"interp_{c}"] = ref_i[c]
        rows.append(ref_i[c].transpose())
    dos = postproc_oracle.dosage(np.vstack(rows), H)
    ref_dos = np.loadtxt(out_e, skiprows=1, delimiter="\t")
    if np.max(np.abs(ref_dos - dos)) > 5.1e-7:
        raise AssertionError(f"{name}: export restatement differs")
    out["dosage"] = dos
    np.savez_compressed(os.path.join(GOLD, f"postproc_{name}.npz"), **out)
    print(f"postproc_{name}: from hmm_{hmm_name}, {n_grid} grid points per chromosome")


def main():
    os.makedirs(GOLD, exist_ok=True)
    only = sys.argv[sys.argv.index("--only") + 1] if "--only" in sys.argv else None
    try:
        if only in (None, "counts"):
            counts_case("h8_count", 1500, 8, 80, 41, True)
            counts_case("h8_plain_emptyrows", 2000, 8, 100, 42, False, drop_rows=100)
            counts_case("h2_count", 1200, 2, 60, 43, True)
            counts_case("h8_ungrouped", 1500, 8, 80, 44, True, drop_every_nth_group=4)
        if only == "counts":
            return
        if only in (None, "compress"):
            compress_case("h8_plain", 2500, 8, 50, 51, False)
            compress_case("h8_count_emptyrows", 1800, 8, 40, 52, True, drop_rows=60)
            compress_case("h2_two_files", 1200, 2, 30, 53, False, two_files=True)
            compress_case("h16_count", 700, 16, 25, 54, True)
        if only == "compress":
            return
        if only == "postproc":
            postproc_case("h8", "h8_full", 25)
            postproc_case("h4", "h4_full", 12)
            return
        if only in (None, "hmm_do"):
            # recombination-shaped tables: entries down to ~1e-32, structural zeros (-inf), near-deterministic
            # stretches; the sparse-expression case has most genes under the expression threshold
            hmm_case("h8_do_full", 8, [70, 45, 30], 35, False, style="do")
            hmm_case("h8_do_short", 8, [70, 45, 30], 36, True, style="do")
            hmm_case("h8_do_sparse_expr", 8, [90, 40], 37, False, style="do", expressed_fraction=0.12)
            hmm_case("h4_do_full", 4, [50, 21], 38, False, style="do")
        if only == "hmm_do":
            return
        if only in (None, "em_mask_values"):
            # stored values under a `-G` mask: multiply(gtmask, axis=2) keeps the surviving entries' values
            em_case("h8_mask_values", 1600, 8, 70, 24,   True,  True,  0.0, True,  1e-4, 999, with_values=True)
        if only == "em_mask_values":
            return
        #        name            R     H  L    seed  count  len    pc   mask   tol   max
        em_case("h2_plain",      1500, 2, 60,  11,   False, False, 0.0, False, 1e-4, 999)
        em_case("h2_len",        1500, 2, 60,  12,   False, True,  0.0, False, 1e-4, 999)
        em_case("h8_len",        3000, 8, 120, 13,   False, True,  0.0, False, 1e-4, 999)
        em_case("h8_count_len",  1200, 8, 90,  14,   True,  True,  0.0, False, 1e-4, 999)
        em_case("h8_pseudo",     2000, 8, 100, 15,   True,  True,  0.5, False, 1e-4, 999)
        em_case("h8_mask",       2500, 8, 80,  16,   False, True,  0.0, True,  1e-4, 999)
        em_case("h8_mask_count", 1500, 8, 80,  17,   True,  False, 0.5, True,  1e-4, 999)
        em_case("h1_len",        800,  1, 50,  18,   False, True,  0.0, False, 1e-4, 999)
        em_case("h8_emptyrows",  2000, 8, 100, 19,   False, True,  0.0, False, 1e-4, 999,
                drop_rows=150)
        em_case("h16_len",       2000, 16, 70, 20,   True,  True,  0.0, False, 1e-4, 999)
        em_case("h8_maxiter",    2000, 8, 100, 21,   False, True,  0.0, False, 0.0,  7)
        em_case("h8_values",     1800, 8, 90,  22,   True,  True,  0.0, False, 1e-4, 999, with_values=True)
        em_case("h2_values_pc",  1500, 2, 60,  23,   False, True,  0.5, False, 1e-4, 999, with_values=True)
        hmm_case("h8_full", 8, [40, 25, 33], 31, False)
        hmm_case("h8_short", 8, [40, 25, 33], 32, True)
        hmm_case("h4_full", 4, [30, 12], 33, False)
        hmm_case("h2_short", 2, [20, 1 + 1], 34, True)
        if only == "hmm":
            return
        postproc_case("h8", "h8_full", 25)
        postproc_case("h4", "h4_full", 12)
    finally:
        shutil.rmtree(WORK, ignore_errors=True)


if __name__ == "__main__":
    main()
