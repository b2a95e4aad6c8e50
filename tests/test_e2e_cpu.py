"""The end-to-end CPU baseline driver (oracle/e2e_oracle.py) and the file writers of scripts/e2e_bench.py
on a small sample (no GPU): the oracle workflow reproduces the reference golden's report text."""
import os
import sys

import numpy as np

from conftest import ROOT, em_case_inputs, golden_files, load_golden

sys.path.insert(0, os.path.join(ROOT, "scripts"))


def test_oracle_quantify_and_reconstruct_from_files(tmp_path):
    import e2e_bench
    from gbrs_amd.alignment import AlignmentPropertyMatrix
    from oracle import e2e_oracle
    g = load_golden([p for p in golden_files("em") if p.endswith("em_h8_len.npz")][0])
    R, L, H, indptr, indices, count, eff_len, groups, gtmask = em_case_inputs(g)
    starts = [min(m) for m in groups]
    lname, hname, gname, grp, lens = e2e_bench.write_support_files(str(tmp_path), L, H, starts, eff_len[0])
    apm = AlignmentPropertyMatrix(shape=(L, H, R), indptr=indptr, indices=indices, haplotype_names=hname,
                                  locus_names=lname)
    aln = str(tmp_path / "a.npz")
    apm.save_npz(aln)
    t = e2e_oracle.quantify(aln, grp, lens, str(tmp_path / "cpu"))
    assert t["em_iterations"] == int(g["num_iters"]) and t["rows"] == R
    for key, name in (("text_isoforms_tpm", "isoforms.tpm"), ("text_genes_tpm", "genes.tpm"),
                      ("text_isoforms_counts", "isoforms.expected_read_counts"),
                      ("text_genes_counts", "genes.expected_read_counts")):
        assert open(tmp_path / f"cpu.multiway.{name}").read() == str(g[key]), name
    # reconstruct on that genes.tpm with synthetic tables
    paths, n_genes = e2e_bench.write_reconstruct_inputs(str(tmp_path), str(tmp_path / "cpu.multiway.genes.tpm"))
    assert n_genes == len(groups)
    t2 = e2e_oracle.reconstruct(str(tmp_path / "cpu.multiway.genes.tpm"), paths["tprob"], paths["avecs"], paths["gpos"],
                                paths["fai"], str(tmp_path / "rec"))
    assert t2["genes"] == n_genes
    gam = np.load(tmp_path / "rec.genoprobs.npz")
    for c in gam.files:
        if gam[c].shape[1]:
            np.testing.assert_allclose(gam[c].sum(axis=0), 1.0, rtol=1e-10)
    assert open(tmp_path / "rec.genotypes.tsv").readline() == "#Gene_ID\tDiplotype\n"


def test_oracle_quantify_with_genotype_file(tmp_path):
    """The diploid pass of the one-core baseline (`quantify -G`, gbrs/emase_utils.py:240-273) reproduces the numbers of
    the reference golden made with the same mask, and writes the calls into the notes column."""
    import e2e_bench
    from gbrs_amd.alignment import AlignmentPropertyMatrix
    from oracle import e2e_oracle
    g = load_golden([p for p in golden_files("em") if p.endswith("em_h8_mask.npz")][0])
    R, L, H, indptr, indices, count, eff_len, groups, gtmask = em_case_inputs(g)
    starts = [min(m) for m in groups]
    lname, hname, gname, grp, lens = e2e_bench.write_support_files(str(tmp_path), L, H, starts, eff_len[0])
    apm = AlignmentPropertyMatrix(shape=(L, H, R), indptr=indptr, indices=indices, haplotype_names=hname,
                                  locus_names=lname)
    aln = str(tmp_path / "a.npz")
    apm.save_npz(aln)
    gt = tmp_path / "gt.tsv"
    calls = {}
    with open(gt, "w") as fh:
        fh.write("#Gene_ID\tDiplotype\n")
        for i, mem in enumerate(groups):
            hs = np.flatnonzero(gtmask[:, mem[0]])
            calls[gname[i]] = "".join(hname[h] for h in (hs if len(hs) == 2 else [hs[0], hs[0]]))
            fh.write(f"{gname[i]}\t{calls[gname[i]]}\n")
    t = e2e_oracle.quantify(aln, grp, lens, str(tmp_path / "cpu"), str(gt))
    assert t["em_iterations"] == int(g["num_iters"]) and t["entries_in_em"] < t["entries"] and "mask" in t
    for key, name in (("text_isoforms_tpm", "isoforms.tpm"), ("text_genes_tpm", "genes.tpm"),
                      ("text_isoforms_counts", "isoforms.expected_read_counts"),
                      ("text_genes_counts", "genes.expected_read_counts")):
        got = open(tmp_path / f"cpu.diploid.{name}").read().splitlines()
        exp = str(g[key]).splitlines()
        assert got[0] == exp[0] + "\tnotes" and len(got) == len(exp)
        assert [ln.rsplit("\t", 1)[0] for ln in got[1:]] == exp[1:], name
        if name.startswith("genes"):
            assert all(ln.rsplit("\t", 1)[1] == calls[ln.split("\t", 1)[0]] for ln in got[1:])
