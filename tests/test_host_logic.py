"""Host-side mirror logic (no GPU): containers, file parsing, report text, CLI surface."""
import numpy as np
import pytest

from conftest import em_case_inputs, golden_files, load_golden


def golden(name):
    return load_golden([p for p in golden_files("em") if p.endswith(f"em_{name}.npz")][0])


def make_apm(g):
    from gbrs_amd.alignment import AlignmentPropertyMatrix
    R, L, H, indptr, indices, count, eff_len, groups, gtmask = em_case_inputs(g)
    apm = AlignmentPropertyMatrix(shape=(L, H, R), indptr=indptr, indices=indices, count=count,
                                  haplotype_names=[chr(65 + h) for h in range(H)],
                                  locus_names=[f"T{l:07d}" for l in range(L)])
    return apm, groups, gtmask, eff_len


def test_report_text_matches_reference_digits():
    """The writer reproduces the reference's TSV byte for byte given the reference's numbers."""
    from gbrs_amd.em import write_locus_table
    import io, os, tempfile
    g = golden("h8_count_len")
    H = int(g["num_haps"])
    hn = [chr(65 + h) for h in range(H)]
    ln = [f"T{l:07d}" for l in range(int(g["num_loci"]))]
    gn = [f"G{i:07d}" for i in range(len(g["group_ptr"]) - 1)]
    theta = g["theta_final"] * (1000000.0 / g["theta_final"].sum())       # report_depths(tpm=True)
    cases = [("text_isoforms_tpm", ln, theta),
             ("text_isoforms_counts", ln, g["expected_counts"]),
             ("text_genes_counts", gn, g["gene_counts"])]
    for key, names, vals in cases:
        with tempfile.TemporaryDirectory() as d:
            p = os.path.join(d, "r.tsv")
            write_locus_table(p, hn, names, vals)
            assert open(p).read() == str(g[key]), key
    # gene TPM: the reference first rescales theta in place (isoform report), then groups, then rescales
    from oracle.em_oracle import EMOracle
    R, L, H, indptr, indices, count, eff_len, groups, gtmask = em_case_inputs(g)
    gene = np.asfortranarray(EMOracle.group_sums(theta, groups))      # memory order of scipy's product
    gene = gene * (1000000.0 / gene.sum())
    with tempfile.TemporaryDirectory() as d:
        p = os.path.join(d, "r.tsv")
        write_locus_table(p, hn, gn, gene)
        assert open(p).read() == str(g["text_genes_tpm"])


def test_mask_haplotype_loci_matches_oracle():
    from oracle.em_oracle import EMOracle
    g = golden("h8_mask")
    apm, groups, gtmask, _ = make_apm(g)
    R, L, H, indptr, indices, count, eff_len, _, _ = em_case_inputs(g)
    o = EMOracle(R, L, H, indptr, indices, count)
    o.apply_genotype_mask(gtmask)
    apm.mask_haplotype_loci(gtmask)
    for h in range(H):
        np.testing.assert_array_equal(apm.indptr[h], o.indptr[h])
        np.testing.assert_array_equal(apm.indices[h], o.indices[h])


def test_length_file_and_groups(tmp_path):
    from gbrs_amd.alignment import AlignmentPropertyMatrix
    from gbrs_amd.em import read_length_file
    g = golden("h8_len")
    apm, groups, _, eff_len = make_apm(g)
    lf = tmp_path / "len.tsv"
    with open(lf, "w") as fh:
        for l, name in enumerate(apm.lname):
            for hn in apm.hname:
                fh.write(f"{name}_{hn}\t{int(g['raw_length'][l])}\n")
    np.testing.assert_array_equal(read_length_file(apm, str(lf), 100), eff_len)
    gf = tmp_path / "g2t.tsv"
    with open(gf, "w") as fh:
        for i, mem in enumerate(groups):
            fh.write(f"G{i:07d}\t" + "\t".join(apm.lname[m] for m in mem) + "\n")
    apm.load_groups(str(gf))
    gp, gm = apm.group_csr()
    np.testing.assert_array_equal(gp, g["group_ptr"])
    np.testing.assert_array_equal(gm, g["group_members"])
    with open(gf, "a") as fh:
        fh.write("GX\tNOPE\n")
    with pytest.raises(KeyError):
        apm.load_groups(str(gf))
    # npz mirror round trip
    apm.save_npz(str(tmp_path / "a.npz"))
    b = AlignmentPropertyMatrix(npzfile=str(tmp_path / "a.npz"))
    assert b.shape == apm.shape and b.hname == apm.hname and b.lname == apm.lname
    for h in range(apm.num_haplotypes):
        np.testing.assert_array_equal(b.indices[h], apm.indices[h])


def test_container_validation():
    from gbrs_amd.alignment import AlignmentPropertyMatrix
    with pytest.raises(RuntimeError, match="three positive integers"):
        AlignmentPropertyMatrix(shape=(3, 0, 4))
    with pytest.raises(RuntimeError, match="does not match"):
        AlignmentPropertyMatrix(shape=(3, 2, 4), haplotype_names=["A"])
    with pytest.raises(RuntimeError, match="Malformed"):
        AlignmentPropertyMatrix(shape=(2, 1, 4), indptr=[np.array([0, 1, 3])], indices=[np.array([0])])


def test_cli_surface_and_error_swallowing(tmp_path, capsys):
    from gbrs_amd import cli
    p = cli.build_parser()
    f = tmp_path / "x.npz"
    f.write_bytes(b"")
    a = p.parse_args(["quantify", "-i", str(f)])
    assert (a.outbase, a.multiread_model, a.pseudocount, a.max_iters, a.tolerance) == \
        ("gbrs.quantified", 4, 0.0, 999, 0.0001)
    assert not a.report_alignment_counts and not a.report_posterior
    r = p.parse_args(["reconstruct", "-e", str(f), "-t", str(f)])
    assert (r.expr_threshold, r.sigma, r.outbase, r.avec_file, r.gpos_file) == (1.5, 0.12, None, None, None)
    # like the reference, a failing subcommand logs and still exits 0
    assert cli.main(["quantify", "-i", str(f), "-M", "9"]) == 0
    assert cli.main(["quantify", "-i", str(f)]) == 0


def pack_mask(gtmask):
    """(H x L) 0/1 mask -> uint32[L] with bit h set where gtmask[h, l] != 0."""
    H = gtmask.shape[0]
    return ((gtmask != 0).astype(np.uint32) << np.arange(H, dtype=np.uint32)[:, None]).sum(axis=0).astype(np.uint32)


def test_genotype_mask_parsing(tmp_path):
    from gbrs_amd.quantify import diplotype_mask, read_genotype_calls, read_genotype_table
    g = golden("h8_mask")
    apm, groups, gtmask, _ = make_apm(g)
    apm.groups = groups
    apm.gname = np.array([f"G{i:07d}" for i in range(len(groups))])
    apm.num_groups = len(groups)
    gt = tmp_path / "gt.tsv"
    with open(gt, "w") as fh:
        fh.write("#Gene_ID\tDiplotype\n")
        for i, mem in enumerate(groups):
            hs = np.flatnonzero(gtmask[:, mem[0]])
            code = "".join(apm.hname[h] for h in (hs if len(hs) == 2 else [hs[0], hs[0]]))
            fh.write(f"G{i:07d}\t{code}\n")
    calls = read_genotype_calls(str(gt))
    table = read_genotype_table(str(gt))
    assert dict(zip(*table)) == calls and len(table[0]) == len(groups)
    m, cg, ct = diplotype_mask(apm, table)
    np.testing.assert_array_equal(m, pack_mask(gtmask))
    assert m.dtype == np.uint32
    assert cg["G0000000"] is not None and ct[apm.lname[groups[0][0]]] == cg["G0000000"]
    assert len(cg) == len(groups) and len(ct) == apm.num_loci
    # the pending device mask, carried out on the host, is the reference's multiply + eliminate_zeros
    from oracle.em_oracle import EMOracle
    R, L, H, indptr, indices, count, eff_len, _, _ = em_case_inputs(g)
    o = EMOracle(R, L, H, indptr, indices, count)
    o.apply_genotype_mask(gtmask)
    apm.set_haplotype_mask(m)
    assert apm.nnz == sum(len(i) for i in o.indices) and apm.haplotype_mask is None
    for h in range(H):
        np.testing.assert_array_equal(apm.indptr[h], o.indptr[h])
        np.testing.assert_array_equal(apm.indices[h], o.indices[h])
    # a gene missing from the file keeps the note None, as the reference's dict.fromkeys tables do
    m2, cg2, ct2 = diplotype_mask(apm, {k: v for k, v in calls.items() if k != "G0000001"})
    assert cg2["G0000001"] is None and ct2[apm.lname[groups[1][0]]] is None
    assert not m2[groups[1]].any()
    # the notes column in one piece is what the per-name look-ups give
    blob, off = ct2.aligned_blob(apm.lname)
    assert [blob[off[k]:off[k + 1]].decode() for k in range(apm.num_loci)] == [str(ct2[t]) for t in apm.lname]
    assert ct2.aligned_blob(list(apm.lname)) is None          # another list object: the writer falls back to look-ups


def test_genotype_table_forms(tmp_path):
    """Lines the one-pass split declines (extra columns, trailing blanks, no final newline) go through the
    reference's per-line rule; a gene listed twice contributes both lines to the mask (gbrs/emase_utils.py:262-268:
    the mask accumulates, the notes keep the later call)."""
    from gbrs_amd.quantify import diplotype_mask, read_genotype_table
    g = golden("h8_mask")
    apm, groups, gtmask, _ = make_apm(g)
    apm.groups = groups
    apm.gname = np.array([f"G{i:07d}" for i in range(len(groups))])
    apm.num_groups = len(groups)
    plain = tmp_path / "plain.tsv"
    plain.write_text("#Gene_ID\tDiplotype\n#second comment\nG0000000\tAB\nG0000001\tCC\nG0000000\tGH")
    assert read_genotype_table(str(plain)) == (["G0000000", "G0000001", "G0000000"], ["AB", "CC", "GH"])
    odd = tmp_path / "odd.tsv"
    odd.write_text("#Gene_ID\tDiplotype\nG0000000\tAB\textra\nG0000001\tCC  \n#G0000002\tDD\nG0000000\tGH\n")
    genes, calls = read_genotype_table(str(odd))
    assert genes == ["G0000000", "G0000001", "#G0000002", "G0000000"] and calls == ["AB", "CC", "DD", "GH"]
    from gbrs_amd.quantify import genotype_mask_from_file
    native = genotype_mask_from_file(apm, str(plain))                # the library's parser: same mask, same notes
    allowed, cg, ct = diplotype_mask(apm, read_genotype_table(str(plain)))
    assert native is not None
    np.testing.assert_array_equal(native[0], allowed)
    for mine, theirs, names in ((native[1], cg, apm.gname), (native[2], ct, apm.lname)):
        assert [mine[k] for k in names] == [theirs[k] for k in names]
        blob_a, off_a = mine.aligned_blob(names)
        blob_b, off_b = theirs.aligned_blob(names)
        assert blob_a == blob_b and np.array_equal(off_a, off_b)
    assert genotype_mask_from_file(apm, str(odd)) is None            # a `#` line after the header: not a gene it knows
    spaced = tmp_path / "spaced.tsv"
    spaced.write_text("#Gene_ID\tDiplotype\nG0000000\tAB\textra\nG0000001\tCC  \n")
    sp = genotype_mask_from_file(apm, str(spaced))
    assert sp is not None and sp[1]["G0000001"] == "CC" and set(sp[0][groups[0]].tolist()) == {0b11}
    assert set(allowed[groups[0]].tolist()) == {0b11000011} and set(allowed[groups[1]].tolist()) == {0b100}
    rest = np.setdiff1d(np.arange(apm.num_loci), np.concatenate([groups[0], groups[1]]))
    assert not allowed[rest].any()
    assert cg["G0000000"] == "GH" and ct[apm.lname[groups[0][0]]] == "GH" and cg["G0000002"] is None
    with pytest.raises(KeyError):
        diplotype_mask(apm, (["G0000000"], ["AZ"]))            # unknown haplotype letter
    with pytest.raises(KeyError):
        diplotype_mask(apm, (["nope"], ["AB"]))                # gene the group file does not list
    bad = tmp_path / "bad.tsv"
    bad.write_text("G0000000\n")
    assert genotype_mask_from_file(apm, str(bad)) is None
    with pytest.raises(ValueError):
        read_genotype_table(str(bad))
    # as many tabs as lines, but two on the first line and none on the second: the reference raises on the one-field
    # line (`g, gt = item[:2]`); the one-pass splitter must not pair the fields up across lines
    bad.write_text("G0000000\tAB\tG0000001\nCD\n")
    assert genotype_mask_from_file(apm, str(bad)) is None
    with pytest.raises(ValueError):
        read_genotype_table(str(bad))
    for text in ("G0000000\tAZ\n", "nope\tAB\n", "G0000000\tABCDEFGHA\n", "G0000000\tA\xc3\xa9\n"):
        bad.write_text(text)
        assert genotype_mask_from_file(apm, str(bad)) is None        # left to the line-by-line path and its errors


def test_read_gene_tpm_native_and_fallback_agree(tmp_path):
    """genes.tpm through the library's number parser (plain tables) and through the Python path (a line it declines)."""
    import numpy as np
    from gbrs_amd import hmm
    rng = np.random.default_rng(5)
    vals = rng.gamma(1.0, 5.0, size=(500, 8)) * (rng.random((500, 8)) < 0.5)
    p = tmp_path / "genes.tpm"
    with open(p, "w") as fh:
        fh.write("locus\t" + "\t".join("ABCDEFGH") + "\ttotal\n")
        for i, row in enumerate(vals):
            fh.write(f"G{i:05d}\t" + "\t".join(repr(float(x)) for x in row) + "\t" + repr(float(row.sum())) + "\n")
    haps, rows, table = hmm.read_gene_tpm(str(p))
    assert haps == list("ABCDEFGH") and rows["G00007"] == 7
    np.testing.assert_array_equal(table, vals)
    with open(p, "a") as fh:                               # "inf" is not what from_chars takes here: Python path
        fh.write("GINF\t" + "\t".join(["inf"] * 8) + "\tinf\n")
    haps2, rows2, table2 = hmm.read_gene_tpm(str(p))
    assert rows2["GINF"] == 500 and np.isinf(table2[-1]).all()
    np.testing.assert_array_equal(table2[:-1], vals)
    with open(p, "a") as fh:
        fh.write("GSHORT\t1.0\n")
    import pytest
    with pytest.raises(ValueError):
        hmm.read_gene_tpm(str(p))
