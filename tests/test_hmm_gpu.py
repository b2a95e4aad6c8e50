"""HIP HMM path vs the reference goldens (needs an MI355X)."""
import numpy as np
import pytest

from conftest import golden_files, hmm_case_inputs, load_golden, viterbi_decision_margins

pytestmark = pytest.mark.gpu


def build(c):
    from gbrs_amd.hmm import DiplotypeHMM
    chroms = c["chroms"]
    return DiplotypeHMM(c["H"], chroms, [len(c["genes"][ch]) for ch in chroms], [c["tprob"][ch] for ch in chroms])


@pytest.mark.parametrize("path", golden_files("hmm"), ids=lambda p: p.split("/")[-1][:-4])
def test_hmm_matches_reference_golden(path):
    g = load_golden(path)
    c = hmm_case_inputs(g)
    chroms = c["chroms"]
    hmm = build(c)
    hmm.set_expression([c["expr"][ch] for ch in chroms], [c["avecs"][ch] for ch in chroms],
                       [c["has_avec"][ch] for ch in chroms], float(g["expr_threshold"]), float(g["sigma"]))
    hmm.run()
    want = ("gamma", "states", "calls", "alpha", "beta", "delta", "scaler", "eprob")
    for ci, ch in enumerate(chroms):
        r = hmm.get(ci, want=want)
        # genotype calls and the ordered Viterbi path: bit-exact
        np.testing.assert_array_equal(r["states"], g[f"states_{ch}"])
        np.testing.assert_array_equal(r["calls"], g[f"calls_{ch}"])
        # log-domain quantities: absolute tolerance on the log value (== relative on the probability)
        np.testing.assert_allclose(r["eprob"], g[f"eprob_{ch}"], rtol=1e-10, atol=1e-10)
        np.testing.assert_allclose(r["alpha"], g[f"alpha_{ch}"], rtol=1e-9, atol=1e-9)
        np.testing.assert_allclose(r["scaler"], g[f"scaler_{ch}"], rtol=1e-9, atol=1e-9)
        np.testing.assert_allclose(r["beta"], g[f"beta_{ch}"], rtol=1e-9, atol=1e-9)
        np.testing.assert_allclose(r["delta"], g[f"delta_{ch}"], rtol=1e-9, atol=1e-9)
        # the calls are argmax decisions: the device delta must sit far inside the smallest decision gap
        gap = viterbi_decision_margins(c["tprob"][ch], g[f"delta_{ch}"]).min()
        assert np.max(np.abs(r["delta"] - g[f"delta_{ch}"])) < 1e-6 * gap
        # posteriors: north-star 1e-4 relative; held to 1e-8
        np.testing.assert_allclose(r["gamma"], g[f"gamma_{ch}"], rtol=1e-8, atol=1e-300)
        np.testing.assert_allclose(r["gamma"].sum(axis=0), 1.0, rtol=1e-12)
    hmm.close()


# the Viterbi values of the blocked scan: by rank convergence (default), by max-plus block operators (round 3), and with a
# tolerance no block can meet, so that every chromosome takes the fallback chain behind the fix-up
DELTA_ROUTES = {"rank": {}, "operators": {"GBRS_TUNING_HMM_DELTA_SPEC": "0"}, "fallback": {"GBRS_TUNING_HMM_DELTA_TOL": "-1"}}


@pytest.mark.parametrize("delta_route", list(DELTA_ROUTES))
@pytest.mark.parametrize("block_genes", [2, 5, 9])
@pytest.mark.parametrize("path", [p for p in golden_files("hmm") if "h8" in p], ids=lambda p: p.split("/")[-1][:-4])
def test_blocked_scan_matches_reference_golden(path, block_genes, delta_route, monkeypatch):
    """The blocked scan of the 36-state recursions (hmm_blocked.inc: block transfer operators on MFMA, a sequential
    combine, the chain kernels inside all blocks at once; Viterbi values by rank convergence or max-plus operators) with
    blocks of 2, 5 and 9 genes, so that the goldens' short chromosomes are cut many times (and blocks of 2 genes cannot
    converge: those chromosomes take the fallback): the same tolerances as the unblocked run, calls bit-exact."""
    monkeypatch.setenv("GBRS_TUNING_HMM_BLOCK_GENES", str(block_genes))
    monkeypatch.setenv("GBRS_TUNING_HMM_BLOCKED", "2")
    for k, v in DELTA_ROUTES[delta_route].items():
        monkeypatch.setenv(k, v)
    g = load_golden(path)
    c = hmm_case_inputs(g)
    chroms = c["chroms"]
    hmm = build(c)
    hmm.set_expression([c["expr"][ch] for ch in chroms], [c["avecs"][ch] for ch in chroms],
                       [c["has_avec"][ch] for ch in chroms], float(g["expr_threshold"]), float(g["sigma"]))
    hmm.run()
    want = ("gamma", "states", "calls", "alpha", "beta", "delta", "scaler")
    for ci, ch in enumerate(chroms):
        r = hmm.get(ci, want=want)
        np.testing.assert_array_equal(r["states"], g[f"states_{ch}"])
        np.testing.assert_array_equal(r["calls"], g[f"calls_{ch}"])
        np.testing.assert_allclose(r["alpha"], g[f"alpha_{ch}"], rtol=1e-9, atol=1e-9)
        np.testing.assert_allclose(r["scaler"], g[f"scaler_{ch}"], rtol=1e-9, atol=1e-9)
        np.testing.assert_allclose(r["beta"], g[f"beta_{ch}"], rtol=1e-9, atol=1e-9)
        np.testing.assert_allclose(r["delta"], g[f"delta_{ch}"], rtol=1e-9, atol=1e-9)
        gap = viterbi_decision_margins(c["tprob"][ch], g[f"delta_{ch}"]).min()
        assert np.max(np.abs(r["delta"] - g[f"delta_{ch}"])) < 1e-6 * gap
        np.testing.assert_allclose(r["gamma"], g[f"gamma_{ch}"], rtol=1e-8, atol=1e-300)
    inf = hmm.info()
    if delta_route == "operators":
        assert inf.last_delta_blocks == 0 and inf.last_delta_fallbacks == 0
    elif delta_route == "fallback":
        many = sum(1 for n in hmm.n_genes if min(64, int(n) // block_genes) >= 2)      # chromosomes cut into 2+ blocks
        assert inf.last_delta_blocks == 0 and inf.last_delta_fallbacks == many
    hmm.close()


@pytest.mark.parametrize("style,expressed", [("benign", 0.5), ("do", 0.5), ("do", 0.1)])
def test_rank_convergence_delta_at_size(style, expressed, monkeypatch):
    """Viterbi values by rank convergence on three long chromosomes, SURVEY's tables and the DO-like ones (recombination
    1e-15 .. 1e-2 per interval, structural zeros = -inf entries) with half and a tenth of the haplotypes expressed: the
    blocks really converge (no fallback, fix-ups far shorter than a block), delta equals the sequential chain's and the
    oracle's, the path is identical."""
    from gbrs_amd import synth
    from gbrs_amd.hmm import DiplotypeHMM
    from oracle import hmm_oracle
    prob = synth.make_hmm_problem(H=8, genes_per_chrom=[2600, 1700, 1100], style=style, expressed_fraction=expressed)
    chroms = prob.chroms
    ex = [np.array([prob.expr[g] for g in prob.gene_ids[c]]) for c in chroms]
    ha = [np.array([g in prob.avecs for g in prob.gene_ids[c]], dtype=np.uint8) for c in chroms]
    av = [np.array([prob.avecs.get(g, np.zeros((8, 8))) for g in prob.gene_ids[c]]) for c in chroms]
    res = {}
    for mode in ("0", "2"):
        monkeypatch.setenv("GBRS_TUNING_HMM_BLOCKED", mode)
        hmm = DiplotypeHMM(8, chroms, [len(prob.gene_ids[c]) for c in chroms], [prob.tprob[c] for c in chroms])
        hmm.set_expression(ex, av, ha, 1.5, 0.12)
        hmm.run()
        res[mode] = [hmm.get(ci, want=("states", "calls", "delta")) for ci in range(3)]
        inf = hmm.info()
        if mode == "2":
            assert inf.last_delta_fallbacks == 0
            assert inf.last_delta_blocks == sum(max(1, min(64, n // 40)) - 1 for n in (2600, 1700, 1100))
            assert 1 <= inf.last_delta_longest_fixup <= 41
        else:
            assert inf.last_delta_blocks == 0
        hmm.close()
    iv = hmm_oracle.init_vector(8)
    for ci, (a, b) in enumerate(zip(res["0"], res["2"])):
        np.testing.assert_array_equal(a["states"], b["states"])
        np.testing.assert_array_equal(a["calls"], b["calls"])
        np.testing.assert_allclose(b["delta"], a["delta"], rtol=1e-10, atol=1e-9)
        ids = prob.gene_ids[chroms[ci]]
        E = np.array([hmm_oracle.emission(prob.expr[g], prob.avecs.get(g), iv) for g in ids])
        d_ref, st_ref, _ = hmm_oracle.viterbi(prob.tprob[chroms[ci]], E, iv)
        np.testing.assert_allclose(b["delta"], d_ref, rtol=1e-10, atol=1e-9)
        np.testing.assert_array_equal(b["states"], st_ref)


@pytest.mark.parametrize("delta_route", list(DELTA_ROUTES))
def test_blocked_scan_two_samples_all_delta_routes(delta_route, monkeypatch):
    """Two samples in the blocked scan (per-sample slots of the guesses, constants and flags), each against its own
    one-sample unblocked run."""
    from gbrs_amd import synth
    from gbrs_amd.hmm import DiplotypeHMM
    prob = synth.make_hmm_problem(H=8, genes_per_chrom=[900, 350, 130], style="do")
    chroms = prob.chroms
    rng = np.random.default_rng(5)
    ha = [np.array([g in prob.avecs for g in prob.gene_ids[c]], dtype=np.uint8) for c in chroms]
    av = [np.array([prob.avecs.get(g, np.zeros((8, 8))) for g in prob.gene_ids[c]]) for c in chroms]
    e0 = [np.array([prob.expr[g] for g in prob.gene_ids[c]]) for c in chroms]
    e1 = [rng.gamma(1.0, 5.0, size=e.shape) * (rng.random(e.shape) < 0.3) for e in e0]
    dims = (8, chroms, [len(prob.gene_ids[c]) for c in chroms], [prob.tprob[c] for c in chroms])
    want = ("states", "calls", "delta", "gamma")
    monkeypatch.setenv("GBRS_TUNING_HMM_BLOCKED", "0")
    single = []
    for ex in (e0, e1):
        hmm = DiplotypeHMM(*dims)
        hmm.set_expression(ex, av, ha, 1.5, 0.12)
        hmm.run()
        single.append([hmm.get(ci, want=want) for ci in range(3)])
        hmm.close()
    monkeypatch.setenv("GBRS_TUNING_HMM_BLOCKED", "2")
    monkeypatch.setenv("GBRS_TUNING_HMM_BLOCK_GENES", "30")
    for k, v in DELTA_ROUTES[delta_route].items():
        monkeypatch.setenv(k, v)
    hmm = DiplotypeHMM(*dims)
    hmm.set_expression([np.stack([a, b]) for a, b in zip(e0, e1)], av, ha, 1.5, 0.12)
    hmm.run()
    inf = hmm.info()
    if delta_route == "rank":
        assert inf.last_delta_fallbacks == 0 and inf.last_delta_blocks > 0
    elif delta_route == "fallback":
        assert inf.last_delta_fallbacks == 2 * 3
    for s in range(2):
        for ci in range(3):
            r = hmm.get(ci, sample=s, want=want)
            np.testing.assert_array_equal(r["states"], single[s][ci]["states"])
            np.testing.assert_array_equal(r["calls"], single[s][ci]["calls"])
            np.testing.assert_allclose(r["delta"], single[s][ci]["delta"], rtol=1e-10, atol=1e-9)
            np.testing.assert_allclose(r["gamma"], single[s][ci]["gamma"], rtol=1e-8, atol=1e-300)
    hmm.close()


def test_blocked_scan_equals_the_unblocked_chains_at_size(monkeypatch):
    """40k genes, 20 chromosomes, one sample: the blocked scan (64 blocks per long chromosome) against the unblocked
    chains of the same library - posteriors, log-domain arrays and delta at 1e-9, Viterbi path identical."""
    from gbrs_amd import synth
    from gbrs_amd.hmm import DiplotypeHMM
    prob = synth.make_hmm_problem(H=8)
    chroms = prob.chroms
    ex = [np.array([prob.expr[g] for g in prob.gene_ids[c]]) for c in chroms]
    ha = [np.array([g in prob.avecs for g in prob.gene_ids[c]], dtype=np.uint8) for c in chroms]
    av = [np.array([prob.avecs.get(g, np.zeros((8, 8))) for g in prob.gene_ids[c]]) for c in chroms]
    res = {}
    for mode in ("0", "2"):
        monkeypatch.setenv("GBRS_TUNING_HMM_BLOCKED", mode)
        hmm = DiplotypeHMM(8, chroms, [len(prob.gene_ids[c]) for c in chroms], [prob.tprob[c] for c in chroms])
        hmm.set_expression(ex, av, ha, 1.5, 0.12)
        hmm.run()
        res[mode] = [hmm.get(ci, want=("gamma", "states", "calls", "alpha", "beta", "delta", "scaler")) for ci in (0, 7, 19)]
        hmm.close()
    for a, b in zip(res["0"], res["2"]):
        np.testing.assert_array_equal(a["states"], b["states"])
        np.testing.assert_array_equal(a["calls"], b["calls"])
        for k in ("alpha", "beta", "delta", "scaler"):
            np.testing.assert_allclose(b[k], a[k], rtol=1e-9, atol=1e-9)
        np.testing.assert_allclose(b["gamma"], a["gamma"], rtol=1e-8, atol=1e-300)


@pytest.mark.parametrize("path", golden_files("hmm")[:2], ids=lambda p: p.split("/")[-1][:-4])
def test_hmm_with_host_emissions_and_sample_batch(path):
    """set_eprob path with the reference's own emissions, 3 identical samples in one launch."""
    g = load_golden(path)
    c = hmm_case_inputs(g)
    chroms = c["chroms"]
    hmm = build(c)
    hmm.set_eprob([np.repeat(g[f"eprob_{ch}"][None], 3, axis=0) for ch in chroms])
    hmm.run()
    for s in range(3):
        for ci, ch in enumerate(chroms):
            r = hmm.get(ci, sample=s)
            np.testing.assert_array_equal(r["states"], g[f"states_{ch}"])
            np.testing.assert_array_equal(r["calls"], g[f"calls_{ch}"])
            np.testing.assert_allclose(r["gamma"], g[f"gamma_{ch}"], rtol=1e-8, atol=1e-300)
    hmm.close()


def test_reconstruct_files(tmp_path, monkeypatch):
    """End-to-end file interface of `gbrs reconstruct` against the golden genotypes.tsv."""
    from gbrs_amd import hmm as H
    from gbrs_amd.synth import diplotype_names
    path = golden_files("hmm")[0]
    g = load_golden(path)
    c = hmm_case_inputs(g)
    chroms = c["chroms"]
    hn = [chr(65 + h) for h in range(c["H"])]
    (tmp_path / "ref.fa.fai").write_text("".join(f"{ch}\t1000\t0\t60\t61\n" for ch in chroms) + "MT\t16299\t0\t60\t61\n")
    monkeypatch.setenv("GBRS_DATA", str(tmp_path))
    with open(tmp_path / "genes.tpm", "w") as fh:
        fh.write("locus\t" + "\t".join(hn) + "\ttotal\n")
        for ch in chroms:
            for gid, v in zip(c["genes"][ch], c["expr"][ch]):
                fh.write(str(gid) + "\t" + "\t".join(repr(float(x)) for x in v) + "\t" + repr(float(v.sum())) + "\n")
    np.savez(tmp_path / "tprob.npz", **{ch: c["tprob"][ch] for ch in chroms})
    av = {}
    gp = {}
    for ch in chroms:
        for gid, has, a in zip(c["genes"][ch], c["has_avec"][ch], c["avecs"][ch]):
            if has:
                av[str(gid)] = a
        arr = np.zeros(len(c["genes"][ch]), dtype=[("f0", "U24"), ("f1", "i8")])
        arr["f0"] = c["genes"][ch]
        gp[ch] = arr
    np.savez(tmp_path / "avecs.npz", **av)
    np.savez(tmp_path / "gpos.npz", **gp)
    out = str(tmp_path / "out")
    H.reconstruct(str(tmp_path / "genes.tpm"), str(tmp_path / "tprob.npz"), str(tmp_path / "avecs.npz"),
                  str(tmp_path / "gpos.npz"), 1.5, 0.12, out)
    assert open(out + ".genotypes.tsv").read() == str(g["tsv_text"])
    gam = np.load(out + ".genoprobs.npz")
    st = np.load(out + ".genotypes.npz")
    names = diplotype_names(hn)
    for ch in chroms:
        np.testing.assert_allclose(gam[ch], g[f"gamma_{ch}"], rtol=1e-8, atol=1e-300)
        assert list(st[ch]) == [names[s] for s in g[f"states_{ch}"]]


def test_hmm_full_size_properties():
    """DO-sized synthetic genome (20 chromosomes, 40k genes): posterior columns sum to one,
    calls equal the oracle's on a sampled chromosome."""
    from gbrs_amd import synth
    from gbrs_amd.hmm import DiplotypeHMM
    from oracle import hmm_oracle
    prob = synth.make_hmm_problem(H=8)
    chroms = prob.chroms
    hmm = DiplotypeHMM(8, chroms, [len(prob.gene_ids[c]) for c in chroms], [prob.tprob[c] for c in chroms])
    ex, av, ha = [], [], []
    for c in chroms:
        ids = prob.gene_ids[c]
        ex.append(np.array([prob.expr[g] for g in ids]))
        ha.append(np.array([g in prob.avecs for g in ids], dtype=np.uint8))
        av.append(np.array([prob.avecs.get(g, np.zeros((8, 8))) for g in ids]))
    hmm.set_expression(ex, av, ha, 1.5, 0.12)
    hmm.run()
    for ci, c in enumerate(chroms):
        r = hmm.get(ci)
        assert np.isfinite(r["gamma"]).all()
        np.testing.assert_allclose(r["gamma"].sum(axis=0), 1.0, rtol=1e-12)
        assert (r["calls"] >= 0).all()
    c = chroms[-1]
    res = hmm_oracle.reconstruct_arrays(prob.hap_names, [c], prob.gene_ids, prob.tprob, prob.expr, prob.avecs)
    r = hmm.get(len(chroms) - 1)
    np.testing.assert_array_equal(r["calls"], res[c]["calls"])
    np.testing.assert_allclose(r["gamma"], res[c]["gamma"], rtol=1e-7, atol=1e-300)
    hmm.close()


@pytest.mark.parametrize("n_samples,mfma_ng,pipeline", [(1, 1, 0), (4, 1, 0), (5, 1, 0), (16, 1, 0), (25, 1, 0), (40, 1, 0),
                                                        (16, 2, 0), (25, 2, 0), (40, 2, 0), (70, 2, 0),
                                                        (16, 1, 16), (25, 1, 16), (40, 1, 16), (70, 1, 16)])
@pytest.mark.parametrize("minus_one", [False, True], ids=["tprob_n", "tprob_n_minus_1"])
def test_hmm_sample_batches_and_short_chromosomes(n_samples, mfma_ng, pipeline, minus_one, monkeypatch):
    """8 founders (the single-wave kernels; from 16 samples on this test sends the alpha and backward sweeps
    through the 16-samples-per-wavefront MFMA kernels that large batches use): distinct samples in one launch, including batch sizes
    that leave a partly filled wave, chromosomes of 1, 2, 3 genes and lengths around the prefetch
    ring and backtrace chunk sizes, both tprob-length conventions; every sample against the oracle."""
    from gbrs_amd import synth
    from gbrs_amd.hmm import DiplotypeHMM
    from oracle import hmm_oracle
    monkeypatch.setenv("GBRS_TUNING_HMM_MFMA", "16")
    monkeypatch.setenv("GBRS_TUNING_HMM_MFMA_NG", str(mfma_ng))   # round 4: one or two groups of 16 samples per wavefront of the sweeps
    # round 4: the batch pass as two pipelined chromosome groups (emission left to the run; default from 96 samples on)
    monkeypatch.setenv("GBRS_TUNING_HMM_PIPELINE", str(pipeline))
    monkeypatch.setenv("GBRS_TUNING_HMM_DLANES", "16")       # and the samples-on-lanes delta chain
    monkeypatch.setenv("GBRS_TUNING_HMM_BPLANES", "5")       # and the samples-on-lanes backpointers (partly filled wavefronts)
    # round 4 switches, off by default: delta as [gene][sample] between those two kernels, XCD-aware 1-D grids of the chain kernels
    monkeypatch.setenv("GBRS_TUNING_HMM_DELTA_ROWS", "1" if n_samples in (25, 70) else "0")
    monkeypatch.setenv("GBRS_TUNING_HMM_XCD", "3" if n_samples == 40 else "0")
    lens = [1, 2, 3, 4, 5, 7, 63, 64, 65, 129, 200]
    probs = [synth.make_hmm_problem(H=8, genes_per_chrom=lens, seed=1234 + s, tprob_len_minus_one=minus_one)
             for s in range(n_samples)]
    p0 = probs[0]
    chroms = p0.chroms
    hmm = DiplotypeHMM(8, chroms, [len(p0.gene_ids[c]) for c in chroms], [p0.tprob[c] for c in chroms])
    ex, av, ha = [], [], []
    for c in chroms:
        ids = p0.gene_ids[c]
        ex.append(np.array([[p.expr[g] for g in ids] for p in probs]))
        ha.append(np.array([g in p0.avecs for g in ids], dtype=np.uint8))
        av.append(np.array([p0.avecs.get(g, np.zeros((8, 8))) for g in ids]))
    hmm.set_expression(ex, av, ha, 1.5, 0.12)
    hmm.run()
    want = ("gamma", "states", "calls", "alpha", "beta", "delta", "scaler")
    for s, p in enumerate(probs):
        res = hmm_oracle.reconstruct_arrays(p0.hap_names, chroms, p0.gene_ids, p0.tprob, p.expr, p0.avecs)
        for ci, c in enumerate(chroms):
            r = hmm.get(ci, sample=s, want=want)
            np.testing.assert_array_equal(r["states"], res[c]["states"], err_msg=f"sample {s} chrom {c}")
            np.testing.assert_array_equal(r["calls"], res[c]["calls"], err_msg=f"sample {s} chrom {c}")
            for k in ("alpha", "beta", "delta", "scaler"):
                np.testing.assert_allclose(r[k], res[c][k], rtol=1e-9, atol=1e-9, err_msg=f"{k} sample {s} chrom {c}")
            np.testing.assert_allclose(r["gamma"], res[c]["gamma"], rtol=1e-8, atol=1e-300)
    hmm.close()


def test_hmm_backpointers_on_lanes_second_sample_group(monkeypatch):
    """260 samples: the samples-on-lanes backpointer kernel with a second, nearly empty group of 256 samples; the
    Viterbi paths and calls equal those of the one-target-per-lane kernel bit for bit."""
    from gbrs_amd import synth
    from gbrs_amd.hmm import DiplotypeHMM
    n_samples = 260
    p0 = synth.make_hmm_problem(H=8, genes_per_chrom=[1, 2, 31, 70], seed=77)
    chroms = p0.chroms
    rng = np.random.default_rng(5)
    ex, av, ha = [], [], []
    for c in chroms:
        ids = p0.gene_ids[c]
        e = np.array([p0.expr[g] for g in ids])
        ex.append(np.stack([e] + [rng.gamma(1.0, 5.0, size=e.shape) * (rng.random(e.shape) < 0.6) for _ in range(n_samples - 1)]))
        ha.append(np.array([g in p0.avecs for g in ids], dtype=np.uint8))
        av.append(np.array([p0.avecs.get(g, np.zeros((8, 8))) for g in ids]))
    out = {}
    for mode in ("0", "1"):
        monkeypatch.setenv("GBRS_TUNING_HMM_BPLANES", mode)
        hmm = DiplotypeHMM(8, chroms, [len(p0.gene_ids[c]) for c in chroms], [p0.tprob[c] for c in chroms])
        hmm.set_expression(ex, av, ha, 1.5, 0.12)
        hmm.run()
        out[mode] = [[hmm.get(ci, sample=s, want=("states", "calls")) for ci in range(len(chroms))] for s in range(n_samples)]
        hmm.close()
    for s in range(n_samples):
        for ci in range(len(chroms)):
            np.testing.assert_array_equal(out["0"][s][ci]["states"], out["1"][s][ci]["states"], err_msg=f"sample {s} chrom {ci}")
            np.testing.assert_array_equal(out["0"][s][ci]["calls"], out["1"][s][ci]["calls"])


def test_hmm_large_batch_default_dispatch():
    """70 samples with the library's own thresholds (no tuning switch): the launch takes the MFMA sweeps and the
    samples-on-lanes delta chain by itself (both from 64 samples on), with a partly filled last group of 16; every
    sample against the oracle."""
    from gbrs_amd import synth
    from gbrs_amd.hmm import DiplotypeHMM
    from oracle import hmm_oracle
    n_samples = 70
    lens = [1, 3, 17, 64, 90]
    probs = [synth.make_hmm_problem(H=8, genes_per_chrom=lens, seed=4321 + s) for s in range(n_samples)]
    p0 = probs[0]
    chroms = p0.chroms
    hmm = DiplotypeHMM(8, chroms, [len(p0.gene_ids[c]) for c in chroms], [p0.tprob[c] for c in chroms])
    ex, av, ha = [], [], []
    for c in chroms:
        ids = p0.gene_ids[c]
        ex.append(np.array([[p.expr[g] for g in ids] for p in probs]))
        ha.append(np.array([g in p0.avecs for g in ids], dtype=np.uint8))
        av.append(np.array([p0.avecs.get(g, np.zeros((8, 8))) for g in ids]))
    hmm.set_expression(ex, av, ha, 1.5, 0.12)
    hmm.run()
    want = ("gamma", "states", "calls", "alpha", "beta", "delta", "scaler")
    for s, p in enumerate(probs):
        res = hmm_oracle.reconstruct_arrays(p0.hap_names, chroms, p0.gene_ids, p0.tprob, p.expr, p0.avecs)
        for ci, c in enumerate(chroms):
            r = hmm.get(ci, sample=s, want=want)
            np.testing.assert_array_equal(r["states"], res[c]["states"], err_msg=f"sample {s} chrom {c}")
            np.testing.assert_array_equal(r["calls"], res[c]["calls"], err_msg=f"sample {s} chrom {c}")
            for k in ("alpha", "beta", "delta", "scaler"):
                np.testing.assert_allclose(r[k], res[c][k], rtol=1e-9, atol=1e-9, err_msg=f"{k} sample {s} chrom {c}")
            np.testing.assert_allclose(r["gamma"], res[c]["gamma"], rtol=1e-8, atol=1e-300)
    hmm.close()


@pytest.mark.parametrize("n_samples,mfma_ng", [(3, 1), (21, 1), (21, 2), (37, 2)])
def test_hmm_do_tables_sample_batches(n_samples, mfma_ng, monkeypatch):
    """DO-like transition tables (entries down to exp(-69), structural zeros, a near-deterministic chain) and
    sparsely expressed samples, single-sample kernels (3) and the 16-samples-per-wave MFMA sweeps (21):
    every sample against the oracle."""
    from gbrs_amd import synth
    from gbrs_amd.hmm import DiplotypeHMM
    from oracle import hmm_oracle
    monkeypatch.setenv("GBRS_TUNING_HMM_MFMA", "16")
    monkeypatch.setenv("GBRS_TUNING_HMM_MFMA_NG", str(mfma_ng))
    monkeypatch.setenv("GBRS_TUNING_HMM_DLANES", "16")       # and the samples-on-lanes delta chain
    lens = [1, 2, 17, 64, 150]
    probs = [synth.make_hmm_problem(H=8, genes_per_chrom=lens, seed=4321 + s, style="do",
                                    expressed_fraction=0.9 if s % 3 else 0.3) for s in range(n_samples)]
    p0 = probs[0]
    chroms = p0.chroms
    hmm = DiplotypeHMM(8, chroms, [len(p0.gene_ids[c]) for c in chroms], [p0.tprob[c] for c in chroms])
    ex, av, ha = [], [], []
    for c in chroms:
        ids = p0.gene_ids[c]
        ex.append(np.array([[p.expr[g] for g in ids] for p in probs]))
        ha.append(np.array([g in p0.avecs for g in ids], dtype=np.uint8))
        av.append(np.array([p0.avecs.get(g, np.zeros((8, 8))) for g in ids]))
    hmm.set_expression(ex, av, ha, 1.5, 0.12)
    hmm.run()
    want = ("gamma", "states", "calls", "alpha", "beta", "scaler")
    for s, p in enumerate(probs):
        res = hmm_oracle.reconstruct_arrays(p0.hap_names, chroms, p0.gene_ids, p0.tprob, p.expr, p0.avecs)
        for ci, c in enumerate(chroms):
            r = hmm.get(ci, sample=s, want=want)
            np.testing.assert_array_equal(r["calls"], res[c]["calls"], err_msg=f"sample {s} chrom {c}")
            np.testing.assert_array_equal(r["states"], res[c]["states"], err_msg=f"sample {s} chrom {c}")
            for k in ("alpha", "beta", "scaler"):
                np.testing.assert_allclose(r[k], res[c][k], rtol=1e-9, atol=1e-9, err_msg=f"{k} sample {s} chrom {c}")
            np.testing.assert_allclose(r["gamma"], res[c]["gamma"], rtol=1e-8, atol=1e-300)
    hmm.close()


@pytest.mark.parametrize("n_samples", [1, 3])
@pytest.mark.parametrize("minus_one", [False, True], ids=["tprob_n", "tprob_n_minus_1"])
def test_hmm_sixteen_founders(n_samples, minus_one):
    """16 founders (136 states, BASELINE config 5's reconstruct): the 4-lanes-per-state chains."""
    from gbrs_amd import synth
    from gbrs_amd.hmm import DiplotypeHMM
    from oracle import hmm_oracle
    lens = [1, 2, 3, 5, 33, 70]
    probs = [synth.make_hmm_problem(H=16, genes_per_chrom=lens, seed=4321 + s, tprob_len_minus_one=minus_one)
             for s in range(n_samples)]
    p0 = probs[0]
    chroms = p0.chroms
    hmm = DiplotypeHMM(16, chroms, [len(p0.gene_ids[c]) for c in chroms], [p0.tprob[c] for c in chroms])
    ex, av, ha = [], [], []
    for c in chroms:
        ids = p0.gene_ids[c]
        ex.append(np.array([[p.expr[g] for g in ids] for p in probs]))
        ha.append(np.array([g in p0.avecs for g in ids], dtype=np.uint8))
        av.append(np.array([p0.avecs.get(g, np.zeros((16, 16))) for g in ids]))
    hmm.set_expression(ex, av, ha, 1.5, 0.12)
    hmm.run()
    want = ("gamma", "states", "calls", "alpha", "beta", "delta", "scaler")
    for s, p in enumerate(probs):
        res = hmm_oracle.reconstruct_arrays(p0.hap_names, chroms, p0.gene_ids, p0.tprob, p.expr, p0.avecs)
        for ci, c in enumerate(chroms):
            r = hmm.get(ci, sample=s, want=want)
            np.testing.assert_array_equal(r["states"], res[c]["states"], err_msg=f"sample {s} chrom {c}")
            np.testing.assert_array_equal(r["calls"], res[c]["calls"], err_msg=f"sample {s} chrom {c}")
            for k in ("alpha", "beta", "delta", "scaler"):
                np.testing.assert_allclose(r[k], res[c][k], rtol=1e-9, atol=1e-9, err_msg=f"{k} sample {s} chrom {c}")
            np.testing.assert_allclose(r["gamma"], res[c]["gamma"], rtol=1e-8, atol=1e-300)
    hmm.close()


@pytest.mark.parametrize("H", [2, 3, 4, 5, 7, 9])
def test_hmm_other_founder_counts(H):
    """Founder counts other than 8 and 16: 3, 4 and 7 run on the single-wave chain kernels (even
    state counts <= 64), 2, 5 and 9 on the generic multi-wave kernels; 5 samples, oracle per sample."""
    from gbrs_amd import synth
    from gbrs_amd.hmm import DiplotypeHMM
    from oracle import hmm_oracle
    lens = [1, 2, 3, 6, 65, 130]
    probs = [synth.make_hmm_problem(H=H, genes_per_chrom=lens, seed=777 + s, tprob_len_minus_one=bool(H % 2))
             for s in range(5)]
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
            np.testing.assert_array_equal(r["states"], res[c]["states"], err_msg=f"sample {s} chrom {c}")
            np.testing.assert_array_equal(r["calls"], res[c]["calls"], err_msg=f"sample {s} chrom {c}")
            for k in ("alpha", "beta", "delta", "scaler"):
                np.testing.assert_allclose(r[k], res[c][k], rtol=1e-9, atol=1e-9, err_msg=f"{k} sample {s} chrom {c}")
            np.testing.assert_allclose(r["gamma"], res[c]["gamma"], rtol=1e-8, atol=1e-300)
    hmm.close()


def test_hmm_log_intermediates_follow_the_last_run():
    """alpha / beta / scaler are made on the first get() that asks for them; a later run with other
    emissions (and another sample count) must not hand back the earlier run's arrays."""
    paths = golden_files("hmm")
    g = load_golden([p for p in paths if "h8_full" in p][0])
    c = hmm_case_inputs(g)
    chroms = c["chroms"]
    hmm = build(c)
    rng = np.random.default_rng(5)
    other = [np.log(rng.dirichlet(np.ones(hmm.S), size=len(c["genes"][ch]))) for ch in chroms]
    hmm.set_eprob([np.stack([e, e]) for e in other])
    hmm.run()
    first = hmm.get(0, sample=1, want=("alpha", "beta", "scaler"))
    hmm.set_eprob([g[f"eprob_{ch}"] for ch in chroms])
    hmm.run()
    for ci, ch in enumerate(chroms):
        r = hmm.get(ci, want=("gamma",))
        np.testing.assert_allclose(r["gamma"], g[f"gamma_{ch}"], rtol=1e-8, atol=1e-300)
        r = hmm.get(ci, want=("beta",))
        np.testing.assert_allclose(r["beta"], g[f"beta_{ch}"], rtol=1e-9, atol=1e-9)
        r = hmm.get(ci, want=("alpha", "scaler"))
        np.testing.assert_allclose(r["alpha"], g[f"alpha_{ch}"], rtol=1e-9, atol=1e-9)
        np.testing.assert_allclose(r["scaler"], g[f"scaler_{ch}"], rtol=1e-9, atol=1e-9)
    assert not np.allclose(first["alpha"], hmm.get(0, want=("alpha",))["alpha"])
    hmm.close()


def test_batched_emission_kernel_matches_the_single_sample_kernel_bit_for_bit():
    """From 4 samples on the emission model runs in emission_batch_kernel (gene-only part hoisted out of
    the sample loop); it must produce exactly the single-sample kernel's emissions, including genes
    below the expression threshold, genes without specificity data and a partly filled sample block."""
    from gbrs_amd import synth
    from gbrs_amd.hmm import DiplotypeHMM
    p0 = synth.make_hmm_problem(H=8, genes_per_chrom=[61, 130], seed=77)
    chroms = p0.chroms
    rng = np.random.default_rng(3)
    ns = 19
    ex, av, ha = [], [], []
    for c in chroms:
        ids = p0.gene_ids[c]
        e0 = np.array([p0.expr[g] for g in ids])
        e = np.stack([e0] + [rng.gamma(1.0, 5.0, size=e0.shape) * (rng.random(e0.shape) < 0.5) for _ in range(ns - 1)])
        e[3] *= 1e-9                                   # a sample below the threshold everywhere
        ex.append(e)
        ha.append(np.array([g in p0.avecs for g in ids], dtype=np.uint8))
        av.append(np.array([p0.avecs.get(g, np.zeros((8, 8))) for g in ids]))
    args = ([len(p0.gene_ids[c]) for c in chroms], [p0.tprob[c] for c in chroms])
    batch = DiplotypeHMM(8, chroms, *args)
    batch.set_expression(ex, av, ha, 1.5, 0.12)
    one = DiplotypeHMM(8, chroms, *args)
    for s in (0, 3, 15, 16, 18):
        one.set_expression([e[s] for e in ex], av, ha, 1.5, 0.12)
        for ci in range(len(chroms)):
            a = batch.get(ci, sample=s, want=("eprob",))["eprob"]
            b = one.get(ci, want=("eprob",))["eprob"]
            np.testing.assert_array_equal(a, b, err_msg=f"sample {s} chromosome {ci}")
    batch.close()
    one.close()


def test_specificity_tables_stay_resident():
    """avecs / has_avec are uploaded once per handle; later samples pass only their expression rows."""
    g = load_golden([p for p in golden_files("hmm") if p.endswith("hmm_h8_do_full.npz")][0])
    c = hmm_case_inputs(g)
    chroms = c["chroms"]
    hmm = build(c)
    ex = [c["expr"][ch] for ch in chroms]
    with pytest.raises(Exception, match="specificity"):
        hmm.set_expression(ex)
    rng = np.random.default_rng(3)
    other = [rng.gamma(1.0, 5.0, size=e.shape) * (rng.random(e.shape) < 0.5) for e in ex]
    hmm.set_expression(other, [c["avecs"][ch] for ch in chroms], [c["has_avec"][ch] for ch in chroms], 1.5, 0.12)
    hmm.run()
    hmm.set_expression(ex, expr_threshold=float(g["expr_threshold"]), sigma=float(g["sigma"]))   # tables reused
    hmm.run()
    for ci, ch in enumerate(chroms):
        r = hmm.get(ci, want=("gamma", "calls", "eprob"))
        np.testing.assert_array_equal(r["calls"], g[f"calls_{ch}"])
        np.testing.assert_allclose(r["eprob"], g[f"eprob_{ch}"], rtol=1e-10, atol=1e-10)
        np.testing.assert_allclose(r["gamma"], g[f"gamma_{ch}"], rtol=1e-8, atol=1e-300)
    hmm.close()


def test_hmm_sixteen_founders_at_size():
    """BASELINE configs[4]'s reconstruct shape: 136 diplotype states on a genome-sized run (20 chromosomes,
    6,000 genes, 2 samples).  Size-independent properties on every chromosome - posterior columns sum to one,
    every gene gets a call in range, the Viterbi path has n + 1 entries - and the oracle on the two shortest
    chromosomes (calls bit-exact, posteriors 1e-8)."""
    from gbrs_amd import synth
    from gbrs_amd.hmm import DiplotypeHMM
    from oracle import hmm_oracle
    lens = [int(round(n * 0.15)) for n in synth.MOUSE_GENES]
    probs = [synth.make_hmm_problem(H=16, genes_per_chrom=lens, seed=9100 + s) for s in range(2)]
    p0 = probs[0]
    chroms = p0.chroms
    hmm = DiplotypeHMM(16, chroms, [len(p0.gene_ids[c]) for c in chroms], [p0.tprob[c] for c in chroms])
    ex, av, ha = [], [], []
    for c in chroms:
        ids = p0.gene_ids[c]
        ex.append(np.array([[p.expr[g] for g in ids] for p in probs]))
        ha.append(np.array([g in p0.avecs for g in ids], dtype=np.uint8))
        av.append(np.array([p0.avecs.get(g, np.zeros((16, 16))) for g in ids]))
    hmm.set_expression(ex, av, ha, 1.5, 0.12)
    hmm.run()
    short = sorted(range(len(chroms)), key=lambda k: lens[k])[:2]
    for s, p in enumerate(probs):
        for ci, c in enumerate(chroms):
            r = hmm.get(ci, sample=s)
            n = lens[ci]
            assert r["gamma"].shape == (136, n) and np.isfinite(r["gamma"]).all()
            np.testing.assert_allclose(r["gamma"].sum(axis=0), 1.0, rtol=1e-12)
            assert len(r["states"]) == n + 1 and r["calls"].min() >= 0 and r["calls"].max() < 136
        sub = [chroms[k] for k in short]
        res = hmm_oracle.reconstruct_arrays(p0.hap_names, sub, p0.gene_ids, p0.tprob, p.expr, p0.avecs)
        for k in short:
            r = hmm.get(k, sample=s)
            np.testing.assert_array_equal(r["calls"], res[chroms[k]]["calls"])
            np.testing.assert_allclose(r["gamma"], res[chroms[k]]["gamma"], rtol=1e-8, atol=1e-300)
    hmm.close()
