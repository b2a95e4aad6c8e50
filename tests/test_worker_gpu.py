"""gbrs_amd.worker (one resident process for many samples) leaves the files the three commands leave (needs an MI355X)."""
import filecmp
import json
import os
import sys

import numpy as np
import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def _write_sample(workdir, seed, rows=30_000, loci=600):
    sys.path.insert(0, os.path.join(ROOT, "scripts"))
    import e2e_bench
    from gbrs_amd import synth
    from gbrs_amd.alignment import AlignmentPropertyMatrix
    inc = synth.make_em_problem(R=rows, H=8, L=loci, seed=seed)
    starts = [m[0] for m in inc.groups]
    eff = inc.effective_length(100)
    lname, hname, gname, grp, lens = e2e_bench.write_support_files(str(workdir), loci, 8, starts, eff[0])
    apm = AlignmentPropertyMatrix(shape=(loci, 8, rows), indptr=inc.indptr, indices=inc.indices, haplotype_names=hname,
                                  locus_names=lname)
    aln = os.path.join(str(workdir), f"sample{seed}.npz")
    apm.save_npz(aln)
    return aln, grp, lens


def test_worker_equals_the_three_commands(tmp_path, monkeypatch):
    """Two samples through SampleWorker.process (alignment file read once per sample, reconstruct tables once per process)
    and through `gbrs quantify` / `gbrs reconstruct` / `gbrs quantify -G` one by one: the same eleven files per sample,
    the genotype calls byte for byte, the .npz outputs array for array, the reports number for number (1e-9)."""
    sys.path.insert(0, os.path.join(ROOT, "scripts"))
    import e2e_bench
    from gbrs_amd import cli
    from gbrs_amd.worker import run_jobs
    monkeypatch.setenv("GBRS_DATA", str(tmp_path))
    samples = [_write_sample(tmp_path, seed) for seed in (5, 6)]
    # the commands, sample by sample (the first quantify also provides the gene list the reconstruct inputs are made for)
    rec = None
    for k, (aln, grp, lens) in enumerate(samples):
        base = str(tmp_path / f"cmd{k}")
        assert cli.main(["quantify", "-i", aln, "-g", grp, "-L", lens, "-o", base]) == 0
        if rec is None:
            rec, _ = e2e_bench.write_reconstruct_inputs(str(tmp_path), base + ".multiway.genes.tpm")
        assert cli.main(["reconstruct", "-e", base + ".multiway.genes.tpm", "-t", rec["tprob"], "-x", rec["avecs"],
                         "-g", rec["gpos"], "-o", base]) == 0
        assert cli.main(["quantify", "-i", aln, "-g", grp, "-L", lens, "-G", base + ".genotypes.tsv", "-o", base]) == 0
    jobs = [dict(alignment_file=aln, group_file=grp, length_file=lens, outbase=str(tmp_path / f"wrk{k}"),
                 tprob_file=rec["tprob"], avec_file=rec["avecs"], gpos_file=rec["gpos"])
            for k, (aln, grp, lens) in enumerate(samples)]
    lines = []
    done, seconds = run_jobs(jobs, device=0, emit=lambda text, flush=True: lines.append(text))
    assert len(done) == 2 and not any("error" in d for d in done), done
    assert all("quantify_diploid" in json.loads(x) for x in lines)
    text_files = [f"{kind}.{level}.{what}" for kind in ("multiway", "diploid") for level in ("isoforms", "genes")
                  for what in ("tpm", "expected_read_counts")] + ["genotypes.tsv"]
    def table(path):
        rows = [line.rstrip("\n").split("\t") for line in open(path)]
        return rows[0], [r[0] for r in rows[1:]], np.array([[float(x) for x in r[1:9 + 1]] for r in rows[1:]]), \
            [r[10:] for r in rows[1:]]
    for k in range(2):
        for suffix in text_files:
            a, b = tmp_path / f"cmd{k}.{suffix}", tmp_path / f"wrk{k}.{suffix}"
            if suffix == "genotypes.tsv":
                assert filecmp.cmp(a, b, shallow=False), suffix           # the calls are argmax decisions: identical
                continue
            # the reports: same header, names and notes; numbers to 1e-9 (the default layout adds its partial sums through
            # LDS float atomics, so two runs differ in the last digits: GBRS_EM_DETERMINISTIC makes them bit-equal)
            ha, na, va, xa = table(a)
            hb, nb, vb, xb = table(b)
            assert ha == hb and na == nb and xa == xb, suffix
            np.testing.assert_allclose(va, vb, rtol=1e-9, atol=1e-300)
        for suffix in ("genoprobs.npz", "genotypes.npz"):
            za, zb = np.load(tmp_path / f"cmd{k}.{suffix}"), np.load(tmp_path / f"wrk{k}.{suffix}")
            assert sorted(za.files) == sorted(zb.files)
            for c in za.files:
                if suffix == "genotypes.npz":
                    np.testing.assert_array_equal(za[c], zb[c])
                else:                      # posteriors of TPM tables that differ in their last digits (see above)
                    np.testing.assert_allclose(za[c], zb[c], rtol=1e-8, atol=1e-300)
    # the samples differ (the worker did not hand the first sample's state to the second)
    assert not filecmp.cmp(tmp_path / "wrk0.multiway.genes.tpm", tmp_path / "wrk1.multiway.genes.tpm", shallow=False)


def test_worker_processes_per_device(tmp_path, monkeypatch):
    """`gbrs worker --devices d0,d1,...`: one resident worker process per listed device (BASELINE configs[3]: samples one
    per GPU, replicas only); on a one-GPU box the list names device 0 twice.  Three samples dealt round-robin to two
    processes: every sample reported once, its files there, calls equal to a single worker's."""
    sys.path.insert(0, os.path.join(ROOT, "scripts"))
    import e2e_bench
    from gbrs_amd import cli
    from gbrs_amd.worker import run_jobs, run_jobs_on_devices
    monkeypatch.setenv("GBRS_DATA", str(tmp_path))
    monkeypatch.setenv("PYTHONPATH", ROOT)
    samples = [_write_sample(tmp_path, seed, rows=20_000, loci=400) for seed in (7, 8, 9)]
    aln, grp, lens = samples[0]
    assert cli.main(["quantify", "-i", aln, "-g", grp, "-L", lens, "-o", str(tmp_path / "seed")]) == 0
    rec, _ = e2e_bench.write_reconstruct_inputs(str(tmp_path), str(tmp_path / "seed.multiway.genes.tpm"))

    def jobs(tag):
        return [dict(alignment_file=a, group_file=g, length_file=l, outbase=str(tmp_path / f"{tag}{k}"),
                     tprob_file=rec["tprob"], avec_file=rec["avecs"], gpos_file=rec["gpos"]) for k, (a, g, l) in enumerate(samples)]
    lines = []
    done, seconds = run_jobs_on_devices(jobs("multi"), [0, 0], emit=lambda text, flush=True: lines.append(text))
    assert sorted(d["outbase"] for d in done) == sorted(j["outbase"] for j in jobs("multi"))
    assert sorted(d["worker"] for d in done) == [0, 0, 1] and not any("error" in d for d in done)
    single, _ = run_jobs(jobs("single"), device=0, emit=None)
    assert not any("error" in d for d in single)
    for k in range(3):
        assert filecmp.cmp(tmp_path / f"multi{k}.genotypes.tsv", tmp_path / f"single{k}.genotypes.tsv", shallow=False)
        assert os.path.getsize(tmp_path / f"multi{k}.diploid.genes.tpm") > 0
