#!/usr/bin/env python3
"""Timing of SURVEY 8(f)'s "next" rows at BASELINE configs[1] size (needs an MI355X): `gbrs compress` on the 40M-read
sample (gbrs/emase_utils.py:22-107), `--report-alignment-counts` (emase/AlignmentPropertyMatrix.py:389-459) and
`gbrs interpolate` / `gbrs export` on the 40k-gene genoprobs.npz (gbrs/gbrs_utils.py:612-697, :863-938).

For every row: the wall clock of the library call on arrays already in host memory, the wall clock of the command as a
fresh process, files in -> files out, and the one-core numpy / Python oracle beside it (on a bounded row subsample, scaled
linearly and said so, where the oracle is a per-read Python loop).  Prints one JSON object.  The device time per kernel
is what `rocprofv3 --kernel-trace --stats -- python3 scripts/next_rows_bench.py --calls-only` lists."""
from __future__ import annotations

import argparse
import json
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "scripts"))


def cli(argv, workdir):
    """One `gbrs` subcommand as a fresh process: wall seconds (a failure the command only logs, as the reference's CLI
    does, is an error here)."""
    import e2e_bench
    return e2e_bench.run_cli(argv, workdir, argv[0])[0]


def subsample(ip, ix, rows_sub, L):
    import numpy as np
    sip, six = [], []
    for h in range(len(ip)):
        keep = ix[h] < rows_sub
        col = np.repeat(np.arange(L, dtype=np.int64), np.diff(ip[h].astype(np.int64)))[keep]
        six.append(ix[h][keep])
        sip.append(np.searchsorted(col, np.arange(L + 1)).astype(np.uint32))
    return sip, six


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=40_000_000)
    ap.add_argument("--haps", type=int, default=8)
    ap.add_argument("--loci", type=int, default=120_000)
    ap.add_argument("--calls-only", action="store_true", help="only the library calls on in-memory arrays (for rocprofv3)")
    ap.add_argument("--compress-cpu-rows", type=int, default=200_000)
    ap.add_argument("--counts-cpu-rows", type=int, default=2_000_000)
    a = ap.parse_args()
    import numpy as np
    import torch
    from gbrs_amd import synth, synth_torch
    from gbrs_amd.alignment import AlignmentPropertyMatrix
    from gbrs_amd.compress import compress_matrix
    from gbrs_amd.counts import alignment_counts
    R, H, L = a.rows, a.haps, a.loci
    prob = synth_torch.make_em_problem_device(R, H, L, synth.SEED_BASE_EM + 1, "cuda:0")
    ip = [t.cpu().numpy().view(np.uint32) for t in prob["indptr"]]
    ix = [t.cpu().numpy().view(np.uint32) for t in prob["indices"]]
    eff = prob["eff_len"].cpu().numpy()
    gene_starts, N = prob["gene_starts"], prob["N"]
    del prob
    torch.cuda.empty_cache()
    lname = [f"T{l:07d}" for l in range(L)]
    hname = [chr(65 + h) for h in range(H)]
    apm = AlignmentPropertyMatrix(shape=(L, H, R), indptr=ip, indices=ix, haplotype_names=hname, locus_names=lname)
    starts = list(gene_starts) + [L]
    apm.groups = [list(range(starts[g], starts[g + 1])) for g in range(len(gene_starts))]
    apm.gname = np.array([f"G{g:07d}" for g in range(len(gene_starts))])
    apm.num_groups = len(apm.groups)
    out = dict(workload=f"configs[1] sample: R={R} reads x H={H} x L={L} isoforms, N={N} entries")

    # ---- compress ------------------------------------------------------------------------------------------------
    compress_matrix(AlignmentPropertyMatrix(shape=(L, H, 1000), indptr=subsample(ip, ix, 1000, L)[0],
                                            indices=subsample(ip, ix, 1000, L)[1]))        # library warm-up
    t0 = time.perf_counter()
    ec = compress_matrix(apm)
    t_call = time.perf_counter() - t0
    out["compress"] = dict(library_call_s=t_call, equivalence_classes=int(ec.num_reads), entries_out=int(ec.nnz),
                           reads_per_s=R / t_call,
                           note="gbrs_compress_create + gbrs_compress_get on host arrays: 1.45 GB of row ids over PCIe, the "
                                "device sort / unique / relabel, the EC arrays back")
    # ---- alignment counts ----------------------------------------------------------------------------------------
    t0 = time.perf_counter()
    aln_i = alignment_counts(apm, grp_wise=False)
    t_iso = time.perf_counter() - t0
    t0 = time.perf_counter()
    aln_g = alignment_counts(apm, grp_wise=True)
    t_gene = time.perf_counter() - t0
    out["alignment_counts"] = dict(isoform_level_s=t_iso, gene_level_s=t_gene,
                                   total_alignments=float(aln_i[0].sum()), locus_unique_reads=float(aln_i[2].sum()),
                                   gene_unique_reads=float(aln_g[2].sum()))
    assert aln_i[0].sum() == N
    if a.calls_only:
        print(json.dumps(out), flush=True)
        return

    # ---- one-core oracles on subsamples --------------------------------------------------------------------------
    from oracle import compress_oracle, counts_oracle
    rs = min(a.compress_cpu_rows, R)
    sip, six = subsample(ip, ix, rs, L)
    t0 = time.perf_counter()
    n_ec, *_ = compress_oracle.compress(rs, L, H, sip, six)
    t_or = time.perf_counter() - t0
    out["compress"]["cpu_baseline"] = dict(
        kind="port", cores=1, seconds=t_or, rows=rs, scaled_to_full_s=t_or * R / rs,
        sample=f"oracle/compress_oracle.py (the reference's per-read Python loop, gbrs/emase_utils.py:60-103) on the first "
               f"{rs} reads: {t_or:.1f} s, {n_ec} classes; x{R / rs:g} linearly in reads for the whole sample")
    out["compress"]["speedup_vs_cpu_library_call"] = t_or * R / rs / t_call
    rs = min(a.counts_cpu_rows, R)
    sip, six = subsample(ip, ix, rs, L)
    t0 = time.perf_counter()
    counts_oracle.alignment_counts(rs, L, H, sip, six)
    t_or = time.perf_counter() - t0
    out["alignment_counts"]["cpu_baseline"] = dict(
        kind="port", cores=1, seconds=t_or, rows=rs, scaled_to_full_s=t_or * R / rs,
        sample=f"oracle/counts_oracle.py (numpy restatement of count_alignments / count_unique_reads) isoform level on the "
               f"first {rs} reads: {t_or:.1f} s; x{R / rs:g} linearly in reads")
    out["alignment_counts"]["speedup_vs_cpu_isoform_level"] = t_or * R / rs / t_iso

    # ---- the commands, files in -> files out ---------------------------------------------------------------------
    import e2e_bench
    work = tempfile.mkdtemp(prefix="gbrs_next_")
    try:
        _, _, _, grp, lens = e2e_bench.write_support_files(work, L, H, gene_starts, eff[0])
        sample = os.path.join(work, "sample.h5")
        apm.save(sample)
        del ec
        out["compress"]["command_wall_s"] = cli(["compress", "-i", sample, "-o", os.path.join(work, "ec.h5")], work)
        out["compress"]["input_bytes"], out["compress"]["output_bytes"] = os.path.getsize(sample), os.path.getsize(os.path.join(work, "ec.h5"))
        t_q = cli(["quantify", "-i", sample, "-g", grp, "-L", lens, "-o", os.path.join(work, "q")], work)
        t_qa = cli(["quantify", "-i", sample, "-g", grp, "-L", lens, "-o", os.path.join(work, "qa"), "-a"], work)
        out["alignment_counts"]["quantify_wall_s"] = t_q
        out["alignment_counts"]["quantify_with_alignment_counts_wall_s"] = t_qa
        # reconstruct -> genoprobs.npz (40k genes x 36 states), then interpolate onto a 64k-marker grid and export
        rec, n_genes = e2e_bench.write_reconstruct_inputs(work, os.path.join(work, "q.multiway.genes.tpm"))
        cli(["reconstruct", "-e", os.path.join(work, "q.multiway.genes.tpm"), "-t", rec["tprob"], "-x", rec["avecs"], "-g",
             rec["gpos"], "-o", os.path.join(work, "rec")], work)
        gpos = np.load(rec["gpos"])
        grid_file = os.path.join(work, "ref.genome_grid.64k.txt")
        n_grid = 0
        grid = {}
        with open(grid_file, "w") as fh:
            fh.write("marker\tchr\tpos\tcM\n")
            for c in synth.MOUSE_CHROMS:
                span = float(gpos[c]["f1"][-1]) if len(gpos[c]) else 1.0
                m = max(int(round(64000 * len(gpos[c]) / max(n_genes, 1))), 2)
                pos = np.linspace(1.0, span, m)
                grid[c] = pos
                for k, x in enumerate(pos):
                    fh.write(f"m{c}_{k}\t{c}\t{int(x)}\t{float(x)!r}\n")
                n_grid += m
        genoprobs = os.path.join(work, "rec.genoprobs.npz")
        t_int = cli(["interpolate", "-i", genoprobs, "-g", grid_file, "-p", rec["gpos"], "-o", os.path.join(work, "interp.npz")], work)
        t_exp = cli(["export", "-i", os.path.join(work, "interp.npz"), "-s", ",".join(hname), "-g", grid_file, "-o",
                     os.path.join(work, "dosage.tsv")], work)
        from oracle import postproc_oracle
        gp = np.load(genoprobs)
        t0 = time.perf_counter()
        rows = []
        for c in synth.MOUSE_CHROMS:
            rows.append(postproc_oracle.interpolate(gpos[c]["f1"].astype(float), gp[c], grid[c]).T)
        t_oi = time.perf_counter() - t0
        t0 = time.perf_counter()
        postproc_oracle.dosage(np.vstack(rows), H)
        t_oe = time.perf_counter() - t0
        out["interpolate_export"] = dict(
            genes=n_genes, grid_points=n_grid, interpolate_command_wall_s=t_int, export_command_wall_s=t_exp,
            cpu_baseline=dict(kind="port", cores=1, interpolate_arithmetic_s=t_oi, export_arithmetic_s=t_oe,
                              sample="oracle/postproc_oracle.py on the whole genoprobs.npz (arithmetic only, arrays in memory)"),
            note="both commands are file handling around a few milliseconds of arithmetic: np.load / savez_compressed / "
                 "savetxt of 64k x 36 numbers and two interpreter + HIP start-ups")
    finally:
        import shutil
        shutil.rmtree(work, ignore_errors=True)
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
