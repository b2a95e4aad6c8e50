#!/usr/bin/env python3
"""End-to-end wall clock of the three-command GBRS pipeline, files in -> reports out: `gbrs quantify` ->
`gbrs reconstruct` -> `gbrs quantify -G genotypes.tsv` (gbrs/emase_utils.py:180-332,
gbrs/gbrs_utils.py:382-609).  BASELINE.json's ">= 50x" is stated on the first two.

  1. builds the BASELINE configs[1] sample with the bench generator and writes it as an EMASE file
     (`.h5` through libhdf5 and/or the `.npz` mirror), with its group and length files;
  2. runs `python -m gbrs_amd quantify ...`, `... reconstruct ...` on its genes.tpm and `... quantify -G` on the
     genotypes.tsv that leaves behind, each as a fresh child process (interpreter start, imports and HIP
     initialisation are inside the measured wall clock) and collects the per-stage times the drivers record
     (GBRS_STAGE_TIMES);
  3. runs oracle/e2e_oracle.py (the numpy restatement of the reference workflow, one core per command) on the
     same files - on the FULL sample when the host has the memory for it (>= 128 GB; three one-core processes side
     by side), else on a row subsample whose row-dependent stages are scaled linearly, and says which.

Prints one JSON object.  Needs an MI355X.  Usage:
    python scripts/e2e_bench.py [--rows N] [--format h5|npz|both] [--cpu-rows M] [--workdir DIR] [--keep]
"""
from __future__ import annotations

import argparse
import json
import os
import shutil
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def write_support_files(workdir, L, H, gene_starts, eff_len_row):
    import numpy as np
    lname = [f"T{l:07d}" for l in range(L)]
    hname = [chr(65 + h) for h in range(H)]
    starts = list(gene_starts) + [L]
    gname = [f"G{g:07d}" for g in range(len(gene_starts))]
    grp = os.path.join(workdir, "ref.gene2transcripts.tsv")
    with open(grp, "w") as fh:
        for g in range(len(gene_starts)):
            fh.write(gname[g] + "\t" + "\t".join(lname[starts[g]:starts[g + 1]]) + "\n")
    lens = os.path.join(workdir, "gbrs.hybridized.targets.info")
    raw = (np.asarray(eff_len_row) + 99).astype(np.int64)       # max(raw - 100 + 1, 1) == eff
    with open(lens, "w") as fh:
        for l in range(L):
            for h in hname:
                fh.write(f"{lname[l]}_{h}\t{raw[l]}\n")
    return lname, hname, gname, grp, lens


def build_sample(workdir, rows, haps, loci, fmt, cpu_rows):
    """Child-process body (`--make-sample`): the full-size sample from the bench generator as alignment
    file(s) + support files, and the row subsample for the one-core baseline.  Runs in a process of its own
    so that the measuring parent never opens the GPU: a second process holding a device context slows every
    large hipMalloc / hipFree of the measured one (layout build 87 -> 410 ms on this pool)."""
    import numpy as np
    import torch
    from gbrs_amd import synth, synth_torch
    from gbrs_amd.alignment import AlignmentPropertyMatrix
    t0 = time.perf_counter()
    prob = synth_torch.make_em_problem_device(rows, haps, loci, synth.SEED_BASE_EM + 1, "cuda:0")
    ip = [t.cpu().numpy().view(np.uint32) for t in prob["indptr"]]
    ix = [t.cpu().numpy().view(np.uint32) for t in prob["indices"]]
    eff = prob["eff_len"].cpu().numpy()
    gene_starts = prob["gene_starts"]
    N = prob["N"]
    del prob
    torch.cuda.empty_cache()
    lname, hname, gname, grp, lens = write_support_files(workdir, loci, haps, gene_starts, eff[0])
    apm = AlignmentPropertyMatrix(shape=(loci, haps, rows), indptr=ip, indices=ix, haplotype_names=hname,
                                  locus_names=lname)
    out = dict(group_file=grp, length_file=lens, N=N, files={}, write_s={})
    if fmt in ("h5", "both"):
        t1 = time.perf_counter()
        p = os.path.join(workdir, "sample.h5")
        apm.save(p)
        out["files"]["h5"] = p
        out["write_s"]["h5"] = time.perf_counter() - t1
    if fmt in ("npz", "both"):
        t1 = time.perf_counter()
        p = os.path.join(workdir, "sample.npz")
        apm.save_npz(p)
        out["files"]["npz"] = p
        out["write_s"]["npz"] = time.perf_counter() - t1
    out["bytes"] = {k: os.path.getsize(v) for k, v in out["files"].items()}
    if cpu_rows >= rows and "npz" in out["files"]:
        out["cpu_sub"], out["cpu_sub_entries"] = out["files"]["npz"], N
    elif cpu_rows >= rows:
        # the whole sample as the `.npz` mirror the one-core baseline reads (members deflated on all cores here; the
        # baseline inflates them on its one)
        from gbrs_amd.npzfast import savez_compressed
        p = os.path.join(workdir, "full.npz")
        members = dict(shape=np.asarray((loci, haps, rows), dtype=np.int64), hname=np.array(hname), lname=np.array(lname))
        for h in range(haps):
            members[f"indptr{h}"] = ip[h]
            members[f"indices{h}"] = ix[h]
        savez_compressed(p, members)
        out["cpu_sub"], out["cpu_sub_entries"] = p, N
    elif cpu_rows:
        out["cpu_sub"], out["cpu_sub_entries"] = write_cpu_subsample(workdir, (ip, ix, eff, lname, hname),
                                                                     min(cpu_rows, rows), loci, haps)
    out["build_s"] = time.perf_counter() - t0
    return out


def build_sample_in_child(workdir, rows, haps, loci, fmt, cpu_rows):
    r = subprocess.run([sys.executable, os.path.abspath(__file__), "--make-sample", "--workdir", workdir,
                        "--rows", str(rows), "--haps", str(haps), "--loci", str(loci), "--format", fmt,
                        "--cpu-rows", str(cpu_rows)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True,
                       env=dict(os.environ, PYTHONPATH=ROOT))
    if r.returncode != 0:
        raise RuntimeError(f"sample generation failed: {r.stderr[-2000:]}")
    return json.loads(r.stdout.strip().split("\n")[-1])


def write_cpu_subsample(workdir, arrays, rows_sub, loci, haps):
    """The first rows_sub reads of the sample as an .npz alignment file for the one-core baseline."""
    import numpy as np
    from gbrs_amd.alignment import AlignmentPropertyMatrix
    ip, ix, eff, lname, hname = arrays
    sip, six = [], []
    for h in range(haps):
        keep = ix[h] < rows_sub
        col = np.repeat(np.arange(loci, dtype=np.int64), np.diff(ip[h].astype(np.int64)))[keep]
        six.append(ix[h][keep])
        sip.append(np.searchsorted(col, np.arange(loci + 1)).astype(np.uint32))
    apm = AlignmentPropertyMatrix(shape=(loci, haps, rows_sub), indptr=sip, indices=six, haplotype_names=hname,
                                  locus_names=lname)
    p = os.path.join(workdir, "sub.npz")
    apm.save_npz(p)
    return p, int(sum(len(x) for x in six))


def write_reconstruct_inputs(workdir, genes_tpm):
    """ref.fa.fai, gene order, transition tables and specificity blocks for the genes of a genes.tpm file
    (SURVEY 8d's HMM recipe: 20 chromosomes in mouse proportions, `eye + 0.01 U` tables of DO length)."""
    import numpy as np
    from gbrs_amd import synth
    with open(genes_tpm) as fh:
        H = len(fh.readline().rstrip().split("\t")) - 2
        ids = [line.split("\t", 1)[0] for line in fh]
    S = H * (H + 1) // 2
    frac = np.cumsum(synth.MOUSE_GENES) / float(sum(synth.MOUSE_GENES))
    cuts = [0] + [int(round(f * len(ids))) for f in frac]
    rng = np.random.default_rng(synth.SEED_HMM)
    tprob, gpos, avecs = {}, {}, {}
    with open(os.path.join(workdir, "ref.fa.fai"), "w") as fh:
        for c in synth.MOUSE_CHROMS:
            fh.write(f"{c}\t100000000\t0\t60\t61\n")
    for k, c in enumerate(synth.MOUSE_CHROMS):
        g = ids[cuts[k]:cuts[k + 1]]
        n = len(g)
        T = np.eye(S)[None, :, :] + 0.01 * rng.random((n, S, S))
        T /= T.sum(axis=1, keepdims=True)
        tprob[c] = np.log(T)
        arr = np.zeros(n, dtype=[("f0", "U24"), ("f1", "i8")])
        arr["f0"] = g
        arr["f1"] = np.arange(n) * 1000
        gpos[c] = arr
        has = rng.random(n) < 0.7
        for i in np.flatnonzero(has):
            a = np.eye(H) + 0.05 * rng.random((H, H))
            avecs[g[i]] = a / a.sum(axis=1, keepdims=True)
    paths = dict(tprob=os.path.join(workdir, "tranprob.npz"), avecs=os.path.join(workdir, "avecs.npz"),
                 gpos=os.path.join(workdir, "ref.gene_pos.ordered.npz"), fai=os.path.join(workdir, "ref.fa.fai"))
    # deflated members, as the reference writes these files (np.savez_compressed: gbrs_utils.py:293, :379); the
    # container is written by gbrs_amd.npzfast.savez_compressed only because it deflates the members on all cores
    from gbrs_amd.npzfast import savez_compressed
    savez_compressed(paths["tprob"], tprob)
    savez_compressed(paths["avecs"], avecs)
    savez_compressed(paths["gpos"], gpos)
    return paths, len(ids)


def run_cli(argv, workdir, tag):
    """One `gbrs` subcommand as a fresh process; returns (wall seconds, stage dict)."""
    stages = os.path.join(workdir, f"stages_{tag}.json")
    env = dict(os.environ, GBRS_DATA=workdir, GBRS_STAGE_TIMES=stages, PYTHONPATH=ROOT)
    t0 = time.time()
    env["GBRS_T0"] = repr(t0)
    r = subprocess.run([sys.executable, "-m", "gbrs_amd"] + argv, env=env, cwd=workdir, stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, text=True)
    wall = time.time() - t0
    if r.returncode != 0 or not os.path.exists(stages):
        raise RuntimeError(f"gbrs {argv[0]} failed (rc {r.returncode}): {r.stderr[-2000:]}")
    with open(stages) as fh:
        st = json.load(fh)
    if st.get("error"):
        raise RuntimeError(f"gbrs {argv[0]} logged an error: {st['error']}")
    return wall, st


def run_workers(workdir, sample, rec, n_workers, samples_each, tag):
    """n_workers resident processes (`python -m gbrs_amd worker`) side by side on the one GPU, each taking samples_each
    samples through quantify -> reconstruct -> quantify -G.  The samples are the same files (page cache warm, as for the
    single commands) under outbases of their own; native decode / format threads are capped at cores / workers.
    Returns wall seconds (launch of the first process to exit of the last) and the per-sample stage times."""
    first = next(iter(sample["files"]))
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    procs = []
    t0 = time.time()
    for w in range(n_workers):
        jobs = [dict(alignment_file=sample["files"][first], group_file=sample["group_file"], length_file=sample["length_file"],
                     outbase=os.path.join(workdir, f"w{tag}_{w}_{k}"), tprob_file=rec["tprob"], avec_file=rec["avecs"],
                     gpos_file=rec["gpos"]) for k in range(samples_each)]
        jf = os.path.join(workdir, f"jobs_{tag}_{w}.json")
        with open(jf, "w") as fh:
            json.dump(jobs, fh)
        env = dict(os.environ, GBRS_DATA=workdir, PYTHONPATH=ROOT, GBRS_T0=repr(t0),
                   GBRS_IO_THREADS=str(max(1, cores // n_workers)))
        procs.append(subprocess.Popen([sys.executable, "-m", "gbrs_amd", "worker", "--jobs", jf], env=env, cwd=workdir,
                                      stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = [p.communicate() for p in procs]
    wall = time.time() - t0
    samples = []
    for p, (out, err) in zip(procs, outs):
        if p.returncode != 0:
            raise RuntimeError(f"worker failed (rc {p.returncode}): {err[-2000:]}")
        for line in out.strip().split("\n"):
            if not line.startswith("{"):            # (EMfactory.run's iteration table goes to stdout, as in the reference)
                continue
            rec_line = json.loads(line)
            if "summary" in rec_line:
                if rec_line["summary"]["failed"]:
                    raise RuntimeError(f"worker reported failed samples: {out[-2000:]}")
            else:
                samples.append(rec_line)
    return wall, samples


def mean_stages(samples):
    """Mean seconds per stage over the samples of a worker run, first sample of a process (tables loaded, library warmed
    up) apart from the later ones."""
    def avg(vals):
        return sum(vals) / len(vals) if vals else None
    out = {}
    for part in ("load_alignment", "wall"):
        out[part] = avg([s[part] for s in samples])
    for cmd in ("quantify", "reconstruct", "quantify_diploid"):
        keys = set().union(*[set(s.get(cmd, {})) for s in samples])
        out[cmd] = {k: avg([s[cmd][k] for s in samples if k in s.get(cmd, {}) and isinstance(s[cmd][k], (int, float))])
                    for k in sorted(keys)}
    return out


def start_oracle(argv):
    env = dict(os.environ, OMP_NUM_THREADS="1", OPENBLAS_NUM_THREADS="1", MKL_NUM_THREADS="1", PYTHONPATH=ROOT)
    return time.time(), subprocess.Popen([sys.executable, os.path.join(ROOT, "oracle", "e2e_oracle.py")] + argv, env=env,
                                         stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)


def finish_oracle(started, what):
    t0, proc = started
    out, err = proc.communicate()
    wall = time.time() - t0
    if proc.returncode != 0:
        raise RuntimeError(f"oracle {what} failed: {err[-2000:]}")
    return wall, json.loads(out.strip().split("\n")[-1])


def run_oracle(argv):
    return finish_oracle(start_oracle(argv), argv[0])


def host_memory_gb():
    try:
        return os.sysconf("SC_PAGE_SIZE") * os.sysconf("SC_PHYS_PAGES") / 2.0**30
    except (ValueError, OSError):
        return 0.0


FULL_CPU_MIN_GB = 128.0        # the one-core oracle needs ~40 GB per quantify process at 40M reads


def best_of(runs):
    best = min(runs, key=lambda r: r["wall_s"])
    return dict(wall_s=best["wall_s"], stages=best["stages"], all_wall_s=[r["wall_s"] for r in runs])


def scaled_quantify(wall, st, scale, iters_full):
    """A subsample's one-core quantify extrapolated to the full sample: stages that walk the reads scale linearly in
    them; parsing the L-sized tables, the genotype table, the reports and process start do not."""
    per_iter = st["em_run"] / max(st["em_iterations"], 1)
    return st["load"] * scale + st["em_setup"] * scale + st.get("mask", 0.0) + per_iter * scale * iters_full \
        + st["reports"] + (wall - st["total_in_process"])


def measure(rows=40_000_000, haps=8, loci=120_000, fmt="h5", cpu_rows=2_000_000, workdir=None, keep=False,
            repeats=2, with_cpu=True, cpu_full="auto", workers=(1, 2, 4), samples_each=2):
    """cpu_full: "auto" = the one-core baseline runs on the whole sample when the host has FULL_CPU_MIN_GB of memory,
    "yes" / "no" force it."""
    own = workdir is None
    workdir = tempfile.mkdtemp(prefix="gbrs_e2e_") if own else workdir
    os.makedirs(workdir, exist_ok=True)
    mem_gb = host_memory_gb()
    full = with_cpu and (cpu_full == "yes" or (cpu_full == "auto" and mem_gb >= FULL_CPU_MIN_GB))
    if full:
        cpu_rows = rows
    res = dict(workload=f"configs[1]/[2]: R={rows} reads x H={haps} x L={loci} isoforms from file: quantify (Model 4, "
                        "tol 1e-4, 4 reports), reconstruct on its genes.tpm (20 chromosomes), quantify -G on the "
                        "genotypes.tsv of that (diploid pass)")
    try:
        sample = build_sample_in_child(workdir, rows, haps, loci, fmt, cpu_rows if with_cpu else 0)
        res["sample"] = dict(entries=sample["N"], file_bytes=sample["bytes"], write_s=sample["write_s"])
        res["quantify"] = {}
        for kind, path in sample["files"].items():
            runs = []
            for _ in range(repeats):
                wall, st = run_cli(["quantify", "-i", path, "-g", sample["group_file"], "-L", sample["length_file"],
                                    "-o", os.path.join(workdir, f"out_{kind}")], workdir, f"q_{kind}")
                runs.append(dict(wall_s=wall, stages=st))
            res["quantify"][kind] = best_of(runs)
        first = next(iter(sample["files"]))
        genes_tpm = os.path.join(workdir, f"out_{first}.multiway.genes.tpm")
        rec, n_genes = write_reconstruct_inputs(workdir, genes_tpm)
        runs = []
        for _ in range(repeats):
            wall, st = run_cli(["reconstruct", "-e", genes_tpm, "-t", rec["tprob"], "-x", rec["avecs"], "-g", rec["gpos"],
                                "-o", os.path.join(workdir, "rec")], workdir, "r")
            runs.append(dict(wall_s=wall, stages=st))
        res["reconstruct"] = dict(best_of(runs), genes=n_genes)
        # third command: the diploid pass on the genotype calls reconstruct just wrote (gbrs/emase_utils.py:240-273)
        genotypes = os.path.join(workdir, "rec.genotypes.tsv")
        runs = []
        for _ in range(repeats):
            wall, st = run_cli(["quantify", "-i", sample["files"][first], "-g", sample["group_file"], "-L",
                                sample["length_file"], "-G", genotypes, "-o", os.path.join(workdir, f"out_{first}")],
                               workdir, "qg")
            runs.append(dict(wall_s=wall, stages=st))
        res["quantify_diploid"] = best_of(runs)
        qbest = min(v["wall_s"] for v in res["quantify"].values())
        res["total_wall_s"] = qbest + res["reconstruct"]["wall_s"]
        res["pipeline_wall_s"] = res["total_wall_s"] + res["quantify_diploid"]["wall_s"]
        if workers:
            # BASELINE configs[3] is 8 samples on 8 GPUs: whether its >= 6x can be host-bound shows on one GPU - resident
            # workers (gbrs_amd.worker) side by side on the card, sharing the host's cores, page cache and PCIe
            conc = {}
            for nw in workers:
                wall, samples = run_workers(workdir, sample, rec, nw, samples_each, f"c{nw}")
                conc[str(nw)] = dict(workers=nw, samples=len(samples), wall_s=wall, samples_per_s=len(samples) / wall,
                                     seconds_per_sample_in_worker=sum(x["wall"] for x in samples) / len(samples),
                                     first_sample_of_a_process_s=sum(x["wall"] for x in samples[::samples_each]) / nw,
                                     stage_means_s=mean_stages(samples))
            base = conc[str(workers[0])]["samples_per_s"] / workers[0]
            res["concurrent"] = dict(
                note="resident workers on ONE GPU: every process takes its samples through quantify -> reconstruct -> "
                     "quantify -G (the three commands' own functions; the alignment file is read once per sample, the "
                     "reconstruct tables once per process); wall = first launch to last exit, interpreter and HIP start-up "
                     "included; a sample as three fresh commands costs pipeline_wall_s",
                samples_per_worker=samples_each, by_workers=conc,
                scaling_vs_one_worker={k: v["samples_per_s"] / (base * workers[0]) for k, v in conc.items()},
                three_commands_samples_per_s=1.0 / res["pipeline_wall_s"])
        if with_cpu:
            sub, n_sub = sample["cpu_sub"], sample["cpu_sub_entries"]
            # three one-core processes side by side (the box's other cores are idle); the two downstream commands read
            # the device path's genes.tpm / genotypes.tsv, which tests hold equal to the oracle's own
            pq = start_oracle(["quantify", sub, sample["group_file"], sample["length_file"], os.path.join(workdir, "cpu")])
            pg = start_oracle(["quantify", sub, sample["group_file"], sample["length_file"], os.path.join(workdir, "cpu"),
                               genotypes])
            pr = start_oracle(["reconstruct", genes_tpm, rec["tprob"], rec["avecs"], rec["gpos"], rec["fai"],
                               os.path.join(workdir, "cpu_rec")])
            wr, tr = finish_oracle(pr, "reconstruct")
            wg, tg = finish_oracle(pg, "quantify -G")
            wq, tq = finish_oracle(pq, "quantify")
            n_cpu = min(cpu_rows, rows)
            scale = rows / float(n_cpu)
            iters_full = res["quantify"][first]["stages"].get("em_iterations", tq["em_iterations"])
            iters_full_g = res["quantify_diploid"]["stages"].get("em_iterations", tg["em_iterations"])
            if scale == 1.0:
                est_q, est_g = wq, wg
                how = (f"oracle/e2e_oracle.py on the whole sample ({n_sub} entries, .npz mirror of the same reads), one "
                       f"core per command, the three commands side by side on a host with {os.cpu_count()} cores and "
                       f"{mem_gb:.0f} GB; nothing scaled or extrapolated")
            else:
                est_q = scaled_quantify(wq, tq, scale, iters_full)
                est_g = scaled_quantify(wg, tg, scale, iters_full_g)
                how = (f"oracle/e2e_oracle.py on the first {n_cpu} reads ({n_sub} entries) of the same sample as an .npz "
                       f"file (host memory {mem_gb:.0f} GB < {FULL_CPU_MIN_GB:.0f} GB needed for the whole sample); load, "
                       f"EM set-up and per-iteration time scaled x{scale:g} linearly in reads at the full sample's "
                       f"iteration counts ({iters_full}, diploid {iters_full_g}), reports and process start unscaled; "
                       "reconstruct run whole")
            res["cpu_baseline"] = dict(
                kind="port", cores=1, host_cores=os.cpu_count(), host_memory_gb=mem_gb, full_size=scale == 1.0, sample=how,
                quantify_measured=dict(wall_s=wq, stages=tq), reconstruct_measured=dict(wall_s=wr, stages=tr),
                quantify_diploid_measured=dict(wall_s=wg, stages=tg),
                quantify_scaled_s=est_q, reconstruct_s=wr, quantify_diploid_scaled_s=est_g, total_s=est_q + wr,
                pipeline_s=est_q + wr + est_g)
            res["speedup_vs_cpu"] = dict(quantify=est_q / qbest, reconstruct=wr / res["reconstruct"]["wall_s"],
                                         quantify_diploid=est_g / res["quantify_diploid"]["wall_s"],
                                         total=(est_q + wr) / res["total_wall_s"],
                                         pipeline=(est_q + wr + est_g) / res["pipeline_wall_s"])
    finally:
        if own and not keep:
            shutil.rmtree(workdir, ignore_errors=True)
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=40_000_000)
    ap.add_argument("--haps", type=int, default=8)
    ap.add_argument("--loci", type=int, default=120_000)
    ap.add_argument("--format", default="h5", choices=["h5", "npz", "both"])
    ap.add_argument("--cpu-rows", type=int, default=2_000_000)
    ap.add_argument("--workdir", default=None)
    ap.add_argument("--keep", action="store_true")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--repeats", type=int, default=2)
    ap.add_argument("--cpu-full", default="auto", choices=["auto", "yes", "no"],
                    help="one-core baseline on the whole sample (auto: when the host has >= 128 GB)")
    ap.add_argument("--workers", default="1,2,4", help="concurrent resident workers to measure on the one GPU (empty: skip); "
                                                       "a GPU box allows at most 6 processes on its card")
    ap.add_argument("--samples-each", type=int, default=2)
    ap.add_argument("--make-sample", action="store_true", help="(internal) child process that writes the sample files")
    a = ap.parse_args()
    if a.make_sample:
        print(json.dumps(build_sample(a.workdir, a.rows, a.haps, a.loci, a.format, a.cpu_rows)), flush=True)
        return
    subprocess.run([sys.executable, "-c", "import __graft_entry__ as g; g.build()"], cwd=ROOT, check=True,
                   stdout=sys.stderr)             # built in a child: the parent never loads the HIP runtime
    workers = tuple(int(x) for x in a.workers.split(",") if x.strip())
    out = measure(a.rows, a.haps, a.loci, a.format, a.cpu_rows, a.workdir, a.keep, a.repeats, not a.no_cpu, a.cpu_full,
                  workers, a.samples_each)
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
