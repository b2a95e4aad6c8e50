#!/usr/bin/env python3
"""`gbrs quantify` / `gbrs reconstruct` wall time against the number of I/O threads of the native file helpers
(GBRS_IO_THREADS; default = hardware_concurrency).  Builds the C2 sample once.  GPU box."""
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "scripts"))
import e2e_bench  # noqa: E402

work = tempfile.mkdtemp(prefix="gbrs_io_probe_")
res = e2e_bench.measure(workdir=work, keep=True, with_cpu=False, repeats=1)
print("baseline:", res["quantify"]["h5"]["wall_s"], res["reconstruct"]["wall_s"], "cpus visible", os.cpu_count(),
      "affinity", len(os.sched_getaffinity(0)), flush=True)
q = ["quantify", "-i", os.path.join(work, "sample.h5"), "-g", os.path.join(work, "ref.gene2transcripts.tsv"), "-L",
     os.path.join(work, "gbrs.hybridized.targets.info"), "-o", os.path.join(work, "probe_q")]
r = ["reconstruct", "-e", os.path.join(work, "out_h5.multiway.genes.tpm"), "-t", os.path.join(work, "tranprob.npz"),
     "-x", os.path.join(work, "avecs.npz"), "-g", os.path.join(work, "ref.gene_pos.ordered.npz"), "-o", os.path.join(work, "probe_r")]
for nt in ("", "8", "16", "24", "32", "64", "128"):
    if nt:
        os.environ["GBRS_IO_THREADS"] = nt
    else:
        os.environ.pop("GBRS_IO_THREADS", None)
    out = []
    for argv, tag in ((q, "q"), (r, "r")):
        runs = []
        for _ in range(3):
            wall, st = e2e_bench.run_cli(argv, work, tag)
            runs.append((wall, st.get("load")))
        out.append(min(runs))
    print(f"threads {nt or 'default':8s} quantify {out[0][0]:.3f} s (load {out[0][1]:.3f})   reconstruct {out[1][0]:.3f} s (load {out[1][1]:.3f})", flush=True)
