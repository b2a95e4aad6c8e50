#!/usr/bin/env python3
"""Where the end-to-end commands spend their wall time: builds the C2 sample files once (scripts/e2e_bench.py), then
runs `gbrs quantify` and `gbrs reconstruct` under cProfile and with -X importtime.  GPU box.  Usage: e2e_profile.py OUTDIR"""
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "scripts"))
import e2e_bench  # noqa: E402

out = os.path.abspath(sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/e2e_prof")
os.makedirs(out, exist_ok=True)
work = tempfile.mkdtemp(prefix="gbrs_e2e_prof_")
res = e2e_bench.measure(workdir=work, keep=True, with_cpu=False, repeats=1)
print("measure:", res["quantify"], res["reconstruct"], flush=True)
env = dict(os.environ, PYTHONPATH=ROOT, GBRS_DATA=work)
prof_env = dict(env, GBRS_ORDERLY_EXIT="1")        # cProfile writes its file at interpreter exit
sample_h5 = "sample.h5"
cmds = {
    "quantify": ["quantify", "-i", os.path.join(work, sample_h5), "-g", os.path.join(work, "ref.gene2transcripts.tsv"), "-L",
                 os.path.join(work, "gbrs.hybridized.targets.info"), "-o", os.path.join(work, "prof_q")],
    "reconstruct": ["reconstruct", "-e", os.path.join(work, "out_h5.multiway.genes.tpm"), "-t", os.path.join(work, "tranprob.npz"),
                    "-x", os.path.join(work, "avecs.npz"), "-g", os.path.join(work, "ref.gene_pos.ordered.npz"), "-o", os.path.join(work, "prof_r")],
}
import time
for name, argv in cmds.items():
    for mode in ("fast", "orderly", "fast", "orderly"):
        e2 = dict(env)
        if mode == "orderly":
            e2["GBRS_ORDERLY_EXIT"] = "1"
        t0 = time.perf_counter()
        subprocess.run([sys.executable, "-m", "gbrs_amd"] + argv, env=e2, cwd=work, capture_output=True, text=True)
        print(f"{name} exit={mode}: wall {time.perf_counter() - t0:.3f} s", flush=True)
for name, argv in cmds.items():
    r = subprocess.run([sys.executable, "-X", "importtime", "-m", "gbrs_amd"] + argv, env=env, cwd=work, capture_output=True, text=True)
    lines = [l for l in r.stderr.split("\n") if l.startswith("import time:")]
    rows = []
    for l in lines[1:]:
        try:
            self_us, cum_us, mod = l[len("import time:"):].split("|")
            rows.append((int(cum_us), mod.strip()))
        except ValueError:
            pass
    rows.sort(reverse=True)
    with open(os.path.join(out, f"{name}_importtime.txt"), "w") as fh:
        fh.write("\n".join(f"{c / 1000:9.1f} ms  {m}" for c, m in rows[:40]) + "\n")
    prof = os.path.join(out, f"{name}.prof")
    subprocess.run([sys.executable, "-m", "cProfile", "-o", prof, "-m", "gbrs_amd"] + argv, env=prof_env, cwd=work, capture_output=True, text=True)
    import pstats
    with open(os.path.join(out, f"{name}_cprofile.txt"), "w") as fh:
        pstats.Stats(prof, stream=fh).sort_stats("cumulative").print_stats(45)
    os.remove(prof)
r = subprocess.run([sys.executable, os.path.join(ROOT, 'scripts', 'cold_create_probe.py'), os.path.join(work, 'sample.h5')], env=env, capture_output=True, text=True)
open(os.path.join(out, 'create_stages.txt'), 'w').write(r.stderr)
import shutil
shutil.rmtree(work, ignore_errors=True)
