#!/bin/bash
# HBM traffic of the reconstruct pass's kernels from the FETCH_SIZE / WRITE_SIZE counters (each its own rocprofv3 --pmc run; the
# counters serialise the kernels; units as in scripts/summarize_prof.py: KB, FETCH_SIZE doubled per the gfx950 note of
# MI355X_MICROARCH.md), next to each kernel's duration from a kernel trace.  Usage: scripts/hmm_traffic_pmc.sh OUT [samples]
OUT=$(realpath -m ${1:-gpurun_out/hmm_traffic}); NS=${2:-256}; R=$PWD; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- python3 $R/scripts/hmm_only.py $NS 2 > $OUT/fetch.log 2>&1 &&
timeout -k 10 400 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -- python3 $R/scripts/hmm_only.py $NS 2 > $OUT/write.log 2>&1 &&
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -- python3 $R/scripts/hmm_only.py $NS 2 > $OUT/kt.log 2>&1
cd $R
python3 - "$OUT" "$NS" <<'PY'
import csv, glob, sys, collections
out, ns = sys.argv[1], int(sys.argv[2])
def per_kernel(d, counter):
    acc = collections.defaultdict(list)
    for f in glob.glob(f"{out}/{d}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter:
                k = r["Kernel_Name"]; k = k[k.find("gbrs::") + 6:] if "gbrs::" in k else k
                acc[k.split("(")[0]].append(float(r["Counter_Value"]))
    return acc
fe, wr = per_kernel("fetch", "FETCH_SIZE"), per_kernel("write", "WRITE_SIZE")
dur = {}
for f in glob.glob(f"{out}/kt/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Name"]; k = k[k.find("gbrs::") + 6:] if "gbrs::" in k else k
        dur[k.split("(")[0]] = (float(r["AverageNs"]) / 1e3, int(r["Calls"]))
print(f"# {ns} samples x 40,000 genes x 36 states; per launch (last launches of each kernel: the timed passes); FETCH_SIZE x 2 KB -> bytes")
tot = 0.0
for k in sorted(fe, key=lambda k: -max(fe[k])):
    f = fe[k][-1] * 2 * 1024; w = (wr.get(k) or [0])[-1] * 1024
    if f + w < (50e6 if ns >= 16 else 2e6): continue
    d = dur.get(k, (0, 0))[0]
    tot += f + w
    print(f"{k[:44]:44s} read {f / 1e9:7.2f} GB  written {w / 1e9:6.2f} GB  side by side {d:9.1f} us  -> {(f + w) / max(d, 1e-9) / 1e6:6.2f} TB/s if it ran alone at that time")
print(f"sum over the pass's kernels: {tot / 1e9:.1f} GB")
PY
