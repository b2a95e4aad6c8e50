#!/bin/bash
# HMM-only run: tests of the blocked scan, time of a pass, rocprofv3 kernel stats (GPU box).  Usage: scripts/hmm_kt.sh OUT [samples]
OUT=$(realpath -m ${1:-gpurun_out/hmm_kt}); NS=${2:-1}
R=$PWD
timeout -k 10 600 python -m pytest tests/test_hmm_gpu.py -m gpu -x -q -k "blocked" 2>&1 | tail -2
python scripts/hmm_only.py $NS 20
cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 $R/scripts/hmm_only.py $NS 20 > $OUT.log 2>&1
cd $R
python3 - "$OUT" <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/*/*kernel_stats.csv")[0]
for r in csv.DictReader(open(f)):
    if "gbrs" in r["Name"] and int(r["Calls"]) >= 20:
        print("%-75s calls %4s avg %9.1f us" % (r["Name"][:75], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
