#!/bin/bash
# As hmm_kt.sh without the tests, under an environment setting.  Usage: scripts/hmm_kt_env.sh OUT samples VAR=VALUE...
OUT=$(realpath -m $1); NS=$2; shift 2
for kv in "$@"; do export "$kv"; done
R=$PWD
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
