#!/bin/bash
# rocprofv3 kernel-trace stats of the EM bench under environment switches (GPU box).
# Usage: scripts/kt_env.sh OUT "<env assignments>;<env assignments>;..." [bench args]
OUT=$(realpath -m ${1:-gpurun_out/kt}); mkdir -p $OUT
IFS=';' read -ra VARS <<< "$2"
shift 2
R=$PWD
export TMPDIR=/tmp
cd /tmp
i=0
for X in "${VARS[@]}"; do
  for kv in $X; do export "$kv"; done
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/v$i -- python3 $R/bench.py --no-e2e --no-cpu-baseline --no-hmm --no-merged-line --no-multi-isoform-line --no-check "$@" > $OUT/v$i.log 2>&1
  echo "== $X"
  f=$(find $OUT/v$i -name "*kernel_stats.csv" | head -1)
  python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:8]:
    print("%-70s calls %6s avg %10.1f ns total %5.1f%%" % (r["Name"][:70], r["Calls"], float(r["AverageNs"]), float(r["Percentage"])))
PY
  i=$((i+1))
done
cd $R
