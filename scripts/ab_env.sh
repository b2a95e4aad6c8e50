#!/bin/bash
# A/B of environment switches on the EM bench (GPU box).  Usage: scripts/ab_env.sh OUT "<env assignments>;<env assignments>;..." [bench args]
OUT=${1:-gpurun_out/ab}; mkdir -p $OUT
IFS=';' read -ra VARS <<< "$2"
shift 2
i=0
for X in "${VARS[@]}"; do
  for rep in a b; do
  env $X timeout -k 10 300 python bench.py --no-e2e --no-cpu-baseline --no-hmm --no-merged-line --no-multi-isoform-line "$@" > $OUT/v${i}_$rep.log 2>&1
  python - "$X" "$OUT/v${i}_$rep.log" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[2]).read().strip().split("\n")[-1])
    print("%-44s estep %.4f ms  step %.4f ms  tiles %d bytes %d ok %s" % (sys.argv[1], d["roofline"]["kernel_ms"], d["ms_per_step"], d["config"]["tiles"], d["roofline"]["kernel_bytes"], d["state_check"]["ok"]))
except Exception as e:
    print(sys.argv[1], "FAILED", e)
PY
  done
  i=$((i+1))
done
