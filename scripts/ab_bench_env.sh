#!/bin/bash
# The EM bench under environment settings, two runs each, alternating (GPU box).  Usage: scripts/ab_bench_env.sh "<ENV=..>;<ENV=..>" [bench args]
IFS=';' read -ra VARS <<< "$1"
shift
for rep in a b; do
for X in "${VARS[@]}"; do
  env $X timeout -k 10 300 python bench.py --no-e2e --no-cpu-baseline --no-hmm --no-merged-line --no-multi-isoform-line --no-check --steps ${STEPS:-300} --warmup ${WARM:-50} "$@" > /tmp/ab_env.log 2>&1
  python - "$X" /tmp/ab_env.log <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[2]).read().strip().split("\n")[-1])
    print("%-50s estep %.4f ms  step %.4f ms  tiles %d ok %s" % (sys.argv[1], d["roofline"]["kernel_ms"], d["ms_per_step"], d["config"]["tiles"], d["state_check"]["ok"]))
except Exception as e:
    print(sys.argv[1], "FAILED", e, open(sys.argv[2]).read()[-300:])
PY
done
done
