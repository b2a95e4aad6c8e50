#!/bin/bash
# Single-sample HMM time under environment switches (GPU box).  Usage: scripts/hmm_single_ab.sh "<ENV=..>;<ENV=..>" [bench args]
IFS=';' read -ra VARS <<< "$1"
shift
for X in "${VARS[@]}"; do
  for rep in a b; do
  env $X timeout -k 10 300 python bench.py --no-e2e --no-cpu-baseline --no-merged-line --no-multi-isoform-line --no-check --steps 20 --warmup 5 --hmm-batch 0 --hmm-batch-large 0 "$@" > /tmp/hmm_ab.log 2>&1
  python - "$X" /tmp/hmm_ab.log <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[2]).read().strip().split("\n")[-1])["hmm"]
    k = d["kernels_ms"]
    print("%-60s pass %.4f ms (wall %.3f)  emission %.3f fwd %.3f bwd %.3f bt %.3f run %.3f  %.1f M genes/s" % (sys.argv[1], d["ms_per_pass"], d["wall_clock"]["ms_per_pass"], k["emission"], k["forward_viterbi"], k["backward_posterior"], k["backtrace"], k["run"], d["value"] / 1e6))
except Exception as e:
    print(sys.argv[1], "FAILED", e, open(sys.argv[2]).read()[-500:])
PY
  done
done
