#!/bin/bash
# The four EM workloads (config-5 shard, merged rows, deterministic, C2) under build variants (GPU box); run-time switches
# such as GBRS_TUNING_TILE_WORDS / GBRS_TUNING_TILE_ORDER are inherited from the environment.
# Usage: scripts/tile_words_check.sh OUTDIR "DEFINES" ["DEFINES" ...]      ("" = the default build)
OUT=${1:-gpurun_out/tilecheck}; shift; mkdir -p $OUT
I=0
for DEF in "$@"; do I=$((I+1))
  rm -f gbrs_amd/csrc/build/em.o gbrs_amd/csrc/build/em_layout.o
  GBRS_HIPCC_EXTRA="$DEF" python -c "import __graft_entry__ as g; g.build()" > $OUT/build.log 2>&1 || { echo "BUILD FAILED $DEF"; continue; }
  for V in "c5 --rows 25000000 --haps 16 --loci 200000 --steps 50 --warmup 5" "merge --merge --steps 300 --warmup 30" "det --flags 32 --steps 50 --warmup 5" "c2 --steps 500 --warmup 50"; do
    set -- $V; N=$1; shift
    timeout -k 10 250 python bench.py --no-e2e --no-cpu-baseline --no-hmm --no-merged-line "$@" > $OUT/v${I}_$N.log 2>&1
    python - "$DEF" $N $OUT/v${I}_$N.log <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[3]).read().strip().split("\n")[-1])
    print("[%s] %-6s estep %.4f ms  step %.4f ms  tiles %d ok %s" % (sys.argv[1], sys.argv[2], d["roofline"]["kernel_ms"], d["ms_per_step"], d["config"]["tiles"], d["state_check"]["ok"]))
except Exception as e:
    print(sys.argv[1], sys.argv[2], "FAILED", e)
PY
  done
done
rm -f gbrs_amd/csrc/build/em.o gbrs_amd/csrc/build/em_layout.o
GBRS_HIPCC_EXTRA="" python -c "import __graft_entry__ as g; g.build()" > /dev/null 2>&1
