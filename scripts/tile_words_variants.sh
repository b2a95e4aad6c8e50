#!/bin/bash
# E-step time against the tile size (words per tile), C2 sample (GPU box).  Usage: scripts/tile_words_variants.sh OUTDIR WORDS...
# (GBRS_TUNING_TILE_WORDS overrides build_tile_layout's rule at run time; GBRS_TUNING_TILE_ORDER=0 for the locus order)
OUT=${1:-gpurun_out/tilewords}; shift; mkdir -p $OUT
python -c "import __graft_entry__ as g; g.build()" > $OUT/build.log 2>&1 || { echo "BUILD FAILED"; exit 1; }
for W in "$@"; do
  GBRS_TUNING_TILE_WORDS=$W timeout -k 10 200 python bench.py --steps 300 --warmup 30 --no-e2e --no-cpu-baseline --no-hmm --no-merged-line > $OUT/w$W.log 2>&1
  python - $W $OUT/w$W.log <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[2]).read().strip().split("\n")[-1])
    print("tile words %-6s estep %.4f ms  step %.4f ms  tiles %d (%.2f rounds of 768) slots %d ok %s" % (sys.argv[1], d["roofline"]["kernel_ms"], d["ms_per_step"], d["config"]["tiles"], d["config"]["tiles"] / 768.0, d["config"]["slots"], d["state_check"]["ok"]))
except Exception as e:
    print(sys.argv[1], "FAILED", e)
PY
done
