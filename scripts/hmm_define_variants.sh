#!/bin/bash
# Compile-time variants of the batched HMM kernels (GPU box): rebuilds the library with each set of defines.
# Usage: scripts/hmm_define_variants.sh OUTDIR "BATCH ..." "-DHMM_NSET_M=4" "-DHMM_DL_AHEAD=8 -DHMM_NSET_M=4" ...   ("" = defaults)
OUT=${1:-gpurun_out/hmmdef}; BATCHES=${2:-256}; shift 2; mkdir -p $OUT
I=0
for DEF in "$@"; do I=$((I+1))
  rm -f gbrs_amd/csrc/build/hmm.o
  GBRS_HIPCC_EXTRA="$DEF" python -c "import __graft_entry__ as g; g.build()" > $OUT/build$I.log 2>&1 || { echo "BUILD FAILED: $DEF"; continue; }
  for B in $BATCHES; do
    timeout -k 10 300 python bench.py --rows 200000 --steps 2 --warmup 1 --no-cpu-baseline --no-merged-line --no-e2e --hmm-batch $B --hmm-batch-large 0 --hmm-reps 3 > $OUT/v${I}_b$B.log 2>&1
    python - $OUT/v${I}_b$B.log $B "$DEF" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().split("\n")[-1])["hmm"]["batched"]
    print("batch %s [%s]: %.3f ms/pass %.1f M genes/s  kernels %s" % (sys.argv[2], sys.argv[3], d["ms_per_pass"], d["value"] / 1e6, {k: round(v, 3) for k, v in d["kernels_ms"].items() if k != "note"}))
except Exception as e:
    print(sys.argv[1], "FAILED", e)
PY
  done
done
rm -f gbrs_amd/csrc/build/hmm.o; GBRS_HIPCC_EXTRA="" python -c "import __graft_entry__ as g; g.build()" > /dev/null 2>&1
