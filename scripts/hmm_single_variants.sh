#!/bin/bash
# Compile-time variants of the one-sample HMM kernels (GPU box).  Usage: scripts/hmm_single_variants.sh OUTDIR "DEFINES" ...
OUT=${1:-gpurun_out/hmmsingle}; shift; mkdir -p $OUT
I=0
for DEF in "$@"; do I=$((I+1))
  rm -f gbrs_amd/csrc/build/hmm.o
  GBRS_HIPCC_EXTRA="$DEF" python -c "import __graft_entry__ as g; g.build()" > $OUT/build$I.log 2>&1 || { echo "BUILD FAILED: $DEF"; continue; }
  timeout -k 10 300 python bench.py --rows 200000 --steps 2 --warmup 1 --no-cpu-baseline --no-merged-line --no-e2e --hmm-batch 0 --hmm-batch-large 0 > $OUT/v$I.log 2>&1
  python - $OUT/v$I.log "$DEF" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().split("\n")[-1])["hmm"]
    print("[%s]: %.3f ms/pass %.1f M genes/s  kernels %s" % (sys.argv[2], d["ms_per_pass"], d["value"] / 1e6, {k: round(v, 3) for k, v in d["kernels_ms"].items() if k != "note"}))
except Exception as e:
    print(sys.argv[1], "FAILED", e)
PY
done
rm -f gbrs_amd/csrc/build/hmm.o; GBRS_HIPCC_EXTRA="" python -c "import __graft_entry__ as g; g.build()" > /dev/null 2>&1
