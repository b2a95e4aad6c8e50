#!/bin/bash
# samples-per-wave of the vector delta chain at a large batch (GPU box)
OUT=${1:-gpurun_out/hmmsb}; B=${2:-256}; mkdir -p $OUT
for SB in 4 8; do
  GBRS_HIPCC_EXTRA="-DHMM_SB=$SB" python -c "import __graft_entry__ as g; g.build()" > $OUT/build.log 2>&1 || echo BUILD FAILED
  timeout -k 10 300 python bench.py --rows 200000 --steps 2 --warmup 1 --no-cpu-baseline --no-merged-line --no-e2e --hmm-batch $B --hmm-batch-large 0 --hmm-reps 3 > $OUT/sb$SB.log 2>&1
  python - $OUT/sb$SB.log $SB <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().split("\n")[-1])["hmm"]["batched"]
    print("SB %s: %.3f ms/pass %.1f M genes/s  kernels %s" % (sys.argv[2], d["ms_per_pass"], d["value"] / 1e6, {k: round(v, 3) for k, v in d["kernels_ms"].items() if k != "note"}))
except Exception as e:
    print(sys.argv[1], "FAILED", e)
PY
done
rm -f gbrs_amd/csrc/build/hmm.o; GBRS_HIPCC_EXTRA="" python -c "import __graft_entry__ as g; g.build()" > /dev/null 2>&1
