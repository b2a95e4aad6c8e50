#!/bin/bash
# Batched HMM line of bench.py under the tuning switches (GPU box).
# Usage: scripts/hmm_batch_variants.sh OUTDIR BATCH "MFMA_MIN DLANES_MIN" [...]   (GBRS_TUNING_HMM_MFMA / _DLANES: smallest batch on the
# MFMA sweeps / the samples-on-lanes delta chain, 0 = never)
OUT=${1:-gpurun_out/hmmb}; B=${2:-64}; shift 2; mkdir -p $OUT
for V in "$@"; do set -- $V; M=$1; D=${2:-0}
  GBRS_TUNING_HMM_DLANES=$D GBRS_TUNING_HMM_MFMA=$M timeout -k 10 300 python bench.py --rows 200000 --steps 2 --warmup 1 --no-cpu-baseline --no-merged-line --no-e2e --hmm-batch $B --hmm-batch-large 0 --hmm-reps 3 > $OUT/b${B}_m${M}_d${D}.log 2>&1
  python - $OUT/b${B}_m${M}_d${D}.log $B "min $M dlanes $D" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().split("\n")[-1])["hmm"]["batched"]
    print("batch %s mfma %s: %.3f ms/pass %.1f M genes/s  kernels %s" % (sys.argv[2], sys.argv[3], d["ms_per_pass"], d["value"] / 1e6, {k: round(v, 3) for k, v in d["kernels_ms"].items() if k != "note"}))
except Exception as e:
    print(sys.argv[1], "FAILED", e)
PY
done
