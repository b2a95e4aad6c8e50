#!/bin/bash
# Standalone (chains serialised) and concurrent kernel durations of the batched HMM line (GPU box).
OUT=$(realpath -m ${1:-gpurun_out/hmmprof}); B=${2:-64}; R=$PWD; mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
for SER in 1 0; do
  export GBRS_TUNING_HMM_SERIAL=$SER
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/ser$SER -- python3 $R/bench.py --rows 200000 --steps 2 --warmup 1 --no-cpu-baseline --no-merged-line --no-e2e --no-check --hmm-batch $B --hmm-batch-large 0 --hmm-reps 3 > $OUT/ser$SER.log 2>&1
  echo "== serial=$SER"; python3 - $OUT/ser$SER <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/*/*_kernel_stats.csv")[0]
for r in csv.DictReader(open(f)):
    n = r["Name"]
    if any(k in n for k in ("mfma", "forward_wave", "backward_wave", "viterbi_bp", "posterior", "emission_batch")):
        print("%9.1f us x%-3s %s" % (float(r["AverageNs"]) / 1e3, r["Calls"], n[:90]))
PY
done
