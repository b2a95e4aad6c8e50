#!/bin/bash
# Tuning aid (GPU box): E-step kernel with pieces compiled out (timing only: the results of an ablated build are wrong).
# Usage: bash scripts/ablate_estep.sh OUTDIR
OUT=${1:-gpurun_out/abl}
mkdir -p $OUT
export GBRS_TUNING_NO_FLOAT_CHECK=1
i=0
for X in ${GBRS_ABLATIONS:-"-DGBRS_FULL" "-DGBRS_ABLATE_ATOMICS" "-DGBRS_ABLATE_FLUSH" "-DGBRS_ABLATE_ROWSUM" "-DGBRS_ABLATE_BATCHES" "-DGBRS_ABLATE_PROLOGUE"}; do
  N=v${i}_$(echo "$X" | sed 's/-DGBRS_//g; s/ABLATE_//g; s/ /+/g')
  GBRS_HIPCC_EXTRA="$X" python -c "import __graft_entry__ as g; g.build()" > /dev/null 2>&1
  timeout -k 10 200 python bench.py --steps 200 --warmup 20 --no-e2e --no-check --no-hmm --no-cpu-baseline --no-merged-line > $OUT/${N}.log 2>&1
  python - $OUT/${N}.log "$X" <<'PY'
import json, sys
try:
    d = [json.loads(l) for l in open(sys.argv[1]).read().strip().split("\n") if l.startswith("{")][-1]
    print("%-60s estep %.4f ms  step %.4f ms" % (sys.argv[2], d["roofline"]["kernel_ms"], d["ms_per_step"]))
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
  i=$((i+1))
done
rm -f gbrs_amd/csrc/build/em.o gbrs_amd/csrc/build/em_layout.o gbrs_amd/csrc/build/hmm.o
GBRS_HIPCC_EXTRA="" python -c "import __graft_entry__ as g; g.build()" > /dev/null 2>&1
