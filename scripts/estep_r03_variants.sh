#!/bin/bash
# Round-3 E-step build variants (GPU box).  Usage: scripts/estep_r03_variants.sh OUTDIR "<variant flags>;<variant flags>;..."
OUT=${1:-gpurun_out/r03_variants}; mkdir -p $OUT
IFS=';' read -ra VARS <<< "${2:--DGBRS_THETA_REGS=0;-DGBRS_THETA_REGS=1;-DGBRS_THETA_REGS=1 -DGBRS_ESTEP_WAVES=4}"
i=0
for SPEC in "${VARS[@]}"; do
  # "<hipcc flags> @ <ENV=value ...>": the part after @ is exported for the bench runs of this variant
  X="${SPEC%%@*}"; E=""; [[ "$SPEC" == *@* ]] && E="${SPEC#*@}"
  N=v${i}
  GBRS_HIPCC_EXTRA="$X" python -c "import __graft_entry__ as g; g.build()" > $OUT/build_$N.log 2>&1 || echo "BUILD FAILED: $X"
  for rep in a b; do
  env $E timeout -k 10 200 python bench.py --steps ${STEPS:-200} --warmup ${WARM:-20} --no-e2e --no-cpu-baseline --no-hmm --no-merged-line --no-multi-isoform-line $BENCH_ARGS > $OUT/${N}_$rep.log 2>&1
  python - "$SPEC" "$OUT/${N}_$rep.log" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[2]).read().strip().split("\n")[-1])
    print("%-100s estep %.4f ms  step %.4f ms  tiles %d ok %s" % (sys.argv[1], d["roofline"]["kernel_ms"], d["ms_per_step"], d["config"]["tiles"], d["state_check"]["ok"]))
except Exception as e:
    print(sys.argv[1], "FAILED", e)
PY
  done
  i=$((i+1))
done
rm -f gbrs_amd/csrc/build/em.o gbrs_amd/csrc/build/em_layout.o gbrs_amd/csrc/build/hmm.o
GBRS_HIPCC_EXTRA="" python -c "import __graft_entry__ as g; g.build()" > /dev/null 2>&1
