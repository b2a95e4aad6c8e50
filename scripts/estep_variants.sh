#!/bin/bash
# E-step time of build variants (GPU box).  Usage: scripts/estep_variants.sh OUTDIR
OUT=${1:-gpurun_out/variants}; mkdir -p $OUT
build() { GBRS_HIPCC_EXTRA="$1" python -c "import __graft_entry__ as g; g.build()" > $OUT/build.log 2>&1 || echo "BUILD FAILED: $1"; }
bench() {  # name, bench flags, [extra bench args]
  timeout -k 10 200 python bench.py --steps 200 --warmup 20 --no-e2e --no-cpu-baseline --no-hmm --no-merged-line --flags $2 $3 > $OUT/$1.log 2>&1
  python - "$1" "$OUT/$1.log" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[2]).read().strip().split("\n")[-1])
    print("%-34s estep %.4f ms  step %.4f ms  words %d tiles %d slots %d ok %s" % (sys.argv[1], d["roofline"]["kernel_ms"], d["ms_per_step"], d["config"]["device_words"], d["config"]["tiles"], d["config"]["slots"], d["state_check"]["ok"]))
except Exception as e:
    print(sys.argv[1], "FAILED", e)
PY
}
C5="--rows 25000000 --haps 16 --loci 200000"
build ""; bench base_c2 0; bench base_c2_b 0; bench base_c5 0 "$C5"; bench base_c2_merge 0 "--merge"
build "-DGBRS_RAW_PD=6"; bench pd6_c2 0
build "-DGBRS_FTAB=0"; bench noftab_c2 0
build "-DGBRS_THETA_PLANES=0"; bench rowmajor_c2 0; bench rowmajor_c5 0 "$C5"
rm -f gbrs_amd/csrc/build/em.o gbrs_amd/csrc/build/em_layout.o gbrs_amd/csrc/build/hmm.o
GBRS_HIPCC_EXTRA="" python -c "import __graft_entry__ as g; g.build()" > /dev/null 2>&1
