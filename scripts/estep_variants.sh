#!/bin/bash
# E-step time of build / layout variants on the C2 sample (GPU box).  Usage: scripts/estep_variants.sh OUTDIR
OUT=${1:-gpurun_out/variants}; mkdir -p $OUT
run() {  # name, hipcc extra, bench flags, [extra bench args]
  GBRS_HIPCC_EXTRA="$2" python -c "import __graft_entry__ as g; g.build()" > /dev/null 2>&1
  timeout -k 10 200 python bench.py --steps 50 --warmup 5 --no-e2e --no-cpu-baseline --no-hmm --no-merged-line --flags $3 $4 > $OUT/$1.log 2>&1
  python - "$1" "$OUT/$1.log" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[2]).read().strip().split("\n")[-1])
    print("%-28s estep %.4f ms  step %.4f ms  words %d tiles %d slots %d ok %s" % (sys.argv[1], d["roofline"]["kernel_ms"], d["ms_per_step"], d["config"]["device_words"], d["config"]["tiles"], d["config"]["slots"], d["state_check"]["ok"]))
except Exception as e:
    print(sys.argv[1], "FAILED", e)
PY
}
run tables_nr2 "" 0
run tables_nr1 "-DGBRS_RECIP_NEWTON=1" 0
run notables_nr2 "-DGBRS_THETA_TABLES=0" 0
run tables_merged "" 1
run tables_c5shard "" 0 "--rows 25000000 --haps 16 --loci 200000"
run notables_c5shard "-DGBRS_THETA_TABLES=0" 0 "--rows 25000000 --haps 16 --loci 200000"
GBRS_HIPCC_EXTRA="" python -c "import __graft_entry__ as g; g.build()" > /dev/null 2>&1
touch gbrs_amd/csrc/em_tiles.inc; python -c "import __graft_entry__ as g; g.build()" > /dev/null 2>&1
