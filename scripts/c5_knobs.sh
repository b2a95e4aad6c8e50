#!/bin/bash
# config-5 shard (25M reads x 16 haplotypes x 200k loci): iteration and E-step time under layout / kernel switches (GPU box).
# Usage: scripts/c5_knobs.sh OUT
OUT=${1:-gpurun_out/c5_knobs}; mkdir -p $OUT
run() {  # name, env...
  local name=$1; shift
  env "$@" python bench.py --rows 25000000 --haps 16 --loci 200000 --no-hmm --no-e2e --no-cpu-baseline --no-merged-line \
      --no-multi-isoform-line --no-check --no-solve > $OUT/$name.json 2> $OUT/$name.err
  python3 - "$OUT/$name.json" "$name" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(f"{sys.argv[2]:28s} ms_per_step {d['ms_per_step']:.4f}  cold {d['cold_start']['ms_per_step']:.4f}  E-step {d['roofline'].get('kernel_ms', 0):.4f}  "
      f"words/read {d['config'].get('words_per_read', d['roofline'].get('words_per_read', 0))}  tiles {d['config'].get('tiles')}  sets {d['config'].get('locus_sets')}")
PY
}
run base GBRS_X=0
run base2 GBRS_X=0
run min_rows_96 GBRS_TUNING_SET_MIN_ROWS=96
run min_rows_384 GBRS_TUNING_SET_MIN_ROWS=384
run no_group_sets GBRS_TUNING_GROUP_SETS=0
run persistent GBRS_TUNING_PERSISTENT=1
run dict_cap_128 GBRS_TUNING_DICT_CAP=128
run dict_cap_96 GBRS_TUNING_DICT_CAP=96
