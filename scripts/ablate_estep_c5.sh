#!/bin/bash
# Tuning aid (GPU box): E-step ablation on a C5-like shard (25M reads x 16 x 200k).  Usage: bash scripts/ablate_estep_c5.sh OUTDIR
OUT=${1:-gpurun_out/abl_c5}
mkdir -p $OUT
export GBRS_TUNING_NO_FLOAT_CHECK=1
for X in "-DGBRS_FULL" "-DGBRS_ABLATE_ATOMICS" "-DGBRS_ABLATE_THETA" "-DGBRS_ABLATE_ROWSUM" \
         "-DGBRS_ABLATE_ATOMICS -DGBRS_ABLATE_THETA -DGBRS_ABLATE_ROWSUM" "-DGBRS_ABLATE_BATCHES" "-DGBRS_NO_MASK_ZLO" "-DGBRS_H16_PD=4" "-DGBRS_H16_PD=16"; do
  N=$(echo "$X" | sed 's/-DGBRS_//g; s/ABLATE_//g; s/ /+/g')
  bash scripts/try_variant_c5.sh $OUT/$N.log "$X"
  tail -1 $OUT/$N.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$N', 'step', round(d['ms_per_step'],4), 'estep', round(d['roofline']['kernel_ms'],4))"
done
