#!/bin/bash
# alpha_mfma_kernel alone (GBRS_TUNING_HMM_SERIAL=1) and side by side, product library and the timing-only HMM_ABL_M_* variants (GPU box)
OUT=${1:-gpurun_out/ablm}; NS=${2:-256}; mkdir -p $OUT
for v in base STORES LOADS RECIP ALL; do
  if [ $v = base ]; then unset GBRS_TUNING_LIB; else export GBRS_TUNING_LIB=$PWD/gbrs_amd/variants/libgbrs_hip_ablm_$v.so; fi
  for serial in 1 0; do
    export GBRS_TUNING_HMM_SERIAL=$serial
    scripts/hmm_timeline.sh $OUT/${v}_$serial $NS > /dev/null 2>&1
    echo "$v serial=$serial: $(grep -E 'alpha_mfma' $OUT/${v}_$serial/timeline.txt | awk '{print $3, $4}') ; backward $(grep -E 'backward_mfma' $OUT/${v}_$serial/timeline.txt | awk '{print $3, $4}') ; $(grep pass: $OUT/${v}_$serial/timeline.txt)"
  done
done
