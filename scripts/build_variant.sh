#!/bin/bash
# Build a variant of libgbrs_hip.so with extra hipcc flags on some translation units (kernel experiments), next to the
# product library: gbrs_amd/variants/libgbrs_hip_<name>.so, selected at run time with GBRS_TUNING_LIB=<path>.
# Usage: scripts/build_variant.sh NAME "<extra hipcc flags>" [units: default "em em_layout"; e.g. "hmm"]
# (run in the build container after the product build; the .so travels with gpurun)
set -e
NAME=$1; EXTRA=$2; UNITS=${3:-em em_layout}
ROOT=$(cd "$(dirname "$0")/.." && pwd); C=$ROOT/gbrs_amd/csrc; B=$C/build/variant_$NAME; mkdir -p $B $ROOT/gbrs_amd/variants
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -ffp-contract=off -Wall"
OBJS=""
for f in common em em_layout hmm hostio; do
  if [[ " $UNITS " == *" $f "* ]]; then /opt/rocm/bin/hipcc $FLAGS $EXTRA -c -o $B/$f.o $C/$f.hip & OBJS="$OBJS $B/$f.o"; else OBJS="$OBJS $C/build/$f.o"; fi
done; wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -pthread -o $ROOT/gbrs_amd/variants/libgbrs_hip_$NAME.so $OBJS -ldl
echo built gbrs_amd/variants/libgbrs_hip_$NAME.so
