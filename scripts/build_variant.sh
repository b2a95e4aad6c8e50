#!/bin/bash
# Build a variant of libgbrs_hip.so with extra hipcc flags on the EM translation units (kernel experiments), next to the
# product library: gbrs_amd/variants/libgbrs_hip_<name>.so, selected at run time with GBRS_TUNING_LIB=<path>.
# Usage: scripts/build_variant.sh NAME "<extra hipcc flags>"      (run in the build container; the .so travels with gpurun)
set -e
NAME=$1; EXTRA=$2
ROOT=$(cd "$(dirname "$0")/.." && pwd); C=$ROOT/gbrs_amd/csrc; B=$C/build/variant_$NAME; mkdir -p $B $ROOT/gbrs_amd/variants
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -ffp-contract=off -Wall"
for f in em em_layout; do /opt/rocm/bin/hipcc $FLAGS $EXTRA -c -o $B/$f.o $C/$f.hip & done; wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -pthread -o $ROOT/gbrs_amd/variants/libgbrs_hip_$NAME.so \
  $C/build/common.o $B/em.o $B/em_layout.o $C/build/hmm.o $C/build/hostio.o -ldl
echo built gbrs_amd/variants/libgbrs_hip_$NAME.so
