#!/bin/bash
# Tuning aid (GPU box): E-step kernel with pieces compiled out.  Usage: bash scripts/ablate_estep.sh OUTDIR
OUT=${1:-gpurun_out/abl}
mkdir -p $OUT
export GBRS_TUNING_NO_FLOAT_CHECK=1
i=0
for X in "-DGBRS_FULL" "-DGBRS_ABLATE_ATOMICS" "-DGBRS_ABLATE_THETA" "-DGBRS_ABLATE_ROWSUM" \
         "-DGBRS_ABLATE_ATOMICS -DGBRS_ABLATE_THETA -DGBRS_ABLATE_ROWSUM" "-DGBRS_ABLATE_BATCHES"; do
  N=v${i}_$(echo "$X" | sed 's/-DGBRS_//g; s/ABLATE_//g; s/ /+/g')
  GBRS_HIPCC_EXTRA="$X" python -c "import __graft_entry__ as g; g.build()" > /dev/null 2>&1
  timeout -k 10 200 python bench.py --steps 30 --warmup 3 --no-hmm --no-cpu-baseline > $OUT/${N}.log 2>&1
  timeout -k 10 200 python bench.py --steps 30 --warmup 3 --no-hmm --no-cpu-baseline --merge > $OUT/${N}_m.log 2>&1
  i=$((i+1))
done
GBRS_HIPCC_EXTRA="-DGBRS_FULL" python -c "import __graft_entry__ as g; g.build()" > /dev/null 2>&1
