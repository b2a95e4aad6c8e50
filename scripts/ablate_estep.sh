#!/bin/bash
# Tuning aid (GPU box): E-step kernel with pieces compiled out, interleave on/off, merged or not.
# Usage: bash scripts/ablate_estep.sh OUTDIR
OUT=${1:-gpurun_out/abl}
mkdir -p $OUT
timeout -k 10 600 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest exit $?" >> $OUT/pytest.log
for V in FULL ATOMICS; do
  if [ $V = FULL ]; then X="-DGBRS_FULL"; else X="-DGBRS_ABLATE_$V"; fi
  GBRS_HIPCC_EXTRA="$X" python -c "import __graft_entry__ as g; g.build()" > /dev/null 2>&1
  for F in 0 4; do
    timeout -k 10 200 python bench.py --steps 30 --warmup 3 --no-hmm --no-cpu-baseline --flags $F > $OUT/${V}_f$F.log 2>&1
    timeout -k 10 200 python bench.py --steps 30 --warmup 3 --no-hmm --no-cpu-baseline --flags $F --merge > $OUT/${V}_f${F}_m.log 2>&1
  done
done
GBRS_HIPCC_EXTRA="-DGBRS_FULL" python -c "import __graft_entry__ as g; g.build()" > /dev/null 2>&1
