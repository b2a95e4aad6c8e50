#!/bin/bash
# Tuning aid (GPU box): build hmm.hip with extra -D flags and run the 16-founder HMM bench.
OUT=$1; shift
mkdir -p $(dirname $OUT)
GBRS_HIPCC_EXTRA="$*" python -c "import __graft_entry__ as g; g.build()" > /dev/null 2>&1
python bench.py --rows 200000 --steps 2 --warmup 1 --no-cpu-baseline --no-merged-line --hmm-haps 16 --hmm-batch 0 --hmm-batch-large 0 --hmm-reps 3 > $OUT 2>&1
GBRS_HIPCC_EXTRA="-DGBRS_FULL" python -c "import __graft_entry__ as g; g.build()" > /dev/null 2>&1
