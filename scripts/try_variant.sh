#!/bin/bash
# Tuning aid (GPU box): build em.hip with extra -D flags and run the C2 bench.  Usage: bash scripts/try_variant.sh OUT "-DFLAG ..."
OUT=$1; shift
mkdir -p $(dirname $OUT)
GBRS_HIPCC_EXTRA="$*" python -c "import __graft_entry__ as g; g.build()" > /dev/null 2>&1
python bench.py --steps 40 --warmup 4 --no-hmm --no-cpu-baseline > $OUT 2>&1
GBRS_HIPCC_EXTRA="-DGBRS_FULL" python -c "import __graft_entry__ as g; g.build()" > /dev/null 2>&1
