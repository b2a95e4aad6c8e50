#!/bin/bash
# Tuning aid (GPU box): build em.hip with extra -D flags and run a C5-like shard (25M reads x 16 x 200k).
OUT=$1; shift
mkdir -p $(dirname $OUT)
GBRS_HIPCC_EXTRA="$*" python -c "import __graft_entry__ as g; g.build()" > /dev/null 2>&1
python bench.py --rows 25000000 --haps 16 --loci 200000 --steps 20 --warmup 2 --no-hmm --no-cpu-baseline --no-merged-line > $OUT 2>&1
GBRS_HIPCC_EXTRA="-DGBRS_FULL" python -c "import __graft_entry__ as g; g.build()" > /dev/null 2>&1
