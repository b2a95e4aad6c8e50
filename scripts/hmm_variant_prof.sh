#!/bin/bash
# Tuning aid (GPU box): build hmm.hip with extra -D flags and record per-kernel times of one
# single-sample HMM bench run.  Usage: bash scripts/hmm_variant_prof.sh OUTDIR "-DFLAG ..."
OUT=$1; shift
mkdir -p $OUT
GBRS_HIPCC_EXTRA="$*" python -c "import __graft_entry__ as g; g.build()" > /dev/null 2>&1
export TMPDIR=/tmp; R=$PWD; cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/kt -- python3 $R/bench.py --rows 200000 --steps 2 --warmup 1 --no-cpu-baseline --no-merged-line --hmm-batch 0 --hmm-batch-large 0 > $R/$OUT/kt.log 2>&1
cd $R; python scripts/summarize_prof.py $OUT > $OUT/summary.txt; rm -f $OUT/kt/*/*trace.csv
GBRS_HIPCC_EXTRA="-DGBRS_FULL" python -c "import __graft_entry__ as g; g.build()" > /dev/null 2>&1
