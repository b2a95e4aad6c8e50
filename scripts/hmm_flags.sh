#!/bin/bash
# Tuning aid (GPU box): 64-sample HMM bench for a list of hmm.hip flag sets.  Usage: bash scripts/hmm_flags.sh OUTDIR "flags" ...
mkdir -p gpurun_out/$1
i=0
for f in "${@:2}"; do
  i=$((i+1))
  GBRS_HIPCC_EXTRA="-DGBRS_FULL $f" python -c "import __graft_entry__ as g; g.build()" > /dev/null 2>&1
  python bench.py --rows 200000 --steps 2 --warmup 1 --no-cpu-baseline --no-merged-line --hmm-samples 64 --hmm-batch 0 --hmm-batch-large 0 --hmm-reps 5 > gpurun_out/$1/v$i.log 2>&1
  tail -1 gpurun_out/$1/v$i.log | python -c "import sys,json; d=json.loads(sys.stdin.read())['hmm']; print('$f', round(d['ms_per_pass'],3), {k:(round(v,3) if isinstance(v,float) else v) for k,v in d['kernels_ms'].items() if k!='note'})"
done
GBRS_HIPCC_EXTRA="-DGBRS_FULL" python -c "import __graft_entry__ as g; g.build()" > /dev/null 2>&1
