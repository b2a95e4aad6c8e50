#!/bin/bash
# rocprofv3 kernel trace + MFMA / VALU counter pass for the 256-sample HMM line (GPU box).  Usage: bash scripts/profile_hmm_mfma.sh OUTDIR
OUT=$(realpath -m ${1:-gpurun_out/prof_hmm_mfma}); R=$PWD
mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
ARGS="--rows 200000 --steps 2 --warmup 1 --no-cpu-baseline --no-merged-line --no-e2e --no-check --hmm-samples 256 --hmm-batch 0 --hmm-batch-large 0 --hmm-reps 2"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -- python3 $R/bench.py $ARGS > $OUT/kt.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU_MFMA_F64 SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc -- python3 $R/bench.py $ARGS > $OUT/pmc.log 2>&1
cd $R
python - <<PY
import collections, csv, glob
print('# 40,000 genes x 36 states x 256 samples per launch; rocprofv3 --kernel-trace --stats')
for f in glob.glob('$OUT/kt/*/*_kernel_stats.csv'):
    for r in csv.DictReader(open(f)):
        n = r['Name']
        if any(k in n for k in ('mfma', 'forward_wave', 'backward_wave', 'viterbi_bp', 'posterior', 'emission_batch', 'backtrace')):
            print('%10.1f us x%-3s %s' % (float(r['AverageNs']) / 1e3, r['Calls'], n[:100]))
print('# rocprofv3 --pmc (own pass), per-launch mean')
for f in glob.glob('$OUT/pmc/*/*_counter_collection.csv'):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name']
        if 'mfma_kernel' in k or 'forward_wave' in k or 'backward_wave' in k:
            agg[k[:60]][r['Counter_Name']].append(float(r['Counter_Value']))
    for k, v in agg.items():
        print(' ', k, {c: round(sum(x) / len(x)) for c, x in v.items()})
PY
rm -rf $OUT/pmc/*/*.csv $OUT/kt/*/*_kernel_trace.csv
