#!/bin/bash
# rocprofv3 counter passes for the HMM kernels at 64 samples (GPU box).  Usage: bash scripts/profile_hmm_batch.sh OUTDIR
OUT=$(realpath -m ${1:-gpurun_out/prof_hmm}); R=$PWD
mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
ARGS="--rows 200000 --steps 2 --warmup 1 --no-cpu-baseline --no-merged-line --hmm-samples 64 --hmm-batch 0 --hmm-batch-large 0 --hmm-reps 2"
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $OUT/pmc_sq -- python3 $R/bench.py $ARGS > $OUT/pmc_sq.log 2>&1
rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_WR --output-format csv -d $OUT/pmc_sq2 -- python3 $R/bench.py $ARGS > $OUT/pmc_sq2.log 2>&1
cd $R
python - <<PY
import collections, csv, glob
for sub in ('pmc_sq', 'pmc_sq2'):
    for f in glob.glob('$OUT/' + sub + '/*/*_counter_collection.csv'):
        agg = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            k = r['Kernel_Name']
            if 'chain_kernel' in k or 'wave_kernel' in k or 'viterbi_bp' in k or 'outputs' in k or 'emission' in k:
                agg[k[:48]][r['Counter_Name']].append(float(r['Counter_Value']))
        print('#', sub, '(per-launch mean)')
        for k, v in agg.items():
            print(' ', k, {c: round(sum(x) / len(x) / 1e6, 3) for c, x in v.items()}, '(millions)')
PY
rm -rf $OUT/pmc_sq/*/*.csv $OUT/pmc_sq2/*/*.csv
