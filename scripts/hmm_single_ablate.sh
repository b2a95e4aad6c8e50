#!/bin/bash
# Timing-only ablations of the one-sample forward chains (GPU box): per-kernel durations from rocprofv3.
# Usage: scripts/hmm_single_ablate.sh OUTDIR "DEFINES" ...      (results of ablated builds are meaningless)
OUT=$(realpath -m ${1:-gpurun_out/hmmabl}); shift; mkdir -p $OUT
R=$PWD; export TMPDIR=/tmp
I=0
for DEF in "$@"; do I=$((I+1))
  rm -f gbrs_amd/csrc/build/hmm.o
  GBRS_HIPCC_EXTRA="$DEF" python -c "import __graft_entry__ as g; g.build()" > $OUT/build$I.log 2>&1 || { echo "BUILD FAILED: $DEF"; continue; }
  rm -rf $OUT/kt$I
  (cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt$I -- python3 $R/bench.py --rows 200000 --steps 2 --warmup 1 --no-cpu-baseline --no-merged-line --no-e2e --hmm-batch 0 --hmm-batch-large 0 --hmm-reps 21 > $OUT/v$I.log 2>&1)
  python - "$DEF" $OUT/kt$I <<'PY'
import csv, glob, sys
out = []
for f in glob.glob(sys.argv[2] + '/*/*_kernel_stats.csv'):
    for r in csv.DictReader(open(f)):
        n = r['Name']
        if 'forward_wave_kernel<36, 3, 1, 0' in n: out.append(('alpha', r))
        if 'forward_wave_kernel<36, 3, 1, 1' in n: out.append(('delta', r))
        if 'backward_wave_kernel<36, 3, 1' in n: out.append(('backward', r))
print('[%s] ' % sys.argv[1] + '  '.join('%s %.0f / %.0f / %.0f us' % (k, float(r['AverageNs']) / 1e3, float(r['MinNs']) / 1e3, float(r['MaxNs']) / 1e3) for k, r in sorted(out)))
PY
done
rm -f gbrs_amd/csrc/build/hmm.o; GBRS_HIPCC_EXTRA="" python -c "import __graft_entry__ as g; g.build()" > /dev/null 2>&1
