#!/bin/bash
# A/B on one box: the E-step on persistent workgroups (default) against one workgroup per tile (GBRS_TUNING_PERSISTENT=0),
# alternating, driver form (--steps 20 --warmup 5) and settled (--steps 300 --warmup 30).
# Usage: scripts/ab_persistent.sh [reps] ["ENV=..." applied to both sides]
reps=${1:-3}; extra_env=${2:-}
out=gpurun_out/ab_persistent.txt; : > $out
for rep in $(seq $reps); do
  for p in 1 0; do
    for form in "20 5" "300 30"; do
      st=${form% *}; wu=${form#* }
      env $extra_env GBRS_TUNING_PERSISTENT=$p python bench.py --steps $st --warmup $wu --no-e2e --no-cpu-baseline --no-hmm --no-merged-line --no-multi-isoform-line --no-check 2>/dev/null |
        python -c "import sys,json; l=[x for x in sys.stdin if x.startswith('{')][-1]; p=json.loads(l); print('persistent=$p steps=$st', 'ms_per_step %.4f' % p['ms_per_step'], 'estep %.4f' % p['roofline']['kernel_ms'], 'tiles', p['config']['tiles'], 'ok', p['state_check']['ok'])" >> $out
    done
  done
done
cat $out
