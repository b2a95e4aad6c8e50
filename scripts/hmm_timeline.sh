#!/bin/bash
# kernel timeline of one reconstruct pass (GPU box).  Usage: scripts/hmm_timeline.sh OUT [samples]
OUT=$(realpath -m ${1:-gpurun_out/hmm_tl}); NS=${2:-1}
R=$PWD
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --output-format csv -d $OUT -- python3 $R/scripts/hmm_only.py $NS 12 > $OUT/run.log 2>&1
cd $R
python3 scripts/hmm_timeline.py $OUT > $OUT/timeline.txt
cat $OUT/timeline.txt
