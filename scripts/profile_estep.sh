#!/bin/bash
# rocprofv3 passes for the EM step kernels (GPU box).  Usage: bash scripts/profile_estep.sh OUTDIR [bench args]
# Kernel trace and every counter group in its own run (never --pmc together with a trace domain).
OUT=$(realpath -m ${1:-gpurun_out/prof}); shift
R=$PWD
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
B="$R/bench.py --no-e2e --no-cpu-baseline --no-hmm --no-merged-line --no-multi-isoform-line --no-check"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -- python3 $B --steps 20 --warmup 2 "$@" > $OUT/kt.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $OUT/pmc_sq -- python3 $B --steps 3 --warmup 1 "$@" > $OUT/pmc_sq.log 2>&1
rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT --output-format csv -d $OUT/pmc_sq2 -- python3 $B --steps 3 --warmup 1 "$@" > $OUT/pmc_sq2.log 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE GRBM_COUNT --output-format csv -d $OUT/pmc_grbm -- python3 $B --steps 3 --warmup 1 "$@" > $OUT/pmc_grbm.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $B --steps 3 --warmup 1 "$@" > $OUT/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $B --steps 3 --warmup 1 "$@" > $OUT/pmc_write.log 2>&1
cd $R
