set -e
mkdir -p gpurun_out/final
python -m pytest tests -m gpu -x -q > gpurun_out/final/pytest.log 2>&1
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/final/smoke.log 2>&1
python bench.py --from-host > gpurun_out/final/bench_default.log 2>&1
python bench.py --rows 200000 --steps 2 --warmup 1 --no-cpu-baseline --no-merged-line --hmm-haps 16 --hmm-batch 8 > gpurun_out/final/bench_h16.log 2>&1
export TMPDIR=/tmp R=$PWD; cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/final/kt -- python3 $R/bench.py --no-cpu-baseline > $R/gpurun_out/final/kt.log 2>&1
cd $R
python scripts/summarize_prof.py gpurun_out/final > gpurun_out/final/kernel_stats.txt
cp gpurun_out/final/kt/*/*_kernel_stats.csv gpurun_out/final/kernel_stats.csv
rm -rf gpurun_out/final/kt
tail -2 gpurun_out/final/pytest.log
