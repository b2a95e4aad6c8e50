# Regenerates the round's evidence under gpurun_out/final on one GPU box (copied into profiles/ afterwards).
set -e
O=gpurun_out/final; mkdir -p $O
python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1
python -c "import __graft_entry__ as g; g.smoke()" >> $O/pytest_gpu.log 2>&1
python bench.py > $O/bench_default.log 2> $O/bench_default.err
python bench.py --no-e2e --no-hmm --no-cpu-baseline --no-merged-line --flags 32 > $O/bench_deterministic_C2.log 2>&1
python bench.py --no-e2e --rows 25000000 --haps 16 --loci 200000 --steps 20 --warmup 2 --no-hmm --no-cpu-baseline --no-merged-line > $O/bench_C5_shard_25M_x16.log 2>&1
python bench.py --no-e2e --rows 200000 --steps 2 --warmup 1 --no-cpu-baseline --no-merged-line --hmm-haps 16 --hmm-batch 8 > $O/bench_hmm_16founders.log 2>&1
python bench.py --gpus 2 --backend gloo --steps 10 --warmup 2 --no-cpu-baseline --hmm-batch-large 0 > $O/bench_gloo_2ranks.log 2>&1
python bench.py --no-e2e --no-hmm --no-cpu-baseline --no-merged-line --rccl-selftest --no-overlap > $O/bench_rccl_selftest_single.log 2>&1
python bench.py --no-e2e --no-hmm --no-cpu-baseline --no-merged-line --rccl-selftest --force-overlap-path > $O/bench_rccl_selftest_pipelined.log 2>&1
export TMPDIR=/tmp R=$PWD; cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/kt -- python3 $R/bench.py --no-e2e --no-cpu-baseline > $R/$O/kt.log 2>&1
cd $R
python scripts/summarize_prof.py $O > $O/kernel_stats.txt
cp $O/kt/*/*_kernel_stats.csv $O/kernel_stats.csv
rm -rf $O/kt
tail -2 $O/pytest_gpu.log
for f in bench_deterministic_C2 bench_C5_shard_25M_x16 bench_gloo_2ranks bench_rccl_selftest_single bench_rccl_selftest_pipelined; do python - $O/$f.log <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().split("\n")[-1])
print(sys.argv[1].split("/")[-1], "n_gpus", d["n_gpus"], "path", d["path"], "ms/step %.4f" % d["ms_per_step"], "estep %.4f" % d["roofline"]["kernel_ms"], "ok", (d.get("state_check") or {}).get("ok"))
PY
done
