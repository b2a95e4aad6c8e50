# Regenerates the round's evidence under gpurun_out/final on one GPU box (copied into profiles/ afterwards).
# Usage: bash scripts/refresh_profiles.sh [a|b|c]   (three parts, each fits one 20-minute gpurun call)
set -e
PART=${1:-a}
O=gpurun_out/final; mkdir -p $O
R=$PWD
summ() { python - "$1" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().split("\n")[-1])
print(sys.argv[1].split("/")[-1], "n_gpus", d["n_gpus"], "path", d["path"], "ms/step %.4f" % d["ms_per_step"], "estep %.4f" % d["roofline"]["kernel_ms"], "ok", (d.get("state_check") or {}).get("ok"))
PY
}
if [ $PART = a ]; then
python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1
python -c "import __graft_entry__ as g; g.smoke()" >> $O/pytest_gpu.log 2>&1
tail -3 $O/pytest_gpu.log
python bench.py > $O/bench_default.log 2> $O/bench_default.err
python bench.py --steps 20 --warmup 5 --no-e2e --no-hmm --no-cpu-baseline --no-merged-line --no-multi-isoform-line > $O/bench_steps20.log 2>&1
python bench.py --no-e2e --no-hmm --no-cpu-baseline --no-merged-line --no-multi-isoform-line --flags 32 > $O/bench_deterministic_C2.log 2>&1
python bench.py --no-e2e --rows 25000000 --haps 16 --loci 200000 --steps 20 --warmup 2 --no-hmm --no-cpu-baseline --no-merged-line --no-multi-isoform-line > $O/bench_C5_shard_25M_x16.log 2>&1
python bench.py --no-e2e --rows 200000 --steps 2 --warmup 1 --no-cpu-baseline --no-merged-line --no-multi-isoform-line --hmm-haps 16 --hmm-batch 8 > $O/bench_hmm_16founders.log 2>&1
for f in bench_steps20 bench_deterministic_C2 bench_C5_shard_25M_x16; do summ $O/$f.log; done
fi
if [ $PART = b ]; then
python bench.py --gpus 2 --backend gloo --steps 10 --warmup 2 --no-cpu-baseline --hmm-batch-large 0 --no-multi-isoform-line > $O/bench_gloo_2ranks.log 2>&1
python bench.py --no-e2e --no-hmm --no-cpu-baseline --no-merged-line --no-multi-isoform-line --rccl-selftest --no-overlap > $O/bench_rccl_selftest_single.log 2>&1
python bench.py --no-e2e --no-hmm --no-cpu-baseline --no-merged-line --no-multi-isoform-line --rccl-selftest --force-overlap-path > $O/bench_rccl_selftest_pipelined.log 2>&1
for f in bench_gloo_2ranks bench_rccl_selftest_single bench_rccl_selftest_pipelined; do summ $O/$f.log; done
python scripts/next_rows_bench.py > $O/next_rows.json 2> $O/next_rows.err
export TMPDIR=/tmp; cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/kt -- python3 $R/bench.py --no-e2e --no-cpu-baseline > $R/$O/kt.log 2>&1
cd $R
python scripts/summarize_prof.py $O > $O/kernel_stats.txt
cp $O/kt/*/*_kernel_stats.csv $O/kernel_stats.csv
rm -rf $O/kt
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/kt_next -- python3 $R/scripts/next_rows_bench.py --calls-only > $R/$O/kt_next.log 2>&1
cd $R
python - $O <<'PY'
import csv, glob, sys
o = sys.argv[1]
with open(o + "/next_rows_kernel_stats.txt", "w") as out:
    out.write("# rocprofv3 --kernel-trace --stats -- python3 scripts/next_rows_bench.py --calls-only\n")
    for f in glob.glob(o + "/kt_next/*/*_kernel_stats.csv"):
        for r in csv.DictReader(open(f)):
            if "gbrs" in r["Name"]:
                out.write("%-110s calls=%5s avg_us=%10.1f total_ms=%9.2f\n" % (r["Name"][:110], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6))
PY
rm -rf $O/kt_next
fi
if [ $PART = c ]; then
for ns in 1 2 4 64 256; do bash scripts/hmm_kt_env.sh $O/hmm_kt_$ns $ns A=1 > $O/hmm_kernels_$ns.txt 2>&1; rm -rf $O/hmm_kt_$ns $O/hmm_kt_$ns.log; done
GBRS_TUNING_HMM_BLOCKED=0 python scripts/hmm_only.py 1 20 > $O/hmm_unblocked_1.txt 2>&1
GBRS_TUNING_HMM_BLOCKED=0 python scripts/hmm_only.py 2 20 >> $O/hmm_unblocked_1.txt 2>&1
python scripts/short_run_clock.py > $O/short_run_clock.txt 2>&1
bash scripts/profile_estep.sh $O/prof_estep
python scripts/summarize_prof.py $O/prof_estep > $O/rocprof_estep_C2.txt
cp $O/prof_estep/kt/*/*_kernel_stats.csv $O/rocprof_estep_kernel_stats_C2.csv
rm -rf $O/prof_estep
python scripts/fuzz_parity.py 300 > $O/fuzz_parity.log 2>&1
tail -3 $O/fuzz_parity.log
fi
