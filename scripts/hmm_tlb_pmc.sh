#!/bin/bash
# address-translation counters of the batch pass's kernels (GPU box; counters serialise the kernels).  Usage: scripts/hmm_tlb_pmc.sh OUT [samples]
OUT=$(realpath -m ${1:-gpurun_out/tlb}); NS=${2:-256}; R=$PWD; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && timeout -k 10 500 rocprofv3 --pmc TCP_UTCL1_REQUEST_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_THRASHING_STALL_sum --kernel-trace --output-format csv -d $OUT -- python3 $R/scripts/hmm_only.py $NS 2 > $OUT/run.log 2>&1
cd $R
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)[0]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"]; k = k[k.find("gbrs::") + 6:] if "gbrs::" in k else k
    k = k.split("(")[0]
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
    n[(k, r["Counter_Name"])] += 1
for k, d in acc.items():
    calls = max(n[(k, c)] for c in d)
    req = d.get("TCP_UTCL1_REQUEST_sum", 0) / calls; hit = d.get("TCP_UTCL1_TRANSLATION_HIT_sum", 0) / calls
    miss = d.get("TCP_UTCL1_TRANSLATION_MISS_sum", 0) / calls; thr = d.get("TCP_UTCL1_THRASHING_STALL_sum", 0) / calls
    if req > 1e5:
        print(f"{k[:60]:60s} calls {calls:3d}  requests {req:12.0f}  hits {hit:12.0f}  misses {miss:12.0f} ({100 * miss / max(req, 1):5.2f} %)  thrashing stalls {thr:10.0f}")
PY
