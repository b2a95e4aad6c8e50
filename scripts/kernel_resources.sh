#!/bin/bash
# Register / scratch / LDS / occupancy table of the kernels in one .hip file (device-only compile).
# Usage: scripts/kernel_resources.sh gbrs_amd/csrc/em.hip [name filter] [extra hipcc flags...]
SRC=$1; FILTER=${2:-.}; shift 2
cd "$(dirname "$SRC")" && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -munsafe-fp-atomics -ffp-contract=off \
  --cuda-device-only -Rpass-analysis=kernel-resource-usage "$@" -c -o /dev/null "$(basename "$SRC")" 2>&1 | python3 -c '
import re, subprocess, sys
flt = sys.argv[1]
cur = None; rows = []
for line in sys.stdin:
    m = re.search(r"remark: (?:\s*)(Function Name|VGPRs|AGPRs|SGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|LDS Size \[bytes/block\]|VGPRs Spill|SGPRs Spill): (\S+)", line)
    if not m: continue
    k, v = m.group(1), m.group(2)
    if k == "Function Name":
        cur = {"name": v}; rows.append(cur)
    elif cur is not None:
        cur[k.split(" [")[0]] = v
names = subprocess.run(["c++filt"], input="\n".join(r["name"] for r in rows), capture_output=True, text=True).stdout.split("\n")
print("%-90s %5s %5s %5s %7s %4s %6s" % ("kernel", "VGPR", "AGPR", "SGPR", "scratch", "occ", "LDS"))
for r, n in zip(rows, names):
    n = re.sub(r"^void ", "", n); n = re.sub(r"\(.*", "", n)
    if re.search(flt, n):
        print("%-90s %5s %5s %5s %7s %4s %6s" % (n[:90], r.get("VGPRs"), r.get("AGPRs"), r.get("SGPRs"), r.get("ScratchSize"), r.get("Occupancy"), r.get("LDS Size")))
' "$FILTER"
