#!/usr/bin/env python3
"""Timeline of the kernels of the LAST reconstruct pass in a rocprofv3 --kernel-trace csv (from `emission` to the pass's last
kernel): start offset and duration of every dispatch, in microseconds.  Usage: python3 scripts/hmm_timeline.py <rocprof out dir>"""
import csv
import glob
import sys

f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f))]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"] for r in rows]
starts = [i for i, n in enumerate(names) if "emission" in n]
if len(starts) < 2:
    raise SystemExit("fewer than two passes in the trace")
lo, hi = starts[-2], starts[-1]                    # the last complete pass
t0 = int(rows[lo]["Start_Timestamp"])
end = 0
for r in rows[lo:hi]:
    s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
    end = max(end, e)
    n = r["Kernel_Name"]
    n = n[n.find("gbrs::") + 6:] if "gbrs::" in n else n
    print(f"{s / 1e3:9.1f} us  +{(e - s) / 1e3:8.1f} us   {n[:110]}")
print(f"pass: {end / 1e3:.1f} us from the emission kernel's start to the last kernel's end")
