#!/usr/bin/env python3
"""Where does a fresh process spend gbrs_em_create?  Loads an alignment file and builds the handle twice
with GBRS_TUNING_BUILD_TIMES=1 (no torch in the process).  Usage: cold_create_probe.py FILE [prewarm_gb]"""
import ctypes as C
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["GBRS_TUNING_BUILD_TIMES"] = "1"
from gbrs_amd.alignment import load_alignment  # noqa: E402
from gbrs_amd.engine import EmEngine  # noqa: E402

t0 = time.perf_counter()
apm = load_alignment(sys.argv[1])
print(f"load {time.perf_counter() - t0:.3f} s", file=sys.stderr)
if len(sys.argv) > 2 and float(sys.argv[2]) > 0:
    hip = C.CDLL("libamdhip64.so")
    p = C.c_void_p()
    n = int(float(sys.argv[2]) * (1 << 30))
    t0 = time.perf_counter()
    assert hip.hipMalloc(C.byref(p), C.c_size_t(n)) == 0
    assert hip.hipMemset(p, 0, C.c_size_t(n)) == 0
    assert hip.hipDeviceSynchronize() == 0
    assert hip.hipFree(p) == 0
    print(f"prewarm {sys.argv[2]} GiB: {time.perf_counter() - t0:.3f} s", file=sys.stderr)
L, H, R = apm.shape
for k in range(2):
    t0 = time.perf_counter()
    e = EmEngine.from_host(R, L, H, apm.indptr, apm.indices, apm.count, None)
    print(f"=== create #{k}: {time.perf_counter() - t0:.3f} s", file=sys.stderr)
    e.close()
