#!/usr/bin/env python3
"""The persistent E-step kernel's life, tile by tile.  Needs a library built with GBRS_HIPCC_EXTRA=-DGBRS_DIAG_TILE_TIMES:
thread 0 of the workgroup that has tile t records (s_memrealtime, 10 ns) 0 loop start, 1 loop end (its wavefront),
2 sums handed in, 3 closing barrier passed, 4 epilogue stores issued, 5 next theta in LDS, 6 next tile's opening barrier passed.
Usage: python3 scripts/tile_timeline_persistent.py [rows]"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402
from gbrs_amd import _lib, synth, synth_torch  # noqa: E402
from gbrs_amd.engine import EmEngine  # noqa: E402

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 40_000_000
prob = synth_torch.make_em_problem_device(rows, 8, 120_000, synth.SEED_BASE_EM + 1, "cuda:0", row_seed=synth.SEED_BASE_EM + 1)
eng = EmEngine.from_device(prob["R"], prob["L"], prob["H"], [t.data_ptr() for t in prob["indptr"]],
                           [t.data_ptr() for t in prob["indices"]], None, prob["eff_len"].data_ptr(), device=0)
eng.prepare(0.0)
n_tiles = int(eng.info().num_tiles)
fn = C.CDLL(os.path.join(ROOT, "gbrs_amd", "libgbrs_hip.so")).gbrs_debug_set_tile_stamps
fn.argtypes = [C.c_void_p]
buf = torch.zeros((n_tiles + 64) * 8, dtype=torch.int64, device="cuda:0")
eng.step(300)
torch.cuda.synchronize()
fn(C.c_void_p(buf.data_ptr()))
eng.step(1)
torch.cuda.synchronize()
fn(C.c_void_p(0))
a = buf.cpu().numpy().reshape(-1, 8)[:n_tiles].astype(np.int64)
base = a[:, 0][a[:, 0] > 0].min()
names = ["loop", "flush (hand in sums)", "closing barrier", "epilogue", "next theta -> LDS", "opening barrier"]
has_next = a[:, 6] > 0
print(f"tiles {n_tiles}, with a successor in their workgroup {int(has_next.sum())}; launch span "
      f"{(a[:, 4].max() - base) / 100:.1f} us (first loop start -> last epilogue)")
for k, nm in enumerate(names):
    sel = has_next if k >= 4 else np.ones(n_tiles, bool)
    d = (a[sel, k + 1] - a[sel, k]) / 100.0
    print(f"  {nm:24s} mean {d.mean():6.2f} us   p50 {np.median(d):6.2f}   p95 {np.percentile(d, 95):6.2f}")
tot = (a[has_next, 6] - a[has_next, 0]) / 100.0
print(f"  tile total (loop start -> next loop start) mean {tot.mean():.2f} us; outside the loop {np.mean((a[has_next, 6] - a[has_next, 1]) / 100.0):.2f} us")

dc, nb = a[:, 7] >> 32, a[:, 7] & 0xffffffff
print("by dictionary size (entries):  tiles  batches  loop  flush  barrierA  epilogue  theta  barrierB   (mean us)")
for lo, hi in ((1, 2), (2, 4), (4, 8), (8, 16), (16, 64), (64, 128), (128, 256), (256, 385)):
    sel = (dc >= lo) & (dc < hi)
    if not sel.any():
        continue
    seg = [np.mean((a[sel & (has_next if k >= 4 else True), k + 1] - a[sel & (has_next if k >= 4 else True), k]) / 100.0) if (sel & (has_next if k >= 4 else True)).any() else float("nan") for k in range(6)]
    print(f"  [{lo:3d},{hi:3d})  {int(sel.sum()):6d}  {nb[sel].mean():7.1f}  " + "  ".join(f"{x:6.2f}" for x in seg))
