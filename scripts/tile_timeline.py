#!/usr/bin/env python3
"""Where does an E-step launch spend its time?  Needs a library built with GBRS_HIPCC_EXTRA=-DGBRS_DIAG_TILE_TIMES:
every tile records start / loop start / loop end / end (s_memrealtime, 10 ns) and the CU it ran on.  Prints the launch
span, the occupancy of the chip's workgroup places over time, the tail, and prologue / loop / epilogue shares.
Usage: python3 scripts/tile_timeline.py [rows]"""
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
lib = _lib.load()
fn = getattr(lib._lib if hasattr(lib, "_lib") else lib, "gbrs_debug_set_tile_stamps", None)
if fn is None:
    fn = C.CDLL(os.path.join(ROOT, "gbrs_amd", "libgbrs_hip.so")).gbrs_debug_set_tile_stamps
fn.argtypes = [C.c_void_p]
buf = torch.zeros((n_tiles + 64) * 8, dtype=torch.int64, device="cuda:0")
eng.step(300)                                  # settle the clock
torch.cuda.synchronize()
fn(C.c_void_p(buf.data_ptr()))
eng.step(1)
torch.cuda.synchronize()
fn(C.c_void_p(0))
a = buf.cpu().numpy().reshape(-1, 8)[:n_tiles].astype(np.int64)
t0, t1, t2, t3, hw, xcc5, t6, t7 = (a[:, k] for k in (0, 1, 2, 3, 4, 5, 6, 7))
xcc, nb = xcc5 & 0xff, xcc5 >> 8
base = t0.min()
t0, t1, t2, t3, t6, t7 = (t - base for t in (t0, t1, t2, t3, t6, t7))
span = t3.max()
print(f"tiles {n_tiles}  launch span {span * 10 / 1000:.1f} us (first start -> last end)")
print(f"prologue in parts (us, thread 0): start -> sums zeroed {np.mean(t7 - t0) / 100:.2f}, -> theta gathered {np.mean(t6 - t7) / 100:.2f}, "
      f"-> barrier passed {np.mean(t1 - t6) / 100:.2f}")
print(f"per tile (us): prologue {np.mean(t1 - t0) / 100:.2f}  loop {np.mean(t2 - t1) / 100:.2f}  epilogue {np.mean(t3 - t2) / 100:.2f}  "
      f"total {np.mean(t3 - t0) / 100:.2f}   largest tile {np.max(t3 - t0) / 100:.2f}, batches {nb.max()}")
cu = ((hw >> 8) & 0xf) | (((hw >> 13) & 0x7) << 4) | ((xcc & 0xf) << 8)       # CU_ID, SE_ID, XCC
print(f"distinct CUs seen: {len(np.unique(cu))}; tiles per CU: min {np.bincount(np.unique(cu, return_inverse=True)[1]).min()} "
      f"max {np.bincount(np.unique(cu, return_inverse=True)[1]).max()}")
# occupancy over time: workgroups in flight per 1-us bucket
edges = np.arange(0, span + 100, 100)
occ = np.zeros(len(edges))
for s, e in zip(t0, t3):
    occ[s // 100:(e // 100) + 1] += 1
in_loop = np.zeros(len(edges))
for s, e in zip(t1, t2):
    in_loop[s // 100:(e // 100) + 1] += 1
print("time_us  workgroups_in_flight  in_batch_loop")
for k in range(0, len(edges), max(1, len(edges) // 30)):
    print(f"{k:7d}  {int(occ[k]):6d}  {int(in_loop[k]):6d}")
places = occ.max()
print(f"peak workgroups in flight {int(places)}; mean over the span {occ[:span // 100 + 1].mean():.0f} = {occ[:span // 100 + 1].mean() / places:.2f} of the peak;"
      f" mean in the batch loop {in_loop[:span // 100 + 1].mean():.0f}")
work = float(np.sum(t3 - t0))
print(f"sum of tile times / (peak places x span) = {work / (places * span):.3f}")
last_start = t0.max()
print(f"last tile starts at {last_start / 100:.1f} us; after that {np.sum(t3 > last_start)} tiles still run; "
      f"time from the moment fewer than 90 % of the places are busy to the end: "
      f"{(span - np.argmax(occ[::-1] >= 0.9 * places) * 100 if False else (len(occ) - 1 - np.argmax(occ[::-1] >= 0.9 * places)) ) } us mark")
per_cu_busy = {}
order = np.argsort(t0)
print("start order == tile order for the first 768:", bool(np.all(np.sort(order[:768]) == np.arange(768))))
