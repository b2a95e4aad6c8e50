"""Multi-GPU EM: rows (reads / ECs) sharded across ranks, one process per GPU.

Each rank builds its own device layout from its row block, every iteration runs the E-step over
its rows into the (L x H) partial-sum vector, the ranks exchange that vector with ONE all-reduce
(RCCL over xGMI when the backend is "nccl"; "gloo" in the CPU tests) and then every rank applies
the identical M-step and stopping rule redundantly.  There is no other
collective on the data path.

The engine is injected: the product passes gbrs_amd.engine.EmEngine (HIP); the CPU tests pass a
numpy stand-in with the same five methods to check that sharding + all-reduce reproduce the
unsharded result without a GPU.
"""
from __future__ import annotations

import numpy as np


def shard_rows(indptr, indices, count, num_rows, rank, world):
    """Row block [r0, r1) of rank `rank`, balanced by entry count, as CSC arrays with local row
    ids.  Returns (r0, r1, indptr_local, indices_local, count_local)."""
    H = len(indptr)
    L = len(indptr[0]) - 1
    per_row = np.zeros(num_rows, dtype=np.int64)
    for h in range(H):
        per_row += np.bincount(indices[h], minlength=num_rows)
    cum = np.concatenate(([0], np.cumsum(per_row)))
    total = cum[-1]
    bounds = [int(np.searchsorted(cum, total * k / world, side='left')) for k in range(world + 1)]
    bounds[0], bounds[-1] = 0, num_rows
    r0, r1 = bounds[rank], bounds[rank + 1]
    ip, ix = [], []
    for h in range(H):
        col = np.repeat(np.arange(L, dtype=np.int64), np.diff(indptr[h].astype(np.int64)))
        keep = (indices[h] >= r0) & (indices[h] < r1)
        ix.append((indices[h][keep].astype(np.int64) - r0).astype(np.uint32))
        ip.append(np.searchsorted(col[keep], np.arange(L + 1)).astype(np.uint32))
    cnt = None if count is None else np.ascontiguousarray(count[r0:r1])
    return r0, r1, ip, ix, cnt


class ShardedEM:
    """Drives one engine per rank.  `allreduce(ptr_or_array, n)` sums the partial vector in place
    across ranks; `engine` exposes prepare_partial / finish_prepare / estep_partial / finish_step
    (+ theta / expected_counts), see gbrs_amd.engine.EmEngine."""

    def __init__(self, engine, allreduce):
        self.engine = engine
        self.allreduce = allreduce
        self.num_iters = 0
        self.err_history = []

    def prepare(self, pseudocount=0.0):
        p, n = self.engine.prepare_partial()
        self.allreduce(p, n)
        self.engine.finish_prepare(pseudocount)

    def step(self):
        p, n = self.engine.estep_partial()
        self.allreduce(p, n)
        return self.engine.finish_step(want_err=True)

    def run(self, model=4, tol=0.001, max_iters=999):
        if model != 4:
            raise RuntimeError('The read normalization model should be 1, 2, 3, or 4.' if model not in (1, 2, 3)
                               else f'Multiread model {model} is not implemented by the MI355X path')
        self.num_iters = 0
        self.err_history = []
        err_sum = 1000000.0
        target = 1000000.0 * tol
        while err_sum > target and self.num_iters < max_iters:
            err_sum = self.step()          # identical on every rank: same reduced vector, same kernels
            self.num_iters += 1
            self.err_history.append(err_sum)
        return self.num_iters


def torch_allreduce(dist, torch, device):
    """all-reduce closure over torch.distributed for a raw device pointer (HIP engine).  Call
    engine.set_stream(torch.cuda.current_stream().cuda_stream) first to drop the two
    synchronisations below."""
    cache = {}

    class _Dev:
        def __init__(self, ptr, n):
            self.__cuda_array_interface__ = dict(shape=(n,), typestr='<f8', data=(ptr, False), version=2)

    def allreduce(ptr, n):
        key = (ptr, n)
        if key not in cache:
            cache[key] = torch.as_tensor(_Dev(ptr, n), device=device)
        torch.cuda.synchronize(device)
        dist.all_reduce(cache[key])
        torch.cuda.synchronize(device)
    return allreduce
