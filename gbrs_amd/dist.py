"""Multi-GPU EM: rows (reads / ECs) sharded across ranks, one process per GPU.

Each rank builds its own device layout from its row block, every iteration runs the E-step over
its rows into the (L x H) partial-sum vector, the ranks exchange that vector with ONE all-reduce
(RCCL over xGMI when the backend is "nccl"; "gloo" in the CPU tests) and then every rank applies
the identical M-step and stopping rule redundantly.  There is no other
collective on the data path.

PipelinedShardedEM hides that all-reduce behind compute: the loci are cut into two ranges that no row
straddles (a gene boundary), each range gets its own engine on the rank, and the all-reduce of one
range's vector runs while the E-step of the other range is on the GPU.

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


# ---- overlap of the all-reduce with the E-step --------------------------------------------------

def split_at_locus(indptr, indices, l_split):
    """The column ranges [0, l_split) and [l_split, L) as two CSC problems of l_split and L - l_split
    loci (columns re-based to 0, original row ids).  Returns ((indptr_a, indices_a), (indptr_b, indices_b))."""
    a_ip, a_ix, b_ip, b_ix = [], [], [], []
    for ip, ix in zip(indptr, indices):
        ip = np.asarray(ip)
        cut = int(ip[l_split])
        a_ip.append(np.ascontiguousarray(ip[:l_split + 1], dtype=np.uint32))
        a_ix.append(np.ascontiguousarray(ix[:cut], dtype=np.uint32))
        b_ip.append((ip[l_split:].astype(np.int64) - cut).astype(np.uint32))
        b_ix.append(np.ascontiguousarray(ix[cut:], dtype=np.uint32))
    return (a_ip, a_ix), (b_ip, b_ix)


def rows_are_disjoint(indices_a, indices_b, num_rows):
    """True when no row has entries in both column ranges (the condition for PipelinedShardedEM)."""
    in_a = np.zeros(num_rows, dtype=bool)
    in_b = np.zeros(num_rows, dtype=bool)
    for ix in indices_a:
        in_a[np.asarray(ix, dtype=np.int64)] = True
    for ix in indices_b:
        in_b[np.asarray(ix, dtype=np.int64)] = True
    return not bool((in_a & in_b).any())


def balanced_gene_boundary(indptr, gene_starts):
    """The gene start closest to the locus that halves the entry count."""
    cum = np.zeros(len(indptr[0]), dtype=np.int64)
    for ip in indptr:
        cum += np.asarray(ip, dtype=np.int64)
    l_half = int(np.searchsorted(cum, cum[-1] / 2))
    gs = np.asarray(gene_starts, dtype=np.int64)
    k = int(np.searchsorted(gs, l_half))
    cands = [int(gs[i]) for i in (k - 1, k) if 0 <= i < len(gs)]
    return min(cands, key=lambda l: abs(l - l_half)) if cands else 0


class PipelinedShardedEM:
    """Two engines per rank over the locus ranges [0, l_split) and [l_split, L) (split_at_locus) that no
    row straddles, so the two ranges are independent EM problems that only share the convergence test.
    Per iteration and range: E-step over the rank's rows -> all-reduce of that engine's partial vector
    -> M-step; the ranges are interleaved so that the all-reduce of one is in flight (RCCL's own
    stream) while the E-step of the other occupies the GPU:

        E_a  AR_a | E_b  AR_b | wait AR_a  M_a  E_a'  AR_a' | wait AR_b  M_b  E_b'  AR_b' | ...

    `start_allreduce(buffer, n)` starts the in-place sum of the engine's partial buffer across ranks
    and returns an object whose wait() orders the engine's stream after it.  (bench.py gives every engine a
    HIP stream of its own and issues the collective in line on that stream - wait() is then a no-op and the
    two ranges overlap on the device without any handle: 114 us instead of 184 us per iteration with a
    one-rank RCCL group, scripts/pipelined_host_cost.py.)  pseudocount must be 0
    (its renormalisation couples the ranges).

    run() applies the reference's stopping rule (EMfactory.py:266-278) over BOTH ranges after every iteration, on
    the device (gbrs_em_pair_check: the two engines of a rank share one stop flag), and looks at the result every
    `check_every` iterations; iterations enqueued past the stopping one are no-ops, so the iteration count, the err_sum
    sequence and theta are those of ShardedEM and of the reference."""

    def __init__(self, engine_a, engine_b, start_allreduce):
        self.eng = (engine_a, engine_b)
        self.start = start_allreduce
        self.num_iters = 0
        self.err_history = []

    def prepare(self, pseudocount=0.0):
        if pseudocount != 0.0:
            raise RuntimeError('PipelinedShardedEM needs pseudocount 0 (the pseudocount renormalisation couples the ranges)')
        pend = [self.start(*e.prepare_partial()) for e in self.eng]
        for e, w in zip(self.eng, pend):
            w.wait()
            e.finish_prepare(0.0)

    def step(self, k=1):
        """k EM iterations (no error report)."""
        if k <= 0:
            return
        # A run() that met its stopping rule left the engines stopped (every later step a no-op, so that theta stays the
        # stopping iteration's).  A step asked for by hand is applied anyway, as EMfactory.update_allelic_expression is
        # (EMfactory.py:214-232 knows no stopping rule): HIP engines clear their device flag in gbrs_em_estep_partial,
        # engines that carry the flag on the host (tests/cpu_engine.py) get it cleared here.
        for e in self.eng:
            if getattr(e, 'stopped', False) is True:
                e.stopped = False
        pend = [self.start(*e.estep_partial()) for e in self.eng]
        for _ in range(k - 1):
            for i, e in enumerate(self.eng):
                pend[i].wait()
                e.finish_step(want_err=False)
                pend[i] = self.start(*e.estep_partial())
        for e, w in zip(self.eng, pend):
            w.wait()
            e.finish_step(want_err=False)
        self.num_iters += k

    def theta(self):
        """(H x L): the two ranges side by side."""
        parts = [e.theta() if callable(e.theta) else e.theta for e in self.eng]
        return np.concatenate([np.asarray(parts[0]), np.asarray(parts[1])], axis=1)

    def _pair(self):
        """The shared stopping rule of the two engines: on the device for HIP engines (EmEngine.pair_*), on the host
        for engines that keep their per-locus totals in numpy (the CPU tests' stand-in)."""
        a, b = self.eng
        return _DevicePair(a, b) if hasattr(a, 'pair_check') else _HostPair(a, b)

    def run(self, model=4, tol=0.001, max_iters=999, check_every=8):
        if model != 4:
            raise RuntimeError('The read normalization model should be 1, 2, 3, or 4.' if model not in (1, 2, 3)
                               else f'Multiread model {model} is not implemented by the MI355X path')
        self.num_iters = 0
        self.err_history = []
        if max_iters <= 0 or not 1000000.0 > 1000000.0 * tol:
            return 0
        pair = self._pair()
        pair.begin(max_iters)
        enqueued = 0
        while enqueued < max_iters:
            k = min(check_every, max_iters - enqueued)
            # k iterations with the rule evaluated after each: E-steps and all-reduces interleave as in step()
            pend = [self.start(*e.estep_partial()) for e in self.eng]
            for it in range(k):
                for i, e in enumerate(self.eng):
                    pend[i].wait()
                    e.finish_step(want_err=False)
                    if i == 1:
                        pair.check(tol)            # after both M-steps of the iteration, before b's next E-step
                    if it + 1 < k:
                        pend[i] = self.start(*e.estep_partial())
            enqueued += k
            done, stopped, hist = pair.status(max_iters)
            self.num_iters, self.err_history = done, [float(x) for x in hist]
            if stopped:
                break
        return self.num_iters


class _DevicePair:
    def __init__(self, a, b):
        self.a, self.b = a, b

    def begin(self, max_iters):
        self.a.pair_begin(self.b, max_iters)

    def check(self, tol):
        self.a.pair_check(self.b, tol)

    def status(self, cap):
        return self.a.pair_status(self.b, cap)


class _HostPair:
    """The same rule for engines that expose `last_totals` = (per-locus totals before, after) of their last M-step and
    honour a `stopped` attribute (tests/cpu_engine.py)."""

    def __init__(self, a, b):
        self.a, self.b = a, b
        self.hist, self.stopped = [], False

    def begin(self, max_iters):
        self.hist, self.stopped = [], False
        self.a.stopped = self.b.stopped = False

    def check(self, tol):
        if self.stopped:
            return
        prev = np.concatenate([self.a.last_totals[0], self.b.last_totals[0]])
        cur = np.concatenate([self.a.last_totals[1], self.b.last_totals[1]])
        err = float(np.abs(cur * (1e6 / cur.sum()) - prev * (1e6 / prev.sum())).sum())
        self.hist.append(err)
        if not err > 1000000.0 * tol:
            self.stopped = self.a.stopped = self.b.stopped = True

    def status(self, cap):
        return len(self.hist), self.stopped, np.asarray(self.hist[:cap])
