#!/usr/bin/env python3
"""bench.py - EM iterations/s (Model 4) and HMM genes/s on MI355X, one JSON line on rank 0.

    python bench.py --gpus 1 --steps 50 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

A "step" is one EM iteration (E-step + M-step + convergence bookkeeping) over one synthetic
DO-shaped sample of BASELINE.json configs[1]: 40M reads x 8 haplotypes x 120k isoforms
(SURVEY.md §8d generator), resident in HBM before the timed region.  With N > 1 every rank holds
its own 40M-read shard of a pooled sample (reads sharded, weak scaling) and the ranks exchange
the H*L expected-count vector with one RCCL all-reduce per iteration; the value
reported is shard-iterations/s summed over ranks.  The N > 1 line also carries `replicas` (configs[3] as
SURVEY §8e reads it: one independent sample per GPU, samples/s, no collective), `time_to_solution` through the
same multi-rank driver, and `hmm` (every rank its own samples, aggregate genes/s, no collective).

The same line carries `roofline` (E-step kernel, HIP-event time measured in the library on its
own stream), `cpu_baseline` (the numpy oracle, 1 core, bounded sample) and an `hmm` object with
the reconstruct numbers for BASELINE.json configs[2].
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy)
F64_VECTOR_PEAK_TFLOPS = 78.6   # MI355X_MICROARCH.md: FP64 vector peak


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    # a run of a few milliseconds ends before the chip has settled on its clock: 50 steps after 5 read 5-10 % lower
    # than 500 after 50 (profiles/r02_estep_experiments.txt, item 8); the defaults measure the settled loop (65 ms)
    ap.add_argument("--steps", type=int, default=500)
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--preroll-ms", type=float, default=40.0,
                    help="untimed EM steps for this long in front of the W warmup steps, so that the K timed steps run at the clock the "
                         "chip settles on under this load (it needs ~25 ms, profiles/r03_short_run_clock.txt); the K steps after W "
                         "warmup steps on the chip as the setup left it are timed first and reported as `cold_start`; 0 = only those")
    ap.add_argument("--rows", type=int, default=40_000_000, help="reads per GPU")
    ap.add_argument("--haps", type=int, default=8)
    ap.add_argument("--loci", type=int, default=120_000)
    ap.add_argument("--merge", action="store_true", help="merge identical rows in the device layout")
    ap.add_argument("--flags", type=int, default=0, help="extra gbrs_em_create flags (tuning)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-hmm", action="store_true")
    ap.add_argument("--no-merged-line", action="store_true")
    ap.add_argument("--no-check", action="store_true", help="skip the post-run tiles-vs-CSC theta comparison")
    ap.add_argument("--no-solve", action="store_true", help="skip the time-to-solution measurements (tol = 1e-4)")
    ap.add_argument("--no-replicas", action="store_true", help="N > 1: skip the one-sample-per-GPU replica measurement (configs[3])")
    ap.add_argument("--from-host", action="store_true", help="also time gbrs_em_create from host (numpy) arrays: PCIe copy + layout build")
    ap.add_argument("--cpu-rows", type=int, default=2_000_000, help="rows of the one-core baseline when it cannot run at full size")
    ap.add_argument("--cpu-iters", type=int, default=60, help="oracle iterations timed for cpu_baseline (stops at 25 s)")
    ap.add_argument("--cpu-full", default="auto", choices=["auto", "yes", "no"],
                    help="one-core baselines on the whole 40M-read sample (auto: when the host has >= 128 GB of memory; "
                         "SURVEY 8d allows the row subsample only on smaller hosts)")
    ap.add_argument("--variant", default="survey", choices=["survey", "multi_isoform"],
                    help="read generator of the timed workload: SURVEY 8d's (the headline) or the multi-isoform one "
                         "(gbrs_amd/synth_torch.py); the default run reports the second as `multi_isoform_variant`")
    ap.add_argument("--no-multi-isoform-line", action="store_true")
    ap.add_argument("--hmm-samples", type=int, default=1)
    ap.add_argument("--hmm-batch", type=int, default=64, help="second HMM measurement with this many samples in one launch (0 = skip)")
    ap.add_argument("--hmm-batch-large", type=int, default=256,
                    help="third HMM measurement at a batch that fills the chip (the alpha / backward sweeps run 16 samples per wavefront on MFMA there; 0 = skip)")
    ap.add_argument("--hmm-reps", type=int, default=7)
    ap.add_argument("--hmm-haps", type=int, default=8, help="founder haplotypes of the HMM measurement (16 = config 5's 136 states)")
    ap.add_argument("--no-overlap", action="store_true", help="N > 1: one engine per GPU, all-reduce not overlapped with the E-step")
    ap.add_argument("--force-overlap-path", action="store_true", help="N = 1: run the two-engine form anyway (its compute-side cost without any exchange)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (gloo only to rehearse N>1 on one GPU)")
    ap.add_argument("--rccl-selftest", action="store_true",
                    help="N = 1: create a one-rank RCCL process group and send every all-reduce of the multi-GPU code "
                         "paths through it (the collectives, stream ordering and raw-pointer tensor views on hardware)")
    ap.add_argument("--no-e2e", action="store_true", help="skip the end-to-end `gbrs quantify` + `gbrs reconstruct` wall-clock measurement")
    ap.add_argument("--e2e-format", default="h5", choices=["h5", "npz", "both"])
    return ap.parse_args()


class DevArray:
    """__cuda_array_interface__ view of a raw device pointer, so torch can wrap the library's
    partial-sum buffer for the RCCL all-reduce."""
    def __init__(self, ptr, n):
        self.__cuda_array_interface__ = dict(shape=(n,), typestr="<f8", data=(ptr, False), version=2)


def timed_region(args, run_steps, barrier, world, torch, dist, dev):
    """W untimed warmup steps, then exactly K steps between two barriers (+ device synchronisation), max over ranks - twice:
    first on the chip as the setup left it (`cold`: the form of rounds 1-3), then again after --preroll-ms of the same steps,
    untimed.  Returns (dt of the second region, dict describing the first and the pre-roll)."""
    def once():
        run_steps(args.warmup)
        barrier()
        t0 = time.perf_counter()
        run_steps(args.steps)
        barrier()
        dt = time.perf_counter() - t0
        if world > 1:
            tt = torch.tensor([dt], dtype=torch.float64, device=dev)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            dt = float(tt.item())
        return dt
    dt = once()
    if args.preroll_ms <= 0:
        return dt, None
    cold = dict(ms_per_step=dt / args.steps * 1e3, value=world * args.steps / dt, unit="iters/s",
                note=f"the first {args.steps} steps after {args.warmup} warmup steps on the chip as the setup left it (what "
                     "rounds 1-3 reported as the headline)")
    n_pre = int(min(20000, max(1, round(args.preroll_ms * 1e-3 / (dt / args.steps)))))     # the same count on every rank
    t0 = time.perf_counter()
    run_steps(n_pre)
    barrier()
    pre_ms = (time.perf_counter() - t0) * 1e3
    dt = once()
    return dt, dict(cold_start=cold, preroll=dict(steps=n_pre, ms=pre_ms, asked_ms=args.preroll_ms,
                                                  note="untimed EM steps in front of the warmup steps of the reported region"))


def em_bench(args, rank, world, torch, dist, keep_host=False):
    from gbrs_amd import _lib, synth, synth_torch
    from gbrs_amd.engine import EmEngine
    dev = f"cuda:{torch.cuda.current_device()}"
    t0 = time.perf_counter()
    prob = synth_torch.make_em_problem_device(args.rows, args.haps, args.loci, synth.SEED_BASE_EM + 1, dev,
                                              row_seed=synth.SEED_BASE_EM + 1 + rank, variant=args.variant)
    t_gen = time.perf_counter() - t0
    host = None
    if keep_host:                 # the same reads for the one-core oracle (cpu_baseline at full size)
        import numpy as np
        host = dict(indptr=[t.cpu().numpy().view(np.uint32) for t in prob["indptr"]],
                    indices=[t.cpu().numpy().view(np.uint32) for t in prob["indices"]],
                    eff_len=prob["eff_len"].cpu().numpy())
    t0 = time.perf_counter()
    eng = EmEngine.from_device(
        prob["R"], prob["L"], prob["H"], [t.data_ptr() for t in prob["indptr"]],
        [t.data_ptr() for t in prob["indices"]], None, prob["eff_len"].data_ptr(),
        device=torch.cuda.current_device(),
        flags=(_lib.GBRS_EM_MERGE_IDENTICAL_ROWS if args.merge else 0) | args.flags)
    t_create = time.perf_counter() - t0
    n_entries = prob["N"]
    t_host = None
    if args.from_host and world == 1:
        import numpy as np
        ip = [t.cpu().numpy().view(np.uint32) for t in prob["indptr"]]
        ix = [t.cpu().numpy().view(np.uint32) for t in prob["indices"]]
        ef = prob["eff_len"].cpu().numpy()
        t1 = time.perf_counter()
        e2 = EmEngine.from_host(prob["R"], prob["L"], prob["H"], ip, ix, None, ef,
                                device=torch.cuda.current_device())
        t_host = time.perf_counter() - t1
        e2.close()
        del ip, ix

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    acc_t = None
    use_dist = world > 1 or args.rccl_selftest
    if use_dist:
        # run the library on torch's current stream: E-step -> all-reduce -> M-step are then ordered
        # on the device and the loop below never synchronises with the host
        eng.set_stream(torch.cuda.current_stream().cuda_stream)
        p, n = eng.prepare_partial()
        acc_t = torch.as_tensor(DevArray(p, n), device=dev)
        dist.all_reduce(acc_t)
        eng.finish_prepare(0.0)
    else:
        eng.prepare(0.0)

    def run_steps(k):
        if not use_dist:
            eng.step(k)
            return
        for _ in range(k):
            eng.estep_partial()
            dist.all_reduce(acc_t)
            eng.finish_step(want_err=False)
        eng.sync()

    dt, region = timed_region(args, run_steps, barrier, world, torch, dist, dev)
    inf = eng.info()
    res = dict(dt=dt, region=region, t_gen=t_gen, t_create=t_create, t_create_host=t_host, N=n_entries, info=inf, host=host,
               mean_loci_per_read=prob.get("mean_loci_per_read"))
    # ---- outside the timed region: is the state the timed steps produced a valid EM state? ----------
    res["check"] = em_state_check(torch, [eng.expected_counts()], float(args.rows) * world)
    if world == 1 and not args.no_check:
        # the same sample through the plain CSC kernels (two passes, global atomics: a different code
        # path with a different summation order) must give the same theta after 3 iterations
        import numpy as np
        eng.prepare(0.0)
        eng.step(3)
        th = eng.theta()
        ec = EmEngine.from_device(
            prob["R"], prob["L"], prob["H"], [t.data_ptr() for t in prob["indptr"]],
            [t.data_ptr() for t in prob["indices"]], None, prob["eff_len"].data_ptr(),
            device=torch.cuda.current_device(), flags=_lib.GBRS_EM_LAYOUT_CSC)
        ec.prepare(0.0)
        ec.step(3)
        th_c = ec.theta()
        ec.close()
        denom = np.maximum(np.abs(th_c), 1e-300)
        rel = float(np.max(np.abs(th - th_c) / denom))
        res["check"]["theta_vs_csc_layout_3_iters_max_rel"] = rel
        res["check"]["ok"] = bool(res["check"]["ok"] and rel < 1e-9)
        # put the engine back where the timed steps left off is not needed: what follows re-prepares
    del prob
    torch.cuda.empty_cache()
    if world == 1 and not args.merge:
        # time to solution with the reference's default stopping rule (tol = 1e-4 TPM units)
        eng.prepare(0.0)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        n_it, hist = eng.run(model=4, tol=1e-4, max_iters=999)
        res["solve"] = dict(iterations=n_it, ms=(time.perf_counter() - t1) * 1e3,
                            final_err_sum=float(hist[-1]) if n_it else None)
    if use_dist and world > 1 and not args.no_solve:
        # N > 1 on one engine per GPU: gbrs_amd.dist.ShardedEM with the all-reduce in line on the engine's stream
        from gbrs_amd.dist import ShardedEM

        _views = {}

        def _ar(ptr, n):
            if ptr not in _views:
                _views[ptr] = torch.as_tensor(DevArray(ptr, n), device=dev)
            dist.all_reduce(_views[ptr])
        drv = ShardedEM(eng, _ar)
        drv.prepare(0.0)
        barrier()
        t1 = time.perf_counter()
        n_it = drv.run(model=4, tol=1e-4, max_iters=999)
        barrier()
        ms = (time.perf_counter() - t1) * 1e3
        tt = torch.tensor([ms], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        res["solve"] = dict(iterations=n_it, ms=float(tt.item()), final_err_sum=drv.err_history[-1] if n_it else None,
                            path="ShardedEM.run (one engine per rank, err_sum read back every iteration)",
                            rows_in_the_pool=args.rows * world)
    if world == 1:
        res["estep_ms"] = inf.last_estep_ms
        res["step_ms"] = inf.last_step_ms
    else:
        # per-launch E-step time: one un-pipelined timed step group without the collective
        eng.step(min(args.steps, 10))
        inf = eng.info()
        res["estep_ms"] = inf.last_estep_ms
        res["step_ms"] = inf.last_step_ms
    eng.close()
    return res


def balanced_gene_boundary_device(prob, torch):
    """The gene start closest to the locus that halves this rank's entry count (0 if there is none)."""
    L = prob["L"]
    cum = torch.zeros(L + 1, dtype=torch.int64, device=prob["indptr"][0].device)
    for ip in prob["indptr"]:
        cum += ip.to(torch.int64)
    l_half = int(torch.searchsorted(cum, cum[-1:] // 2)[0])
    gs = prob["gene_starts"]
    k = int(gs.searchsorted(l_half))
    cands = [int(gs[i]) for i in (k - 1, k) if 0 <= i < len(gs) and 0 < int(gs[i]) < L]
    return min(cands, key=lambda l: abs(l - l_half)) if cands else 0


def split_problem_device(prob, l_split, torch):
    """Cut the sample's loci at l_split into two problems of l_split and L - l_split loci (device
    tensors; columns re-based, original row ids).  Returns (half_a, half_b), each (indptr, indices,
    eff_len, num_loci), or None when some row has entries on both sides or a side is empty."""
    R = prob["R"]
    dev = prob["indptr"][0].device
    a_ip, a_ix, b_ip, b_ix = [], [], [], []
    in_a = torch.zeros(R, dtype=torch.bool, device=dev)
    in_b = torch.zeros(R, dtype=torch.bool, device=dev)
    for ip, ix in zip(prob["indptr"], prob["indices"]):
        cut = int(ip[l_split])
        a_ip.append(ip[:l_split + 1].contiguous())
        b_ip.append((ip[l_split:] - cut).contiguous())
        a_ix.append(ix[:cut])
        b_ix.append(ix[cut:])
        in_a[a_ix[-1].long()] = True
        in_b[b_ix[-1].long()] = True
    if bool((in_a & in_b).any()) or not (bool(in_a.any()) and bool(in_b.any())):
        return None
    el = prob["eff_len"]
    return ((a_ip, a_ix, el[:, :l_split].contiguous(), l_split),
            (b_ip, b_ix, el[:, l_split:].contiguous(), prob["L"] - l_split))


def pipelined_setup(args, rank, world, torch, dist):
    """Fallible part of the two-engine path: generate the shard, cut it at a gene boundary, build the two
    engines.  Touches no data-path collective (only the cut broadcast, which every rank reaches before
    anything can fail locally); any local failure is reported through the returned (state, error) pair and
    agreed on by the caller with one MIN all-reduce before the first data collective."""
    from gbrs_amd import synth, synth_torch
    from gbrs_amd.engine import EmEngine
    devno = torch.cuda.current_device()
    dev = f"cuda:{devno}"
    t0 = time.perf_counter()
    prob = synth_torch.make_em_problem_device(args.rows, args.haps, args.loci, synth.SEED_BASE_EM + 1, dev,
                                              row_seed=synth.SEED_BASE_EM + 1 + rank)
    t_gen = time.perf_counter() - t0
    # one cut for all ranks (the all-reduce is over a locus range): rank 0's balanced gene boundary
    ls = torch.tensor([balanced_gene_boundary_device(prob, torch)], dtype=torch.int64, device=dev)
    if world > 1:
        dist.broadcast(ls, src=0)
    l_split = int(ls.item())
    state, err = None, None
    try:
        cut = split_problem_device(prob, l_split, torch) if 0 < l_split < prob["L"] else None
        if cut is None:
            err = "no gene boundary that no row straddles"
        else:
            half_a, half_b = cut
            t0 = time.perf_counter()
            engs = [EmEngine.from_device(prob["R"], nl, prob["H"], [t.data_ptr() for t in ip],
                                         [t.data_ptr() for t in ix], None, el.data_ptr(), device=devno,
                                         flags=args.flags | 128)      # GBRS_EM_SIDE_BY_SIDE: tiles sized for the two ranges together
                    for ip, ix, el, nl in (half_a, half_b)]
            state = dict(engs=engs, t_gen=t_gen, t_create=time.perf_counter() - t0, N=prob["N"], l_split=l_split)
    except Exception as ex:                           # noqa: BLE001 - reported, then agreed on by all ranks
        err = f"{type(ex).__name__}: {ex}"
    del prob
    torch.cuda.empty_cache()
    return state, err


def em_bench_pipelined(args, rank, world, torch, dist, state):
    """N > 1: every rank holds one shard of rows, cut into two locus ranges with an engine each, so that
    the RCCL all-reduce of one range overlaps the E-step of the other (gbrs_amd.dist.PipelinedShardedEM).
    Runs after all ranks agreed that their setup succeeded: a failure from here on is fatal."""
    from gbrs_amd.dist import PipelinedShardedEM
    dev = f"cuda:{torch.cuda.current_device()}"
    engs = state["engs"]
    # Every range runs on a stream of its own: E-step -> all-reduce -> M-step are ordered on that stream and nothing
    # orders the two ranges against each other, so the device overlaps the collective of one with the E-step of
    # the other by itself.  The collective is issued in line (no handle, no wait) on the one default group:
    # its communicator serialises the two ranges' all-reduces in issue order, the same on every rank, and
    # that serialisation is also what staggers the ranges (measured with a one-rank RCCL group,
    # scripts/pipelined_host_cost.py: 114 us per iteration against 184 us for async handles on one stream,
    # whose event hops between the compute and the collective stream cost ~10 us each).
    torch.cuda.synchronize()                          # the engines were built on the default stream
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    for e, st in zip(engs, streams):
        e.set_stream(st.cuda_stream)
    views, turn = {}, {}

    class _Done:                                   # the collective is already in the range's stream
        def wait(self):
            pass

    class _OnStream:                               # an engine that says whose buffer the next all-reduce is
        def __init__(self, eng, st):
            self._eng, self._st = eng, st

        def __getattr__(self, name):
            return getattr(self._eng, name)

        def estep_partial(self):
            turn["stream"] = self._st
            return self._eng.estep_partial()

        def prepare_partial(self):
            turn["stream"] = self._st
            return self._eng.prepare_partial()

    def start_allreduce(ptr, n):
        if world == 1 and not args.rccl_selftest:
            return _Done()                         # --force-overlap-path alone: nothing to exchange
        if ptr not in views:
            views[ptr] = torch.as_tensor(DevArray(ptr, n), device=dev)
        with torch.cuda.stream(turn["stream"]):
            dist.all_reduce(views[ptr])
        return _Done()

    drv = PipelinedShardedEM(_OnStream(engs[0], streams[0]), _OnStream(engs[1], streams[1]), start_allreduce)
    drv.prepare(0.0)

    def barrier():
        for e in engs:
            e.sync()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    dt, region = timed_region(args, drv.step, barrier, world, torch, dist, dev)
    check = em_state_check(torch, [e.expected_counts() for e in engs], float(args.rows) * world)
    # time to solution through the same two-engine loop: prepare, then the reference's stopping rule (EMfactory.py:266-278)
    # over both locus ranges on the device (gbrs_em_pair_check), looked at every 8 iterations; max over ranks
    solve = None
    if not args.no_solve:
        drv.prepare(0.0)
        barrier()
        t1 = time.perf_counter()
        n_it = drv.run(model=4, tol=1e-4, max_iters=999)
        barrier()
        ms = (time.perf_counter() - t1) * 1e3
        if world > 1:
            tt = torch.tensor([ms], dtype=torch.float64, device=dev)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            ms = float(tt.item())
        solve = dict(iterations=n_it, ms=ms, final_err_sum=drv.err_history[-1] if drv.err_history else None,
                     path="PipelinedShardedEM.run (two engines per rank, pair stopping rule on the device)",
                     rows_in_the_pool=args.rows * world)
    # per-launch E-step time of the two halves: local steps without the collective
    estep_ms = step_ms = 0.0
    infos = []
    for e in engs:
        e.step(min(args.steps, 10))
        inf = e.info()
        estep_ms += inf.last_estep_ms
        step_ms += inf.last_step_ms
        infos.append(inf)
    for e in engs:
        e.close()
    return dict(dt=dt, region=region, t_gen=state["t_gen"], t_create=state["t_create"], t_create_host=None, N=state["N"],
                info=infos[0], infos=infos, estep_ms=estep_ms, step_ms=step_ms, l_split=state["l_split"],
                check=check, **({"solve": solve} if solve else {}))


def em_state_check(torch, counts_list, expect_total):
    """Outside the timed region: the expected read counts of the last E-step must be finite, non-negative
    and add up to the number of reads (every read's posterior sums to 1, EMfactory.py:302 /
    AlignmentPropertyMatrix.py:335-342)."""
    import numpy as np
    tot = float(sum(np.asarray(c, dtype=np.float64).sum() for c in counts_list))
    ok = all(bool(np.isfinite(c).all()) and bool((np.asarray(c) >= 0).all()) for c in counts_list)
    out = dict(sum_expected_counts=tot, finite_nonnegative=ok)
    if expect_total is not None:
        out["expected"] = float(expect_total)
        out["rel_err"] = abs(tot - expect_total) / max(expect_total, 1.0)
        out["ok"] = bool(ok and out["rel_err"] < 1e-9)
    return out


def max_over_ranks(x, torch, dist, world, dev):
    if world == 1:
        return float(x)
    t = torch.tensor([float(x)], dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def gather_over_ranks(x, torch, dist, world, dev):
    if world == 1:
        return [float(x)]
    t = torch.zeros(world, dtype=torch.float64, device=dev)
    t[dist.get_rank()] = float(x)
    dist.all_reduce(t)
    return [float(v) for v in t.tolist()]


def replica_bench(args, rank, world, torch, dist):
    """BASELINE.json configs[3] read as SURVEY 8(e) reads it: N samples, one per GPU, every rank solves ITS OWN sample
    (independent theta: replicas only, no collective on the data path) - layout build from the CSC arrays in HBM,
    prepare, then EMfactory.run's loop with the reference's default stopping rule (tol = 1e-4).  Reported as samples/s:
    N samples / the slowest rank's time, the region bracketed by barriers."""
    from gbrs_amd import synth, synth_torch
    from gbrs_amd.engine import EmEngine
    dev = f"cuda:{torch.cuda.current_device()}"
    # a sample of its own per rank: its own abundances, gene sizes and reads (seed offset 1000 * (rank + 1))
    seed = synth.SEED_BASE_EM + 1 + 1000 * (rank + 1)
    prob = synth_torch.make_em_problem_device(args.rows, args.haps, args.loci, seed, dev, row_seed=seed,
                                              variant=args.variant)

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()

    def one_sample():
        t0 = time.perf_counter()
        eng = EmEngine.from_device(prob["R"], prob["L"], prob["H"], [t.data_ptr() for t in prob["indptr"]],
                                   [t.data_ptr() for t in prob["indices"]], None, prob["eff_len"].data_ptr(),
                                   device=torch.cuda.current_device(), flags=args.flags)
        t1 = time.perf_counter()
        eng.prepare(0.0)
        n_it, hist = eng.run(model=4, tol=1e-4, max_iters=999)
        t2 = time.perf_counter()
        tot = float(eng.expected_counts().sum())
        eng.close()
        return t1 - t0, t2 - t1, n_it, tot

    one_sample()                                   # untimed: code objects, allocator pools, clocks
    barrier()
    t0 = time.perf_counter()
    t_create, t_solve, n_it, tot = one_sample()
    barrier()
    dt = max_over_ranks(time.perf_counter() - t0, torch, dist, world, dev)
    solve_max = max_over_ranks(t_solve, torch, dist, world, dev)
    create_max = max_over_ranks(t_create, torch, dist, world, dev)
    iters = gather_over_ranks(n_it, torch, dist, world, dev)
    mass = gather_over_ranks(abs(tot - args.rows) / args.rows, torch, dist, world, dev)
    del prob
    torch.cuda.empty_cache()
    return dict(metric="samples/s: one sample per GPU, layout build from arrays in HBM + prepare + run(tol=1e-4); no collective",
                value=world / dt, unit="samples/s", n_gpus=world, seconds=dt, scaling="weak",
                em_only=dict(value=world / solve_max, unit="samples/s", seconds=solve_max,
                             note="prepare + EMfactory.run loop alone (the slowest rank), layout already built"),
                layout_build_seconds=create_max, iterations_per_rank=[int(v) for v in iters],
                mass_conservation_rel_err_per_rank=mass,
                workload=f"R={args.rows} reads x H={args.haps} x L={args.loci} isoforms per sample, a different sample "
                         f"(abundances, gene sizes, reads) on every rank")


def hmm_bench_ranks(args, rank, world, torch, dist):
    """N > 1: the HMM is independent per (sample, chromosome) (gbrs_utils.py:498-599) - replicas only, no collective.
    Every rank runs its own samples through the same pass as the N = 1 line (one sample: the CLI's shape; a batch that
    fills the chip); genes/s = all ranks' gene x sample units / the slowest rank's pass time."""
    dev = f"cuda:{torch.cuda.current_device()}"
    out = None
    for key, ns in (("single", 1), ("batched_large", args.hmm_batch_large)):
        if ns <= 0 or (key == "batched_large" and args.hmm_haps != 8):
            continue
        if world > 1:
            dist.barrier()
        b = hmm_bench(args, torch, ns=ns, with_cpu=False, sample_seed=1 + rank)
        ms = max_over_ranks(b["ms_per_pass"], torch, dist, world, dev)
        wall = max_over_ranks(b["wall_clock"]["ms_per_pass"], torch, dist, world, dev)
        units = b["genes"] * ns * world
        rec = dict(value=units / (ms * 1e-3), unit="genes/s", n_gpus=world, samples_per_gpu=ns, ms_per_pass_slowest_rank=ms,
                   wall_clock=dict(value=units / (wall * 1e-3), unit="genes/s", ms_per_pass_slowest_rank=wall),
                   per_gpu_roofline_frac=b["roofline"]["frac"] * b["ms_per_pass"] / ms, genes=b["genes"], states=b["states"])
        if key == "single":
            out = dict(metric=b["metric"] + "; one process per GPU, no collective", scaling="weak", **rec)
        else:
            out["batched_large"] = rec
    return out


def host_memory_gb():
    try:
        return os.sysconf("SC_PAGE_SIZE") * os.sysconf("SC_PHYS_PAGES") / 2.0**30
    except (ValueError, OSError):
        return 0.0


def cpu_full_size(args):
    return args.cpu_full == "yes" or (args.cpu_full == "auto" and host_memory_gb() >= 128.0)


def em_cpu_baseline(args, host=None):
    """The numpy oracle (op-for-op restatement of the reference) on one core: on the whole workload - the reads the
    timed region processed, copied back from HBM - when the host has the memory, else on a row subsample."""
    import numpy as np
    from gbrs_amd import synth
    from oracle.em_oracle import EMOracle
    if host is not None:
        R, L, H = args.rows, args.loci, args.haps
        t0 = time.perf_counter()
        o = EMOracle(R, L, H, host["indptr"], host["indices"], None)
        t_build = time.perf_counter() - t0
        t0 = time.perf_counter()
        o.prepare(0.0, host["eff_len"])
        t_prep = time.perf_counter() - t0
        nnz = int(sum(len(i) for i in host["indices"]))
        o.em_step()
        t0 = time.perf_counter()
        n = 0
        while n < 3 or (n < args.cpu_iters and time.perf_counter() - t0 < 12.0):
            o.em_step()
            n += 1
        dt = (time.perf_counter() - t0) / n
        return dict(value=1.0 / dt, unit="iters/s", cores=1, kind="port", full_size=True,
                    prepare_s=t_prep, ms_per_iter=dt * 1e3,
                    sample=f"numpy oracle on the whole workload: the R={R} reads (N={nnz} entries) of the timed region "
                           f"copied back from HBM, prepare {t_prep:.1f} s, then {n} iterations after one untimed, "
                           f"{dt * 1e3:.0f} ms/iter; nothing scaled; host has {os.cpu_count()} cores and "
                           f"{host_memory_gb():.0f} GB, reference is single-threaded")
    R = min(args.cpu_rows, args.rows)
    inc = synth.make_em_problem(R=R, H=args.haps, L=args.loci, seed=synth.SEED_BASE_EM + 1)
    o = EMOracle(inc.num_rows, inc.num_loci, inc.num_haps, inc.indptr, inc.indices, None)
    o.prepare(0.0, inc.effective_length(100))
    o.em_step()
    t0 = time.perf_counter()
    n = 0
    while n < args.cpu_iters and time.perf_counter() - t0 < 25.0:
        o.em_step()
        n += 1
    dt = (time.perf_counter() - t0) / max(n, 1)
    scale = args.rows / R
    return dict(value=1.0 / (dt * scale), unit="iters/s", cores=1, kind="port", full_size=False,
                sample=f"numpy oracle, {n} iterations at R={R} rows (N={inc.nnz} entries), "
                       f"{dt * 1e3:.1f} ms/iter measured, scaled x{scale:g} linearly in rows to the "
                       f"{args.rows}-row workload (host memory {host_memory_gb():.0f} GB < 128 GB: SURVEY 8d's subsample "
                       f"rule); host has {os.cpu_count()} cores, reference is single-threaded")


def hmm_bench(args, torch, ns=None, with_cpu=True, sample_seed=1):
    import numpy as np
    from gbrs_amd import synth
    from gbrs_amd.hmm import DiplotypeHMM
    HH = args.hmm_haps
    prob = synth.make_hmm_problem(H=HH)
    chroms = prob.chroms
    ns = args.hmm_samples if ns is None else ns
    hmm = DiplotypeHMM(HH, chroms, [len(prob.gene_ids[c]) for c in chroms], [prob.tprob[c] for c in chroms],
                       device=torch.cuda.current_device())
    rng = np.random.default_rng(sample_seed)
    ex, av, ha = [], [], []
    for c in chroms:
        ids = prob.gene_ids[c]
        e = np.array([prob.expr[g] for g in ids])
        if sample_seed != 1:                       # another rank's sample: its own expression values
            e = rng.gamma(1.0, 5.0, size=e.shape) * (rng.random(e.shape) < 0.5)
        if ns > 1:
            e = np.stack([e] + [rng.gamma(1.0, 5.0, size=e.shape) * (rng.random(e.shape) < 0.5)
                                for _ in range(ns - 1)])
        ex.append(e)
        ha.append(np.array([g in prob.avecs for g in ids], dtype=np.uint8))
        av.append(np.array([prob.avecs.get(g, np.zeros((HH, HH))) for g in ids]))
    hmm.set_expression(ex, av, ha, 1.5, 0.12)       # uploads the specificity tables once; they stay resident
    for _ in range(3 if ns == 1 else 1):            # untimed passes (a single sample's pass is 1.5 ms: let the clock settle)
        hmm.run()
    emis, tot, wall = [], [], []
    for _ in range(args.hmm_reps):
        t0 = time.perf_counter()
        hmm.set_expression(ex, expr_threshold=1.5, sigma=0.12)   # H2D copy of the expression rows + emission kernel
        hmm.run()                                   # blocking: returns when the three chains and the posterior are done
        wall.append((time.perf_counter() - t0) * 1e3)
        inf = hmm.info()
        emis.append(inf.last_emission_ms)
        tot.append(inf.last_emission_ms + inf.last_run_ms)   # device time; the run's chains overlap
    inf = hmm.info()
    rank = dict(blocks_fixed_up=int(inf.last_delta_blocks), longest_fixup_genes=int(inf.last_delta_longest_fixup),
                fallbacks=int(inf.last_delta_fallbacks),
                note="blocked scan (1-4 samples): Viterbi values by rank convergence - blocks whose values were matched to the "
                     "block before them, the longest such fix-up, (sample, chromosome) pairs recomputed by the sequential chain")
    ms = float(np.median(tot))
    wall_ms = float(np.median(wall))
    t0 = time.perf_counter()
    for ci in range(len(chroms)):                   # what `gbrs reconstruct` keeps: posteriors, path, calls (sample 0)
        hmm.get(ci, 0)
    fetch_ms = (time.perf_counter() - t0) * 1e3
    units = prob.num_genes * ns
    S_ = HH * (HH + 1) // 2
    # 16 S^2 bytes of transition tables per gene (shared by the samples of a batch) + 64 S per gene x sample
    algo = prob.num_genes * (16 * S_ * S_ + 64 * S_ * ns)
    out = dict(metric="HMM gene x sample /s (emission+forward+backward+posterior+Viterbi)",
               value=units / (ms * 1e-3), unit="genes/s", ms_per_pass=ms, n_samples=ns,
               wall_clock=dict(ms_per_pass=wall_ms, value=units / (wall_ms * 1e-3), unit="genes/s",
                               fetch_one_sample_ms=fetch_ms,
                               note="host wall clock of set_expression (expression rows over PCIe + emission kernel) "
                                    "+ run, median of the timed passes; the transition and specificity tables are "
                                    "resident on the handle; `value` above is device-event time of the same passes"),
               genes=prob.num_genes, states=HH * (HH + 1) // 2,
               **({"delta_rank_convergence": rank} if rank["blocks_fixed_up"] or rank["fallbacks"] else {}),
               kernels_ms=dict(emission=inf.last_emission_ms, forward_viterbi=inf.last_forward_ms,
                               backward_posterior=inf.last_backward_ms, backtrace=inf.last_backtrace_ms,
                               run=inf.last_run_ms,
                               note="forward, backward and Viterbi chains run concurrently: run < sum"),
               roofline=dict(bound="hbm", achieved=algo / (ms * 1e-3) / 1e9,
                             peak=HBM_PEAK_GBS, unit="GB/s",
                             frac=algo / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, traffic=None,
                             algorithmic_bytes=algo,
                             note=("one sample = 3 chains x 20 chromosomes = 60 wavefronts on a 1,024-SIMD chip: " if ns == 1 else "")
                                  + "sequential recursion over genes: latency-bound, not bandwidth-bound; the "
                                  "transition tables are priced once per batch (SURVEY 8d), the per-sample "
                                  "vectors once per sample"))
    hmm.close()
    try:                                              # measured HBM traffic of this exact workload, from the committed counter passes
        with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as fh:
            pt = json.load(fh).get(f"hmm_S{S_}_G{prob.num_genes}_N{ns}")
        if pt:
            out["roofline"].update(traffic=pt["bytes_per_pass"], traffic_source=pt["source"],
                                   traffic_rate_GBs=pt["bytes_per_pass"] / (ms * 1e-3) / 1e9,
                                   traffic_frac_of_peak=pt["bytes_per_pass"] / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                   measured_in_run={"achieved": True, "traffic": False})
    except (OSError, ValueError):
        pass
    if with_cpu and args.hmm_batch > 0 and args.hmm_batch != ns:
        b = hmm_bench(args, torch, ns=args.hmm_batch, with_cpu=False)
        out["batched"] = {k: b[k] for k in ("value", "unit", "ms_per_pass", "n_samples", "wall_clock", "kernels_ms", "roofline")}
    if with_cpu and args.hmm_batch_large > 0 and args.hmm_batch_large not in (ns, args.hmm_batch) and HH == 8:
        b = hmm_bench(args, torch, ns=args.hmm_batch_large, with_cpu=False)
        out["batched_large"] = {k: b[k] for k in ("value", "unit", "ms_per_pass", "n_samples", "wall_clock", "kernels_ms", "roofline")}
        out["batched_large"]["note"] = "alpha and backward sweeps: 16 samples per wavefront on v_mfma_f64_16x16x4 (from 96 samples on); delta chain: vector max-plus"
    if with_cpu and not args.no_cpu_baseline:
        from oracle import hmm_oracle
        # the whole single-sample workload when it has 8 founders (about 10 s of oracle time), else a slice
        sub = prob if HH == 8 else synth.make_hmm_problem(H=8, genes_per_chrom=[1500, 1500], seed=synth.SEED_HMM)
        t0 = time.perf_counter()
        hmm_oracle.reconstruct_arrays(sub.hap_names, sub.chroms, sub.gene_ids, sub.tprob, sub.expr, sub.avecs)
        dt = time.perf_counter() - t0
        out["cpu_baseline"] = dict(value=sub.num_genes / dt, unit="genes/s", cores=1, kind="port",
                                   sample=f"numpy oracle on {len(sub.chroms)} chromosomes, {sub.num_genes} genes, "
                                          f"1 sample ({dt:.1f} s); host has {os.cpu_count()} cores, reference is "
                                          f"single-threaded")
    return out


def self_launch(args):
    """`python bench.py --gpus N` with N > 1 and no torchrun environment: start N fresh ranks as child
    processes (torch.distributed.run) BEFORE this process imports torch or touches a GPU, relay their
    output and exit with their status.  Nothing is ever exec'ed from a process that initialised HIP."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    print("[bench] launching " + " ".join(cmd), file=sys.stderr, flush=True)
    proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, text=True, env=env)
    lines = []
    for ln in proc.stdout:
        sys.stdout.write(ln)
        sys.stdout.flush()
        lines.append(ln)
    rc = proc.wait()
    if rc != 0:
        raise SystemExit(rc)
    recs = []
    for ln in lines:
        ln = ln.strip()
        if ln.startswith("{"):
            try:
                recs.append(json.loads(ln))
            except ValueError:
                pass
    if len(recs) != 1 or recs[0].get("n_gpus") != args.gpus:
        print(f"[bench] expected one JSON line with n_gpus={args.gpus}, got {[r.get('n_gpus') for r in recs]}",
              file=sys.stderr, flush=True)
        raise SystemExit(3)
    if args.backend == "nccl" and recs[0].get("rccl_ranks") != args.gpus:
        # a run that fell back to fewer RCCL ranks than GPUs asked for is not a scaling measurement
        print(f"[bench] --gpus {args.gpus} over RCCL but the line says rccl_ranks={recs[0].get('rccl_ranks')}",
              file=sys.stderr, flush=True)
        raise SystemExit(4)
    raise SystemExit(0)


def main():
    args = parse()
    if os.environ.get("GBRS_BENCH_WATCHDOG"):        # rehearsals: where is every rank after N seconds without finishing?
        import faulthandler
        faulthandler.dump_traceback_later(int(os.environ["GBRS_BENCH_WATCHDOG"]), repeat=True, file=sys.stderr)
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        self_launch(args)                             # never returns
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks")
    e2e = None
    if world == 1 and not args.no_e2e and not args.merge and args.rows == 40_000_000 and args.haps == 8 \
            and args.variant == "survey":
        # End-to-end `gbrs quantify` + `gbrs reconstruct` (file in -> reports out) as fresh child processes,
        # measured BEFORE this process opens the GPU: a second process holding a device context slows the
        # large allocations of the measured one.  Outside the timed EM region; its own object in the line.
        # (the library is built in a child as well: loading libgbrs_hip.so here, before torch brings in its own
        # HIP runtime, leaves this process with two runtimes and no visible device)
        import subprocess
        subprocess.run([sys.executable, "-c", "import __graft_entry__ as g; g.build()"], cwd=ROOT, check=True,
                       stdout=sys.stderr)
        sys.path.insert(0, os.path.join(ROOT, "scripts"))
        import e2e_bench
        try:
            e2e = e2e_bench.measure(args.rows, args.haps, args.loci, args.e2e_format, args.cpu_rows, repeats=3,
                                    with_cpu=not args.no_cpu_baseline, cpu_full=args.cpu_full)
        except Exception as ex:                         # noqa: BLE001 - reported in the line, never hidden
            e2e = {"error": f"{type(ex).__name__}: {ex}"}
    import torch
    import torch.distributed as dist
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: gbrs_amd has no CPU path")
    ndev = torch.cuda.device_count()
    if world > 1 and args.backend == "nccl" and ndev < world:
        raise SystemExit(f"bench.py: --gpus {world} over RCCL needs {world} GPUs, {ndev} visible "
                         "(--backend gloo rehearses the multi-rank path on fewer)")
    local = local % ndev
    torch.cuda.set_device(local)
    if world > 1:
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device(f"cuda:{local}"))
        else:
            dist.init_process_group(args.backend)
    elif args.rccl_selftest:
        import socket
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        dist.init_process_group(args.backend, init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1,
                                **({"device_id": torch.device(f"cuda:{local}")} if args.backend == "nccl" else {}))
    import __graft_entry__
    if rank == 0:
        import contextlib
        with contextlib.redirect_stdout(sys.stderr):      # stdout carries the one JSON line and nothing else
            __graft_entry__.build()
    if world > 1:
        dist.barrier()

    em = None
    path = "single-engine"
    path_note = None
    if (world > 1 or args.force_overlap_path) and not args.no_overlap and not args.merge:
        state, err = pipelined_setup(args, rank, world, torch, dist)
        ok = torch.tensor([1 if state is not None else 0], device=f"cuda:{local}")
        if world > 1:                                  # go / no-go agreed before any data-path collective
            dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        if int(ok.item()) == 1:
            path = "pipelined"
            em = em_bench_pipelined(args, rank, world, torch, dist, state)   # a failure here is fatal
        else:
            if state is not None:
                for e in state["engs"]:
                    e.close()
            del state
            torch.cuda.empty_cache()
            path_note = f"two-engine setup declined on some rank (this rank: {err}); one engine per GPU"
            print(f"[bench] rank {rank}: {path_note}", file=sys.stderr, flush=True)
    want_full_cpu = rank == 0 and world == 1 and not args.no_cpu_baseline and cpu_full_size(args)
    if em is None:
        em = em_bench(args, rank, world, torch, dist, keep_host=want_full_cpu)
    inf = em["info"]
    infos = em.get("infos", [inf])
    ms_per_step = em["dt"] / args.steps * 1e3
    value = world * args.steps / em["dt"]
    algo = sum(int(i.algorithmic_bytes) for i in infos)
    moved = sum(int(i.bytes_per_iter) for i in infos)
    estep_bytes = sum(int(i.estep_bytes) for i in infos)
    words = sum(int(i.num_device_words) for i in infos)
    estep_s = em["estep_ms"] * 1e-3
    # price on min(B_iter, bytes actually moved) so a smaller device format never inflates the fraction
    priced_estep = min(algo, estep_bytes)
    priced_step = min(algo, moved)
    achieved = priced_estep / estep_s / 1e9 if estep_s > 0 else None
    useful_flop = 4.0 * em["N"]                         # per stored entry: one FMA into den, one into A
    def short(n):
        return f"{n / 1e6:g}M" if n >= 1_000_000 and n % 100_000 == 0 else (f"{n / 1e3:g}k" if n >= 1000 and n % 100 == 0 else str(n))
    is_c2 = (args.rows, args.haps, args.loci) == (40_000_000, 8, 120_000)
    is_c5_shard = (args.rows, args.haps, args.loci) == (25_000_000, 16, 200_000)
    cfg_name = "configs[1]: single DO sample" if is_c2 and world == 1 else \
        ("configs[3]-shaped pool: one shard per GPU" if is_c2 else
         ("configs[4]: one GPU's shard of the 200M-read x 16-haplotype sample" if is_c5_shard else "custom shape"))
    line = {
        "metric": f"EM iterations/s (EMASE Model 4, {short(args.rows)} reads x {args.haps} haplotypes x "
                  f"{short(args.loci)} isoforms per GPU)",
        "value": value, "unit": "iters/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": ms_per_step, **(em.get("region") or {}),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f64", "data": "synthetic",
        "path": path, "rccl_ranks": world if ((world > 1 or args.rccl_selftest) and args.backend == "nccl") else 0,
        "backend": args.backend if world > 1 else None,
        "config": {"workload": ("" if args.variant == "survey" else f"[{args.variant} reads, not the headline] ") +
                               f"{cfg_name}, R={args.rows} reads x H={args.haps} x "
                               f"L={args.loci} isoforms, N={em['N']} alignment entries, quantify Model 4, "
                               f"tol=0 fixed iterations" + (f", rows sharded: one {short(args.rows)}-read shard of a "
                               f"{short(args.rows * world)}-read pool per GPU + {'RCCL' if args.backend == 'nccl' else args.backend} "
                               "all-reduce of the H*L vector per iteration" if world > 1 else "")
                               + (f"; loci cut at gene boundary {em['l_split']} into two engines per GPU, the "
                                  "all-reduce of one range overlapped with the E-step of the other"
                                  if "l_split" in em else ""),
                   "layout": int(inf.layout), "merge_identical_rows": bool(args.merge),
                   "device_rows": sum(int(i.num_device_rows) for i in infos),
                   "device_words": words,
                   "words_per_read": words / max(sum(int(i.num_rows) for i in infos) / len(infos), 1),
                   "entries_per_read": em["N"] / args.rows,
                   "tiles": sum(int(i.num_tiles) for i in infos), "slots": sum(int(i.num_slots) for i in infos),
                   "heavy_loci": sum(int(i.num_heavy_loci) for i in infos),
                   "light_loci": sum(int(i.num_light_loci) for i in infos),
                   "long_rows": sum(int(i.num_long_rows) for i in infos),
                   "locus_sets": sum(int(i.num_locus_sets) for i in infos)},
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS if achieved else None,
                     "traffic": None, "kernel": "tile_estep_kernel (E-step)", "kernel_ms": em["estep_ms"],
                     "limited_by": "vector issue + LDS (valu_util, lds_idx_active below); the per-launch working set "
                                   "fits the 256 MB Infinity Cache, so the HBM fraction is context, not the bound",
                     "kernel_bytes": estep_bytes, "priced_bytes": priced_estep,
                     "algorithmic_bytes": algo,
                     "note": "achieved = E-step bytes (word stream, tile headers, dictionary, theta gather, slot "
                             "stores) / E-step launch time from HIP events on the library's stream; the device "
                             "format is a re-encoding of the CSC input (one 32-bit word per (read, locus) pair "
                             "with the haplotype mask inside; a read whose alignments share one mask is one word on its locus "
                             "SET, config.locus_sets), so it moves fewer bytes than SURVEY 8d's B_iter "
                             "(algorithmic_bytes); words_per_read depends on the generator (<= 2 loci per read)",
                     "whole_step": {"bytes": moved, "priced_bytes": priced_step, "ms": ms_per_step,
                                    "achieved": priced_step / (ms_per_step * 1e-3) / 1e9,
                                    "frac": priced_step / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                    "vs_survey_B_iter": {"achieved": algo / (ms_per_step * 1e-3) / 1e9,
                                                         "frac": algo / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS}},
                     "compute": {"useful_f64_flop_per_launch": useful_flop,
                                 "achieved_tflops": useful_flop / estep_s / 1e12 if estep_s > 0 else None,
                                 "peak_tflops": F64_VECTOR_PEAK_TFLOPS,
                                 "frac": useful_flop / estep_s / 1e12 / F64_VECTOR_PEAK_TFLOPS if estep_s > 0 else None,
                                 "note": "2 FMAs per stored alignment entry (row sum and column sum); the kernel "
                                         "is bound by vector issue, not by HBM (see valu_util)"}},
        "setup_s": {"generate": em["t_gen"], "create_layout": em["t_create"],
                    "create_from_host_arrays": em["t_create_host"]},
        "state_check": em.get("check"),
    }
    if path_note:
        line["path_note"] = path_note
    if "solve" in em:
        line["time_to_solution"] = dict(em["solve"], rule="err_sum <= 1e6*tol, tol=1e-4 (gbrs quantify default)")
    # Counter figures of the E-step kernel from the committed rocprofv3 PMC passes (separate runs of this
    # same command, scripts/profile_estep.sh): HBM traffic = FETCH_SIZE x2 (gfx950 correction,
    # MI355X_MICROARCH.md) + WRITE_SIZE per launch; VALU issue utilisation = SQ_INSTS_VALU x 4 cycles /
    # (1024 SIMDs x launch duration x measured shader clock).  Only quoted for the exact workload measured.
    try:
        with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as fh:
            pt = json.load(fh)
        key = f"R{args.rows}_H{args.haps}_L{args.loci}_merge{int(args.merge)}"
        if key in pt and path == "single-engine":
            line["roofline"]["traffic"] = pt[key]["bytes_per_launch"]
            line["roofline"]["traffic_source"] = pt[key]["source"]
            # traffic and the counter-derived fields below come from the committed rocprofv3 --pmc passes of this same
            # command (a counter run cannot share a process with the timed region), not from this run
            line["roofline"]["measured_in_run"] = {"achieved": True, "kernel_ms": True, "traffic": False, "pmc_fields": False}
            for k in ("valu_insts_per_launch", "valu_util", "shader_clock_mhz", "lds_bank_conflict_cycles",
                      "lds_idx_active_cycles", "lds_util", "valu_insts_per_word", "pmc_round"):
                if k in pt[key]:
                    line["roofline"][k] = pt[key][k]
    except (OSError, ValueError):
        pass
    if world > 1:
        # the rest of BASELINE.json's metric at N > 1 (every rank takes part: the timings are max-over-ranks)
        if not args.no_replicas and not args.merge:
            line["replicas"] = replica_bench(args, rank, world, torch, dist)
        if not args.no_hmm:
            line["hmm"] = hmm_bench_ranks(args, rank, world, torch, dist)
    if rank == 0:
        if not args.no_cpu_baseline and world == 1:
            line["cpu_baseline"] = em_cpu_baseline(args, em.pop("host", None))
            line["speedup_vs_cpu"] = value / line["cpu_baseline"]["value"]
        elif world > 1:
            line["cpu_baseline"] = None        # rank 0 at N = 1 only (the driver's N = 1 run carries it)
        if not args.merge and not args.no_merged_line and world == 1:
            import copy
            a2 = copy.copy(args)
            a2.merge = True
            a2.no_check = True
            m = em_bench(a2, rank, world, torch, dist)
            line["merged_rows_variant"] = {
                "note": "same sample with identical reads merged into weighted rows while building the device "
                        "layout (GBRS_EM_MERGE_IDENTICAL_ROWS, what `gbrs compress` does first); not the headline",
                "value": args.steps / m["dt"], "unit": "iters/s", "ms_per_step": m["dt"] / args.steps * 1e3,
                **(m.get("region") or {}),
                "estep_kernel_ms": m["estep_ms"], "device_rows": int(m["info"].num_device_rows),
                "device_words": int(m["info"].num_device_words), "layout_bytes": int(m["info"].bytes_per_iter),
                "state_check": m.get("check")}
        if not args.merge and not args.no_multi_isoform_line and world == 1 and args.variant == "survey":
            import copy
            a3 = copy.copy(args)
            a3.variant = "multi_isoform"
            v = em_bench(a3, rank, world, torch, dist)
            vi = v["info"]
            v_estep_s = v["estep_ms"] * 1e-3
            v_priced = min(int(vi.algorithmic_bytes), int(vi.estep_bytes))
            line["multi_isoform_variant"] = {
                "note": "second EM workload, not the headline: a read aligns to 1 + Poisson(2) isoforms of its gene "
                        "(capped at the gene's size) and its haplotype mask differs from locus to locus (each "
                        "(locus, haplotype) alignment dropped with p = 0.1): gbrs_amd/synth_torch.py "
                        "make_multi_isoform_device; same sample model, R, H, L as the headline",
                "value": args.steps / v["dt"], "unit": "iters/s", "ms_per_step": v["dt"] / args.steps * 1e3,
                **(v.get("region") or {}),
                "entries": v["N"], "entries_per_read": v["N"] / args.rows,
                "loci_per_read": v["mean_loci_per_read"],
                "words_per_read": int(vi.num_device_words) / max(int(vi.num_rows), 1),
                "device_words": int(vi.num_device_words), "tiles": int(vi.num_tiles), "slots": int(vi.num_slots),
                "long_rows": int(vi.num_long_rows),
                "estep_kernel_ms": v["estep_ms"], "estep_bytes": int(vi.estep_bytes),
                "algorithmic_bytes": int(vi.algorithmic_bytes),
                "roofline": {"bound": "hbm", "achieved": v_priced / v_estep_s / 1e9 if v_estep_s > 0 else None,
                             "peak": HBM_PEAK_GBS, "unit": "GB/s",
                             "frac": v_priced / v_estep_s / 1e9 / HBM_PEAK_GBS if v_estep_s > 0 else None,
                             "priced_bytes": v_priced, "traffic": None},
                "time_to_solution": v.get("solve"), "state_check": v.get("check")}
        if not args.no_hmm and world == 1:
            line["hmm"] = hmm_bench(args, torch)
        if e2e is not None:
            line["end_to_end"] = e2e
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
    if world > 1 or args.rccl_selftest:
        dist.destroy_process_group()
    chk = em.get("check") or {}
    if chk.get("ok") is False:
        raise SystemExit("bench.py: the post-run state check failed: " + json.dumps(chk))


if __name__ == "__main__":
    main()
