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
reported is shard-iterations/s summed over ranks.

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


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--rows", type=int, default=40_000_000, help="reads per GPU")
    ap.add_argument("--haps", type=int, default=8)
    ap.add_argument("--loci", type=int, default=120_000)
    ap.add_argument("--merge", action="store_true", help="merge identical rows in the device layout")
    ap.add_argument("--flags", type=int, default=0, help="extra gbrs_em_create flags (tuning)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-hmm", action="store_true")
    ap.add_argument("--no-merged-line", action="store_true")
    ap.add_argument("--from-host", action="store_true", help="also time gbrs_em_create from host (numpy) arrays: PCIe copy + layout build")
    ap.add_argument("--cpu-rows", type=int, default=2_000_000)
    ap.add_argument("--cpu-iters", type=int, default=60, help="oracle iterations timed for cpu_baseline (stops at 25 s)")
    ap.add_argument("--hmm-samples", type=int, default=1)
    ap.add_argument("--hmm-batch", type=int, default=64, help="second HMM measurement with this many samples in one launch (0 = skip)")
    ap.add_argument("--hmm-reps", type=int, default=5)
    ap.add_argument("--hmm-haps", type=int, default=8, help="founder haplotypes of the HMM measurement (16 = config 5's 136 states)")
    ap.add_argument("--no-overlap", action="store_true", help="N > 1: one engine per GPU, all-reduce not overlapped with the E-step")
    ap.add_argument("--force-overlap-path", action="store_true", help="N = 1: run the two-engine form anyway (its compute-side cost without any exchange)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (gloo only to rehearse N>1 on one GPU)")
    return ap.parse_args()


class DevArray:
    """__cuda_array_interface__ view of a raw device pointer, so torch can wrap the library's
    partial-sum buffer for the RCCL all-reduce."""
    def __init__(self, ptr, n):
        self.__cuda_array_interface__ = dict(shape=(n,), typestr="<f8", data=(ptr, False), version=2)


def em_bench(args, rank, world, torch, dist):
    from gbrs_amd import _lib, synth, synth_torch
    from gbrs_amd.engine import EmEngine
    dev = f"cuda:{torch.cuda.current_device()}"
    t0 = time.perf_counter()
    prob = synth_torch.make_em_problem_device(args.rows, args.haps, args.loci, synth.SEED_BASE_EM + 1, dev,
                                              row_seed=synth.SEED_BASE_EM + 1 + rank)
    t_gen = time.perf_counter() - t0
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
    del prob
    torch.cuda.empty_cache()

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    acc_t = None
    if world > 1:
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
        if world == 1:
            eng.step(k)
            return
        for _ in range(k):
            eng.estep_partial()
            dist.all_reduce(acc_t)
            eng.finish_step(want_err=False)
        eng.sync()

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
    inf = eng.info()
    res = dict(dt=dt, t_gen=t_gen, t_create=t_create, t_create_host=t_host, N=n_entries, info=inf)
    if world == 1 and not args.merge:
        # time to solution with the reference's default stopping rule (tol = 1e-4 TPM units)
        eng.prepare(0.0)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        n_it, hist = eng.run(model=4, tol=1e-4, max_iters=999)
        res["solve"] = dict(iterations=n_it, ms=(time.perf_counter() - t1) * 1e3,
                            final_err_sum=float(hist[-1]) if n_it else None)
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


def em_bench_pipelined(args, rank, world, torch, dist):
    """N > 1: every rank holds one shard of rows, cut into two locus ranges with an engine each, so that
    the RCCL all-reduce of one range overlaps the E-step of the other (gbrs_amd.dist.PipelinedShardedEM).
    Returns None when the sample cannot be cut (the caller falls back to one engine per rank)."""
    from gbrs_amd import synth, synth_torch
    from gbrs_amd.dist import PipelinedShardedEM
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
    cut = split_problem_device(prob, l_split, torch) if 0 < l_split < prob["L"] else None
    ok = torch.tensor([1 if cut is not None else 0], device=dev)
    if world > 1:
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)          # every rank takes the same path
    if int(ok.item()) == 0:
        return None
    half_a, half_b = cut
    t0 = time.perf_counter()
    engs = [EmEngine.from_device(prob["R"], nl, prob["H"], [t.data_ptr() for t in ip],
                                 [t.data_ptr() for t in ix], None, el.data_ptr(), device=devno, flags=args.flags)
            for ip, ix, el, nl in (half_a, half_b)]
    t_create = time.perf_counter() - t0
    n_entries, H = prob["N"], prob["H"]
    L = prob["L"]
    del prob, half_a, half_b, cut
    torch.cuda.empty_cache()
    stream = torch.cuda.current_stream().cuda_stream
    for e in engs:
        e.set_stream(stream)
    views = {}

    class _Done:                                   # world == 1 (--force-overlap-path): nothing to exchange
        def wait(self):
            pass

    def start_allreduce(ptr, n):
        if world == 1:
            return _Done()
        if ptr not in views:
            views[ptr] = torch.as_tensor(DevArray(ptr, n), device=dev)
        return dist.all_reduce(views[ptr], async_op=True)

    drv = PipelinedShardedEM(engs[0], engs[1], start_allreduce)
    drv.prepare(0.0)
    def barrier():
        for e in engs:
            e.sync()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    drv.step(args.warmup)
    barrier()
    t0 = time.perf_counter()
    drv.step(args.steps)
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
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
    return dict(dt=dt, t_gen=t_gen, t_create=t_create, t_create_host=None, N=n_entries, info=infos[0], infos=infos,
                estep_ms=estep_ms, step_ms=step_ms, l_split=l_split)


def em_cpu_baseline(args):
    """The numpy oracle (op-for-op restatement of the reference) on one core, bounded sample."""
    import numpy as np
    from gbrs_amd import synth
    from oracle.em_oracle import EMOracle
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
    return dict(value=1.0 / (dt * scale), unit="iters/s", cores=1, kind="port",
                sample=f"numpy oracle, {n} iterations at R={R} rows (N={inc.nnz} entries), "
                       f"{dt * 1e3:.1f} ms/iter measured, scaled x{scale:g} linearly in rows to the "
                       f"{args.rows}-row workload; host has {os.cpu_count()} cores, reference is single-threaded")


def hmm_bench(args, torch, ns=None, with_cpu=True):
    import numpy as np
    from gbrs_amd import synth
    from gbrs_amd.hmm import DiplotypeHMM
    HH = args.hmm_haps
    prob = synth.make_hmm_problem(H=HH)
    chroms = prob.chroms
    ns = args.hmm_samples if ns is None else ns
    hmm = DiplotypeHMM(HH, chroms, [len(prob.gene_ids[c]) for c in chroms], [prob.tprob[c] for c in chroms],
                       device=torch.cuda.current_device())
    rng = np.random.default_rng(1)
    ex, av, ha = [], [], []
    for c in chroms:
        ids = prob.gene_ids[c]
        e = np.array([prob.expr[g] for g in ids])
        if ns > 1:
            e = np.stack([e] + [rng.gamma(1.0, 5.0, size=e.shape) * (rng.random(e.shape) < 0.5)
                                for _ in range(ns - 1)])
        ex.append(e)
        ha.append(np.array([g in prob.avecs for g in ids], dtype=np.uint8))
        av.append(np.array([prob.avecs.get(g, np.zeros((HH, HH))) for g in ids]))
    hmm.set_expression(ex, av, ha, 1.5, 0.12)
    hmm.run()
    emis, tot = [], []
    for _ in range(args.hmm_reps):
        hmm.set_expression(ex, av, ha, 1.5, 0.12)    # includes the H2D copy of expr; kernel time is taken from events
        hmm.run()
        inf = hmm.info()
        emis.append(inf.last_emission_ms)
        tot.append(inf.last_emission_ms + inf.last_run_ms)   # device time; the run's chains overlap
    inf = hmm.info()
    ms = float(np.median(tot))
    units = prob.num_genes * ns
    S_ = HH * (HH + 1) // 2
    # 16 S^2 bytes of transition tables per gene (shared by the samples of a batch) + 64 S per gene x sample
    algo = prob.num_genes * (16 * S_ * S_ + 64 * S_ * ns)
    out = dict(metric="HMM gene x sample /s (emission+forward+backward+posterior+Viterbi)",
               value=units / (ms * 1e-3), unit="genes/s", ms_per_pass=ms, n_samples=ns,
               genes=prob.num_genes, states=HH * (HH + 1) // 2,
               kernels_ms=dict(emission=inf.last_emission_ms, forward_viterbi=inf.last_forward_ms,
                               backward_posterior=inf.last_backward_ms, backtrace=inf.last_backtrace_ms,
                               run=inf.last_run_ms,
                               note="forward, backward and Viterbi chains run concurrently: run < sum"),
               roofline=dict(bound="hbm", achieved=algo / (ms * 1e-3) / 1e9,
                             peak=HBM_PEAK_GBS, unit="GB/s",
                             frac=algo / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, traffic=None,
                             algorithmic_bytes=algo,
                             note="sequential recursion over genes: latency-bound, not bandwidth-bound; the "
                                  "transition tables are priced once per batch (SURVEY 8d), the per-sample "
                                  "vectors once per sample"))
    hmm.close()
    if with_cpu and args.hmm_batch > 0 and args.hmm_batch != ns:
        b = hmm_bench(args, torch, ns=args.hmm_batch, with_cpu=False)
        out["batched"] = {k: b[k] for k in ("value", "unit", "ms_per_pass", "n_samples", "kernels_ms", "roofline")}
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


def main():
    args = parse()
    import torch
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: gbrs_amd has no CPU path")
    local = local % torch.cuda.device_count()
    torch.cuda.set_device(local)
    if world > 1:
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device(f"cuda:{local}"))
        else:
            dist.init_process_group(args.backend)
    import __graft_entry__
    if rank == 0:
        __graft_entry__.build()
    if world > 1:
        dist.barrier()

    em = None
    if (world > 1 or args.force_overlap_path) and not args.no_overlap and not args.merge:
        try:
            em = em_bench_pipelined(args, rank, world, torch, dist)
        except Exception as ex:                        # keep the run alive on the single-engine path
            print(f"[bench] rank {rank}: overlapped path failed ({type(ex).__name__}: {ex}); "
                  "falling back to one engine per GPU", file=sys.stderr, flush=True)
            em = None
        if world > 1:                                  # all ranks take the same path
            ok = torch.tensor([1 if em is not None else 0], device=f"cuda:{local}")
            dist.all_reduce(ok, op=dist.ReduceOp.MIN)
            if int(ok.item()) == 0:
                em = None
                torch.cuda.empty_cache()
    if em is None:
        em = em_bench(args, rank, world, torch, dist)
    inf = em["info"]
    infos = em.get("infos", [inf])
    ms_per_step = em["dt"] / args.steps * 1e3
    value = world * args.steps / em["dt"]
    algo = sum(int(i.algorithmic_bytes) for i in infos)
    moved = sum(int(i.bytes_per_iter) for i in infos)
    priced = min(algo, moved)
    estep_s = em["estep_ms"] * 1e-3
    line = {
        "metric": "EM iterations/s (EMASE Model 4, 40M reads x 8 haplotypes x 120k isoforms per GPU)",
        "value": value, "unit": "iters/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f64", "data": "synthetic",
        "config": {"workload": f"configs[1]: single DO sample, R={args.rows} reads x H={args.haps} x "
                               f"L={args.loci} isoforms, N={em['N']} alignment entries, quantify Model 4, "
                               f"tol=0 fixed iterations" + (", rows sharded one 40M-read shard per GPU + "
                               "RCCL all-reduce of the H*L vector per iteration" if world > 1 else "")
                               + (f"; loci cut at gene boundary {em['l_split']} into two engines per GPU, the "
                                  "all-reduce of one range overlapped with the E-step of the other"
                                  if "l_split" in em else ""),
                   "layout": int(inf.layout), "merge_identical_rows": bool(args.merge),
                   "device_rows": sum(int(i.num_device_rows) for i in infos),
                   "device_words": sum(int(i.num_device_words) for i in infos),
                   "tiles": sum(int(i.num_tiles) for i in infos), "slots": sum(int(i.num_slots) for i in infos),
                   "long_rows": sum(int(i.num_long_rows) for i in infos)},
        "roofline": {"bound": "hbm", "achieved": priced / estep_s / 1e9 if estep_s > 0 else None,
                     "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": priced / estep_s / 1e9 / HBM_PEAK_GBS if estep_s > 0 else None,
                     "traffic": None, "kernel": "E-step", "kernel_ms": em["estep_ms"],
                     "step_ms_events": em["step_ms"], "algorithmic_bytes": algo, "layout_bytes": moved,
                     "priced_bytes": priced},
        "setup_s": {"generate": em["t_gen"], "create_layout": em["t_create"],
                    "create_from_host_arrays": em["t_create_host"]},
    }
    if "solve" in em:
        line["time_to_solution"] = dict(em["solve"], rule="err_sum <= 1e6*tol, tol=1e-4 (gbrs quantify default)")
    # HBM traffic of the E-step kernel from the committed rocprofv3 PMC passes (separate runs of this
    # same command, scripts/profile_estep.sh): FETCH_SIZE x2 (gfx950 correction, MI355X_MICROARCH.md)
    # + WRITE_SIZE, per launch.  Only quoted for the exact workload it was measured on.
    try:
        with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as fh:
            pt = json.load(fh)
        key = f"R{args.rows}_H{args.haps}_L{args.loci}_merge{int(args.merge)}"
        if key in pt:
            line["roofline"]["traffic"] = pt[key]["bytes_per_launch"]
            line["roofline"]["traffic_source"] = pt[key]["source"]
    except (OSError, ValueError):
        pass
    if rank == 0:
        if not args.no_cpu_baseline:
            line["cpu_baseline"] = em_cpu_baseline(args)
            line["speedup_vs_cpu"] = (value / world) / line["cpu_baseline"]["value"]
        if not args.merge and not args.no_merged_line and world == 1:
            import copy
            a2 = copy.copy(args)
            a2.merge = True
            m = em_bench(a2, rank, world, torch, dist)
            line["merged_rows_variant"] = {
                "note": "same sample with identical reads merged into weighted rows while building the device "
                        "layout (GBRS_EM_MERGE_IDENTICAL_ROWS, what `gbrs compress` does first); not the headline",
                "value": args.steps / m["dt"], "unit": "iters/s", "ms_per_step": m["dt"] / args.steps * 1e3,
                "estep_kernel_ms": m["estep_ms"], "device_rows": int(m["info"].num_device_rows),
                "device_words": int(m["info"].num_device_words), "layout_bytes": int(m["info"].bytes_per_iter)}
        if not args.no_hmm:
            line["hmm"] = hmm_bench(args, torch)
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
