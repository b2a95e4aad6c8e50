#!/usr/bin/env python3
"""Host cost of the calls of the two-engine (pipelined) EM loop on one GPU, with a one-rank RCCL group:
how long the interpreter spends enqueueing one iteration, against the time the GPU needs for it.
Usage (GPU box): python scripts/pipelined_host_cost.py [steps]"""
import os
import socket
import sys
import time
import types

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
    import torch
    import torch.distributed as dist
    import __graft_entry__
    __graft_entry__.build()
    import bench
    torch.cuda.set_device(0)
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1,
                            device_id=torch.device("cuda:0"))
    args = types.SimpleNamespace(rows=40_000_000, haps=8, loci=120_000, flags=0)
    state, err = bench.pipelined_setup(args, 0, 1, torch, dist)
    if state is None:
        raise SystemExit(err)
    engs = state["engs"]
    stream = torch.cuda.current_stream().cuda_stream
    for e in engs:
        e.set_stream(stream)
    views = []
    pend = []
    for e in engs:
        p, n = e.prepare_partial()
        views.append(torch.as_tensor(bench.DevArray(p, n), device="cuda:0"))
    for e, v in zip(engs, views):
        dist.all_reduce(v)
        e.finish_prepare(0.0)
    torch.cuda.synchronize()

    def loop(mode, k):
        cost = dict(estep=0.0, allreduce=0.0, wait=0.0, finish=0.0)
        pc = time.perf_counter
        t_all = pc()
        pend = [None, None]
        for i, e in enumerate(engs):
            t = pc(); e.estep_partial(); cost["estep"] += pc() - t
            t = pc()
            pend[i] = dist.all_reduce(views[i], async_op=True) if mode == "async" else (dist.all_reduce(views[i]) if mode == "sync" else None)
            cost["allreduce"] += pc() - t
        for _ in range(k - 1):
            for i, e in enumerate(engs):
                t = pc()
                if mode == "async":
                    pend[i].wait()
                cost["wait"] += pc() - t
                t = pc(); e.finish_step(want_err=False); cost["finish"] += pc() - t
                t = pc(); e.estep_partial(); cost["estep"] += pc() - t
                t = pc()
                pend[i] = dist.all_reduce(views[i], async_op=True) if mode == "async" else (dist.all_reduce(views[i]) if mode == "sync" else None)
                cost["allreduce"] += pc() - t
        for i, e in enumerate(engs):
            if mode == "async":
                pend[i].wait()
            e.finish_step(want_err=False)
        host = pc() - t_all
        torch.cuda.synchronize()
        total = pc() - t_all
        return host, total, cost

    for mode in ("none", "sync", "async"):
        loop(mode, 30)
        host, total, cost = loop(mode, steps)
        print("%-5s: host enqueue %.1f us/iter, wall %.1f us/iter; per iteration (both engines): %s" % (
            mode, host / steps * 1e6, total / steps * 1e6,
            {k: round(v / steps * 1e6, 1) for k, v in cost.items()}), flush=True)

    # every engine on a stream and a process group of its own, collectives issued in line (no handle, no wait):
    # nothing orders the two ranges against each other, the GPU overlaps them by itself
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    groups = [dist.new_group([0]), dist.new_group([0])]
    torch.cuda.synchronize()
    for e, st in zip(engs, streams):
        e.set_stream(st.cuda_stream)

    def loop2(k, collective=True):
        pc = time.perf_counter
        t_all = pc()
        for it in range(k):
            for i, e in enumerate(engs):
                with torch.cuda.stream(streams[i]):
                    if it:
                        e.finish_step(want_err=False)
                    e.estep_partial()
                    if collective:
                        dist.all_reduce(views[i], group=groups[i])
        for i, e in enumerate(engs):
            with torch.cuda.stream(streams[i]):
                e.finish_step(want_err=False)
        host = pc() - t_all
        torch.cuda.synchronize()
        return host, pc() - t_all

    for coll, label in ((False, "no collective"), (True, "in-line all-reduce on a group per engine"),
                        (True, "in-line all-reduce, both engines on the default group")):
        if "default" in label:
            groups = [None, None]
        loop2(30, coll)
        host, total = loop2(steps, coll)
        print("two streams, %s: host enqueue %.1f us/iter, wall %.1f us/iter" % (label, host / steps * 1e6, total / steps * 1e6),
              flush=True)
    for e in engs:
        e.close()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
