"""N > 1 path on CPU: rows sharded over 2 gloo ranks + one all-reduce per iteration reproduce the
unsharded oracle (no GPU: the engine is the numpy stand-in from tests/cpu_engine.py)."""
import os
import socket
import sys

import numpy as np
import pytest

from conftest import ROOT, em_case_inputs, golden_files, load_golden


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, path, out_dir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch
    import torch.distributed as dist
    from conftest import em_case_inputs as inputs, load_golden as load
    from cpu_engine import NumpyEngine
    from gbrs_amd.dist import ShardedEM, shard_rows
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    g = load(path)
    R, L, H, indptr, indices, count, eff_len, groups, gtmask = inputs(g)
    r0, r1, ip, ix, cnt = shard_rows(indptr, indices, count, R, rank, world)
    eng = NumpyEngine(r1 - r0, L, H, ip, ix, cnt, eff_len)

    def allreduce(arr, n):
        t = torch.from_numpy(arr)           # shares memory with the engine's partial buffer
        dist.all_reduce(t)
    drv = ShardedEM(eng, allreduce)
    drv.prepare(float(g["pseudocount"]))
    n = drv.run(model=4, tol=float(g["tol"]), max_iters=int(g["max_iters"]))
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), theta=eng.theta, n=n, rows=np.array([r0, r1]),
             err=np.array(drv.err_history))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("name", ["h8_count_len", "h8_pseudo", "h2_len"])
def test_two_rank_sharded_em_matches_reference(tmp_path, name):
    import torch.multiprocessing as mp
    path = [p for p in golden_files("em") if p.endswith(f"em_{name}.npz")][0]
    g = load_golden(path)
    port = _free_port()
    mp.spawn(_worker, args=(2, port, path, str(tmp_path)), nprocs=2, join=True)
    a = np.load(tmp_path / "rank0.npz")
    b = np.load(tmp_path / "rank1.npz")
    # both ranks hold the same answer, and it is the reference's
    np.testing.assert_array_equal(a["theta"], b["theta"])
    assert int(a["n"]) == int(b["n"]) == int(g["num_iters"])
    np.testing.assert_allclose(a["theta"], g["theta_final"], rtol=1e-9, atol=1e-300)
    np.testing.assert_allclose(a["err"], g["err_history"], rtol=1e-7)
    # the shards partition the rows
    assert a["rows"][0] == 0 and a["rows"][1] == b["rows"][0] and b["rows"][1] == int(g["num_rows"])


def test_shard_rows_balances_entries():
    from gbrs_amd import synth
    from gbrs_amd.dist import shard_rows
    inc = synth.make_em_problem(R=5000, H=4, L=200, seed=1)
    tot = 0
    for k in range(4):
        r0, r1, ip, ix, cnt = shard_rows(inc.indptr, inc.indices, None, inc.num_rows, k, 4)
        n = sum(len(x) for x in ix)
        tot += n
        assert abs(n - inc.nnz / 4) < 0.02 * inc.nnz
        for h in range(4):
            assert int(ip[h][-1]) == len(ix[h]) and (ix[h] < (r1 - r0)).all()
    assert tot == inc.nnz
