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


def _pipelined_worker(rank, world, port, path, out_dir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch
    import torch.distributed as dist
    from conftest import em_case_inputs as inputs, load_golden as load
    from cpu_engine import NumpyEngine
    from gbrs_amd.dist import (PipelinedShardedEM, balanced_gene_boundary, rows_are_disjoint, shard_rows,
                               split_at_locus)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    g = load(path)
    R, L, H, indptr, indices, count, eff_len, groups, gtmask = inputs(g)
    l_split = balanced_gene_boundary(indptr, sorted(min(m) for m in groups))
    r0, r1, ip, ix, cnt = shard_rows(indptr, indices, count, R, rank, world)
    (a_ip, a_ix), (b_ip, b_ix) = split_at_locus(ip, ix, l_split)
    assert 0 < l_split < L and rows_are_disjoint(a_ix, b_ix, r1 - r0)
    eng_a = NumpyEngine(r1 - r0, l_split, H, a_ip, a_ix, cnt, None if eff_len is None else eff_len[:, :l_split])
    eng_b = NumpyEngine(r1 - r0, L - l_split, H, b_ip, b_ix, cnt, None if eff_len is None else eff_len[:, l_split:])

    class Done:
        def wait(self):
            pass

    def start_allreduce(arr, n):
        dist.all_reduce(torch.from_numpy(arr))      # shares memory with the engine's partial buffer
        return Done()
    drv = PipelinedShardedEM(eng_a, eng_b, start_allreduce)
    drv.prepare(0.0)
    drv.step(int(g["num_iters"]))              # exactly the reference's iteration count
    theta_fixed = drv.theta()
    drv.prepare(0.0)                           # and again under the driver's own stopping rule
    n = drv.run(model=4, tol=float(g["tol"]), max_iters=int(g["max_iters"]), check_every=4)
    theta_run = drv.theta()
    drv.step(1)                                # a step by hand after a run that stopped is applied, not swallowed
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), theta=theta_fixed, theta_run=theta_run, n=n, l_split=l_split,
             err=np.array(drv.err_history), theta_plus1=drv.theta(), n_plus1=drv.num_iters)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("name", ["h8_count_len", "h2_len"])
def test_two_rank_pipelined_halves_match_reference(tmp_path, name):
    """Rows sharded over 2 ranks AND loci cut at a gene boundary into two engines per rank whose
    all-reduces interleave: after the reference's number of iterations theta is the reference's; under
    the driver's own stopping rule - evaluated over both ranges after every iteration, looked at every 4 -
    the run stops at the reference's iteration with the reference's err_sum sequence (EMfactory.py:266-278)."""
    import torch.multiprocessing as mp
    path = [p for p in golden_files("em") if p.endswith(f"em_{name}.npz")][0]
    g = load_golden(path)
    port = _free_port()
    mp.spawn(_pipelined_worker, args=(2, port, path, str(tmp_path)), nprocs=2, join=True)
    a = np.load(tmp_path / "rank0.npz")
    b = np.load(tmp_path / "rank1.npz")
    np.testing.assert_array_equal(a["theta"], b["theta"])
    assert int(a["n"]) == int(b["n"]) == int(g["num_iters"])
    np.testing.assert_allclose(a["err"], g["err_history"], rtol=1e-7)
    np.testing.assert_allclose(a["theta"], g["theta_final"], rtol=1e-9, atol=1e-300)
    np.testing.assert_allclose(a["theta_run"], g["theta_final"], rtol=1e-9, atol=1e-300)   # theta after run(): not a step further
    # run() -> step(1): one more iteration of the oracle from the final state
    from oracle.em_oracle import EMOracle
    R, L, H, indptr, indices, count, eff_len, groups, gtmask = em_case_inputs(g)
    o = EMOracle(R, L, H, indptr, indices, count)
    o.prepare(0.0, eff_len)
    for _ in range(int(g["num_iters"]) + 1):
        o.em_step()
    assert int(a["n_plus1"]) == int(g["num_iters"]) + 1
    np.testing.assert_allclose(a["theta_plus1"], o.theta, rtol=1e-9, atol=1e-300)
    assert not np.allclose(a["theta_plus1"], a["theta_run"], rtol=1e-12, atol=0)


def test_split_at_locus_partitions_entries():
    from gbrs_amd import synth
    from gbrs_amd.dist import rows_are_disjoint, split_at_locus
    inc = synth.make_em_problem(R=3000, H=4, L=120, seed=3)
    (a_ip, a_ix), (b_ip, b_ix) = split_at_locus(inc.indptr, inc.indices, 50)
    for h in range(4):
        assert len(a_ip[h]) == 51 and len(b_ip[h]) == 71 and a_ip[h][0] == 0 and b_ip[h][0] == 0
        assert int(a_ip[h][-1]) == len(a_ix[h]) and int(b_ip[h][-1]) == len(b_ix[h])
        assert len(a_ix[h]) + len(b_ix[h]) == len(inc.indices[h])
        np.testing.assert_array_equal(np.diff(a_ip[h].astype(np.int64)), np.diff(inc.indptr[h].astype(np.int64))[:50])
        np.testing.assert_array_equal(np.diff(b_ip[h].astype(np.int64)), np.diff(inc.indptr[h].astype(np.int64))[50:])
        np.testing.assert_array_equal(np.concatenate([a_ix[h], b_ix[h]]), inc.indices[h])
    # rows_are_disjoint: a row with entries on both sides of the cut is reported
    assert rows_are_disjoint([np.array([0, 1], dtype=np.uint32)], [np.array([2, 3], dtype=np.uint32)], 4)
    assert not rows_are_disjoint([np.array([0, 1], dtype=np.uint32)], [np.array([1, 3], dtype=np.uint32)], 4)


def test_balanced_gene_boundary_picks_the_gene_start_nearest_to_half_the_entries():
    from gbrs_amd.dist import balanced_gene_boundary
    # 10 loci, 1 haplotype: entries per locus 5,5,5,5,0,0,10,10,0,0 -> half (20 entries) is reached at locus 4
    ip = [np.array([0, 5, 10, 15, 20, 20, 20, 30, 40, 40, 40], dtype=np.uint32)]
    assert balanced_gene_boundary(ip, [0, 3, 6, 8]) == 3
    assert balanced_gene_boundary(ip, [0, 5, 9]) == 5
    assert balanced_gene_boundary(ip, [0]) == 0            # a single gene: no cut (callers fall back)
