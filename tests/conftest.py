import glob
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLD = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def golden_files(prefix):
    return sorted(glob.glob(os.path.join(GOLD, prefix + "_*.npz")))


def load_golden(path):
    with np.load(path, allow_pickle=False) as z:
        return {k: z[k] for k in z.files}


def em_case_inputs(g):
    """(R, L, H, indptr, indices, count, eff_len, groups, gtmask) from an em_*.npz fixture."""
    H, L, R = int(g["num_haps"]), int(g["num_loci"]), int(g["num_rows"])
    indptr = [g[f"indptr{h}"] for h in range(H)]
    indices = [g[f"indices{h}"] for h in range(H)]
    count = g["count"] if bool(g["has_count"]) else None
    eff_len = g["eff_len"] if bool(g["has_len"]) else None
    gp, gm = g["group_ptr"], g["group_members"]
    groups = [list(gm[gp[i]:gp[i + 1]]) for i in range(len(gp) - 1)]
    gtmask = g["gtmask"] if bool(g["has_mask"]) else None
    return R, L, H, indptr, indices, count, eff_len, groups, gtmask


def em_case_values(g):
    """Stored alignment values of an em_*.npz fixture (list per haplotype), or None."""
    H = int(g["num_haps"])
    return [g[f"values{h}"] for h in range(H)] if "values0" in g else None


def hmm_case_inputs(g):
    H = int(g["num_haps"])
    chroms = [str(c) for c in g["chroms"]]
    out = dict(H=H, chroms=chroms)
    for k in ("genes", "tprob", "expr", "has_avec", "avecs"):
        out[k] = {c: g[f"{k}_{c}"] for c in chroms}
    return out


@pytest.fixture(scope="session")
def hip_lib():
    from gbrs_amd import _lib
    return _lib.load()


@pytest.fixture(scope="session", autouse=True)
def _torch_cuda_first(request):
    """The few GPU tests that hand device pointers to the C ABI use torch as allocator; initialise
    its HIP context at session start (late initialisation, after other tests have spawned worker
    processes and left numpy in the reference's `np.seterr(all='raise')` state, was flaky)."""
    if request.config.getoption("-m") and "not gpu" in request.config.getoption("-m"):
        return
    try:
        import torch
        if torch.cuda.is_available():
            torch.cuda.init()
            torch.zeros(1, device="cuda")
    except Exception:
        pass


def viterbi_decision_margins(T, delta):
    """Gap between the best and the second-best candidate at every argmax the reference's backtrace
    takes (gbrs_utils.py:586-596): the last column of delta, then delta[:, i] + T[i][sid] walking back.
    A device delta that differs from the reference's by much less than the smallest gap gives the same
    calls."""
    S, n = delta.shape
    v = delta[:, n - 1]
    top = np.sort(v)[::-1]
    gaps = [top[0] - top[1]] if S > 1 else []
    sid = int(v.argmax())
    for i in reversed(range(min(n, len(T)))):
        v = delta[:, i] + T[i][sid]
        top = np.sort(v)[::-1]
        if S > 1:
            gaps.append(top[0] - top[1])
        sid = int(v.argmax())
    return np.asarray(gaps)
