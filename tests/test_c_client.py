"""The C-ABI from C: tests/native/em_client.c is compiled as C99 against include/gbrs_hip.h, linked with the
built libgbrs_hip.so and run on reference goldens (no Python between the client and the library)."""
import os
import shutil
import struct
import subprocess

import numpy as np
import pytest

from conftest import ROOT, em_case_inputs, golden_files, load_golden

LIB_DIR = os.path.join(ROOT, "gbrs_amd")


def _build_client(tmp_path, name="em_client"):
    cc = shutil.which("gcc")
    if cc is None:
        pytest.skip("gcc not installed")
    if not os.path.exists(os.path.join(LIB_DIR, "libgbrs_hip.so")):
        pytest.skip("libgbrs_hip.so not built")
    exe = tmp_path / name
    cmd = [cc, "-std=c99", "-pedantic", "-Wall", "-Wextra", "-Werror", "-O1", "-I", os.path.join(ROOT, "include"),
           os.path.join(ROOT, "tests", "native", name + ".c"), "-o", str(exe),
           "-L", LIB_DIR, "-lgbrs_hip", f"-Wl,-rpath,{LIB_DIR}", "-Wl,-rpath,/opt/rocm/lib"]
    build = subprocess.run(cmd, capture_output=True, text=True)
    assert build.returncode == 0, build.stderr[-3000:]
    return exe


def test_header_compiles_as_c99_and_client_links(tmp_path):
    """CPU: the header is plain C (-std=c99 -pedantic -Werror) and every symbol the clients use resolves."""
    _build_client(tmp_path)
    _build_client(tmp_path, "hmm_client")


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["em_h8_count_len", "em_h2_plain", "em_h8_pseudo"])
def test_c_client_matches_reference_golden(tmp_path, name):
    paths = [p for p in golden_files("em") if p.endswith(name + ".npz")]
    if not paths:
        pytest.skip(f"no golden {name}")
    g = load_golden(paths[0])
    R, L, H, indptr, indices, count, eff_len, _, gtmask = em_case_inputs(g)
    assert gtmask is None
    exe = _build_client(tmp_path)
    pseudo = float(g["pseudocount"]) if "pseudocount" in g else 0.0
    tol = float(g["tol"]) if "tol" in g else 1e-4
    max_iters = int(g["max_iters"]) if "max_iters" in g else 999
    with open(tmp_path / "in.bin", "wb") as fh:
        fh.write(struct.pack("<QIIIIddII", R, L, H, int(count is not None), int(eff_len is not None), pseudo, tol, max_iters, 0))
        for h in range(H):
            fh.write(np.ascontiguousarray(indptr[h], dtype=np.uint32).tobytes())
            fh.write(struct.pack("<I", len(indices[h])))
            fh.write(np.ascontiguousarray(indices[h], dtype=np.uint32).tobytes())
        if count is not None:
            fh.write(np.ascontiguousarray(count, dtype=np.float64).tobytes())
        if eff_len is not None:
            fh.write(np.ascontiguousarray(eff_len, dtype=np.float64).tobytes())
    run = subprocess.run([str(exe), str(tmp_path / "in.bin"), str(tmp_path / "out.bin")], capture_output=True, text=True,
                         timeout=300)
    assert run.returncode == 0, (run.stdout, run.stderr[-2000:])
    raw = open(tmp_path / "out.bin", "rb").read()
    n_iters = struct.unpack_from("<i", raw)[0]
    theta = np.frombuffer(raw, dtype=np.float64, count=H * L, offset=8).reshape(H, L)
    counts = np.frombuffer(raw, dtype=np.float64, count=H * L, offset=8 + 8 * H * L).reshape(H, L)
    assert n_iters == int(g["num_iters"])
    np.testing.assert_allclose(theta, g["theta_final"], rtol=1e-9, atol=1e-12)
    np.testing.assert_allclose(counts, g["expected_counts"], rtol=1e-9, atol=1e-12)


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["hmm_h8_full", "hmm_h8_do_full", "hmm_h8_short", "hmm_h4_full"])
def test_c_hmm_client_matches_reference_golden(tmp_path, name):
    from conftest import hmm_case_inputs
    paths = [p for p in golden_files("hmm") if p.endswith(name + ".npz")]
    if not paths:
        pytest.skip(f"no golden {name}")
    g = load_golden(paths[0])
    c = hmm_case_inputs(g)
    H, chroms = c["H"], c["chroms"]
    S = H * (H + 1) // 2
    exe = _build_client(tmp_path, "hmm_client")
    with open(tmp_path / "in.bin", "wb") as fh:
        fh.write(struct.pack("<iidd", H, len(chroms), float(g["expr_threshold"]), float(g["sigma"])))
        for ch in chroms:
            n = len(c["genes"][ch])
            tp = np.ascontiguousarray(c["tprob"][ch], dtype=np.float64)
            fh.write(struct.pack("<ii", n, tp.shape[0]))
            fh.write(tp.tobytes())
            fh.write(np.ascontiguousarray(c["expr"][ch], dtype=np.float64).reshape(n, H).tobytes())
            fh.write(np.ascontiguousarray(c["avecs"][ch], dtype=np.float64).reshape(n, H, H).tobytes())
            flags = np.zeros((n + 7) // 8 * 8, dtype=np.uint8)
            flags[:n] = np.asarray(c["has_avec"][ch], dtype=np.uint8)
            fh.write(flags.tobytes())
    run = subprocess.run([str(exe), str(tmp_path / "in.bin"), str(tmp_path / "out.bin")], capture_output=True, text=True,
                         timeout=300)
    assert run.returncode == 0, (run.stdout, run.stderr[-2000:])
    raw = open(tmp_path / "out.bin", "rb").read()
    at = 0
    for ch in chroms:
        n = len(c["genes"][ch])
        gamma = np.frombuffer(raw, dtype=np.float64, count=S * n, offset=at).reshape(S, n)
        at += 8 * S * n
        calls = np.frombuffer(raw, dtype=np.int32, count=n, offset=at)
        at += 4 * n
        n_path = struct.unpack_from("<i", raw, at)[0]
        at += 4
        states = np.frombuffer(raw, dtype=np.int32, count=n_path, offset=at)
        at += 4 * n_path
        np.testing.assert_allclose(gamma, g[f"gamma_{ch}"], rtol=1e-8, atol=1e-300)
        np.testing.assert_array_equal(calls, g[f"calls_{ch}"])
        np.testing.assert_array_equal(states, g[f"states_{ch}"])
    assert at == len(raw)
