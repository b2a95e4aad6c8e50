"""AddressSanitizer + UndefinedBehaviorSanitizer build of the library's host-side C++ (CPU build only:
GPU sanitizers are not available on this pool).  gbrs_amd/csrc/hostio.hip holds every host-only entry
point of libgbrs_hip (report writer, number formatter, length-table parser, HDF5 chunk decoder); it is
compiled with g++ -fsanitize=address,undefined together with tests/native/hostio_driver.cpp, which feeds
them valid and malformed inputs.  Any sanitizer report or failed check fails the test."""
import os
import shutil
import subprocess

import pytest

from conftest import ROOT


def test_hostio_under_asan_ubsan(tmp_path):
    cxx = shutil.which("g++")
    if cxx is None:
        pytest.skip("g++ not installed")
    exe = tmp_path / "hostio_asan"
    cmd = [cxx, "-std=c++17", "-O1", "-g", "-fno-omit-frame-pointer", "-fsanitize=address,undefined",
           "-fno-sanitize-recover=undefined", "-DGBRS_HOST_ONLY", "-Wall", "-Wextra", "-x", "c++",
           os.path.join(ROOT, "gbrs_amd", "csrc", "hostio.hip"), os.path.join(ROOT, "tests", "native", "hostio_driver.cpp"),
           "-o", str(exe), "-pthread", "-ldl"]
    build = subprocess.run(cmd, capture_output=True, text=True)
    if build.returncode != 0 and "asan" in build.stderr.lower() and "cannot find" in build.stderr.lower():
        pytest.skip("libasan is not installed")
    assert build.returncode == 0, build.stderr[-3000:]
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1")
    run = subprocess.run([str(exe), str(tmp_path)], capture_output=True, text=True, env=env, timeout=600)
    assert run.returncode == 0, (run.stdout[-2000:], run.stderr[-4000:])
    assert "hostio sanitizer driver: ok" in run.stdout
    assert "runtime error" not in run.stderr and "AddressSanitizer" not in run.stderr
