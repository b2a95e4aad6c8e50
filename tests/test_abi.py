"""The C-ABI library loads and exports every symbol include/gbrs_hip.h declares (no compute)."""
import os
import re

from conftest import ROOT


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "gbrs_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(gbrs_[a-z_0-9]+)\s*\(", text)))


def test_header_symbols_exported(hip_lib):
    from gbrs_amd import _lib
    names = declared_symbols()
    assert len(names) >= 25
    for n in names:
        assert hasattr(hip_lib, n), f"{n} declared in gbrs_hip.h but not exported"
    assert sorted(_lib.EXPORTS) == names
    assert hip_lib.gbrs_abi_version() == 5


def test_header_constants_match_python_mirror():
    """Every status code and gbrs_em_create flag of the header has the same value in gbrs_amd/_lib.py."""
    from gbrs_amd import _lib
    text = open(os.path.join(ROOT, "include", "gbrs_hip.h")).read()
    consts = dict(re.findall(r"^#define\s+(GBRS_(?:EM|ERR|OK)[A-Z_]*)\s+\(?(-?\d+)u?\)?\s*$", text, flags=re.M))
    consts.update(re.findall(r"^\s+(GBRS_(?:OK|ERR_[A-Z_]+))\s*=\s*(-?\d+),", text, flags=re.M))      # the status enum
    assert {"GBRS_OK", "GBRS_ERR_FLOAT", "GBRS_EM_DETERMINISTIC", "GBRS_EM_KEEP_CSC", "GBRS_EM_SIDE_BY_SIDE"} <= set(consts)
    for name, value in consts.items():
        assert getattr(_lib, name) == int(value), name
    flags = sorted(int(v) for k, v in consts.items() if k.startswith("GBRS_EM_") and int(v) > 0)
    assert flags == [1 << i for i in range(len(flags))]          # distinct bits, none skipped


def test_struct_sizes(tmp_path):
    """The ctypes mirrors of the info structs have the size and field offsets a C compiler gives the
    declarations in include/gbrs_hip.h (the header must also compile as plain C)."""
    import ctypes as C
    import shutil
    import subprocess
    from gbrs_amd import _lib
    assert C.sizeof(_lib.EmInfo) == 8 * 8 + 4 * 4 + 8 * 8
    assert C.sizeof(_lib.HmmInfo) == 2 * 8 + 4 * 8 + 2 * 4 + 8 + 4 * 4
    cc = shutil.which("gcc") or shutil.which("cc")
    if cc is None:
        return
    src = tmp_path / "sz.c"
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "gbrs_hip.h"\n'
                   'int main(void){printf("%zu %zu %zu %zu %zu %zu\\n", sizeof(gbrs_em_info_t), '
                   'sizeof(gbrs_hmm_info_t), offsetof(gbrs_em_info_t, estep_bytes), '
                   'offsetof(gbrs_em_info_t, num_tiles), offsetof(gbrs_hmm_info_t, last_run_ms), '
                   'offsetof(gbrs_em_info_t, last_estep_ms));return 0;}\n')
    exe = tmp_path / "sz"
    subprocess.run([cc, "-std=c99", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)],
                   check=True)
    got = [int(x) for x in subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.split()]
    assert got == [C.sizeof(_lib.EmInfo), C.sizeof(_lib.HmmInfo), _lib.EmInfo.estep_bytes.offset,
                   _lib.EmInfo.num_tiles.offset, _lib.HmmInfo.last_run_ms.offset, _lib.EmInfo.last_estep_ms.offset]


def test_no_cpu_fallback_without_device(hip_lib):
    """On a box without a GPU every compute entry point must fail loudly, never compute."""
    import ctypes as C
    import numpy as np
    import pytest
    from gbrs_amd import _lib
    if hip_lib.gbrs_device_count() > 0:
        pytest.skip("a HIP device is visible")
    ip = [np.zeros(3, dtype=np.uint32)]
    ix = [np.zeros(0, dtype=np.uint32)]
    h = C.c_void_p()
    st = hip_lib.gbrs_em_create(1, 2, 1, _lib.ptr_table(ip), _lib.ptr_table(ix), None, None, 0, 0, C.byref(h))
    assert st == _lib.GBRS_ERR_NO_DEVICE
    with pytest.raises(_lib.GbrsHipError):
        _lib.check(st)
