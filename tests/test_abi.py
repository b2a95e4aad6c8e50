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
    assert hip_lib.gbrs_abi_version() == 1


def test_struct_sizes():
    import ctypes as C
    from gbrs_amd import _lib
    assert C.sizeof(_lib.EmInfo) == 8 * 8 + 4 * 4 + 3 * 8
    assert C.sizeof(_lib.HmmInfo) == 2 * 8 + 4 * 8 + 2 * 4 + 8


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
