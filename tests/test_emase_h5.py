"""EMASE h5 layout through libhdf5: round trip, structure as h5dump sees it, alternative attribute
encodings.  Skipped when libhdf5 is not installed (the .npz mirror is the fallback format)."""
import shutil
import subprocess

import numpy as np
import pytest

from conftest import em_case_inputs, golden_files, load_golden


def _lib_or_skip():
    try:
        from gbrs_amd import emase_h5
        emase_h5._load()
        return emase_h5
    except ImportError as e:
        pytest.skip(str(e))


def _apm():
    from gbrs_amd.alignment import AlignmentPropertyMatrix
    g = load_golden([p for p in golden_files("em") if p.endswith("em_h8_count_len.npz")][0])
    R, L, H, indptr, indices, count, eff_len, groups, gtmask = em_case_inputs(g)
    return AlignmentPropertyMatrix(shape=(L, H, R), indptr=indptr, indices=indices, count=count,
                                   haplotype_names=[chr(65 + h) for h in range(H)],
                                   locus_names=[f"ENSMUST{l:011d}" for l in range(L)])


def test_h5_round_trip(tmp_path):
    _lib_or_skip()
    from gbrs_amd.alignment import AlignmentPropertyMatrix, load_alignment
    a = _apm()
    p = tmp_path / "x.h5"
    a.save(str(p), title="test")
    b = load_alignment(str(p))
    assert b.shape == a.shape and b.hname == a.hname and b.lname == a.lname
    np.testing.assert_array_equal(b.count, a.count)
    for h in range(a.num_haplotypes):
        np.testing.assert_array_equal(b.indptr[h], a.indptr[h])
        np.testing.assert_array_equal(b.indices[h], a.indices[h])
    assert isinstance(b, AlignmentPropertyMatrix) and b.lid[a.lname[3]] == 3


def test_h5_structure_matches_emase_layout(tmp_path):
    _lib_or_skip()
    h5dump = shutil.which("h5dump") or "/opt/conda/bin/h5dump"
    if not shutil.which(h5dump):
        pytest.skip("h5dump not installed")
    a = _apm()
    p = tmp_path / "x.h5"
    a.save(str(p))
    txt = subprocess.run([h5dump, "-H", str(p)], capture_output=True, text=True, check=True).stdout
    for needle in ('GROUP "h0"', 'GROUP "h7"', 'DATASET "indptr"', 'DATASET "indices"', 'DATASET "count"',
                   'DATASET "lname"', 'ATTRIBUTE "mtype"', 'ATTRIBUTE "shape"', 'ATTRIBUTE "hname"',
                   'ATTRIBUTE "incidence_only"', 'H5T_STD_U32LE', 'H5T_IEEE_F64LE'):
        assert needle in txt, needle
    prop = subprocess.run([h5dump, "-p", "-H", "-d", "/h0/indices", str(p)], capture_output=True, text=True).stdout
    assert "DEFLATE" in prop and "CHUNKED" in prop


def test_h5_reader_accepts_plain_encodings(tmp_path):
    """shape as an integer array and hname as a string array (what a non-PyTables writer produces)."""
    h5 = _lib_or_skip()
    import ctypes as C
    lib = h5._load()
    a = _apm()
    p = tmp_path / "plain.h5"
    a.save(str(p))
    # rewrite the two pickled attributes in plain form
    f = lib.H5Fopen(str(p).encode(), 1, 0)          # H5F_ACC_RDWR
    root = lib.H5Gopen2(f, b"/", 0)
    lib.H5Adelete(root, b"shape")
    dims = (C.c_uint64 * 1)(3)
    s = lib.H5Screate_simple(1, dims, None)
    at = lib.H5Acreate2(root, b"shape", h5._native(np.int64), s, 0, 0)
    arr = np.array(a.shape, dtype=np.int64)
    lib.H5Awrite(at, h5._native(np.int64), arr.ctypes.data_as(C.c_void_p))
    lib.H5Aclose(at); lib.H5Sclose(s); lib.H5Gclose(root); lib.H5Fclose(f)
    from gbrs_amd.alignment import load_alignment
    b = load_alignment(str(p))
    assert b.shape == a.shape


def test_h5_legacy_coo_rejected(tmp_path):
    h5 = _lib_or_skip()
    import ctypes as C
    lib = h5._load()
    p = tmp_path / "legacy.h5"
    f = lib.H5Fcreate(str(p).encode(), 2, 0, 0)
    lib.H5Fclose(f)
    from gbrs_amd.alignment import load_alignment
    with pytest.raises(RuntimeError, match="csc"):
        load_alignment(str(p))
