"""gbrs_amd/npzfast.py (the `.npz` readers of `gbrs reconstruct`) against numpy.load: stored and deflated members,
the native central-directory / member-stack helpers of libgbrs_hip and their pure-Python fallbacks, odd members,
damaged files.  Host code only."""
import os
import zipfile

import numpy as np
import pytest

from gbrs_amd import npzfast


def _blocks(n, seed=0):
    rng = np.random.default_rng(seed)
    return {f"ENSMUSG{k:011d}": rng.random((8, 8)) for k in range(n)}


@pytest.mark.parametrize("compressed", [False, True], ids=["stored", "deflated"])
@pytest.mark.parametrize("native", [True, False], ids=["native", "python"])
def test_stack_matches_numpy(tmp_path, compressed, native, monkeypatch):
    blocks = _blocks(3000)
    path = tmp_path / "avecs.npz"
    (np.savez_compressed if compressed else np.savez)(path, **blocks)
    if not native:
        monkeypatch.setattr(npzfast.FastNpz, "_native_directory", lambda self: None)
    z = npzfast.FastNpz(str(path))
    assert (z._native is not None) == native
    assert sorted(z.files) == sorted(blocks)
    names = list(blocks)[::3] + list(blocks)[1::7]
    got = z.stack(names, (8, 8))
    want = np.stack([blocks[n] for n in names])
    np.testing.assert_array_equal(got, want)
    ref = np.load(path)
    for n in names[:25]:
        np.testing.assert_array_equal(z[n], ref[n])
    z.close()


def test_stack_mixed_members_and_wrong_shape(tmp_path):
    """Members with another dtype / header go through the per-member path; another shape is an error."""
    blocks = _blocks(50)
    blocks["as_float32"] = np.arange(64, dtype=np.float32).reshape(8, 8)
    blocks["fortran"] = np.asfortranarray(np.arange(64, dtype=np.float64).reshape(8, 8))
    path = tmp_path / "mixed.npz"
    np.savez_compressed(path, **blocks)
    z = npzfast.FastNpz(str(path))
    names = list(blocks)
    got = z.stack(names, (8, 8))
    np.testing.assert_array_equal(got, np.stack([np.asarray(blocks[n], dtype=np.float64) for n in names]))
    np.savez(tmp_path / "bad.npz", a=np.zeros((8, 8)), b=np.zeros((4, 4)))
    z2 = npzfast.FastNpz(str(tmp_path / "bad.npz"))
    with pytest.raises(ValueError):
        z2.stack(["a", "b"], (8, 8))


def test_large_members_and_read_many(tmp_path):
    rng = np.random.default_rng(3)
    arrays = {f"chr{c}": rng.random((300, 36, 36)) for c in range(1, 6)}
    path = tmp_path / "tprob.npz"
    np.savez_compressed(path, **arrays)
    z = npzfast.FastNpz(str(path))
    for name, a in zip(arrays, z.read_many(list(arrays))):
        np.testing.assert_array_equal(a, arrays[name])
    np.savez(path, **arrays)                      # stored: mapped, not copied
    z = npzfast.FastNpz(str(path))
    np.testing.assert_array_equal(z["chr3"], arrays["chr3"])


def test_native_directory_matches_zipfile(tmp_path):
    lib = pytest.importorskip("gbrs_amd._lib")
    try:
        lib.load()
    except OSError:
        pytest.skip("libgbrs_hip.so not built")
    path = tmp_path / "many.npz"
    with zipfile.ZipFile(path, "w", zipfile.ZIP_DEFLATED) as zf:
        for k in range(2000):
            with zf.open(f"m{k}.npy", "w", force_zip64=(k % 5 == 0)) as fh:      # zip64 local headers, as numpy writes
                np.lib.format.write_array(fh, np.full((k % 7 + 1,), k, dtype=np.int32))
        zf.comment = b"a trailing comment moves the end record"
    z = npzfast.FastNpz(str(path))
    assert z._native is not None
    with zipfile.ZipFile(path) as zf:
        for info in zf.infolist():
            m = z._info[info.filename[:-4]]
            assert (m.compress_type, m.compress_size, m.file_size, m.header_offset) == \
                   (info.compress_type, info.compress_size, info.file_size, info.header_offset)
    np.testing.assert_array_equal(z["m13"], np.full((7,), 13, dtype=np.int32))


def test_damaged_files_fall_back_or_fail_cleanly(tmp_path):
    path = tmp_path / "ok.npz"
    np.savez(path, **_blocks(20))
    raw = open(path, "rb").read()
    # central directory cut off: neither parser may read out of bounds; numpy's error is the user's error
    open(tmp_path / "cut.npz", "wb").write(raw[:len(raw) - 40])
    with pytest.raises(Exception):
        npzfast.FastNpz(str(tmp_path / "cut.npz"))
    # a local header offset pointing outside the file
    bad = bytearray(raw)
    cd = raw.rfind(b"PK\x01\x02")
    bad[cd + 42:cd + 46] = (len(raw) + 1000).to_bytes(4, "little")
    open(tmp_path / "off.npz", "wb").write(bytes(bad))
    z = npzfast.FastNpz(str(tmp_path / "off.npz"))
    last = list(z.files)[-1]
    with pytest.raises(Exception):
        z.stack([last, list(z.files)[0]], (8, 8))


def test_bit_rot_is_reported_like_numpy_load(tmp_path, monkeypatch):
    """A member whose payload no longer matches its CRC-32 raises BadZipFile when read on its own (numpy.load's
    behaviour, which the reference inherits), on its own and through the bulk paths (read_many, stack)."""
    import zipfile
    from gbrs_amd import npzfast
    rng = np.random.default_rng(3)
    arrays = {f"g{k:03d}": rng.random((8, 8)) for k in range(20)}
    arrays["big"] = rng.random(100_000)
    for writer, name in ((np.savez, "stored.npz"), (np.savez_compressed, "deflated.npz")):
        p = tmp_path / name
        writer(p, **arrays)
        raw = bytearray(p.read_bytes())
        z = npzfast.FastNpz(str(p))
        zi = z._info["g007"]
        _, off, csize = z._payload(zi)
        z.close()
        raw[off + csize - 3] ^= 0x5A                       # inside the member's data, sizes unchanged
        bad = tmp_path / ("bad_" + name)
        bad.write_bytes(bytes(raw))
        z = npzfast.FastNpz(str(bad))
        assert z._info["g007"].CRC is not None
        np.testing.assert_array_equal(z["g006"], arrays["g006"])
        assert z["g006"].flags.writeable and z["g006"].flags.aligned
        with pytest.raises((zipfile.BadZipFile, zlib_error())):
            z["g007"]
        # the bulk paths check by default (on the native threads that produce the bytes); GBRS_VERIFY_CRC=0 opts out
        assert npzfast.VERIFY_ALL
        with pytest.raises((zipfile.BadZipFile, zlib_error())):
            z.read_many(list(arrays))
        genes = [f"g{k:03d}" for k in range(20)]
        with pytest.raises((zipfile.BadZipFile, zlib_error())):
            z.stack(genes, (8, 8))
        np.testing.assert_array_equal(z.stack(genes[:7], (8, 8)), np.stack([arrays[n] for n in genes[:7]]))
        z.close()
    # damage in a LARGE deflated member: only the bulk path's own check can see it
    p = tmp_path / "tables.npz"
    big = {f"c{k}": rng.random(300_000) for k in range(6)}
    np.savez_compressed(p, **big)
    z = npzfast.FastNpz(str(p))
    zi = z._info["c3"]
    _, off, csize = z._payload(zi)
    z.close()
    raw = bytearray(p.read_bytes())
    raw[off + csize // 2] ^= 0x01
    bad = tmp_path / "bad_tables.npz"
    bad.write_bytes(bytes(raw))
    z = npzfast.FastNpz(str(bad))
    with pytest.raises((zipfile.BadZipFile, zlib_error())):
        z.read_many(list(big))
    z.close()


def zlib_error():
    import zlib
    return zlib.error
