"""EMASE alignment files in the PyTables/HDF5 layout, read and written through libhdf5 (ctypes).

Layout, as written by the reference (emase/Sparse3DMatrix.py:400-444,
emase/AlignmentPropertyMatrix.py:478-525; second writer emase/AlignmentMatrixFactory.py:84-142):

    /                attrs  incidence_only (bool)  mtype = 'csc_matrix'  shape = (L, H, R)  hname = [..]
    /h0 .. /h{H-1}   groups with CArrays  indptr uint32[L+1], indices uint32[nnz] (row ids), [data]
    /count           float64[R]  (EC multiplicities, optional)
    /lname           fixed-length byte strings [L];   /rname optional
    all arrays chunked + zlib level 1

PyTables stores tuple/list attributes as pickled byte strings and scalars natively.  Neither
PyTables nor h5py exists in this image, so the reader accepts every encoding HDF5 allows for
these fields (pickled string, integer array, string array, bool/enum/int scalar) and the writer
emits the PyTables conventions as documented; a file written by real PyTables could not be
produced here to confirm.  What has been verified, and how:
  * group / dataset names, dtypes, chunking, shuffle + deflate filters: `h5dump -H -p` of this
    writer's output (tests/test_emase_h5.py);
  * attribute encodings - pickled protocol-0 strings for `shape` / `hname`, a native string for
    `mtype`, an 8-bit integer for `incidence_only`: by this module's own reader and `h5dump` only,
    never by PyTables itself (the reader therefore also takes integer arrays, string arrays, enum
    booleans and variable-length strings for them);
  * values (`incidence_only` false -> /h*/data) and the legacy COO form (files without `mtype` or
    `incidence_only`: /h*/coor + /h*/data, Sparse3DMatrix.py:68-78, :93-99): files written by the
    test-suite with libhdf5 in the shapes the reference's reader expects.

Large index arrays are decoded in parallel: HDF5 chunks are independent deflate streams, so the
reader takes every chunk's file address from H5Dget_chunk_info and inflates (libdeflate when
present, else zlib) + unshuffles the chunks on a thread pool straight into the output array; the
HDF5 library itself is only used for metadata there.
"""
from __future__ import annotations

import ctypes as C
import ctypes.util
import os
import pickle

import numpy as np

hid_t = C.c_int64
_lib = None

H5F_ACC_RDONLY, H5F_ACC_TRUNC = 0, 2
H5P_DEFAULT = 0
H5S_ALL = 0
H5T_INTEGER, H5T_FLOAT, H5T_STRING, H5T_ENUM = 0, 1, 3, 8
H5S_SCALAR = 0


def _load():
    global _lib
    if _lib is not None:
        return _lib
    cands = [os.environ.get('GBRS_LIBHDF5'), '/opt/conda/lib/libhdf5.so', ctypes.util.find_library('hdf5')]
    err = None
    for c in cands:
        if not c:
            continue
        try:
            lib = C.CDLL(c)
            break
        except OSError as e:     # noqa: PERF203
            err = e
    else:
        raise ImportError(f'libhdf5 not found (set GBRS_LIBHDF5=/path/to/libhdf5.so): {err}; '
                          'use the .npz mirror format instead (AlignmentPropertyMatrix.save_npz)')
    lib.H5open()
    lib.H5Eset_auto2.argtypes = [hid_t, C.c_void_p, C.c_void_p]
    lib.H5Eset_auto2(0, None, None)          # we raise Python errors ourselves
    for name in ('H5Fopen', 'H5Fcreate', 'H5Gopen2', 'H5Gcreate2', 'H5Dopen2', 'H5Dcreate2', 'H5Dget_space',
                 'H5Dget_type', 'H5Aopen', 'H5Aget_type', 'H5Aget_space', 'H5Acreate2', 'H5Tcopy',
                 'H5Screate_simple', 'H5Screate', 'H5Pcreate', 'H5Dget_create_plist'):
        getattr(lib, name).restype = hid_t
    lib.H5Tget_size.restype = C.c_size_t
    lib.H5Sget_simple_extent_npoints.restype = C.c_int64
    H, VP, CP, U, I, SZ = hid_t, C.c_void_p, C.c_char_p, C.c_uint, C.c_int, C.c_size_t
    sig = dict(
        H5Fopen=[CP, U, H], H5Fcreate=[CP, U, H, H], H5Fclose=[H],
        H5Gopen2=[H, CP, H], H5Gcreate2=[H, CP, H, H, H], H5Gclose=[H],
        H5Dopen2=[H, CP, H], H5Dcreate2=[H, CP, H, H, H, H, H], H5Dget_space=[H], H5Dget_type=[H],
        H5Dread=[H, H, H, H, H, VP], H5Dwrite=[H, H, H, H, H, VP], H5Dclose=[H],
        H5Aexists=[H, CP], H5Aopen=[H, CP, H], H5Aget_type=[H], H5Aget_space=[H], H5Aread=[H, H, VP],
        H5Acreate2=[H, CP, H, H, H, H], H5Awrite=[H, H, VP], H5Aclose=[H], H5Adelete=[H, CP],
        H5Tcopy=[H], H5Tset_size=[H, SZ], H5Tget_class=[H], H5Tget_size=[H], H5Tget_sign=[H],
        H5Tis_variable_str=[H], H5Tclose=[H],
        H5Screate_simple=[I, VP, VP], H5Screate=[I], H5Sclose=[H], H5Sget_simple_extent_ndims=[H],
        H5Sget_simple_extent_dims=[H, VP, VP],
        H5Pcreate=[H], H5Pset_chunk=[H, I, VP], H5Pset_deflate=[H, U], H5Pclose=[H],
        H5Lexists=[H, CP, H],
        H5Dget_create_plist=[H], H5Pget_chunk=[H, I, VP], H5Pget_nfilters=[H], H5Pget_layout=[H],
        H5Pget_filter2=[H, U, VP, VP, VP, SZ, CP, VP], H5Pset_shuffle=[H], H5Tget_order=[H],
    )
    for name, args in sig.items():
        getattr(lib, name).argtypes = args
    # the chunk queries (HDF5 >= 1.10.5) and the user-block query only serve the parallel chunk decoder: without them
    # every dataset goes through H5Dread
    optional = dict(H5Dget_num_chunks=[H, H, VP], H5Dget_chunk_info=[H, H, C.c_uint64, VP, VP, VP, VP],
                    H5Fget_create_plist=[H], H5Pget_userblock=[H, VP], H5Iget_file_id=[H])
    lib.gbrs_has_chunk_queries = True
    for name, args in optional.items():
        try:
            getattr(lib, name).argtypes = args
        except AttributeError:
            lib.gbrs_has_chunk_queries = False
    if lib.gbrs_has_chunk_queries:
        lib.H5Fget_create_plist.restype = hid_t
        lib.H5Iget_file_id.restype = hid_t
    _lib = lib
    return lib


def _g(name):
    return C.c_int64.in_dll(_load(), name).value


def _native(dtype):
    dtype = np.dtype(dtype)
    table = {'uint32': 'H5T_NATIVE_UINT32_g', 'int32': 'H5T_NATIVE_INT32_g', 'uint64': 'H5T_NATIVE_UINT64_g',
             'int64': 'H5T_NATIVE_INT64_g', 'float64': 'H5T_NATIVE_DOUBLE_g', 'float32': 'H5T_NATIVE_FLOAT_g',
             'int8': 'H5T_NATIVE_INT8_g', 'uint8': 'H5T_NATIVE_UINT8_g', 'uint16': 'H5T_NATIVE_UINT16_g',
             'int16': 'H5T_NATIVE_INT16_g'}
    return _g(table[dtype.name])


def _check(v, what):
    if (v.value if isinstance(v, hid_t) else v) < 0:
        raise RuntimeError(f'HDF5 error in {what}')
    return v


def _id(v):
    return v.value if isinstance(v, hid_t) else int(v)


def _dims(space):
    lib = _load()
    nd = lib.H5Sget_simple_extent_ndims(space)
    dims = (C.c_uint64 * max(nd, 1))()
    if nd > 0:
        lib.H5Sget_simple_extent_dims(space, dims, None)
    return tuple(int(dims[i]) for i in range(nd))


def _read_typed(read, obj, ftype, space):
    """Read a dataset/attribute into a numpy array given its file type and dataspace."""
    lib = _load()
    shape = _dims(space)
    cls = lib.H5Tget_class(ftype)
    size = int(lib.H5Tget_size(ftype))
    if cls in (H5T_INTEGER, H5T_ENUM):
        signed = lib.H5Tget_sign(ftype) == 1 if cls == H5T_INTEGER else True
        dt = np.dtype(f"{'i' if signed else 'u'}{size}")
        out = np.empty(shape, dtype=dt)
        mem = _native(dt) if cls == H5T_INTEGER else ftype
        _check(read(obj, mem, out.ctypes.data_as(C.c_void_p)), 'read')
        return out
    if cls == H5T_FLOAT:
        dt = np.dtype(f'f{size}')
        out = np.empty(shape, dtype=dt)
        _check(read(obj, _native(dt), out.ctypes.data_as(C.c_void_p)), 'read')
        return out
    if cls == H5T_STRING:
        if lib.H5Tis_variable_str(ftype) > 0:
            n = int(np.prod(shape)) if shape else 1
            ptrs = (C.c_char_p * n)()
            _check(read(obj, ftype, ptrs), 'read')
            vals = [ptrs[i] or b'' for i in range(n)]
            return np.array(vals, dtype=object).reshape(shape) if shape else np.array(vals[0], dtype=object)
        out = np.empty(shape, dtype=f'S{size}')
        _check(read(obj, ftype, out.ctypes.data_as(C.c_void_p)), 'read')
        return out
    raise RuntimeError(f'unsupported HDF5 type class {cls}')


H5Z_FILTER_DEFLATE, H5Z_FILTER_SHUFFLE = 1, 2
H5D_CHUNKED = 2
PARALLEL_MIN_BYTES = 1 << 22          # below this the plain H5Dread is as fast


def _inflater():
    """(name, fn(raw bytes, out_size) -> bytes): libdeflate through ctypes when the image has it (about
    three times zlib's speed per core; ctypes and zlib both release the GIL while they run)."""
    global _inflate_impl
    try:
        return _inflate_impl
    except NameError:
        pass
    import zlib
    impl = ('zlib', lambda raw, n: zlib.decompress(raw, bufsize=n))
    for cand in (os.environ.get('GBRS_LIBDEFLATE'), '/opt/conda/lib/libdeflate.so', ctypes.util.find_library('deflate')):
        if not cand:
            continue
        try:
            ld = C.CDLL(cand)
            ld.libdeflate_alloc_decompressor.restype = C.c_void_p
            ld.libdeflate_zlib_decompress.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t, C.c_void_p, C.c_size_t,
                                                     C.POINTER(C.c_size_t)]
            ld.libdeflate_zlib_decompress.restype = C.c_int
            import threading
            local = threading.local()

            def inflate(raw, n, ld=ld, local=local):
                d = getattr(local, 'd', None)
                if d is None:
                    d = local.d = C.c_void_p(ld.libdeflate_alloc_decompressor())
                out = (C.c_char * n)()
                got = C.c_size_t(0)
                if ld.libdeflate_zlib_decompress(d, raw, len(raw), out, n, C.byref(got)) != 0:
                    return zlib.decompress(raw, bufsize=n)
                return memoryview(out)[:got.value]
            impl = ('libdeflate', inflate)
            break
        except (OSError, AttributeError):
            continue
    _inflate_impl = impl
    return impl


def _read_chunks_parallel(path, d, ftype, space):
    """Decode a 1-D chunked dataset (numeric, little endian, filters within {shuffle, deflate}) chunk by
    chunk on a thread pool.  Returns the array, or None when the dataset does not qualify."""
    lib = _load()
    if not lib.gbrs_has_chunk_queries:
        return None
    shape = _dims(space)
    cls, size = lib.H5Tget_class(ftype), int(lib.H5Tget_size(ftype))
    if len(shape) != 1 or cls not in (H5T_INTEGER, H5T_FLOAT) or lib.H5Tget_order(ftype) != 0:
        return None
    # chunk addresses are relative to the file's base address: a file with a user block (none that PyTables writes)
    # goes through H5Dread
    fid = lib.H5Iget_file_id(d)
    if _id(fid) < 0:
        return None
    try:
        fcpl = lib.H5Fget_create_plist(fid)
        if _id(fcpl) < 0:
            return None
        try:
            ub = C.c_uint64(1)
            if lib.H5Pget_userblock(fcpl, C.byref(ub)) < 0 or ub.value != 0:
                return None
        finally:
            lib.H5Pclose(fcpl)
    finally:
        lib.H5Fclose(fid)
    n = shape[0]
    if n * size < PARALLEL_MIN_BYTES:
        return None
    if cls == H5T_INTEGER:
        dt = np.dtype(f"<{'i' if lib.H5Tget_sign(ftype) == 1 else 'u'}{size}")
    else:
        dt = np.dtype(f'<f{size}')
    plist = lib.H5Dget_create_plist(d)
    try:
        if lib.H5Pget_layout(plist) != H5D_CHUNKED:
            return None
        cdim = (C.c_uint64 * 1)()
        if lib.H5Pget_chunk(plist, 1, cdim) != 1:
            return None
        chunk = int(cdim[0])
        filters = []
        for k in range(lib.H5Pget_nfilters(plist)):
            flags, ncd, cfg = C.c_uint(0), C.c_size_t(0), C.c_uint(0)
            filters.append(lib.H5Pget_filter2(plist, k, C.byref(flags), C.byref(ncd), None, 0, None, C.byref(cfg)))
        if filters not in ([], [H5Z_FILTER_DEFLATE], [H5Z_FILTER_SHUFFLE], [H5Z_FILTER_SHUFFLE, H5Z_FILTER_DEFLATE]):
            return None
    finally:
        lib.H5Pclose(plist)
    nchunks = C.c_uint64(0)
    if lib.H5Dget_num_chunks(d, space, C.byref(nchunks)) < 0:
        return None
    todo = []
    off, fmask, addr, csize = (C.c_uint64 * 1)(), C.c_uint(0), C.c_uint64(0), C.c_uint64(0)
    for k in range(nchunks.value):
        if lib.H5Dget_chunk_info(d, space, k, off, C.byref(fmask), C.byref(addr), C.byref(csize)) < 0:
            return None
        todo.append((int(off[0]), int(fmask.value), int(addr.value), int(csize.value)))
    out = np.zeros(n, dtype=dt)            # chunks that were never written read as the fill value 0
    try:
        from . import _lib as _native
        lib_native = _native.load()
    except ImportError:
        lib_native = None
    if lib_native is not None:
        # pread + inflate + un-shuffle on C++ threads straight into `out` (the library's host-side helper)
        tab = np.array(todo, dtype=np.uint64).reshape(len(todo), 4)
        start = np.ascontiguousarray(tab[:, 0])
        mask = np.ascontiguousarray(tab[:, 1].astype(np.uint32))
        addr = np.ascontiguousarray(tab[:, 2])
        stored = np.ascontiguousarray(tab[:, 3])
        pos = {f: k for k, f in enumerate(filters)}
        st = lib_native.gbrs_decode_chunks(
            os.fsencode(path), len(todo), _native.ptr(addr), _native.ptr(stored), _native.ptr(start), _native.ptr(mask),
            chunk, size, n, pos.get(H5Z_FILTER_SHUFFLE, -1), pos.get(H5Z_FILTER_DEFLATE, -1), _native.ptr(out),
            int(os.environ.get('GBRS_IO_THREADS', 0)))
        if st == _native.GBRS_OK:
            return out
        if st != _native.GBRS_ERR_UNSUPPORTED:          # a damaged file is an error; a missing inflate library is not
            _native.check(st)
        out[:] = 0
    return _decode_chunks_python(path, todo, filters, chunk, size, n, out)


def _decode_chunks_python(path, todo, filters, chunk, size, n, out):
    """The same decode on a Python thread pool (zlib / ctypes release the GIL): used when the native
    library is not built."""
    raw_view = out.view(np.uint8)
    _, inflate = _inflater()
    fd = os.open(path, os.O_RDONLY)

    def decode(item):
        start, mask, address, stored = item
        buf = os.pread(fd, stored, address)
        count = min(chunk, n - start)
        # filters were applied in pipeline order when writing: undo them back to front; bit k of the
        # chunk's mask means filter k was skipped for this chunk
        for k in reversed(range(len(filters))):
            if mask & (1 << k):
                continue
            if filters[k] == H5Z_FILTER_DEFLATE:
                buf = inflate(buf, chunk * size)
            else:                                   # byte shuffle: plane b holds byte b of every element
                planes = np.frombuffer(buf, dtype=np.uint8, count=chunk * size).reshape(size, chunk)
                raw_view[start * size:(start + count) * size].reshape(count, size)[:] = planes[:, :count].T
                buf = None
        if buf is not None:
            raw_view[start * size:(start + count) * size] = np.frombuffer(buf, dtype=np.uint8, count=count * size)

    try:
        from concurrent.futures import ThreadPoolExecutor
        workers = int(os.environ.get('GBRS_IO_THREADS', 0)) or min(32, len(os.sched_getaffinity(0)))
        with ThreadPoolExecutor(max_workers=max(1, workers)) as pool:
            list(pool.map(decode, todo, chunksize=4))
    finally:
        os.close(fd)
    return out


def _read_dataset(loc, name, path=None):
    lib = _load()
    d = _check(lib.H5Dopen2(loc, name.encode(), H5P_DEFAULT), f'open dataset {name}')
    try:
        ftype, space = lib.H5Dget_type(d), lib.H5Dget_space(d)
        try:
            if path is not None and not os.environ.get('GBRS_H5_SERIAL'):
                fast = _read_chunks_parallel(path, d, ftype, space)
                if fast is not None:
                    return fast
            return _read_typed(lambda o, t, p: lib.H5Dread(o, t, H5S_ALL, H5S_ALL, H5P_DEFAULT, p),
                               d, ftype, space)
        finally:
            lib.H5Tclose(ftype)
            lib.H5Sclose(space)
    finally:
        lib.H5Dclose(d)


def _read_attr(loc, name):
    lib = _load()
    if lib.H5Aexists(loc, name.encode()) <= 0:
        raise AttributeError(name)
    a = _check(lib.H5Aopen(loc, name.encode(), H5P_DEFAULT), f'open attribute {name}')
    try:
        ftype, space = lib.H5Aget_type(a), lib.H5Aget_space(a)
        try:
            return _read_typed(lambda o, t, p: lib.H5Aread(o, t, p), a, ftype, space)
        finally:
            lib.H5Tclose(ftype)
            lib.H5Sclose(space)
    finally:
        lib.H5Aclose(a)


class _AttrUnpickler(pickle.Unpickler):
    """Only what a shape tuple / name list needs: builtins plus numpy scalars.  An EMASE file is
    input data, not code."""
    _ALLOWED = {('numpy.core.multiarray', 'scalar'), ('numpy._core.multiarray', 'scalar'), ('numpy', 'dtype')}

    def find_class(self, module, name):
        if (module, name) in self._ALLOWED:
            return super().find_class(module, name)
        raise pickle.UnpicklingError(f'{module}.{name} is not allowed in an EMASE attribute')


def _unpickle_maybe(v):
    """PyTables keeps non-scalar Python attributes as pickled byte strings."""
    if isinstance(v, np.ndarray) and v.dtype.kind == 'S' and v.shape == ():
        v = v.item()
    if isinstance(v, np.ndarray) and v.dtype == object and v.shape == ():
        v = v.item()
    if isinstance(v, (bytes, np.bytes_)):
        raw = bytes(v)
        try:
            import io
            return _AttrUnpickler(io.BytesIO(raw)).load()
        except Exception:
            return raw
    return v


def _as_bool(v):
    v = _unpickle_maybe(v)
    if isinstance(v, np.ndarray):
        v = v.ravel()[0] if v.size else 0
    if isinstance(v, (bytes, np.bytes_)):
        return bytes(v).strip(b'\x00').lower() not in (b'', b'0', b'false')
    return bool(v)


def _coo_to_csc(coor, data, R, L):
    """Legacy COO component (Sparse3DMatrix.py:93-99: coo_matrix((data, coor), shape=(R, L)) followed by
    .tocsc() in finalize(), which orders by column then row and adds up duplicate coordinates)."""
    coor = np.asarray(coor)
    if coor.ndim != 2 or 2 not in coor.shape:
        raise RuntimeError('legacy EMASE file: /h*/coor must hold (row, column) pairs')
    rows, cols = (coor[0], coor[1]) if coor.shape[0] == 2 else (coor[:, 0], coor[:, 1])
    rows = rows.astype(np.int64)
    cols = cols.astype(np.int64)
    vals = np.asarray(data, dtype=np.float64)
    if len(vals) != len(rows):
        raise RuntimeError('legacy EMASE file: /h*/coor and /h*/data differ in length')
    if len(rows) and (rows.min() < 0 or rows.max() >= R or cols.min() < 0 or cols.max() >= L):
        raise ValueError('legacy EMASE file: coordinate outside the matrix shape')
    order = np.lexsort((rows, cols))
    rows, cols, vals = rows[order], cols[order], vals[order]
    if len(rows):
        first = np.concatenate(([True], (rows[1:] != rows[:-1]) | (cols[1:] != cols[:-1])))
        if not first.all():
            vals = np.add.reduceat(vals, np.flatnonzero(first))
            rows, cols = rows[first], cols[first]
    indptr = np.searchsorted(cols, np.arange(L + 1)).astype(np.uint32)
    return indptr, rows.astype(np.uint32), vals


def load_into(apm, path, on_names=None):
    """Fill an AlignmentPropertyMatrix from an EMASE h5 file (Sparse3DMatrix.py:42-102,
    AlignmentPropertyMatrix.py:70-83).  Stored values other than 1 end up in `apm.values` (they set the
    starting point of the EM: EMfactory.prepare normalises them, EMfactory.py:95-98); an incidence-only
    file, or one whose values are all 1, leaves `apm.values` None.  `on_names` (optional) is called as soon as the
    haplotype and locus names are known, before the large index arrays are decoded (the caller parses its group
    file on a thread meanwhile: the decode runs in native code without the interpreter lock)."""
    lib = _load()
    f = lib.H5Fopen(os.fsencode(path), H5F_ACC_RDONLY, H5P_DEFAULT)
    if f < 0:
        raise OSError(f'cannot open {path} as HDF5')
    try:
        root = _check(lib.H5Gopen2(f, b'/', H5P_DEFAULT), 'open /')
        try:
            # a file lacking either attribute is the legacy COO form with values (Sparse3DMatrix.py:69-78)
            try:
                mtype = _unpickle_maybe(_read_attr(root, 'mtype'))
                incidence_only = _as_bool(_read_attr(root, 'incidence_only'))
                if isinstance(mtype, (bytes, np.bytes_)):
                    mtype = bytes(mtype).decode()
                mtype = str(mtype).rstrip('\x00')
            except AttributeError:
                mtype, incidence_only = 'coo_matrix', False
            if mtype not in ('csc_matrix', 'coo_matrix'):
                raise RuntimeError('Only csc or coo matrices are supported.')
            try:
                shape = _unpickle_maybe(_read_attr(root, 'shape'))
            except AttributeError:
                raise RuntimeError(f'{path} is not an EMASE alignment file: the root attribute `shape` is missing')
            apm.shape = tuple(int(x) for x in np.asarray(shape).ravel())
            if len(apm.shape) != 3 or min(apm.shape) < 1:
                raise RuntimeError('The shape must be a tuple of three positive integers.')
            L, H, R = apm.shape
            try:
                hname = _unpickle_maybe(_read_attr(root, 'hname'))
                apm.hname = [x.decode() if isinstance(x, (bytes, np.bytes_)) else str(x)
                             for x in (hname.ravel() if isinstance(hname, np.ndarray) else hname)]
            except AttributeError:
                apm.hname = None
            if lib.H5Lexists(root, b'lname', H5P_DEFAULT) > 0:
                ln = _read_dataset(root, 'lname')
                apm.lname = ln.ravel().astype('U').tolist() if ln.dtype.kind == 'S' else \
                    [x.decode() if isinstance(x, (bytes, np.bytes_)) else str(x) for x in ln.ravel()]
            if on_names is not None:
                on_names()
            apm.indptr, apm.indices, values = [], [], []
            for h in range(H):
                g = _check(lib.H5Gopen2(f, f'/h{h}'.encode(), H5P_DEFAULT), f'open /h{h}')
                try:
                    if mtype == 'csc_matrix':
                        apm.indptr.append(np.ascontiguousarray(_read_dataset(g, 'indptr'), dtype=np.uint32))
                        apm.indices.append(np.ascontiguousarray(_read_dataset(g, 'indices', path), dtype=np.uint32))
                        values.append(None if incidence_only else
                                      np.ascontiguousarray(_read_dataset(g, 'data', path), dtype=np.float64))
                    else:
                        ip, ix, v = _coo_to_csc(_read_dataset(g, 'coor'), _read_dataset(g, 'data', path), R, L)
                        apm.indptr.append(ip)
                        apm.indices.append(ix)
                        values.append(v)
                finally:
                    lib.H5Gclose(g)
            if any(v is not None and len(v) and not (v == 1.0).all() for v in values):
                apm.values = [np.ones(len(apm.indices[h])) if v is None else v for h, v in enumerate(values)]
            if lib.H5Lexists(root, b'count', H5P_DEFAULT) > 0:
                apm.count = np.ascontiguousarray(_read_dataset(root, 'count', path), dtype=np.float64)
        finally:
            lib.H5Gclose(root)
    finally:
        lib.H5Fclose(f)


# ------------------------------------------------------------------------------------------ writer

def _write_str_attr(loc, name, value: bytes):
    lib = _load()
    t = lib.H5Tcopy(_g('H5T_C_S1_g'))
    lib.H5Tset_size(t, max(len(value), 1))
    s = lib.H5Screate(H5S_SCALAR)
    a = _check(lib.H5Acreate2(loc, name.encode(), t, s, H5P_DEFAULT, H5P_DEFAULT), f'attr {name}')
    buf = C.create_string_buffer(value, max(len(value), 1))
    lib.H5Awrite(a, t, buf)
    lib.H5Aclose(a); lib.H5Sclose(s); lib.H5Tclose(t)


def _write_scalar_attr(loc, name, value, dtype):
    lib = _load()
    arr = np.array(value, dtype=dtype)
    s = lib.H5Screate(H5S_SCALAR)
    a = _check(lib.H5Acreate2(loc, name.encode(), _native(dtype), s, H5P_DEFAULT, H5P_DEFAULT),
               f'attr {name}')
    lib.H5Awrite(a, _native(dtype), arr.ctypes.data_as(C.c_void_p))
    lib.H5Aclose(a); lib.H5Sclose(s)


def _write_carray(loc, name, arr, title='', complevel=1):
    lib = _load()
    arr = np.ascontiguousarray(arr)
    n = int(arr.shape[0])
    dims = (C.c_uint64 * 1)(n)
    space = lib.H5Screate_simple(1, dims, None)
    plist = lib.H5Pcreate(_g('H5P_CLS_DATASET_CREATE_ID_g'))
    if n > 0:
        chunk = (C.c_uint64 * 1)(min(n, 1 << 16))
        lib.H5Pset_chunk(plist, 1, chunk)
        if arr.dtype.kind != 'S':
            lib.H5Pset_shuffle(plist)           # tables.Filters(complevel=1) shuffles by default
        lib.H5Pset_deflate(plist, complevel)
    if arr.dtype.kind == 'S':
        t = lib.H5Tcopy(_g('H5T_C_S1_g'))
        lib.H5Tset_size(t, arr.dtype.itemsize)
        ftype = mtype = t
    else:
        t = None
        ftype = mtype = _native(arr.dtype)
    d = _check(lib.H5Dcreate2(loc, name.encode(), ftype, space, H5P_DEFAULT, plist, H5P_DEFAULT),
               f'create {name}')
    if n > 0:
        _check(lib.H5Dwrite(d, mtype, H5S_ALL, H5S_ALL, H5P_DEFAULT,
                            arr.ctypes.data_as(C.c_void_p)), f'write {name}')
    _write_str_attr(d, 'CLASS', b'CARRAY')
    _write_str_attr(d, 'VERSION', b'1.1')
    _write_str_attr(d, 'TITLE', title.encode())
    lib.H5Dclose(d); lib.H5Pclose(plist); lib.H5Sclose(space)
    if t is not None:
        lib.H5Tclose(t)


def _group_attrs(g, title=''):
    _write_str_attr(g, 'CLASS', b'GROUP')
    _write_str_attr(g, 'VERSION', b'1.0')
    _write_str_attr(g, 'TITLE', title.encode())


def save(apm, path, title=None, complib='zlib', incidence_only=True, shallow=False, **_ignored):
    """Write the EMASE h5 layout (Sparse3DMatrix.save :400-444 + AlignmentPropertyMatrix.save :478-525).
    incidence_only=True (the reference's default) writes the structure alone; False adds /h*/data with
    `apm.values` (ones when the matrix has none)."""
    lib = _load()
    f = lib.H5Fcreate(os.fsencode(path), H5F_ACC_TRUNC, H5P_DEFAULT, H5P_DEFAULT)
    if f < 0:
        raise OSError(f'cannot create {path}')
    try:
        root = lib.H5Gopen2(f, b'/', H5P_DEFAULT)
        _group_attrs(root, title or '')
        _write_str_attr(root, 'PYTABLES_FORMAT_VERSION', b'2.1')
        _write_scalar_attr(root, 'incidence_only', 1 if incidence_only else 0, np.int8)
        _write_str_attr(root, 'mtype', b'csc_matrix')
        _write_str_attr(root, 'shape', pickle.dumps(tuple(int(x) for x in apm.shape), 0))
        L, H, R = apm.shape
        for h in range(H):
            g = _check(lib.H5Gcreate2(f, f'/h{h}'.encode(), H5P_DEFAULT, H5P_DEFAULT, H5P_DEFAULT),
                       f'group h{h}')
            _group_attrs(g, f'Sparse matrix components for Haplotype {h}')
            _write_carray(g, 'indptr', apm.indptr[h].astype(np.uint32))
            _write_carray(g, 'indices', apm.indices[h].astype(np.uint32))
            if not incidence_only:
                vals = getattr(apm, 'values', None)
                _write_carray(g, 'data', np.ones(len(apm.indices[h])) if vals is None
                              else np.asarray(vals[h], dtype=np.float64))
            lib.H5Gclose(g)
        if apm.count is not None:
            _write_carray(root, 'count', np.asarray(apm.count, dtype=np.float64), 'Equivalence Class Counts')
        if not shallow:
            if apm.hname is not None:
                _write_str_attr(root, 'hname', pickle.dumps(list(apm.hname), 0))
            if apm.lname is not None:
                _write_carray(root, 'lname', np.array(apm.lname, dtype='S'), 'Locus Names')
        lib.H5Gclose(root)
    finally:
        lib.H5Fclose(f)
