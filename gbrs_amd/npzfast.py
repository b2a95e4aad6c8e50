"""Fast readers for the `.npz` inputs of `gbrs reconstruct` / `gbrs quantify`.

`numpy.load(path)[name]` re-opens the zip member, re-parses its header and CRC-checks the payload on
every access (~0.2-0.4 ms per member): the reference pays that once per gene for avecs.npz
(gbrs/gbrs_utils.py:490, ~45 % of its run time).  An `.npz` is a plain zip of `.npy`
members, so this module reads the central directory once and then

  * maps stored (uncompressed) members straight out of the page cache (`numpy.frombuffer` on an mmap:
    no copy for the 0.5 GB transition tables, ~1 us per small member), and
  * inflates compressed members with raw zlib - large ones on a thread pool (zlib releases the GIL).

Only C-ordered, non-object `.npy` members are handled here; anything else falls back to numpy.load.
"""
from __future__ import annotations

import ast
import mmap
import os
import struct
import zipfile
import zlib

import numpy as np


class _Member:
    __slots__ = ('filename', 'compress_type', 'compress_size', 'file_size', 'header_offset', 'CRC')

    def __init__(self, filename, compress_type, compress_size, file_size, header_offset, crc=None):
        self.filename = filename
        self.compress_type = compress_type
        self.compress_size = compress_size
        self.file_size = file_size
        self.header_offset = header_offset
        self.CRC = crc                  # from the central directory; None = unknown (never checked)


# CRC-32 of the members (numpy.load checks every member on every access and raises BadZipFile: the reference inherits
# that).  Here: every member is checked once when it is read - the bulk paths (the transition tables inflated on all
# cores, the per-gene blocks copied by gbrs_npz_stack) check on the native thread that has just produced the bytes
# (libdeflate's / zlib's crc32, 0.4 GB of tables in ~15 ms spread over the cores); a mismatch sends the member to the
# one-at-a-time reader, which raises BadZipFile with the member's name.  GBRS_VERIFY_CRC=0 switches the checks of the bulk
# paths and of large stored members off (a member read on its own is always checked when it is deflated or small).
VERIFY_ALL = os.environ.get('GBRS_VERIFY_CRC', '1') not in ('', '0')
STORED_CHECK_MAX = 4 << 20


def _check_crc(zi, payload, path):
    if zi.CRC is not None and (zlib.crc32(payload) & 0xFFFFFFFF) != zi.CRC:
        raise zipfile.BadZipFile(f'Bad CRC-32 for file {zi.filename!r} in {path}')


class FastNpz:
    def __init__(self, path):
        self.path = path
        self._fh = open(path, 'rb')
        self._mm = mmap.mmap(self._fh.fileno(), 0, access=mmap.ACCESS_READ)
        self._info = {}
        self._native = None
        self._index = {}
        for k, zi in enumerate(self._central_directory()):
            name = zi.filename[:-4] if zi.filename.endswith('.npy') else zi.filename
            self._info[name] = zi
            self._index[name] = k
        self.files = list(self._info)
        self._fallback = None

    def _central_directory(self):
        """The members' (name, method, sizes, header offset) records.  zipfile.ZipFile builds a ZipInfo with
        date / attribute decoding per member (0.15 s for the 33k genes of an avecs.npz); this reads the
        same central directory with one struct call per member and falls back to zipfile for anything
        unusual (zip64 end record, multi-disk, comments)."""
        mm = self._mm
        native = self._native_directory()
        if native is not None:
            return native
        try:
            tail_at = max(0, len(mm) - 65557)
            eocd = mm.rfind(b'PK\x05\x06', tail_at)
            if eocd < 0:
                raise ValueError
            _, disk, cd_disk, n_here, n_total, cd_size, cd_off, _ = struct.unpack_from('<IHHHHIIH', mm, eocd)
            if disk or cd_disk or n_here != n_total or n_total == 0xFFFF or cd_off == 0xFFFFFFFF:
                raise ValueError
            out = []
            pos = cd_off
            unpack = struct.Struct('<IHHHHHHIIIHHHHHII').unpack_from
            for _ in range(n_total):
                (sig, _, _, flags, method, _, _, crc, csize, usize, nlen, xlen, clen, _, _, _, hoff) = unpack(mm, pos)
                if sig != 0x02014b50:
                    raise ValueError
                name = mm[pos + 46:pos + 46 + nlen].decode('utf-8' if flags & 0x800 else 'cp437')
                if csize == 0xFFFFFFFF or usize == 0xFFFFFFFF or hoff == 0xFFFFFFFF:
                    # zip64 sizes live in the extra field (numpy writes members with force_zip64)
                    extra = mm[pos + 46 + nlen:pos + 46 + nlen + xlen]
                    k = 0
                    vals = None
                    while k + 4 <= len(extra):
                        tag, ln = struct.unpack_from('<HH', extra, k)
                        if tag == 1:
                            vals = list(struct.unpack_from('<' + 'Q' * (ln // 8), extra, k + 4))
                            break
                        k += 4 + ln
                    if vals is None:
                        raise ValueError
                    if usize == 0xFFFFFFFF:
                        usize = vals.pop(0)
                    if csize == 0xFFFFFFFF:
                        csize = vals.pop(0)
                    if hoff == 0xFFFFFFFF:
                        hoff = vals.pop(0)
                zi = _Member(name, method, csize, usize, hoff, crc)
                out.append(zi)
                pos += 46 + nlen + xlen + clen
            return out
        except (ValueError, struct.error, IndexError):
            with zipfile.ZipFile(self.path) as zf:
                return [_Member(z.filename, z.compress_type, z.compress_size, z.file_size, z.header_offset, z.CRC)
                        for z in zf.infolist()]

    def _native_directory(self):
        """The same records through libgbrs_hip's gbrs_zip_directory (one pass in C: ~2 ms for the 33k genes of an
        avecs.npz against ~0.1 s for the struct loop below); None when the library is not built or declines."""
        try:
            from . import _lib
            import ctypes as C
            lib = _lib.load()
        except Exception:      # noqa: BLE001 - the pure-Python parser below needs no library
            return None
        view = np.frombuffer(self._mm, dtype=np.uint8)
        n, nbytes = C.c_uint64(0), C.c_uint64(0)
        if lib.gbrs_zip_directory(_lib.ptr(view), view.size, 0, None, None, None, None, None, None, 0, C.byref(n), C.byref(nbytes)):
            return None
        count = int(n.value)
        method = np.empty(count, dtype=np.uint16)
        csize = np.empty(count, dtype=np.uint64)
        usize = np.empty(count, dtype=np.uint64)
        hoff = np.empty(count, dtype=np.uint64)
        crc = np.empty(count, dtype=np.uint32)
        names = np.empty(int(nbytes.value), dtype=np.uint8)
        if lib.gbrs_zip_directory(_lib.ptr(view), view.size, count, _lib.ptr(method), _lib.ptr(csize), _lib.ptr(usize),
                                  _lib.ptr(hoff), _lib.ptr(crc), _lib.ptr(names), names.size, C.byref(n), C.byref(nbytes)):
            return None
        try:
            text = names.tobytes().decode('utf-8')
        except UnicodeDecodeError:
            return None                                      # cp437 names: leave them to the flag-aware parser
        labels = text.split('\n')[:-1]
        if len(labels) != count:
            return None
        self._native = (method, csize, usize, hoff, crc)     # arrays for stack() / read_many()
        crcs = crc.tolist()
        return [_Member(labels[k], int(method[k]), int(csize[k]), int(usize[k]), int(hoff[k]), crcs[k]) for k in range(count)]

    def __contains__(self, name):
        return name in self._info

    def close(self):
        try:
            self._mm.close()
        except (BufferError, ValueError):      # arrays handed out still point into the map
            pass
        self._fh.close()

    # ---- one member ------------------------------------------------------------------------------
    def _payload(self, zi):
        """(compression method, offset of the member's data in the file, stored size)."""
        o = zi.header_offset
        sig, = struct.unpack_from('<I', self._mm, o)
        if sig != 0x04034b50:
            raise ValueError(f'{self.path}: bad local header for {zi.filename}')
        nlen, xlen = struct.unpack_from('<HH', self._mm, o + 26)
        return zi.compress_type, o + 30 + nlen + xlen, zi.compress_size

    @staticmethod
    def _npy_header(buf, base=0):
        """(dtype, shape, data offset relative to base) of the .npy image starting at buf[base]."""
        if bytes(buf[base:base + 6]) != b'\x93NUMPY':
            raise ValueError('not an .npy member')
        major = buf[base + 6]
        if major == 1:
            hlen, = struct.unpack_from('<H', buf, base + 8)
            start = base + 10
        else:
            hlen, = struct.unpack_from('<I', buf, base + 8)
            start = base + 12
        meta = ast.literal_eval(bytes(buf[start:start + hlen]).decode('latin1'))
        dt = np.dtype(meta['descr'])
        if meta['fortran_order'] or dt.hasobject:
            raise ValueError('unsupported .npy layout')
        return dt, tuple(meta['shape']), start + hlen - base

    def _numpy_load(self, name):
        if self._fallback is None:
            self._fallback = np.load(self.path, allow_pickle=False)
        return self._fallback[name]

    def __getitem__(self, name):
        zi = self._info[name]
        try:
            method, off, csize = self._payload(zi)
            if method == zipfile.ZIP_STORED:
                dt, shape, doff = self._npy_header(self._mm, off)
                small = csize <= STORED_CHECK_MAX
                if small or VERIFY_ALL:
                    _check_crc(zi, memoryview(self._mm)[off:off + csize], self.path)
                a = np.frombuffer(self._mm, dtype=dt, count=int(np.prod(shape, dtype=np.int64)),
                                  offset=off + doff).reshape(shape)
                # small members are handed out as arrays of their own (aligned, writable, as numpy.load's are); the
                # large transition tables stay views of the page cache
                return a.copy() if small else a
            if method == zipfile.ZIP_DEFLATED:
                raw = zlib.decompress(self._mm[off:off + csize], -15, zi.file_size)
                _check_crc(zi, raw, self.path)
                dt, shape, doff = self._npy_header(raw)
                return np.frombuffer(bytearray(raw), dtype=dt, count=int(np.prod(shape, dtype=np.int64)), offset=doff).reshape(shape)
        except ValueError:
            pass
        return self._numpy_load(name)

    # ---- many members ----------------------------------------------------------------------------
    def read_many(self, names, threads=None):
        """The members `names` as a list of arrays; large compressed ones are inflated in parallel."""
        names = list(names)
        big = sum(self._info[n].file_size for n in names) > (8 << 20) and \
            any(self._info[n].compress_type != zipfile.ZIP_STORED for n in names)
        if not big or len(names) < 2:
            return [self[n] for n in names]
        native = self._read_many_native(names)
        if native is not None:
            return native
        from concurrent.futures import ThreadPoolExecutor
        workers = threads or int(os.environ.get('GBRS_IO_THREADS', 0)) or min(32, len(os.sched_getaffinity(0)))
        with ThreadPoolExecutor(max_workers=max(1, min(workers, len(names)))) as pool:
            return list(pool.map(self.__getitem__, names))

    def _read_many_native(self, names):
        """Deflated members through gbrs_zip_read_members (libdeflate when the machine has it, all cores, largest
        member first); None when the library or a member declines."""
        if self._native is None:
            return None
        try:
            import ctypes as C
            from . import _lib
            lib = _lib.load()
            which = np.fromiter((self._index[n] for n in names), dtype=np.int64, count=len(names))
            m_all, c_all, u_all, h_all, crc_all = self._native
            method, csize, usize, hoff, crc = (np.ascontiguousarray(a[which]) for a in (m_all, c_all, u_all, h_all, crc_all))
            if not np.isin(method, (0, 8)).all():
                return None
            images = [np.empty(int(u), dtype=np.uint8) for u in usize]
            ptrs = (C.c_void_p * len(names))(*[im.ctypes.data for im in images])
            view = np.frombuffer(self._mm, dtype=np.uint8)
            # (a member that fails its CRC-32 makes the call decline: the caller's one-at-a-time path then names it)
            if lib.gbrs_zip_read_members(_lib.ptr(view), view.size, len(names), _lib.ptr(hoff), _lib.ptr(method),
                                         _lib.ptr(csize), _lib.ptr(usize), _lib.ptr(crc) if VERIFY_ALL else None, ptrs, 0):
                return None
            out = []
            for im in images:
                dt, shape, doff = self._npy_header(im)
                out.append(np.frombuffer(im, dtype=dt, count=int(np.prod(shape, dtype=np.int64)), offset=doff).reshape(shape))
            return out
        except (OSError, AttributeError, KeyError, ValueError):
            return None

    def stack(self, names, shape, dtype=np.float64):
        """Equally shaped small members (the per-gene blocks of avecs.npz) as one [len(names), *shape]
        array.  Members whose .npy header equals the first one's (same dtype, shape, order - the normal
        case) are copied as raw bytes without parsing their header again."""
        want = tuple(int(x) for x in shape)
        out = np.empty((len(names),) + want, dtype=dtype)
        if not len(names):
            return out
        if self._native is not None and self._stack_native(names, want, out):
            return out
        nbytes = out[0].nbytes
        flat = memoryview(out.reshape(-1).view(np.uint8))
        mm, info, unpack = self._mm, self._info, struct.unpack_from
        ref_hdr = None
        for k, n in enumerate(names):
            zi = info[n]
            o = zi.header_offset
            nlen, xlen = unpack('<HH', mm, o + 26)
            ds = o + 30 + nlen + xlen
            if zi.compress_type == zipfile.ZIP_STORED:
                buf, base = mm, ds
            elif zi.compress_type == zipfile.ZIP_DEFLATED:
                buf, base = zlib.decompress(mm[ds:ds + zi.compress_size], -15, zi.file_size), 0
            else:
                buf = None
            if buf is not None and ref_hdr is not None and buf[base:base + len(ref_hdr)] == ref_hdr:
                start = base + len(ref_hdr)
                flat[k * nbytes:(k + 1) * nbytes] = buf[start:start + nbytes]
                continue
            a = self[n]                                   # first member, or one with a header of its own
            if a.shape != want:
                raise ValueError(f'{self.path}: member {n} has shape {a.shape}, expected {want}')
            out[k] = a
            if ref_hdr is None and buf is not None and a.dtype == out.dtype:
                try:
                    _, _, doff = self._npy_header(buf, base)
                    ref_hdr = bytes(buf[base:base + doff])
                except ValueError:
                    pass
        return out


    def _stack_native_impl(self, names, want, out):
        from . import _lib
        first = self[names[0]]
        if first.shape != want or first.dtype != out.dtype:
            return False
        zi = self._info[names[0]]
        method, off, csize = self._payload(zi)
        image = self._mm[off:off + csize] if method == zipfile.ZIP_STORED else zlib.decompress(self._mm[off:off + csize], -15, zi.file_size)
        _, _, doff = self._npy_header(image)
        header = np.frombuffer(bytes(image[:doff]), dtype=np.uint8)
        which = np.fromiter((self._index[n] for n in names), dtype=np.int64, count=len(names))
        m_all, c_all, u_all, h_all, crc_all = self._native
        crc_k = np.ascontiguousarray(crc_all[which])
        method_k = np.ascontiguousarray(m_all[which])
        csize_k = np.ascontiguousarray(c_all[which])
        usize_k = np.ascontiguousarray(u_all[which])
        hoff_k = np.ascontiguousarray(h_all[which])
        fallback = np.empty(len(names), dtype=np.uint8)
        view = np.frombuffer(self._mm, dtype=np.uint8)
        flat = out.reshape(len(names), -1).view(np.uint8)
        status = _lib.load().gbrs_npz_stack(_lib.ptr(view), view.size, len(names), _lib.ptr(hoff_k), _lib.ptr(method_k),
                                            _lib.ptr(csize_k), _lib.ptr(usize_k), _lib.ptr(crc_k) if VERIFY_ALL else None,
                                            _lib.ptr(header), header.size,
                                            out[0].nbytes, _lib.ptr(flat), _lib.ptr(fallback), 0)
        if status:
            return False
        for k in np.flatnonzero(fallback):                   # a member with a header of its own
            a = self[names[k]]
            if a.shape != want:
                raise ValueError(f'{self.path}: member {names[k]} has shape {a.shape}, expected {want}')
            out[k] = a
        return True

    def _stack_native(self, names, want, out):
        try:
            return self._stack_native_impl(names, want, out)
        except (OSError, AttributeError, KeyError):
            return False


def savez_compressed(path, arrays, level=6, threads=None):
    """numpy.savez_compressed with the members deflated on a thread pool (zlib releases the GIL) and the zip
    container written by hand: local header + raw deflate stream per member, central directory, end
    record.  The result is an ordinary `.npz` that numpy.load reads."""
    import io
    from concurrent.futures import ThreadPoolExecutor
    path = str(path)
    if not path.endswith('.npz'):
        path += '.npz'

    def pack(item):
        name, arr = item
        bio = io.BytesIO()
        np.lib.format.write_array(bio, np.asanyarray(arr), allow_pickle=False)
        raw = bio.getvalue()
        comp = zlib.compressobj(level, zlib.DEFLATED, -15)
        data = comp.compress(raw) + comp.flush()
        return (name + '.npy').encode(), zlib.crc32(raw), len(raw), data

    items = list(arrays.items())
    workers = threads or int(os.environ.get('GBRS_IO_THREADS', 0)) or min(32, len(os.sched_getaffinity(0)))
    if len(items) > 1 and workers > 1:
        with ThreadPoolExecutor(max_workers=min(workers, len(items))) as pool:
            packed = list(pool.map(pack, items))
    else:
        packed = [pack(it) for it in items]
    central = []
    with open(path, 'wb') as fh:
        for fname, crc, usize, data in packed:
            off = fh.tell()
            if max(usize, len(data), off) >= 0xFFFFFFFF:
                raise ValueError('member too large for the plain zip format; use numpy.savez_compressed')
            fh.write(struct.pack('<IHHHHHIIIHH', 0x04034b50, 20, 0, 8, 0, 0x21, crc, len(data), usize, len(fname), 0))
            fh.write(fname)
            fh.write(data)
            central.append(struct.pack('<IHHHHHHIIIHHHHHII', 0x02014b50, 20, 20, 0, 8, 0, 0x21, crc, len(data), usize,
                                       len(fname), 0, 0, 0, 0, 0, off) + fname)
        cd_off = fh.tell()
        for rec in central:
            fh.write(rec)
        cd_size = fh.tell() - cd_off
        if len(central) > 0xFFFF:
            raise ValueError('too many members for the plain zip format')
        fh.write(struct.pack('<IHHHHIIH', 0x06054b50, 0, 0, len(central), len(central), cd_size, cd_off, 0))
