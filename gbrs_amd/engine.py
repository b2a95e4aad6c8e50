"""Thin object wrapper over the gbrs_em_* C ABI (include/gbrs_hip.h).  No arithmetic here."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib


def _allowed_array(allowed, L):
    al = np.ascontiguousarray(allowed, dtype=np.uint32)
    if al.shape != (L,):
        raise RuntimeError('allowed must hold one haplotype bit mask per locus')
    return al


class EmEngine:
    def __init__(self, handle, shape, keepalive=None):
        self._h = handle
        self.shape = shape            # (L, H, R)
        self._keep = keepalive

    # ---- construction ----------------------------------------------------------------------
    @classmethod
    def from_host(cls, R, L, H, indptr, indices, count=None, eff_len=None, device=0, flags=0, allowed=None):
        lib = _lib.load()
        ip = [np.ascontiguousarray(p, dtype=np.uint32) for p in indptr]
        ix = [np.ascontiguousarray(i, dtype=np.uint32) for i in indices]
        for h in range(H):
            if len(ip[h]) != L + 1 or int(ip[h][-1]) != len(ix[h]):
                raise RuntimeError(f'Malformed CSC arrays for haplotype {h}.')
        cnt = None if count is None else np.ascontiguousarray(count, dtype=np.float64)
        eff = None if eff_len is None else np.ascontiguousarray(eff_len, dtype=np.float64)
        if cnt is not None and len(cnt) != R:
            raise RuntimeError('count has the wrong length')
        if eff is not None and eff.shape != (H, L):
            raise RuntimeError('eff_len must be (H x L)')
        h = C.c_void_p()
        if allowed is not None:         # `-G`: uint32[L] haplotype bits per locus, applied on the device
            al = _allowed_array(allowed, L)
            _lib.check(lib.gbrs_em_create_masked(R, L, H, _lib.ptr_table(ip), _lib.ptr_table(ix), _lib.ptr(cnt),
                                                 _lib.ptr(eff), _lib.ptr(al), device, flags, C.byref(h)))
        else:
            _lib.check(lib.gbrs_em_create(R, L, H, _lib.ptr_table(ip), _lib.ptr_table(ix), _lib.ptr(cnt),
                                          _lib.ptr(eff), device, flags, C.byref(h)))
        return cls(h, (L, H, R))

    @classmethod
    def from_device(cls, R, L, H, indptr_ptrs, indices_ptrs, count_ptr=None, eff_len_ptr=None,
                    device=0, flags=0, allowed=None):
        """All array arguments are raw device addresses (ints); the arrays are only read during the call.
        `allowed` (the `-G` mask) is a host uint32[L] array."""
        lib = _lib.load()
        h = C.c_void_p()
        if allowed is not None:
            al = _allowed_array(allowed, L)
            _lib.check(lib.gbrs_em_create_masked_device(R, L, H, _lib.raw_table(indptr_ptrs),
                                                        _lib.raw_table(indices_ptrs), count_ptr, eff_len_ptr,
                                                        _lib.ptr(al), device, flags, C.byref(h)))
        else:
            _lib.check(lib.gbrs_em_create_device(R, L, H, _lib.raw_table(indptr_ptrs),
                                                 _lib.raw_table(indices_ptrs), count_ptr, eff_len_ptr,
                                                 device, flags, C.byref(h)))
        return cls(h, (L, H, R))

    def set_initial_values(self, values):
        """Per-entry stored values (aligned with the indices given to from_host, which must have been
        called with GBRS_EM_KEEP_CSC): prepare() then starts from their per-read normalisation."""
        vals = [np.ascontiguousarray(v, dtype=np.float64) for v in values]
        _lib.check(_lib.load().gbrs_em_set_initial_values(self._h, _lib.ptr_table(vals)))

    def close(self):
        if self._h is not None:
            _lib.load().gbrs_em_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- calls -----------------------------------------------------------------------------
    def prepare(self, pseudocount=0.0):
        _lib.check(_lib.load().gbrs_em_prepare(self._h, float(pseudocount)))

    def step(self, n=1, want_err=False):
        err = C.c_double(0.0)
        _lib.check(_lib.load().gbrs_em_step(self._h, int(n), C.byref(err) if want_err else None))
        return err.value

    def run(self, model=4, tol=0.001, max_iters=999):
        n_it = C.c_int(0)
        cap = max(int(max_iters), 1)
        hist = np.zeros(cap, dtype=np.float64)
        _lib.check(_lib.load().gbrs_em_run(self._h, int(model), float(tol), int(max_iters), C.byref(n_it),
                                           _lib.ptr(hist), cap, None))
        return int(n_it.value), hist[:n_it.value].copy()

    def theta(self):
        L, H, R = self.shape
        out = np.empty((H, L), dtype=np.float64)
        _lib.check(_lib.load().gbrs_em_get(self._h, _lib.ptr(out), None))
        return out

    def expected_counts(self):
        L, H, R = self.shape
        out = np.empty((H, L), dtype=np.float64)
        _lib.check(_lib.load().gbrs_em_get(self._h, None, _lib.ptr(out)))
        return out

    def set_theta(self, theta):
        L, H, R = self.shape
        t = np.ascontiguousarray(theta, dtype=np.float64).reshape(H, L)
        _lib.check(_lib.load().gbrs_em_set_theta(self._h, _lib.ptr(t)))

    def group_sums(self, group_ptr, members, which=0):
        L, H, R = self.shape
        gp = np.ascontiguousarray(group_ptr, dtype=np.int64)
        mem = np.ascontiguousarray(members, dtype=np.int64)
        G = len(gp) - 1
        out = np.empty((H, G), dtype=np.float64)
        _lib.check(_lib.load().gbrs_em_group_sums(self._h, G, _lib.ptr(gp), _lib.ptr(mem), which, _lib.ptr(out)))
        return out

    def info(self):
        inf = _lib.EmInfo()
        _lib.check(_lib.load().gbrs_em_info(self._h, C.byref(inf)))
        return inf

    # ---- multi-GPU building blocks -----------------------------------------------------------
    def _partial(self, fn):
        p = C.c_void_p()
        n = C.c_uint64(0)
        _lib.check(fn(self._h, C.byref(p), C.byref(n)))
        return p.value, int(n.value)

    def estep_partial(self):
        return self._partial(_lib.load().gbrs_em_estep_partial)

    def prepare_partial(self):
        return self._partial(_lib.load().gbrs_em_prepare_partial)

    def finish_step(self, want_err=True):
        err = C.c_double(0.0)
        _lib.check(_lib.load().gbrs_em_finish_step(self._h, C.byref(err) if want_err else None))
        return err.value

    def finish_prepare(self, pseudocount=0.0):
        _lib.check(_lib.load().gbrs_em_finish_prepare(self._h, float(pseudocount)))

    def sync(self):
        _lib.check(_lib.load().gbrs_em_sync(self._h))

    # ---- the stopping rule over two handles (the two locus ranges of one sample) ---------------
    def pair_begin(self, other, max_iters):
        _lib.check(_lib.load().gbrs_em_pair_begin(self._h, other._h, int(max_iters)))

    def pair_check(self, other, tol):
        _lib.check(_lib.load().gbrs_em_pair_check(self._h, other._h, float(tol)))

    def pair_status(self, other, cap):
        """(iterations applied, stopped, err_sum sequence) - synchronises both handles."""
        it, st = C.c_int(0), C.c_int(0)
        hist = np.zeros(max(int(cap), 1), dtype=np.float64)
        _lib.check(_lib.load().gbrs_em_pair_status(self._h, other._h, C.byref(it), C.byref(st), _lib.ptr(hist), len(hist)))
        return int(it.value), bool(st.value), hist[:min(int(it.value), len(hist))].copy()

    def stream(self):
        return _lib.load().gbrs_em_stream(self._h)

    def set_stream(self, stream_ptr):
        _lib.check(_lib.load().gbrs_em_set_stream(self._h, stream_ptr))
