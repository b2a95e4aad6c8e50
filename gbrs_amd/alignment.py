"""Host-side container for the alignment incidence tensor.

Mirror of the parts of the reference's AlignmentPropertyMatrix that `gbrs quantify` touches
(emase/AlignmentPropertyMatrix.py:28-130, emase/Sparse3DMatrix.py:26-102): shape (L, H, R), one
CSC (R x L) incidence matrix per haplotype, optional EC ``count``, haplotype / locus names and
gene groups.  Only the *structure* is kept (``incidence_only``): the values of the reference's
matrices are reset to 1 at the top of every EM step (Sparse3DMatrix.py:220-228), and the device
never materialises them.

File formats
  * ``.h5``  EMASE/PyTables layout (gbrs_amd.emase_h5, needs libhdf5)
  * ``.npz`` mirror of the same fields for machines without libhdf5:
        shape=(L,H,R)  hname  lname  [count]  indptr{h}  indices{h}  [values{h}]

``values`` (normally None) holds stored alignment values other than 1, which only legacy files and
files saved with ``incidence_only=False`` carry; they set the starting point of the EM
(EMfactory.prepare normalises them per read, EMfactory.py:95-98) and nothing else.
"""
from __future__ import annotations

import numpy as np


class AlignmentPropertyMatrix:
    def __init__(self, shape=None, indptr=None, indices=None, count=None, haplotype_names=None,
                 locus_names=None, grpfile=None, h5file=None, npzfile=None, values=None, on_names=None):
        """`on_names(self)` (optional) is called once the haplotype and locus names are known - for an HDF5 file that is
        before the large index arrays are decoded, so the caller can parse name-keyed side files meanwhile."""
        self.hname = None
        self.lname = None
        self.lid = None
        self.gname = None
        self.groups = None
        self.num_groups = 0
        self.count = None
        self.values = None          # per-haplotype float64 arrays aligned with indices, or None = all ones
        self.haplotype_mask = None  # uint32[L], bit h = (h, l) kept: a `-G` mask the device applies (set_haplotype_mask)
        groups_thread = None
        if h5file is not None:
            from . import emase_h5
            if grpfile is not None or on_names is not None:
                # the group file is parsed while the index arrays are decoded (native code, no interpreter lock)
                import threading
                box = {}

                def parse_groups():
                    try:
                        if self.lname is None:
                            raise RuntimeError('Locus IDs are not available.')
                        self.load_groups(grpfile)
                    except BaseException as ex:      # noqa: BLE001 - re-raised on the calling thread below
                        box['error'] = ex

                def start():
                    nonlocal groups_thread
                    self.num_loci, self.num_haplotypes, self.num_reads = self.shape
                    if self.lname is not None:       # before anything that runs beside the decode looks a name up
                        self.lid = dict(zip(self.lname, np.arange(len(self.lname))))
                    if on_names is not None:
                        on_names(self)
                    if grpfile is not None:
                        groups_thread = threading.Thread(target=parse_groups, name='gbrs-groups')
                        groups_thread.start()
                try:
                    emase_h5.load_into(self, h5file, on_names=start)
                finally:
                    if groups_thread is not None:
                        groups_thread.join()
                if 'error' in box:
                    raise box['error']
            else:
                emase_h5.load_into(self, h5file)
        elif npzfile is not None:
            self._load_npz(npzfile)
            if on_names is not None:
                self.num_loci, self.num_haplotypes, self.num_reads = self.shape
                if self.lname is not None:
                    self.lid = dict(zip(self.lname, np.arange(len(self.lname))))
                on_names(self)
        else:
            if shape is None or len(shape) != 3 or (np.array(shape) < 1).any():
                raise RuntimeError('The shape must be a tuple of three positive integers.')
            self.shape = tuple(int(x) for x in shape)
            L, H, R = self.shape
            if indptr is None:
                indptr = [np.zeros(L + 1, dtype=np.uint32) for _ in range(H)]
                indices = [np.zeros(0, dtype=np.uint32) for _ in range(H)]
            self.indptr = [np.ascontiguousarray(p, dtype=np.uint32) for p in indptr]
            self.indices = [np.ascontiguousarray(i, dtype=np.uint32) for i in indices]
            if count is not None:
                self.count = np.ascontiguousarray(count, dtype=np.float64)
            if values is not None:
                self.values = [np.ascontiguousarray(v, dtype=np.float64) for v in values]
            if haplotype_names is not None:
                if len(haplotype_names) != H:
                    raise RuntimeError('The number of names does not match to the matrix shape.')
                self.hname = list(haplotype_names)
            if locus_names is not None:
                if len(locus_names) != L:
                    raise RuntimeError('The number of names does not match to the matrix shape.')
                self.lname = list(locus_names)
        self._finish_init()
        if grpfile is not None and groups_thread is None:
            self.load_groups(grpfile)

    def _finish_init(self):
        self.num_loci, self.num_haplotypes, self.num_reads = self.shape
        L, H, R = self.shape
        if len(self.indptr) != H or len(self.indices) != H:
            raise RuntimeError('The number of haplotype matrices does not match to the matrix shape.')
        for h in range(H):
            if len(self.indptr[h]) != L + 1 or int(self.indptr[h][-1]) != len(self.indices[h]):
                raise RuntimeError(f'Malformed CSC arrays for haplotype {h}.')
        if self.count is not None and len(self.count) != R:
            raise RuntimeError('The length of count does not match to the matrix shape.')
        if self.values is not None and [len(v) for v in self.values] != [len(i) for i in self.indices]:
            raise RuntimeError('The stored values do not match the index arrays.')
        if self.lname is not None and (self.lid is None or len(self.lid) != self.num_loci):
            self.lid = dict(zip(self.lname, np.arange(self.num_loci)))
        self.finalized = True

    # ---- groups (AlignmentPropertyMatrix.py:113-130) ---------------------------------------
    def load_groups(self, grpfile):
        if self.lid is None:
            raise RuntimeError('Locus IDs are not available.')
        self.gname = []
        self.groups = []
        with open(grpfile) as fh:
            for curline in fh:
                item = curline.rstrip().split('\t')
                self.gname.append(item[0])
                self.groups.append([self.lid[t] for t in item[1:]])
        self.gname = np.array(self.gname)
        self.num_groups = len(self.gname)
        # while the caller's thread still decodes the index arrays: the reports and `-G` need both
        self.group_csr()
        from .em import _blob_cached
        _blob_cached(self.gname)

    def group_csr(self):
        """(group_ptr int64[G+1], members int64[...]) with members ascending and unique per group:
        the column structure of the reference's grp_conv_mat (EMfactory.py:41-47)."""
        cached = getattr(self, '_group_csr', None)
        if cached is not None and cached[0] is self.groups:
            return cached[1]
        sizes = np.fromiter(map(len, self.groups), dtype=np.int64, count=len(self.groups))
        gene = np.repeat(np.arange(len(self.groups), dtype=np.int64), sizes)
        mem = np.fromiter((x for g in self.groups for x in g), dtype=np.int64, count=int(sizes.sum()))
        order = np.lexsort((mem, gene))                      # ascending members inside every group ...
        gene, mem = gene[order], mem[order]
        if len(mem):
            keep = np.concatenate(([True], (gene[1:] != gene[:-1]) | (mem[1:] != mem[:-1])))   # ... without repeats
            gene, mem = gene[keep], mem[keep]
        ptr = np.searchsorted(gene, np.arange(len(self.groups) + 1)).astype(np.int64)
        self._group_csr = (self.groups, (ptr, np.ascontiguousarray(mem)))
        return self._group_csr[1]

    # ---- structure edits -------------------------------------------------------------------
    def set_haplotype_mask(self, allowed):
        """The `-G` restriction of `gbrs quantify` (gbrs/emase_utils.py:271-273: `multiply(gtmask, axis=2)` +
        `eliminate_zeros()`) as a per-locus bit mask, uint32[L] with bit h set where (haplotype h, locus l) stays.
        The host arrays are left alone: gbrs_em_create_masked drops the columns on the device.  Anything that
        needs the edited structure on the host (save, nnz) calls apply_haplotype_mask() first."""
        allowed = np.ascontiguousarray(allowed, dtype=np.uint32)
        if allowed.shape != (self.num_loci,):
            raise RuntimeError('The haplotype mask does not match to the matrix shape.')
        self.haplotype_mask = allowed if self.haplotype_mask is None else (self.haplotype_mask & allowed)

    def apply_haplotype_mask(self):
        """Carry a pending device-side mask out on the host arrays (the reference's eager behaviour)."""
        if self.haplotype_mask is None:
            return
        allowed, self.haplotype_mask = self.haplotype_mask, None
        H = self.num_haplotypes
        self.mask_haplotype_loci(((allowed[None, :] >> np.arange(H, dtype=np.uint32)[:, None]) & 1).astype(np.float64))

    def mask_haplotype_loci(self, gtmask):
        """Drop every entry (h, l) with gtmask[h, l] == 0 from the host arrays: `multiply(gtmask, axis=2)` followed
        by `eliminate_zeros()` per haplotype (gbrs/emase_utils.py:271-273)."""
        gtmask = np.asarray(gtmask)
        L, H, R = self.shape
        for h in range(H):
            width = np.diff(self.indptr[h].astype(np.int64))
            keep_col = gtmask[h, :] != 0.0
            keep = np.repeat(keep_col, width)
            self.indices[h] = np.ascontiguousarray(self.indices[h][keep])
            if self.values is not None:
                self.values[h] = np.ascontiguousarray(self.values[h][keep])
            self.indptr[h] = np.concatenate(([0], np.cumsum(np.where(keep_col, width, 0)))).astype(np.uint32)

    @property
    def nnz(self):
        self.apply_haplotype_mask()
        return int(sum(len(i) for i in self.indices))

    # ---- npz mirror ------------------------------------------------------------------------
    def _load_npz(self, path):
        from .npzfast import FastNpz
        z = FastNpz(path)                                   # members inflated in parallel (zlib drops the GIL)
        try:
            self.shape = tuple(int(x) for x in z['shape'])
            L, H, R = self.shape
            want = [f'indptr{h}' for h in range(H)] + [f'indices{h}' for h in range(H)]
            want += ['count'] if 'count' in z else []
            want += [f'values{h}' for h in range(H)] if 'values0' in z else []
            got = dict(zip(want, z.read_many(want)))
            self.indptr = [np.ascontiguousarray(got[f'indptr{h}'], dtype=np.uint32) for h in range(H)]
            self.indices = [np.ascontiguousarray(got[f'indices{h}'], dtype=np.uint32) for h in range(H)]
            if 'count' in got:
                self.count = np.ascontiguousarray(got['count'], dtype=np.float64)
            if 'values0' in got:
                self.values = [np.ascontiguousarray(got[f'values{h}'], dtype=np.float64) for h in range(H)]
            if 'hname' in z:
                self.hname = z['hname'].astype('U').tolist()
            if 'lname' in z:
                self.lname = z['lname'].astype('U').tolist()
        finally:
            z.close()

    def save_npz(self, path):
        self.apply_haplotype_mask()
        out = dict(shape=np.asarray(self.shape, dtype=np.int64))
        for h in range(self.num_haplotypes):
            out[f'indptr{h}'] = self.indptr[h]
            out[f'indices{h}'] = self.indices[h]
        if self.count is not None:
            out['count'] = self.count
        if self.values is not None:
            for h in range(self.num_haplotypes):
                out[f'values{h}'] = self.values[h]
        if self.hname is not None:
            out['hname'] = np.array(self.hname)
        if self.lname is not None:
            out['lname'] = np.array(self.lname)
        np.savez_compressed(path, **out)

    def save(self, h5file, **kw):
        if str(h5file).endswith('.npz'):
            return self.save_npz(h5file)
        from . import emase_h5
        self.apply_haplotype_mask()
        emase_h5.save(self, h5file, **kw)


def load_alignment(path, grpfile=None, on_names=None):
    """Open an EMASE alignment file by extension (.npz mirror or PyTables-layout HDF5)."""
    if str(path).endswith('.npz'):
        return AlignmentPropertyMatrix(npzfile=path, grpfile=grpfile, on_names=on_names)
    return AlignmentPropertyMatrix(h5file=path, grpfile=grpfile, on_names=on_names)
