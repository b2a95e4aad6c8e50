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
        shape=(L,H,R)  hname  lname  [count]  indptr{h}  indices{h}
"""
from __future__ import annotations

import numpy as np


class AlignmentPropertyMatrix:
    def __init__(self, shape=None, indptr=None, indices=None, count=None, haplotype_names=None,
                 locus_names=None, grpfile=None, h5file=None, npzfile=None):
        self.hname = None
        self.lname = None
        self.lid = None
        self.gname = None
        self.groups = None
        self.num_groups = 0
        self.count = None
        if h5file is not None:
            from . import emase_h5
            emase_h5.load_into(self, h5file)
        elif npzfile is not None:
            self._load_npz(npzfile)
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
            if haplotype_names is not None:
                if len(haplotype_names) != H:
                    raise RuntimeError('The number of names does not match to the matrix shape.')
                self.hname = list(haplotype_names)
            if locus_names is not None:
                if len(locus_names) != L:
                    raise RuntimeError('The number of names does not match to the matrix shape.')
                self.lname = list(locus_names)
        self._finish_init()
        if grpfile is not None:
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
        if self.lname is not None:
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

    def group_csr(self):
        """(group_ptr int64[G+1], members int64[...]) with members ascending and unique per group:
        the column structure of the reference's grp_conv_mat (EMfactory.py:41-47)."""
        ptr = [0]
        mem = []
        for g in self.groups:
            m = sorted(set(int(x) for x in g))
            mem.extend(m)
            ptr.append(len(mem))
        return np.asarray(ptr, dtype=np.int64), np.asarray(mem, dtype=np.int64)

    # ---- structure edits -------------------------------------------------------------------
    def mask_haplotype_loci(self, gtmask):
        """Drop every entry (h, l) with gtmask[h, l] == 0: `multiply(gtmask, axis=2)` followed by
        `eliminate_zeros()` per haplotype (gbrs/emase_utils.py:271-273)."""
        gtmask = np.asarray(gtmask)
        L, H, R = self.shape
        for h in range(H):
            width = np.diff(self.indptr[h].astype(np.int64))
            keep_col = gtmask[h, :] != 0.0
            self.indices[h] = np.ascontiguousarray(self.indices[h][np.repeat(keep_col, width)])
            self.indptr[h] = np.concatenate(([0], np.cumsum(np.where(keep_col, width, 0)))).astype(np.uint32)

    @property
    def nnz(self):
        return int(sum(len(i) for i in self.indices))

    # ---- npz mirror ------------------------------------------------------------------------
    def _load_npz(self, path):
        with np.load(path, allow_pickle=False) as z:
            self.shape = tuple(int(x) for x in z['shape'])
            L, H, R = self.shape
            self.indptr = [np.ascontiguousarray(z[f'indptr{h}'], dtype=np.uint32) for h in range(H)]
            self.indices = [np.ascontiguousarray(z[f'indices{h}'], dtype=np.uint32) for h in range(H)]
            if 'count' in z.files:
                self.count = np.ascontiguousarray(z['count'], dtype=np.float64)
            if 'hname' in z.files:
                self.hname = [str(x) for x in z['hname']]
            if 'lname' in z.files:
                self.lname = [str(x) for x in z['lname']]

    def save_npz(self, path):
        out = dict(shape=np.asarray(self.shape, dtype=np.int64))
        for h in range(self.num_haplotypes):
            out[f'indptr{h}'] = self.indptr[h]
            out[f'indices{h}'] = self.indices[h]
        if self.count is not None:
            out['count'] = self.count
        if self.hname is not None:
            out['hname'] = np.array(self.hname)
        if self.lname is not None:
            out['lname'] = np.array(self.lname)
        np.savez_compressed(path, **out)

    def save(self, h5file, **kw):
        if str(h5file).endswith('.npz'):
            return self.save_npz(h5file)
        from . import emase_h5
        emase_h5.save(self, h5file, **kw)


def load_alignment(path, grpfile=None):
    """Open an EMASE alignment file by extension (.npz mirror or PyTables-layout HDF5)."""
    if str(path).endswith('.npz'):
        return AlignmentPropertyMatrix(npzfile=path, grpfile=grpfile)
    return AlignmentPropertyMatrix(h5file=path, grpfile=grpfile)
