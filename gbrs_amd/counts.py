"""`--report-alignment-counts`: alignment counts, allele-unique and locus-unique read counts per
locus or gene (emase/AlignmentPropertyMatrix.py:389-459 and _bundle_inline :155-188), computed by
libgbrs_hip (gbrs_alignment_counts).  Integer sums, exact."""
from __future__ import annotations

import numpy as np

from . import _lib


def _group_map(apm):
    """locus -> gene map and the gene names; -1 = in no gene: its entries vanish from the bundled matrix, exactly as
    the product with grp_conv_mat drops them (AlignmentPropertyMatrix.py:155-188)."""
    if not apm.num_groups:
        raise RuntimeError('No group information is available for bundling.')
    group = np.full(apm.num_loci, -1, dtype=np.int32)
    for g, members in enumerate(apm.groups):
        m = np.asarray(members, dtype=np.int64)
        if (group[m] >= 0).any():
            shared = apm.lname[int(m[group[m] >= 0][0])] if apm.lname is not None else int(m[group[m] >= 0][0])
            raise RuntimeError(f'Locus {shared} is listed in more than one group; the MI355X alignment-count '
                               'path needs every locus in at most one group.')
        group[m] = g
    return group, list(apm.gname)


class AlignmentCounter:
    """The alignments of one matrix on the device (gbrs_counts_create) for several sets of counts: `quantify -a`
    writes the isoform-level and the gene-level report from one upload."""

    def __init__(self, apm, device=0):
        import ctypes as C
        self._lib = _lib.load()
        self.apm = apm
        L, H, R = apm.shape
        cnt = None if apm.count is None else np.ascontiguousarray(apm.count, dtype=np.float64)
        self._h = C.c_void_p()
        _lib.check(self._lib.gbrs_counts_create(R, L, H, _lib.ptr_table(apm.indptr), _lib.ptr_table(apm.indices),
                                                _lib.ptr(cnt), device, C.byref(self._h)))

    def counts(self, grp_wise=False):
        """(aln (H x Lo), allele_unique (H x Lo), locus_unique (Lo), names)."""
        if self._h is None:
            raise RuntimeError('The counter has been closed.')
        apm = self.apm
        L, H, _ = apm.shape
        group, Lo, names = None, L, apm.lname
        if grp_wise:
            group, names = _group_map(apm)
            Lo = apm.num_groups
        aln = np.empty((H, Lo)); uniq = np.empty((H, Lo)); lu = np.empty(Lo)
        _lib.check(self._lib.gbrs_counts_get(self._h, _lib.ptr(group), Lo, _lib.ptr(aln), _lib.ptr(uniq), _lib.ptr(lu)))
        return aln, uniq, lu, names

    def close(self):
        if self._h is not None:
            self._lib.gbrs_counts_destroy(self._h)
            self._h = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def alignment_counts(apm, grp_wise=False, device=0):
    """Returns (aln (H x Lo), allele_unique (H x Lo), locus_unique (Lo), names)."""
    lib = _lib.load()
    L, H, R = apm.shape
    group, Lo, names = None, L, apm.lname
    if grp_wise:
        group, names = _group_map(apm)
        Lo = apm.num_groups
    aln = np.empty((H, Lo)); uniq = np.empty((H, Lo)); lu = np.empty(Lo)
    cnt = None if apm.count is None else np.ascontiguousarray(apm.count, dtype=np.float64)
    _lib.check(lib.gbrs_alignment_counts(R, L, H, _lib.ptr_table(apm.indptr), _lib.ptr_table(apm.indices),
                                         _lib.ptr(cnt), _lib.ptr(group), Lo, device,
                                         _lib.ptr(aln), _lib.ptr(uniq), _lib.ptr(lu)))
    return aln, uniq, lu, names


def report_alignment_counts(apm, filename, grp_wise=False, device=0, counter=None):
    """File format of AlignmentPropertyMatrix.report_alignment_counts (:442-459): per locus the alignment counts and
    the allele-unique counts of every haplotype, then the locus-unique count, each as str(float64).  Written by the
    library's table writer (gbrs_write_locus_table: the same digits), 2.1 M numbers per sample at DO size."""
    import os
    from .em import _blob_cached
    aln, uniq, lu, names = counter.counts(grp_wise) if counter is not None else \
        alignment_counts(apm, grp_wise=grp_wise, device=device)
    values = np.ascontiguousarray(np.vstack((aln, uniq)))          # (2H x Lo): value(row r, column c) at c * Lo + r
    totals = np.ascontiguousarray(lu, dtype=np.float64)
    names = names if isinstance(names, list) else [str(x) for x in names]
    name_blob, name_off = _blob_cached(names)
    head = 'locus\t' + '\t'.join([f'aln_{h}' for h in apm.hname]) + '\t' + \
        '\t'.join([f'uniq_{h}' for h in apm.hname]) + '\t' + 'locus_uniq' + '\n'
    n_cols, n_rows = values.shape
    _lib.check(_lib.load().gbrs_write_locus_table(os.fsencode(filename), head.encode(), _lib.ptr(values), n_rows, n_cols, 1,
                                                  n_rows, _lib.ptr(totals), name_blob, _lib.ptr(name_off), None, None, None))
