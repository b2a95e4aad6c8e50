#!/usr/bin/env python3
"""Write the three tiny EMASE .h5 fixtures that pin gbrs_amd/emase_h5.py against files PyTables itself produced.

Needs PyTables (`pip install tables`), which the build image of this repository does not have; run it once wherever
PyTables is installed and commit the three files it leaves under tests/golden/ - tests/test_emase_h5.py then loads them
(and reports the tests as skipped, with this script's name, while they are absent).  Pure PyTables + numpy, no GBRS import.
The calls are the ones the reference's writers make: emase/Sparse3DMatrix.py:400-444 + AlignmentPropertyMatrix.py:478-525
(`save`), AlignmentMatrixFactory.py:84-142 (`produce`), and the legacy COO layout Sparse3DMatrix.py:93-99 reads back."""
import os
import numpy as np
import tables

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
L, H, R = 5, 2, 7                                   # loci, haplotypes, reads
rng = np.random.default_rng(20241008)
dense = [(rng.random((R, L)) < 0.4) * np.round(rng.random((R, L)) + 0.5, 3) for _ in range(H)]     # (R x L) per haplotype
hname, lname = ["A", "B"], [f"ENSMUST{l:011d}" for l in range(L)]
count = rng.integers(1, 9, size=R).astype(float)


def csc(m):                                         # column-major (data, row indices, column pointers) of a dense matrix
    rows, cols = np.nonzero(m.T)[1], np.nonzero(m.T)[0]
    return m[rows, cols], rows, np.concatenate(([0], np.cumsum(np.bincount(cols, minlength=m.shape[1]))))


def write(path, incidence_only, legacy_coo=False):
    fh = tables.open_file(path, "w", title="pytables fixture")
    fil = tables.Filters(complevel=1, complib="zlib")
    if not legacy_coo:
        fh.set_node_attr(fh.root, "incidence_only", incidence_only)
        fh.set_node_attr(fh.root, "mtype", "csc_matrix")
    fh.set_node_attr(fh.root, "shape", (L, H, R))
    fh.set_node_attr(fh.root, "hname", hname)
    fh.create_carray(fh.root, "lname", obj=lname, title="Locus Names", filters=fil)
    fh.create_carray(fh.root, "count", obj=count, title="Equivalence Class Counts", filters=fil)
    for h in range(H):
        grp = fh.create_group(fh.root, f"h{h}", f"Sparse matrix components for Haplotype {h}")
        data, rows, ptr = csc(dense[h])
        if legacy_coo:                              # (2 x nnz) coordinates (row, column) + values
            cols = np.repeat(np.arange(L), np.diff(ptr))
            fh.create_carray(grp, "coor", obj=np.vstack((rows, cols)).astype("uint32"), filters=fil)
            fh.create_carray(grp, "data", obj=data.astype(float), filters=fil)
            continue
        fh.create_carray(grp, "indptr", obj=ptr.astype("uint32"), filters=fil)
        fh.create_carray(grp, "indices", obj=rows.astype("uint32"), filters=fil)
        if not incidence_only:
            fh.create_carray(grp, "data", obj=data.astype(float), filters=fil)
    fh.close()


write(os.path.join(OUT, "pytables_csc_incidence.h5"), True)
write(os.path.join(OUT, "pytables_csc_values.h5"), False)
write(os.path.join(OUT, "pytables_legacy_coo.h5"), False, legacy_coo=True)
np.savez(os.path.join(OUT, "pytables_expected.npz"), dense=np.stack(dense), count=count, hname=np.array(hname),
         lname=np.array(lname), tables_version=np.array(tables.__version__))
print("wrote", OUT, "with PyTables", tables.__version__)
