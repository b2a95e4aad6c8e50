"""CPU ORACLE (test infrastructure) for `gbrs compress`.

Restatement of the equivalence-class loop of gbrs/emase_utils.py:60-103 on in-memory CSC arrays:
key of a row = per-haplotype sorted locus lists; classes in order of first appearance (dict
insertion order); count = sum of member counts (1 per read when the input has no count).
Parity pin: tests/golden/compress_*.npz were written by oracle/gen_golden.py from the reference's own
compress() run on in-memory matrices (its two PyTables file boundaries, the h5 load and the h5 save, are
served from / captured into memory by a subclass of the reference's AlignmentPropertyMatrix; the
equivalence-class loop runs untouched), and this restatement was checked bit for bit against it there:
class order, structure and counts, with EC counts, empty rows, 16 haplotypes and two input files.
"""
import numpy as np


def compress(R, L, H, indptr, indices, count=None):
    w = np.ones(R) if count is None else np.asarray(count, dtype=np.float64)
    per_row = [[[] for _ in range(H)] for _ in range(R)]
    for h in range(H):
        ptr = np.asarray(indptr[h], dtype=np.int64)
        col = np.repeat(np.arange(L, dtype=np.int64), np.diff(ptr))
        for r, l in zip(np.asarray(indices[h], dtype=np.int64), col):
            per_row[r][h].append(int(l))
    ec = {}
    for r in range(R):
        key = ':'.join(','.join(map(str, sorted(per_row[r][h]))) for h in range(H))
        ec[key] = ec.get(key, 0) + w[r]
    counts = np.array(list(ec.values()), dtype=np.float64)
    ip, ix = [], []
    for h in range(H):
        rows, cols = [], []
        for row_id, key in enumerate(ec):
            part = key.split(':')[h]
            if part != '':
                for l in map(int, part.split(',')):
                    rows.append(row_id)
                    cols.append(l)
        rows, cols = np.asarray(rows, dtype=np.int64), np.asarray(cols, dtype=np.int64)
        order = np.lexsort((rows, cols))
        ix.append(rows[order].astype(np.uint32))
        ip.append(np.searchsorted(cols[order], np.arange(L + 1)).astype(np.uint32))
    return len(ec), ip, ix, counts
