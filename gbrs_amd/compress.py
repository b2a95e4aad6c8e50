"""`gbrs compress` on the MI355X path: same arguments and output file as
gbrs/emase_utils.py:22-107; the equivalence-class construction runs in HIP (gbrs_compress_*)."""
from __future__ import annotations

import ctypes as C
import logging

import numpy as np

from . import _lib
from .alignment import AlignmentPropertyMatrix, load_alignment

logger = logging.getLogger('gbrs')


def compress_matrix(apm, device=0):
    """AlignmentPropertyMatrix -> AlignmentPropertyMatrix of equivalence classes with counts."""
    lib = _lib.load()
    L, H, R = apm.shape
    cnt = None if apm.count is None else np.ascontiguousarray(apm.count, dtype=np.float64)
    h = C.c_void_p()
    n_ecs = C.c_uint64(0)
    nnz = np.zeros(H, dtype=np.uint64)
    _lib.check(lib.gbrs_compress_create(R, L, H, _lib.ptr_table(apm.indptr), _lib.ptr_table(apm.indices),
                                        _lib.ptr(cnt), device, C.byref(h), C.byref(n_ecs), _lib.ptr(nnz)))
    try:
        ip = [np.zeros(L + 1, dtype=np.uint32) for _ in range(H)]
        ix = [np.zeros(int(nnz[k]), dtype=np.uint32) for k in range(H)]
        count = np.zeros(int(n_ecs.value), dtype=np.float64)
        _lib.check(lib.gbrs_compress_get(h, _lib.ptr_table(ip), _lib.ptr_table(ix), _lib.ptr(count)))
    finally:
        lib.gbrs_compress_destroy(h)
    return AlignmentPropertyMatrix(shape=(L, H, max(int(n_ecs.value), 1)), indptr=ip, indices=ix,
                                   count=count if n_ecs.value else np.zeros(1), haplotype_names=apm.hname,
                                   locus_names=apm.lname)


def compress(emase_files: list, output_file: str, comp_lib: str = 'zlib', device: int = 0) -> None:
    """Compress EMASE file(s) to an alignment incidence matrix of equivalence classes."""
    for x in emase_files:
        logger.info(f'EMASE file: {x}')
    logger.info(f'Output File: {output_file}')
    logger.info(f'Compression Library: {comp_lib}')
    mats = []
    for aln_file in emase_files:
        logger.info(f'Loading EMASE file: {aln_file}')
        m = load_alignment(aln_file)
        logger.debug(f'Number Loci: {m.num_loci}')
        logger.debug(f'Number Haplotypes: {m.num_haplotypes}')
        logger.debug(f'Number Reads: {m.num_reads}')
        mats.append(m)
    first = mats[0]
    if len(mats) > 1:          # the files' reads one after another (:46-77 iterates file by file)
        L, H, _ = first.shape
        off, ips, ixs, cnts = 0, [], [], []
        for m in mats:
            if m.shape[:2] != (L, H):
                raise RuntimeError('The EMASE files do not share loci / haplotypes.')
            cnts.append(np.ones(m.num_reads) if m.count is None else m.count)
            ixs.append([m.indices[h].astype(np.int64) + off for h in range(H)])
            ips.append([m.indptr[h].astype(np.int64) for h in range(H)])
            off += m.num_reads
        indptr, indices = [], []
        for h in range(H):
            cols = np.concatenate([np.repeat(np.arange(L), np.diff(ips[k][h])) for k in range(len(mats))])
            rows = np.concatenate([ixs[k][h] for k in range(len(mats))])
            order = np.lexsort((rows, cols))
            indices.append(rows[order].astype(np.uint32))
            indptr.append(np.searchsorted(cols[order], np.arange(L + 1)).astype(np.uint32))
        first = AlignmentPropertyMatrix(shape=(L, H, off), indptr=indptr, indices=indices,
                                        count=np.concatenate(cnts), haplotype_names=first.hname,
                                        locus_names=first.lname)
    logger.debug('Creating unique ECs')
    ec = compress_matrix(first, device=device)
    logger.info('Constructing APM')
    logger.debug(f'Number ECs: {ec.num_reads}')
    logger.info(f'Saving EMASE Formatted File: {output_file}')
    ec.save(output_file, complib=comp_lib)
    logger.info('Done')
