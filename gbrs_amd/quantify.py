"""`gbrs quantify` on the MI355X path.

Drop-in for the workflow of gbrs/emase_utils.py:180-332 - same arguments and defaults, same output
files (`<outbase>.multiway.*` or, with a genotype file, `<outbase>.diploid.*`), same log vocabulary -
organised here as: resolve the support files, load the alignment tensor, optionally restrict it to the
called diplotypes, run the EM on the device (gbrs_amd.em.EMfactory -> libgbrs_hip), write the reports.
"""
from __future__ import annotations

import logging
import os
import re
import time

import numpy as np

from .alignment import load_alignment
from .em import EMfactory

logger = logging.getLogger('gbrs')

# support files looked up under $GBRS_DATA when the caller names none (gbrs/emase_utils.py:206-219)
DEFAULT_GROUP_FILE = 'ref.gene2transcripts.tsv'
DEFAULT_LENGTH_FILE = 'gbrs.hybridized.targets.info'


def _default_support_file(given, default_name, what_is_lost):
    if given is not None:
        return given
    path = os.path.join(os.getenv('GBRS_DATA', '.'), default_name)
    if not os.path.exists(path):
        logger.warning(what_is_lost)
    return path


_TWO_TABS_ON_A_LINE = re.compile(r'\t[^\t\n]*\t')


def read_genotype_table(genotype_file):
    """(gene ids, diplotype strings) of a `genotypes.tsv` in file order, one pair per line (a gene listed twice
    appears twice): `#Gene_ID<TAB>Diplotype` header, then `<gene><TAB><call>[<TAB>...]` lines.  Only the comment
    lines that open the file are skipped (gbrs/emase_utils.py:262: dropwhile(is_comment))."""
    with open(genotype_file) as fh:
        text = fh.read()
    pos = 0
    while text.startswith('#', pos):
        nl = text.find('\n', pos)
        pos = len(text) if nl < 0 else nl + 1
    body = text[pos:]
    if not body:
        return [], []
    n_lines = body.count('\n') + (0 if body.endswith('\n') else 1)
    # the plain form - exactly two tab-separated fields per line, no other white space - splits in one pass
    # (as many tabs as lines AND no line with two of them = exactly one tab on every line: a line with two tabs beside one
    # with none would otherwise pair the fields up wrongly, where the reference raises on the one-field line, `g, gt = item[:2]`)
    if body.count('\t') == n_lines and not (' ' in body or '\r' in body or '\n\n' in body or body.startswith('\n')) \
            and _TWO_TABS_ON_A_LINE.search(body) is None:
        tokens = body.replace('\n', '\t').split('\t')
        if not body.endswith('\n'):
            tokens.append('')
        if len(tokens) == 2 * n_lines + 1 and tokens[-1] == '':
            return tokens[0:-1:2], tokens[1:-1:2]
    genes, calls = [], []
    for line in body.splitlines():
        g, gt = line.rstrip().split('\t')[:2]           # ValueError on a one-field line, as in the reference
        genes.append(g)
        calls.append(gt)
    return genes, calls


def read_genotype_calls(genotype_file):
    """{gene id: diplotype string}; for a gene listed twice the later line (what gtcall_g keeps,
    gbrs/emase_utils.py:265)."""
    return dict(zip(*read_genotype_table(genotype_file)))


class CallNotes:
    """The `notes` argument of the report writers for one naming level: name -> diplotype call, None where the
    gene has no call (printed as `None`, as dict.fromkeys leaves it in the reference: gbrs/emase_utils.py:251-253).
    Behaves like the reference's dict for look-ups; a writer that lists exactly the names it was built over gets
    the whole notes column in one piece (aligned_blob)."""

    def __init__(self, names, text, has_call):
        self.names = names                 # the list / array the table rows are named by
        self.text = text                   # 'U' or 'S' array aligned with names: the call, 'None' without one
        self.has_call = has_call           # bool array aligned with names
        self._index = None

    def __getitem__(self, name):
        if self._index is None:
            self._index = {str(n): k for k, n in enumerate(self.names)}
        k = self._index[str(name)]
        if not self.has_call[k]:
            return None
        v = self.text[k]
        return v.decode() if isinstance(v, bytes) else str(v)

    def __len__(self):
        return len(self.text)

    def aligned_blob(self, row_names):
        """(bytes of str(call) for every row laid end to end, int64 offsets), or None when row_names is not the
        name list this column was built over (or a call is not plain ASCII)."""
        if row_names is not self.names:
            return None
        n, item = len(self.text), self.text.dtype.itemsize
        if self.text.dtype.kind == 'S':
            code = np.frombuffer(self.text.tobytes(), dtype=np.uint8).reshape(n, item)
        else:
            code = np.frombuffer(self.text.tobytes(), dtype=np.uint32).reshape(n, item // 4)
            if (code > 127).any():
                return None
        used = code != 0
        off = np.zeros(n + 1, dtype=np.int64)
        np.cumsum(used.sum(axis=1), out=off[1:])
        return code[used].astype(np.uint8).tobytes(), off


def _mark_loci(aln_mat, gidx, bits, text, note_type):
    """allowed bits per locus and the isoform notes from per-line (gene index, haplotype bits, call) in file
    order: every line marks the loci of its gene; an isoform keeps the call of the last line that names it."""
    L = aln_mat.num_loci
    allowed = np.zeros(L, dtype=np.uint32)
    iso_text, iso_has = np.full(L, 'None', dtype=note_type), np.zeros(L, dtype=bool)
    ptr, mem = aln_mat.group_csr()
    n = len(gidx)
    start, size = ptr[gidx], ptr[gidx + 1] - ptr[gidx]
    line_of = np.repeat(np.arange(n, dtype=np.int64), size)
    first = np.zeros(n, dtype=np.int64)
    np.cumsum(size[:-1], out=first[1:])
    loci = mem[start[line_of] + (np.arange(len(line_of), dtype=np.int64) - first[line_of])]
    np.bitwise_or.at(allowed, loci, bits[line_of])
    iso_text[loci] = text[line_of]                               # a repeated index keeps the last line's call
    iso_has[loci] = True
    return allowed, CallNotes(aln_mat.lname, iso_text, iso_has)


def genotype_mask_from_file(aln_mat, genotype_file):
    """diplotype_mask(aln_mat, read_genotype_table(genotype_file)) with the table parsed by the library
    (gbrs_parse_genotype_table); None when the file has a line that needs the interpreter's own parsing and errors."""
    from . import _lib
    from .em import _blob, _blob_cached
    G, H = len(aln_mat.gname), aln_mat.num_haplotypes
    with open(genotype_file, 'rb') as fh:
        raw = fh.read()
    gene_blob, gene_off = _blob_cached(aln_mat.gname)
    hap_blob, hap_off = _blob(aln_mat.hname)
    width = 8
    gene_bits = np.zeros(G, dtype=np.uint32)
    gene_call = np.zeros(G, dtype=f'S{width}')
    last_line = np.full(G, -1, dtype=np.int32)
    import ctypes as C
    n_lines = C.c_int64(0)
    st = _lib.load().gbrs_parse_genotype_table(raw, len(raw), gene_blob, _lib.ptr(gene_off), G, hap_blob,
                                               _lib.ptr(hap_off), H, _lib.ptr(gene_bits), _lib.ptr(gene_call), width,
                                               _lib.ptr(last_line), C.byref(n_lines))
    if st != 0:
        if st < 0:
            _lib.check(st)
        return None
    has = last_line >= 0
    gene_text = np.where(has, gene_call, np.bytes_(b'None')).astype(f'S{width}')
    gnotes = CallNotes(aln_mat.gname, gene_text, has)
    # the called genes in the order of their last lines: an isoform listed under two genes keeps the later one's call
    gidx = np.flatnonzero(has)
    gidx = gidx[np.argsort(last_line[gidx], kind='stable')]
    allowed, tnotes = _mark_loci(aln_mat, gidx, gene_bits[gidx], gene_text[gidx], f'S{width}')
    return allowed, gnotes, tnotes


def diplotype_mask(aln_mat, calls):
    """(allowed, gene notes, isoform notes) for genotype calls given as (gene ids, diplotype strings) in file
    order, or as a {gene: call} dict.

    allowed is uint32[L] with bit h set where haplotype h is one of the letters called for a gene the locus
    belongs to - the (H x L) 0/1 `gtmask` of gbrs/emase_utils.py:249-268 packed per locus; a gene listed twice
    contributes both lines, a locus in no called gene keeps nothing.  The notes are CallNotes over aln_mat.gname /
    aln_mat.lname: the call of the gene's (isoform's) last line, None without one."""
    from operator import itemgetter
    genes, texts = (list(calls.keys()), list(calls.values())) if isinstance(calls, dict) else calls
    L, G = aln_mat.num_loci, len(aln_mat.gname)
    n = len(genes)
    text = np.asarray(texts, dtype='U') if n else np.zeros(0, dtype='U1')
    width = text.dtype.itemsize // 4
    note_type = f'U{max(width, 4)}'
    gene_text, gene_has = np.full(G, 'None', dtype=note_type), np.zeros(G, dtype=bool)
    gnotes = CallNotes(aln_mat.gname, gene_text, gene_has)
    if n == 0:
        return (np.zeros(L, dtype=np.uint32), gnotes,
                CallNotes(aln_mat.lname, np.full(L, 'None', dtype=note_type), np.zeros(L, dtype=bool)))
    gene_index = {name: k for k, name in enumerate(aln_mat.gname.tolist())}
    found = itemgetter(*genes)(gene_index)                       # KeyError on a gene the group file does not list
    gidx = np.asarray(found if n > 1 else (found,), dtype=np.int64)
    # letters -> haplotype bits, through the distinct characters of the calls
    code = np.frombuffer(text.tobytes(), dtype=np.uint32).reshape(n, width)
    hap_index = {name: k for k, name in enumerate(aln_mat.hname)}
    uniq, inv = np.unique(code, return_inverse=True)
    bit_of = np.array([np.uint32(1) << np.uint32(hap_index[chr(cp)]) if cp else np.uint32(0)     # KeyError on an
                       for cp in uniq.tolist()], dtype=np.uint32)                                # unknown letter
    bits = np.bitwise_or.reduce(bit_of[inv].reshape(n, width), axis=1)
    allowed, tnotes = _mark_loci(aln_mat, gidx, bits, text, note_type)
    gene_text[gidx] = text                                       # a repeated index keeps the last line's call
    gene_has[gidx] = True
    return allowed, gnotes, tnotes


def _write_expression_reports(em, outbase, with_groups, isoform_notes, gene_notes, report_posterior):
    """The 2-5 files `gbrs quantify` leaves behind, in the reference's order (the isoform TPM report
    comes first because it rescales theta in place, which the later reports inherit)."""
    def emit(label, suffix, writer, **kw):
        path = f'{outbase}.{suffix}'
        logger.info(f'Generating {label}: {path}')
        writer(filename=path, **kw)

    from .em import ReportPool
    em.report_pool = ReportPool()           # the tables are formatted and written while the next one is being fetched
    try:
        _emit_reports(emit, em, with_groups, isoform_notes, gene_notes, report_posterior)
    finally:
        pool, em.report_pool = em.report_pool, None
        pool.finish()


def _emit_reports(emit, em, with_groups, isoform_notes, gene_notes, report_posterior):
    emit('isoform TPMs', 'isoforms.tpm', em.report_depths, tpm=True, notes=isoform_notes)
    emit('isoform Read Counts', 'isoforms.expected_read_counts', em.report_read_counts, notes=isoform_notes)
    if report_posterior:
        emit('Posterior Probabilities', 'posterior.h5', em.export_posterior_probability)
    if with_groups:
        emit('gene TPMs', 'genes.tpm', em.report_depths, tpm=True, grp_wise=True, notes=gene_notes)
        emit('gene Read Counts', 'genes.expected_read_counts', em.report_read_counts, grp_wise=True,
             notes=gene_notes)


def quantify(alignment_file: str, group_file: str = None, length_file: str = None,
             genotype_file: str = None, outbase: str = 'gbrs.quantified', multiread_model: int = 4,
             pseudocount: float = 0.0, max_iters: int = 999, tolerance: float = 0.0001,
             report_alignment_counts: bool = False, report_posterior: bool = False,
             device: int = 0, merge_identical_rows: bool = False, stage_times: dict = None,
             one_shot: bool = False, alignment=None, target_lengths=None) -> None:
    """Quantify allele-specific expression from an EMASE alignment file.  `stage_times` (optional
    dict) receives the wall-clock seconds of the stages: load, mask, em_setup, em_run, reports,
    alignment_counts.  `alignment` / `target_lengths` (extension, gbrs_amd.worker): the file's contents as
    load_alignment() and read_length_file() return them, when a resident process has them already - the
    multiway and the diploid pass of one sample read the same file."""
    clock = time.perf_counter
    marks = stage_times if stage_times is not None else {}
    group_file = _default_support_file(
        group_file, DEFAULT_GROUP_FILE,
        'A group file is not given. Group-level results will not be reported.')
    length_file = _default_support_file(
        length_file, DEFAULT_LENGTH_FILE,
        'A length file is not given. Transcript length adjustment will *not* be performed.')
    for label, value in (('Alignment File', alignment_file), ('Group File', group_file),
                         ('Length File', length_file), ('Genotype File', genotype_file),
                         ('Outbase', outbase), ('Multiread Model', multiread_model),
                         ('Pseudocount', pseudocount), ('Tolerance', tolerance),
                         ('Report Alignment Counts', report_alignment_counts),
                         ('Report Posterior', report_posterior)):
        logger.info(f'{label}: {value}')

    from . import _lib
    _lib.warm_up_device_async(device)      # HIP start-up overlaps with reading the files
    t0 = clock()
    logger.info(f'Loading EMASE file: {alignment_file}')
    # the length table only needs the names, which an HDF5 file yields before its index arrays are decoded: it is
    # parsed on a thread meanwhile (the decode is native code and holds no interpreter lock)
    from concurrent.futures import ThreadPoolExecutor
    from .em import read_length_file
    side = ThreadPoolExecutor(max_workers=1)
    lengths_pending = []

    def names_known(apm):
        if length_file is not None and target_lengths is None:
            lengths_pending.append(side.submit(read_length_file, apm, length_file, 100))
    if alignment is not None:
        aln_mat = alignment
        aln_mat.haplotype_mask = None              # a mask of an earlier pass is not this pass's
    else:
        aln_mat = load_alignment(alignment_file, grpfile=group_file, on_names=names_known)
    marks['load'] = clock() - t0

    t0 = clock()
    gene_notes = isoform_notes = None
    if genotype_file is None:
        outbase = f'{outbase}.multiway'
    else:
        outbase = f'{outbase}.diploid'
        logger.info(f'Loading and processing genotype calls from: {genotype_file}')
        allowed, gene_notes, isoform_notes = genotype_mask_from_file(aln_mat, genotype_file) or \
            diplotype_mask(aln_mat, read_genotype_table(genotype_file))
        aln_mat.set_haplotype_mask(allowed)        # applied on the device when the EM handle is built
    logger.debug(f'Outbase now: {outbase}')
    marks['mask'] = clock() - t0

    logger.info('Running EMASE')
    t0 = clock()
    em = EMfactory(aln_mat, device=device, merge_identical_rows=merge_identical_rows, one_shot=one_shot)
    if target_lengths is not None:
        em.set_target_lengths(target_lengths)
        em.prepare(pseudocount=pseudocount)
    elif lengths_pending:
        em.set_target_lengths(lengths_pending[0].result())
        em.prepare(pseudocount=pseudocount)
    else:
        em.prepare(pseudocount=pseudocount, lenfile=length_file)
    side.shutdown(wait=False)
    marks['em_setup'] = clock() - t0
    t0 = clock()
    em.run(model=multiread_model, tol=tolerance, max_iters=max_iters, verbose=True)
    marks['em_run'] = clock() - t0
    marks['em_iterations'] = em.num_iters

    t0 = clock()
    _write_expression_reports(em, outbase, group_file is not None, isoform_notes, gene_notes, report_posterior)
    em.close()
    marks['reports'] = clock() - t0

    if report_alignment_counts:
        t0 = clock()
        from .counts import AlignmentCounter, report_alignment_counts as write_counts
        # the reference reloads the file for this report (gbrs/emase_utils.py:318-331), so the counts are always those
        # of the unmasked alignments; a `-G` mask here is a note for the device and leaves the host arrays as loaded -
        # unless something (--report-posterior) has carried it out on them since
        fresh = aln_mat if genotype_file is None or aln_mat.haplotype_mask is not None else \
            load_alignment(alignment_file, grpfile=group_file)
        with AlignmentCounter(fresh, device=device) as counter:       # one upload of the alignments for both levels
            for level, grp_wise in (('isoform', False), ('gene', True)):
                if grp_wise and group_file is None:
                    continue
                path = f'{outbase}.{level}s.alignment_counts'
                logger.info(f'Generating {level} Alignment Counts: {path}')
                write_counts(fresh, path, grp_wise=grp_wise, device=device, counter=counter)
        marks['alignment_counts'] = clock() - t0
    logger.debug('Done')
