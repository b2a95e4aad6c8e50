"""`gbrs quantify` on the MI355X path.

Drop-in for the workflow of gbrs/emase_utils.py:180-332 - same arguments and defaults, same output
files (`<outbase>.multiway.*` or, with a genotype file, `<outbase>.diploid.*`), same log vocabulary -
organised here as: resolve the support files, load the alignment tensor, optionally restrict it to the
called diplotypes, run the EM on the device (gbrs_amd.em.EMfactory -> libgbrs_hip), write the reports.
"""
from __future__ import annotations

import logging
import os
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


def read_genotype_calls(genotype_file):
    """{gene id: diplotype string} from a `genotypes.tsv` (`#Gene_ID<TAB>Diplotype` header, then one
    line per gene).  Only the comment lines that open the file are skipped (gbrs/emase_utils.py:248)."""
    calls = {}
    in_header = True
    with open(genotype_file) as fh:
        for line in fh:
            if in_header and line.startswith('#'):
                continue
            in_header = False
            fields = line.rstrip().split('\t')
            calls[fields[0]] = fields[1]
    return calls


def diplotype_mask(aln_mat, calls):
    """(mask, gene notes, isoform notes) for a set of genotype calls.

    mask is (H x L) with 1 where the haplotype is one of the two letters called for the locus's gene
    (gbrs/emase_utils.py:245-269); the notes map every gene / isoform name to its call, None when
    the gene has none (those print as `None` in the reports' notes column, as in the reference)."""
    hap_index = {name: k for k, name in enumerate(aln_mat.hname)}
    gene_index = {name: k for k, name in enumerate(aln_mat.gname)}
    mask = np.zeros((aln_mat.num_haplotypes, aln_mat.num_loci))
    gene_notes = {str(g): None for g in aln_mat.gname}
    isoform_notes = {t: None for t in aln_mat.lname}
    for gene, call in calls.items():
        members = np.asarray(aln_mat.groups[gene_index[gene]], dtype=np.int64)
        letters = [hap_index[ch] for ch in call]
        mask[np.ix_(letters, members)] = 1.0
        gene_notes[gene] = call
        for t in members:
            isoform_notes[aln_mat.lname[t]] = call
    return mask, gene_notes, isoform_notes


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
             device: int = 0, merge_identical_rows: bool = False, stage_times: dict = None) -> None:
    """Quantify allele-specific expression from an EMASE alignment file.  `stage_times` (optional
    dict) receives the wall-clock seconds of the stages: load, mask, em_setup, em_run, reports,
    alignment_counts."""
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
        if length_file is not None:
            lengths_pending.append(side.submit(read_length_file, apm, length_file, 100))
    aln_mat = load_alignment(alignment_file, grpfile=group_file, on_names=names_known)
    marks['load'] = clock() - t0

    t0 = clock()
    gene_notes = isoform_notes = None
    if genotype_file is None:
        outbase = f'{outbase}.multiway'
    else:
        outbase = f'{outbase}.diploid'
        logger.info(f'Loading and processing genotype calls from: {genotype_file}')
        mask, gene_notes, isoform_notes = diplotype_mask(aln_mat, read_genotype_calls(genotype_file))
        aln_mat.mask_haplotype_loci(mask)
    logger.debug(f'Outbase now: {outbase}')
    marks['mask'] = clock() - t0

    logger.info('Running EMASE')
    t0 = clock()
    em = EMfactory(aln_mat, device=device, merge_identical_rows=merge_identical_rows)
    if lengths_pending:
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
        from .counts import report_alignment_counts as write_counts
        # the EM above may have masked the tensor (-G); the reference reloads the file for this report
        # (gbrs/emase_utils.py:318-331), so the counts are always those of the unmasked alignments
        fresh = load_alignment(alignment_file, grpfile=group_file) if genotype_file is not None else aln_mat
        for level, grp_wise in (('isoform', False), ('gene', True)):
            if grp_wise and group_file is None:
                continue
            path = f'{outbase}.{level}s.alignment_counts'
            logger.info(f'Generating {level} Alignment Counts: {path}')
            write_counts(fresh, path, grp_wise=grp_wise, device=device)
        marks['alignment_counts'] = clock() - t0
    logger.debug('Done')
