"""`gbrs quantify` workflow on the MI355X path: same arguments, defaults, log lines and output
files as gbrs/emase_utils.py:180-332 (load alignment file + groups, optional genotype mask,
EMASE EM, 2-6 report files).  The EM runs through gbrs_amd.em.EMfactory (HIP kernels)."""
from __future__ import annotations

import logging
import os
from itertools import dropwhile

import numpy as np

from .alignment import load_alignment
from .em import EMfactory

logger = logging.getLogger('gbrs')


def is_comment(s: str) -> bool:
    return s.startswith('#')


def genotype_mask(aln_mat, genotype_file):
    """(gtmask H x L, gene calls, transcript calls) from a genotypes.tsv
    (gbrs/emase_utils.py:245-269)."""
    hid = dict(zip(aln_mat.hname, np.arange(aln_mat.num_haplotypes)))
    gid = dict(zip(aln_mat.gname, np.arange(len(aln_mat.gname))))
    gtmask = np.zeros((aln_mat.num_haplotypes, aln_mat.num_loci))
    gtcall_g = dict.fromkeys(aln_mat.gname)
    gtcall_t = dict.fromkeys(aln_mat.lname)
    with open(genotype_file) as fh:
        for curline in dropwhile(is_comment, fh):
            item = curline.rstrip().split('\t')
            g, gt = item[:2]
            gtcall_g[g] = gt
            hid2set = np.array([hid[c] for c in gt])
            tid2set = np.array(aln_mat.groups[gid[g]])
            gtmask[tuple(np.meshgrid(hid2set, tid2set))] = 1.0
            for t in tid2set:
                gtcall_t[aln_mat.lname[t]] = gt
    return gtmask, gtcall_g, gtcall_t


def quantify(alignment_file: str, group_file: str = None, length_file: str = None,
             genotype_file: str = None, outbase: str = 'gbrs.quantified', multiread_model: int = 4,
             pseudocount: float = 0.0, max_iters: int = 999, tolerance: float = 0.0001,
             report_alignment_counts: bool = False, report_posterior: bool = False,
             device: int = 0, merge_identical_rows: bool = False) -> None:
    """Quantify expected read counts."""
    data_dir = os.getenv('GBRS_DATA', '.')
    if group_file is None:
        group_file = os.path.join(data_dir, 'ref.gene2transcripts.tsv')
        if not os.path.exists(group_file):
            logger.warning('A group file is not given. Group-level results will not be reported.')
    if length_file is None:
        length_file = os.path.join(data_dir, 'gbrs.hybridized.targets.info')
        if not os.path.exists(length_file):
            logger.warning('A length file is not given. Transcript length adjustment will *not* be performed.')
    report_group_counts = group_file is not None

    logger.info(f'Alignment File: {alignment_file}')
    logger.info(f'Group File: {group_file}')
    logger.info(f'Length File: {length_file}')
    logger.info(f'Genotype File: {genotype_file}')
    logger.info(f'Outbase: {outbase}')
    logger.info(f'Multiread Model: {multiread_model}')
    logger.info(f'Pseudocount: {pseudocount}')
    logger.info(f'Tolerance: {tolerance}')
    logger.info(f'Report Alignment Counts: {report_alignment_counts}')
    logger.info(f'Report Posterior: {report_posterior}')

    logger.info(f'Loading EMASE file: {alignment_file}')
    aln_mat = load_alignment(alignment_file, grpfile=group_file)

    if genotype_file is not None:
        outbase = f'{outbase}.diploid'
        logger.debug(f'Outbase now: {outbase}')
        logger.info(f'Loading and processing genotype calls from: {genotype_file}')
        gtmask, gtcall_g, gtcall_t = genotype_mask(aln_mat, genotype_file)
        aln_mat.mask_haplotype_loci(gtmask)
    else:
        outbase = f'{outbase}.multiway'
        logger.debug(f'Outbase now: {outbase}')
        gtcall_g = None
        gtcall_t = None

    logger.info('Running EMASE')
    em_factory = EMfactory(aln_mat, device=device, merge_identical_rows=merge_identical_rows)
    em_factory.prepare(pseudocount=pseudocount, lenfile=length_file)
    em_factory.run(model=multiread_model, tol=tolerance, max_iters=max_iters, verbose=True)

    logger.info(f'Generating isoform TPMs: {outbase}.isoforms.tpm')
    em_factory.report_depths(filename=f'{outbase}.isoforms.tpm', tpm=True, notes=gtcall_t)
    logger.info(f'Generating isoform Read Counts: {outbase}.isoforms.expected_read_counts')
    em_factory.report_read_counts(filename=f'{outbase}.isoforms.expected_read_counts', notes=gtcall_t)
    if report_posterior:
        logger.info(f'Generating Posterior Probabilities: {outbase}.posterior.h5')
        em_factory.export_posterior_probability(filename=f'{outbase}.posterior.h5')
    if report_group_counts:
        logger.info(f'Generating gene TPMs: {outbase}.genes.tpm')
        em_factory.report_depths(filename=f'{outbase}.genes.tpm', tpm=True, grp_wise=True, notes=gtcall_g)
        logger.info(f'Generating gene Read Counts: {outbase}.genes.expected_read_counts')
        em_factory.report_read_counts(filename=f'{outbase}.genes.expected_read_counts', grp_wise=True,
                                      notes=gtcall_g)
    em_factory.close()

    if report_alignment_counts:
        from .counts import report_alignment_counts as write_counts
        alnmat = load_alignment(alignment_file, grpfile=group_file)
        logger.info(f'Generating isoform Alignment Counts: {outbase}.isoforms.alignment_counts')
        write_counts(alnmat, f'{outbase}.isoforms.alignment_counts', grp_wise=False, device=device)
        if report_group_counts:
            logger.info(f'Generating gene Alignment Counts: {outbase}.genes.alignment_counts')
            write_counts(alnmat, f'{outbase}.genes.alignment_counts', grp_wise=True, device=device)
    logger.debug('Done')
