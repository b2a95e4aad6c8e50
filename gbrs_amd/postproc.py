"""`gbrs interpolate` and `gbrs export` on the MI355X path.  Files in and out are those of
gbrs_utils.interpolate (gbrs/gbrs_utils.py:612-697) and gbrs_utils.export (:863-938); the piecewise-linear
interpolation and the 36 -> 8 dosage product run in HIP kernels (gbrs_interpolate, gbrs_genoprob_dosage)."""
from __future__ import annotations

import logging
import os

import numpy as np

from . import _lib
from .hmm import get_chromosome_info

logger = logging.getLogger('gbrs')

GRID_FILE = 'ref.genome_grid.64k.txt'
GENE_POSITION_FILE = 'ref.gene_pos.ordered.npz'


def _support_file(given, default_name):
    """A support file named on the command line, or its default under $GBRS_DATA."""
    return given if given is not None else os.path.join(os.getenv('GBRS_DATA', '.'), default_name)


def read_grid(grid_file):
    """The marker grid: {chromosome: positions (column 4, as float64)} in file order.  One header line;
    columns are tab separated with the chromosome in the second one."""
    chroms, positions = [], []
    with open(grid_file) as fh:
        fh.readline()
        for line in fh:
            fields = line.rstrip().split('\t')
            chroms.append(fields[1])
            positions.append(fields[3] if len(fields) > 3 else 'nan')
    positions = np.asarray(positions, dtype=np.float64)
    chroms = np.asarray(chroms)
    grid = {}
    for c in dict.fromkeys(chroms.tolist()):               # first-appearance order
        grid[c] = positions[chroms == c]
    return grid


def interpolate_arrays(x_gene, gamma, x_grid, device=0):
    """gamma (S x n) at gene positions x_gene (n) -> (S x len(x_grid)).  The curve is held constant outside the
    genes: a knot at 0 repeats the first column and one past the last grid point repeats the last (:664-676,
    :684-688); knots are sorted stably, as scipy's interp1d(assume_sorted=False) does."""
    gamma = np.asarray(gamma, dtype=np.float64)
    n_states = gamma.shape[0]
    xq = np.ascontiguousarray(x_grid, dtype=np.float64)
    knots = np.concatenate(([0.0], np.asarray(x_gene, dtype=np.float64), [xq[-1] + 1.0]))
    values = np.concatenate((gamma[:, :1], gamma, gamma[:, -1:]), axis=1)
    order = np.argsort(knots, kind='stable')
    knots = np.ascontiguousarray(knots[order])
    values = np.ascontiguousarray(values[:, order])
    out = np.empty((n_states, len(xq)), dtype=np.float64)
    lib = _lib.load()
    status = lib.gbrs_interpolate(n_states, len(knots), _lib.ptr(knots), _lib.ptr(values), len(xq), _lib.ptr(xq),
                                  _lib.ptr(out), device)
    if status == _lib.GBRS_ERR_INVALID and b'interpolation range' in lib.gbrs_last_error():
        raise ValueError(lib.gbrs_last_error().decode())          # what interp1d raises for a point outside the knots
    _lib.check(status)
    return out


def interpolate(genoprob_file: str, grid_file: str = None, gpos_file: str = None, output_file: str = None,
                device: int = 0) -> None:
    """Gene-level genotype probabilities -> probabilities on the marker grid, chromosome by chromosome."""
    default_gpos = gpos_file is None
    gpos_file = _support_file(gpos_file, GENE_POSITION_FILE)
    grid_file = _support_file(grid_file, GRID_FILE)
    if output_file is None:
        output_file = f'gbrs.interpolated.{os.path.basename(genoprob_file)}'
    for label, value in (('Genotype Probability File', genoprob_file), ('Grid File', grid_file),
                         ('Gene Position File', gpos_file), ('Output File', output_file)):
        logger.info(f'{label}: {value}')
    try:
        gene_positions = np.load(gpos_file)
    except Exception:
        if default_gpos:
            logger.error(f"Please make sure if $GBRS_DATA is set correctly: {os.getenv('GBRS_DATA', '.')}")
        raise
    logger.info('Loading chromosome information')
    get_chromosome_info(os.getenv('GBRS_DATA', '.'))       # the reference insists on ref.fa.fai here as well (:651)
    logger.info(f'Loading grid file: {grid_file}')
    grid = read_grid(grid_file)
    logger.info(f'Loading GBRS genotype probability file: {genoprob_file}')
    gene_probs = np.load(genoprob_file)
    on_grid = {}
    for chrom, markers in grid.items():
        if chrom not in gene_positions.files or chrom not in gene_probs.files:
            continue
        logger.debug(f'Working on {chrom}')
        where = np.asarray([record[1] for record in gene_positions[chrom]], dtype=np.float64)
        on_grid[chrom] = interpolate_arrays(where, gene_probs[chrom], markers, device=device)
    logger.info(f'Saving interpolate probability file: {output_file}')
    np.savez_compressed(output_file, **on_grid)
    logger.info('Done')


def export(genoprob_file: str, strains: list, grid_file: str = None, output_file: str = None,
           device: int = 0) -> None:
    """Grid-level diplotype probabilities -> founder dosages (one row per marker, one column per strain), as a
    tab separated table with six decimals."""
    grid_file = _support_file(grid_file, GRID_FILE)
    if output_file is None:
        output_file = f'{os.path.splitext(genoprob_file)[0]}.tsv'
    for label, value in (('Genotype Probabilities File', genoprob_file), ('Strains', strains),
                         ('Grid File', grid_file), ('Output File', output_file)):
        logger.info(f'{label}: {value}')
    n_strains = len(strains)
    n_states = n_strains * (n_strains + 1) // 2
    logger.info(f'Loading grid file: {grid_file}')
    grid = read_grid(grid_file)
    logger.debug(f'Number of grids: {sum(len(v) for v in grid.values())}')
    logger.info(f'Loading GBRS genotype probability file: {genoprob_file}')
    probs = np.load(genoprob_file)
    by_marker = np.ascontiguousarray(np.concatenate([probs[c].T for c in grid], axis=0), dtype=np.float64)
    if by_marker.shape[1] != n_states:
        raise ValueError(f'shapes {by_marker.shape} and ({n_states},{n_strains}) not aligned')
    logger.info('Converting genotype probability')
    dosage = np.empty((by_marker.shape[0], n_strains), dtype=np.float64)
    _lib.check(_lib.load().gbrs_genoprob_dosage(n_strains, by_marker.shape[0], _lib.ptr(by_marker), _lib.ptr(dosage),
                                                device))
    logger.info(f'Saving GBRS quant format: {output_file}')
    np.savetxt(output_file, dosage, fmt='%.6f', delimiter='\t', header='\t'.join(strains))
    logger.info('Done')
