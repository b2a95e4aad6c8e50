"""`gbrs interpolate` and `gbrs export` on the MI355X path: same files in and out as
gbrs_utils.interpolate (gbrs/gbrs_utils.py:612-697) and gbrs_utils.export (:863-938); the
interpolation and the 36 -> 8 dosage product run in HIP kernels (gbrs_interpolate,
gbrs_genoprob_dosage)."""
from __future__ import annotations

import logging
import os
from collections import OrderedDict, defaultdict

import numpy as np

from . import _lib
from .hmm import get_chromosome_info

logger = logging.getLogger('gbrs')


def interpolate_arrays(x_gene, gamma, x_grid, device=0):
    """gamma (S x n) at gene positions x_gene (n) -> (S x len(x_grid)), end points padded as the
    reference does (:664-676, :684-688)."""
    gamma = np.asarray(gamma, dtype=np.float64)
    S = gamma.shape[0]
    x = np.append([0.0], np.asarray(x_gene, dtype=np.float64))
    x = np.append(x, [x_grid[-1] + 1.0])
    y = np.hstack((gamma[:, 0][:, np.newaxis], gamma))
    y = np.hstack((y, y[:, -1][:, np.newaxis]))
    order = np.argsort(x, kind='mergesort')            # interp1d(assume_sorted=False)
    x = np.ascontiguousarray(x[order])
    y = np.ascontiguousarray(np.take(y, order, axis=1))
    xq = np.ascontiguousarray(x_grid, dtype=np.float64)
    out = np.empty((S, len(xq)), dtype=np.float64)
    st = _lib.load().gbrs_interpolate(S, len(x), _lib.ptr(x), _lib.ptr(y), len(xq), _lib.ptr(xq), _lib.ptr(out),
                                      device)
    if st == _lib.GBRS_ERR_INVALID and b'interpolation range' in _lib.load().gbrs_last_error():
        raise ValueError(_lib.load().gbrs_last_error().decode())
    _lib.check(st)
    return out


def interpolate(genoprob_file: str, grid_file: str = None, gpos_file: str = None, output_file: str = None,
                device: int = 0) -> None:
    data_dir = os.getenv('GBRS_DATA', '.')
    if gpos_file is None:
        gpos_file = os.path.join(data_dir, 'ref.gene_pos.ordered.npz')
        try:
            x_gene = np.load(gpos_file)
        except Exception:
            logger.error(f'Please make sure if $GBRS_DATA is set correctly: {data_dir}')
            raise
    else:
        x_gene = np.load(gpos_file)
    if grid_file is None:
        grid_file = os.path.join(data_dir, 'ref.genome_grid.64k.txt')
    if output_file is None:
        output_file = f'gbrs.interpolated.{os.path.basename(genoprob_file)}'
    logger.info(f'Genotype Probability File: {genoprob_file}')
    logger.info(f'Grid File: {grid_file}')
    logger.info(f'Gene Position File: {gpos_file}')
    logger.info(f'Output File: {output_file}')
    logger.info('Loading chromosome information')
    get_chromosome_info(data_dir)            # the reference requires ref.fa.fai here too (:651)
    logger.info(f'Loading grid file: {grid_file}')
    x_grid = defaultdict(list)
    with open(grid_file) as fh:
        fh.readline()
        for line in fh:
            item = line.rstrip().split('\t')
            x_grid[item[1]].append(float(item[3]))
    x_grid = dict(x_grid)
    logger.info(f'Loading GBRS genotype probability file: {genoprob_file}')
    gamma_gene = np.load(genoprob_file)
    out = dict()
    for c in x_grid.keys():
        if c in x_gene.files and c in gamma_gene.files:
            logger.debug(f'Working on {c}')
            xs = [float(row[1]) for row in x_gene[c]]
            out[c] = interpolate_arrays(xs, gamma_gene[c], x_grid[c], device=device)
    logger.info(f'Saving interpolate probability file: {output_file}')
    np.savez_compressed(output_file, **out)
    logger.info('Done')


def export(genoprob_file: str, strains: list, grid_file: str = None, output_file: str = None,
           device: int = 0) -> None:
    data_dir = os.getenv('GBRS_DATA', '.')
    if grid_file is None:
        grid_file = os.path.join(data_dir, 'ref.genome_grid.64k.txt')
    if output_file is None:
        output_file = f'{os.path.splitext(genoprob_file)[0]}.tsv'
    logger.info(f'Genotype Probabilities File: {genoprob_file}')
    logger.info(f'Strains: {strains}')
    logger.info(f'Grid File: {grid_file}')
    logger.info(f'Output File: {output_file}')
    num_strains = len(strains)
    logger.info(f'Loading grid file: {grid_file}')
    with open(grid_file) as fh:
        next(fh)
        grid = OrderedDict()
        for line in fh:
            chrom = line.rstrip().split('\t')[1]
            grid[chrom] = grid.get(chrom, 0) + 1
    logger.debug(f'Number of grids: {sum(grid.values())}')
    logger.info(f'Loading GBRS genotype probability file: {genoprob_file}')
    gprob = np.load(genoprob_file)
    gprob_mat = np.vstack([gprob[c].transpose() for c in grid.keys()])
    S = num_strains * (num_strains + 1) // 2
    if gprob_mat.shape[1] != S:
        raise ValueError(f'shapes {gprob_mat.shape} and ({S},{num_strains}) not aligned')
    logger.info('Converting genotype probability')
    gp = np.ascontiguousarray(gprob_mat, dtype=np.float64)
    conv = np.empty((gp.shape[0], num_strains), dtype=np.float64)
    _lib.check(_lib.load().gbrs_genoprob_dosage(num_strains, gp.shape[0], _lib.ptr(gp), _lib.ptr(conv), device))
    logger.info(f'Saving GBRS quant format: {output_file}')
    np.savetxt(output_file, conv, fmt='%.6f', delimiter='\t', header='\t'.join(strains))
    logger.info('Done')
