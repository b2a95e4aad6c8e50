"""`gbrs reconstruct` on MI355X: same inputs, outputs and defaults as
gbrs_utils.reconstruct (gbrs/gbrs_utils.py:382-609); the emission model, forward, backward,
posterior, Viterbi and backtrace all run in HIP kernels through include/gbrs_hip.h.
No CPU fallback: without libgbrs_hip.so / a gfx950 device every entry point raises.
"""
from __future__ import annotations

import ctypes as C
import logging
import os
from collections import OrderedDict
from itertools import combinations_with_replacement

import numpy as np

from . import _lib

logger = logging.getLogger('gbrs')


class DiplotypeHMM:
    """Device handle for one set of transition tables (sample independent)."""

    def __init__(self, num_haps, chroms, n_genes, tprob, device=0):
        """chroms: names; n_genes[c]; tprob[c] float64 [n_t, S, S] log T[i][to, from]."""
        lib = _lib.load()
        self.H = int(num_haps)
        self.S = self.H * (self.H + 1) // 2
        self.chroms = list(chroms)
        self.n_genes = np.asarray(n_genes, dtype=np.int32)
        self._tp = [np.ascontiguousarray(t, dtype=np.float64) for t in tprob]
        for c, t in zip(self.chroms, self._tp):
            if t.ndim != 3 or t.shape[1:] != (self.S, self.S):
                raise ValueError(f'tprob[{c}] has shape {t.shape}, expected (n, {self.S}, {self.S})')
        self.n_trans = np.asarray([len(t) for t in self._tp], dtype=np.int32)
        for c, n, nt in zip(self.chroms, self.n_genes, self.n_trans):
            if nt < n - 1:
                raise IndexError(f'index {n - 2} is out of bounds for axis 0 with size {nt}')
        h = C.c_void_p()
        _lib.check(lib.gbrs_hmm_create(self.H, len(self.chroms), _lib.ptr(self.n_genes),
                                       _lib.ptr(self.n_trans), _lib.ptr_table(self._tp), device,
                                       C.byref(h)))
        self._h = h
        self.n_samples = 0

    def close(self):
        if getattr(self, '_h', None) is not None:
            _lib.load().gbrs_hmm_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_expression(self, expr, avecs, has_avec, expr_threshold=1.5, sigma=0.12):
        """expr[c] [n_samples, n_c, H] (or [n_c, H]); avecs[c] [n_c, H, H]; has_avec[c] bool [n_c]."""
        ex, av, ha = [], [], []
        ns = None
        for c, n in enumerate(self.n_genes):
            e = np.ascontiguousarray(expr[c], dtype=np.float64)
            if e.ndim == 2:
                e = e[None]
            if e.shape[1:] != (n, self.H):
                raise ValueError(f'expr[{c}] has shape {e.shape}')
            ns = e.shape[0] if ns is None else ns
            if e.shape[0] != ns:
                raise ValueError('inconsistent number of samples')
            ex.append(np.ascontiguousarray(e))
            a = np.ascontiguousarray(avecs[c], dtype=np.float64)
            if a.shape != (n, self.H, self.H):
                raise ValueError(f'avecs[{c}] has shape {a.shape}')
            av.append(a)
            ha.append(np.ascontiguousarray(has_avec[c], dtype=np.uint8))
        _lib.check(_lib.load().gbrs_hmm_set_expression(
            self._h, ns, _lib.ptr_table(ex), _lib.ptr_table(av), _lib.ptr_table(ha),
            float(expr_threshold), float(sigma)))
        self.n_samples = ns

    def set_eprob(self, eprob):
        ep = []
        ns = None
        for c, n in enumerate(self.n_genes):
            e = np.ascontiguousarray(eprob[c], dtype=np.float64)
            if e.ndim == 2:
                e = e[None]
            if e.shape[1:] != (n, self.S):
                raise ValueError(f'eprob[{c}] has shape {e.shape}')
            ns = e.shape[0] if ns is None else ns
            ep.append(np.ascontiguousarray(e))
        _lib.check(_lib.load().gbrs_hmm_set_eprob(self._h, ns, _lib.ptr_table(ep)))
        self.n_samples = ns

    def run(self):
        _lib.check(_lib.load().gbrs_hmm_run(self._h))

    def get(self, chrom, sample=0, want=('gamma', 'states', 'calls')):
        c = chrom if isinstance(chrom, int) else self.chroms.index(chrom)
        n, S = int(self.n_genes[c]), self.S
        m = min(n, int(self.n_trans[c]))
        bufs = dict(gamma=np.empty((S, n)), states=np.empty(m + 1, dtype=np.int32),
                    calls=np.empty(n, dtype=np.int32), alpha=np.empty((S, n)), beta=np.empty((S, n)),
                    delta=np.empty((S, n)), scaler=np.empty(n), eprob=np.empty((n, S)))
        args = [(_lib.ptr(bufs[k]) if k in want else None)
                for k in ('gamma', 'states', 'calls', 'alpha', 'beta', 'delta', 'scaler', 'eprob')]
        _lib.check(_lib.load().gbrs_hmm_get(self._h, sample, c, *args))
        return {k: bufs[k] for k in want}

    def info(self):
        inf = _lib.HmmInfo()
        _lib.check(_lib.load().gbrs_hmm_info(self._h, C.byref(inf)))
        return inf


def get_chromosome_info(data_dir=None):
    """Chromosome order from $GBRS_DATA/ref.fa.fai (gbrs_utils.py:24-38)."""
    data_dir = os.getenv('GBRS_DATA', '.') if data_dir is None else data_dir
    fai_file = os.path.join(data_dir, 'ref.fa.fai')
    chr_lens = OrderedDict()
    try:
        with open(fai_file) as fh:
            for line in fh:
                item = line.rstrip('\n').split()
                if len(item) >= 2:
                    chr_lens[item[0][:8]] = int(item[1])
    except FileNotFoundError:
        raise ValueError('Make sure if $GBRS_DATA is set correctly, and that "ref.fa.fai" is in that '
                         f'directory. Currently it is: {data_dir}')
    return chr_lens


def reconstruct(expression_file: str, tprob_file: str, avec_file: str = None, gpos_file: str = None,
                expr_threshold: float = 1.5, sigma: float = 0.12, outbase: str = None,
                device: int = 0) -> None:
    """Reconstruct the genome based upon gene-level TPM quantities."""
    data_dir = os.getenv('GBRS_DATA', '.')
    if outbase is None:
        out_gtype = 'gbrs.reconstructed.genotypes.tsv'
        out_gprob = 'gbrs.reconstructed.genoprobs.npz'
    else:
        out_gtype = f'{outbase}.genotypes.tsv'
        out_gprob = f'{outbase}.genoprobs.npz'
    out_gtype_ordered = f'{os.path.splitext(out_gtype)[0]}.npz'
    if avec_file is None:
        avec_file = os.path.join(data_dir, 'avecs.npz')
    if gpos_file is None:
        gpos_file = os.path.join(data_dir, 'ref.gene_pos.ordered.npz')

    logger.info(f'Expression File: {expression_file}')
    logger.info(f'Transition Probabilities File: {tprob_file}')
    logger.info(f'Alignment Specificity File: {avec_file}')
    logger.info(f'Gene Position File: {gpos_file}')
    logger.info(f'Expression Threshold: {expr_threshold}')
    logger.info(f'Sigma: {sigma}')
    logger.info(f'Outbase: {outbase}')

    logger.info('Loading chromosome information')
    chrs = list(get_chromosome_info(data_dir).keys())

    logger.info(f'Loading alignment specificity: {avec_file}')
    avecs = np.load(avec_file)
    avec_keys = set(avecs.files)

    logger.info(f'Loading gene meta data: {gpos_file}')
    gene_pos = np.load(gpos_file)
    gid_genome_order = {}
    for c in gene_pos.files:
        arr = gene_pos[c]
        ids = [row[0] for row in arr]
        gid_genome_order[c] = [g.decode() if isinstance(g, bytes) else str(g) for g in ids]

    logger.info(f'Loading expression level data: {expression_file}')
    expr = {}
    with open(expression_file) as fh:
        haplotypes = fh.readline().rstrip().split('\t')[1:-1]
        for curline in fh:
            item = curline.rstrip().split('\t')
            expr[item[0]] = np.array(list(map(float, item[1:-1])))
    num_haps = len(haplotypes)
    genotypes = [h1 + h2 for h1, h2 in combinations_with_replacement(haplotypes, 2)]

    logger.info(f'Loading transition probabilities: {tprob_file}')
    tprob = np.load(tprob_file)
    tprob_keys = set(tprob.files)
    use = [c for c in chrs if c in tprob_keys]

    ex, av, ha, ng, tp = [], [], [], [], []
    for c in use:
        ids = gid_genome_order[c]
        ex.append(np.array([expr[g] for g in ids], dtype=np.float64).reshape(len(ids), num_haps))
        has = np.array([g in avec_keys for g in ids], dtype=np.uint8)
        a = np.zeros((len(ids), num_haps, num_haps))
        for i, g in enumerate(ids):
            if has[i]:
                a[i] = avecs[g]
        av.append(a)
        ha.append(has)
        ng.append(len(ids))
        tp.append(tprob[c])

    gamma, viterbi_states, gtcall_g = {}, {}, {}
    if use:
        hmm = DiplotypeHMM(num_haps, use, ng, tp, device=device)
        logger.info('Getting forward probability')
        hmm.set_expression(ex, av, ha, expr_threshold, sigma)
        logger.info('Getting backward probability')
        hmm.run()
        logger.info('Getting forward-backward probability')
        for ci, c in enumerate(use):
            res = hmm.get(ci)
            gamma[c] = res['gamma']
            viterbi_states[c] = [genotypes[s] for s in res['states']]
            for g, s in zip(gid_genome_order[c], res['calls']):
                if s >= 0:
                    gtcall_g[g] = genotypes[s]
        hmm.close()

    logger.info(f'Saving Reconstructed Genotype Probabilities: {out_gprob}')
    np.savez_compressed(out_gprob, **gamma)
    logger.info(f'Saving Reconstructed Genotypes: {out_gtype}')
    with open(out_gtype, 'w') as fhout:
        fhout.write('#Gene_ID\tDiplotype\n')
        for g in sorted(gtcall_g.keys()):
            fhout.write(f'{g}\t{gtcall_g[g]}\n')
    logger.info(f'Saving Reconstructed Ordered Genotypes: {out_gtype_ordered}')
    np.savez_compressed(out_gtype_ordered, **viterbi_states)
    logger.info('Done')
