"""`gbrs reconstruct` on MI355X: same inputs, outputs and defaults as
gbrs_utils.reconstruct (gbrs/gbrs_utils.py:382-609); the emission model, forward, backward,
posterior, Viterbi and backtrace all run in HIP kernels through include/gbrs_hip.h.
No CPU fallback: without libgbrs_hip.so / a gfx950 device every entry point raises.
"""
from __future__ import annotations

import ctypes as C
import logging
import os
import time
from collections import OrderedDict
from itertools import combinations_with_replacement

import numpy as np

from . import _lib
from .npzfast import FastNpz, savez_compressed

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

    def set_expression(self, expr, avecs=None, has_avec=None, expr_threshold=1.5, sigma=0.12):
        """expr[c] [n_samples, n_c, H] (or [n_c, H]); avecs[c] [n_c, H, H]; has_avec[c] bool [n_c].
        avecs / has_avec are sample independent and stay on the device once given: pass them with the
        first sample (or batch) and leave them out afterwards."""
        if (avecs is None) != (has_avec is None):
            raise ValueError('avecs and has_avec go together')
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
            ex.append(e)
            if avecs is not None:
                a = np.ascontiguousarray(avecs[c], dtype=np.float64)
                if a.shape != (n, self.H, self.H):
                    raise ValueError(f'avecs[{c}] has shape {a.shape}')
                av.append(a)
                ha.append(np.ascontiguousarray(has_avec[c], dtype=np.uint8))
        _lib.check(_lib.load().gbrs_hmm_set_expression(
            self._h, ns, _lib.ptr_table(ex), _lib.ptr_table(av) if av else None,
            _lib.ptr_table(ha) if ha else None, float(expr_threshold), float(sigma)))
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
    """Chromosome names (cut to 8 characters) -> lengths, in the order of $GBRS_DATA/ref.fa.fai: the
    order in which `gbrs reconstruct` walks the genome (gbrs_utils.py:24-38)."""
    data_dir = os.getenv('GBRS_DATA', '.') if data_dir is None else data_dir
    fai = os.path.join(data_dir, 'ref.fa.fai')
    if not os.path.isfile(fai):
        raise ValueError('Make sure if $GBRS_DATA is set correctly, and that "ref.fa.fai" is in that '
                         f'directory. Currently it is: {data_dir}')
    lengths = OrderedDict()
    with open(fai) as fh:
        for fields in map(str.split, fh):
            if len(fields) >= 2:
                lengths[fields[0][:8]] = int(fields[1])
    return lengths


def read_gene_tpm(expression_file):
    """(haplotype letters, {gene id: row}, TPM matrix [genes x H]) from a `.genes.tpm` report.  The first
    column is the gene id and the last the total; `gbrs reconstruct` assumes there is no notes column,
    i.e. the multiway report (gbrs_utils.py:450-459).  The numbers of the whole file go
    through one C-level parse instead of a float() per cell."""
    with open(expression_file) as fh:
        haplotypes = fh.readline().rstrip().split('\t')[1:-1]
        body = fh.read()
    lines = body.splitlines()
    width = len(haplotypes) + 1
    if lines and len(lines) == body.count('\n') + (0 if body.endswith('\n') else 1):
        # plain table: the library parses the numbers (std::from_chars, ~10 ns each against strtod's ~100)
        raw = body.encode()
        table = np.empty((len(lines), width), dtype=np.float64)
        try:
            status = _lib.load().gbrs_parse_number_table(raw, len(raw), len(lines), width, _lib.ptr(table))
        except (ImportError, OSError):
            status = 1
        if status == 0:
            ids = [line.partition('\t')[0] for line in lines]
            return haplotypes, {g: k for k, g in enumerate(ids)}, table[:, :-1]
    ids, cells = [], []
    for line in lines:
        gid, _, rest = line.rstrip().partition('\t')
        ids.append(gid)
        cells.append(rest)
    flat = np.fromstring('\t'.join(cells), dtype=np.float64, sep='\t') if cells else np.zeros(0)
    if flat.size != len(ids) * width:
        raise ValueError(f'{expression_file}: every line must hold {width} numbers after the gene id')
    table = flat.reshape(len(ids), width)[:, :-1]
    return haplotypes, {g: k for k, g in enumerate(ids)}, table


def read_gene_order(gpos_file):
    """{chromosome: [gene ids in genome order]} from `ref.gene_pos.ordered.npz`, whose arrays hold
    (gene id, position) records with the id as bytes or str (gbrs_utils.py:437-447)."""
    order = {}
    z = FastNpz(gpos_file)
    for c in z.files:
        a = z[c]
        if a.dtype.names:                                   # structured records: first field is the id
            col = a[a.dtype.names[0]]
            order[c] = col.astype('U').tolist()
        else:
            order[c] = [gid.decode() if isinstance(gid, bytes) else str(gid) for gid, *_ in a]
    z.close()
    return order


class ReconstructContext:
    """Everything `gbrs reconstruct` reads that does not depend on the sample - genome order, gene order, transition
    tables, specificity blocks - and the device handle that holds the tables.  One command builds one and drops it; a
    resident process (gbrs_amd.worker) keeps it across samples, so that a sample costs its expression rows only."""

    def __init__(self, tprob_file, avec_file=None, gpos_file=None, device=0):
        data_dir = os.getenv('GBRS_DATA', '.')
        self.tprob_file = tprob_file
        self.avec_file = avec_file or os.path.join(data_dir, 'avecs.npz')
        self.gpos_file = gpos_file or os.path.join(data_dir, 'ref.gene_pos.ordered.npz')
        self.device = device
        self.data_dir = data_dir
        self.loaded = False
        self.hmm = None
        self.num_haps = None

    def load(self, marks=None):
        """The files (the big transition tables inflate on the library's threads while the small files are parsed)."""
        if self.loaded:
            return
        logger.info('Loading chromosome information')
        genome = list(get_chromosome_info(self.data_dir))
        from concurrent.futures import ThreadPoolExecutor
        tprob = FastNpz(self.tprob_file)
        self.chroms = [c for c in genome if c in tprob]              # chromosomes without a table are skipped (:495)
        reader = ThreadPoolExecutor(max_workers=1)
        tables_pending = reader.submit(tprob.read_many, self.chroms)
        logger.info(f'Loading alignment specificity: {self.avec_file}')
        self.avecs = FastNpz(self.avec_file)
        logger.info(f'Loading gene meta data: {self.gpos_file}')
        self.gene_order = read_gene_order(self.gpos_file)
        self._tables_pending, self._reader = tables_pending, reader
        self.loaded = True

    def tables(self):
        if self._tables_pending is not None:
            logger.info(f'Loading transition probabilities: {self.tprob_file}')
            self._tables = self._tables_pending.result()             # stored members: views of the page cache, no copy
            self._reader.shutdown(wait=False)
            self._tables_pending = None
        return self._tables

    def specificity(self, num_haps):
        """Per chromosome (blocks [n, H, H], present flags [n]): sample independent, made once."""
        if getattr(self, '_spec', None) is None or self.num_haps != num_haps:
            spec = []
            for c in self.chroms:
                ids = self.gene_order[c]
                n = len(ids)
                present = np.fromiter((g in self.avecs for g in ids), dtype=np.uint8, count=n)
                blocks = np.zeros((n, num_haps, num_haps), dtype=np.float64)
                have = np.flatnonzero(present)
                if len(have):
                    blocks[have] = self.avecs.stack([ids[k] for k in have], (num_haps, num_haps))
                spec.append((blocks, present))
            self._spec, self.num_haps = spec, num_haps
            self.avecs.close()
        return self._spec

    def close(self):
        if self.hmm is not None:
            self.hmm.close()
            self.hmm = None


def reconstruct(expression_file: str, tprob_file: str, avec_file: str = None, gpos_file: str = None,
                expr_threshold: float = 1.5, sigma: float = 0.12, outbase: str = None,
                device: int = 0, stage_times: dict = None, context: ReconstructContext = None) -> None:
    """`gbrs reconstruct`: diplotype posteriors and Viterbi calls along every chromosome from
    gene-level TPMs.  Same inputs, defaults and three output files as gbrs_utils.reconstruct
    (gbrs_utils.py:382-609); the emission model and the three recursions run on the device.
    `stage_times` (optional dict) receives wall-clock seconds per stage.  `context` (extension,
    gbrs_amd.worker): the sample-independent inputs and the device handle of an earlier call on the same
    tables - the transition and specificity tables then stay where they are and the sample moves its expression
    rows only."""
    clock = time.perf_counter
    marks = stage_times if stage_times is not None else {}
    stem = 'gbrs.reconstructed' if outbase is None else outbase
    out_calls, out_post, out_path = f'{stem}.genotypes.tsv', f'{stem}.genoprobs.npz', f'{stem}.genotypes.npz'
    ctx = context if context is not None else ReconstructContext(tprob_file, avec_file, gpos_file, device)
    for label, value in (('Expression File', expression_file), ('Transition Probabilities File', tprob_file),
                         ('Alignment Specificity File', ctx.avec_file), ('Gene Position File', ctx.gpos_file),
                         ('Expression Threshold', expr_threshold), ('Sigma', sigma), ('Outbase', outbase)):
        logger.info(f'{label}: {value}')

    _lib.warm_up_device_async(device)      # HIP start-up overlaps with reading the files
    t0 = clock()
    # the transition tables are the big read (0.4 GB of deflate streams at DO size): it starts on the library's
    # threads and is collected when the tables are needed, after the small files have been parsed
    ctx.load()
    chroms, gene_order = ctx.chroms, ctx.gene_order
    logger.info(f'Loading expression level data: {expression_file}')
    haplotypes, expr_row, expr_table = read_gene_tpm(expression_file)
    num_haps = len(haplotypes)
    diplotypes = [a + b for a, b in combinations_with_replacement(haplotypes, 2)]
    rows = []
    for c in chroms:
        ids = gene_order[c]
        r = expr_table[[expr_row[g] for g in ids]] if len(ids) else np.zeros((0, num_haps))   # KeyError: gene without TPM
        rows.append(np.ascontiguousarray(r, dtype=np.float64).reshape(len(ids), num_haps))
    first_use = ctx.hmm is None or ctx.hmm.H != num_haps
    spec = ctx.specificity(num_haps) if first_use else None
    tables = ctx.tables() if first_use else None
    marks['load'] = clock() - t0

    posterior, path_names, calls = {}, {}, {}
    if chroms:
        t0 = clock()
        if first_use:
            ctx.close()
            ctx.hmm = DiplotypeHMM(num_haps, chroms, [len(gene_order[c]) for c in chroms], tables, device=device)
        hmm = ctx.hmm
        marks['tables_to_device'] = clock() - t0
        t0 = clock()
        logger.info('Getting forward probability')
        if first_use:
            hmm.set_expression(rows, [x[0] for x in spec], [x[1] for x in spec], expr_threshold, sigma)
        else:
            hmm.set_expression(rows, expr_threshold=expr_threshold, sigma=sigma)
        logger.info('Getting backward probability')
        hmm.run()
        logger.info('Getting forward-backward probability')
        for k, c in enumerate(chroms):
            res = hmm.get(k)
            posterior[c] = res['gamma']
            path_names[c] = [diplotypes[s] for s in res['states']]
            calls.update((gid, diplotypes[s]) for gid, s in zip(gene_order[c], res['calls']) if s >= 0)
        if context is None:
            ctx.close()
        marks['hmm'] = clock() - t0

    t0 = clock()
    logger.info(f'Saving Reconstructed Genotype Probabilities: {out_post}')
    savez_compressed(out_post, posterior)
    logger.info(f'Saving Reconstructed Genotypes: {out_calls}')
    with open(out_calls, 'w') as out:
        out.write('#Gene_ID\tDiplotype\n')
        out.writelines(f'{gid}\t{calls[gid]}\n' for gid in sorted(calls))
    logger.info(f'Saving Reconstructed Ordered Genotypes: {out_path}')
    savez_compressed(out_path, {c: np.asarray(v) for c, v in path_names.items()})
    marks['save'] = clock() - t0
    logger.info('Done')
