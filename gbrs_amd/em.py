"""EMfactory: the reference's EM driver interface backed by libgbrs_hip.so.

Same constructor, method names, argument meaning, printed progress table and error behaviour as
emase/EMfactory.py:15-392 for the Model-4 path that `gbrs quantify` drives
(gbrs/emase_utils.py:282-316).  All arithmetic of prepare / run happens in HIP kernels through
the C ABI in include/gbrs_hip.h; this module never computes an EM quantity on the host and has
no CPU fallback.
"""
from __future__ import annotations

import ctypes as C
import time

import numpy as np

from . import _lib


class EMfactory:
    """A class that coordinates Expectation-Maximization (MI355X HIP path)."""

    def __init__(self, alignments, device: int = 0, merge_identical_rows: bool = False,
                 csc_layout: bool = False, extra_flags: int = 0):
        self.probability = alignments
        self.grp_conv_mat = None          # kept for attribute parity; groups live in probability
        self.t2t_mat = None               # Models 1-3 only (EMfactory.py:48-59): never built
        self.target_lengths = None
        self.device = device
        self.flags = (_lib.GBRS_EM_MERGE_IDENTICAL_ROWS if merge_identical_rows else 0) | \
                     (_lib.GBRS_EM_LAYOUT_CSC if csc_layout else 0) | int(extra_flags)   # tuning switches of gbrs_hip.h
        self._h = None
        self._theta = None                # host copy of allelic_expression (H x L)
        self._theta_dirty = False         # host copy edited, device not yet updated
        self.num_iters = 0
        self.err_history = []

    # ------------------------------------------------------------------ handle management
    def _create(self):
        lib = _lib.load()
        apm = self.probability
        L, H, R = apm.shape
        tab_p = _lib.ptr_table(apm.indptr)
        tab_i = _lib.ptr_table(apm.indices)
        h = C.c_void_p()
        eff = None
        if self.target_lengths is not None:
            eff = np.ascontiguousarray(self.target_lengths, dtype=np.float64)
        cnt = None if apm.count is None else np.ascontiguousarray(apm.count, dtype=np.float64)
        _lib.check(lib.gbrs_em_create(R, L, H, tab_p, tab_i, _lib.ptr(cnt), _lib.ptr(eff),
                                      self.device, self.flags, C.byref(h)))
        self._h = h

    def close(self):
        if self._h is not None:
            _lib.load().gbrs_em_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def info(self):
        inf = _lib.EmInfo()
        _lib.check(_lib.load().gbrs_em_info(self._h, C.byref(inf)))
        return inf

    # ------------------------------------------------------------------ reference interface
    def prepare(self, pseudocount: float = 0.0, lenfile: str = None, read_length: int = 100) -> None:
        """Initializes the probability of read origin according to the alignment profile
        (EMfactory.py:27-111)."""
        apm = self.probability
        if lenfile is not None:
            self.target_lengths = read_length_file(apm, lenfile, read_length)
            if not np.all(self.target_lengths > 0.0):
                raise RuntimeError('There exist transcripts missing length information.')
        self.close()
        self._create()
        _lib.check(_lib.load().gbrs_em_prepare(self._h, float(pseudocount)))
        self._theta = None
        self._theta_dirty = False

    def _require(self):
        if self._h is None:
            raise RuntimeError('prepare() has not been called.')

    def _push(self):
        if self._theta_dirty:
            _lib.check(_lib.load().gbrs_em_set_theta(self._h, _lib.ptr(np.ascontiguousarray(self._theta))))
            self._theta_dirty = False

    @property
    def allelic_expression(self):
        self._require()
        if self._theta is None:
            L, H, R = self.probability.shape
            out = np.empty((H, L), dtype=np.float64)
            _lib.check(_lib.load().gbrs_em_get(self._h, _lib.ptr(out), None))
            self._theta = out
        return self._theta

    @allelic_expression.setter
    def allelic_expression(self, value):
        self._require()
        L, H, R = self.probability.shape
        self._theta = np.ascontiguousarray(value, dtype=np.float64).reshape(H, L)
        self._theta_dirty = True

    def expected_read_counts(self):
        """probability.sum(axis=READ) of the last E-step (EMfactory.py:302), (H x L)."""
        self._require()
        L, H, R = self.probability.shape
        out = np.empty((H, L), dtype=np.float64)
        _lib.check(_lib.load().gbrs_em_get(self._h, None, _lib.ptr(out)))
        return out

    def _group_sums(self, which):
        apm = self.probability
        if not apm.num_groups:
            raise RuntimeError('No group information is available.')
        self._push()
        gptr, mem = apm.group_csr()
        H = apm.num_haplotypes
        out = np.empty((H, apm.num_groups), dtype=np.float64)
        _lib.check(_lib.load().gbrs_em_group_sums(self._h, apm.num_groups, _lib.ptr(gptr), _lib.ptr(mem),
                                                  which, _lib.ptr(out)))
        # scipy hands the reference this product as the transpose of a C-ordered (G x H) array;
        # keep that memory order so a later full `.sum()` adds in the same sequence
        return np.asfortranarray(out)

    def get_allelic_expression(self, at_group_level: bool = False):
        if at_group_level:
            return self._group_sums(0)
        return self.allelic_expression.copy()

    def update_allelic_expression(self, model: int = 4) -> None:
        """A single EM step (EMfactory.py:214-232)."""
        self._check_model(model)
        self._require()
        self._push()
        _lib.check(_lib.load().gbrs_em_step(self._h, 1, None))
        self._theta = None

    @staticmethod
    def _check_model(model):
        if model not in (1, 2, 3, 4):
            raise RuntimeError('The read normalization model should be 1, 2, 3, or 4.')
        if model != 4:
            raise RuntimeError(f'Multiread model {model} is not implemented by the MI355X path '
                               '(only Model 4: Gene*Isoform*Allele).')

    def run(self, model: int, tol: float = 0.001, max_iters: int = 999, verbose: bool = True) -> None:
        """Runs EM iterations (EMfactory.py:234-287)."""
        # the reference leaves numpy in this error state after run()
        np.seterr(all='raise')
        np.seterr(under='ignore')
        self._check_model(model)
        self._require()
        self._push()
        lib = _lib.load()
        if verbose:
            print('')
            print('Iter No  Time (hh:mm:ss)    Total change (TPM)  ')
            print('-------  ---------------  ----------------------')
        time0 = time.time()
        n_it = C.c_int(0)
        cap = max(int(max_iters), 1)
        hist = np.zeros(cap, dtype=np.float64)
        _lib.check(lib.gbrs_em_run(self._h, int(model), float(tol), int(max_iters), C.byref(n_it),
                                   _lib.ptr(hist), cap))
        self.num_iters = int(n_it.value)
        self.err_history = [float(x) for x in hist[:self.num_iters]]
        self._theta = None
        if verbose:
            # the device loop does not hand control back per iteration; elapsed time is the run's
            delmin, s = divmod(int(time.time() - time0), 60)
            h, m = divmod(delmin, 60)
            for i, err_sum in enumerate(self.err_history):
                print(' %5d      %4d:%02d:%02d     %9.1f / 1000000' % (i + 1, h, m, s, err_sum))

    # ------------------------------------------------------------------ reports
    def report_read_counts(self, filename, grp_wise=False, reorder='as-is', notes=None):
        """Export read counts (EMfactory.py:289-331)."""
        if grp_wise:
            lname = self.probability.gname
            expected_read_counts = self._group_sums(1)
        else:
            lname = self.probability.lname
            expected_read_counts = self.expected_read_counts()
        total_read_counts = expected_read_counts.sum(axis=0)
        _write_report(filename, self.probability.hname, lname, expected_read_counts, total_read_counts,
                      reorder, notes)

    def report_depths(self, filename, tpm=True, grp_wise=False, reorder='as-is', notes=None) -> None:
        """Exports expected depths (EMfactory.py:333-380).  As in the reference, tpm=True at the
        isoform level rescales allelic_expression itself."""
        if grp_wise:
            lname = self.probability.gname
            depths = self._group_sums(0)
        else:
            lname = self.probability.lname
            depths = self.allelic_expression
        if tpm:
            depths *= 1000000.0 / depths.sum()
            if not grp_wise:
                self._theta_dirty = True
        total_depths = depths.sum(axis=0)
        _write_report(filename, self.probability.hname, lname, depths, total_depths, reorder, notes)

    def export_posterior_probability(self, filename: str, title: str = 'Posterior Probability') -> None:
        """The reference saves with incidence_only=True (EMfactory.py:392 ->
        AlignmentPropertyMatrix.py:484), i.e. the structure without posterior values."""
        self.probability.save(filename, title=title)


def read_length_file(apm, lenfile, read_length=100):
    """target_lengths (H x L) as EMfactory.prepare builds it (EMfactory.py:60-94)."""
    hid = dict(zip(apm.hname, np.arange(len(apm.hname))))
    tl = np.zeros((apm.num_loci, apm.num_haplotypes))
    if apm.num_haplotypes > 1:
        with open(lenfile) as fh:
            for curline in fh:
                item = curline.rstrip().split('\t')
                locus, hap = item[0].split('_')
                tl[apm.lid[locus], hid[hap]] = max(float(item[1]) - read_length + 1.0, 1.0)
    elif apm.num_haplotypes > 0:
        with open(lenfile) as fh:
            for curline in fh:
                item = curline.rstrip().split('\t')
                tl[apm.lid[item[0]], 0] = max(float(item[1]) - read_length + 1.0, 1.0)
    else:
        raise RuntimeError('There is something wrong with your emase-format alignment file.')
    return np.ascontiguousarray(tl.transpose())


def _write_report(filename, hname, lname, values, totals, reorder, notes):
    if reorder == 'decreasing':
        report_order = np.argsort(totals.flatten())[::-1]
    elif reorder == 'increasing':
        report_order = np.argsort(totals.flatten())
    elif reorder == 'as-is':
        report_order = np.arange(len(lname))
    cntdata = np.vstack((values, totals))
    with open(filename, 'w') as fhout:
        fhout.write('locus\t' + '\t'.join(hname) + '\ttotal')
        if notes is not None:
            fhout.write('\tnotes')
        fhout.write('\n')
        for locus_id in report_order:
            lname_cur = lname[locus_id]
            fhout.write('\t'.join([lname_cur] + list(map(str, cntdata[:, locus_id].ravel()))))
            if notes is not None:
                fhout.write(f'\t{notes[lname_cur]}')
            fhout.write('\n')
