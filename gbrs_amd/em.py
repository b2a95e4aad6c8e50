"""EMfactory: the reference's EM driver interface backed by libgbrs_hip.so.

Same constructor, method names, argument meaning, printed progress table and error behaviour as
emase/EMfactory.py:15-392 for the Model-4 path that `gbrs quantify` drives
(gbrs/emase_utils.py:282-316).  All arithmetic of prepare / run happens in HIP kernels through
the C ABI in include/gbrs_hip.h; this module never computes an EM quantity on the host and has
no CPU fallback.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

from . import _lib


class EMfactory:
    """A class that coordinates Expectation-Maximization (MI355X HIP path)."""

    def __init__(self, alignments, device: int = 0, merge_identical_rows: bool = False,
                 csc_layout: bool = False, extra_flags: int = 0, deterministic: bool = False,
                 one_shot: bool = False):
        self.probability = alignments
        self.grp_conv_mat = None          # kept for attribute parity; groups live in probability
        self.t2t_mat = None               # Models 1-3 only (EMfactory.py:48-59): never built
        self.target_lengths = None
        self.device = device
        self.flags = (_lib.GBRS_EM_MERGE_IDENTICAL_ROWS if merge_identical_rows else 0) | \
                     (_lib.GBRS_EM_LAYOUT_CSC if csc_layout else 0) | \
                     (_lib.GBRS_EM_DETERMINISTIC if deterministic else 0) | \
                     (_lib.GBRS_EM_ONE_SHOT if one_shot else 0) | int(extra_flags)   # tuning switches of gbrs_hip.h
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
        vals = getattr(apm, 'values', None)
        flags = self.flags | (_lib.GBRS_EM_KEEP_CSC if vals is not None else 0)
        allowed = getattr(apm, 'haplotype_mask', None)      # `-G`: the device drops the masked columns
        if allowed is not None:
            allowed = np.ascontiguousarray(allowed, dtype=np.uint32)
            _lib.check(lib.gbrs_em_create_masked(R, L, H, tab_p, tab_i, _lib.ptr(cnt), _lib.ptr(eff),
                                                 _lib.ptr(allowed), self.device, flags, C.byref(h)))
        else:
            _lib.check(lib.gbrs_em_create(R, L, H, tab_p, tab_i, _lib.ptr(cnt), _lib.ptr(eff),
                                          self.device, flags, C.byref(h)))
        self._h = h
        if vals is not None:
            # the file stores alignment values: they fix the starting point (EMfactory.py:95-98)
            vals = [np.ascontiguousarray(v, dtype=np.float64) for v in vals]
            _lib.check(lib.gbrs_em_set_initial_values(h, _lib.ptr_table(vals)))

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
            self.set_target_lengths(read_length_file(apm, lenfile, read_length))
        self.close()
        self._create()
        _lib.check(_lib.load().gbrs_em_prepare(self._h, float(pseudocount)))
        self._theta = None
        self._theta_dirty = False

    def set_target_lengths(self, lengths):
        """Effective lengths (H x L) as read_length_file returns them, with prepare()'s check (EMfactory.py:88-90)."""
        self.target_lengths = lengths
        if not np.all(self.target_lengths > 0.0):
            raise RuntimeError('There exist transcripts missing length information.')

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
        """Runs EM iterations (EMfactory.py:234-287): the whole loop, stopping rule included, executes
        on the device (gbrs_em_run); the host sees it once per batch of 8 iterations."""
        np.seterr(all='raise', under='ignore')      # the state the reference leaves numpy in
        self._check_model(model)
        self._require()
        self._push()
        cap = max(int(max_iters), 1)
        hist = np.zeros(cap, dtype=np.float64)
        stamps = np.zeros(cap, dtype=np.float64)
        n_it = C.c_int(0)
        _lib.check(_lib.load().gbrs_em_run(self._h, int(model), float(tol), int(max_iters), C.byref(n_it),
                                           _lib.ptr(hist), cap, _lib.ptr(stamps)))
        self.num_iters = int(n_it.value)
        self.err_history = hist[:self.num_iters].tolist()
        self._theta = None
        if verbose:
            _print_progress(self.err_history, stamps[:self.num_iters])

    # ------------------------------------------------------------------ reports
    report_pool = None      # a ReportPool: report_* hand their files to it instead of writing them before returning

    def _level(self, grp_wise, which):
        """(row names, H x n values) of theta (which=0) or of the expected read counts (which=1) at
        isoform or gene level."""
        apm = self.probability
        if grp_wise:
            return apm.gname, self._group_sums(which)
        return apm.lname, (self.allelic_expression if which == 0 else self.expected_read_counts())

    def report_read_counts(self, filename, grp_wise=False, reorder='as-is', notes=None):
        """Expected read counts of the last E-step as a TSV table (file format of EMfactory.py:289-331)."""
        names, values = self._level(grp_wise, 1)
        write_locus_table(filename, self.probability.hname, names, values, reorder, notes, pool=self.report_pool)

    def report_depths(self, filename, tpm=True, grp_wise=False, reorder='as-is', notes=None) -> None:
        """Depths (theta) as a TSV table, scaled to TPM on request (file format of EMfactory.py:333-380).
        Reference behaviour kept: at isoform level the TPM scaling is applied to allelic_expression
        itself, so it carries over to whatever is reported next (:352-354)."""
        names, values = self._level(grp_wise, 0)
        if tpm:
            values *= 1000000.0 / values.sum()
            if not grp_wise:
                self._theta_dirty = True
        write_locus_table(filename, self.probability.hname, names, values, reorder, notes, pool=self.report_pool)

    def export_posterior_probability(self, filename: str, title: str = 'Posterior Probability') -> None:
        """The reference saves with incidence_only=True (EMfactory.py:392 ->
        AlignmentPropertyMatrix.py:484), i.e. the structure without posterior values."""
        self.probability.save(filename, title=title)


def _print_progress(err_history, stamps):
    """The reference's progress table (EMfactory.py:259-262, :284-287); the time column holds the
    moment the host learned of the iteration (one stamp per device batch of 8)."""
    print('')
    print('Iter No  Time (hh:mm:ss)    Total change (TPM)  ')
    print('-------  ---------------  ----------------------')
    for k, (err, t) in enumerate(zip(err_history, stamps), start=1):
        secs = int(t)
        print(' %5d      %4d:%02d:%02d     %9.1f / 1000000' % (k, secs // 3600, secs // 60 % 60, secs % 60, err))


def read_length_file(apm, lenfile, read_length=100):
    """Effective lengths (H x L) from a `<locus>_<haplotype>TAB<length>` table (plain `<locus>` ids when
    there is one haplotype): max(length - read_length + 1, 1), what EMfactory.prepare builds at
    EMfactory.py:60-94.  As there, an id is cut at its underscore into exactly two parts, so a locus id
    that itself contains '_' is an error."""
    L, H = apm.num_loci, apm.num_haplotypes
    if H < 1:
        raise RuntimeError('There is something wrong with your emase-format alignment file.')
    eff = np.zeros((H, L))
    # plain files (every line `<known locus>_<known haplotype> TAB <number>`) are parsed natively; anything
    # else goes through the line-by-line path below, which raises what the reference raises
    with open(lenfile, 'rb') as fh:
        text = fh.read()
    name_blob, name_off = _blob_cached(apm.lname)
    hap_blob, hap_off = _blob(apm.hname)
    st = _lib.load().gbrs_parse_length_table(text, len(text), name_blob, _lib.ptr(name_off), L, hap_blob,
                                             _lib.ptr(hap_off), H, float(read_length), _lib.ptr(eff))
    if st == 0:
        return eff
    if st < 0:
        _lib.check(st)
    del text
    eff[:] = 0.0
    hap_row = {name: k for k, name in enumerate(apm.hname)}
    with open(lenfile) as fh:
        for line in fh:
            key, value = line.rstrip().split('\t')[:2]
            if H > 1:
                locus, hap = key.split('_')           # ValueError on a second underscore, as in the reference
                row = hap_row[hap]
            else:
                locus, row = key, 0
            eff[row, apm.lid[locus]] = max(float(value) - read_length + 1.0, 1.0)
    return eff


def _blob(strings):
    """(utf-8 bytes of the strings laid end to end, int64 offsets [n + 1])."""
    enc = [x.encode() for x in strings]
    off = np.zeros(len(enc) + 1, dtype=np.int64)
    np.cumsum(np.fromiter(map(len, enc), dtype=np.int64, count=len(enc)), out=off[1:])
    return b''.join(enc), off


_blob_cache = {}


def _blob_cached(names):
    """_blob of a name list, remembered per list object: the isoform (and gene) names are written into
    two reports each.  A numpy string array (the gene names) is laid out without touching its elements."""
    key = id(names)
    hit = _blob_cache.get(key)
    if hit is None or hit[0] is not names:
        made = None
        if isinstance(names, np.ndarray) and names.dtype.kind == 'U' and names.ndim == 1:
            made = _blob_of_unicode_array(names)
        if made is None:
            made = _blob([str(x) for x in names])
        if len(_blob_cache) > 4:
            _blob_cache.clear()
        hit = _blob_cache[key] = (names, made)
    return hit[1]


def _blob_of_unicode_array(arr):
    """(bytes, offsets) of an ASCII 'U' array from its code points, None when a name is not plain ASCII (or holds
    an inner NUL, which the fixed-width storage cannot tell from padding)."""
    n, width = len(arr), arr.dtype.itemsize // 4
    if n == 0 or width == 0:
        return b'', np.zeros(n + 1, dtype=np.int64)
    code = np.frombuffer(np.ascontiguousarray(arr).tobytes(), dtype=np.uint32).reshape(n, width)
    if (code > 127).any():
        return None
    used = code != 0
    length = used.sum(axis=1)
    if (used != (np.arange(width)[None, :] < length[:, None])).any():
        return None
    off = np.zeros(n + 1, dtype=np.int64)
    np.cumsum(length, out=off[1:])
    return code[used].astype(np.uint8).tobytes(), off


def write_locus_table(filename, hap_names, row_names, values, reorder='as-is', notes=None, pool=None):
    """`locus <haplotypes...> total [notes]` table: one line per locus (or gene) with the per-haplotype
    values, their sum and, when `notes` maps names to text, that text.  Numbers are written in their
    shortest round-trip form, which is what str(numpy.float64) gives in the reference's writers; the
    text is produced by libgbrs_hip's gbrs_write_locus_table (1.5 M numbers per sample at DO size).
    `pool` (a ReportPool) takes the formatting and writing onto a thread: the caller goes on to the next report and
    collects errors with pool.finish(); `values` must not change until then."""
    values = np.asarray(values, dtype=np.float64)
    totals = values.sum(axis=0)
    order = None
    if reorder in ('increasing', 'decreasing'):
        order = np.argsort(totals.ravel())
        if reorder == 'decreasing':
            order = order[::-1]
        order = np.ascontiguousarray(order, dtype=np.int64)
    elif reorder != 'as-is':
        raise ValueError(f'unknown reorder option: {reorder}')
    if not (values.flags.c_contiguous or values.flags.f_contiguous):
        values = np.ascontiguousarray(values)
    totals = np.ascontiguousarray(totals.ravel(), dtype=np.float64)
    n_haps, n_rows = values.shape
    item = values.itemsize
    names = row_names if isinstance(row_names, (list, np.ndarray)) else [str(x) for x in row_names]
    if len(names) != n_rows:
        raise RuntimeError('The number of names does not match to the matrix shape.')
    name_blob, name_off = _blob_cached(names)
    note_blob = note_off = None
    if notes is not None:
        aligned = notes.aligned_blob(row_names) if hasattr(notes, 'aligned_blob') else None
        note_blob, note_off = aligned if aligned is not None else _blob([str(notes[n]) for n in names])
    head = '\t'.join(['locus', *hap_names, 'total'] + (['notes'] if notes is not None else [])) + '\n'

    def emit():       # every array the call reads is held by this closure
        _lib.check(_lib.load().gbrs_write_locus_table(
            os.fsencode(filename), head.encode(), _lib.ptr(values), n_rows, n_haps, values.strides[1] // item,
            values.strides[0] // item, _lib.ptr(totals), name_blob, _lib.ptr(name_off), note_blob,
            _lib.ptr(note_off) if note_off is not None else None, _lib.ptr(order) if order is not None else None))
    if pool is None:
        emit()
    else:
        pool.submit(emit)


class ReportPool:
    """Report files written behind the caller's back: `gbrs quantify` leaves four tables of 10^5 lines each, and the
    numbers of the next one can be fetched from the device while the previous one is formatted and written."""

    def __init__(self):
        from concurrent.futures import ThreadPoolExecutor
        self._pool = ThreadPoolExecutor(max_workers=4)
        self._pending = []

    def submit(self, fn):
        self._pending.append(self._pool.submit(fn))

    def finish(self):
        """Wait for every table; the first failure is raised here."""
        pending, self._pending = self._pending, []
        try:
            for f in pending:
                f.result()
        finally:
            self._pool.shutdown(wait=True)
