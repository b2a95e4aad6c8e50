/*
 * gbrs_hip.h - C ABI of libgbrs_hip.so: the MI355X (gfx950) implementation of the two
 * numeric hot paths of churchill-lab/gbrs.
 *
 * The reference has no FFI seam (it is pure Python); the entry points below are cut at the
 * seams its own callers use, so a maintainer can bind them with ctypes and
 * keep `gbrs quantify` / `gbrs reconstruct` unchanged above this line.  All citations are
 * relative to /root/reference/src/gbrs/.
 *
 *   EM  (gbrs quantify -M 4):   emase/EMfactory.py:20-287 as driven from
 *                                gbrs/emase_utils.py:282-316
 *   HMM (gbrs reconstruct):     gbrs/gbrs_utils.py:463-599
 *
 * Conventions
 *   - plain pointers and sizes only; the caller owns every host buffer, the library copies in
 *     during the call and never keeps a host pointer after it returns;
 *   - every function returns 0 on success and a negative gbrs_status on failure, with the text
 *     available from gbrs_last_error() (thread-local).  The Python shim turns that into the
 *     RuntimeError / FloatingPointError the reference would have raised;
 *   - calls are blocking and synchronous unless the name ends in _async; one HIP stream per
 *     handle; different handles may be driven from different threads (ctypes drops the GIL);
 *   - matrices named "H x L" are row-major with the haplotype index slowest, exactly the
 *     ndarray layout of EMfactory.allelic_expression.
 */
#ifndef GBRS_HIP_H
#define GBRS_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GBRS_ABI_VERSION 5

enum gbrs_status {
    GBRS_OK = 0,
    GBRS_ERR_INVALID = -1,      /* bad argument (RuntimeError in the reference)            */
    GBRS_ERR_HIP = -2,          /* HIP runtime failure, message carries hipGetErrorString  */
    GBRS_ERR_NO_DEVICE = -3,    /* no gfx950 device visible: there is NO CPU fallback      */
    GBRS_ERR_FLOAT = -4,        /* 0/0 or overflow where the reference's np.seterr(all='raise')
                                   (EMfactory.py:256) raises FloatingPointError              */
    GBRS_ERR_UNSUPPORTED = -5,  /* models 1-3 (EMfactory.py:160-203): out of scope         */
    GBRS_ERR_STATE = -6         /* call order violated (e.g. run before prepare)           */
};

const char *gbrs_last_error(void);
int gbrs_abi_version(void);
/* Number of visible HIP devices, or a negative status. */
int gbrs_device_count(void);
/* Optional: pays the process's one-time device costs now (runtime start-up, context, loading the library's code
 * objects, the first small copy) - about 0.1-0.3 s that otherwise land in the first create call.  The Python host
 * calls it on a thread while the input files are read. */
int gbrs_warm_up(int device);

/* ------------------------------------------------------------------------------------------
 * EM: EMASE Model 4 over the alignment incidence tensor.
 * ---------------------------------------------------------------------------------------- */

typedef struct gbrs_em gbrs_em_t;

/* flags for gbrs_em_create* */
#define GBRS_EM_DEFAULT 0u
/* Merge rows with identical alignment patterns into one weighted row while building the device
 * layout (what `gbrs compress`, gbrs/emase_utils.py:60-103, does as a separate command).  Same
 * fixed point; off by default so that every input row is processed every iteration. */
#define GBRS_EM_MERGE_IDENTICAL_ROWS 1u
/* Keep the reference's CSC arrays as the device layout (two passes with global float64 atomics).
 * The default is the packed-row-tile layout; this one is the simple cross-check. */
#define GBRS_EM_LAYOUT_CSC 2u
/* Tuning switches for the order of rows inside a tile.
 * Stream order (default): rows are grouped by length and each group of 64/len lanes walks its own
 * contiguous run of the tile's sorted rows over successive batches, so a lane stays on one locus
 * list for long stretches and the per-lane register accumulation rarely spills to LDS atomics.
 * GBRS_EM_NO_STREAMS falls back to the previous defaults: sorted order for raw reads, interleaved
 * order (locus lists dealt round-robin across the 64 lanes of a batch) when `count` is given or
 * rows are merged; GBRS_EM_NO_INTERLEAVE / GBRS_EM_FORCE_INTERLEAVE select among those two. */
#define GBRS_EM_NO_INTERLEAVE 4u
#define GBRS_EM_FORCE_INTERLEAVE 8u
#define GBRS_EM_NO_STREAMS 16u
/* Bit-reproducible sums: the E-step's LDS float64 atomics are replaced by a fixed-order reduction
 * (every wavefront of a tile owns a private copy of the tile's partial sums; lanes that hand in sums
 * for one locus are added by a fixed tree; copies, slots and block sums are added in index order), so
 * two runs on the same input give bit-identical theta and therefore the same iteration count even when
 * err_sum lands next to 1e6*tol (EMfactory.py:266).  The tiles are cut smaller (at most
 * (4160/8 - 1)/H loci each); not available with GBRS_EM_LAYOUT_CSC or H > 16. */
#define GBRS_EM_DETERMINISTIC 32u
/* Keep the uploaded CSC arrays on the handle until gbrs_em_set_initial_values has run (files that
 * store alignment values other than 1). */
#define GBRS_EM_KEEP_CSC 64u
/* This handle is one of two locus ranges of one sample that run side by side on the device (the two engines of
 * a rank in gbrs_amd/dist.py PipelinedShardedEM): the layout sizes its tiles for the two together. */
#define GBRS_EM_SIDE_BY_SIDE 128u
/* The process builds this one handle and exits (the `gbrs quantify` command): the multi-gigabyte temporaries of the
 * layout build and the uploaded CSC copy stay allocated until gbrs_em_destroy instead of being freed when create
 * returns (on some hosts a hipMalloc that follows a large hipFree stalls for 0.1-0.2 s; the whole create is ~45 ms
 * shorter).  gbrs_em_info.retained_build_bytes reports what is held.  Without the flag create frees them in one pass
 * before it returns, so several live handles cost their layouts only. */
#define GBRS_EM_ONE_SHOT 256u
/* Locus sets off.  By default a read whose alignments to several loci all carry the same haplotype mask may be stored
 * as one word on the id of its locus SET: its denominator is sum_h m_h * (theta[l1,h] + theta[l2,h] + ...) and every
 * member locus receives the same count/den, so the set behaves like one locus whose theta is the sum of its members'.
 * The sets exist inside the tiles only (a tile sums the members' theta for a set entry and stores the entry's sums once
 * per member locus); theta, the expected counts and the M-step are per locus as before.  Same arithmetic up to the
 * association of those sums (agrees with the plain form to ~1e-15 relative).  The layout takes the sets when they
 * remove >= 15 % of the words, leave >= 100 words per id and number at most twice the loci, and never for weighted rows
 * (count given or identical rows merged); gbrs_em_info.num_locus_sets says how many it found (0: not taken).  The flag keeps one word per
 * (read, locus) pair whatever the sample. */
#define GBRS_EM_NO_LOCUS_SETS 512u

/*
 * Replaces: AlignmentPropertyMatrix(h5file=...) as consumed by EMfactory.__init__
 * (emase/Sparse3DMatrix.py:80-92 CSC branch, emase/AlignmentPropertyMatrix.py:72-73 count,
 * emase/EMfactory.py:60-94 target_lengths).
 *
 *   num_rows  R reads / equivalence classes        num_loci  L isoforms      num_haps  H (<= 32)
 *   indptr[h]   uint32[L+1]  column pointers of haplotype h's (R x L) CSC incidence matrix
 *   indices[h]  uint32[nnz_h] row (read) ids, any order inside a column, no duplicates
 *   count       double[R] EC multiplicities, or NULL (every row counts once)
 *   eff_len     double[H*L] row-major max(len - read_length + 1, 1), or NULL (no length model)
 *   device      HIP device ordinal
 */
int gbrs_em_create(uint64_t num_rows, uint32_t num_loci, uint32_t num_haps,
                   const uint32_t *const *indptr, const uint32_t *const *indices,
                   const double *count, const double *eff_len,
                   int device, uint32_t flags, gbrs_em_t **out);

/* Same, but every array pointer (indptr[h], indices[h], count, eff_len) is a DEVICE pointer on
 * `device`; the pointer tables indptr/indices themselves are host arrays of H device pointers.
 * The inputs are only read during the call. */
int gbrs_em_create_device(uint64_t num_rows, uint32_t num_loci, uint32_t num_haps,
                          const uint32_t *const *indptr, const uint32_t *const *indices,
                          const double *count, const double *eff_len,
                          int device, uint32_t flags, gbrs_em_t **out);

/*
 * The same with the `-G` genotype mask of `gbrs quantify` applied on the device.  Replaces
 * gbrs/emase_utils.py:240-273: `aln_mat.multiply(gtmask, axis=2)` followed by `eliminate_zeros()` per haplotype,
 * i.e. every stored entry (h, l, r) whose haplotype h is not one of the two called for the gene of locus l leaves
 * the structure before EMfactory sees it (rows left without entries drop out, as they do there).
 *   allowed  HOST uint32[L] in both variants, bit h set = entries of (haplotype h, locus l) stay; NULL = no mask.
 * The mask selects whole CSC columns, so the library uploads the arrays as they are and moves the surviving
 * columns together on the device; no host pass over the entries.  gbrs_em_info.num_entries is the masked count;
 * values given to gbrs_em_set_initial_values still line up with the indices arrays as passed here.
 */
int gbrs_em_create_masked(uint64_t num_rows, uint32_t num_loci, uint32_t num_haps,
                          const uint32_t *const *indptr, const uint32_t *const *indices,
                          const double *count, const double *eff_len, const uint32_t *allowed,
                          int device, uint32_t flags, gbrs_em_t **out);
int gbrs_em_create_masked_device(uint64_t num_rows, uint32_t num_loci, uint32_t num_haps,
                                 const uint32_t *const *indptr, const uint32_t *const *indices,
                                 const double *count, const double *eff_len, const uint32_t *allowed,
                                 int device, uint32_t flags, gbrs_em_t **out);

/* Stored alignment values (an EMASE file saved with incidence_only = False, or a legacy COO file:
 * emase/Sparse3DMatrix.py:84-88, :93-99): values[h] double[nnz_h], aligned with indices[h] as given to
 * create.  EMfactory.prepare normalises them per read and that is all they are used for
 * (EMfactory.py:95-98; every E-step starts again from ones, Sparse3DMatrix.py:220-228), so this call
 * computes prepare()'s column sums from them once; prepare then uses those instead of 1/nnz_row.
 * Needs GBRS_EM_KEEP_CSC at create; call it once, before gbrs_em_prepare*. */
int gbrs_em_set_initial_values(gbrs_em_t *em, const double *const *values);

/* Replaces EMfactory.prepare (EMfactory.py:95-111): theta0 = sum_r count[r]/nnz_row / eff_len,
 * then the optional pseudocount rule (:105-111). */
int gbrs_em_prepare(gbrs_em_t *em, double pseudocount);

/* n_iters EM steps (EMfactory.update_allelic_expression, EMfactory.py:214-232) without looking
 * at the stopping rule; err_sum_out (nullable) receives the last step's total TPM change. */
int gbrs_em_step(gbrs_em_t *em, int n_iters, double *err_sum_out);

/* Replaces EMfactory.run (EMfactory.py:234-287).  model must be 4.  Stops when
 * err_sum <= 1e6*tol or after max_iters steps.  err_hist (nullable) receives up to
 * err_hist_cap per-iteration err_sum values (the numbers the reference prints); elapsed_s (nullable,
 * same capacity) the wall-clock seconds since the start of the run at which each iteration was
 * complete on the host's side - the "Time" column of the reference's progress table (:284-287).
 * The device loop hands control back once per batch of 8 iterations, so the iterations of a batch
 * share one time stamp. */
int gbrs_em_run(gbrs_em_t *em, int model, double tol, int max_iters,
                int *n_iters_out, double *err_hist, int err_hist_cap, double *elapsed_s);

/* theta (H x L) = EMfactory.allelic_expression; expected_counts (H x L) = probability.sum(READ)
 * of the last E-step (EMfactory.py:302), i.e. theta_before * A of the last iteration - on the single-GPU path formed
 * as theta * effective_length from the theta that iteration's M-step left (the same number up to two roundings), when
 * this call or gbrs_em_group_sums asks for it.  They stay those of the last iteration when gbrs_em_set_theta replaces
 * theta afterwards (the reports rescale theta to TPM first).  Either pointer may be NULL. */
int gbrs_em_get(gbrs_em_t *em, double *theta, double *expected_counts);

/* Overwrite theta (H x L), e.g. to resume from a checkpoint. */
int gbrs_em_set_theta(gbrs_em_t *em, const double *theta);

/* Gene-level sums: EMfactory.get_allelic_expression(at_group_level=True) (EMfactory.py:140-142)
 * and the grp_wise branch of report_read_counts (:305).  group_ptr int64[G+1], members int64[..]
 * locus ids.  which: 0 = theta, 1 = expected counts.  out is (H x G) row-major. */
int gbrs_em_group_sums(gbrs_em_t *em, int64_t num_groups, const int64_t *group_ptr,
                       const int64_t *members, int which, double *out);

/* Multi-GPU building blocks (rows sharded across ranks): one E-step over this
 * handle's rows accumulated into the handle's own partial buffer; the caller all-reduces that
 * buffer (RCCL) and then finishes the step.  partial_dev returns the DEVICE pointer of the
 * (L x H, locus-major) float64 partial-sum buffer and its element count. */
int gbrs_em_estep_partial(gbrs_em_t *em, void **partial_dev, uint64_t *n_elems);
/* M-step + stopping-rule bookkeeping on the all-reduced buffer.  With err_sum_out == NULL the error
 * pass (sum |dTPM|, iteration counter) is deferred into the next E-step launch; it is completed by
 * the next call that reports it (gbrs_em_finish_step with err_sum_out, gbrs_em_step, gbrs_em_sync). */
int gbrs_em_finish_step(gbrs_em_t *em, double *err_sum_out);
/* Same pair for prepare(): partial sums of count/nnz_row, then the division and pseudocount. */
int gbrs_em_prepare_partial(gbrs_em_t *em, void **partial_dev, uint64_t *n_elems);
int gbrs_em_finish_prepare(gbrs_em_t *em, double pseudocount);
/* The HIP stream (hipStream_t) the handle launches on, so the caller can order a collective. */
void *gbrs_em_stream(gbrs_em_t *em);
/* Launch on the caller's stream instead (e.g. torch's current stream, so that an RCCL all-reduce
 * issued through torch.distributed is ordered with the E-step without host synchronisation).
 * The caller keeps ownership of the stream. */
int gbrs_em_set_stream(gbrs_em_t *em, void *stream);
int gbrs_em_sync(gbrs_em_t *em);

/* Two handles that hold the two locus ranges of ONE sample (cut where no row straddles, GBRS_EM_SIDE_BY_SIDE; the two
 * engines of a rank in gbrs_amd/dist.py PipelinedShardedEM) share only the stopping rule of EMfactory.run
 * (emase/EMfactory.py:266-278): err_sum runs over the loci of both, scaled by the totals of both.  These calls evaluate
 * it on the device, with no host synchronisation in the loop, so that the pair stops at the reference's iteration:
 *   pair_begin   clears both handles' step scalars and the pair's counters (max_iters sizes the error history)
 *   pair_check   after gbrs_em_finish_step of BOTH handles for an iteration (a's first): enqueues the rule on b's
 *                stream behind a's M-step; when err_sum <= 1e6 * tol it raises both handles' stop flags - every later
 *                kernel of either is a no-op, theta stays the stopping iteration's; a's next M-step waits for it
 *   pair_status  synchronises and returns the iterations applied, the stop flag and the err_sum sequence */
int gbrs_em_pair_begin(gbrs_em_t *a, gbrs_em_t *b, int max_iters);
int gbrs_em_pair_check(gbrs_em_t *a, gbrs_em_t *b, double tol);
int gbrs_em_pair_status(gbrs_em_t *a, gbrs_em_t *b, int *iters_done, int *stopped, double *err_hist, int err_hist_cap);

typedef struct gbrs_em_info {
    uint64_t num_rows;          /* R as given                                               */
    uint64_t num_entries;       /* N = sum_h nnz_h                                          */
    uint64_t num_device_rows;   /* rows in the device layout (== R unless merged)           */
    uint64_t num_device_words;  /* 32-bit (row, locus) words streamed per E-step            */
    uint64_t bytes_per_iter;    /* bytes the E+M step kernels move per iteration (layout)   */
    uint64_t algorithmic_bytes; /* 4N + 4(R+1) [+8R] + 8HL*2 [+8HL]             */
    double   last_estep_ms;     /* HIP-event time of the E-step launch (mean of the timed    */
    double   last_step_ms;      /* steps: every 8th of a gbrs_em_step call) / of a full step */
    uint32_t num_loci, num_haps;
    uint32_t layout;            /* 0 = csc-direct, 1 = packed row tiles                     */
    uint32_t reserved;
    uint64_t num_tiles;         /* layout 1: workgroup tiles                                */
    uint64_t num_slots;         /* layout 1: (tile, locus) partial-sum slots                */
    uint64_t num_long_rows;     /* layout 1: rows handled by the long-row kernel            */
    uint64_t num_heavy_loci;    /* layout 1: loci with more than 16 slots (one wavefront each in the
                                   gather: emase/AlignmentPropertyMatrix.py:288-298 column sums)   */
    uint64_t num_light_loci;    /* layout 1: loci with 2..16 slots (summed in place)        */
    uint64_t estep_bytes;       /* bytes the E-step kernel itself moves per launch: word stream, tile
                                   headers, dictionary, theta gather, slot stores [, row weights]  */
    uint64_t retained_build_bytes; /* device bytes of build temporaries the handle still holds
                                   (GBRS_EM_ONE_SHOT; 0 otherwise)                                */
    uint64_t num_locus_sets;       /* distinct locus sets of the tile layout (GBRS_EM_NO_LOCUS_SETS: 0) */
} gbrs_em_info_t;
int gbrs_em_info(gbrs_em_t *em, gbrs_em_info_t *info);

/* `--report-alignment-counts` (emase/AlignmentPropertyMatrix.py:389-459), stand-alone (the
 * reference reloads the alignment file for it, gbrs/emase_utils.py:318-331).  Inputs as for
 * gbrs_em_create.  locus_group (nullable) int32[L] maps every locus to an output column, which
 * gives the gene-level report after _bundle_inline(reset=True) (:155-188); -1 = the locus is in no
 * group: its entries drop out, as they do from the product with grp_conv_mat.  num_out_loci = G
 * then, ignored (= L) otherwise.  Outputs, any nullable: aln_counts and allele_unique (H x Lout)
 * row-major, locus_unique (Lout).  Sums of EC counts in float64: exact for integer counts. */
int gbrs_alignment_counts(uint64_t num_rows, uint32_t num_loci, uint32_t num_haps,
                          const uint32_t *const *indptr, const uint32_t *const *indices,
                          const double *count, const int32_t *locus_group, uint32_t num_out_loci,
                          int device, double *aln_counts, double *allele_unique, double *locus_unique);

/* The same in three calls, for callers that want both levels (the `quantify -a` command writes the isoform-level and
 * the gene-level report, AlignmentPropertyMatrix.py:442-459 twice): create uploads the alignments once, every get
 * computes one set of counts (locus_group / num_out_loci as above) and re-uses the device workspace of the last. */
typedef struct gbrs_counts gbrs_counts_t;
int gbrs_counts_create(uint64_t num_rows, uint32_t num_loci, uint32_t num_haps,
                       const uint32_t *const *indptr, const uint32_t *const *indices,
                       const double *count, int device, gbrs_counts_t **out);
int gbrs_counts_get(gbrs_counts_t *c, const int32_t *locus_group, uint32_t num_out_loci,
                    double *aln_counts, double *allele_unique, double *locus_unique);
int gbrs_counts_destroy(gbrs_counts_t *c);

int gbrs_em_destroy(gbrs_em_t *em);

/* `gbrs compress` numeric body (gbrs/emase_utils.py:60-103): rows with identical alignment patterns
 * collapse into equivalence classes (ECs) whose count is the sum of the member counts; classes are
 * numbered in order of first appearance (the reference's dict insertion order, :77/:95); rows
 * without any alignment form one class with the empty key, as they do there.  Inputs as for
 * gbrs_em_create (several files = their rows concatenated by the caller).  create returns the
 * number of classes and the entries per haplotype so that the caller can size the outputs of get:
 * indptr_out[h] uint32[L+1], indices_out[h] uint32[nnz_per_hap[h]] (class ids ascending inside a
 * column), count_out double[num_ecs]. */
typedef struct gbrs_compress gbrs_compress_t;
int gbrs_compress_create(uint64_t num_rows, uint32_t num_loci, uint32_t num_haps,
                         const uint32_t *const *indptr, const uint32_t *const *indices,
                         const double *count, int device, gbrs_compress_t **out,
                         uint64_t *num_ecs, uint64_t *nnz_per_hap);
int gbrs_compress_get(gbrs_compress_t *c, uint32_t *const *indptr_out, uint32_t *const *indices_out,
                      double *count_out);
int gbrs_compress_destroy(gbrs_compress_t *c);

/* ------------------------------------------------------------------------------------------
 * HMM: per-chromosome forward-backward + Viterbi over the S = H(H+1)/2 diplotype states.
 * ---------------------------------------------------------------------------------------- */

typedef struct gbrs_hmm gbrs_hmm_t;

/*
 * Replaces the `tprob = np.load(tprob_file)` tables as consumed by gbrs_utils.py:500-599.
 *   n_genes[c]  genes on chromosome c (genome order)       n_trans[c]  len(tprob[c])
 *   tprob[c]    double[n_trans[c]][S][S], natural-log, T[i][to][from], C order
 * n_trans[c] may be n_genes[c]-1 (tables written by get_transition_prob, gbrs_utils.py:273-278)
 * or >= n_genes[c] (DO tables); both backtrace conventions of gbrs_utils.py:589-596 are kept.
 */
int gbrs_hmm_create(int num_haps, int n_chrom, const int32_t *n_genes, const int32_t *n_trans,
                    const double *const *tprob, int device, gbrs_hmm_t **out);

/* Emission model on the device: gbrs_utils.py:463-492 with get_genotype_probability (:80-98).
 *   expr[c]     double[n_samples][n_genes[c]][H]   gene-level TPM per haplotype
 *   avecs[c]    double[n_genes[c]][H][H]           alignment specificity (row i = haplotype i)
 *   has_avec[c] uint8[n_genes[c]]  0 -> naive avecs with sigma fixed 0.450 (:483-487)
 * avecs / has_avec are sample independent: they are copied to the device when given and stay resident
 * on the handle, so later calls (the next sample or batch of samples) may pass NULL for both and move
 * only the expression rows.
 */
int gbrs_hmm_set_expression(gbrs_hmm_t *hmm, int n_samples, const double *const *expr,
                            const double *const *avecs, const uint8_t *const *has_avec,
                            double expr_threshold, double sigma);

/* Alternative to set_expression: caller-computed log emissions eprob[c] double[n_samples][n_c][S]. */
int gbrs_hmm_set_eprob(gbrs_hmm_t *hmm, int n_samples, const double *const *eprob);

/* forward (:500-526) + Viterbi delta/backpointers (:567-579), backward + posterior (:530-560),
 * backtrace (:580-598) for every (sample, chromosome). */
int gbrs_hmm_run(gbrs_hmm_t *hmm);

/* Results of one (sample, chromosome).  Every pointer is nullable.
 *   gamma  double[S][n_c]  C order, as saved in genoprobs.npz (gbrs_utils.py:558-563)
 *   states int32[n'+1]     ordered Viterbi path as saved in genotypes.npz, n' = min(n_c, n_trans)
 *   calls  int32[n_c]      state index written to genotypes.tsv per gene, -1 = no entry
 *   alpha, beta, delta  double[S][n_c];  scaler double[n_c];  eprob double[n_c][S]
 * alpha, scaler and beta are the reference's log-domain intermediates; gbrs reconstruct saves none of
 * them, so gbrs_hmm_run leaves them out and the first get that asks for one makes them for the
 * whole last run (one extra device pass). */
int gbrs_hmm_get(gbrs_hmm_t *hmm, int sample, int chrom, double *gamma, int32_t *states,
                 int32_t *calls, double *alpha, double *beta, double *delta, double *scaler,
                 double *eprob);

typedef struct gbrs_hmm_info {
    uint64_t total_genes;        /* sum_c n_genes[c]                                        */
    uint64_t algorithmic_bytes;  /* per sample: sum_c n_c * (16 S^2 + 64 S)     */
    /* Device time of the phases of the last run.  For 8 founders (S = 36) the forward, backward and
     * Viterbi chains run concurrently on three streams, so forward (the longer of alpha and
     * delta + backpointers) and backward (sweep + outputs) overlap; last_run_ms is the whole of
     * gbrs_hmm_run on the device. */
    double   last_emission_ms, last_forward_ms, last_backward_ms, last_backtrace_ms;
    int32_t  num_states, n_samples;
    double   last_run_ms;
    /* Blocked scan (1-4 samples): the Viterbi values of the last run by rank convergence - blocks whose values were
     * matched to the block before them, the longest such fix-up in genes, and (sample, chromosome) pairs that were
     * recomputed by the sequential chain because a block did not converge.  0 on every other path. */
    int32_t  last_delta_blocks, last_delta_longest_fixup, last_delta_fallbacks, reserved0;
} gbrs_hmm_info_t;
int gbrs_hmm_info(gbrs_hmm_t *hmm, gbrs_hmm_info_t *info);

int gbrs_hmm_destroy(gbrs_hmm_t *hmm);

/* `gbrs interpolate` numeric body (gbrs/gbrs_utils.py:684-688): the rows of y (S x n_points, C
 * order) given at ascending positions x are linearly interpolated onto x_grid with scipy
 * interp1d(kind='linear')'s operation order; out is (S x n_grid).  The caller pads the gene
 * positions / posterior columns with the reference's two end points (:664-676, :684-685).  Grid
 * points outside [x[0], x[n_points-1]] fail with scipy's ValueError text. */
int gbrs_interpolate(int num_states, int n_points, const double *x, const double *y,
                     int n_grid, const double *x_grid, double *out, int device);

/* `gbrs export` numeric body (gbrs_utils.py:888-927): n_rows x S diplotype probabilities times the
 * (S x H) matrix 0.5 * (founder count in diplotype) -> n_rows x H founder dosages. */
int gbrs_genoprob_dosage(int num_haps, int64_t n_rows, const double *gprob, double *out, int device);

/* ------------------------------------------------------------------------------------------
 * Report text (host side, no device work): the `locus <haplotypes> total [notes]` tables of
 * EMfactory.report_read_counts / report_depths (emase/EMfactory.py:289-380).
 * ---------------------------------------------------------------------------------------- */

/* One double in the form str(numpy.float64) / repr(float) give it (shortest round-trip digits, fixed
 * notation for 1e-4 <= |x| < 1e16, otherwise d.ddde+XX).  out needs 32 bytes; returns the length. */
int gbrs_format_double(double v, char *out32);

/* Writes `header_line` and then, for k = 0 .. n_rows-1 and r = order ? order[k] : k, the line
 *     name r  TAB  value(r, 0) TAB ... TAB value(r, n_cols-1)  TAB  totals[r]  [TAB note r]  LF
 * with value(r, c) = values[r * row_stride + c * col_stride] (strides in elements, so both the
 * (H x L) row-major matrix of gbrs_em_get and its transpose are taken as they lie in memory);
 * totals[r] is the caller's column sum (kept outside so that its summation order stays the
 * caller's); names / notes are byte blobs addressed by n_rows + 1 offsets, name r =
 * names[name_off[r] .. name_off[r+1]); notes / note_off may both be NULL. */
int gbrs_write_locus_table(const char *path, const char *header_line, const double *values, int64_t n_rows,
                           int32_t n_cols, int64_t row_stride, int64_t col_stride, const double *totals,
                           const char *names, const int64_t *name_off, const char *notes,
                           const int64_t *note_off, const int64_t *order);

/* HDF5 chunk decoding for the EMASE reader (emase/Sparse3DMatrix.py:80-92 reads the h<k>/indices arrays through
 * PyTables one array at a time): the chunks of a 1-D dataset, located by the caller with
 * H5Dget_chunk_info (file address, stored size, first element, filter mask), are read with pread and
 * inflated (+ un-shuffled) on `threads` threads (0 = all cores) straight into `out`.  shuffle_pos /
 * deflate_pos are the positions of those filters in the dataset's pipeline, -1 when absent. */
int gbrs_decode_chunks(const char *path, int64_t n_chunks, const uint64_t *file_addr, const uint64_t *stored_bytes,
                       const uint64_t *elem_start, const uint32_t *filter_mask, uint64_t chunk_elems,
                       uint32_t elem_size, uint64_t n_elems, int32_t shuffle_pos, int32_t deflate_pos, void *out,
                       int32_t threads);
/* 1 = libdeflate, 2 = zlib, 0 = neither could be loaded at run time. */
int gbrs_inflate_backend(void);

/* The length table of EMfactory.prepare (emase/EMfactory.py:60-94) parsed natively: text is the whole
 * file (`<locus>_<haplotype> TAB <length>` lines, plain `<locus>` keys when n_haps == 1), names / haps
 * are byte blobs with n + 1 offsets, eff_out is (n_haps x n_loci) row-major and receives
 * max(length - read_length + 1, 1) for the listed pairs (other elements untouched).  Returns 0 when
 * every line was plain, 1 when some line needs the caller's own permissive line-by-line parsing and
 * error reporting (unknown name, second underscore, a number in a form from_chars does not take). */
int gbrs_parse_length_table(const char *text, int64_t text_len, const char *names, const int64_t *name_off,
                            int64_t n_loci, const char *haps, const int64_t *hap_off, int32_t n_haps,
                            double read_length, double *eff_out);

/* The genotype call table of `gbrs quantify -G` (gbrs/emase_utils.py:262-268) parsed natively: text is the whole file
 * (`#` lines that open it are skipped, then `<gene> TAB <call>[ TAB ...]` lines, every character of a call a
 * haplotype name); gene_names / haps are byte blobs with n + 1 offsets.  Per gene g (caller zeroes / presets the
 * arrays): gene_bits[g] |= the haplotype bits of each of its lines (the reference's mask accumulates over lines),
 * gene_call[g * call_width ...] = the call of its last line, zero padded, gene_last_line[g] = that line's index
 * (what the notes keep; -1 preset = no call).  n_lines receives the number of data lines.  Returns 0 when every
 * line was plain, 1 when some line needs the caller's own permissive line-by-line parsing and error reporting
 * (unknown gene or haplotype letter, a line without a second field, bytes outside printable ASCII, a call longer
 * than call_width). */
int gbrs_parse_genotype_table(const char *text, int64_t text_len, const char *gene_names, const int64_t *gene_off,
                              int64_t n_genes, const char *haps, const int64_t *hap_off, int32_t n_haps,
                              uint32_t *gene_bits, char *gene_call, int32_t call_width, int32_t *gene_last_line,
                              int64_t *n_lines);

/* The numbers of a `label TAB v1 TAB ... TAB vn` table (the genes.tpm report `gbrs reconstruct` reads,
 * gbrs/gbrs_utils.py:450-459): text holds exactly n_rows such lines (header removed), out receives n_rows x n_cols
 * doubles.  Returns 0, or 1 when a line is not of that plain form (the caller then parses it its own way). */
int gbrs_parse_number_table(const char *text, int64_t text_len, int64_t n_rows, int32_t n_cols, double *out);

/* `.npz` inputs of `gbrs reconstruct` (gbrs/gbrs_utils.py:420-441 opens avecs.npz with numpy.load and reads one
 * member per gene, :490 - zipfile re-parses the member's header and CRC-checks it on every access).
 * gbrs_zip_directory reads the central directory of a zip image (buf/len = the mapped file) once: member k's
 * compression method (0 stored, 8 deflate), compressed and plain sizes and local-header offset, and the member
 * names as one blob with an LF after each name.  Arrays hold `cap` members and `names_cap` bytes; the member count
 * and the blob size come back in n_members / names_len, so a first call with cap = 0 sizes the second.
 * GBRS_ERR_UNSUPPORTED: multi-disk archive (the caller falls back to zipfile). */
int gbrs_zip_directory(const uint8_t *buf, uint64_t len, uint64_t cap, uint16_t *method, uint64_t *csize,
                       uint64_t *usize, uint64_t *header_off, uint32_t *crc32 /* nullable: the members' CRC-32 */,
                       char *names, uint64_t names_cap, uint64_t *n_members, uint64_t *names_len);
/* n members' plain contents (their .npy images), member k into out[k] (usize[k] bytes, caller allocated), copied or
 * inflated on `threads` threads (0 = all cores), largest member first.  crc32 (nullable): the members' CRC-32 from the
 * central directory, checked on the thread that produced the bytes - numpy.load / zipfile check every member and raise
 * BadZipFile (the reference inherits that at gbrs_utils.py:420-441); a mismatch returns GBRS_ERR_INVALID. */
int gbrs_zip_read_members(const uint8_t *buf, uint64_t len, int64_t n, const uint64_t *header_off, const uint16_t *method,
                          const uint64_t *csize, const uint64_t *usize, const uint32_t *crc32, uint8_t *const *out,
                          int32_t threads);
/* n equally shaped .npy members (the per-gene 8 x 8 blocks) -> out[k * item_bytes ...], on `threads` threads
 * (0 = all cores): a member whose .npy image is exactly npy_header followed by item_bytes of data is copied
 * (stored) or inflated (raw deflate) into place; any other member - and, with crc32 given, one that fails its CRC-32 -
 * gets needs_fallback[k] = 1 and is left to the caller. */
int gbrs_npz_stack(const uint8_t *buf, uint64_t len, int64_t n, const uint64_t *header_off, const uint16_t *method,
                   const uint64_t *csize, const uint64_t *usize, const uint32_t *crc32, const uint8_t *npy_header,
                   uint64_t npy_header_len, uint64_t item_bytes, uint8_t *out, uint8_t *needs_fallback, int32_t threads);

#ifdef __cplusplus
}
#endif
#endif /* GBRS_HIP_H */
