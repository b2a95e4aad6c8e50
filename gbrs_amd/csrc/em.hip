// EMASE Model-4 EM on MI355X (gfx950).  See include/gbrs_hip.h for the boundary and DESIGN.md
// for the data layout and kernel inventory.
//
// One EM step of the reference (emase/EMfactory.py:214-232 with the Model-4 branch :204-208)
// never needs the per-entry posterior: with
//       den[r]  = sum_{(h,l) in row r} theta[h,l]                         (normalize_reads READ)
//       A[h,l]  = sum_{r in column (h,l)} count[r] / den[r]               (sum READ)
// the update is theta'[h,l] = theta[h,l] * A[h,l] / eff_len[h,l], and the expected read counts
// of that E-step are theta[h,l] * A[h,l].  Everything below computes den and A.
#include "common.h"
#include "em_layout.h"

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdlib>

namespace gbrs {

// ------------------------------------------------------------------------------------------
// Small per-iteration kernels (M-step, convergence), shared by every E-step layout.
// theta, acc, counts, eff_len are locus-major on the device: index l*H + h, so one locus is one
// 64-byte line at H = 8.
// ------------------------------------------------------------------------------------------

struct EmScalars {          // lives in device memory, one per handle
    double s_prev, s_new;   // sum of theta over everything before / after the step
    double err_sum;         // total TPM change of the last step
    double pc_before, pc_after;
    int stop;               // set once err_sum <= target (EMfactory.py:267)
    int iters_done;         // EM steps applied to theta
    int float_error;        // a referenced row had den == 0 (FloatingPointError in the reference)
    int ticket;             // arrival counter of err_finish_kernel (reset by its last block)
};

constexpr int RED_BLOCKS = 256;   // partial slots of the two-level deterministic reductions
constexpr int RED_THREADS = 256;

// theta' = theta * A / len, counts = theta * A, per-locus totals before and after, block partials.
// mode 0: EM step.  mode 1: prepare (theta treated as 1 everywhere).
template <int MODE>
__global__ void __launch_bounds__(RED_THREADS)
mstep_kernel(uint32_t L, uint32_t H, double *__restrict__ theta, const double *__restrict__ acc,
             const double *__restrict__ eff_len, double *__restrict__ counts,
             double *__restrict__ tot_prev, double *__restrict__ tot_new,
             double *__restrict__ msums, uint32_t cap, const EmScalars *__restrict__ sc) {
    __shared__ double lds[16];
    if (MODE == 0 && sc->stop) return;
    double p_prev = 0.0, p_new = 0.0;
    for (uint32_t l = blockIdx.x * blockDim.x + threadIdx.x; l < L; l += gridDim.x * blockDim.x) {
        double tp = 0.0, tn = 0.0;
        const size_t base = (size_t)l * H;
        for (uint32_t h = 0; h < H; ++h) {
            const double t = MODE == 0 ? theta[base + h] : 1.0;
            const double c = t * acc[base + h];
            double tnew = c;
            if (eff_len) tnew = c / eff_len[base + h];
            counts[base + h] = c;
            theta[base + h] = tnew;
            tp += t;
            tn += tnew;
        }
        tot_prev[l] = tp;
        tot_new[l] = tn;
        p_prev += tp;
        p_new += tn;
    }
    double a = block_sum(p_prev, lds);
    double b = block_sum(p_new, lds);
    if (threadIdx.x == 0) {
        msums[blockIdx.x] = a;
        msums[cap + blockIdx.x] = b;
    }
}

__device__ __forceinline__ double reduce_partials(const double *__restrict__ p, int n, double *lds) {
    double v = 0.0;
    for (int i = threadIdx.x; i < n; i += blockDim.x) v += p[i];
    return block_sum(v, lds);
}

// Element-parallel M-step for H a power of two: one thread per (locus, haplotype), the H lanes of
// a locus are adjacent, so every access is a coalesced 8-byte stream; per-locus totals by shuffles.
// A comes from `acc` (tile epilogues for single-tile loci, gather_kernel for the rest; loci without
// any read stay 0).  Block sums of theta before/after go to RED_BLOCKS accumulator slots.
template <int MODE>
__global__ void __launch_bounds__(RED_THREADS)
mstep_elem_kernel(uint32_t L, uint32_t H, double *__restrict__ theta, const double *__restrict__ acc,
                  const double *__restrict__ acc_extra, const double *__restrict__ eff_len,
                  double *__restrict__ counts, double *__restrict__ tot_prev, double *__restrict__ tot_new,
                  double *__restrict__ msums, uint32_t cap, const EmScalars *__restrict__ sc) {
    __shared__ double lds[16];
    if (MODE == 0 && sc->stop) return;
    const uint64_t n = (uint64_t)L * H;
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    double t = 0.0, tn = 0.0;
    if (i < n) {
        const uint32_t l = (uint32_t)(i / H), h = (uint32_t)(i & (H - 1));
        double a = acc[i];
        if (acc_extra) a += acc_extra[i];
        t = MODE == 0 ? theta[i] : 1.0;
        const double c = t * a;
        tn = eff_len ? c / eff_len[i] : c;
        counts[i] = c;
        theta[i] = tn;
        double tp = t, tq = tn;
        for (uint32_t off = 1; off < H; off <<= 1) {
            tp += __shfl_xor(tp, off, WAVE);
            tq += __shfl_xor(tq, off, WAVE);
        }
        if (h == 0) {
            tot_prev[l] = tp;
            tot_new[l] = tq;
        }
    }
    double a = block_sum(t, lds);
    double b = block_sum(tn, lds);
    if (threadIdx.x == 0) {                 // one slot per workgroup: no atomics, and a fixed summation order
        msums[blockIdx.x] = a;
        msums[cap + blockIdx.x] = b;
    }
}

constexpr int ERR_BLOCKS = 64;
// err_sum = sum_l | tot_new[l]*1e6/S_new - tot_prev[l]*1e6/S_prev |  (EMfactory.py:268-278), then the
// bookkeeping of the step (err history, iteration counter, stop flag; :266-267).  Every block
// reduces the M-step's block partials to S_prev / S_new itself (fixed order), writes its partial
// of err_sum and takes a ticket; the block that draws the last ticket sums the partials in fixed
// order, so the result does not depend on the order in which blocks finish.
// Arguments of the error pass, as one value: gather_kernel carries them for the deferred form.
struct ErrArgs {
    uint32_t L;
    int nblocks;                 // workgroups of the M-step launch = block sums to add up
    uint32_t cap;                // offset of the second block-sum array / of the error partials (x2)
    const double *tot_prev, *tot_new;
    double *partials;
    EmScalars *sc;
    double target_err;
    double *err_hist;
    int err_hist_cap;
    uint32_t n_err_blocks;       // workgroups that run the pass (0: none)
};

__device__ __forceinline__ void err_finish_body(const ErrArgs &ea, unsigned bid) {
    const uint32_t L = ea.L;
    const int nblocks = ea.nblocks;
    const double *__restrict__ tot_prev = ea.tot_prev, *__restrict__ tot_new = ea.tot_new;
    double *__restrict__ partials = ea.partials;
    EmScalars *__restrict__ sc = ea.sc;
    const double target_err = ea.target_err;
    double *__restrict__ err_hist = ea.err_hist;
    const int err_hist_cap = ea.err_hist_cap;
    const unsigned nb = ea.n_err_blocks;
    __shared__ double lds[16];
    __shared__ double s_sums[2];
    __shared__ int s_last;
    // (one read of the flag per workgroup: a partner handle's pair_err_kernel may raise it while this pass is running)
    if (threadIdx.x == 0) s_last = sc->stop;
    __syncthreads();
    if (s_last) return;
    __syncthreads();                                            // s_last is written again below
    double a = reduce_partials(partials, nblocks, lds);
    double b = reduce_partials(partials + ea.cap, nblocks, lds);
    if (threadIdx.x == 0) {
        s_sums[0] = a;
        s_sums[1] = b;
    }
    __syncthreads();
    const double cp = 1000000.0 / s_sums[0], cn = 1000000.0 / s_sums[1];
    double e = 0.0;
    for (uint32_t l = bid * blockDim.x + threadIdx.x; l < L; l += nb * blockDim.x)
        e += fabs(tot_new[l] * cn - tot_prev[l] * cp);
    e = block_sum(e, lds);
    if (threadIdx.x == 0) {
        partials[2 * (size_t)ea.cap + bid] = e;
        __threadfence();                                        // publish before the ticket
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // (the compiler may drop the fence's own wait)
        const int ticket = atomicAdd(&sc->ticket, 1);
        s_last = ticket == (int)nb - 1;
        if (s_last) __threadfence();                            // acquire the other blocks' partials
    }
    __syncthreads();
    if (!s_last) return;
    double tot = 0.0;
    for (int i = threadIdx.x; i < (int)nb; i += blockDim.x)
        tot += __hip_atomic_load(&partials[2 * (size_t)ea.cap + i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    tot = block_sum(tot, lds);
    if (threadIdx.x == 0) {
        sc->ticket = 0;
        sc->s_prev = s_sums[0];
        sc->s_new = s_sums[1];
        sc->err_sum = tot;
        const int it = sc->iters_done;
        if (err_hist && it < err_hist_cap) err_hist[it] = tot;
        sc->iters_done = it + 1;
        if (!(tot > target_err)) sc->stop = 1;
        if (!(s_sums[0] > 0.0) || !(s_sums[1] > 0.0) || tot != tot) sc->float_error = 1;
    }
}

__global__ void __launch_bounds__(RED_THREADS)
err_finish_kernel(ErrArgs ea) {
    err_finish_body(ea, blockIdx.x);
}

// The stopping rule of EMfactory.run (EMfactory.py:266-278) over TWO handles that hold the two locus ranges of one
// sample (gbrs_amd/dist.py PipelinedShardedEM): err_sum = sum over the loci of both ranges of
// | tot_new * 1e6 / S_new - tot_prev * 1e6 / S_prev | with S the totals over both.  One workgroup; when the sum is at or
// under the target it raises both handles' stop flags, so that every later kernel of either handle is a no-op and theta
// stays the stopping iteration's.
struct PairSide {
    uint32_t L;
    int nblocks;
    uint32_t cap;
    const double *tot_prev, *tot_new, *msums;
    EmScalars *sc;
};
struct PairState { int iters, stop; double err; };
__global__ void __launch_bounds__(1024)
pair_err_kernel(PairSide a, PairSide b, double target_err, PairState *__restrict__ ps, double *__restrict__ hist,
                int hist_cap) {
    __shared__ double lds[16];
    __shared__ double s_sum[2];
    if (ps->stop) return;
    double sp = 0.0, sn = 0.0;
    for (int i = threadIdx.x; i < a.nblocks; i += blockDim.x) { sp += a.msums[i]; sn += a.msums[a.cap + i]; }
    for (int i = threadIdx.x; i < b.nblocks; i += blockDim.x) { sp += b.msums[i]; sn += b.msums[b.cap + i]; }
    sp = block_sum(sp, lds);
    sn = block_sum(sn, lds);
    if (threadIdx.x == 0) { s_sum[0] = sp; s_sum[1] = sn; }
    __syncthreads();
    const double cp = 1000000.0 / s_sum[0], cn = 1000000.0 / s_sum[1];
    double e = 0.0;
    for (uint32_t l = threadIdx.x; l < a.L; l += blockDim.x) e += fabs(a.tot_new[l] * cn - a.tot_prev[l] * cp);
    for (uint32_t l = threadIdx.x; l < b.L; l += blockDim.x) e += fabs(b.tot_new[l] * cn - b.tot_prev[l] * cp);
    e = block_sum(e, lds);
    if (threadIdx.x == 0) {
        const int it = ps->iters;
        if (hist && it < hist_cap) hist[it] = e;
        ps->iters = it + 1;
        ps->err = e;
        if (!(e > target_err)) {
            ps->stop = 1;
            a.sc->stop = 1;
            b.sc->stop = 1;
        }
        if (!(s_sum[0] > 0.0) || !(s_sum[1] > 0.0) || e != e) a.sc->float_error = 1;
    }
}

// pseudocount rule, EMfactory.py:105-111: every haplotype of a locus with any nonzero haplotype
// gets +pc, then the whole matrix is rescaled to its previous total.
__global__ void __launch_bounds__(RED_THREADS)
pseudo_add_kernel(uint32_t L, uint32_t H, double pc, double *__restrict__ theta,
                  double *__restrict__ partials) {
    __shared__ double lds[16];
    double before = 0.0, after = 0.0;
    for (uint32_t l = blockIdx.x * blockDim.x + threadIdx.x; l < L; l += gridDim.x * blockDim.x) {
        const size_t base = (size_t)l * H;
        bool any = false;
        double s = 0.0;
        for (uint32_t h = 0; h < H; ++h) {
            const double t = theta[base + h];
            any |= (t != 0.0);
            s += t;
        }
        before += s;
        if (any) {
            double s2 = 0.0;
            for (uint32_t h = 0; h < H; ++h) {
                const double t = theta[base + h] + pc;
                theta[base + h] = t;
                s2 += t;
            }
            after += s2;
        } else {
            after += s;
        }
    }
    double a = block_sum(before, lds);
    double b = block_sum(after, lds);
    if (threadIdx.x == 0) {
        partials[blockIdx.x] = a;
        partials[RED_BLOCKS + blockIdx.x] = b;
    }
}

__global__ void __launch_bounds__(RED_THREADS)
pseudo_scale_kernel(uint64_t n, int nblocks, double *__restrict__ theta,
                    const double *__restrict__ partials, EmScalars *__restrict__ sc) {
    __shared__ double lds[16];
    __shared__ double s_f;
    double a = reduce_partials(partials, nblocks, lds);
    double b = reduce_partials(partials + RED_BLOCKS, nblocks, lds);
    if (threadIdx.x == 0) {
        s_f = a / b;
        if (blockIdx.x == 0) {
            sc->pc_before = a;
            sc->pc_after = b;
        }
    }
    __syncthreads();
    const double f = s_f;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (uint64_t)gridDim.x * blockDim.x)
        theta[i] *= f;
}

// (H x L) row-major host order <-> (L x H) locus-major device order
__global__ void transpose_hl_to_lh(uint32_t L, uint32_t H, const double *__restrict__ src,
                                   double *__restrict__ dst) {
    const uint64_t n = (uint64_t)L * H;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t l = i / H, h = i % H;
        dst[i] = src[(uint64_t)h * L + l];
    }
}
__global__ void transpose_lh_to_hl(uint32_t L, uint32_t H, const double *__restrict__ src,
                                   double *__restrict__ dst) {
    const uint64_t n = (uint64_t)L * H;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t h = i / L, l = i % L;
        dst[i] = src[(uint64_t)l * H + h];
    }
}

// Gene-level segmented sum (EMfactory.py:140-142): out[h][g] = sum over member loci in ascending
// order -- the accumulation order of scipy's csc right-multiplication, so given the same theta
// the result is the reference's bit for bit.  One thread per (gene, haplotype).
__global__ void group_sum_kernel(uint32_t H, int64_t G, const int64_t *__restrict__ gptr,
                                 const int64_t *__restrict__ members,
                                 const double *__restrict__ src_lh, double *__restrict__ out_hg) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= G * (int64_t)H) return;
    const int64_t g = i / H;
    const uint32_t h = i % H;
    double s = 0.0;
    for (int64_t k = gptr[g]; k < gptr[g + 1]; ++k) s += src_lh[(size_t)members[k] * H + h];
    out_hg[(size_t)h * G + g] = s;
}

// ------------------------------------------------------------------------------------------
// E-step, layout 0 ("csc-direct"): works straight on the reference's CSC arrays.  Two passes
// with global float64 atomics; kept as the simple correct baseline and as the cross-check for
// the tiled layout.
// ------------------------------------------------------------------------------------------

template <bool ONES>
__global__ void __launch_bounds__(256)
csc_den_kernel(uint64_t n, uint32_t ncols, uint32_t L, uint32_t H,
               const uint64_t *__restrict__ col_ptr, const uint32_t *__restrict__ ent_row,
               const double *__restrict__ theta, double *__restrict__ den,
               const EmScalars *__restrict__ sc) {
    if (!ONES && sc->stop) return;
    const uint64_t k = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k - (threadIdx.x & 63) >= n) return;
    const bool live = k < n;
    const uint32_t c = entry_column(col_ptr, ncols, live ? k : n - 1, n);
    if (!live) return;
    const uint32_t h = c / L, l = c - h * L;
    const double t = ONES ? 1.0 : theta[(size_t)l * H + h];
    atomicAdd(&den[ent_row[k]], t);
}

template <bool ONES>
__global__ void __launch_bounds__(256)
csc_acc_kernel(uint64_t n, uint32_t ncols, uint32_t L, uint32_t H,
               const uint64_t *__restrict__ col_ptr, const uint32_t *__restrict__ ent_row,
               const double *__restrict__ count, const double *__restrict__ den,
               double *__restrict__ acc, EmScalars *__restrict__ sc) {
    if (!ONES && sc->stop) return;
    const uint64_t k = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k - (threadIdx.x & 63) >= n) return;
    const bool live = k < n;
    const uint32_t c = entry_column(col_ptr, ncols, live ? k : n - 1, n);
    double w = 0.0;
    if (live) {
        const uint32_t r = ent_row[k];
        const double d = den[r];
        const double cnt = count ? count[r] : 1.0;
        if (d > 0.0) w = cnt / d; else sc->float_error = 1;
    }
    const uint32_t c0 = __shfl(c, 0, WAVE);
    if (__all(c == c0)) {
        w = wave_sum(w);
        if ((threadIdx.x & 63) == 0) {
            const uint32_t h = c0 / L, l = c0 - h * L;
            atomicAdd(&acc[(size_t)l * H + h], w);
        }
    } else if (live) {
        const uint32_t h = c / L, l = c - h * L;
        atomicAdd(&acc[(size_t)l * H + h], w);
    }
}

// Stored alignment values (files saved with incidence_only = False, legacy COO files): EMfactory.prepare
// normalises them per read and sums them per column (EMfactory.py:95-98), once; every later E-step
// starts from ones again (Sparse3DMatrix.reset).  den[r] = sum of the row's values, then
// acc[l][h] += count[r] * value / den[r].
__global__ void __launch_bounds__(256)
csc_den_values_kernel(uint64_t n, const uint32_t *__restrict__ ent_row, const double *__restrict__ vals,
                      double *__restrict__ den) {
    const uint64_t k = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k < n) atomicAdd(&den[ent_row[k]], vals[k]);
}

__global__ void __launch_bounds__(256)
csc_acc_values_kernel(uint64_t n, uint32_t ncols, uint32_t L, uint32_t H, const uint64_t *__restrict__ col_ptr,
                      const uint32_t *__restrict__ ent_row, const double *__restrict__ vals,
                      const double *__restrict__ count, const double *__restrict__ den, double *__restrict__ acc,
                      int *__restrict__ bad) {
    const uint64_t k = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k - (threadIdx.x & 63) >= n) return;
    const bool live = k < n;
    const uint32_t c = entry_column(col_ptr, ncols, live ? k : n - 1, n);
    if (!live) return;
    const uint32_t r = ent_row[k];
    const double d = den[r];
    if (!(d > 0.0) || !(vals[k] >= 0.0)) { *bad = 1; return; }
    const uint32_t h = c / L, l = c - h * L;
    atomicAdd(&acc[(size_t)l * H + h], (count ? count[r] : 1.0) * vals[k] / d);
}

// `-G` haplotype mask (gbrs/emase_utils.py:240-273: multiply(gtmask, axis=2) + eliminate_zeros): the mask is
// per (haplotype, locus), i.e. it drops whole CSC columns, so the masked tensor is the kept columns moved up
// against each other.  dst_ptr are the column offsets after the mask (dropped columns have width 0), src_ptr the
// offsets in the arrays as uploaded; one thread per surviving entry.
template <typename T>
__global__ void __launch_bounds__(256)
compact_columns_kernel(uint64_t n, uint32_t ncols, const uint64_t *__restrict__ dst_ptr,
                       const uint64_t *__restrict__ src_ptr, const T *__restrict__ src, T *__restrict__ dst) {
    const uint64_t k = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k - (threadIdx.x & 63) >= n) return;
    const bool live = k < n;
    const uint32_t c = entry_column(dst_ptr, ncols, live ? k : n - 1, n);
    if (live) dst[k] = src[src_ptr[c] + (k - dst_ptr[c])];
}

// largest row id of the uploaded arrays (inputs that arrive as device pointers cannot be checked on
// the host, and an out-of-range id would make the scatter kernels fault)
__global__ void __launch_bounds__(256)
max_row_kernel(uint64_t n, const uint32_t *__restrict__ ent_row, unsigned int *__restrict__ out) {
    unsigned int m = 0;
    for (uint64_t k = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; k < n; k += (uint64_t)gridDim.x * blockDim.x)
        m = max(m, ent_row[k]);
    for (int off = 32; off > 0; off >>= 1) m = max(m, (unsigned int)__shfl_down((int)m, off, WAVE));
    if ((threadIdx.x & 63) == 0) atomicMax(out, m);
}

// `--report-alignment-counts` on the CSC arrays (AlignmentPropertyMatrix.py:389-459).
//   nnz_row[r]   number of stored entries of row r   (sum LOCUS then HAPLOTYPE)
//   nloc_row[r]  number of distinct loci of row r    (nnz of the HAPLOTYPE-summed matrix)
__global__ void __launch_bounds__(256)
csc_rowstat_kernel(uint64_t n, uint32_t ncols, uint32_t L, const uint64_t *__restrict__ col_ptr,
                   const uint32_t *__restrict__ ent_row, uint32_t *__restrict__ nnz_row) {
    const uint64_t k = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k < n) atomicAdd(&nnz_row[ent_row[k]], 1u);
}

#include "em_tiles.inc"

}  // namespace gbrs

using namespace gbrs;

// ------------------------------------------------------------------------------------------
// Handle
// ------------------------------------------------------------------------------------------

struct gbrs_em {
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = true;
    hipEvent_t ev0 = nullptr, ev1 = nullptr, ev2 = nullptr;
    uint64_t R = 0, N = 0;
    uint32_t L = 0, H = 0;
    bool has_count = false, has_len = false, prepared = false;
    uint32_t flags = 0;

    int layout = 0;               // 0 = csc-direct, 1 = packed row tiles
    uint32_t persist_groups = 0;  // > 0: the E-step of a step runs on this many persistent workgroups (em_tiles.inc)
    uint32_t view = 1;            // 2: the tile layout is built over half-loci (em_layout.h): tL() loci of tH() haplotypes
    uint32_t tH() const { return H / view; }
    uint32_t tL() const { return L * view; }
    bool acc_needs_extra = false; // last E-step left the long-row sums in tl.acc_extra for the M-step to add
    bool acc_external = false;    // the caller all-reduces acc (sharded): always materialise all of it
    TileLayout tl;

    // layout 0
    DevBuf<uint32_t> ent_row;
    DevBuf<uint64_t> col_ptr;     // H*L + 1
    DevBuf<uint64_t> col_ptr_src; // H*L + 1: column offsets of the caller's arrays when a haplotype mask dropped columns
    bool masked = false;          // (kept with GBRS_EM_KEEP_CSC, for gbrs_em_set_initial_values)
    DevBuf<double> den;           // R

    DevBuf<double> count, eff_len;                 // R ; L*H locus-major
    DevBuf<double> acc_init;                       // L*H: prepare()'s column sums when the file stores values
    bool has_init = false, keep_csc = false;
    bool stopped = false;          // the device's stop flag is set: every step kernel is a no-op until it is cleared
    DevBuf<double> theta, acc, counts;             // L*H locus-major
    bool counts_stale = false;                     // the fused M-step ran since `counts` was written (em_refresh_counts)
    DevBuf<double> tot_prev, tot_new;              // L
    DevBuf<double> partials;                       // 3 * RED_BLOCKS (pseudocount reductions)
    DevBuf<double> msums;                          // M-step block sums [2][msum_cap] + ERR_BLOCKS error partials
    uint32_t msum_cap = 0, msum_blocks = 0;        // capacity; workgroups of the last M-step launch
    DevBuf<double> scratch_hl;                     // L*H staging for host <-> device transposes
    DevBuf<double> err_hist;
    DevBuf<EmScalars> scalars;
    int err_hist_cap = 0;
    double last_estep_ms = 0.0, last_step_ms = 0.0;   // means over the steps of the last call
    bool time_steps = true;
    // The error pass of a step may be deferred into the gather launch of the next step (it runs there
    // on extra workgroups beside the gather's, one launch and ~9 us less per iteration); it is
    // flushed as its own launch before anything reads the scalars.
    bool err_pending = false;
    double err_pending_target = -1.0;
    hipEvent_t ev_after_estep = nullptr;      // timed step: recorded between the E-step kernels and the gather
    std::vector<hipEvent_t> ev_pool;                   // 3 events per timed step of gbrs_em_step

    // Pair mode (gbrs_em_pair_*): this handle and a partner hold the two locus ranges of one sample; the stopping rule is
    // evaluated over both on the device.  The first handle of the pair owns the state.
    struct PairScalars { int iters, stop; double err; };
    DevBuf<PairScalars> pair_sc;
    DevBuf<double> pair_hist;
    int pair_hist_cap = 0;
    hipEvent_t pair_ev_mstep = nullptr;    // recorded after every M-step of a handle in pair mode
    hipEvent_t pair_ev_err = nullptr;      // recorded after the pair's error pass (first handle)

    int red_blocks() const { return (int)std::min<uint64_t>(RED_BLOCKS, (L + RED_THREADS - 1) / RED_THREADS); }
};

// gbrs_counts_*: the alignments of `--report-alignment-counts` on the device, and the workspace of the last get
struct gbrs_counts {
    int device = 0;
    uint64_t R = 0, n = 0;
    uint32_t L = 0, H = 0;
    gbrs::DevBuf<uint32_t> ent_row;
    gbrs::DevBuf<uint64_t> col_ptr;
    gbrs::DevBuf<double> d_count;
    hipStream_t s = nullptr;
    gbrs::CountsWork *work = nullptr;
};

namespace {

int em_flush_err(gbrs_em *em);

int em_check_float(gbrs_em *em, EmScalars &host) {
    GBRS_TRY(em_flush_err(em));
    GBRS_HIP_CHECK(hipMemcpyAsync(&host, em->scalars.p, sizeof(EmScalars), hipMemcpyDeviceToHost, em->stream));
    GBRS_HIP_CHECK(hipStreamSynchronize(em->stream));
    em->stopped = host.stop != 0;
    static const bool no_check = std::getenv("GBRS_TUNING_NO_FLOAT_CHECK") != nullptr;   // ablation builds only
    if (host.float_error && !no_check)
        return fail(GBRS_ERR_FLOAT, "invalid value encountered in divide (a read's alignments all have zero abundance)");
    return GBRS_OK;
}

// E-step over the tiled layout: tiles -> partials, long rows -> acc_extra, gather -> acc.
ErrArgs em_err_args(gbrs_em *em, double target_err);
int em_flush_err(gbrs_em *em);

template <int HT, bool ONES>
int em_estep_tiles_h(gbrs_em *em) {
    const TileLayout &tl = em->tl;
    if (tl.n_tiles) {
        // a deferred error pass of the previous step rides on this launch as extra workgroups
        ErrArgs ea = em_err_args(em, em->err_pending_target);
        if (ONES || !em->err_pending) ea.n_err_blocks = 0;
        em->err_pending = false;
        const dim3 grid((unsigned)tl.n_tiles + ea.n_err_blocks), block(TILE_THREADS);
        const double *ww = tl.weighted ? tl.word_weight.p : (const double *)nullptr;
        const SetArgs sets{em->tL(), tl.n_sets ? tl.set_ptr.p : nullptr, tl.set_members.p, tl.dest_list.p, tl.dict_b.p, tl.dest_b.p};
#define GBRS_LAUNCH_TILES(W, D)                                                                                        \
        hipLaunchKernelGGL((tile_estep_kernel<HT, W, ONES, D>), grid, block, 0, em->stream, em->tH(), tl.tiles.p, tl.words.p, \
                           tl.dict.p, ww, em->theta.p, tl.partials.p, tl.slot_dest.p, em->acc.p, em->scalars.p,       \
                           (uint32_t)tl.n_tiles, ea, sets)
        if (!ONES && !tl.deterministic && em->persist_groups > 0 && HT > 0 && HT <= 8) {
            // persistent workgroups (em_tiles.inc): as many as the chip holds at once, each walking several tiles with the
            // next tile's header, dictionary and theta fetched under the current tile's batch loop
            const uint32_t G = std::min<uint32_t>(em->persist_groups, (uint32_t)tl.n_tiles);
            const dim3 pgrid(G + ea.n_err_blocks);
#define GBRS_LAUNCH_PERSISTENT(HH, W, OW)                                                                              \
            hipLaunchKernelGGL((tile_estep_persistent_kernel<HH, W, OW>), pgrid, block, 0, em->stream, em->tH(), tl.tiles.p,  \
                               tl.words.p, tl.dict.p, ww, em->theta.p, tl.partials.p, tl.slot_dest.p,                  \
                               (int64_t)(em->acc.p - tl.partials.p), em->scalars.p, (uint32_t)tl.n_tiles, G, ea, sets)
            if (HT == 8 && !tl.weighted && tl.all_one_word) GBRS_LAUNCH_PERSISTENT(HT == 8 ? 8 : 1, false, HT == 8);
            else if (tl.weighted) GBRS_LAUNCH_PERSISTENT(HT, true, false);
            else GBRS_LAUNCH_PERSISTENT(HT, false, false);
#undef GBRS_LAUNCH_PERSISTENT
        } else if (tl.deterministic) {           // fixed-order sums instead of LDS float atomics (GBRS_EM_DETERMINISTIC)
            if (tl.weighted) GBRS_LAUNCH_TILES(true, true); else GBRS_LAUNCH_TILES(false, true);
#if !defined(GBRS_NO_ONEWORD)
        } else if (HT == 8 && !tl.weighted && tl.all_one_word) {
            // no row of the layout has more than one word (single-locus reads, or locus sets): no row sums at all
            hipLaunchKernelGGL((tile_estep_kernel<HT == 8 ? 8 : 1, false, ONES, false, HT == 8>), grid, block, 0, em->stream, em->tH(),
                               tl.tiles.p, tl.words.p, tl.dict.p, ww, em->theta.p, tl.partials.p, tl.slot_dest.p, em->acc.p,
                               em->scalars.p, (uint32_t)tl.n_tiles, ea, sets);
#endif
        } else {
            if (tl.weighted) GBRS_LAUNCH_TILES(true, false); else GBRS_LAUNCH_TILES(false, false);
        }
#undef GBRS_LAUNCH_TILES
    }
    return GBRS_OK;
}
template <bool ONES>
int em_estep_tiles(gbrs_em *em, bool materialize, bool skip_gather = false) {
    if (ONES || !em->tl.n_tiles) GBRS_TRY(em_flush_err(em));   // prepare / no tile launch to ride on

    TileLayout &tl = em->tl;
    switch (em->tH()) {
        case 1: GBRS_TRY((em_estep_tiles_h<1, ONES>(em))); break;
        case 2: GBRS_TRY((em_estep_tiles_h<2, ONES>(em))); break;
        case 4: GBRS_TRY((em_estep_tiles_h<4, ONES>(em))); break;
        case 8: GBRS_TRY((em_estep_tiles_h<8, ONES>(em))); break;
        case 16: GBRS_TRY((em_estep_tiles_h<16, ONES>(em))); break;
        default: GBRS_TRY((em_estep_tiles_h<0, ONES>(em))); break;
    }
    if (tl.n_long) {
        GBRS_HIP_CHECK(hipMemsetAsync(tl.acc_extra.p, 0, tl.acc_extra.bytes(), em->stream));
        if (tl.deterministic)
            hipLaunchKernelGGL(long_rows_estep_serial_kernel<ONES>, dim3(1), dim3(64), 0, em->stream, tl.n_long, em->tH(),
                               tl.long_ptr.p, tl.long_loc.p, tl.long_mask.p, tl.long_weight.p, em->theta.p,
                               tl.acc_extra.p, em->scalars.p);
        else
            hipLaunchKernelGGL(long_rows_estep_kernel<ONES>, dim3((unsigned)((tl.n_long + 3) / 4)), dim3(256), 0, em->stream,
                               tl.n_long, em->tH(), tl.long_ptr.p, tl.long_loc.p, tl.long_mask.p, tl.long_weight.p,
                               em->theta.p, tl.acc_extra.p, em->scalars.p);
    }
    if (em->ev_after_estep) {
        GBRS_HIP_CHECK(hipEventRecord(em->ev_after_estep, em->stream));
        em->ev_after_estep = nullptr;
    }
    // (the gather walks the layout's loci: half-loci of tH() haplotypes under the half-locus view - the same elements of acc)
    const uint32_t gH = em->tH(), gL = em->tL();
    uint32_t HP = 1;
    while (HP < gH) HP <<= 1;
    const bool pow2 = (gH & (gH - 1)) == 0;
    const bool all = materialize || em->acc_external || !pow2 || em->view > 1;     // write every element of acc
    const uint64_t elems = all ? (uint64_t)gL * gH : (uint64_t)tl.n_light * gH;
    const unsigned light = (unsigned)((elems + 255) / 256);
    const unsigned heavy = (unsigned)((tl.n_heavy + 3) / 4);
    em->acc_needs_extra = !all && tl.n_long > 0;
    if (skip_gather) return GBRS_OK;           // the fused gather + M-step kernel follows (em_one_step)
    if (light + heavy > 0)
        hipLaunchKernelGGL(gather_kernel, dim3(light + heavy), dim3(256), 0, em->stream, gL, gH,
                           HP, light, heavy, (uint32_t)tl.n_heavy, all ? 1 : 0, (uint32_t)tl.n_light, tl.light_loci.p,
                           tl.slot_ptr.p, tl.heavy_loci.p, tl.locus_class.p, tl.partials.p,
                           (tl.n_long && all) ? tl.acc_extra.p : (const double *)nullptr, em->acc.p, em->scalars.p,
                           ONES ? 0 : 1);
    GBRS_HIP_CHECK(hipGetLastError());
    return GBRS_OK;
}

// E-step: fills em->acc with A (sum of count/den per column).  ONES: theta treated as 1.
template <bool ONES>
int em_estep(gbrs_em *em, bool materialize = false) {
    if (em->layout == 1) return em_estep_tiles<ONES>(em, materialize);
    em->acc_needs_extra = false;
    const uint64_t n = em->N;
    const uint32_t ncols = em->H * em->L;
    GBRS_HIP_CHECK(hipMemsetAsync(em->acc.p, 0, em->acc.bytes(), em->stream));
    if (n == 0) return GBRS_OK;
    GBRS_HIP_CHECK(hipMemsetAsync(em->den.p, 0, em->den.bytes(), em->stream));
    const unsigned grid = (unsigned)((n + 255) / 256);
    hipLaunchKernelGGL(csc_den_kernel<ONES>, dim3(grid), dim3(256), 0, em->stream, n, ncols, em->L,
                       em->H, em->col_ptr.p, em->ent_row.p, em->theta.p, em->den.p, em->scalars.p);
    hipLaunchKernelGGL(csc_acc_kernel<ONES>, dim3(grid), dim3(256), 0, em->stream, n, ncols, em->L,
                       em->H, em->col_ptr.p, em->ent_row.p, em->has_count ? em->count.p : nullptr,
                       em->den.p, em->acc.p, em->scalars.p);
    GBRS_HIP_CHECK(hipGetLastError());
    return GBRS_OK;
}

// Gather + M-step in one launch (tile layout, H a power of two, single GPU): an element of a locus
// with at most one slot takes A straight from `acc` (written by the tile epilogue), a locus with a
// few slots sums them in place, and the loci with many slots get one workgroup each (the leading
// workgroups), which reduces the slots in fixed order and applies the M-step to that locus itself.
// No intermediate A vector is written for the gathered loci and there is one launch less per step.
//
// Round 3: the launch was ~13 us of a ~115 us iteration and none of it was bandwidth - it is the number of
// DEPENDENT global loads on its longest path (a round trip is ~0.7 us): stop flag -> locus list -> slot_ptr ->
// four slot rows at a time -> ... .  Now every path is two round trips: (1) the stop flag, the (locus, first row,
// end row) record of a heavy / light locus, or class, theta, A and length of a plain element, all in flight
// together; (2) all the slot rows of the locus at once (a light locus has at most HEAVY_SLOTS of them; a heavy
// workgroup takes 16 rows per thread and round).  The stop flag only guards the stores.
#ifndef GBRS_MSTEP_EPT
#define GBRS_MSTEP_EPT 4
#endif
constexpr int MSTEP_EPT = GBRS_MSTEP_EPT;       // elements per thread of the elementwise workgroups (measured 1, 2, 4, 8)
constexpr int HEAVY_ROWS = 16;     // slot rows a thread of a heavy workgroup has in flight
__global__ void __launch_bounds__(RED_THREADS)
mstep_gather_kernel(uint32_t L, uint32_t H, uint32_t HP, uint32_t heavy_blocks, uint32_t light_blocks, uint32_t n_light,
                    const uint32_t *__restrict__ heavy_range, const uint32_t *__restrict__ light_range,
                    const uint8_t *__restrict__ locus_class,
                    const double *__restrict__ slot_sums, const double *__restrict__ acc,
                    const double *__restrict__ acc_extra, double *__restrict__ theta,
                    const double *__restrict__ eff_len, double *__restrict__ counts, double *__restrict__ tot_prev,
                    double *__restrict__ tot_new, double *__restrict__ msums, uint32_t cap,
                    const EmScalars *__restrict__ sc) {
    __shared__ double lds[16];
    __shared__ double heavy_part[RED_THREADS / 64][16];
    const int stop = sc->stop;          // scalar load, in flight beside the vector loads below; looked at before the stores
    double t = 0.0, tn = 0.0;
    if (blockIdx.x < heavy_blocks) {
        // one workgroup per many-slot locus: the largest loci of a deep sample have hundreds of slots whose rows
        // were written by tiles all over the chip a moment ago
        const uint32_t l = heavy_range[3 * blockIdx.x], k0 = heavy_range[3 * blockIdx.x + 1], k1 = heavy_range[3 * blockIdx.x + 2];
        const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
        const uint32_t h = threadIdx.x & (HP - 1), sub = threadIdx.x / HP, nsub = blockDim.x / HP;
        const bool live = h < H;
        const size_t il = (size_t)l * H + (live ? h : 0);
        const double t_old = theta[il], ax = acc_extra ? acc_extra[il] : 0.0, ln = eff_len ? eff_len[il] : 1.0;
        double a = 0.0;
        for (uint32_t k = k0 + sub; k < k1; k += HEAVY_ROWS * nsub) {          // fixed order: rounds, then rows of a round
            double r[HEAVY_ROWS];
#pragma unroll
            for (int m = 0; m < HEAVY_ROWS; ++m) {
                const uint32_t km = k + (uint32_t)m * nsub;
                r[m] = (live && km < k1) ? slot_sums[(size_t)km * H + h] : 0.0;
            }
#pragma unroll
            for (int w = 1; w < HEAVY_ROWS; w <<= 1)
#pragma unroll
                for (int m = 0; m + w < HEAVY_ROWS; m += 2 * w) r[m] += r[m + w];
            a += r[0];
        }
        for (uint32_t off = HP; off < 64; off <<= 1) a += __shfl_xor(a, off, WAVE);
        if (lane < (int)HP) heavy_part[wv][lane] = a;        // the wavefronts' partial sums, added in a fixed order
        __syncthreads();
        if (wv == 0) {
            a = 0.0;
            if (lane < (int)HP)
                for (uint32_t w2 = 0; w2 < blockDim.x / 64; ++w2) a += heavy_part[w2][lane];
            const bool mine = lane < (int)HP && live;
            if (mine) {
                t = t_old;
                const double c = t * (a + ax);
                tn = eff_len ? c / ln : c;
                if (!stop) theta[il] = tn;             // (the expected counts are theta' * len: made when asked for)
            }
            double tp = t, tq = tn;                       // lanes 0 .. H-1 hold the locus, the others 0
            for (uint32_t off = 1; off < HP; off <<= 1) {
                tp += __shfl_xor(tp, off, WAVE);
                tq += __shfl_xor(tq, off, WAVE);
            }
            if (lane == 0 && !stop) {
                tot_prev[l] = tp;
                tot_new[l] = tq;
            }
        }
    } else if (blockIdx.x < heavy_blocks + light_blocks) {
        // one thread per (light locus, haplotype): the locus's 2..HEAVY_SLOTS slot rows all at once, added in slot order
        const uint64_t i = (uint64_t)(blockIdx.x - heavy_blocks) * blockDim.x + threadIdx.x;
        const bool live = i < (uint64_t)n_light * H;
        const uint32_t li = live ? (uint32_t)(i / H) : 0, h = (uint32_t)(i & (H - 1));
        const uint32_t l = light_range[3 * li], k0 = light_range[3 * li + 1], k1 = light_range[3 * li + 2];
        const size_t il = (size_t)l * H + h;
        const double t_old = theta[il], ax = acc_extra ? acc_extra[il] : 0.0, ln = eff_len ? eff_len[il] : 1.0;
        double r[HEAVY_SLOTS];
#pragma unroll
        for (int m = 0; m < HEAVY_SLOTS; ++m) r[m] = k0 + m < k1 ? slot_sums[(size_t)(k0 + m) * H + h] : 0.0;
        double a = ax;
#pragma unroll
        for (int m = 0; m < HEAVY_SLOTS; ++m) a += r[m];
        if (live) {
            t = t_old;
            const double c = t * a;
            tn = eff_len ? c / ln : c;
            if (!stop) theta[il] = tn;
        }
        double tp = t, tq = tn;                           // the H lanes of a locus are adjacent
        for (uint32_t off = 1; off < H; off <<= 1) {
            tp += __shfl_xor(tp, off, WAVE);
            tq += __shfl_xor(tq, off, WAVE);
        }
        if (live && h == 0 && !stop) {
            tot_prev[l] = tp;
            tot_new[l] = tq;
        }
    } else {
        // MSTEP_EPT elements per thread of the loci with at most one slot: class, theta, A and length in one batch of
        // loads, no second round trip
        const uint64_t n = (uint64_t)L * H;
        const uint64_t i0 = (uint64_t)(blockIdx.x - heavy_blocks - light_blocks) * MSTEP_EPT * blockDim.x + threadIdx.x;
        uint32_t cls[MSTEP_EPT];
        double av[MSTEP_EPT], tv[MSTEP_EPT], lv[MSTEP_EPT];
#pragma unroll
        for (int e = 0; e < MSTEP_EPT; ++e) {
            const uint64_t i = i0 + (uint64_t)e * blockDim.x;
            cls[e] = 3;
            av[e] = tv[e] = 0.0;
            lv[e] = 1.0;
            if (i < n) {
                cls[e] = locus_class[(uint32_t)(i / H)];
                tv[e] = theta[i];
                av[e] = acc[i];                           // one slot: stored by its tile; none: stays 0
                if (acc_extra) av[e] += acc_extra[i];
                if (eff_len) lv[e] = eff_len[i];
            }
        }
#pragma unroll
        for (int e = 0; e < MSTEP_EPT; ++e) {
            const uint64_t i = i0 + (uint64_t)e * blockDim.x;
            const uint32_t l = (uint32_t)(i / H), h = (uint32_t)(i & (H - 1));
            const bool mine = i < n && cls[e] < 2;        // classes 2 and 3: the workgroups above
            double te = 0.0, tne = 0.0;
            if (mine) {
                te = tv[e];
                const double c = te * av[e];
                tne = eff_len ? c / lv[e] : c;
                if (!stop) theta[i] = tne;
            }
            double tp = te, tq = tne;                     // the H lanes of a locus are adjacent, same class
            for (uint32_t off = 1; off < H; off <<= 1) {
                tp += __shfl_xor(tp, off, WAVE);
                tq += __shfl_xor(tq, off, WAVE);
            }
            if (mine && h == 0 && !stop) {
                tot_prev[l] = tp;
                tot_new[l] = tq;
            }
            t += te;
            tn += tne;
        }
    }
    const double a = block_sum(t, lds);
    const double b = block_sum(tn, lds);
    if (threadIdx.x == 0 && !stop) {
        msums[blockIdx.x] = a;
        msums[cap + blockIdx.x] = b;
    }
}

// M-step launch: element-parallel when H is a power of two, thread-per-locus otherwise.
template <int MODE>
int em_launch_mstep(gbrs_em *em) {
    const bool pow2 = (em->H & (em->H - 1)) == 0;
    const double *len = em->has_len ? em->eff_len.p : nullptr;
    if (!pow2) {
        const int nb = em->red_blocks();
        em->msum_blocks = nb;
        hipLaunchKernelGGL(mstep_kernel<MODE>, dim3(nb), dim3(RED_THREADS), 0, em->stream, em->L, em->H,
                           em->theta.p, em->acc.p, len, em->counts.p, em->tot_prev.p, em->tot_new.p,
                           em->msums.p, em->msum_cap, em->scalars.p);
        return GBRS_OK;
    }
    const uint64_t n = (uint64_t)em->L * em->H;
    const unsigned nb = (unsigned)((n + RED_THREADS - 1) / RED_THREADS);
    em->msum_blocks = nb;
    hipLaunchKernelGGL(mstep_elem_kernel<MODE>, dim3(nb), dim3(RED_THREADS), 0, em->stream, em->L, em->H, em->theta.p,
                       em->acc.p, em->acc_needs_extra ? em->tl.acc_extra.p : (const double *)nullptr, len,
                       em->counts.p, em->tot_prev.p, em->tot_new.p, em->msums.p, em->msum_cap, em->scalars.p);
    return GBRS_OK;
}

// The expected read counts theta * A of the last iteration (EMfactory.py:302: the last E-step's posterior summed over the
// reads) equal theta' * len, theta' being what that iteration's M-step left: the fused gather + M-step launch does not
// store them (7.7 of its ~55 MB per iteration at C2), they are made from theta when somebody asks (one more rounding:
// within 2 ulp of the product the unfused kernels store).
__global__ void __launch_bounds__(256)
counts_from_theta_kernel(uint64_t n, const double *__restrict__ theta, const double *__restrict__ eff_len, double *__restrict__ counts) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) counts[i] = eff_len ? theta[i] * eff_len[i] : theta[i];
}
int em_refresh_counts(gbrs_em *em) {
    if (!em->counts_stale) return GBRS_OK;
    const uint64_t n = (uint64_t)em->L * em->H;
    hipLaunchKernelGGL(counts_from_theta_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, em->stream, n, em->theta.p,
                       em->has_len ? em->eff_len.p : (const double *)nullptr, em->counts.p);
    GBRS_HIP_CHECK(hipGetLastError());
    em->counts_stale = false;
    return GBRS_OK;
}

// Everything after the E-step of one EM iteration.
ErrArgs em_err_args(gbrs_em *em, double target_err) {
    ErrArgs ea;
    ea.L = em->L;
    ea.nblocks = (int)em->msum_blocks;
    ea.cap = em->msum_cap;
    ea.tot_prev = em->tot_prev.p;
    ea.tot_new = em->tot_new.p;
    ea.partials = em->msums.p;
    ea.sc = em->scalars.p;
    ea.target_err = target_err;
    ea.err_hist = em->err_hist.p;
    ea.err_hist_cap = em->err_hist_cap;
    ea.n_err_blocks = (uint32_t)std::min<uint64_t>(ERR_BLOCKS, (em->L + RED_THREADS - 1) / RED_THREADS);
    return ea;
}

// Runs a deferred error pass now (before anything that reads or resets the step scalars).
int em_flush_err(gbrs_em *em) {
    if (!em->err_pending) return GBRS_OK;
    em->err_pending = false;
    const ErrArgs ea = em_err_args(em, em->err_pending_target);
    hipLaunchKernelGGL(err_finish_kernel, dim3(ea.n_err_blocks), dim3(RED_THREADS), 0, em->stream, ea);
    GBRS_HIP_CHECK(hipGetLastError());
    return GBRS_OK;
}

// Everything after the E-step of one EM iteration.  `defer`: leave the error pass to the next
// step's gather launch (em_estep_tiles) or to em_flush_err.
// tile layout, H a power of two, A not handed out for an all-reduce: gather and M-step share a launch
bool em_can_fuse_mstep(const gbrs_em *em) {
    static const bool off = [] { const char *e = std::getenv("GBRS_TUNING_NO_FUSED_MSTEP"); return e && std::atoi(e) != 0; }();
    // (half-locus view: the fused kernel's per-locus totals would be per half; gather and M-step stay two launches there)
    return !off && em->layout == 1 && (em->H & (em->H - 1)) == 0 && !em->acc_external && em->view == 1;
}

int em_launch_mstep_gather(gbrs_em *em) {
    const TileLayout &tl = em->tl;
    uint32_t HP = 1;
    while (HP < em->H) HP <<= 1;
    const uint64_t n = (uint64_t)em->L * em->H;
    const unsigned elem_blocks = (unsigned)((n + (uint64_t)RED_THREADS * MSTEP_EPT - 1) / ((uint64_t)RED_THREADS * MSTEP_EPT));
    const unsigned heavy_blocks = (unsigned)tl.n_heavy;         // one workgroup per many-slot locus
    const unsigned light_blocks = (unsigned)((tl.n_light * em->H + RED_THREADS - 1) / RED_THREADS);
    em->msum_blocks = elem_blocks + heavy_blocks + light_blocks;
    hipLaunchKernelGGL(mstep_gather_kernel, dim3(em->msum_blocks), dim3(RED_THREADS), 0, em->stream, em->L,
                       em->H, HP, heavy_blocks, light_blocks, (uint32_t)tl.n_light, tl.heavy_range.p, tl.light_range.p,
                       tl.locus_class.p, tl.partials.p, em->acc.p,
                       tl.n_long ? tl.acc_extra.p : (const double *)nullptr, em->theta.p,
                       em->has_len ? em->eff_len.p : (const double *)nullptr, em->counts.p, em->tot_prev.p,
                       em->tot_new.p, em->msums.p, em->msum_cap, em->scalars.p);
    return GBRS_OK;
}

int em_finish_step(gbrs_em *em, double target_err, bool defer = false, bool fused = false) {
    GBRS_TRY(em_flush_err(em));
    if (fused) GBRS_TRY(em_launch_mstep_gather(em));
    else GBRS_TRY(em_launch_mstep<0>(em));
    em->counts_stale = fused;
    if (defer && em->layout == 1) {
        em->err_pending = true;
        em->err_pending_target = target_err;
        return GBRS_OK;
    }
    const ErrArgs ea = em_err_args(em, target_err);
    hipLaunchKernelGGL(err_finish_kernel, dim3(ea.n_err_blocks), dim3(RED_THREADS), 0, em->stream, ea);
    GBRS_HIP_CHECK(hipGetLastError());
    return GBRS_OK;
}

// One iteration; `ev` (3 events) times it.  An event record is a barrier packet that costs the
// queue ~4 us of idle time on this hardware, so callers time a sample of the iterations (every
// EM_TIME_STRIDE-th), not each one.
constexpr int EM_TIME_STRIDE = 8;
int em_one_step(gbrs_em *em, double target_err, hipEvent_t *ev = nullptr, bool defer_err = false) {
    const bool timed = ev != nullptr && em->time_steps;
    if (timed) GBRS_HIP_CHECK(hipEventRecord(ev[0], em->stream));
    // ev[1] closes the E-step kernel itself (tile layout: before the gather; CSC layout: after the pass)
    em->ev_after_estep = timed && em->layout == 1 ? ev[1] : nullptr;
    const bool fused = em_can_fuse_mstep(em);
    if (fused) GBRS_TRY(em_estep_tiles<false>(em, false, /* skip_gather */ true));
    else GBRS_TRY(em_estep<false>(em));
    em->ev_after_estep = nullptr;
    if (timed && em->layout != 1) GBRS_HIP_CHECK(hipEventRecord(ev[1], em->stream));
    GBRS_TRY(em_finish_step(em, target_err, defer_err, fused));
    if (timed) GBRS_HIP_CHECK(hipEventRecord(ev[2], em->stream));
    return GBRS_OK;
}

int em_read_times(gbrs_em *em) {
    if (!em->time_steps) return GBRS_OK;
    float a = 0.f, b = 0.f;
    if (hipEventElapsedTime(&a, em->ev0, em->ev1) == hipSuccess) em->last_estep_ms = a;
    if (hipEventElapsedTime(&b, em->ev0, em->ev2) == hipSuccess) em->last_step_ms = b;
    return GBRS_OK;
}

int em_finish_prepare(gbrs_em *em, double pseudocount) {
    GBRS_TRY(em_launch_mstep<1>(em));
    em->counts_stale = false;
    const int nb = em->red_blocks();
    if (pseudocount > 0.0) {
        hipLaunchKernelGGL(pseudo_add_kernel, dim3(nb), dim3(RED_THREADS), 0, em->stream, em->L,
                           em->H, pseudocount, em->theta.p, em->partials.p);
        hipLaunchKernelGGL(pseudo_scale_kernel, dim3(nb), dim3(RED_THREADS), 0, em->stream,
                           (uint64_t)em->L * em->H, nb, em->theta.p, em->partials.p, em->scalars.p);
    }
    GBRS_HIP_CHECK(hipMemsetAsync(em->partials.p, 0, em->partials.bytes(), em->stream));
    // A of the loci without any slot must read 0 in the steps (the tile path then writes only the loci
    // it has rows for); prepare's full gather left the long rows' contribution there
    GBRS_HIP_CHECK(hipMemsetAsync(em->acc.p, 0, em->acc.bytes(), em->stream));
    GBRS_HIP_CHECK(hipGetLastError());
    em->prepared = true;
    return GBRS_OK;
}

int em_reset_scalars(gbrs_em *em, bool keep_iters) {
    GBRS_TRY(em_flush_err(em));
    EmScalars host;
    std::memset(&host, 0, sizeof(host));
    if (keep_iters) {
        EmScalars cur;
        GBRS_HIP_CHECK(hipMemcpyAsync(&cur, em->scalars.p, sizeof(cur), hipMemcpyDeviceToHost, em->stream));
        GBRS_HIP_CHECK(hipStreamSynchronize(em->stream));
        host.iters_done = cur.iters_done;
    }
    em->stopped = false;
    GBRS_HIP_CHECK(hipMemcpyAsync(em->scalars.p, &host, sizeof(host), hipMemcpyHostToDevice, em->stream));
    GBRS_HIP_CHECK(hipStreamSynchronize(em->stream));
    return GBRS_OK;
}

int em_ensure_hist(gbrs_em *em, int cap) {
    if (cap <= em->err_hist_cap) return GBRS_OK;
    GBRS_HIP_CHECK(hipStreamSynchronize(em->stream));
    GBRS_TRY(em->err_hist.alloc((size_t)cap));
    em->err_hist_cap = cap;
    return GBRS_OK;
}

// Concatenate the per-haplotype CSC arrays on the device: column id c = h*L + l, col_ptr[c] is the
// offset of the column's first entry in ent_row.  Validates monotone indptr and (host inputs) row ids.
// allowed (host, nullable): uint32[L], bit h set = the entries of (haplotype h, locus l) stay; the other columns
// are dropped on the device (compact_columns_kernel).  src_ptr_dev (nullable) then receives the column offsets of
// the arrays as given, for callers that have to move a second per-entry array the same way.
int upload_csc(uint64_t R, uint32_t L, uint32_t H, const uint32_t *const *indptr,
               const uint32_t *const *indices, bool on_device, DevBuf<uint32_t> &ent_row,
               DevBuf<uint64_t> &col_ptr_dev, uint64_t &n_out, const uint32_t *allowed = nullptr,
               DevBuf<uint64_t> *src_ptr_dev = nullptr, hipStream_t stream = nullptr) {
    const hipMemcpyKind kind = on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice;
    std::vector<uint64_t> col_ptr((size_t)H * L + 1), src_ptr;
    if (allowed) src_ptr.resize((size_t)H * L + 1);
    std::vector<uint32_t> tmp(L + 1);
    uint64_t n = 0, n_src = 0;
    std::vector<uint64_t> hap_off(H + 1);
    for (uint32_t h = 0; h < H; ++h) {
        if (!indptr[h]) return fail(GBRS_ERR_INVALID, "indptr[%u] is NULL", h);
        if (on_device) {
            GBRS_HIP_CHECK(hipMemcpy(tmp.data(), indptr[h], (L + 1) * sizeof(uint32_t), hipMemcpyDeviceToHost));
        } else {
            std::memcpy(tmp.data(), indptr[h], (L + 1) * sizeof(uint32_t));
        }
        if (tmp[0] != 0) return fail(GBRS_ERR_INVALID, "indptr[%u][0] != 0", h);
        hap_off[h] = n_src;
        for (uint32_t l = 0; l < L; ++l) {
            if (tmp[l + 1] < tmp[l]) return fail(GBRS_ERR_INVALID, "indptr[%u] is not non-decreasing at %u", h, l);
            col_ptr[(size_t)h * L + l] = n;
            if (allowed) {
                src_ptr[(size_t)h * L + l] = n_src + tmp[l];
                if ((allowed[l] >> h) & 1u) n += tmp[l + 1] - tmp[l];
            } else {
                n += tmp[l + 1] - tmp[l];
            }
        }
        n_src += tmp[L];
    }
    hap_off[H] = n_src;
    col_ptr[(size_t)H * L] = n;
    GBRS_TRY(col_ptr_dev.alloc(col_ptr.size()));
    GBRS_HIP_CHECK(hipMemcpy(col_ptr_dev.p, col_ptr.data(), col_ptr_dev.bytes(), hipMemcpyHostToDevice));
    GBRS_TRY(ent_row.alloc(std::max<uint64_t>(n, 1)));
    DevBuf<uint32_t> staged;                      // masked: the arrays as given, before the columns move up
    DevBuf<uint64_t> src_local;
    if (allowed) {
        src_ptr[(size_t)H * L] = n_src;
        GBRS_TRY(staged.alloc(std::max<uint64_t>(n_src, 1)));
        DevBuf<uint64_t> &sp = src_ptr_dev ? *src_ptr_dev : src_local;
        GBRS_TRY(sp.alloc(src_ptr.size()));
        GBRS_HIP_CHECK(hipMemcpy(sp.p, src_ptr.data(), sp.bytes(), hipMemcpyHostToDevice));
    }
    uint32_t *dst = allowed ? staged.p : ent_row.p;
    for (uint32_t h = 0; h < H; ++h) {
        const uint64_t cnt = hap_off[h + 1] - hap_off[h];
        if (cnt == 0) continue;
        if (!indices[h]) return fail(GBRS_ERR_INVALID, "indices[%u] is NULL", h);
        GBRS_HIP_CHECK(hipMemcpy(dst + hap_off[h], indices[h], cnt * sizeof(uint32_t), kind));
    }
    if (allowed && n) {
        const DevBuf<uint64_t> &sp = src_ptr_dev ? *src_ptr_dev : src_local;
        hipLaunchKernelGGL(compact_columns_kernel<uint32_t>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, n,
                           H * L, col_ptr_dev.p, sp.p, staged.p, ent_row.p);
        GBRS_HIP_CHECK(hipStreamSynchronize(stream));
        GBRS_HIP_CHECK(hipGetLastError());
    }
    n_out = n;
    return GBRS_OK;
}

// Row ids of uploaded CSC arrays are range-checked on the device before any kernel indexes per-row
// storage with them (host and device inputs alike).
int check_row_ids(uint64_t n, const uint32_t *ent_row, uint64_t R, hipStream_t stream) {
    if (n == 0) return GBRS_OK;
    DevBuf<unsigned int> d_max;
    GBRS_TRY(d_max.alloc(1));
    GBRS_HIP_CHECK(hipMemsetAsync(d_max.p, 0, sizeof(unsigned int), stream));
    hipLaunchKernelGGL(max_row_kernel, dim3(1024), dim3(256), 0, stream, n, ent_row, d_max.p);
    unsigned int mx = 0;
    GBRS_HIP_CHECK(hipMemcpyAsync(&mx, d_max.p, sizeof(mx), hipMemcpyDeviceToHost, stream));
    GBRS_HIP_CHECK(hipStreamSynchronize(stream));
    if (mx >= R) return fail(GBRS_ERR_INVALID, "indices hold row id %u >= num_rows", mx);
    return GBRS_OK;
}

int em_create_impl(uint64_t R, uint32_t L, uint32_t H, const uint32_t *const *indptr,
                   const uint32_t *const *indices, const double *count, const double *eff_len,
                   const uint32_t *allowed, int device, uint32_t flags, bool on_device, gbrs_em_t **out) {
    RoctxRange roctx_range("gbrs_em_create");
    if (!out) return fail(GBRS_ERR_INVALID, "out is NULL");
    *out = nullptr;
    if (H < 1 || H > 32 || L < 1 || R < 1 || R > 0xFFFFFFFFull)
        return fail(GBRS_ERR_INVALID, "The shape must be a tuple of three positive integers (H <= 32, R < 2^32).");
    if (!indptr || !indices) return fail(GBRS_ERR_INVALID, "indptr/indices tables are NULL");
    GBRS_TRY(select_device(device));
    gbrs_em *em = new gbrs_em();
    struct Guard { gbrs_em *p; ~Guard() { if (p) gbrs_em_destroy(p); } } guard{em};
    em->device = device;
    em->R = R; em->L = L; em->H = H; em->flags = flags;
    em->has_count = count != nullptr;
    em->has_len = eff_len != nullptr;
    GBRS_HIP_CHECK(hipStreamCreateWithFlags(&em->stream, hipStreamDefault));
    GBRS_HIP_CHECK(hipEventCreate(&em->ev0));
    GBRS_HIP_CHECK(hipEventCreate(&em->ev1));
    GBRS_HIP_CHECK(hipEventCreate(&em->ev2));
    const hipMemcpyKind kind = on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice;
    uint64_t n = 0;
    StageTimer stg("create");
    em->masked = allowed != nullptr;
    GBRS_TRY(upload_csc(R, L, H, indptr, indices, on_device, em->ent_row, em->col_ptr, n, allowed,
                        (allowed && (flags & GBRS_EM_KEEP_CSC)) ? &em->col_ptr_src : nullptr, em->stream));
    stg.mark("upload csc");
    em->N = n;
    const size_t LH = (size_t)L * H;
    GBRS_TRY(em->den.alloc(R));
    GBRS_TRY(em->theta.alloc(LH));
    GBRS_TRY(em->acc.alloc(LH));
    GBRS_TRY(em->counts.alloc(LH));
    GBRS_TRY(em->scratch_hl.alloc(LH));
    GBRS_TRY(em->tot_prev.alloc(L));
    GBRS_TRY(em->tot_new.alloc(L));
    GBRS_TRY(em->partials.alloc(3 * RED_BLOCKS));
    em->msum_cap = (uint32_t)(((uint64_t)L * H + RED_THREADS - 1) / RED_THREADS * 2 + L + RED_BLOCKS);   // elementwise + light workgroups + one per many-slot locus
    GBRS_TRY(em->msums.alloc(2 * (size_t)em->msum_cap + ERR_BLOCKS));
    GBRS_TRY(em->scalars.alloc(1));
    GBRS_HIP_CHECK(hipMemset(em->scalars.p, 0, sizeof(EmScalars)));
    GBRS_HIP_CHECK(hipMemset(em->theta.p, 0, em->theta.bytes()));
    GBRS_HIP_CHECK(hipMemset(em->acc.p, 0, em->acc.bytes()));
    GBRS_HIP_CHECK(hipMemset(em->partials.p, 0, em->partials.bytes()));
    GBRS_HIP_CHECK(hipMemset(em->msums.p, 0, em->msums.bytes()));
    GBRS_HIP_CHECK(hipMemset(em->counts.p, 0, em->counts.bytes()));
    if (count) {
        GBRS_TRY(em->count.alloc(R));
        GBRS_HIP_CHECK(hipMemcpy(em->count.p, count, R * sizeof(double), kind));
    }
    if (eff_len) {
        GBRS_TRY(em->eff_len.alloc(LH));
        GBRS_HIP_CHECK(hipMemcpy(em->scratch_hl.p, eff_len, LH * sizeof(double), kind));
        hipLaunchKernelGGL(transpose_hl_to_lh, dim3(1024), dim3(256), 0, em->stream, L, H,
                           em->scratch_hl.p, em->eff_len.p);
        GBRS_HIP_CHECK(hipStreamSynchronize(em->stream));
    }
    GBRS_HIP_CHECK(hipDeviceSynchronize());
    GBRS_TRY(check_row_ids(n, em->ent_row.p, R, em->stream));
    stg.mark("vectors, checks");
    if ((flags & GBRS_EM_DETERMINISTIC) && ((flags & GBRS_EM_LAYOUT_CSC) || H > 16 || n >= 0xFFFFFFFFull))
        return fail(GBRS_ERR_UNSUPPORTED, "GBRS_EM_DETERMINISTIC needs the tiled layout (H <= 16, not GBRS_EM_LAYOUT_CSC): "
                                          "the CSC kernels accumulate with global float atomics");
    if (!(flags & GBRS_EM_LAYOUT_CSC) && H <= 16 && n < 0xFFFFFFFFull) {
        // Row order inside a tile: the stream order (gbrs_hip.h) by default - every lane walks a
        // contiguous piece of the tile's sorted rows, so it stays on one locus list for long stretches
        // (E-step on C2: raw reads 0.158 -> 0.152 ms, merged distinct rows 0.110 -> 0.056 ms against
        // the interleaved order that used to be their default).
        int row_order = 2;
        if (flags & GBRS_EM_FORCE_INTERLEAVE) row_order = 1;
        else if (flags & GBRS_EM_NO_STREAMS) {
            const bool distinct = count != nullptr || (flags & GBRS_EM_MERGE_IDENTICAL_ROWS);
            row_order = (distinct && !(flags & GBRS_EM_NO_INTERLEAVE)) ? 1 : 0;
        }
        em->tl.retain_temporaries = (flags & GBRS_EM_ONE_SHOT) != 0;
        // 16 haplotypes as half-loci on the 8-haplotype kernels (em_layout.h; the review's "two halves of 8"): built, parity-green,
        // NO gain - one GPU's shard of config 5: E-step 0.1518 ms against 0.1525 (the words double, the cost per word halves) and
        // the iteration 0.1997 against 0.1744 (gather and M-step as two launches over every element).  GBRS_TUNING_HALF_LOCI=1
        // switches it on; weighted rows and the deterministic mode never take it.
        {
            const char *env = std::getenv("GBRS_TUNING_HALF_LOCI");
            const bool want = env ? std::atoi(env) != 0 : false;
            const bool weighted = count != nullptr || (flags & GBRS_EM_MERGE_IDENTICAL_ROWS);
            em->view = (H == 16 && want && !weighted && !(flags & GBRS_EM_DETERMINISTIC) && (uint64_t)L * 2 < (1u << 27)) ? 2u : 1u;
        }
        GBRS_TRY(build_tile_layout(em->tl, R, em->tL(), em->tH(), n, em->ent_row.p, em->col_ptr.p,
                                   count ? em->count.p : nullptr, (flags & GBRS_EM_MERGE_IDENTICAL_ROWS) != 0,
                                   row_order, (flags & GBRS_EM_DETERMINISTIC) != 0,
                                   em->stream, (flags & GBRS_EM_SIDE_BY_SIDE) ? 2u : 1u,
                                   (flags & GBRS_EM_NO_LOCUS_SETS) == 0 && !count &&
                                       !(flags & GBRS_EM_MERGE_IDENTICAL_ROWS),     // (weighted rows: their tiles are dictionary-bound)
                                   0, em->view));
        em->layout = 1;
        {
            // persistent E-step workgroups: one per place the chip has for them (tile_estep_kernel's launch bounds: 3 per CU
            // for unweighted rows of <= 8 haplotypes, else 2), shared between handles that run side by side
            // GBRS_TUNING_PERSISTENT=1 switches them on: built, parity-green and measured in round 4 - 8-10 % SLOWER than one
            // workgroup per tile on the C2 sample (profiles/r04_estep_experiments.txt), so off by default
            const char *env = std::getenv("GBRS_TUNING_PERSISTENT");
            int n_cu = 0;
            GBRS_HIP_CHECK(hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, device));
            const unsigned per_cu = (em->tl.weighted || em->tH() > 8) ? 2u : 3u;
            const unsigned share = (flags & GBRS_EM_SIDE_BY_SIDE) ? 2u : 1u;
            unsigned groups = per_cu * (unsigned)std::max(n_cu, 1) / share;
            if (const char *g = std::getenv("GBRS_TUNING_PERSISTENT_GROUPS"); g && std::atoi(g) > 0) groups = (unsigned)std::atoi(g);
            em->persist_groups = (env && std::atoi(env) != 0) ? std::max(groups, 1u) : 0u;
        }
        stg.mark("build_tile_layout");
        // the CSC copy and the per-row denominators are only needed by layout 0 (and, until
        // gbrs_em_set_initial_values has run, when the caller announced stored values)
        em->keep_csc = (flags & GBRS_EM_KEEP_CSC) != 0;
        if (!em->keep_csc) {
            if (em->tl.retain_temporaries) {  // one-shot process: left to the handle's destructor (common.h, DeferFrees)
                DeferFrees with_the_layout(&em->tl.retired, &em->tl.retired_bytes);
                em->ent_row.release();
                em->den.release();
            } else {
                em->ent_row.release();
                em->den.release();
            }
        }
        stg.mark("release csc copy");
    }
    guard.p = nullptr;
    *out = em;
    return GBRS_OK;
}

}  // namespace

struct gbrs_compress {
    int device = 0;
    uint32_t L = 0, H = 0;
    CompressResult res;
};

extern "C" {

int gbrs_compress_create(uint64_t R, uint32_t L, uint32_t H, const uint32_t *const *indptr,
                         const uint32_t *const *indices, const double *count, int device,
                         gbrs_compress_t **out, uint64_t *num_ecs, uint64_t *nnz_per_hap) {
    RoctxRange roctx_range("gbrs_compress_create");
    if (!out) return fail(GBRS_ERR_INVALID, "out is NULL");
    *out = nullptr;
    if (H < 1 || H > 16 || L < 1 || R < 1 || R > 0xFFFFFFFFull || !indptr || !indices)
        return fail(GBRS_ERR_INVALID, "The shape must be a tuple of three positive integers (H <= 16, R < 2^32).");
    GBRS_TRY(select_device(device));
    hipStream_t s = nullptr;
    GBRS_HIP_CHECK(hipStreamCreateWithFlags(&s, hipStreamDefault));
    struct SG { hipStream_t s; ~SG() { (void)hipStreamDestroy(s); } } sg{s};
    DevBuf<uint32_t> ent_row;
    DevBuf<uint64_t> col_ptr;
    DevBuf<double> d_count;
    uint64_t n = 0;
    GBRS_TRY(upload_csc(R, L, H, indptr, indices, false, ent_row, col_ptr, n));
    if (count) {
        GBRS_TRY(d_count.alloc(R));
        GBRS_HIP_CHECK(hipMemcpy(d_count.p, count, R * sizeof(double), hipMemcpyHostToDevice));
    }
    GBRS_HIP_CHECK(hipDeviceSynchronize());
    GBRS_TRY(check_row_ids(n, ent_row.p, R, s));
    gbrs_compress *c = new gbrs_compress();
    c->device = device; c->L = L; c->H = H;
    const int st = compress_device(c->res, R, L, H, n, ent_row.p, col_ptr.p, count ? d_count.p : nullptr, s);
    if (st != GBRS_OK) { delete c; return st; }
    if (num_ecs) *num_ecs = c->res.num_ecs;
    if (nnz_per_hap) {
        std::vector<uint64_t> cp((size_t)H * L + 1);
        GBRS_HIP_CHECK(hipMemcpy(cp.data(), c->res.col_ptr.p, cp.size() * 8, hipMemcpyDeviceToHost));
        for (uint32_t h = 0; h < H; ++h) nnz_per_hap[h] = cp[(size_t)(h + 1) * L] - cp[(size_t)h * L];
    }
    *out = c;
    return GBRS_OK;
}

int gbrs_compress_get(gbrs_compress_t *c, uint32_t *const *indptr_out, uint32_t *const *indices_out, double *count_out) {
    if (!c || !indptr_out || !indices_out) return fail(GBRS_ERR_INVALID, "NULL argument");
    GBRS_TRY(select_device(c->device));
    const uint32_t L = c->L, H = c->H;
    std::vector<uint64_t> cp((size_t)H * L + 1);
    GBRS_HIP_CHECK(hipMemcpy(cp.data(), c->res.col_ptr.p, cp.size() * 8, hipMemcpyDeviceToHost));
    for (uint32_t h = 0; h < H; ++h) {
        const uint64_t base = cp[(size_t)h * L], cnt = cp[(size_t)(h + 1) * L] - base;
        for (uint32_t l = 0; l <= L; ++l) indptr_out[h][l] = (uint32_t)(cp[(size_t)h * L + l] - base);
        if (cnt) GBRS_HIP_CHECK(hipMemcpy(indices_out[h], c->res.indices.p + base, cnt * 4, hipMemcpyDeviceToHost));
    }
    if (count_out && c->res.num_ecs)
        GBRS_HIP_CHECK(hipMemcpy(count_out, c->res.count.p, c->res.num_ecs * 8, hipMemcpyDeviceToHost));
    return GBRS_OK;
}

int gbrs_compress_destroy(gbrs_compress_t *c) {
    if (c) { (void)hipSetDevice(c->device); delete c; }
    return GBRS_OK;
}

}  // extern "C"

extern "C" {

int gbrs_em_create(uint64_t R, uint32_t L, uint32_t H, const uint32_t *const *indptr,
                   const uint32_t *const *indices, const double *count, const double *eff_len,
                   int device, uint32_t flags, gbrs_em_t **out) {
    return em_create_impl(R, L, H, indptr, indices, count, eff_len, nullptr, device, flags, false, out);
}

int gbrs_em_create_device(uint64_t R, uint32_t L, uint32_t H, const uint32_t *const *indptr,
                          const uint32_t *const *indices, const double *count, const double *eff_len,
                          int device, uint32_t flags, gbrs_em_t **out) {
    return em_create_impl(R, L, H, indptr, indices, count, eff_len, nullptr, device, flags, true, out);
}

int gbrs_em_create_masked(uint64_t R, uint32_t L, uint32_t H, const uint32_t *const *indptr,
                          const uint32_t *const *indices, const double *count, const double *eff_len,
                          const uint32_t *allowed, int device, uint32_t flags, gbrs_em_t **out) {
    if (allowed && H < 32)
        for (uint32_t l = 0; l < L; ++l)
            if (allowed[l] >> H) return fail(GBRS_ERR_INVALID, "allowed[%u] names a haplotype >= num_haps", l);
    return em_create_impl(R, L, H, indptr, indices, count, eff_len, allowed, device, flags, false, out);
}

int gbrs_em_create_masked_device(uint64_t R, uint32_t L, uint32_t H, const uint32_t *const *indptr,
                                 const uint32_t *const *indices, const double *count, const double *eff_len,
                                 const uint32_t *allowed, int device, uint32_t flags, gbrs_em_t **out) {
    if (allowed && H < 32)
        for (uint32_t l = 0; l < L; ++l)
            if (allowed[l] >> H) return fail(GBRS_ERR_INVALID, "allowed[%u] names a haplotype >= num_haps", l);
    return em_create_impl(R, L, H, indptr, indices, count, eff_len, allowed, device, flags, true, out);
}

namespace {
// partial sums of count/nnz_row into em->acc (every element written)
int em_prepare_partial(gbrs_em *em) {
    GBRS_TRY(select_device(em->device));
    GBRS_TRY(em_reset_scalars(em, false));
    if (em->has_init) {                       // stored alignment values: their normalised column sums
        GBRS_HIP_CHECK(hipMemcpyAsync(em->acc.p, em->acc_init.p, em->acc.bytes(), hipMemcpyDeviceToDevice, em->stream));
        em->acc_needs_extra = false;
        return GBRS_OK;
    }
    return em_estep<true>(em, true);
}
}  // namespace

int gbrs_em_set_initial_values(gbrs_em_t *em, const double *const *values) {
    if (!em || !values) return fail(GBRS_ERR_INVALID, "NULL argument");
    if (!em->ent_row.p || !em->den.p)
        return fail(GBRS_ERR_STATE, "the handle no longer holds the CSC arrays: create it with GBRS_EM_KEEP_CSC "
                                    "and call gbrs_em_set_initial_values once, before prepare");
    GBRS_TRY(select_device(em->device));
    const uint64_t n = em->N;
    const size_t LH = (size_t)em->L * em->H;
    GBRS_TRY(em->acc_init.alloc(LH));
    GBRS_HIP_CHECK(hipMemsetAsync(em->acc_init.p, 0, em->acc_init.bytes(), em->stream));
    if (n) {
        DevBuf<double> vals;
        DevBuf<int> bad;
        GBRS_TRY(vals.alloc(n));
        GBRS_TRY(bad.alloc(1));
        // values[h] lines up with indices[h] as given to create: under a haplotype mask they are staged whole and
        // then moved the way the row ids were (compact_columns_kernel)
        const DevBuf<uint64_t> &given = em->masked ? em->col_ptr_src : em->col_ptr;
        if (!given.p) return fail(GBRS_ERR_STATE, "the handle no longer knows the layout of the caller's arrays");
        std::vector<uint64_t> cp((size_t)em->H * em->L + 1);
        GBRS_HIP_CHECK(hipMemcpy(cp.data(), given.p, cp.size() * 8, hipMemcpyDeviceToHost));
        DevBuf<double> staged;
        if (em->masked) GBRS_TRY(staged.alloc(std::max<uint64_t>(cp.back(), 1)));
        double *dst = em->masked ? staged.p : vals.p;
        for (uint32_t h = 0; h < em->H; ++h) {
            const uint64_t b = cp[(size_t)h * em->L], cnt = cp[(size_t)(h + 1) * em->L] - b;
            if (!cnt) continue;
            if (!values[h]) return fail(GBRS_ERR_INVALID, "values[%u] is NULL", h);
            GBRS_HIP_CHECK(hipMemcpy(dst + b, values[h], cnt * sizeof(double), hipMemcpyHostToDevice));
        }
        if (em->masked) {
            hipLaunchKernelGGL(compact_columns_kernel<double>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, em->stream,
                               n, em->H * em->L, em->col_ptr.p, em->col_ptr_src.p, staged.p, vals.p);
            GBRS_HIP_CHECK(hipStreamSynchronize(em->stream));
        }
        GBRS_HIP_CHECK(hipMemsetAsync(bad.p, 0, sizeof(int), em->stream));
        GBRS_HIP_CHECK(hipMemsetAsync(em->den.p, 0, em->den.bytes(), em->stream));
        const unsigned grid = (unsigned)((n + 255) / 256);
        hipLaunchKernelGGL(csc_den_values_kernel, dim3(grid), dim3(256), 0, em->stream, n, em->ent_row.p, vals.p, em->den.p);
        hipLaunchKernelGGL(csc_acc_values_kernel, dim3(grid), dim3(256), 0, em->stream, n, em->H * em->L, em->L, em->H,
                           em->col_ptr.p, em->ent_row.p, vals.p, em->has_count ? em->count.p : (const double *)nullptr,
                           em->den.p, em->acc_init.p, bad.p);
        int hbad = 0;
        GBRS_HIP_CHECK(hipMemcpyAsync(&hbad, bad.p, sizeof(int), hipMemcpyDeviceToHost, em->stream));
        GBRS_HIP_CHECK(hipStreamSynchronize(em->stream));
        GBRS_HIP_CHECK(hipGetLastError());
        if (hbad) {
            em->acc_init.release();
            return fail(GBRS_ERR_FLOAT, "invalid value encountered in divide (a read's stored alignment values are "
                                        "negative or add up to zero)");
        }
    }
    em->has_init = true;
    if (em->layout == 1 && em->keep_csc) {     // the tiled layout needs neither array from here on
        em->ent_row.release();
        em->den.release();
        em->col_ptr_src.release();
        em->keep_csc = false;
    }
    return GBRS_OK;
}

int gbrs_em_prepare_partial(gbrs_em_t *em, void **partial_dev, uint64_t *n_elems) {
    if (!em) return fail(GBRS_ERR_INVALID, "handle is NULL");
    em->acc_external = true;                  // sharded rows: the caller all-reduces the whole A vector from now on
    GBRS_TRY(em_prepare_partial(em));
    if (partial_dev) *partial_dev = em->acc.p;
    if (n_elems) *n_elems = (uint64_t)em->L * em->H;
    return GBRS_OK;
}

int gbrs_em_finish_prepare(gbrs_em_t *em, double pseudocount) {
    if (!em) return fail(GBRS_ERR_INVALID, "handle is NULL");
    GBRS_TRY(select_device(em->device));
    GBRS_TRY(em_finish_prepare(em, pseudocount));
    EmScalars host;
    return em_check_float(em, host);
}

int gbrs_em_prepare(gbrs_em_t *em, double pseudocount) {
    RoctxRange roctx_range("gbrs_em_prepare");
    if (!em) return fail(GBRS_ERR_INVALID, "handle is NULL");
    GBRS_TRY(em_prepare_partial(em));
    return gbrs_em_finish_prepare(em, pseudocount);
}

int gbrs_em_estep_partial(gbrs_em_t *em, void **partial_dev, uint64_t *n_elems) {
    if (!em) return fail(GBRS_ERR_INVALID, "handle is NULL");
    if (!em->prepared) return fail(GBRS_ERR_STATE, "prepare() has not been called");
    GBRS_TRY(select_device(em->device));
    // a run that met its stopping rule (gbrs_em_run, or the pair's rule: gbrs_em_pair_check) left the device's stop flag
    // set, which turns every step kernel into a no-op; a step asked for by hand is applied anyway, as in gbrs_em_step
    if (em->stopped) GBRS_TRY(em_reset_scalars(em, true));
    em->acc_external = true;
    GBRS_TRY(em_estep<false>(em, true));
    if (partial_dev) *partial_dev = em->acc.p;
    if (n_elems) *n_elems = (uint64_t)em->L * em->H;
    return GBRS_OK;
}

int gbrs_em_finish_step(gbrs_em_t *em, double *err_sum_out) {
    if (!em) return fail(GBRS_ERR_INVALID, "handle is NULL");
    GBRS_TRY(select_device(em->device));
    // without a request for err_sum the error pass is deferred into the next E-step launch
    GBRS_TRY(em_finish_step(em, -1.0, /* defer */ err_sum_out == nullptr));
    if (em->pair_ev_mstep) GBRS_HIP_CHECK(hipEventRecord(em->pair_ev_mstep, em->stream));
    if (err_sum_out) {
        EmScalars host;
        GBRS_TRY(em_check_float(em, host));
        *err_sum_out = host.err_sum;
    }
    return GBRS_OK;
}

int gbrs_em_step(gbrs_em_t *em, int n_iters, double *err_sum_out) {
    RoctxRange roctx_range("gbrs_em_step");
    if (!em) return fail(GBRS_ERR_INVALID, "handle is NULL");
    if (!em->prepared) return fail(GBRS_ERR_STATE, "prepare() has not been called");
    GBRS_TRY(select_device(em->device));
    // a run that met its stopping rule left the device's stop flag set; a step asked for by hand is applied anyway
    // (EMfactory.update_allelic_expression knows no stopping rule)
    if (em->stopped) GBRS_TRY(em_reset_scalars(em, true));
    const int timed = em->time_steps ? std::min((n_iters + EM_TIME_STRIDE - 1) / EM_TIME_STRIDE, 64) : 0;
    while ((int)em->ev_pool.size() < 3 * timed) {
        hipEvent_t e;
        GBRS_HIP_CHECK(hipEventCreate(&e));
        em->ev_pool.push_back(e);
    }
    for (int i = 0; i < n_iters; ++i) {
        const int slot = i / EM_TIME_STRIDE;
        const bool t = i % EM_TIME_STRIDE == 0 && slot < timed;
        GBRS_TRY(em_one_step(em, -1.0, t ? &em->ev_pool[3 * slot] : nullptr, i + 1 < n_iters));
    }
    EmScalars host;
    GBRS_TRY(em_check_float(em, host));
    if (timed > 0) {
        double se = 0.0, ss = 0.0;
        for (int i = 0; i < timed; ++i) {
            float a = 0.f, b = 0.f;
            (void)hipEventElapsedTime(&a, em->ev_pool[3 * i], em->ev_pool[3 * i + 1]);
            (void)hipEventElapsedTime(&b, em->ev_pool[3 * i], em->ev_pool[3 * i + 2]);
            se += a;
            ss += b;
        }
        em->last_estep_ms = se / timed;
        em->last_step_ms = ss / timed;
    }
    if (err_sum_out) *err_sum_out = host.err_sum;
    return GBRS_OK;
}

int gbrs_em_run(gbrs_em_t *em, int model, double tol, int max_iters, int *n_iters_out,
                double *err_hist, int err_hist_cap, double *elapsed_s) {
    RoctxRange roctx_range("gbrs_em_run");
    if (!em) return fail(GBRS_ERR_INVALID, "handle is NULL");
    if (model < 1 || model > 4)
        return fail(GBRS_ERR_INVALID, "The read normalization model should be 1, 2, 3, or 4.");
    if (model != 4)
        return fail(GBRS_ERR_UNSUPPORTED, "multiread model %d is not implemented by the HIP path (only Model 4)", model);
    if (!em->prepared) return fail(GBRS_ERR_STATE, "prepare() has not been called");
    GBRS_TRY(select_device(em->device));
    if (max_iters < 0) max_iters = 0;
    GBRS_TRY(em_ensure_hist(em, std::max(max_iters, 1)));
    GBRS_TRY(em_reset_scalars(em, false));
    const double target = 1000000.0 * tol;
    // the reference tests `err_sum > target` with err_sum = 1e6 before the first step
    int done = 0;
    EmScalars host;
    std::memset(&host, 0, sizeof(host));
    if (1000000.0 > target) {
        // Steps are enqueued in batches of 8 (one host synchronisation each); a device-side stop flag
        // turns the steps after the stopping iteration into no-ops, so theta is exactly the stopping
        // iteration's value.
        const int batch = 8;
        bool first = true;
        const auto t_start = std::chrono::steady_clock::now();
        while (done < max_iters) {
            const int nb = std::min(batch, max_iters - done);
            hipEvent_t ev[3] = {em->ev0, em->ev1, em->ev2};
            for (int i = 0; i < nb; ++i) {     // the run's first step is timed; a batch's last error pass is not deferred
                GBRS_TRY(em_one_step(em, target, first ? ev : nullptr, i + 1 < nb));
                first = false;
            }
            GBRS_TRY(em_check_float(em, host));
            if (elapsed_s) {       // the host sees a batch at a time: its iterations share the batch's completion time
                const double t = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_start).count();
                for (int i = done; i < host.iters_done && i < err_hist_cap; ++i) elapsed_s[i] = t;
            }
            done = host.iters_done;
            if (host.stop) break;
        }
    }
    em_read_times(em);
    if (n_iters_out) *n_iters_out = done;
    if (err_hist && err_hist_cap > 0 && done > 0) {
        const int ncopy = std::min(done, err_hist_cap);
        GBRS_HIP_CHECK(hipMemcpy(err_hist, em->err_hist.p, ncopy * sizeof(double), hipMemcpyDeviceToHost));
    }
    return GBRS_OK;
}

int gbrs_em_get(gbrs_em_t *em, double *theta, double *expected_counts) {
    if (!em) return fail(GBRS_ERR_INVALID, "handle is NULL");
    GBRS_TRY(select_device(em->device));
    const size_t LH = (size_t)em->L * em->H;
    for (int which = 0; which < 2; ++which) {
        double *dst = which == 0 ? theta : expected_counts;
        if (!dst) continue;
        if (which == 1) GBRS_TRY(em_refresh_counts(em));
        hipLaunchKernelGGL(transpose_lh_to_hl, dim3(1024), dim3(256), 0, em->stream, em->L, em->H,
                           which == 0 ? em->theta.p : em->counts.p, em->scratch_hl.p);
        GBRS_HIP_CHECK(hipMemcpyAsync(dst, em->scratch_hl.p, LH * sizeof(double), hipMemcpyDeviceToHost, em->stream));
        GBRS_HIP_CHECK(hipStreamSynchronize(em->stream));
    }
    return GBRS_OK;
}

int gbrs_em_set_theta(gbrs_em_t *em, const double *theta) {
    if (!em || !theta) return fail(GBRS_ERR_INVALID, "NULL argument");
    GBRS_TRY(select_device(em->device));
    const size_t LH = (size_t)em->L * em->H;
    // the expected counts stay those of the last iteration (the reports rescale theta to TPM in place and then ask for
    // the counts, EMfactory.py:352-354 before :302): made from the theta about to be replaced, if not stored yet
    GBRS_TRY(em_refresh_counts(em));
    GBRS_HIP_CHECK(hipMemcpyAsync(em->scratch_hl.p, theta, LH * sizeof(double), hipMemcpyHostToDevice, em->stream));
    hipLaunchKernelGGL(transpose_hl_to_lh, dim3(1024), dim3(256), 0, em->stream, em->L, em->H,
                       em->scratch_hl.p, em->theta.p);
    GBRS_HIP_CHECK(hipStreamSynchronize(em->stream));
    em->prepared = true;
    return GBRS_OK;
}

int gbrs_em_group_sums(gbrs_em_t *em, int64_t G, const int64_t *group_ptr, const int64_t *members,
                       int which, double *out) {
    if (!em || !group_ptr || !out || G < 0) return fail(GBRS_ERR_INVALID, "bad argument");
    if (G == 0) return GBRS_OK;
    GBRS_TRY(select_device(em->device));
    const int64_t nm = group_ptr[G];
    for (int64_t g = 0; g < G; ++g)
        if (group_ptr[g + 1] < group_ptr[g]) return fail(GBRS_ERR_INVALID, "group_ptr not monotone");
    for (int64_t k = 0; k < nm; ++k)
        if (members[k] < 0 || members[k] >= (int64_t)em->L)
            return fail(GBRS_ERR_INVALID, "group member %lld out of range", (long long)members[k]);
    DevBuf<int64_t> d_ptr, d_mem;
    DevBuf<double> d_out;
    GBRS_TRY(d_ptr.alloc(G + 1));
    GBRS_TRY(d_mem.alloc(std::max<int64_t>(nm, 1)));
    GBRS_TRY(d_out.alloc((size_t)G * em->H));
    GBRS_HIP_CHECK(hipMemcpyAsync(d_ptr.p, group_ptr, (G + 1) * sizeof(int64_t), hipMemcpyHostToDevice, em->stream));
    if (nm) GBRS_HIP_CHECK(hipMemcpyAsync(d_mem.p, members, nm * sizeof(int64_t), hipMemcpyHostToDevice, em->stream));
    const int64_t total = G * (int64_t)em->H;
    if (which != 0) GBRS_TRY(em_refresh_counts(em));
    hipLaunchKernelGGL(group_sum_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, em->stream,
                       em->H, G, d_ptr.p, d_mem.p, which == 0 ? em->theta.p : em->counts.p, d_out.p);
    GBRS_HIP_CHECK(hipMemcpyAsync(out, d_out.p, (size_t)G * em->H * sizeof(double), hipMemcpyDeviceToHost, em->stream));
    GBRS_HIP_CHECK(hipStreamSynchronize(em->stream));
    return GBRS_OK;
}

void *gbrs_em_stream(gbrs_em_t *em) { return em ? (void *)em->stream : nullptr; }

#if defined(GBRS_DIAG_TILE_TIMES)
// diagnostic build only: device buffer of 8 x uint64 per E-step workgroup (nullptr switches the stamps off)
int gbrs_debug_set_tile_stamps(void *device_ptr) {
    unsigned long long *p = static_cast<unsigned long long *>(device_ptr);
    GBRS_HIP_CHECK(hipMemcpyToSymbol(HIP_SYMBOL(g_tile_stamps), &p, sizeof(p)));
    return GBRS_OK;
}
#endif

int gbrs_em_set_stream(gbrs_em_t *em, void *stream) {
    if (!em) return fail(GBRS_ERR_INVALID, "handle is NULL");
    GBRS_TRY(select_device(em->device));
    GBRS_HIP_CHECK(hipStreamSynchronize(em->stream));
    if (em->own_stream && em->stream) (void)hipStreamDestroy(em->stream);
    em->stream = (hipStream_t)stream;
    em->own_stream = false;
    return GBRS_OK;
}

int gbrs_em_sync(gbrs_em_t *em) {
    if (!em) return fail(GBRS_ERR_INVALID, "handle is NULL");
    GBRS_TRY(em_flush_err(em));
    GBRS_HIP_CHECK(hipStreamSynchronize(em->stream));
    return GBRS_OK;
}

int gbrs_em_pair_begin(gbrs_em_t *a, gbrs_em_t *b, int max_iters) {
    if (!a || !b || a == b) return fail(GBRS_ERR_INVALID, "two distinct handles are needed");
    if (a->device != b->device) return fail(GBRS_ERR_INVALID, "the two handles of a pair live on one device");
    GBRS_TRY(select_device(a->device));
    for (gbrs_em *em : {a, b}) {
        GBRS_TRY(em_reset_scalars(em, false));           // (synchronises the handle's stream)
        if (!em->pair_ev_mstep) GBRS_HIP_CHECK(hipEventCreateWithFlags(&em->pair_ev_mstep, hipEventDisableTiming));
    }
    if (!a->pair_ev_err) GBRS_HIP_CHECK(hipEventCreateWithFlags(&a->pair_ev_err, hipEventDisableTiming));
    const int cap = std::max(max_iters, 1);
    if (!a->pair_sc.p) GBRS_TRY(a->pair_sc.alloc(1));
    if (cap > a->pair_hist_cap) {
        GBRS_TRY(a->pair_hist.alloc((size_t)cap));
        a->pair_hist_cap = cap;
    }
    GBRS_HIP_CHECK(hipMemset(a->pair_sc.p, 0, sizeof(gbrs_em::PairScalars)));
    return GBRS_OK;
}

int gbrs_em_pair_check(gbrs_em_t *a, gbrs_em_t *b, double tol) {
    if (!a || !b || !a->pair_sc.p || !a->pair_ev_err || !a->pair_ev_mstep || !b->pair_ev_mstep)
        return fail(GBRS_ERR_STATE, "gbrs_em_pair_begin has not been called on this pair");
    GBRS_TRY(select_device(a->device));
    // on b's stream (whose M-step has just been enqueued), after a's M-step; a's next M-step waits for the verdict
    GBRS_HIP_CHECK(hipStreamWaitEvent(b->stream, a->pair_ev_mstep, 0));
    static_assert(sizeof(PairState) == sizeof(gbrs_em::PairScalars), "one struct, two names");
    const PairSide sa{a->L, (int)a->msum_blocks, a->msum_cap, a->tot_prev.p, a->tot_new.p, a->msums.p, a->scalars.p};
    const PairSide sb{b->L, (int)b->msum_blocks, b->msum_cap, b->tot_prev.p, b->tot_new.p, b->msums.p, b->scalars.p};
    hipLaunchKernelGGL(pair_err_kernel, dim3(1), dim3(1024), 0, b->stream, sa, sb, 1000000.0 * tol,
                       reinterpret_cast<PairState *>(a->pair_sc.p), a->pair_hist.p, a->pair_hist_cap);
    GBRS_HIP_CHECK(hipGetLastError());
    GBRS_HIP_CHECK(hipEventRecord(a->pair_ev_err, b->stream));
    GBRS_HIP_CHECK(hipStreamWaitEvent(a->stream, a->pair_ev_err, 0));
    return GBRS_OK;
}

int gbrs_em_pair_status(gbrs_em_t *a, gbrs_em_t *b, int *iters_done, int *stopped, double *err_hist, int err_hist_cap) {
    if (!a || !b || !a->pair_sc.p) return fail(GBRS_ERR_STATE, "gbrs_em_pair_begin has not been called on this pair");
    GBRS_TRY(select_device(a->device));
    EmScalars ha, hb;
    GBRS_TRY(em_check_float(a, ha));                     // flushes deferred passes, synchronises, reports 0/0
    GBRS_TRY(em_check_float(b, hb));
    gbrs_em::PairScalars ps;
    GBRS_HIP_CHECK(hipMemcpy(&ps, a->pair_sc.p, sizeof(ps), hipMemcpyDeviceToHost));
    if (iters_done) *iters_done = ps.iters;
    if (stopped) *stopped = ps.stop;
    if (err_hist && err_hist_cap > 0 && ps.iters > 0)
        GBRS_HIP_CHECK(hipMemcpy(err_hist, a->pair_hist.p, std::min(ps.iters, std::min(err_hist_cap, a->pair_hist_cap)) * sizeof(double),
                                 hipMemcpyDeviceToHost));
    return GBRS_OK;
}

int gbrs_em_info(gbrs_em_t *em, gbrs_em_info_t *info) {
    if (!em || !info) return fail(GBRS_ERR_INVALID, "NULL argument");
    std::memset(info, 0, sizeof(*info));
    const uint64_t HL = (uint64_t)em->H * em->L;
    info->num_rows = em->R;
    info->num_entries = em->N;
    info->num_device_rows = em->R;
    info->num_device_words = em->N;
    info->algorithmic_bytes = 4 * em->N + 4 * (em->R + 1) + (em->has_count ? 8 * em->R : 0) +
                              16 * HL + (em->has_len ? 8 * HL : 0);
    // layout 0: two passes over the row ids + den zero/atomic/read + theta/acc traffic
    info->bytes_per_iter = 8 * em->N + 8 * em->R * 3 + 8 * HL * 4;
    info->estep_bytes = 8 * em->N + 8 * em->R * 3 + 8 * HL * 2;
    if (em->layout == 1) {
        const TileLayout &tl = em->tl;
        info->num_device_rows = tl.n_rows + tl.n_long;
        info->num_tiles = tl.n_tiles;
        info->num_slots = tl.n_slots;
        info->num_long_rows = tl.n_long;
        info->num_device_words = tl.n_batches * 64;
        // E-step: word stream, tile headers, dictionary, theta gather + partial store per slot,
        // [row weights]; gather: slot index + partials read back + acc store; M-step etc.: 5 H*L vectors
        // rows = theta rows gathered and sum rows stored by the tiles: one per slot, one per (slot, member) for a locus set
        const uint64_t rows = tl.n_dest_rows ? tl.n_dest_rows : tl.n_slots;
        info->bytes_per_iter = 4 * tl.n_batches * 64 + 16 * tl.n_tiles + 4 * tl.n_slots +
                               8 * rows * em->tH() * 3 + 4 * rows + 4 * ((uint64_t)em->tL() + 1) +
                               (tl.weighted ? 8 * tl.n_batches * 64 : 0) + 8 * HL * 6;
        info->num_heavy_loci = tl.n_heavy;
        info->num_light_loci = tl.n_light;
        // the E-step launch alone: words, tile headers, dictionary + slot destinations, theta gathered
        // once per slot, one partial-sum row stored per slot [, the per-word row weights]
        info->estep_bytes = 4 * tl.n_batches * 64 + 16 * tl.n_tiles + 4 * tl.n_slots + 4 * rows +
                            8 * rows * em->tH() * 2 + (tl.weighted ? 8 * tl.n_batches * 64 : 0) +
                            (tl.n_sets ? 8 * tl.n_slots : 0);       // dict_b / dest_b beside the dictionary
        if (tl.n_sets) info->bytes_per_iter += 8 * tl.n_slots;
    }
    info->last_estep_ms = em->last_estep_ms;
    info->last_step_ms = em->last_step_ms;
    info->num_loci = em->L;
    info->num_haps = em->H;
    info->layout = em->layout;
    info->retained_build_bytes = em->tl.retired_bytes;
    info->num_locus_sets = em->layout == 1 ? em->tl.n_sets : 0;
    return GBRS_OK;
}

int gbrs_counts_create(uint64_t R, uint32_t L, uint32_t H, const uint32_t *const *indptr,
                       const uint32_t *const *indices, const double *count, int device, gbrs_counts_t **out) {
    if (!out) return fail(GBRS_ERR_INVALID, "out is NULL");
    *out = nullptr;
    if (H < 1 || H > 32 || L < 1 || R < 1 || R > 0xFFFFFFFFull || !indptr || !indices)
        return fail(GBRS_ERR_INVALID, "The shape must be a tuple of three positive integers (H <= 32, R < 2^32).");
    GBRS_TRY(select_device(device));
    gbrs_counts *c = new gbrs_counts();
    struct Guard { gbrs_counts *p; ~Guard() { if (p) gbrs_counts_destroy(p); } } guard{c};
    c->device = device;
    c->R = R; c->L = L; c->H = H;
    GBRS_HIP_CHECK(hipStreamCreateWithFlags(&c->s, hipStreamDefault));
    GBRS_TRY(upload_csc(R, L, H, indptr, indices, false, c->ent_row, c->col_ptr, c->n));
    if (count) {
        GBRS_TRY(c->d_count.alloc(R));
        GBRS_HIP_CHECK(hipMemcpy(c->d_count.p, count, R * sizeof(double), hipMemcpyHostToDevice));
    }
    GBRS_HIP_CHECK(hipDeviceSynchronize());
    GBRS_TRY(check_row_ids(c->n, c->ent_row.p, R, c->s));
    c->work = counts_work_new();
    guard.p = nullptr;
    *out = c;
    return GBRS_OK;
}

int gbrs_counts_get(gbrs_counts_t *c, const int32_t *locus_group, uint32_t num_out_loci, double *aln_counts,
                    double *allele_unique, double *locus_unique) {
    RoctxRange roctx_range("gbrs_counts_get");
    if (!c) return fail(GBRS_ERR_INVALID, "handle is NULL");
    const uint32_t L = c->L, H = c->H;
    const uint32_t Lout = locus_group ? num_out_loci : L;
    if (Lout < 1 || Lout >= (1u << 27)) return fail(GBRS_ERR_INVALID, "bad number of output loci");
    if (locus_group)
        for (uint32_t l = 0; l < L; ++l)
            if (locus_group[l] < -1 || (locus_group[l] >= 0 && (uint32_t)locus_group[l] >= Lout))
                return fail(GBRS_ERR_INVALID, "locus_group[%u] out of range", l);
    GBRS_TRY(select_device(c->device));
    DevBuf<double> d_aln, d_uniq, d_lu;
    DevBuf<int32_t> d_group;
    if (locus_group) {
        GBRS_TRY(d_group.alloc(L));
        GBRS_HIP_CHECK(hipMemcpy(d_group.p, locus_group, L * sizeof(int32_t), hipMemcpyHostToDevice));
    }
    GBRS_TRY(d_aln.alloc((size_t)H * Lout));
    GBRS_TRY(d_uniq.alloc((size_t)H * Lout));
    GBRS_TRY(d_lu.alloc(Lout));
    GBRS_TRY(alignment_counts_device(c->R, L, H, c->n, c->ent_row.p, c->col_ptr.p, c->d_count.p,
                                     locus_group ? d_group.p : nullptr, Lout, d_aln.p, d_uniq.p, d_lu.p, c->s, c->work));
    if (aln_counts) GBRS_HIP_CHECK(hipMemcpy(aln_counts, d_aln.p, d_aln.bytes(), hipMemcpyDeviceToHost));
    if (allele_unique) GBRS_HIP_CHECK(hipMemcpy(allele_unique, d_uniq.p, d_uniq.bytes(), hipMemcpyDeviceToHost));
    if (locus_unique) GBRS_HIP_CHECK(hipMemcpy(locus_unique, d_lu.p, d_lu.bytes(), hipMemcpyDeviceToHost));
    return GBRS_OK;
}

int gbrs_counts_destroy(gbrs_counts_t *c) {
    if (!c) return GBRS_OK;
    (void)hipSetDevice(c->device);
    if (c->s) {
        (void)hipStreamSynchronize(c->s);
        (void)hipStreamDestroy(c->s);
    }
    if (c->work) counts_work_free(c->work);
    delete c;
    return GBRS_OK;
}

int gbrs_alignment_counts(uint64_t R, uint32_t L, uint32_t H, const uint32_t *const *indptr,
                          const uint32_t *const *indices, const double *count, const int32_t *locus_group,
                          uint32_t num_out_loci, int device, double *aln_counts, double *allele_unique,
                          double *locus_unique) {
    gbrs_counts_t *c = nullptr;
    GBRS_TRY(gbrs_counts_create(R, L, H, indptr, indices, count, device, &c));
    const int st = gbrs_counts_get(c, locus_group, num_out_loci, aln_counts, allele_unique, locus_unique);
    (void)gbrs_counts_destroy(c);
    return st;
}

int gbrs_em_destroy(gbrs_em_t *em) {
    if (!em) return GBRS_OK;
    (void)hipSetDevice(em->device);
    if (em->stream) (void)hipStreamSynchronize(em->stream);
    if (em->ev0) (void)hipEventDestroy(em->ev0);
    if (em->ev1) (void)hipEventDestroy(em->ev1);
    if (em->ev2) (void)hipEventDestroy(em->ev2);
    for (auto e : em->ev_pool) (void)hipEventDestroy(e);
    if (em->pair_ev_mstep) (void)hipEventDestroy(em->pair_ev_mstep);
    if (em->pair_ev_err) (void)hipEventDestroy(em->pair_ev_err);
    if (em->stream && em->own_stream) (void)hipStreamDestroy(em->stream);
    delete em;
    return GBRS_OK;
}

}  // extern "C"

// gbrs_warm_up (common.hip): loads this file's code object
namespace gbrs {
__global__ void warm_em_kernel() {}
void warm_em(hipStream_t st) { hipLaunchKernelGGL(warm_em_kernel, dim3(1), dim3(64), 0, st); }
}  // namespace gbrs
