// `gbrs reconstruct` HMM on MI355X (gfx950): emission model, forward + Viterbi, backward +
// posterior, backtrace.  Reference: gbrs/gbrs_utils.py:63-98 and :463-599.
//
// Parallel decomposition: one workgroup per (sample, chromosome); the recursion over genes is
// sequential by nature, so each step is made short instead: the S x S transition block of the
// step is spread over S*LPS threads (LPS lanes per state, shuffle-reduced), the next step's
// block is prefetched into registers while the current one is being reduced, and there is one
// workgroup barrier per gene.  No dense contraction is reshaped for MFMA: the path is exp/log
// and latency bound.
#include "common.h"

#include <algorithm>
#include <cfloat>
#include <climits>
#include <cmath>
#include <cstdlib>
#include <functional>

namespace gbrs {

constexpr double TINY = 4.9406564584124654e-324;   // np.nextafter(0, 1)
constexpr int MAX_H = 16;   // S <= 136 (CC-style 16 founders); DO is H = 8, S = 36

// ------------------------------------------------------------------------------------------
// Emission.  Operation order follows get_genotype_probability:
// builtin (sequential) sums, v / norm divisions, exp(d / (-2 sigma^2)), p / sum(p), log(p + tiny).
// Compiled with -ffp-contract=off so a*a + s is never fused.
// ------------------------------------------------------------------------------------------

__device__ __forceinline__ double seq_norm(const double *v, int H, int stride) {
    double s = 0.0;
    for (int x = 0; x < H; ++x) s += v[x * stride] * v[x * stride];
    return sqrt(s);
}

// EM_GENES consecutive genes of one sample per 64-thread workgroup, EM_LANES lanes per gene.  The
// genes' specificity matrices, expression vectors and results pass through LDS so that every global
// access is a coalesced stream (a thread reading its own 512-byte matrix straight from HBM touches
// 64 different lines per instruction).  The lanes of a gene split the unit-row pass by rows and the
// diplotype pass by states; every per-gene reduction the reference does with builtin sum() is
// still evaluated sequentially in its order (each lane repeats it), so results do not depend on
// the split.  An 8-gene block (8 lanes per gene, measured best of 2-16) needs 7 KB of LDS at H = 8.
#ifndef HMM_EM_LANES
#define HMM_EM_LANES 8
#endif
constexpr int EM_LANES = HMM_EM_LANES;
constexpr int EM_GENES = 64 / EM_LANES;
#ifndef HMM_EM_BATCH_MIN
#define HMM_EM_BATCH_MIN 4      // samples from which emission_batch_kernel replaces emission_kernel
#endif
#ifndef HMM_EM_BATCH_SPB
#define HMM_EM_BATCH_SPB 16     // samples walked by one workgroup of emission_batch_kernel
#endif
constexpr int EM_BATCH_MIN = HMM_EM_BATCH_MIN, EM_BATCH_SPB = HMM_EM_BATCH_SPB;

// one-wavefront workgroups: LDS instructions of a wavefront execute in order, so a compiler fence replaces the barrier
// (__syncthreads would also wait for the global loads and stores in flight)
__device__ __forceinline__ void em_wave_fence() {
    asm volatile("" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    asm volatile("" ::: "memory");
}

__global__ void __launch_bounds__(64)
emission_kernel(int H, int S, int64_t n_genes, int n_samples, const double *__restrict__ expr,
                const double *__restrict__ avecs, const uint8_t *__restrict__ has_avec,
                const double *__restrict__ init_vec, double expr_threshold, double sigma,
                double *__restrict__ eprob, double *__restrict__ peprob) {
    extern __shared__ double lds[];
    const int HH = H * H, av_stride = HH + 1, ex_stride = H + 1, out_stride = S + 1;
    double *l_av = lds, *l_ex = l_av + EM_GENES * av_stride, *l_out = l_ex + EM_GENES * ex_stride,
           *l_pe = l_out + EM_GENES * out_stride;
    const int64_t blocks_per_sample = (n_genes + EM_GENES - 1) / EM_GENES;
    const int sample = (int)(blockIdx.x / blocks_per_sample);
    const int64_t g0 = (blockIdx.x % blocks_per_sample) * EM_GENES;
    const int ng = (int)min((int64_t)EM_GENES, n_genes - g0);
    const int tid = threadIdx.x;
    for (int x = tid; x < ng * HH; x += 64) l_av[(x / HH) * av_stride + x % HH] = avecs[g0 * HH + x];
    const double *ex_src = expr + ((int64_t)sample * n_genes + g0) * H;
    for (int x = tid; x < ng * H; x += 64) l_ex[(x / H) * ex_stride + x % H] = ex_src[x];
    __syncthreads();

    const int lg = tid / EM_LANES, q = tid % EM_LANES;
    const bool in_range = lg < ng;
    const double *e = l_ex + lg * ex_stride;
    double *out = l_out + lg * out_stride, *pout = l_pe + lg * out_stride;
    double *U = l_av + lg * av_stride;
    // states of this lane: a contiguous chunk [s_lo, s_hi) of the upper-triangular order
    const int spl = (S + EM_LANES - 1) / EM_LANES;
    const int s_lo = min(S, q * spl), s_hi = min(S, s_lo + spl);
    double esum = 0.0;
    if (in_range)
        for (int x = 0; x < H; ++x) esum += e[x];
    const bool live = in_range && !(esum < expr_threshold);
    bool naive = false;
    if (live) {
        naive = !has_avec[g0 + lg];
        // unit_vector() of the specificity rows i = q, q + EM_LANES, ... once, in place (the same
        // quotients the reference recomputes for every diplotype)
        for (int i = q; i < H; i += EM_LANES) {
            double sm = 0.0, qq = 0.0;
            for (int x = 0; x < H; ++x) {
                const double a = naive ? (x == i ? 1.0 : 0.0001) : U[i * H + x];
                sm += a;
                qq += a * a;
            }
            const double rn = sm > 1e-6 ? sqrt(qq) : 0.0;
            for (int x = 0; x < H; ++x) {
                const double a = naive ? (x == i ? 1.0 : 0.0001) : U[i * H + x];
                U[i * H + x] = rn != 0.0 ? a / rn : a;
            }
        }
    }
    __syncthreads();
    if (in_range && !live) {
        for (int s = s_lo; s < s_hi; ++s) out[s] = init_vec[s];
    } else if (live) {
        const double sg = naive ? 0.450 : sigma;
        const double denom = -2 * sg * sg;
        double u[MAX_H];                       // profile unit vector
        {
            const bool norm = esum > 1e-6;
            const double nrm = norm ? seq_norm(e, H, 1) : 1.0;
            for (int x = 0; x < H; ++x) u[x] = norm ? e[x] / nrm : e[x];
        }
        int i = 0, rem = s_lo;                 // (i, j) of state s_lo
        while (i < H && rem >= H - i) { rem -= H - i; ++i; }
        int j = i + rem;
        for (int s = s_lo; s < s_hi; ++s) {
            double d = 0.0;
            if (j == i) {
                for (int x = 0; x < H; ++x) {
                    const double t = u[x] - U[i * H + x];
                    d += t * t;
                }
            } else {
                double gs = 0.0, gq = 0.0;
                for (int x = 0; x < H; ++x) {
                    const double w = U[i * H + x] + U[j * H + x];
                    gs += w;
                    gq += w * w;
                }
                const bool norm = gs > 1e-6;
                const double gn = norm ? sqrt(gq) : 1.0;
                for (int x = 0; x < H; ++x) {
                    const double w = U[i * H + x] + U[j * H + x];
                    const double t = u[x] - (norm ? w / gn : w);
                    d += t * t;
                }
            }
            out[s] = exp(d / denom);
            if (++j == H) { ++i; j = i; }
        }
    }
    __syncthreads();
    if (in_range && !live) {
        for (int s = s_lo; s < s_hi; ++s) pout[s] = exp(init_vec[s]);
    } else if (live) {
        double psum = 0.0;
        for (int s = 0; s < S; ++s) psum += out[s];
        for (int s = s_lo; s < s_hi; ++s) {
            const double pr = out[s] / psum + TINY;
            out[s] = log(pr);
            pout[s] = pr;                      // pe = exp(e) without the round trip through the logarithm
        }
    }
    __syncthreads();
    // the log emissions, and pe = exp(e) for the probability-domain sweeps
    double *dst = eprob + ((int64_t)sample * n_genes + g0) * S;
    double *pdst = peprob + ((int64_t)sample * n_genes + g0) * S;
    for (int x = tid; x < ng * S; x += 64) {
        dst[x] = l_out[(x / S) * out_stride + x % S];
        pdst[x] = l_pe[(x / S) * out_stride + x % S];
    }
}

// The same emissions for many samples at once (H == EM_LANES founders).  Everything that depends on the
// gene only - the unit specificity rows and the S unit diplotype vectors, i.e. all but 9 of the ~260
// divisions and all the square roots but one - is computed once per workgroup and kept in registers
// (each of a gene's 8 lanes owns ceil(S/8) states), then the workgroup walks its samples: expression
// row in, unit profile (lane q divides element q), distances, exp, the sequential sum, log, rows out.
// Operation for operation the arithmetic of emission_kernel, so both produce the same bits.
template <int HT>
__global__ void __launch_bounds__(64)
emission_batch_kernel(int64_t n_genes, int n_samples, int samples_per_block, const double *__restrict__ expr,
                      const double *__restrict__ avecs, const uint8_t *__restrict__ has_avec,
                      const double *__restrict__ init_vec, double expr_threshold, double sigma,
                      double *__restrict__ eprob, double *__restrict__ peprob,
                      int64_t gene_begin, int64_t gene_end /* the launch covers genes [gene_begin, gene_end) of the n_genes */) {
    constexpr int H = HT, S = H * (H + 1) / 2, HH = H * H;
    static_assert(HT == EM_LANES, "lane q of a gene owns specificity row q and profile element q");
    constexpr int SPL = (S + EM_LANES - 1) / EM_LANES;
    constexpr int av_stride = HH + 1, ex_stride = H + 1, out_stride = S + 1;
    __shared__ double l_av[EM_GENES * av_stride], l_ex[EM_GENES * ex_stride], l_u[EM_GENES * ex_stride],
        l_out[EM_GENES * out_stride], l_pe[EM_GENES * out_stride], l_init[2 * S];
    const int64_t g0 = gene_begin + (int64_t)blockIdx.x * EM_GENES;
    const int ng = (int)min((int64_t)EM_GENES, gene_end - g0);
    const int s_begin = blockIdx.y * samples_per_block, s_end = min(n_samples, s_begin + samples_per_block);
    const int tid = threadIdx.x;
    for (int x = tid; x < ng * HH; x += 64) l_av[(x / HH) * av_stride + x % HH] = avecs[g0 * HH + x];
    // one expression element per lane and sample, fetched one sample ahead
    const bool has_e = tid < ng * H;
    double e_next = has_e && s_begin < s_end ? expr[((int64_t)s_begin * n_genes + g0) * H + tid] : 0.0;
    em_wave_fence();
    const int lg = tid / EM_LANES, q = tid % EM_LANES;
    const bool in_range = lg < ng;
    double *U = l_av + lg * av_stride;
    const bool naive = in_range && !has_avec[g0 + lg];
    if (in_range) {                            // unit_vector() of specificity row q, in place
        double sm = 0.0, qq = 0.0;
        for (int x = 0; x < H; ++x) {
            const double a = naive ? (x == q ? 1.0 : 0.0001) : U[q * H + x];
            sm += a;
            qq += a * a;
        }
        const double rn = sm > 1e-6 ? sqrt(qq) : 0.0;
        for (int x = 0; x < H; ++x) {
            const double a = naive ? (x == q ? 1.0 : 0.0001) : U[q * H + x];
            U[q * H + x] = rn != 0.0 ? a / rn : a;
        }
    }
    em_wave_fence();
    const int s_lo = min(S, q * SPL), s_hi = min(S, s_lo + SPL);
    double G[SPL][H];                          // unit diplotype vectors of this lane's states
    {
        int i = 0, rem = s_lo;
        while (i < H && rem >= H - i) { rem -= H - i; ++i; }
        int j = i + rem;
#pragma unroll
        for (int t = 0; t < SPL; ++t) {
            const bool on = in_range && s_lo + t < s_hi;
            const int ii = on ? i : 0, jj = on ? j : 0;
            if (jj == ii) {
#pragma unroll
                for (int x = 0; x < H; ++x) G[t][x] = U[ii * H + x];
            } else {
                double gs = 0.0, gq = 0.0;
#pragma unroll
                for (int x = 0; x < H; ++x) {
                    const double w = U[ii * H + x] + U[jj * H + x];
                    gs += w;
                    gq += w * w;
                }
                const bool norm = gs > 1e-6;
                const double gn = norm ? sqrt(gq) : 1.0;
#pragma unroll
                for (int x = 0; x < H; ++x) {
                    const double w = U[ii * H + x] + U[jj * H + x];
                    G[t][x] = norm ? w / gn : w;
                }
            }
            if (on && ++j == H) { ++i; j = i; }
        }
    }
    const double sg = naive ? 0.450 : sigma;
    const double denom = -2 * sg * sg;
    const double *e = l_ex + lg * ex_stride;
    double *out = l_out + lg * out_stride, *pout = l_pe + lg * out_stride;
    if (tid < S) {                             // log and linear start values: e and pe of a gene below the threshold
        l_init[tid] = init_vec[tid];
        l_init[S + tid] = exp(init_vec[tid]);
    }
    for (int sample = s_begin; sample < s_end; ++sample) {
        if (has_e) l_ex[(tid / H) * ex_stride + tid % H] = e_next;
        if (has_e && sample + 1 < s_end) e_next = expr[((int64_t)(sample + 1) * n_genes + g0) * H + tid];
        em_wave_fence();
        double esum = 0.0;
        if (in_range)
            for (int x = 0; x < H; ++x) esum += e[x];
        const bool live = in_range && !(esum < expr_threshold);
        if (live) {
            const bool norm = esum > 1e-6;
            const double nrm = norm ? seq_norm(e, H, 1) : 1.0;
            l_u[lg * ex_stride + q] = norm ? e[q] / nrm : e[q];
        }
        em_wave_fence();
        if (in_range && !live) {
#pragma unroll
            for (int t = 0; t < SPL; ++t)
                if (s_lo + t < s_hi) {
                    out[s_lo + t] = l_init[s_lo + t];
                    pout[s_lo + t] = l_init[S + s_lo + t];
                }
        } else if (live) {
            double u[H];
#pragma unroll
            for (int x = 0; x < H; ++x) u[x] = l_u[lg * ex_stride + x];
#pragma unroll
            for (int t = 0; t < SPL; ++t) {
                if (s_lo + t >= s_hi) break;
                double d = 0.0;
#pragma unroll
                for (int x = 0; x < H; ++x) {
                    const double dd = u[x] - G[t][x];
                    d += dd * dd;
                }
                out[s_lo + t] = exp(d / denom);
            }
        }
        em_wave_fence();
        if (live) {
            double psum = 0.0;
            for (int s = 0; s < S; ++s) psum += out[s];
            for (int s = s_lo; s < s_hi; ++s) {
                const double pr = out[s] / psum + TINY;
                out[s] = log(pr);
                pout[s] = pr;                  // pe = exp(e) without the round trip through the logarithm
            }
        }
        em_wave_fence();
        double *dst = eprob + ((int64_t)sample * n_genes + g0) * S;
        double *pdst = peprob + ((int64_t)sample * n_genes + g0) * S;
        for (int x = tid; x < ng * S; x += 64) {
            dst[x] = l_out[(x / S) * out_stride + x % S];
            pdst[x] = l_pe[(x / S) * out_stride + x % S];
        }
    }
}

// ------------------------------------------------------------------------------------------
// Forward + Viterbi.  Block = S*LPS threads rounded up to a wave; thread (j, q) owns row j of the
// step's transition block T[i-1][j][:] at columns k = q + LPS*m.
// ------------------------------------------------------------------------------------------

// The chain kernels with two register sets are the blocked scan's (1,000 one-wavefront workgroups per kernel, three such
// kernels at a time on 1,024 SIMDs): they are built for HMM_BLK_WAVES wavefronts per SIMD, so that the three sides of a
// pass can share the SIMDs instead of queueing for whole register files; with three sets (60 wavefronts per pass) a
// wavefront may take the whole file.
#ifndef HMM_BLK_WAVES
#define HMM_BLK_WAVES 2
#endif
#define HMM_CHAIN_WAVES(NSET) ((NSET) <= 2 ? HMM_BLK_WAVES : 1)

// Where the per-(sample, gene) rows of eprob / peprob / xsum / bhat / delta / invz / bscale sit: row = sample * sample_stride +
// gene * gene_stride.  {genes_per_sample, 1} is [sample][gene] (a sample's genes contiguous: what gbrs_hmm_get copies out);
// {1, n_samples} is [gene][sample] (the 16 samples of a wavefront's step contiguous).
struct RowMap {
    int64_t sample_stride, gene_stride;
    // XCD-aware 1-D grids of the batch chain kernels (GBRS_TUNING_HMM_XCD): workgroup L runs on XCD L % 8.  The chromosomes are
    // dealt to 8 / xcd_span sets of xcd_span XCDs each (chromosome order index o -> set o % (8 / span)), and a chromosome's
    // xcd_groups sample groups round-robin over the XCDs of its set: its transition blocks are fetched into xcd_span L2s
    // instead of eight.  0 groups: plain 2-D grid (blockIdx.x = group, blockIdx.y = order index).
    int32_t xcd_groups = 0, n_order = 0, xcd_span = 1;
    __device__ __forceinline__ bool place(int &bx, int &by) const {
        if (xcd_groups > 0) {
            const int xcd = bx & 7, slot = bx >> 3;
            const int sets = 8 / xcd_span, set = xcd / xcd_span, lane = xcd % xcd_span;
            const int per = (xcd_groups + xcd_span - 1) / xcd_span;          // groups of a chromosome on one XCD
            by = set + sets * (slot / per);
            bx = lane + xcd_span * (slot % per);
            return by < n_order && bx < xcd_groups;
        }
        return true;
    }
};

struct ChromDesc {
    int64_t gene_off;     // offset of the chromosome's first gene in the per-sample gene axis
    int64_t trans_off;    // offset (in S*S blocks) of tprob[c][0] in the transition buffer
    int64_t bp_off;       // offset (in S entries) of the chromosome's backpointer rows
    int64_t chunk_off;    // offset (in BT_B-row chunks) of the chromosome's backtrace chunk maps
    int32_t n_genes, n_trans;
    // Blocked scan (hmm_blocked.inc): a descriptor may stand for a run of genes of a chromosome whose recursion starts
    // from an injected boundary vector instead of the chromosome's own start / end.
    int32_t inject;       // -1: the natural start; else the slot of the block's boundary vector
    int32_t real_chrom;   // the chromosome whose last gene this descriptor ends on (forward / delta), else -1
};

// ---- cross-lane helpers on DPP (no LDS crossbar on the recursion's critical path) ------------
template <int CTRL>
__device__ __forceinline__ double dpp_f64(double x) {
    int lo = __double2loint(x), hi = __double2hiint(x);
    lo = __builtin_amdgcn_update_dpp(lo, lo, CTRL, 0xf, 0xf, false);
    hi = __builtin_amdgcn_update_dpp(hi, hi, CTRL, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}
#if defined(HMM_ABLATE_SAMEBLOCK)
#define HMM_BLK(i) 0
#else
#define HMM_BLK(i) (i)
#endif
#if defined(HMM_PRED_LOADS)
#define HMM_LOAD_PRED if (act)
#else
#define HMM_LOAD_PRED
#endif
constexpr int DPP_QUAD_XOR1 = 0xB1;      // quad_perm:[1,0,3,2]
constexpr int DPP_QUAD_XOR2 = 0x4E;      // quad_perm:[2,3,0,1]
constexpr int DPP_ROW_HALF_MIRROR = 0x141;
constexpr int DPP_ROW_MIRROR = 0x140;

// sum over the 4 lanes of a quad (LPS = 4 lanes per state); result in every lane of the quad
__device__ __forceinline__ double quad_sum(double v) {
    v += dpp_f64<DPP_QUAD_XOR1>(v);
    v += dpp_f64<DPP_QUAD_XOR2>(v);
    return v;
}

// (max value, its lowest index) over the quad; ties keep the lower index (np.argmax)
__device__ __forceinline__ void quad_argmax(double &v, int &k) {
    {
        const double ov = dpp_f64<DPP_QUAD_XOR1>(v);
        const int ok = __builtin_amdgcn_update_dpp(k, k, DPP_QUAD_XOR1, 0xf, 0xf, false);
        if (ov > v || (ov == v && ok < k)) { v = ov; k = ok; }
    }
    {
        const double ov = dpp_f64<DPP_QUAD_XOR2>(v);
        const int ok = __builtin_amdgcn_update_dpp(k, k, DPP_QUAD_XOR2, 0xf, 0xf, false);
        if (ov > v || (ov == v && ok < k)) { v = ov; k = ok; }
    }
}

// sum over all 64 lanes, same value in every lane: quad, half-row and row steps on DPP, the four
// row totals through v_readlane
__device__ __forceinline__ double wave_sum_dpp(double v) {
    v += dpp_f64<DPP_QUAD_XOR1>(v);
    v += dpp_f64<DPP_QUAD_XOR2>(v);
    v += dpp_f64<DPP_ROW_HALF_MIRROR>(v);
    v += dpp_f64<DPP_ROW_MIRROR>(v);
    const int lo = __double2loint(v), hi = __double2hiint(v);
    double t = __hiloint2double(__builtin_amdgcn_readlane(hi, 0), __builtin_amdgcn_readlane(lo, 0));
    t += __hiloint2double(__builtin_amdgcn_readlane(hi, 16), __builtin_amdgcn_readlane(lo, 16));
    t += __hiloint2double(__builtin_amdgcn_readlane(hi, 32), __builtin_amdgcn_readlane(lo, 32));
    t += __hiloint2double(__builtin_amdgcn_readlane(hi, 48), __builtin_amdgcn_readlane(lo, 48));
    return t;
}

// Workgroup barrier for the recursion loops: waits for this wave's LDS traffic only.  A plain
// __syncthreads() also drains vmcnt, i.e. it would wait every step for the step's global stores
// and for the transition block prefetched for the NEXT step (1-2 us each) - the whole point of
// the prefetch is that those stay in flight across the barrier.
__device__ __forceinline__ void lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

// sum of the S values of an LDS vector, computed redundantly by every wave (no extra barrier)
__device__ __forceinline__ double wave_sum_lds(const double *x, int S) {
    double s = 0.0;
    for (int j = threadIdx.x & 63; j < S; j += 64) s += x[j];
    return wave_sum_dpp(s);
}

// P = exp(T) (forward) and its per-block transpose (backward), made once per transition table:
// the tables are sample independent, so every sample and every later call reuses them.
__global__ void exp_blocks_kernel(int S, int64_t n_blocks, const double *__restrict__ t,
                                  double *__restrict__ p, double *__restrict__ pt) {
    const int64_t total = n_blocks * S * S;
    for (int64_t x = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; x < total;
         x += (int64_t)gridDim.x * blockDim.x) {
        const int64_t b = x / ((int64_t)S * S);
        const int r = (int)(x - b * S * S);
        const int jj = r / S, kk = r % S;
        const double v = exp(t[x]);
        p[x] = v;
        pt[b * S * S + (int64_t)kk * S + jj] = v;
    }
}

// The chain kernels read their tables in lane order: with LPS lanes per state (1 for the single-wave
// kernels, 4 for the quad chains) and KPL = S / LPS columns per lane, double2 number m2 of lane
// t = LPS*j + q (state j, row part q) sits at double offset (m2 * LPS*S + t) * 2 of the block, so one
// wave-instruction reads a contiguous run (1 KiB for the quad chains) instead of one 16-byte piece
// from each of up to 64 rows (at S = 136 a step streams 148 KB; row-strided lanes ran it at a
// seventh of this).  p_q = exp(T), pt_q = exp(T) transposed, t_q = T, all in that order.
__global__ void lane_blocks_kernel(int S, int LPS, int KPL, int64_t n_blocks, const double *__restrict__ t,
                                   double *__restrict__ p_q, double *__restrict__ pt_q,
                                   double *__restrict__ t_q) {
    const int64_t bs = (int64_t)S * S, total = n_blocks * bs;
    const int lanes2 = 2 * LPS * S;
    for (int64_t x = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; x < total;
         x += (int64_t)gridDim.x * blockDim.x) {
        const int64_t b = x / bs;
        const int within = (int)(x - b * bs);
        const int m2 = within / lanes2, rem = within % lanes2;
        const int tl = rem / 2, e = rem % 2;
        const int jj = tl / LPS, q = tl % LPS;
        const int kk = q * KPL + 2 * m2 + e;
        const double v = t[b * bs + (int64_t)jj * S + kk];
        t_q[x] = v;
        p_q[x] = exp(v);
        pt_q[x] = exp(t[b * bs + (int64_t)kk * S + jj]);
    }
}

__device__ __forceinline__ double fast_recip_pos(double d) {
    double r = __builtin_amdgcn_rcp(d);
    double e = fma(-d, r, 1.0);
    r = fma(r, e, r);
    e = fma(-d, r, 1.0);
    return fma(r, e, r);
}

// pe = exp(eprob): every gene at once, before the sequential sweeps
__global__ void exp_emission_kernel(int64_t n, const double *__restrict__ e, double *__restrict__ pe) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) pe[i] = exp(e[i]);
}

// Forward + Viterbi (gbrs_utils.py:500-526, :567-579), one workgroup per (chromosome, sample,
// role): role 0 runs the alpha recursion, role 1 the delta/backpointer recursion; the two chains
// only share the transition tables, so they run side by side on different CUs.
//
// alpha: the reference's step  alpha_i[j] = log(sum_k exp(alpha_{i-1}[k] + T[j,k]) + tiny) + e_i[j],
// alpha_i -= log(sum_j exp(alpha_i[j]))  is carried in the probability domain so that no exp/log
// and no global load sits on the sequential critical path:
//     x_j = (sum_k y_{i-1}[k] * P[j,k]) / Z_{i-1} + tiny        P = exp(T), precomputed
//     y_j = x_j * pe_i[j]                Z_i = sum_j y_j         pe = exp(e), precomputed
// The kernel stores x and 1/Z; posterior_kernel recomputes alpha-hat = x*pe/Z from them and
// hmm_outputs_kernel makes the log-domain alpha and scaler of the reference (same quantities up to
// rounding), both in parallel over genes.
// delta stays in the log domain on T itself: additions and max only, i.e. exact.
template <int KMAX, int MAXT, bool EXACT>
__global__ void __launch_bounds__(MAXT)
forward_viterbi_kernel(int S, int64_t genes_per_sample, int64_t bp_per_sample,
                       const ChromDesc *__restrict__ chroms, const double *__restrict__ tprob,
                       const double *__restrict__ pprob, const double *__restrict__ eprob,
                       const double *__restrict__ peprob, const double *__restrict__ init_vec,
                       double *__restrict__ xsum, double *__restrict__ invz,
                       double *__restrict__ delta, uint16_t *__restrict__ bp,
                       int32_t *__restrict__ last_state) {
    constexpr int LPS = 4;
    extern __shared__ double lds[];           // v[2][S]
    const ChromDesc cd = chroms[blockIdx.x];
    const int sample = blockIdx.y;
    const int role = blockIdx.z;
    const int n = cd.n_genes;
    if (n <= 0) return;
    const int j = threadIdx.x / LPS, q = threadIdx.x % LPS;
    const bool active = j < S;
    const bool owner = active && q == 0;
    const int KPT = EXACT ? KMAX : (S + LPS - 1) / LPS;   // columns per lane (<= KMAX)
    // EXACT: S == LPS * KMAX, so every (lane, m) is a real column and no access needs a predicate;
    // the padding threads (j >= S) then simply shadow the last state and never store
    const int jr = EXACT ? min(j, S - 1) : j;
    const int64_t g0 = (int64_t)sample * genes_per_sample + cd.gene_off;
    const int n_steps = min(n, cd.n_trans + 1);   // gene i needs T[i-1]
    double *buf = lds;
    int cur = 0;

#if defined(HMM_ONLY_DELTA)
    if (role == 0) return;
#elif defined(HMM_ONLY_ALPHA)
    if (role == 1) return;
#endif
    if (role == 0) {
        const double *E = eprob + g0 * S, *PE = peprob + g0 * S;
        double *XS = xsum + g0 * S, *IZ = invz + g0;
        const double *P = pprob + cd.trans_off * (int64_t)S * S;
        double y_own = 0.0;
        if (owner) {
            const double a0 = init_vec[j] + E[j];
            y_own = exp(a0);
            buf[j] = y_own;
            XS[j] = exp(init_vec[j]);             // so that log(x) + e reproduces init + e
        }
        // two register sets for the transition block: the step on one set refills the other for
        // the step after it, so no block is ever copied between registers
        double pa[KMAX], pb[KMAX];
#pragma unroll
        for (int m = 0; m < KMAX; ++m) {
            const int k = q * KPT + m;          // contiguous chunk per lane: wide loads
            pa[m] = ((EXACT || (active && m < KPT && k < S)) && cd.n_trans > 0) ? P[(int64_t)jr * S + k] : 0.0;
            pb[m] = 0.0;
        }
        double pe_next = (owner && n > 1) ? PE[(int64_t)S + j] : 0.0;
        __syncthreads();
        double inv_z = fast_recip_pos(wave_sum_lds(buf, S));
        if (owner) {
            if (j == 0) IZ[0] = inv_z;
        }
        auto step = [&](int i, const double (&pc)[KMAX], double (&pn)[KMAX]) {
            const double pe = pe_next;
            if (owner && i + 1 < n) pe_next = PE[(int64_t)(i + 1) * S + j];
            if (i < cd.n_trans) {                 // prefetch P[i] for the next step
                const double *Pn = P + (int64_t)HMM_BLK(i) * S * S;
#pragma unroll
                for (int m = 0; m < KMAX; ++m) {
                    const int k = q * KPT + m;
                    if (EXACT || (active && m < KPT && k < S)) pn[m] = Pn[(int64_t)jr * S + k];
                }
            }
            const double *y_prev = buf + cur * S;
            double s0 = 0.0, s1 = 0.0, s2 = 0.0;
#pragma unroll
            for (int m = 0; m < KMAX; ++m) {
                const int k = q * KPT + m;
                if (EXACT || (active && m < KPT && k < S)) {
                    const double yp = y_prev[k];
                    if (m % 3 == 0) s0 = fma(yp, pc[m], s0);
                    else if (m % 3 == 1) s1 = fma(yp, pc[m], s1);
                    else s2 = fma(yp, pc[m], s2);
                }
            }
            const double x = quad_sum((s0 + s1) + s2) * inv_z + TINY;
            const int nxt = cur ^ 1;
            if (owner) {
                y_own = x * pe;
                buf[nxt * S + j] = y_own;
                XS[(int64_t)i * S + j] = x;
            }
            lds_barrier();
            cur = nxt;
            inv_z = fast_recip_pos(wave_sum_lds(buf + cur * S, S));
            if (owner) {
                if (j == 0) IZ[i] = inv_z;
            }
        };
        int i = 1;
        for (; i + 1 < n_steps; i += 2) {
            step(i, pa, pb);
            step(i + 1, pb, pa);
        }
        if (i < n_steps) step(i, pa, pb);
        return;
    }

    // role 1: Viterbi
    const double *E = eprob + g0 * S;
    double *DL = delta + g0 * S;
    uint16_t *BP = bp + ((int64_t)sample * bp_per_sample + cd.bp_off) * S;
    const double *T = tprob + cd.trans_off * (int64_t)S * S;
    const bool phantom = cd.n_trans >= n;         // one more max-step with T[n-1] (backtrace quirk)
    double d_own = 0.0;
    if (owner) {
        d_own = init_vec[j] + E[j];
        buf[j] = d_own;
        DL[j] = d_own;
    }
    double ta[KMAX], tb[KMAX];
#pragma unroll
    for (int m = 0; m < KMAX; ++m) {
        const int k = q * KPT + m;          // contiguous chunk per lane: wide loads
        ta[m] = ((EXACT || (active && m < KPT && k < S)) && cd.n_trans > 0) ? T[(int64_t)jr * S + k] : 0.0;
        tb[m] = 0.0;
    }
    double e_next = (owner && n > 1) ? E[(int64_t)S + j] : 0.0;
    __syncthreads();
    // step i: delta_i from delta_{i-1} and T[i-1] (in tc), backpointers of transition i-1; `real`
    // false = the phantom max-step with T[n-1] that only yields backpointers
    auto step = [&](int i, bool real, const double (&tc)[KMAX], double (&tn)[KMAX]) {
        const double e = e_next;
        if (owner && i + 1 < n) e_next = E[(int64_t)(i + 1) * S + j];
        if (i < cd.n_trans) {
            const double *Tn = T + (int64_t)HMM_BLK(i) * S * S;
#pragma unroll
            for (int m = 0; m < KMAX; ++m) {
                const int k = q * KPT + m;
                if (EXACT || (active && m < KPT && k < S)) tn[m] = Tn[(int64_t)jr * S + k];
            }
        }
        const double *d_prev = buf + cur * S;
        double best = -DBL_MAX;
        int best_k = 0x7fffffff;
#pragma unroll
        for (int m = 0; m < KMAX; ++m) {
            const int k = q * KPT + m;
            if (EXACT || (active && m < KPT && k < S)) {
                const double dv = d_prev[k] + tc[m];
                if (dv > best) { best = dv; best_k = k; }   // ascending k: first max kept
            }
        }
        quad_argmax(best, best_k);
        const int nxt = cur ^ 1;
        if (owner) {
            BP[(int64_t)(i - 1) * S + j] = (uint16_t)best_k;
            if (real) {
                d_own = best + e;
                buf[nxt * S + j] = d_own;
                DL[(int64_t)i * S + j] = d_own;
            }
        }
        if (real) {
            lds_barrier();
            cur = nxt;
        }
    };
    int i = 1;
    for (; i + 1 < n_steps; i += 2) {
        step(i, true, ta, tb);
        step(i + 1, true, tb, ta);
    }
    bool on_a = true;                        // which set holds the block for step i
    if (i < n_steps) {
        step(i, true, ta, tb);
        ++i;
        on_a = false;
    }
    if (phantom && i == n) {                 // n_steps == n here: the last real step prefetched T[n-1]
        if (on_a) step(n, false, ta, tb); else step(n, false, tb, ta);
    }
    // sid = argmax delta[:, n-1] (first max)
    __syncthreads();
    if (threadIdx.x == 0) {
        const double *dl = DL + (int64_t)(n - 1) * S;
        double b = dl[0];
        int bk = 0;
        for (int s = 1; s < S; ++s)
            if (dl[s] > b) { b = dl[s]; bk = s; }
        last_state[(int64_t)sample * gridDim.x + blockIdx.x] = bk;
    }
}

// Backward (gbrs_utils.py:530-550) in the probability domain:
//     bhat_i[j] = (sum_k P_i[k,j] * bhat_{i+1}[k] * pe_{i+1}[k]) / Z_i          bhat_{n-1} = 1/Z_{n-1}
// pprob_t holds the transposed blocks Pt[i][j][k] = exp(T[i][k][j]) so that a thread reads a row.
// beta = log(bhat) is produced by hmm_outputs_kernel, the posterior by posterior_kernel.
template <int KMAX, int MAXT, bool EXACT>
__global__ void __launch_bounds__(MAXT)
backward_kernel(int S, int64_t genes_per_sample, const ChromDesc *__restrict__ chroms,
                const double *__restrict__ pprob_t, const double *__restrict__ peprob,
                const double *__restrict__ invz, double *__restrict__ bhat) {
    constexpr int LPS = 4;
    extern __shared__ double lds[];           // w[2][S]
    const ChromDesc cd = chroms[blockIdx.x];
    const int sample = blockIdx.y;
    const int n = cd.n_genes;
    if (n <= 0) return;
    const int j = threadIdx.x / LPS, q = threadIdx.x % LPS;
    const bool active = j < S;
    const bool owner = active && q == 0;
    const int KPT = EXACT ? KMAX : (S + LPS - 1) / LPS;   // columns per lane (<= KMAX)
    // EXACT: S == LPS * KMAX, so every (lane, m) is a real column and no access needs a predicate;
    // the padding threads (j >= S) then simply shadow the last state and never store
    const int jr = EXACT ? min(j, S - 1) : j;
    const int64_t g0 = (int64_t)sample * genes_per_sample + cd.gene_off;
    const double *PE = peprob + g0 * S, *IZ = invz + g0;
    double *BH = bhat + g0 * S;
    const double *Pt = pprob_t + cd.trans_off * (int64_t)S * S;
    double *w_buf = lds;

    // The reference indexes tprob[c][i] for i = n-2 .. 0, so it needs n_trans >= n-1 (host check).
    int cur = 0;
    if (owner) {
        const int64_t o = (int64_t)(n - 1) * S + j;
        const double bh = IZ[n - 1];                       // exp(scaler[n-1])
        BH[o] = bh;
        w_buf[j] = bh * PE[o];
    }
    double ta[KMAX], tb[KMAX];
#pragma unroll
    for (int m = 0; m < KMAX; ++m) {
        const int k = q * KPT + m;          // contiguous chunk per lane: wide loads
        ta[m] = ((EXACT || (active && m < KPT && k < S)) && n >= 2) ? Pt[((int64_t)(n - 2) * S + jr) * S + k] : 0.0;
        tb[m] = 0.0;
    }
    double pe_nx = (owner && n >= 2) ? PE[(int64_t)(n - 2) * S + j] : 0.0;
    double iz_nx = n >= 2 ? IZ[n - 2] : 0.0;
    __syncthreads();
    auto step = [&](int i, const double (&tc)[KMAX], double (&tn)[KMAX]) {
        const double pe = pe_nx, iz = iz_nx;
        if (i >= 1) {
            iz_nx = IZ[i - 1];
            if (owner) pe_nx = PE[(int64_t)(i - 1) * S + j];
            const double *Tn = Pt + (int64_t)HMM_BLK(i - 1) * S * S;
#pragma unroll
            for (int m = 0; m < KMAX; ++m) {
                const int k = q * KPT + m;
                if (EXACT || (active && m < KPT && k < S)) tn[m] = Tn[(int64_t)jr * S + k];
            }
        }
        const double *w_next = w_buf + cur * S;
        double s0 = 0.0, s1 = 0.0, s2 = 0.0;
#pragma unroll
        for (int m = 0; m < KMAX; ++m) {
            const int k = q * KPT + m;
            if (EXACT || (active && m < KPT && k < S)) {
                const double wv = w_next[k];
                if (m % 3 == 0) s0 = fma(wv, tc[m], s0);
                else if (m % 3 == 1) s1 = fma(wv, tc[m], s1);
                else s2 = fma(wv, tc[m], s2);
            }
        }
        const double bh = quad_sum((s0 + s1) + s2) * iz;
        const int nxt = cur ^ 1;
        if (owner) {
            w_buf[nxt * S + j] = bh * pe;
            BH[(int64_t)i * S + j] = bh;
        }
        lds_barrier();
        cur = nxt;
    };
    int i = n - 2;
    for (; i >= 1; i -= 2) {
        step(i, ta, tb);
        step(i - 1, tb, ta);
    }
    if (i == 0) step(0, ta, tb);
}

// ------------------------------------------------------------------------------------------
// Single-wave recursions (even S <= 64, i.e. the 8-founder S = 36 case): one lane per state holds
// its whole transition row in registers, the previous step's vector goes through LDS as broadcast
// 128-bit reads, and because one wave executes its LDS instructions in order there is no
// s_barrier, no cross-lane reduction and no predicate in the loop.  A lone wave retires one
// instruction every ~8 cycles on these dependent chains, so the step time is the instruction
// count: this form has ~40% of the multi-wave kernels' instructions per step.
// Lanes j >= S shadow state S-1 and never store.
// ------------------------------------------------------------------------------------------
template <int SS>
__device__ __forceinline__ void load_row(const double *__restrict__ row, double (&dst)[SS]) {
    const double2 *src = reinterpret_cast<const double2 *>(row);
#pragma unroll
    for (int m = 0; m < SS / 2; ++m) {
        const double2 v = src[m];
        dst[2 * m] = v.x;
        dst[2 * m + 1] = v.y;
    }
}

template <int SS>
__device__ __forceinline__ void read_vec(const double *lds_vec, double2 (&v)[SS / 2]) {
    const double2 *src = reinterpret_cast<const double2 *>(lds_vec);
#pragma unroll
    for (int m = 0; m < SS / 2; ++m) v[m] = src[m];
    asm volatile("" ::: "memory");
}

// In-place refill of a register set: the reload of element m must issue after the product that
// consumed the old element m (otherwise old and new values are live together and the set costs
// twice its registers; the scheduler hoists independent loads to the top of the step).  The empty
// asm makes the load's per-lane offset formally depend on that product.
__device__ __forceinline__ void pin_after(unsigned &lane_off, double consumed) {
    asm volatile("" : "+v"(lane_off) : "v"(consumed));
}
// uniform base (SGPR pair) + 32-bit unsigned lane offset: the saddr form of global_load, no per-load
// 64-bit address arithmetic
__device__ __forceinline__ double2 load_pinned(const char *uniform_base, unsigned lane_off_bytes) {
    return *reinterpret_cast<const double2 *>(uniform_base + lane_off_bytes);
}

// lane j's row from a lane-ordered block (lane_blocks_kernel, LPS = 1): double2 m at (m*SS + j)*16 bytes
template <int SS>
__device__ __forceinline__ void load_lane(const double *__restrict__ block, int lane, double (&dst)[SS]) {
    const char *src = reinterpret_cast<const char *>(block) + lane * 16;
#if defined(HMM_ABLATE_HALF_LOADS)
    constexpr int NLOAD = SS / 4;                        // timing only: the second half keeps what it had
#else
    constexpr int NLOAD = SS / 2;
#endif
#pragma unroll
    for (int m = 0; m < NLOAD; ++m) {
        const double2 v = *reinterpret_cast<const double2 *>(src + m * (SS * 16));
        dst[2 * m] = v.x;
        dst[2 * m + 1] = v.y;
    }
}

// index of the first maximum, from four running maxima over interleaved index classes (k = 0, 1, 2, 3 mod 4), each with
// the index where it was first reached: on equal values the smaller index wins, which is np.argmax's rule
__device__ __forceinline__ int first_max4(double m0, int k0, double m1, int k1, double m2, int k2, double m3, int k3) {
    if (m1 > m0 || (m1 == m0 && k1 < k0)) { m0 = m1; k0 = k1; }
    if (m3 > m2 || (m3 == m2 && k3 < k2)) { m2 = m3; k2 = k3; }
    if (m2 > m0 || (m2 == m0 && k2 < k0)) k0 = k2;
    return k0;
}

// orders this wave's LDS writes before its later LDS reads for the compiler; the hardware keeps
// one wave's DS instructions in order
__device__ __forceinline__ void wave_lds_fence() {
    asm volatile("" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    asm volatile("" ::: "memory");
}

// Grid (sample, chromosome rank): the samples of a chromosome are neighbours in dispatch order, so
// they stream the same transition blocks at the same time (one HBM fetch per XCD instead of one
// per sample), and order[] lists the chromosomes longest first (no long chain starting last).
//
// Transition blocks are prefetched NSET-1 steps ahead into a ring of NSET register sets (the loop is
// unrolled NSET-fold so every set index is a compile-time constant): an HBM miss costs ~0.4 us,
// a step ~0.25 us, so one step of lead leaves the miss exposed.  Order o <-> the o-th step of the
// sweep; set o % NSET holds its block, the step refills the set its predecessor just released.
//
// Forward: ROLE 0 = alpha recursion, ROLE 1 = delta recursion (values only: the backpointers are
// argmax_k(delta_t[k] + T[t][j][k]) of stored rows, which viterbi_bp_kernel evaluates for every t
// in parallel instead of on the sequential path).  The two are separate launches on separate
// streams (each keeps its own register budget).
template <int SS, int NSET, int SB, int ROLE, int HB, bool BP = false /* role 1: write the backpointers of every step too */>
__global__ void __launch_bounds__(64, HMM_CHAIN_WAVES(NSET))
forward_wave_kernel(int n_samples, int64_t genes_per_sample, const ChromDesc *__restrict__ chroms,
                    const int32_t *__restrict__ order, const double *__restrict__ tprob, const double *__restrict__ pprob,
                    const double *__restrict__ eprob, const double *__restrict__ peprob,
                    const double *__restrict__ init_vec, double *__restrict__ xsum,
                    double *__restrict__ invz, double *__restrict__ delta,
                    int32_t *__restrict__ last_state, const double *__restrict__ inject, int n_real_chrom,
                    int inject_slots /* boundary vectors per sample in `inject` */,
                    const int32_t *__restrict__ only_if = nullptr /* [sample][chromosome]: run only where set (the fallback of
                                                                     the rank-convergence delta, hmm_blocked.inc) */,
                    uint16_t *__restrict__ bp = nullptr, int64_t bp_per_sample = 0 /* BP instances */) {
    static_assert(SS % 2 == 0 && SS <= 64, "one lane per state, 16-byte aligned rows");
    __shared__ __attribute__((aligned(16))) double buf[SB][2][SS];
    const int chrom = order[blockIdx.y];
    const ChromDesc cd = chroms[chrom];
    // injected start (blocked scan): the descriptor's first gene belongs to the previous block - its vector comes from
    // `inject` ([sample][slot][state]) and nothing is stored for it
    const bool injected = cd.inject >= 0;
    constexpr int role = ROLE;
    const int n = cd.n_genes;
    if (n <= 0) return;
    const int j = threadIdx.x;
    const bool act = j < SS;
    const int jr = act ? j : SS - 1;
    // the SB samples of this wave share every transition block it loads; a sample index past the
    // end shadows the last sample and never stores
    bool sv[SB];
    int64_t g0[SB];
#pragma unroll
    for (int b = 0; b < SB; ++b) {
        const int sample = blockIdx.x * SB + b;
        sv[b] = sample < n_samples;
        g0[b] = (int64_t)min(sample, n_samples - 1) * genes_per_sample + cd.gene_off;
    }
    if (only_if) {
        bool wanted = false;
#pragma unroll
        for (int b = 0; b < SB; ++b) {
            sv[b] = sv[b] && only_if[(int64_t)(blockIdx.x * SB + b) * n_real_chrom + chrom] != 0;
            wanted = wanted || sv[b];
        }
        if (!wanted) return;
    }
    const int n_ord = min(n, cd.n_trans + 1) - 1;   // step i = o + 1 needs T[o]
    int cur = 0;
    // role 0 streams P = exp(T) and pe = exp(e); role 1 streams T and e
    const double *BLK = (role == 0 ? pprob : tprob) + cd.trans_off * (int64_t)SS * SS;
    const double *EM = role == 0 ? peprob : eprob;
    // Loads are unconditional with clamped block indices (a conditional refill would make the
    // compiler merge whole register sets at every join): past the end they re-read the last block.
    const int last_o = max(n_ord - 1, 0);
    const bool any = cd.n_trans > 0 && n > 1;
    double pr[NSET][SS], em_r[NSET][SB];
#pragma unroll
    for (int u = 0; u < NSET; ++u) {
        const int ou = min(u, last_o);
        if (any) load_lane<SS>(BLK + (int64_t)HMM_BLK(ou) * SS * SS, jr, pr[u]);
        else {
#pragma unroll
            for (int m = 0; m < SS; ++m) pr[u][m] = 0.0;
        }
#pragma unroll
        for (int b = 0; b < SB; ++b) em_r[u][b] = any ? EM[(g0[b] + ou + 1) * SS + jr] : 0.0;
    }
    auto prefetch = [&](int o, double (&pn)[SS], double (&em_n)[SB]) {
        const int of = min(o + NSET - 1, last_o);
#if defined(HMM_ABLATE_LOADS)                            // timing-only builds (scripts/hmm_single_ablate.sh): the sets keep their first blocks
        (void)pn; (void)em_n; (void)of;
        return;
#endif
        load_lane<SS>(BLK + (int64_t)HMM_BLK(of) * SS * SS, jr, pn);
#pragma unroll
        for (int b = 0; b < SB; ++b) em_n[b] = EM[(g0[b] + of + 1) * SS + jr];
    };

    if constexpr (role == 0) {
        double y_own[SB];
#pragma unroll
        for (int b = 0; b < SB; ++b) {
            if (injected) {
                y_own[b] = inject[((int64_t)min((int)blockIdx.x * SB + b, n_samples - 1) * inject_slots + cd.inject) * SS + jr];
            } else {
                y_own[b] = exp(init_vec[jr] + eprob[g0[b] * SS + jr]);
                if (act && sv[b]) xsum[g0[b] * SS + j] = exp(init_vec[j]);   // so that log(x) + e reproduces init + e
            }
            if (act) buf[b][0][j] = y_own[b];
        }
        wave_lds_fence();
        // Z of the vector in buf[.][cur] comes out of the same broadcast reads as the products:
        // every lane adds it up itself, no cross-lane operation
        auto finish = [&](int b, int i_prev, double z) {
#if defined(HMM_ABLATE_RECIP)
            const double inv_z = __hiloint2double(0x3FF00000 | (__double2hiint(z) & 0xFFFF), 0);
#else
            const double inv_z = fast_recip_pos(z);
#endif
#if !defined(HMM_ABLATE_STORES)
            if (act && sv[b]) {
                if (j == 0 && (i_prev > 0 || !injected)) invz[g0[b] + i_prev] = inv_z;
            }
#endif
            return inv_z;
        };
        auto step = [&](int o, const double (&pc)[SS], const double (&pe)[SB], double (&pn)[SS], double (&pe_n)[SB]) {
            const int i = o + 1;
            double pe_now[SB];
#pragma unroll
            for (int b = 0; b < SB; ++b) pe_now[b] = pe[b];
            prefetch(o, pn, pe_n);
#pragma unroll
            for (int b = 0; b < SB; ++b) {
                // HB > 0: the broadcast reads go out HB at a time before the products that use them
                // (a compiler left alone keeps ~4 in flight and the step becomes a chain of LDS
                // latencies); a batch costs 4*HB registers, so HB is sized to what the sets leave
                constexpr int NV = SS / 2, BATCH = HB > 0 ? HB : NV;
                const double2 *yv = reinterpret_cast<const double2 *>(buf[b][cur]);
                double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0, z0 = 0.0, z1 = 0.0, z2 = 0.0, z3 = 0.0;
#pragma unroll
                for (int mb = 0; mb < NV; mb += BATCH) {
                    double2 yv_r[BATCH];
                    if constexpr (HB > 0) {
#pragma unroll
                        for (int q = 0; q < BATCH; ++q)
                            if (mb + q < NV) yv_r[q] = yv[mb + q];
                        asm volatile("" ::: "memory");
                    }
#pragma unroll
                    for (int q = 0; q < BATCH; ++q) {
                        const int m = mb + q;
                        if (m >= NV) break;
                        const double2 v = HB > 0 ? yv_r[q] : yv[m];
                        if (m & 1) {
                            s2 = fma(v.x, pc[2 * m], s2);
                            s3 = fma(v.y, pc[2 * m + 1], s3);
                            z2 += v.x;
                            z3 += v.y;
                        } else {
                            s0 = fma(v.x, pc[2 * m], s0);
                            s1 = fma(v.y, pc[2 * m + 1], s1);
                            z0 += v.x;
                            z1 += v.y;
                        }
                    }
                }
                const double inv_z = finish(b, i - 1, (z0 + z1) + (z2 + z3));
                const double x = ((s0 + s1) + (s2 + s3)) * inv_z + TINY;
                y_own[b] = x * pe_now[b];
                if (act) buf[b][cur ^ 1][j] = y_own[b];
#if !defined(HMM_ABLATE_STORES)
                if (act && sv[b]) xsum[(g0[b] + i) * SS + j] = x;
#endif
            }
            cur ^= 1;
            wave_lds_fence();
        };
        int o = 0;
        for (; o + NSET <= n_ord; o += NSET) {
#pragma unroll
            for (int u = 0; u < NSET; ++u)
                step(o + u, pr[u], em_r[u], pr[(u + NSET - 1) % NSET], em_r[(u + NSET - 1) % NSET]);
        }
#pragma unroll
        for (int u = 0; u < NSET - 1; ++u)
            if (o + u < n_ord) step(o + u, pr[u], em_r[u], pr[(u + NSET - 1) % NSET], em_r[(u + NSET - 1) % NSET]);
#pragma unroll
        for (int b = 0; b < SB; ++b) {
            const double2 *yv = reinterpret_cast<const double2 *>(buf[b][cur]);
            double z0 = 0.0, z1 = 0.0, z2 = 0.0, z3 = 0.0;
#pragma unroll
            for (int m = 0; m < SS / 2; ++m) {
                const double2 v = yv[m];
                if (m & 1) { z2 += v.x; z3 += v.y; } else { z0 += v.x; z1 += v.y; }
            }
            finish(b, n_ord, (z0 + z1) + (z2 + z3));
        }
        return;
    }

    // role 1: Viterbi values in the log domain (adds and max only: exact)
#pragma unroll
    for (int b = 0; b < SB; ++b) {
        double d0;
        if (injected) {
            d0 = inject[((int64_t)min((int)blockIdx.x * SB + b, n_samples - 1) * inject_slots + cd.inject) * SS + jr];
        } else {
            d0 = init_vec[jr] + EM[g0[b] * SS + jr];
            if (act && sv[b]) delta[g0[b] * SS + j] = d0;
        }
        if (act) buf[b][0][j] = d0;
    }
    wave_lds_fence();
    auto step = [&](int o, const double (&tc)[SS], const double (&e)[SB], double (&tn)[SS], double (&e_n)[SB]) {
        const int i = o + 1;
        double e_now[SB];
#pragma unroll
        for (int b = 0; b < SB; ++b) e_now[b] = e[b];
        prefetch(o, tn, e_n);
#pragma unroll
        for (int b = 0; b < SB; ++b) {
            constexpr int NV = SS / 2, BATCH = HB > 0 ? HB : NV;
            const double2 *dp = reinterpret_cast<const double2 *>(buf[b][cur]);
            double m0 = 0.0, m1 = 0.0, m2 = 0.0, m3 = 0.0;
            int k0 = 0, k1 = 1, k2 = 2, k3 = 3;          // BP: where each running maximum was first reached
#pragma unroll
            for (int mb = 0; mb < NV; mb += BATCH) {
                double2 dp_r[BATCH];
                if constexpr (HB > 0) {
#pragma unroll
                    for (int q = 0; q < BATCH; ++q)
                        if (mb + q < NV) dp_r[q] = dp[mb + q];
                    asm volatile("" ::: "memory");
                }
#pragma unroll
                for (int q = 0; q < BATCH; ++q) {
                    const int m = mb + q;
                    if (m >= NV) break;
                    const double2 v = HB > 0 ? dp_r[q] : dp[m];
                    const double a = v.x + tc[2 * m], c = v.y + tc[2 * m + 1];
                    if (m == 0) { m0 = a; m1 = c; }
                    else if (m == 1) { m2 = a; m3 = c; }
                    else if (m & 1) {
                        if constexpr (BP) { k2 = a > m2 ? 2 * m : k2; k3 = c > m3 ? 2 * m + 1 : k3; }
                        m2 = fmax(m2, a); m3 = fmax(m3, c);
                    } else {
                        if constexpr (BP) { k0 = a > m0 ? 2 * m : k0; k1 = c > m1 ? 2 * m + 1 : k1; }
                        m0 = fmax(m0, a); m1 = fmax(m1, c);
                    }
                }
            }
            if constexpr (BP) {
                // backpointer row of the gene this step starts from (gbrs_utils.py:567-579 takes argmax(delta[:, t] + T[t][j]):
                // the first maximum): merge the four running maxima, the smaller index on equal values
                if (act && sv[b]) bp[((int64_t)(blockIdx.x * SB + b) * bp_per_sample + cd.bp_off + o) * SS + j] =
                    (uint16_t)first_max4(m0, k0, m1, k1, m2, k2, m3, k3);
            }
            const double d = fmax(fmax(m0, m1), fmax(m2, m3)) + e_now[b];
            if (act) buf[b][cur ^ 1][j] = d;
#if !defined(HMM_ABLATE_STORES)
            if (act && sv[b]) delta[(g0[b] + i) * SS + j] = d;
#endif
        }
        cur ^= 1;
        wave_lds_fence();
    };
    int o = 0;
    for (; o + NSET <= n_ord; o += NSET) {
#pragma unroll
        for (int u = 0; u < NSET; ++u)
            step(o + u, pr[u], em_r[u], pr[(u + NSET - 1) % NSET], em_r[(u + NSET - 1) % NSET]);
    }
#pragma unroll
    for (int u = 0; u < NSET - 1; ++u)
        if (o + u < n_ord) step(o + u, pr[u], em_r[u], pr[(u + NSET - 1) % NSET], em_r[(u + NSET - 1) % NSET]);
    if constexpr (BP) {
        // the chromosome's last gene has a backpointer row too when its transition block exists (tprob of length n, the DO
        // convention): argmax(delta[:, n-1] + T[n-1][j]) - no step of the recursion uses that block
        if (cd.real_chrom >= 0 && cd.n_trans > n_ord) {
            double tl[SS];
            load_lane<SS>(BLK + (int64_t)n_ord * SS * SS, jr, tl);
#pragma unroll
            for (int b = 0; b < SB; ++b) {
                const double2 *dp = reinterpret_cast<const double2 *>(buf[b][cur]);
                double m0 = 0.0, m1 = 0.0, m2 = 0.0, m3 = 0.0;
                int k0 = 0, k1 = 1, k2 = 2, k3 = 3;
#pragma unroll
                for (int m = 0; m < SS / 2; ++m) {
                    const double2 v = dp[m];
                    const double a = v.x + tl[2 * m], c = v.y + tl[2 * m + 1];
                    if (m == 0) { m0 = a; m1 = c; }
                    else if (m == 1) { m2 = a; m3 = c; }
                    else if (m & 1) { k2 = a > m2 ? 2 * m : k2; k3 = c > m3 ? 2 * m + 1 : k3; m2 = fmax(m2, a); m3 = fmax(m3, c); }
                    else { k0 = a > m0 ? 2 * m : k0; k1 = c > m1 ? 2 * m + 1 : k1; m0 = fmax(m0, a); m1 = fmax(m1, c); }
                }
                if (act && sv[b]) bp[((int64_t)(blockIdx.x * SB + b) * bp_per_sample + cd.bp_off + n_ord) * SS + j] =
                    (uint16_t)first_max4(m0, k0, m1, k1, m2, k2, m3, k3);
            }
        }
    }
    if (j < SB && blockIdx.x * SB + j < n_samples && cd.real_chrom >= 0) {      // sid = argmax delta[:, n-1] (first max)
        const double *dl = buf[j][cur];
        double bv = dl[0];
        int bk = 0;
        for (int s = 1; s < SS; ++s)
            if (dl[s] > bv) { bv = dl[s]; bk = s; }
        last_state[(int64_t)(blockIdx.x * SB + j) * n_real_chrom + cd.real_chrom] = bk;
    }
}

// ------------------------------------------------------------------------------------------
// The same three chains for state counts beyond one wave (S = 4*KMAX, KMAX even; S = 136 for 16
// founders): 4 adjacent lanes per state, each holding a contiguous quarter of the state's
// transition row in registers (KMAX doubles), quad sums / maxima on DPP, the vector double
// buffered in LDS with one LDS-only barrier per step.  A step moves 8*S*S bytes (148 KB at
// S = 136) through one CU, ~1-2 us, so one step of lead (two register sets) hides the HBM miss.
// ROLE 0 = alpha (Z from the lanes' own quarter sums + quad sum), 1 = delta values, 2 = free-running
// backward (see backward_wave_kernel).  Padding lanes shadow the last state and never store.
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ double quad_max(double v) {
    v = fmax(v, dpp_f64<DPP_QUAD_XOR1>(v));
    v = fmax(v, dpp_f64<DPP_QUAD_XOR2>(v));
    return v;
}

template <int LPS>
__device__ __forceinline__ double group_sum(double v) {
    v += dpp_f64<DPP_QUAD_XOR1>(v);
    if (LPS == 4) v += dpp_f64<DPP_QUAD_XOR2>(v);
    return v;
}
template <int LPS>
__device__ __forceinline__ double group_max(double v) {
    v = fmax(v, dpp_f64<DPP_QUAD_XOR1>(v));
    if (LPS == 4) v = fmax(v, dpp_f64<DPP_QUAD_XOR2>(v));
    return v;
}

template <int LPS, int KMAX, int NSET, int HBQ_SUM, int ROLE>
__global__ void __launch_bounds__(((LPS * LPS * KMAX + 63) / 64) * 64)
group_chain_kernel(int64_t genes_per_sample, const ChromDesc *__restrict__ chroms,
                  const int32_t *__restrict__ order, const double *__restrict__ blocks,
                  const double *__restrict__ em, const double *__restrict__ eprob,
                  const double *__restrict__ init_vec, double *__restrict__ out_vec /* xsum | delta | bhat */,
                  double *__restrict__ scal /* invz | - | bscale */,
                  int32_t *__restrict__ last_state) {
    static_assert(KMAX % 2 == 0 && (LPS == 2 || LPS == 4), "16-byte aligned row parts, DPP group of 2 or 4");
    constexpr int S = LPS * KMAX, NV = KMAX / 2;   // NSET register sets refilled in place: NSET steps of lead
    __shared__ __attribute__((aligned(16))) double buf[2][S];
    const int chrom = order[blockIdx.y];
    const ChromDesc cd = chroms[chrom];
    const int sample = blockIdx.x;
    const int n = cd.n_genes;
    if (n <= 0) return;
    const int j = threadIdx.x / LPS, q = threadIdx.x % LPS;
    const bool valid = j < S;
    const int jr = valid ? j : S - 1;
    const bool owner = valid && q == 0;
    const int64_t g0 = (int64_t)sample * genes_per_sample + cd.gene_off;
    // tables in lane order (lane_blocks_kernel): double2 m of this lane at byte (m*LPS*S + lane)*16
    const double *BLK0 = blocks + cd.trans_off * (int64_t)S * S;
    unsigned row_off = min((int)threadIdx.x, LPS * S - 1) * 16;
    constexpr int M_STRIDE = LPS * S * 16;
    // order o: forward roles step i = o + 1 on block o; backward gene i = n-2-o on block i
    const int n_ord = ROLE == 2 ? n - 1 : min(n, cd.n_trans + 1) - 1;
    const int last_o = max(n_ord - 1, 0);
    auto blk_of = [&](int o) { return ROLE == 2 ? n - 2 - o : o; };
    auto em_of = [&](int o) { return ROLE == 2 ? n - 2 - o : o + 1; };
    double pr[NSET][KMAX], em_r[NSET];
#pragma unroll
    for (int u = 0; u < NSET; ++u) {
        const int ou = min(u, last_o);
        if (n_ord > 0) {
            const char *src = reinterpret_cast<const char *>(BLK0 + (int64_t)HMM_BLK(blk_of(ou)) * S * S) + row_off;
#pragma unroll
            for (int m = 0; m < NV; ++m) {
                const double2 g = *reinterpret_cast<const double2 *>(src + (size_t)m * M_STRIDE);
                pr[u][2 * m] = g.x;
                pr[u][2 * m + 1] = g.y;
            }
        } else {
#pragma unroll
            for (int m = 0; m < KMAX; ++m) pr[u][m] = 0.0;
        }
        em_r[u] = n_ord > 0 ? em[(g0 + em_of(ou)) * S + jr] : 0.0;
    }
    int cur = 0;
    double own = 0.0;                               // this state's current vector entry
    if (ROLE == 0) {
        own = exp(init_vec[jr] + eprob[g0 * S + jr]);
        if (owner) out_vec[g0 * S + j] = exp(init_vec[j]);          // so that log(x) + e reproduces init + e
    } else if (ROLE == 1) {
        own = init_vec[jr] + em[g0 * S + jr];
        if (owner) out_vec[g0 * S + j] = own;
    } else {
        own = em[(g0 + n - 1) * S + jr];           // pe_{n-1} * bt_{n-1}, bt_{n-1} = 1
        if (owner) {
            out_vec[(g0 + n - 1) * S + j] = 1.0;
            if (j == 0) scal[g0 + n - 1] = 1.0;
        }
    }
    if (owner) buf[0][j] = own;
    __syncthreads();
    auto alpha_finish = [&](int i_prev, double z) {
        const double inv_z = fast_recip_pos(z);
        if (owner) {
            if (j == 0) scal[g0 + i_prev] = inv_z;
        }
        return inv_z;
    };
    auto step = [&](int o, bool rescale, double (&pc)[KMAX], double &e_slot) {
        const double e = e_slot;
        const int of = min(o + NSET, last_o);
        e_slot = em[(g0 + em_of(of)) * S + jr];
        const double *nblk = BLK0 + (int64_t)HMM_BLK(blk_of(of)) * S * S;
        const double2 *src = reinterpret_cast<const double2 *>(buf[cur] + q * KMAX);
        double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0, z0 = 0.0, z1 = 0.0;
        // LDS reads go out in batches sized to the register budget (168 with 9 waves per workgroup)
        constexpr int HBQ = ROLE == 1 ? (NV + 1) / 2 : HBQ_SUM;
#pragma unroll
        for (int mb = 0; mb < NV; mb += HBQ) {
            double2 v[HBQ];
#pragma unroll
            for (int t = 0; t < HBQ; ++t)
                if (mb + t < NV) v[t] = src[mb + t];
            asm volatile("" ::: "memory");
#pragma unroll
            for (int t = 0; t < HBQ; ++t) {
                const int m = mb + t;
                if (m >= NV) break;
                if (ROLE == 1) {
                    const double x = v[t].x + pc[2 * m], y = v[t].y + pc[2 * m + 1];
                    if (m == 0) { a0 = x; a1 = y; }
                    else if (m == 1) { a2 = x; a3 = y; }
                    else if (m & 1) { a2 = fmax(a2, x); a3 = fmax(a3, y); }
                    else { a0 = fmax(a0, x); a1 = fmax(a1, y); }
                } else {
                    if (m & 1) { a2 = fma(v[t].x, pc[2 * m], a2); a3 = fma(v[t].y, pc[2 * m + 1], a3); }
                    else { a0 = fma(v[t].x, pc[2 * m], a0); a1 = fma(v[t].y, pc[2 * m + 1], a1); }
                    if (ROLE == 0 || rescale) {
                        if (m & 1) z1 += v[t].x + v[t].y; else z0 += v[t].x + v[t].y;
                    }
                }
                pin_after(row_off, (m & 1) ? a3 : a1);          // in-place refill: see forward_wave_kernel
                const double2 g = load_pinned(reinterpret_cast<const char *>(nblk) + (size_t)m * M_STRIDE, row_off);
                pc[2 * m] = g.x;
                pc[2 * m + 1] = g.y;
            }
        }
        const int i = ROLE == 2 ? n - 2 - o : o + 1;
        if (ROLE == 0) {
            const double inv_z = alpha_finish(i - 1, group_sum<LPS>(z0 + z1));
            const double x = group_sum<LPS>((a0 + a1) + (a2 + a3)) * inv_z + TINY;
            own = x * e;
            if (owner) out_vec[(g0 + i) * S + j] = x;
        } else if (ROLE == 1) {
            own = group_max<LPS>(fmax(fmax(a0, a1), fmax(a2, a3))) + e;
            if (owner) out_vec[(g0 + i) * S + j] = own;
        } else {
            const double r = rescale ? fast_recip_pos(group_sum<LPS>(z0 + z1)) : 1.0;
            const double bh = group_sum<LPS>((a0 + a1) + (a2 + a3)) * r;
            own = bh * e;
            if (owner) {
                out_vec[(g0 + i) * S + j] = bh;
                if (j == 0) scal[g0 + i] = r;
            }
        }
        if (owner) buf[cur ^ 1][j] = own;
        cur ^= 1;
        lds_barrier();
    };
    int o = 0;                                   // backward: rescale every step (one loop body)
    for (; o + NSET <= n_ord; o += NSET) {
#pragma unroll
        for (int u = 0; u < NSET; ++u) step(o + u, true, pr[u], em_r[u]);
    }
#pragma unroll
    for (int u = 0; u < NSET - 1; ++u)
        if (o + u < n_ord) step(o + u, true, pr[u], em_r[u]);
    if (ROLE == 0) {
        const double2 *src = reinterpret_cast<const double2 *>(buf[cur] + q * KMAX);
        double z0 = 0.0, z1 = 0.0;
#pragma unroll
        for (int m = 0; m < NV; ++m) {
            const double2 v = src[m];
            if (m & 1) z1 += v.x + v.y; else z0 += v.x + v.y;
        }
        alpha_finish(n_ord, group_sum<LPS>(z0 + z1));
    }
    if (ROLE == 1 && threadIdx.x == 0) {       // sid = argmax delta[:, n-1] (first max)
        const double *dl = buf[cur];
        double bv = dl[0];
        int bk = 0;
        for (int t = 1; t < S; ++t)
            if (dl[t] > bv) { bv = dl[t]; bk = t; }
        last_state[(int64_t)sample * gridDim.y + chrom] = bk;
    }
}

// Backpointers bp[t][j] = argmax_k(delta_t[k] + T[t][j][k]) (first max, np.argmax) for
// t < min(n, n_trans): the quantity the reference's backtrace recomputes along the path
// (gbrs_utils.py:594), including the row of T[n-1] when len(tprob) >= n.  One workgroup per
// (t, chromosome): the block is staged once in LDS and applied to every sample's delta row.
__global__ void __launch_bounds__(256)
viterbi_bp_kernel(int S, int n_samples, int64_t genes_per_sample, int64_t bp_per_sample,
                  const ChromDesc *__restrict__ chroms, const double *__restrict__ tprob,
                  const double *__restrict__ delta, uint16_t *__restrict__ bp) {
    extern __shared__ double lds[];            // T block, rows padded to S + 1
    const ChromDesc cd = chroms[blockIdx.y];
    const int t = blockIdx.x;
    if (t >= min(cd.n_genes, cd.n_trans)) return;
    const double *T = tprob + (cd.trans_off + t) * (int64_t)S * S;
    const int stride = S + 1;
    for (int x = threadIdx.x; x < S * S; x += blockDim.x) lds[(x / S) * stride + x % S] = T[x];
    __syncthreads();
    const int per_pass = blockDim.x / S;       // samples per pass
    const int sl = threadIdx.x / S, j = threadIdx.x % S;
    if (sl >= per_pass) return;
    const double *row = lds + j * stride;
    for (int s0 = 0; s0 < n_samples; s0 += per_pass) {
        const int sample = s0 + sl;
        if (sample >= n_samples) break;
        const double *d = delta + ((int64_t)sample * genes_per_sample + cd.gene_off + t) * S;
        double best = d[0] + row[0];
        int best_k = 0;
        for (int k = 1; k < S; ++k) {
            const double dv = d[k] + row[k];
            if (dv > best) { best = dv; best_k = k; }
        }
        bp[((int64_t)sample * bp_per_sample + cd.bp_off + t) * S + j] = (uint16_t)best_k;
    }
}

// Backpointers for few samples and a single-wave state count: one wavefront takes BPW_ROWS consecutive genes of a
// chromosome; lane j loads row j of every block from the lane-ordered table (coalesced, as the chain kernels do), the
// delta row goes through LDS as broadcast reads.  viterbi_bp_kernel above launches one 256-thread workgroup per gene
// and stages the block in LDS - with one sample that is 52 k workgroups of which 36 threads work (0.12 ms, on the
// critical path of the Viterbi chain); this form is ~5 k wavefronts.
constexpr int BPW_ROWS = 8;
template <int SS>
__global__ void __launch_bounds__(64)
viterbi_bp_wave_kernel(int n_samples, int64_t genes_per_sample, int64_t bp_per_sample,
                       const ChromDesc *__restrict__ chroms, const double *__restrict__ tprob_q,
                       const double *__restrict__ delta, uint16_t *__restrict__ bp) {
    __shared__ __attribute__((aligned(16))) double drow[SS];
    const ChromDesc cd = chroms[blockIdx.y];
    const int rows = min(cd.n_genes, cd.n_trans);
    const int t0 = blockIdx.x * BPW_ROWS;
    if (t0 >= rows) return;
    const int j = threadIdx.x;
    const bool act = j < SS;
    const int jr = act ? j : SS - 1;
    for (int t = t0; t < min(t0 + BPW_ROWS, rows); ++t) {
        double tr[SS];
        load_lane<SS>(tprob_q + (cd.trans_off + t) * (int64_t)SS * SS, jr, tr);
        for (int sample = 0; sample < n_samples; ++sample) {
            const double *d = delta + ((int64_t)sample * genes_per_sample + cd.gene_off + t) * SS;
            wave_lds_fence();
            if (act) drow[j] = d[j];
            wave_lds_fence();
            const double2 *dv = reinterpret_cast<const double2 *>(drow);
            double best = 0.0;
            int best_k = 0;
#pragma unroll
            for (int m = 0; m < SS / 2; ++m) {           // first maximum, np.argmax's rule
                const double2 v = dv[m];
                const double a = v.x + tr[2 * m], c = v.y + tr[2 * m + 1];
                if (m == 0 || a > best) { best = a; best_k = 2 * m; }
                if (c > best) { best = c; best_k = 2 * m + 1; }
            }
            if (act) bp[((int64_t)sample * bp_per_sample + cd.bp_off + t) * SS + j] = (uint16_t)best_k;
        }
    }
}

// Backpointers for the quad chains: same result as viterbi_bp_kernel, but the block comes straight
// from the lane-ordered table (coalesced, no 148 KB LDS image) into registers and is reused for
// every sample; 4 lanes per state, quad argmax with np.argmax's first-max rule.
template <int KMAX>
__global__ void __launch_bounds__(((KMAX * 16 + 63) / 64) * 64)
viterbi_bp_quad_kernel(int n_samples, int64_t genes_per_sample, int64_t bp_per_sample,
                       const ChromDesc *__restrict__ chroms, const double *__restrict__ tprob_q,
                       const double *__restrict__ delta, uint16_t *__restrict__ bp) {
    constexpr int S = 4 * KMAX, NV = KMAX / 2, M_STRIDE = 4 * S * 16;
    __shared__ __attribute__((aligned(16))) double drow[S];
    const ChromDesc cd = chroms[blockIdx.y];
    const int t = blockIdx.x;
    if (t >= min(cd.n_genes, cd.n_trans)) return;
    const int j = threadIdx.x / 4, q = threadIdx.x % 4;
    const bool owner = j < S && q == 0;
    const char *src = reinterpret_cast<const char *>(tprob_q + (cd.trans_off + t) * (int64_t)S * S) +
                      min((int)threadIdx.x, 4 * S - 1) * 16;
    double tr[KMAX];
#pragma unroll
    for (int m = 0; m < NV; ++m) {
        const double2 g = *reinterpret_cast<const double2 *>(src + (size_t)m * M_STRIDE);
        tr[2 * m] = g.x;
        tr[2 * m + 1] = g.y;
    }
    for (int sample = 0; sample < n_samples; ++sample) {
        const double *d = delta + ((int64_t)sample * genes_per_sample + cd.gene_off + t) * S;
        __syncthreads();
        for (int x = threadIdx.x; x < S; x += blockDim.x) drow[x] = d[x];
        __syncthreads();
        const double2 *dv = reinterpret_cast<const double2 *>(drow + q * KMAX);
        double best = -DBL_MAX;
        int best_k = 0x7fffffff;
#pragma unroll
        for (int m = 0; m < NV; ++m) {
            const double2 v = dv[m];
            const double x = v.x + tr[2 * m], y = v.y + tr[2 * m + 1];
            if (x > best) { best = x; best_k = q * KMAX + 2 * m; }          // ascending k: first max kept
            if (y > best) { best = y; best_k = q * KMAX + 2 * m + 1; }
        }
        quad_argmax(best, best_k);
        if (owner) bp[((int64_t)sample * bp_per_sample + cd.bp_off + t) * S + j] = (uint16_t)best_k;
    }
}

// Backward sweep that does not wait for the forward sweep: instead of the forward normalisers it
// carries its own scale,
//     bt_{n-1} = 1        bt_i[j] = r_i * sum_k Pt_i[j][k] * pe_{i+1}[k] * bt_{i+1}[k]
// with r_i = 1 / sum_k(pe_{i+1}[k] * bt_{i+1}[k]) on every NSET-th step and 1 otherwise (the sums
// come out of the broadcast reads the products use).  The reference's beta differs from log(bt) by
// a per-gene constant, log C_i = sum_{t>=i} log(1/Z_t) - sum_{i<=t<=n-2} log(r_t), which
// beta_corr_kernel adds afterwards; the posterior is scale free.
template <int SS, int NSET, int SB>
__global__ void __launch_bounds__(64, HMM_CHAIN_WAVES(NSET))
backward_wave_kernel(int n_samples, int64_t genes_per_sample, const ChromDesc *__restrict__ chroms,
                     const int32_t *__restrict__ order, const double *__restrict__ pprob_t,
                     const double *__restrict__ peprob, double *__restrict__ bhat, double *__restrict__ bscale,
                     const double *__restrict__ inject, int inject_slots) {
    static_assert(SS % 2 == 0 && SS <= 64, "one lane per state, 16-byte aligned rows");
    __shared__ __attribute__((aligned(16))) double buf[SB][2][SS];
    const ChromDesc cd = chroms[order[blockIdx.y]];
    const int n = cd.n_genes;
    if (n <= 0) return;
    const bool injected = cd.inject >= 0;       // blocked scan: the last gene is the next block's; its pe * bt comes from `inject`
    const int j = threadIdx.x;
    const bool act = j < SS;
    const int jr = act ? j : SS - 1;
    bool sv[SB];
    int64_t g0[SB];
#pragma unroll
    for (int b = 0; b < SB; ++b) {
        const int sample = blockIdx.x * SB + b;
        sv[b] = sample < n_samples;
        g0[b] = (int64_t)min(sample, n_samples - 1) * genes_per_sample + cd.gene_off;
    }
    const double *Pt = pprob_t + cd.trans_off * (int64_t)SS * SS;
    int cur = 0;
#pragma unroll
    for (int b = 0; b < SB; ++b) {
        const int64_t o = (g0[b] + n - 1) * SS + jr;
        if (injected) {
            if (act) buf[b][0][j] = inject[((int64_t)min((int)blockIdx.x * SB + b, n_samples - 1) * inject_slots + cd.inject) * SS + jr];
        } else {
            if (act) buf[b][0][j] = peprob[o];
            if (act && sv[b]) {
                bhat[o] = 1.0;
                if (j == 0) bscale[g0[b] + n - 1] = 1.0;
            }
        }
    }
    // order o <-> gene i = n-2-o, transition block i (the host checked n_trans >= n-1)
    const int n_ord = n - 1;
    const int last_o = max(n_ord - 1, 0);
    double tr[NSET][SS], pe_r[NSET][SB];
#pragma unroll
    for (int u = 0; u < NSET; ++u) {
        const int i = n - 2 - min(u, last_o);
        if (n_ord > 0) load_lane<SS>(Pt + (int64_t)HMM_BLK(i) * SS * SS, jr, tr[u]);
        else {
#pragma unroll
            for (int m = 0; m < SS; ++m) tr[u][m] = 0.0;
        }
#pragma unroll
        for (int b = 0; b < SB; ++b) pe_r[u][b] = n_ord > 0 ? peprob[(g0[b] + i) * SS + jr] : 0.0;
    }
    wave_lds_fence();
    auto step = [&](int o, bool rescale, const double (&tc)[SS], const double (&pe)[SB], double (&tn)[SS],
                    double (&pe_n)[SB]) {
        const int i = n - 2 - o;
        double pe_now[SB];
#pragma unroll
        for (int b = 0; b < SB; ++b) pe_now[b] = pe[b];
        const int in = n - 2 - min(o + NSET - 1, last_o);     // clamped: see forward_wave_kernel
#pragma unroll
        for (int b = 0; b < SB; ++b) pe_n[b] = peprob[(g0[b] + in) * SS + jr];
        load_lane<SS>(Pt + (int64_t)HMM_BLK(in) * SS * SS, jr, tn);
#pragma unroll
        for (int b = 0; b < SB; ++b) {
            // all broadcast reads of the vector go out first (a compiler left alone keeps ~4 in
            // flight and the step becomes a chain of LDS latencies)
            double2 wv[SS / 2];
            read_vec<SS>(buf[b][cur], wv);
            double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0, z0 = 0.0, z1 = 0.0;
#pragma unroll
            for (int m = 0; m < SS / 2; ++m) {
                const double2 v = wv[m];
                if (m & 1) {
                    s2 = fma(v.x, tc[2 * m], s2);
                    s3 = fma(v.y, tc[2 * m + 1], s3);
                    if (rescale) z1 += v.x + v.y;
                } else {
                    s0 = fma(v.x, tc[2 * m], s0);
                    s1 = fma(v.y, tc[2 * m + 1], s1);
                    if (rescale) z0 += v.x + v.y;
                }
            }
            const double r = rescale ? fast_recip_pos(z0 + z1) : 1.0;
            const double bh = ((s0 + s1) + (s2 + s3)) * r;
            if (act) buf[b][cur ^ 1][j] = bh * pe_now[b];
            if (act && sv[b]) {
                bhat[(g0[b] + i) * SS + j] = bh;
                if (j == 0) bscale[g0[b] + i] = r;
            }
        }
        cur ^= 1;
        wave_lds_fence();
    };
    int o = 0;
    for (; o + NSET <= n_ord; o += NSET) {
#pragma unroll
        for (int u = 0; u < NSET; ++u)
            step(o + u, u == 0, tr[u], pe_r[u], tr[(u + NSET - 1) % NSET], pe_r[(u + NSET - 1) % NSET]);
    }
#pragma unroll
    for (int u = 0; u < NSET - 1; ++u)
        if (o + u < n_ord)
            step(o + u, u == 0, tr[u], pe_r[u], tr[(u + NSET - 1) % NSET], pe_r[(u + NSET - 1) % NSET]);
}

// ------------------------------------------------------------------------------------------
// Large sample batches, 36 states: the alpha and backward sweeps of 16 samples per wavefront on
// v_mfma_f64_16x16x4_f64 (D[16x16] += A[16x4] * B[4x16]).  A step is x = M y with M the 36x36 block of the
// gene and y one column per sample, so the samples are the N dimension and nothing is broadcast through LDS.
//
// Operand maps (one f64 per lane): A[i = lane & 15][k = lane >> 4], B[k = lane >> 4][n = lane & 15];
// D: register r of lane (g = lane >> 4, n = lane & 15) is D[row g + 4r][n].  Three row blocks of 16 cover 48
// slots, nine k blocks of 4 the summation, and the point of the arrangement: register r of row block rb is
// exactly the B operand of k block kb = 4 rb + r of the next step (k slot 4 kb + g = 16 rb + 4 r + g = its
// row), so a step's result feeds the next one without any lane movement.  Lane (g, n) owns the slots
// 4 q + g, q = 0..8, of sample n; slot 4 q + g holds state 9 g + q (mf_state), so a lane's nine states are
// 72 contiguous bytes of every per-state row it loads or stores.
// Slots 36..39 (third row block, register 1) are rows of ones: the vector's sum Z arrives in every lane
// group out of the same products; slots 40..47 are zero rows.  mfma_blocks_kernel lays the 27 A operands of
// every transition block out in lane order, two operands per 16-byte load.
// The f64 MFMA rate of this chip equals its vector rate (one 16x16x4 = 16 passes = 64 cycles, measured 30 ns:
// scripts/probes/mfma_f64_rate.hip), so the gain is the instruction count: 27 MFMAs + ~50 vector
// instructions per step and 16 samples instead of ~140 vector instructions per step and sample.
// Samples past the end shadow the last sample: same inputs, same results, and they store them to the same
// places - no predicate anywhere in the loop.
// Measured (profiles/r02_hmm_mfma_batches.txt): launched alone a sweep takes 1.25 us per step against the
// 0.83 us of its 27 MFMAs; side by side with the delta chain every sweep slows down about twofold, at any batch
// size and in either form - f64 MFMA and vector f64 share the double-precision units, so the three chains add
// their times up wherever they share SIMDs.  At 64 samples (160 + 160 + 640 wavefronts) this form therefore
// ties with the vector sweeps (6.2 vs 6.1 ms per pass); from about 96 samples on, when the chip is full, its
// fourfold smaller instruction count wins (256 samples: 17.0 vs 25.9 ms).  With the delta chain on
// delta_lanes_kernel (below) the tie at 64 samples turns into a small win (7.07 vs 7.31 ms per pass with the
// emission), so hmm_launch switches both at 64 samples (HMM_MFMA_MIN, HMM_DLANES_MIN).
// ------------------------------------------------------------------------------------------
typedef double mfma_d4 __attribute__((ext_vector_type(4)));
struct __attribute__((aligned(8))) mf_pair { double x, y; };     // 16 bytes at 8-byte alignment (a lane's row piece starts at 72 g)
constexpr int MF_S = 36, MF_RB = 3, MF_KB = 9, MF_OPS = MF_RB * MF_KB, MF_PAIRS = (MF_OPS + 1) / 2;
constexpr int MF_BLK = MF_PAIRS * 128;                           // doubles per block: [pair][lane][2]
constexpr int MF_Q = 9;                                          // states per lane
__host__ __device__ constexpr int mf_state(int slot) { return 9 * (slot & 3) + (slot >> 2); }

// operand w = kb * 3 + rb of block b: fwd = exp(T)[to = row][from = k] (alpha), bwd = its transpose (backward)
__global__ void mfma_blocks_kernel(int64_t n_blocks, const double *__restrict__ t, double *__restrict__ fwd,
                                   double *__restrict__ bwd) {
    const int64_t total = n_blocks * MF_BLK;
    for (int64_t x = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; x < total;
         x += (int64_t)gridDim.x * blockDim.x) {
        const int64_t b = x / MF_BLK;
        const int at = (int)(x - b * MF_BLK);
        const int w = 2 * (at / 128) + (at & 1), l = (at >> 1) & 63;
        double vf = 0.0, vb = 0.0;
        if (w < MF_OPS) {
            const int kb = w / MF_RB, rb = w % MF_RB;
            const int out = 16 * rb + (l & 15), in = mf_state(4 * kb + (l >> 4));      // k slot <= 35
            if (out < MF_S) {
                vf = exp(t[b * (MF_S * MF_S) + mf_state(out) * MF_S + in]);
                vb = exp(t[b * (MF_S * MF_S) + in * MF_S + mf_state(out)]);
            } else if (out < MF_S + 4) {
                vf = vb = 1.0;
            }
        }
        fwd[x] = vf;
        bwd[x] = vb;
    }
}

__device__ __forceinline__ void mfma_load_block(const double *__restrict__ blk, int lane, double (&a)[2 * MF_PAIRS]) {
    const double2 *src = reinterpret_cast<const double2 *>(blk) + lane;
#pragma unroll
    for (int p = 0; p < MF_PAIRS; ++p) {
#if defined(HMM_DIAG_FEWER_PAIRS)                // timing only: 10 of the 14 operand pairs are loaded, the rest keep what they had
        if (p >= 10) break;
#endif
        const double2 v = src[p * 64];
        a[2 * p] = v.x;
        a[2 * p + 1] = v.y;
    }
}

// a lane's nine states of one per-state row (72 contiguous bytes)
__device__ __forceinline__ void mfma_load_row(const double *__restrict__ row9, double (&v)[MF_Q]) {
    const mf_pair *src = reinterpret_cast<const mf_pair *>(row9);
#pragma unroll
    for (int m = 0; m < 4; ++m) {
        const mf_pair pr = src[m];
        v[2 * m] = pr.x;
        v[2 * m + 1] = pr.y;
    }
    v[8] = row9[8];
}
__device__ __forceinline__ void mfma_store_row(double *__restrict__ row9, const double (&v)[MF_Q]) {
    mf_pair *dst = reinterpret_cast<mf_pair *>(row9);
#pragma unroll
    for (int m = 0; m < 4; ++m) dst[m] = mf_pair{v[2 * m], v[2 * m + 1]};
    row9[8] = v[8];
}

// The same rows, stored coalesced (round 4).  Lane (g, n) holds nine states of sample n: stored straight from the lanes a
// 16-byte store instruction touches three cache lines of each of the 16 samples' rows, five instructions 240 partial lines per
// step - two thirds of all the cache-line transactions of a sweep's wavefront, and what the sweeps of a large batch waited for
// (a timing-only build without these stores: alpha 4.5 -> 3.4 ms alone, 9.3 -> 6.5 side by side).  Staged through LDS as
// [sample][state], the wavefront writes the 16 rows as 288 consecutive 16-byte pieces: every line once.
// dst[k]: where piece 64 k + lane goes (its sample's row of gene 0); stage: 16 x 36 doubles of this wavefront.
constexpr int MF_STAGE_STORES = (16 * MF_S / 2 + 63) / 64;                  // 5
__device__ __forceinline__ void mfma_stage_targets(double *__restrict__ base /* [sample][gene][state] at the chromosome's first gene */,
                                                   int first_sample, int n_samples, int64_t sample_stride /* rows */, int lane,
                                                   double *(&dst)[MF_STAGE_STORES]) {
#pragma unroll
    for (int k = 0; k < MF_STAGE_STORES; ++k) {
        const int c = min(64 * k + lane, 16 * MF_S / 2 - 1);
        const int sample = min(first_sample + c / (MF_S / 2), n_samples - 1);
        dst[k] = base + (int64_t)sample * sample_stride * MF_S + 2 * (c % (MF_S / 2));
    }
}
__device__ __forceinline__ void mfma_store_rows_staged(double *__restrict__ stage, const double (&v)[MF_Q], int lane,
                                                       double *const (&dst)[MF_STAGE_STORES], int64_t gene) {
    double *w = stage + (lane & 15) * MF_S + 9 * (lane >> 4);
#pragma unroll
    for (int q = 0; q < MF_Q; ++q) w[q] = v[q];
    asm volatile("" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    asm volatile("" ::: "memory");
    const double2 *rd = reinterpret_cast<const double2 *>(stage);
#pragma unroll
    for (int k = 0; k < MF_STAGE_STORES; ++k)
        if (64 * k + lane < 16 * MF_S / 2) *reinterpret_cast<double2 *>(dst[k] + gene * MF_S) = rd[64 * k + lane];
    asm volatile("" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    asm volatile("" ::: "memory");
}

// D = M y for the three row blocks (third first: it carries Z, whose reciprocal the rest waits for)
__device__ __forceinline__ void mfma_matvec(const double (&a)[2 * MF_PAIRS], const double (&y)[MF_Q], mfma_d4 &d0, mfma_d4 &d1,
                                            mfma_d4 &d2) {
    d0 = d1 = d2 = mfma_d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int kb = 0; kb < MF_KB; ++kb) {
        d2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a[kb * 3 + 2], y[kb], d2, 0, 0, 0);
        d0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a[kb * 3 + 0], y[kb], d0, 0, 0, 0);
        d1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a[kb * 3 + 1], y[kb], d1, 0, 0, 0);
    }
}

__device__ __forceinline__ double mfma_own(const mfma_d4 &d0, const mfma_d4 &d1, const mfma_d4 &d2, int q) {
    return q < 4 ? d0[q] : q < 8 ? d1[q - 4] : d2[0];
}

// alpha sweep (forward_wave_kernel role 0, same stored quantities: x, 1/Z).
// NG (round 4): groups of 16 samples per wavefront; one load of the step's 27 operands serves all of them and the groups'
// products and vector tails are independent instruction streams.  Built to test whether the operand traffic (320 + 320
// streams of 14 KB per step at 256 samples) is what the chains of a large batch share: it is not - NG = 2 halves that
// traffic and the wavefront count and the 256-sample pass takes the same 16.3-16.7 ms (smaller batches lose: a step is
// twice as long).  GBRS_TUNING_HMM_MFMA_NG=2 selects it; parity-tested at 16-70 samples.
template <int NSET, int NG>
__global__ void __launch_bounds__(64)
alpha_mfma_kernel(int n_samples, RowMap rm, const ChromDesc *__restrict__ chroms,
                  const int32_t *__restrict__ order, const double *__restrict__ amat,
                  const double *__restrict__ eprob, const double *__restrict__ peprob,
                  const double *__restrict__ init_vec, double *__restrict__ xsum, double *__restrict__ invz) {
#if defined(HMM_DIAG_EXCL_MFMA)                  // diagnostic builds: the wavefront takes its SIMD's whole register file, nothing runs beside it
    asm volatile("v_accvgpr_write_b32 a255, 0" ::: "a255");
#endif
    constexpr int S = MF_S;
    int bx = blockIdx.x, by = blockIdx.y;
    if (!rm.place(bx, by)) return;
    const ChromDesc cd = chroms[order[by]];
    const int n = cd.n_genes;
    if (n <= 0) return;
    const int lane = threadIdx.x, g = lane >> 4;
    const int n_ord = min(n, cd.n_trans + 1) - 1;       // step i = o + 1 needs block o
    const int last_o = max(n_ord - 1, 0);
    const double *BLK = amat + cd.trans_off * (int64_t)MF_BLK;
    const double *pe_row[NG];                            // + i * S: my states of gene i
    double *x_row[NG], *iz[NG];
    double y[NG][MF_Q];
#pragma unroll
    for (int cg = 0; cg < NG; ++cg) {
        const int sample = min((bx * NG + cg) * 16 + (lane & 15), n_samples - 1);   // past the end: shadows the last sample
        const int64_t g0 = (int64_t)sample * rm.sample_stride + cd.gene_off * rm.gene_stride;
        pe_row[cg] = peprob + g0 * S + 9 * g;
        x_row[cg] = xsum + g0 * S + 9 * g;
        iz[cg] = invz + g0;
        double e0[MF_Q], iv[MF_Q], x0[MF_Q];
        mfma_load_row(eprob + g0 * S + 9 * g, e0);
        mfma_load_row(init_vec + 9 * g, iv);
#pragma unroll
        for (int q = 0; q < MF_Q; ++q) {
            y[cg][q] = exp(iv[q] + e0[q]);
            x0[q] = exp(iv[q]);                          // so that log(x) + e reproduces init + e
        }
        mfma_store_row(x_row[cg], x0);
    }
    __shared__ __attribute__((aligned(16))) double stage[NG][16 * MF_S];
    double *x_dst[NG][MF_STAGE_STORES];
#pragma unroll
    for (int cg = 0; cg < NG; ++cg)
        mfma_stage_targets(xsum + cd.gene_off * rm.gene_stride * S, (bx * NG + cg) * 16, n_samples, rm.sample_stride, lane, x_dst[cg]);
    double a[NSET][2 * MF_PAIRS], pe[NSET][NG][MF_Q];
#pragma unroll
    for (int u = 0; u < NSET; ++u) {
        const int ou = min(u, last_o);
        if (n_ord > 0) {
            mfma_load_block(BLK + (int64_t)HMM_BLK(ou) * MF_BLK, lane, a[u]);
#pragma unroll
            for (int cg = 0; cg < NG; ++cg) mfma_load_row(pe_row[cg] + (int64_t)(ou + 1) * rm.gene_stride * S, pe[u][cg]);
        }
    }
    auto step = [&](int o, const double (&ac)[2 * MF_PAIRS], const double (&pc)[NG][MF_Q], double (&an)[2 * MF_PAIRS], double (&pn)[NG][MF_Q]) {
        mfma_d4 d0[NG], d1[NG], d2[NG];
#pragma unroll
        for (int cg = 0; cg < NG; ++cg) mfma_matvec(ac, y[cg], d0[cg], d1[cg], d2[cg]);
        double pe_now[NG][MF_Q];
#pragma unroll
        for (int cg = 0; cg < NG; ++cg)
#pragma unroll
            for (int q = 0; q < MF_Q; ++q) pe_now[cg][q] = pc[cg][q];
        // refill the set the previous step released (clamped index: past the end it re-reads the last block)
        const int of = min(o + NSET - 1, last_o);
#if !defined(HMM_ABL_M_LOADS)                    // timing-only builds (HMM_ABL_M_*): what a step of the sweep waits for
#if defined(HMM_ABL_M_QUARTER)                   // ... the operand block of every fourth step only (what four wavefronts sharing
        if ((o & 3) == 0)                        //     their operands through LDS would fetch)
#endif
        mfma_load_block(BLK + (int64_t)HMM_BLK(of) * MF_BLK, lane, an);
#pragma unroll
        for (int cg = 0; cg < NG; ++cg) mfma_load_row(pe_row[cg] + (int64_t)(of + 1) * rm.gene_stride * S, pn[cg]);
#else
        (void)an; (void)pn; (void)of;
#endif
#pragma unroll
        for (int cg = 0; cg < NG; ++cg) {
#if defined(HMM_ABL_M_RECIP)
            const double inv_z = __hiloint2double(0x7FE00000 - __double2hiint(d2[cg][1]), 0);
#else
            const double inv_z = fast_recip_pos(d2[cg][1]);        // Z of the previous vector
#endif
#if !defined(HMM_ABL_M_STORES)
            if (g == 0) iz[cg][o * rm.gene_stride] = inv_z;
#endif
            double x[MF_Q];
#pragma unroll
            for (int q = 0; q < MF_Q; ++q) {
                x[q] = mfma_own(d0[cg], d1[cg], d2[cg], q) * inv_z + TINY;
                y[cg][q] = x[q] * pe_now[cg][q];
            }
#if !defined(HMM_ABL_M_STORES)
            mfma_store_rows_staged(stage[cg], x, lane, x_dst[cg], (o + 1) * rm.gene_stride);
#endif
        }
    };
    int o = 0;
    for (; o + NSET <= n_ord; o += NSET) {
#pragma unroll
        for (int u = 0; u < NSET; ++u) step(o + u, a[u], pe[u], a[(u + NSET - 1) % NSET], pe[(u + NSET - 1) % NSET]);
    }
#pragma unroll
    for (int u = 0; u < NSET - 1; ++u)
        if (o + u < n_ord) step(o + u, a[u], pe[u], a[(u + NSET - 1) % NSET], pe[(u + NSET - 1) % NSET]);
    // Z of the last vector: my nine states, then the four lane groups of my sample
#pragma unroll
    for (int cg = 0; cg < NG; ++cg) {
        double z = 0.0;
#pragma unroll
        for (int q = 0; q < MF_Q; ++q) z += y[cg][q];
        z += __shfl_xor(z, 16, 64);
        z += __shfl_xor(z, 32, 64);
        if (g == 0) iz[cg][n_ord * rm.gene_stride] = fast_recip_pos(z);
    }
}

// free-running backward sweep (backward_wave_kernel's quantities, rescaled on every step: bscale is general)
template <int NSET, int NG>
__global__ void __launch_bounds__(64)
backward_mfma_kernel(int n_samples, RowMap rm, const ChromDesc *__restrict__ chroms,
                     const int32_t *__restrict__ order, const double *__restrict__ amat_t,
                     const double *__restrict__ peprob, double *__restrict__ bhat, double *__restrict__ bscale) {
#if defined(HMM_DIAG_EXCL_MFMA)                  // diagnostic builds: the wavefront takes its SIMD's whole register file, nothing runs beside it
    asm volatile("v_accvgpr_write_b32 a255, 0" ::: "a255");
#endif
    constexpr int S = MF_S;
    int bx = blockIdx.x, by = blockIdx.y;
    if (!rm.place(bx, by)) return;
    const ChromDesc cd = chroms[order[by]];
    const int n = cd.n_genes;
    if (n <= 0) return;
    const int lane = threadIdx.x, g = lane >> 4;
    const double *BLK = amat_t + cd.trans_off * (int64_t)MF_BLK;
    const double *pe_row[NG];
    double *b_row[NG], *bs[NG];
    double w[NG][MF_Q];                               // pe_{i+1} * bt_{i+1}
#pragma unroll
    for (int cg = 0; cg < NG; ++cg) {
        const int sample = min((bx * NG + cg) * 16 + (lane & 15), n_samples - 1);
        const int64_t g0 = (int64_t)sample * rm.sample_stride + cd.gene_off * rm.gene_stride;
        pe_row[cg] = peprob + g0 * S + 9 * g;
        b_row[cg] = bhat + g0 * S + 9 * g;
        bs[cg] = bscale + g0;
        double one[MF_Q];
#pragma unroll
        for (int q = 0; q < MF_Q; ++q) one[q] = 1.0;
        mfma_load_row(pe_row[cg] + (int64_t)(n - 1) * rm.gene_stride * S, w[cg]);
        mfma_store_row(b_row[cg] + (int64_t)(n - 1) * rm.gene_stride * S, one);
        if (g == 0) bs[cg][(n - 1) * rm.gene_stride] = 1.0;
    }
    // order o <-> gene i = n-2-o, transition block i (the host checked n_trans >= n-1)
    const int n_ord = n - 1;
    const int last_o = max(n_ord - 1, 0);
    __shared__ __attribute__((aligned(16))) double stage[NG][16 * MF_S];
    double *b_dst[NG][MF_STAGE_STORES];
#pragma unroll
    for (int cg = 0; cg < NG; ++cg)
        mfma_stage_targets(bhat + cd.gene_off * rm.gene_stride * S, (bx * NG + cg) * 16, n_samples, rm.sample_stride, lane, b_dst[cg]);
    double a[NSET][2 * MF_PAIRS], pe[NSET][NG][MF_Q];
#pragma unroll
    for (int u = 0; u < NSET; ++u) {
        const int i = n - 2 - min(u, last_o);
        if (n_ord > 0) {
            mfma_load_block(BLK + (int64_t)HMM_BLK(i) * MF_BLK, lane, a[u]);
#pragma unroll
            for (int cg = 0; cg < NG; ++cg) mfma_load_row(pe_row[cg] + (int64_t)i * rm.gene_stride * S, pe[u][cg]);
        }
    }
    auto step = [&](int o, const double (&ac)[2 * MF_PAIRS], const double (&pc)[NG][MF_Q], double (&an)[2 * MF_PAIRS], double (&pn)[NG][MF_Q]) {
        const int i = n - 2 - o;
        mfma_d4 d0[NG], d1[NG], d2[NG];
#pragma unroll
        for (int cg = 0; cg < NG; ++cg) mfma_matvec(ac, w[cg], d0[cg], d1[cg], d2[cg]);
        double pe_now[NG][MF_Q];
#pragma unroll
        for (int cg = 0; cg < NG; ++cg)
#pragma unroll
            for (int q = 0; q < MF_Q; ++q) pe_now[cg][q] = pc[cg][q];
        const int in = n - 2 - min(o + NSET - 1, last_o);
#if defined(HMM_ABL_M_QUARTER)
        if ((o & 3) == 0)
#endif
        mfma_load_block(BLK + (int64_t)HMM_BLK(in) * MF_BLK, lane, an);
#pragma unroll
        for (int cg = 0; cg < NG; ++cg) mfma_load_row(pe_row[cg] + (int64_t)in * rm.gene_stride * S, pn[cg]);
#pragma unroll
        for (int cg = 0; cg < NG; ++cg) {
            const double r = fast_recip_pos(d2[cg][1]);
            if (g == 0) bs[cg][i * rm.gene_stride] = r;
            double bh[MF_Q];
#pragma unroll
            for (int q = 0; q < MF_Q; ++q) {
                bh[q] = mfma_own(d0[cg], d1[cg], d2[cg], q) * r;
                w[cg][q] = bh[q] * pe_now[cg][q];
            }
            mfma_store_rows_staged(stage[cg], bh, lane, b_dst[cg], i * rm.gene_stride);
        }
    };
    int o = 0;
    for (; o + NSET <= n_ord; o += NSET) {
#pragma unroll
        for (int u = 0; u < NSET; ++u) step(o + u, a[u], pe[u], a[(u + NSET - 1) % NSET], pe[(u + NSET - 1) % NSET]);
    }
#pragma unroll
    for (int u = 0; u < NSET - 1; ++u)
        if (o + u < n_ord) step(o + u, a[u], pe[u], a[(u + NSET - 1) % NSET], pe[(u + NSET - 1) % NSET]);
}

// ------------------------------------------------------------------------------------------
// Sample batches, 36 states: the delta (max-plus) chain with the samples on the lanes.  A workgroup of nine wavefronts
// carries 16 samples of one chromosome; each 16-lane row of a wavefront owns ONE target state (row r of wavefront w:
// state 4w + r) and its lanes are the 16 samples.  Every lane keeps its sample's 36 previous values in registers (LDS,
// [sample][state], b128 reads) and a row's 36 transition entries sit 16 to a register across the row's lanes and enter
// the additions through the DPP row broadcast of v_fmac_f64 (row_newbcast:k; of the double-precision arithmetic only
// v_fmac_f64 has a DPP form, v_add_f64 / v_max_f64 are VOP3-only, so the sum is entry * 1.0 + previous, one rounding
// of the exact sum like the addition; each lane uses each previous value once, so the accumulate-in-place costs no
// copy): 36 (add, max) pairs per lane and step, no copy of the table per sample, no broadcast of the vector, and 16-sample groups so that even a 64-sample
// batch spreads over 80 CUs.  (Tried before, 64 samples per workgroup: uniform global loads of the entries -
// s_load_dwordx16, ~0.5 us each on a miss: 21 us per step; the block staged in LDS and read with uniform addresses: a
// 16-byte broadcast read returns 1 KB to the register file, ~24 cycles: 6.8 us per step; the DPP form with four
// and twelve wavefronts per 64 samples: 6.4 and 4.3 us per step - one workgroup per CU, a third of the chip at 256
// samples, and every double-precision instruction waiting on the one before it.)  Adds and maxima only, in any order:
// the values are those of the other forms bit for bit.
// ------------------------------------------------------------------------------------------
// DL_T target states per lane (round 4 experiment).  A lane needs its sample's whole 36-vector in registers whatever it
// computes, so with one target per lane the 576 threads of a workgroup read 166 KB from LDS per step and nine wavefronts
// meet at the barrier; with three per lane it is 55 KB and three wavefronts (one 16-lane row = target states 3 r .. 3 r + 2),
// the same (add, max) pairs - and SLOWER: 4.03 against 3.15 ms alone, 9.5 against 8.2 beside the sweeps (256 samples): the
// kernel is bound by the dependent double-precision instructions of a wavefront, not by the LDS pipe.  1 stays.
#ifndef HMM_DL_TARGETS
#define HMM_DL_TARGETS 1
#endif
constexpr int DL_T = HMM_DL_TARGETS;
static_assert(MF_S % (4 * DL_T) == 0, "rows of DL_T targets, four rows per wavefront");
constexpr int DL_WAVES = MF_S / (4 * DL_T), DL_SAMPLES = 16, DL_STRIDE = MF_S + 2;   // 38 doubles per sample: b128 reads of 16 lanes on 64 distinct banks
constexpr int DL_GROUPS = (MF_S + 15) / 16;                                     // 36 entries per row and step, 3 registers
#ifndef HMM_DL_AHEAD
#define HMM_DL_AHEAD 4
#endif
constexpr int DL_AHEAD = HMM_DL_AHEAD;                                          // steps between a fetch and its use

// x + (lane N of my 16-lane row of t), in one instruction: v_fmac_f64 is the one double-precision operation with a DPP
// form (row_newbcast only), so the sum is written t * 1.0 + x - one rounding of the exact sum, the value v_add_f64 gives.
template <int N>
__device__ __forceinline__ double add_row_bcast(double x, double t, double one) {
    asm("v_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(x) : "v"(t), "v"(one), "n"(N));
    return x;
}

// max over k of dp[k] + T[j][k]: four from-states at a time (independent instructions next to each other: a dependent
// double-precision instruction waits several issue slots), four running maxima
template <int K>
__device__ __forceinline__ void dl_from(const double (&t)[DL_GROUPS], const double (&dp)[MF_S], double one, double (&m)[4]) {
    const double a0 = add_row_bcast<(K + 0) % 16>(dp[K + 0], t[(K + 0) / 16], one);
    const double a1 = add_row_bcast<(K + 1) % 16>(dp[K + 1], t[(K + 1) / 16], one);
    const double a2 = add_row_bcast<(K + 2) % 16>(dp[K + 2], t[(K + 2) / 16], one);
    const double a3 = add_row_bcast<(K + 3) % 16>(dp[K + 3], t[(K + 3) / 16], one);
    if constexpr (K == 0) {
        m[0] = a0; m[1] = a1; m[2] = a2; m[3] = a3;
    } else {
        m[0] = fmax(m[0], a0); m[1] = fmax(m[1], a1); m[2] = fmax(m[2], a2); m[3] = fmax(m[3], a3);
    }
    if constexpr (K + 4 < MF_S) dl_from<K + 4>(t, dp, one, m);
}

__global__ void __launch_bounds__(64 * DL_WAVES)
delta_lanes_kernel(int n_samples, RowMap rm /* rows of eprob, and the grid's placement */, RowMap drows /* rows of delta */, const ChromDesc *__restrict__ chroms,
                   const int32_t *__restrict__ order, const double *__restrict__ tprob, const double *__restrict__ eprob,
                   const double *__restrict__ init_vec, double *__restrict__ delta, int32_t *__restrict__ last_state,
                   int n_chrom /* of the handle: the launch may cover a group of them */) {
    constexpr int S = MF_S, BLK = S * S;
    __shared__ __attribute__((aligned(16))) double dbuf[2][DL_SAMPLES][DL_STRIDE];
    int bx = blockIdx.x, by = blockIdx.y;
    if (!rm.place(bx, by)) return;
    const int chrom = order[by];
    const ChromDesc cd = chroms[chrom];
    const int n = cd.n_genes;
    if (n <= 0) return;
    const int tid = threadIdx.x, c = tid & 15, j0 = (tid >> 4) * DL_T;   // sample slot, first of the row's DL_T target states
    const int sample_raw = bx * DL_SAMPLES + c;
    const int sample = min(sample_raw, n_samples - 1);           // slots past the end shadow the last sample (same values, same stores)
    const int64_t g0 = (int64_t)sample * rm.sample_stride + cd.gene_off * rm.gene_stride;
    const int n_ord = min(n, cd.n_trans + 1) - 1;               // step i = o + 1 needs block o
    const int last_o = max(n_ord - 1, 0);
    const double *e_at = eprob + g0 * S + j0;                   // + i * S + t: my states of gene i
    // The step's 16 new rows leave through LDS: the workgroup stores them as consecutive 16-byte pieces (every cache line
    // once) instead of each row's few bytes of every sample's row.
    constexpr int ROW_PIECES = S / 2, PIECES = DL_SAMPLES * ROW_PIECES, THREADS = 64 * DL_WAVES;
    constexpr int ST = (PIECES + THREADS - 1) / THREADS;
    double *d_out[ST];
    int st_off[ST];                                             // of the piece in a dbuf plane, in doubles; -1: none
#pragma unroll
    for (int q = 0; q < ST; ++q) {
        const int piece = tid + q * THREADS;
        const bool on = piece < PIECES;
        const int slot = on ? piece / ROW_PIECES : 0, pc = on ? piece % ROW_PIECES : 0;
        st_off[q] = on ? slot * DL_STRIDE + 2 * pc : -1;
        d_out[q] = delta + ((int64_t)min(bx * DL_SAMPLES + slot, n_samples - 1) * drows.sample_stride + cd.gene_off * drows.gene_stride) * S + 2 * pc;
    }
    // row j of a block is 36 consecutive entries; register g of slot c holds entry 16 g + c
    const double *TB = tprob + cd.trans_off * (int64_t)BLK + (int64_t)j0 * S + c;
    const int third = c < S - 32 ? 32 : S - 1 - c;
    // the entries and the emissions of a step are fetched DL_AHEAD steps before it: a step is shorter than a trip to HBM
    double t_ring[DL_AHEAD][DL_T][DL_GROUPS], e_ring[DL_AHEAD][DL_T];
    double one = 1.0;
    asm volatile("" : "+v"(one));                            // a register operand for the DPP form
    auto fetch_step = [&](int o, int slot) {                   // step i = o + 1; clamped: past the end it re-reads the last one
        const int oc = min(o, last_o);
        const double *src = TB + (int64_t)HMM_BLK(oc) * BLK;
#pragma unroll
        for (int t = 0; t < DL_T; ++t) {
            e_ring[slot][t] = e_at[(int64_t)(oc + 1) * rm.gene_stride * S + t];
            t_ring[slot][t][0] = src[t * S];
            t_ring[slot][t][1] = src[t * S + 16];
            t_ring[slot][t][2] = src[t * S + third];            // entries 32..35 in slots 0..3; the other slots re-read entry 35, unused
        }
    };
    {
#pragma unroll
        for (int t = 0; t < DL_T; ++t) dbuf[0][c][j0 + t] = init_vec[j0 + t] + e_at[t];
        if (n_ord > 0) {
#pragma unroll
            for (int u = 0; u < DL_AHEAD; ++u) fetch_step(u, u);
        }
    }
    __syncthreads();
    int cur = 0;
    auto store_rows = [&](int gene) {                           // the rows in dbuf[cur], all threads
#pragma unroll
        for (int q = 0; q < ST; ++q)
            if (st_off[q] >= 0)
                *reinterpret_cast<double2 *>(d_out[q] + (int64_t)gene * drows.gene_stride * S) = *reinterpret_cast<const double2 *>(&dbuf[cur][0][0] + st_off[q]);
    };
    store_rows(0);
    auto step = [&](int o, int u) {                            // u: a constant after unrolling (the ring is in registers)
        double dp[S];
        const double2 *src = reinterpret_cast<const double2 *>(&dbuf[cur][c][0]);
#pragma unroll
        for (int k2 = 0; k2 < S / 2; ++k2) {
            const double2 pr = src[k2];
            dp[2 * k2] = pr.x;
            dp[2 * k2 + 1] = pr.y;
        }
        double dn[DL_T];
#pragma unroll
        for (int t = 0; t < DL_T; ++t) {
            double m[4];
            dl_from<0>(t_ring[u][t], dp, one, m);
            dn[t] = fmax(fmax(m[0], m[1]), fmax(m[2], m[3])) + e_ring[u][t];
        }
        fetch_step(o + DL_AHEAD, u);                            // after the last use of the slot: its registers are free again
#pragma unroll
        for (int t = 0; t < DL_T; ++t) dbuf[cur ^ 1][c][j0 + t] = dn[t];
        cur ^= 1;
        __syncthreads();
        // (this buffer is written again two steps on, behind the next barrier: the reads of store_rows are done by then)
        store_rows(o + 1);
    };
    int ob = 0;
    for (; ob + DL_AHEAD <= n_ord; ob += DL_AHEAD) {           // whole groups: no branch between a fetch and its use
#pragma unroll
        for (int u = 0; u < DL_AHEAD; ++u) step(ob + u, u);
    }
#pragma unroll
    for (int u = 0; u < DL_AHEAD - 1; ++u)
        if (ob + u < n_ord) step(ob + u, u);                    // block-uniform
    if (tid < DL_SAMPLES && sample_raw < n_samples) {           // sid = argmax delta[:, n-1] (first max)
        double bv = dbuf[cur][c][0];
        int bk = 0;
        for (int s2 = 1; s2 < S; ++s2) {
            const double v = dbuf[cur][c][s2];
            if (v > bv) { bv = v; bk = s2; }
        }
        last_state[(int64_t)sample_raw * n_chrom + chrom] = bk;
    }
}

// log C_i of backward_wave_kernel's comment for one (chromosome, sample): a suffix sum over genes
__global__ void __launch_bounds__(256)
beta_corr_kernel(int64_t genes_per_sample, const ChromDesc *__restrict__ chroms,
                 const double *__restrict__ invz, const double *__restrict__ bscale,
                 double *__restrict__ bcorr) {
    __shared__ double part[256];
    const ChromDesc cd = chroms[blockIdx.y];
    const int n = cd.n_genes;
    if (n <= 0) return;
    const int64_t g0 = (int64_t)blockIdx.x * genes_per_sample + cd.gene_off;
    const int t = threadIdx.x;
    const int seg = (n + 255) / 256;
    const int lo = min(n, t * seg), hi = min(n, lo + seg);
    auto term = [&](int i) { return log(invz[g0 + i]) - (i <= n - 2 ? log(bscale[g0 + i]) : 0.0); };
    double sum = 0.0;
    for (int i = hi - 1; i >= lo; --i) sum += term(i);
    part[t] = sum;
    __syncthreads();
    if (t == 0) {                               // exclusive suffix sums of the 256 segment totals
        double run = 0.0;
        for (int x = 255; x >= 0; --x) {
            const double v = part[x];
            part[x] = run;
            run += v;
        }
    }
    __syncthreads();
    double run = part[t];
    for (int i = hi - 1; i >= lo; --i) {
        run += term(i);
        bcorr[g0 + i] = run;
    }
}

// Posterior gamma = ahat*bhat / sum_j(ahat*bhat), ahat = x*pe/Z recomputed here (gbrs_utils.py:558-560): the
// only per-state array gbrs reconstruct saves, part of every run.  One thread per (sample, gene, state)
// element and POST_GROUPS groups of OUT_ROWS rows per workgroup: a pure stream (four arrays in, one
// out), so every thread puts all its 4*POST_GROUPS loads in flight before the first barrier
// (one group per workgroup ran at 2.2 TB/s, latency-bound).
constexpr int POST_GROUPS = 4;
__global__ void __launch_bounds__(1024)
posterior_kernel(int S, int OUT_ROWS, int64_t n_rows, const double *__restrict__ xsum,
                 const double *__restrict__ peprob, const double *__restrict__ invz,
                 const double *__restrict__ bhat, double *__restrict__ gamma,
                 int64_t gene_begin, int64_t range_len, int64_t genes_per_sample) {
    // n_rows rows are walked: row v is gene gene_begin + v % range_len of sample v / range_len (the whole handle:
    // gene_begin 0, range_len = genes_per_sample; a chromosome group of the pipelined batch pass: its gene range)
    extern __shared__ double lds[];               // g[POST_GROUPS][OUT_ROWS * S], norm[POST_GROUPS * OUT_ROWS]
    const int per = OUT_ROWS * S;
    double *l_g = lds, *l_norm = lds + POST_GROUPS * per;
    const int64_t r0 = (int64_t)blockIdx.x * OUT_ROWS * POST_GROUPS;
    const int t = threadIdx.x;
    const int row = t / S;
    auto actual = [&](int64_t v) { return (v / range_len) * genes_per_sample + gene_begin + v % range_len; };
    double ah[POST_GROUPS], bh[POST_GROUPS];
    bool live[POST_GROUPS];
#pragma unroll
    for (int q = 0; q < POST_GROUPS; ++q) {
        const int64_t rq = r0 + (int64_t)q * OUT_ROWS;
        const int nr = (int)max((int64_t)0, min((int64_t)OUT_ROWS, n_rows - rq));
        live[q] = t < nr * S;
        ah[q] = bh[q] = 0.0;
        if (live[q]) {
            const int64_t ar = actual(rq + row), o = ar * S + (t - row * S);
            ah[q] = (xsum[o] * peprob[o]) * invz[ar];            // alpha-hat = y / Z, y = x * pe as in the sweep
            bh[q] = bhat[o];
        }
    }
#pragma unroll
    for (int q = 0; q < POST_GROUPS; ++q)
        if (live[q]) l_g[q * per + t] = ah[q] * bh[q];
    __syncthreads();
    for (int rr = t; rr < POST_GROUPS * OUT_ROWS; rr += blockDim.x) {
        if (r0 + rr >= n_rows) continue;
        const double *g = l_g + (rr / OUT_ROWS) * per + (rr % OUT_ROWS) * S;
        double norm = 0.0;
        for (int s = 0; s < S; ++s) norm += g[s];               // sequential over states, as ndarray.sum(axis=0)
        l_norm[rr] = norm;
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < POST_GROUPS; ++q)
        if (live[q]) gamma[actual(r0 + (int64_t)q * OUT_ROWS + row) * S + (t - row * S)] = ah[q] * bh[q] / l_norm[q * OUT_ROWS + row];
}

// The reference's log-domain intermediates, one thread per (sample, gene, state); made on the first
// gbrs_hmm_get() that asks for one:
//   parts & 1  alpha = log(x) + e - log(Z), scaler = -log(Z)                        (gbrs_utils.py:515-524)
//   parts & 4  beta = log(bhat) [+ log C_i of the free-running backward sweep]      (gbrs_utils.py:542-549)
__global__ void __launch_bounds__(1024)
hmm_outputs_kernel(int S, int OUT_ROWS /* (sample, gene) rows per workgroup */, int64_t n_rows, int parts,
                   const double *__restrict__ eprob, const double *__restrict__ xsum,
                   const double *__restrict__ invz, const double *__restrict__ bhat,
                   const double *__restrict__ bcorr /* nullable */, double *__restrict__ alpha,
                   double *__restrict__ scaler, double *__restrict__ beta) {
    const int64_t r0 = (int64_t)blockIdx.x * OUT_ROWS;
    const int nr = (int)min((int64_t)OUT_ROWS, n_rows - r0);
    const int t = threadIdx.x;
    const bool live = t < nr * S;
    const int64_t o = r0 * S + t;
    const int row = t / S;
    if (parts & 1) {
        if (t < nr) scaler[r0 + t] = log(invz[r0 + t]);              // -log(Z)
        if (live) alpha[o] = (log(xsum[o]) + eprob[o]) + log(invz[r0 + row]);
    }
    if ((parts & 4) && live) {
        const double lb = log(bhat[o]);
        beta[o] = bcorr ? lb + bcorr[r0 + row] : lb;
    }
}

// Backtrace (gbrs_utils.py:587-597): states[m] = argmax delta[:, n-1], states[t] = bp[t][states[t+1]]
// for t = m-1 .. 0 (m = min(n, n_trans)).  A chain of 3,000 dependent lookups, cut into chunks of
// BT_B rows that are walked in parallel: backtrace_maps_kernel walks every chunk from each of the S
// possible entry states (lane = entry state) and records where it leaves the chunk;
// backtrace_write_kernel composes the exit maps of the chunks above its own (a few dozen lookups
// in LDS) to find its true entry state, walks its chunk once more and writes the path.
constexpr int BT_B = 64;

__global__ void __launch_bounds__(64)
backtrace_maps_kernel(int S, int64_t bp_per_sample, int64_t chunks_per_sample,
                      const ChromDesc *__restrict__ chroms, const uint16_t *__restrict__ bp,
                      uint16_t *__restrict__ exit_map) {
    extern __shared__ uint16_t stage[];        // BT_B * S
    const ChromDesc cd = chroms[blockIdx.y];
    const int sample = blockIdx.z;
    const int m = min(cd.n_genes, cd.n_trans);
    const int lo = blockIdx.x * BT_B;
    if (lo >= m) return;
    const int hi = min(m, lo + BT_B);
    const uint16_t *BP = bp + ((int64_t)sample * bp_per_sample + cd.bp_off + lo) * S;
    for (int x = threadIdx.x; x < (hi - lo) * S; x += 64) stage[x] = BP[x];
    __syncthreads();
    for (int s = threadIdx.x; s < S; s += 64) {
        int sid = s;
        for (int t = hi - lo - 1; t >= 0; --t) sid = stage[t * S + sid];
        exit_map[((int64_t)sample * chunks_per_sample + cd.chunk_off + blockIdx.x) * S + s] = (uint16_t)sid;
    }
}

__global__ void __launch_bounds__(64)
backtrace_write_kernel(int S, int64_t genes_per_sample, int64_t bp_per_sample, int64_t states_per_sample,
                       int64_t chunks_per_sample, int n_chrom, const ChromDesc *__restrict__ chroms,
                       const uint16_t *__restrict__ bp, const uint16_t *__restrict__ exit_map,
                       const int32_t *__restrict__ last_state, int32_t *__restrict__ states,
                       int32_t *__restrict__ calls, int chrom_base /* the launch covers chromosomes chrom_base + blockIdx.y */) {
    extern __shared__ uint16_t stage[];        // max(BT_B, chunks above) * S maps, then BT_B path entries
    const int chrom = chrom_base + (int)blockIdx.y;
    const ChromDesc cd = chroms[chrom];
    const int sample = blockIdx.z;
    const int n = cd.n_genes;
    const int m = min(n, cd.n_trans);
    const int n_chunks = (m + BT_B - 1) / BT_B;
    const int c = blockIdx.x;
    if (c >= max(n_chunks, 1)) return;
    int32_t *ST = states + (int64_t)sample * states_per_sample + cd.gene_off + chrom;
    int32_t *CL = calls + (int64_t)sample * genes_per_sample + cd.gene_off;
    const int last = last_state[(int64_t)sample * n_chrom + chrom];
    if (c == 0) {
        for (int i = m + threadIdx.x; i < n; i += 64) CL[i] = -1;
        if (threadIdx.x == 0) ST[m] = last;
    }
    if (m <= 0) return;
    const int lo = c * BT_B, hi = min(m, lo + BT_B);
    // entry state of this chunk: the last state pushed through the chunks above, top down
    const int above = n_chunks - 1 - c;
    const uint16_t *EX = exit_map + ((int64_t)sample * chunks_per_sample + cd.chunk_off + c + 1) * S;
    for (int x = threadIdx.x; x < above * S; x += 64) stage[x] = EX[x];
    __syncthreads();
    int sid = last;
    for (int cc = above - 1; cc >= 0; --cc) sid = stage[cc * S + sid];    // uniform: every thread walks
    __syncthreads();
    const uint16_t *BP = bp + ((int64_t)sample * bp_per_sample + cd.bp_off + lo) * S;
    uint16_t *rows = stage, *path = stage + BT_B * S;
    for (int x = threadIdx.x; x < (hi - lo) * S; x += 64) rows[x] = BP[x];
    __syncthreads();
    if (threadIdx.x == 0)
        for (int t = hi - lo - 1; t >= 0; --t) {
            sid = rows[t * S + sid];
            path[t] = (uint16_t)sid;
        }
    __syncthreads();
    for (int t = threadIdx.x; t < hi - lo; t += 64) {
        const int v = path[t];
        ST[lo + t] = v;
        CL[lo + t] = v;
    }
}

#include "hmm_blocked.inc"

// Backpointers for many samples (S = 36): the SAMPLES on the lanes.  viterbi_bp_kernel gives a sample's 36 targets a
// lane each, so every lane re-reads the sample's delta row from memory (36 loads per lane and sample, two or three
// distinct addresses per wavefront): 13 G loads at 256 samples, bound by the address units (4.56 ms alone, and the
// posterior next to it runs at a third of its speed).  Here a lane holds its sample's delta row in registers (18 16-byte
// loads), the transition block is read from LDS at wave-uniform addresses (one broadcast b128 read per two entries), and
// a (target, source) pair is four vector instructions: add, compare, maximum, select of the index.  Four targets make one
// 8-byte store of the sample's backpointer row.  First maximum, as np.argmax.
#ifndef HMM_BPL_MIN
#define HMM_BPL_MIN 32        // samples from which viterbi_bp_lanes_kernel replaces viterbi_bp_kernel
#endif
constexpr int BPL_WAVES = 4;
template <int SS>
__global__ void __launch_bounds__(64 * BPL_WAVES)
viterbi_bp_lanes_kernel(int n_samples, RowMap drows /* of delta */, int64_t bp_per_sample,
                        const ChromDesc *__restrict__ chroms, const double *__restrict__ tprob,
                        const double *__restrict__ delta, uint16_t *__restrict__ bp) {
    static_assert(SS % 4 == 0, "four backpointers per store");
    __shared__ __attribute__((aligned(16))) double tl[SS * SS];
    const ChromDesc cd = chroms[blockIdx.y];
    const int t = blockIdx.x;
    if (t >= min(cd.n_genes, cd.n_trans)) return;
    const double2 *T = reinterpret_cast<const double2 *>(tprob + (cd.trans_off + t) * (int64_t)SS * SS);
    for (int x = threadIdx.x; x < SS * SS / 2; x += blockDim.x) reinterpret_cast<double2 *>(tl)[x] = T[x];
    const int sample = blockIdx.z * blockDim.x + threadIdx.x;
    const bool act = sample < n_samples;
    const int64_t row = (int64_t)(act ? sample : n_samples - 1) * drows.sample_stride + (cd.gene_off + t) * drows.gene_stride;
    const double2 *d2 = reinterpret_cast<const double2 *>(delta + row * SS);
    double d[SS];
#pragma unroll
    for (int m = 0; m < SS / 2; ++m) {
        const double2 v = d2[m];
        d[2 * m] = v.x;
        d[2 * m + 1] = v.y;
    }
    __syncthreads();
    uint2 *out = reinterpret_cast<uint2 *>(bp + ((int64_t)(act ? sample : n_samples - 1) * bp_per_sample + cd.bp_off + t) * SS);
    uint64_t packed = 0;                          // the last four backpointers, 16 bits each
    // the row's 72 bytes are stored together at the end: written as they came (8 bytes every fourth state, hundreds of
    // cycles apart) the counters showed 3.3 GB of HBM writes for 0.74 GB of backpointers
    uint2 res[SS / 4];
#pragma unroll
    for (int q4 = 0; q4 < SS / 4; ++q4) {          // (nine register slots; the four states of a slot in a rolled loop)
#pragma unroll 1
        for (int j = 4 * q4; j < 4 * q4 + 4; ++j) {
            const double2 *tr = reinterpret_cast<const double2 *>(tl + j * SS);
            double best = 0.0;
            uint32_t best_k = 0;
#pragma unroll
            for (int m = 0; m < SS / 2; ++m) {
                const double2 tv = tr[m];
                const double a = d[2 * m] + tv.x, c = d[2 * m + 1] + tv.y;
                if (m == 0) {
                    best = a;
                } else {
                    best_k = a > best ? 2 * m : best_k;
                    best = max_f64(best, a);
                }
                best_k = c > best ? 2 * m + 1 : best_k;
                best = max_f64(best, c);
            }
            packed = (packed >> 16) | ((uint64_t)best_k << 48);
        }
        res[q4] = make_uint2((uint32_t)packed, (uint32_t)(packed >> 32));
    }
    if (act) {
#pragma unroll
        for (int q = 0; q < SS / 4; ++q) out[q] = res[q];
    }
}

// ---- post-processing of the posteriors (gbrs_utils.py:612-697 interpolate, :863-938 export) ---

// Linear interpolation of the rows of y (S x n points at ascending x) onto xq, operation order of
// scipy interp1d(kind='linear'):  slope = (y_hi - y_lo) / (x_hi - x_lo);  y = slope * (x - x_lo) + y_lo
__global__ void interpolate_kernel(int S, int n, const double *__restrict__ xs, const double *__restrict__ y,
                                   int m, const double *__restrict__ xq, double *__restrict__ out) {
    const int64_t id = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= (int64_t)S * m) return;
    const int s = (int)(id / m), g = (int)(id - (int64_t)s * m);
    const double x = xq[g];
    int lo = 0, hi = n;                        // searchsorted(xs, x, side='left')
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (xs[mid] < x) lo = mid + 1; else hi = mid;
    }
    const int idx = min(max(lo, 1), n - 1);
    const double x_lo = xs[idx - 1], x_hi = xs[idx];
    const double y_lo = y[(int64_t)s * n + idx - 1], y_hi = y[(int64_t)s * n + idx];
    const double slope = (y_hi - y_lo) / (x_hi - x_lo);
    out[(int64_t)s * m + g] = slope * (x - x_lo) + y_lo;
}

// 36 -> 8 dosage: out[r][h] = sum_g gprob[r][g] * 0.5 * (#h in genotype g)  (convmat, :896-901, :927)
__global__ void dosage_kernel(int H, int S, int64_t rows, const double *__restrict__ gprob,
                              double *__restrict__ out) {
    const int64_t id = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= rows * H) return;
    const int64_t r = id / H;
    const int h = (int)(id - r * H);
    const double *p = gprob + r * S;
    double acc = 0.0;
    int g = 0;
    for (int a = 0; a < H; ++a)
        for (int b = a; b < H; ++b, ++g) {
            const double w = 0.5 * ((a == h) + (b == h));
            acc += p[g] * w;
        }
    out[id] = acc;
}

}  // namespace gbrs

using namespace gbrs;

struct gbrs_hmm {
    int device = 0;
    hipStream_t stream = nullptr, stream_b = nullptr, stream_c = nullptr;   // see hmm_launch
    hipEvent_t ev[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
    hipEvent_t ev_fork = nullptr, ev_b = nullptr, ev_c1 = nullptr, ev_c = nullptr;
    hipEvent_t ev_ops[2] = {nullptr, nullptr};   // blocked scan: the alpha / backward operator kernels are done
    int H = 0, S = 0, n_chrom = 0, n_samples = 0;
    std::vector<ChromDesc> chroms;
    int64_t total_genes = 0, total_trans = 0, total_bp = 0, total_chunks = 0;
    int max_bp_rows = 0;                      // max over chromosomes of min(n_genes, n_trans)
    bool have_eprob = false, ran = false;
    bool pe_ready = false;                    // peprob = exp(eprob) is current (the emission kernels write both)
    bool free_backward = false;               // last run used the free-running backward sweep (beta needs bcorr)
    bool logs_ready = false;                  // alpha / beta / scaler of the last run have been made (hmm_make_logs)
    DevBuf<ChromDesc> d_chroms;
    DevBuf<int32_t> d_order;                  // chromosome indices, longest first
    DevBuf<double> tprob, pprob, pprob_t, init_vec;   // log T, exp(T), exp(T) transposed per block
    DevBuf<double> tprob_q;                   // S = 36 / 136: log T in the chain kernels' lane order (pprob, pprob_t too)
    bool quad = false;                        // tables are in lane order
    DevBuf<double> amat_f, amat_b;            // S = 36, large batches: the MFMA operands of exp(T) and its transpose (made on first use)
    DevBuf<double> expr, avecs, eprob, peprob, xsum, bhat, alpha, beta, gamma, delta, scaler, invz;
    double *expr_stage = nullptr;      // pinned host image of `expr` for small batches (gbrs_hmm_set_expression)
    size_t expr_stage_n = 0;
    DevBuf<double> bscale, bcorr;             // free-running backward: per-gene scale and log correction
    DevBuf<uint8_t> has_avec;
    DevBuf<uint16_t> bp, bt_exit;             // backpointers; per-chunk exit maps of the backtrace
    DevBuf<int32_t> last_state, states, calls;
    double t_emis = 0, t_fwd = 0, t_bwd = 0, t_bt = 0, t_run = 0;
    // blocked scan (hmm_blocked.inc; 36 states, up to HMM_BLOCKED_MAX samples): the chromosomes cut into blocks
    // Two block structures (round 4): the forward one (alpha, delta) may open every chromosome with a long block that is
    // chained DIRECTLY from the chromosome's start while the other blocks' operators are being built (no operator for it);
    // the backward one closes every chromosome with such a block.  [0] forward, [1] backward.
    int n_vb = 0, blk_samples = 0;            // n_vb: the larger of the two block counts (buffers are sized by it)
    int n_blk[2] = {0, 0}, n_head[2] = {0, 0};
    bool last_blocked = false;                // the last run's backward chains started from injected vectors
    RowMap delta_rows{0, 1};                  // how the last run laid `delta` out ([sample][gene], or [gene][sample] for the large batches)
    bool last_delta_spec = false;             // the last run's delta came from the rank-convergence scheme: delta_apply_kernel is due
    DevBuf<BlockRange> d_ranges[2];
    DevBuf<int32_t> d_first_block[2];         // blocks of chromosome c: first_block[c] .. first_block[c+1]
    DevBuf<int32_t> d_vorder[2];              // block indices, the directly chained ones first (n_head of them)
    DevBuf<ChromDesc> d_vfwd, d_vbwd;         // the blocks as descriptors of the forward / delta and of the backward chains
    std::vector<BlockRange> h_ranges0;        // host copy of d_ranges[0]
    DevBuf<double> dspec_c;                   // rank-convergence delta: every block's constant against the block before it,
    DevBuf<int32_t> dspec_g, dspec_fail;      // the gene from which its stored values stand, per (sample, chromosome) failure flags
    hipStream_t stream_h[3] = {nullptr, nullptr, nullptr};   // the direct chains of alpha / backward / delta
    hipEvent_t ev_head[3] = {nullptr, nullptr, nullptr};
    // Pipelined batch pass (round 4, hmm_launch_groups): the chromosomes in two groups of consecutive chromosomes, each with
    // its own emission -> chains -> posterior / backtrace pipeline on its own three streams, so that the emission of the
    // second group and the posteriors / backtraces of whichever group is done run beside the chains that set the pass's length.
    bool emission_pending = false;            // gbrs_hmm_set_expression left the emission kernel to the next run
    double em_thr = 0.0, em_sigma = 0.0;
    int n_groups = 0;
    int grp_lo[2] = {0, 0}, grp_hi[2] = {0, 0}, grp_order_off[2] = {0, 0}, grp_max_bp[2] = {0, 0};
    DevBuf<int32_t> d_order_grp;              // per group: its chromosomes, longest first
    hipStream_t stream_g[3] = {nullptr, nullptr, nullptr};   // the second group's streams (the first uses stream / stream_b / stream_c)
    hipEvent_t gev_em[2] = {nullptr, nullptr}, gev_b[2] = {nullptr, nullptr}, gev_c[2] = {nullptr, nullptr},
               gev_done[2] = {nullptr, nullptr}, gev_start = nullptr;
    DevBuf<double> g_f, g_b, g_d, inj_f, inj_b, inj_d;   // block operators [sample][block][36][36], boundary vectors [sample][block][36]
    DevBuf<int32_t> e_f, e_b;                 // power-of-two exponents of the operators' columns
};

namespace {

int hmm_alloc_samples(gbrs_hmm *h, int n_samples) {
    if (n_samples == h->n_samples && h->eprob.p) return GBRS_OK;
    const size_t gs = (size_t)h->total_genes * n_samples;
    GBRS_TRY(h->eprob.alloc(gs * h->S));
    GBRS_TRY(h->peprob.alloc(gs * h->S));
    GBRS_TRY(h->xsum.alloc(gs * h->S));
    GBRS_TRY(h->bhat.alloc(gs * h->S));
    GBRS_TRY(h->invz.alloc(gs));
    GBRS_TRY(h->bscale.alloc(gs));
    GBRS_TRY(h->gamma.alloc(gs * h->S));
    GBRS_TRY(h->delta.alloc(gs * h->S));
    h->alpha.release(); h->beta.release(); h->scaler.release(); h->bcorr.release();   // made on demand (hmm_make_logs)
    GBRS_TRY(h->bp.alloc(std::max<size_t>((size_t)h->total_bp * n_samples * h->S, 1)));
    GBRS_TRY(h->bt_exit.alloc(std::max<size_t>((size_t)h->total_chunks * n_samples * h->S, 1)));
    GBRS_TRY(h->last_state.alloc((size_t)h->n_chrom * n_samples));
    GBRS_TRY(h->states.alloc(((size_t)h->total_genes + h->n_chrom) * n_samples));
    GBRS_TRY(h->calls.alloc(gs));
    h->n_samples = n_samples;
    return GBRS_OK;
}

// posterior_kernel / hmm_outputs_kernel over every (sample, gene) row of the handle
void launch_posterior(gbrs_hmm *h, hipStream_t st) {
    const int S = h->S;
    const int64_t rows = h->total_genes * h->n_samples;
    const int out_rows = std::max(1, 1024 / S), per_block = out_rows * POST_GROUPS;
    const dim3 grid((unsigned)((rows + per_block - 1) / per_block)), block(((out_rows * S + 63) / 64) * 64);
    const size_t lds = (size_t)POST_GROUPS * (out_rows * S + out_rows) * sizeof(double);
    hipLaunchKernelGGL(posterior_kernel, grid, block, lds, st, S, out_rows, rows, h->xsum.p, h->peprob.p, h->invz.p,
                       h->bhat.p, h->gamma.p, (int64_t)0, h->total_genes, h->total_genes);
}

// the same over the genes [gene_begin, gene_begin + range_len) of every sample
void launch_posterior_range(gbrs_hmm *h, hipStream_t st, int64_t gene_begin, int64_t range_len) {
    const int S = h->S;
    const int64_t rows = range_len * h->n_samples;
    if (rows <= 0) return;
    const int out_rows = std::max(1, 1024 / S), per_block = out_rows * POST_GROUPS;
    const dim3 grid((unsigned)((rows + per_block - 1) / per_block)), block(((out_rows * S + 63) / 64) * 64);
    const size_t lds = (size_t)POST_GROUPS * (out_rows * S + out_rows) * sizeof(double);
    hipLaunchKernelGGL(posterior_kernel, grid, block, lds, st, S, out_rows, rows, h->xsum.p, h->peprob.p, h->invz.p,
                       h->bhat.p, h->gamma.p, gene_begin, range_len, h->total_genes);
}

void launch_logs(gbrs_hmm *h, int parts, hipStream_t st) {
    const int S = h->S;
    const int64_t rows = h->total_genes * h->n_samples;
    const int out_rows = std::max(1, 1024 / S);
    const dim3 grid((unsigned)((rows + out_rows - 1) / out_rows)), block(((out_rows * S + 63) / 64) * 64);
    hipLaunchKernelGGL(hmm_outputs_kernel, grid, block, 0, st, S, out_rows, rows, parts, h->eprob.p, h->xsum.p,
                       h->invz.p, h->bhat.p, h->free_backward ? h->bcorr.p : (const double *)nullptr,
                       h->alpha.p, h->scaler.p, h->beta.p);
}

// The log-domain intermediates of the reference (alpha, scaler, beta).  gbrs reconstruct saves none of
// them, so a run leaves them out (three of the seven per-state arrays the outputs pass would move, and the
// suffix sum of the backward sweep's scale constants); the first gbrs_hmm_get() that asks makes them
// for every sample of the last run.
int hmm_make_logs(gbrs_hmm *h) {
    if (h->logs_ready) return GBRS_OK;
    const size_t gs = (size_t)h->total_genes * h->n_samples;
    if (gs == 0) { h->logs_ready = true; return GBRS_OK; }
    if (!h->alpha.p) GBRS_TRY(h->alpha.alloc(gs * h->S));
    if (!h->beta.p) GBRS_TRY(h->beta.alloc(gs * h->S));
    if (!h->scaler.p) GBRS_TRY(h->scaler.alloc(gs));
    if (h->free_backward) {
        if (!h->bcorr.p) GBRS_TRY(h->bcorr.alloc(gs));
        if (h->last_blocked)            // the blocks' backward chains become one free-running chain again (hmm_blocked.inc)
            hipLaunchKernelGGL(blocked_bscale_fix_kernel, dim3(h->n_blk[1], h->n_samples), dim3(64), 0, h->stream, h->total_genes,
                               h->n_vb, h->d_ranges[1].p, h->peprob.p, h->bhat.p, h->inj_b.p, h->bscale.p);
        hipLaunchKernelGGL(beta_corr_kernel, dim3(h->n_samples, h->n_chrom), dim3(256), 0, h->stream, h->total_genes,
                           h->d_chroms.p, h->invz.p, h->bscale.p, h->bcorr.p);
    }
    launch_logs(h, 1 | 4, h->stream);
    GBRS_HIP_CHECK(hipGetLastError());
    GBRS_HIP_CHECK(hipStreamSynchronize(h->stream));
    h->logs_ready = true;
    return GBRS_OK;
}

#ifndef HMM_NSET
#define HMM_NSET 3        // register sets of the one-sample-per-wave recursions
#endif
#ifndef HMM_HOIST_A
#define HMM_HOIST_A 18      // LDS reads (double2) hoisted per batch in the one-sample alpha / delta recursions
#endif
#ifndef HMM_HOIST_D
#define HMM_HOIST_D 18
#endif
#ifndef HMM_HOIST_BLK
#define HMM_HOIST_BLK 6     // the same in the blocked scan's chain instances
#endif
#ifndef HMM_SB
#define HMM_SB 2          // samples per wave in large batches (n_samples >= HMM_BATCH_MIN)
#endif
#ifndef HMM_BATCH_MIN
#define HMM_BATCH_MIN 24  // below this every sample gets waves of its own (measured: 16 samples 3.4 vs 3.6 ms, 32 samples 5.1 vs 4.8 ms)
#endif
#ifndef HMM_NSET_B
#define HMM_NSET_B 3      // register sets of the HMM_SB-samples-per-wave recursions
#endif
#ifndef HMM_MFMA_MIN
#define HMM_MFMA_MIN 64   // 36 states, at least this many samples: alpha and backward sweeps of 16 samples per wave on MFMA (see the kernels' comment)
#endif
#ifndef HMM_DLANES_MIN
#define HMM_DLANES_MIN 64 // 36 states, at least this many samples: delta chain with the samples on the lanes (measured with the MFMA sweeps beside it:
                          // 8-32 samples the one-state-per-lane kernels win, 64 a wash, 128: 10.3 vs 12.1 ms, 256: 19.4 vs 22.1 ms)
#endif
#ifndef HMM_NSET_M
#define HMM_NSET_M 3      // register sets (transition blocks in flight) of the MFMA sweeps
#endif
#ifndef HMM_NSET_M2
#define HMM_NSET_M2 3     // the same with two sample groups per wavefront
#endif
#ifndef HMM_MFMA_NG2_MIN
#define HMM_MFMA_NG2_MIN (1 << 30)   // samples from which a wavefront of the MFMA sweeps carries two groups of 16: never by
#endif                               // default - measured (round 4): 256 samples 16.3-16.7 ms either way, 128: 9.2 -> 12.0, 64: 6.9 -> 9.3

#ifndef HMM_BLOCKED_MAX
#define HMM_BLOCKED_MAX 4     // 36 states, at most this many samples: the blocked scan (the sum-product operators cost 36 columns per block and sample;
                              // round 4, Viterbi values by rank convergence: 0.63 / 1.01 / 1.51 / 1.85 ms at 1 / 2 / 3 / 4 samples against 1.9-2.0 on the
                              // chains; 5 samples 2.25 against 2.0)
#endif
#ifndef HMM_DELTA_AFTER_OPS
#define HMM_DELTA_AFTER_OPS 0     // measured: 0.653 against 0.630 ms (the operators do not get faster without the delta chains beside them)
#endif
#ifndef HMM_DELTA_INTERLEAVED
#define HMM_DELTA_INTERLEAVED 0     // delta as [gene][sample] for the large batches: parity-green, no gain (15.72 against 15.73 ms), off
#endif
#ifndef HMM_BP_AFTER_SWEEPS
#define HMM_BP_AFTER_SWEEPS 0
#endif
#ifndef HMM_XCD_SPAN
#define HMM_XCD_SPAN 2        // XCDs a chromosome's sample groups are spread over under GBRS_TUNING_HMM_XCD (1, 2 or 4)
#endif
#ifndef HMM_XCD_GRIDS
#define HMM_XCD_GRIDS 0       // batch chain kernels on XCD-aware 1-D grids (GBRS_TUNING_HMM_XCD)
#endif
#ifndef HMM_DELTA_SPEC
#define HMM_DELTA_SPEC 1      // blocked scan: Viterbi values by rank convergence (one chain per block + fix-up) instead of max-plus block operators
#endif
#ifndef HMM_BLOCK_GENES
#define HMM_BLOCK_GENES 40    // genes per block aimed at (at most HMM_BLOCKS_MAX blocks per chromosome)
#endif
#ifndef HMM_BLOCKS_MAX
#define HMM_BLOCKS_MAX 64
#endif

// The block structures of the handle's chromosomes and the buffers of the blocked scan for n_samples samples.
#ifndef HMM_HEAD_PERCENT
#define HMM_HEAD_PERCENT 0    // share of a chromosome's genes that is chained directly while the operators of the rest are built.
                              // Round 4: built (GBRS_TUNING_HMM_HEAD=20..60), parity-green, SLOWER - beside the operator kernels,
                              // which keep every CU and the memory system busy, a directly chained block runs at ~1.7 us per step
                              // instead of 0.4 (wave priority, s_setprio 3, did not change that): 40k genes, one sample 1.06 ms without,
                              // 1.19 / 1.42 / 1.64 / 1.85 ms with 20 / 30 / 40 / 50 % (profiles/r04_hmm_experiments.txt)
#endif
int hmm_prepare_blocks(gbrs_hmm *h) {
    const int S = h->S;
    if (h->n_vb == 0) {
        int block_genes = HMM_BLOCK_GENES, blocks_max = HMM_BLOCKS_MAX, head_pct = HMM_HEAD_PERCENT;
        if (const char *env = std::getenv("GBRS_TUNING_HMM_BLOCK_GENES"); env && std::atoi(env) > 1) block_genes = std::atoi(env);
        if (const char *env = std::getenv("GBRS_TUNING_HMM_BLOCKS_MAX"); env && std::atoi(env) > 0) blocks_max = std::atoi(env);
        if (const char *env = std::getenv("GBRS_TUNING_HMM_HEAD"); env) head_pct = std::max(0, std::min(90, std::atoi(env)));
        std::vector<ChromDesc> vf, vb;
        for (int dir = 0; dir < 2; ++dir) {
            std::vector<BlockRange> ranges;
            std::vector<int32_t> first(h->n_chrom + 1, 0), heads, rest;
            for (int c = 0; c < h->n_chrom; ++c) {
                const ChromDesc &cd = h->chroms[c];
                const int n = cd.n_genes;
                // the directly chained part: the first genes (forward) / the last ones (backward); none on a chromosome
                // too short to leave three blocks beside it
                int direct = (int)((int64_t)n * head_pct / 100);
                if (n - direct < 3 * block_genes || direct < block_genes) direct = 0;
                const int rest_n = n - direct;
                const int nb = std::max(1, std::min(blocks_max - (direct ? 1 : 0), rest_n / block_genes));
                const int len = (rest_n + nb - 1) / nb;
                std::vector<std::pair<int, int>> cuts;     // [lo, hi) of the chromosome's blocks, ascending
                if (dir == 0 && direct) cuts.emplace_back(0, direct);
                const int r0 = dir == 0 ? direct : 0, r1 = dir == 0 ? n : rest_n;
                for (int lo = r0; lo < r1; lo += len) cuts.emplace_back(lo, std::min(r1, lo + len));
                if (dir == 1 && direct) cuts.emplace_back(rest_n, n);
                first[c] = (int32_t)ranges.size();
                for (size_t q = 0; q < cuts.size(); ++q) {
                    const int lo = cuts[q].first, hi = cuts[q].second;
                    const int v = (int)ranges.size();
                    const bool is_direct = direct && (dir == 0 ? q == 0 : q + 1 == cuts.size());
                    ranges.push_back(BlockRange{cd.gene_off, cd.trans_off, lo, hi, n, c, is_direct ? 1 : 0, cd.bp_off});
                    (is_direct ? heads : rest).push_back(v);
                    ChromDesc d = cd;               // chunk_off is not used by the chain kernels
                    if (dir == 0) {
                        if (lo > 0) {               // forward / delta: starts on the previous block's last gene
                            d.gene_off = cd.gene_off + lo - 1;
                            d.trans_off = cd.trans_off + lo - 1;
                            d.bp_off = cd.bp_off + lo - 1;      // the delta chain writes the backpointer rows of its steps
                            d.n_genes = hi - lo + 1;
                            d.inject = v;
                        } else {
                            d.n_genes = hi;
                        }
                        d.n_trans = d.n_genes;      // every step of the block has its transition block
                        // the chromosome's last block: one more block exactly when the chromosome has T[n-1] (its last
                        // backpointer row, forward_wave_kernel BP)
                        if (hi == n) d.n_trans = d.n_genes - 1 + (cd.n_trans >= n ? 1 : 0);
                        d.real_chrom = hi == n ? c : -1;
                        vf.push_back(d);
                    } else {
                        d.gene_off = cd.gene_off + lo;
                        d.trans_off = cd.trans_off + lo;
                        if (hi < n) {               // backward: ends on the next block's first gene
                            d.n_genes = hi - lo + 1;
                            d.inject = v;
                        } else {
                            d.n_genes = n - lo;
                        }
                        d.n_trans = d.n_genes;
                        d.real_chrom = -1;
                        vb.push_back(d);
                    }
                }
            }
            first[h->n_chrom] = (int32_t)ranges.size();
            const int VB = (int)ranges.size();
            std::vector<int32_t> order(heads);
            order.insert(order.end(), rest.begin(), rest.end());
            GBRS_TRY(h->d_ranges[dir].alloc(VB));
            GBRS_TRY(h->d_first_block[dir].alloc(first.size()));
            GBRS_TRY(h->d_vorder[dir].alloc(VB));
            GBRS_HIP_CHECK(hipMemcpy(h->d_ranges[dir].p, ranges.data(), VB * sizeof(BlockRange), hipMemcpyHostToDevice));
            GBRS_HIP_CHECK(hipMemcpy(h->d_first_block[dir].p, first.data(), first.size() * sizeof(int32_t), hipMemcpyHostToDevice));
            GBRS_HIP_CHECK(hipMemcpy(h->d_vorder[dir].p, order.data(), VB * sizeof(int32_t), hipMemcpyHostToDevice));
            h->n_blk[dir] = VB;
            h->n_head[dir] = (int)heads.size();
            if (dir == 0) h->h_ranges0 = ranges;
        }
        GBRS_TRY(h->d_vfwd.alloc(vf.size()));
        GBRS_TRY(h->d_vbwd.alloc(vb.size()));
        GBRS_HIP_CHECK(hipMemcpy(h->d_vfwd.p, vf.data(), vf.size() * sizeof(ChromDesc), hipMemcpyHostToDevice));
        GBRS_HIP_CHECK(hipMemcpy(h->d_vbwd.p, vb.data(), vb.size() * sizeof(ChromDesc), hipMemcpyHostToDevice));
        h->n_vb = std::max(h->n_blk[0], h->n_blk[1]);
        for (int k = 0; k < 3; ++k) {
            if (!h->stream_h[k]) GBRS_HIP_CHECK(hipStreamCreateWithFlags(&h->stream_h[k], hipStreamNonBlocking));
            if (!h->ev_head[k]) GBRS_HIP_CHECK(hipEventCreateWithFlags(&h->ev_head[k], hipEventDisableTiming));
        }
    }
    if (h->blk_samples < h->n_samples) {
        const size_t nb = (size_t)h->n_vb * h->n_samples;
        GBRS_TRY(h->g_f.alloc(nb * S * S)); GBRS_TRY(h->g_b.alloc(nb * S * S)); GBRS_TRY(h->g_d.alloc(nb * S * S));
        GBRS_TRY(h->e_f.alloc(nb * S)); GBRS_TRY(h->e_b.alloc(nb * S));
        GBRS_TRY(h->inj_f.alloc(nb * S)); GBRS_TRY(h->inj_b.alloc(nb * S)); GBRS_TRY(h->inj_d.alloc(nb * S));
        GBRS_TRY(h->dspec_c.alloc(nb)); GBRS_TRY(h->dspec_g.alloc(nb));
        GBRS_TRY(h->dspec_fail.alloc((size_t)h->n_samples * h->n_chrom));
        h->blk_samples = h->n_samples;
    }
    return GBRS_OK;
}

// The emission kernel gbrs_hmm_set_expression would have launched (all genes), when a deferred one has to be made up
// outside the pipelined pass (a get() of the emissions before any run, or a run that takes another path).
int hmm_flush_emission(gbrs_hmm *h) {
    if (!h->emission_pending) return GBRS_OK;
    const dim3 grid((unsigned)((h->total_genes + EM_GENES - 1) / EM_GENES), (unsigned)((h->n_samples + EM_BATCH_SPB - 1) / EM_BATCH_SPB));
    hipLaunchKernelGGL(emission_batch_kernel<EM_LANES>, grid, dim3(64), 0, h->stream, h->total_genes, h->n_samples,
                       EM_BATCH_SPB, h->expr.p, h->avecs.p, h->has_avec.p, h->init_vec.p, h->em_thr, h->em_sigma,
                       h->eprob.p, h->peprob.p, (int64_t)0, h->total_genes);
    GBRS_HIP_CHECK(hipGetLastError());
    GBRS_HIP_CHECK(hipStreamSynchronize(h->stream));
    h->emission_pending = false;
    h->pe_ready = true;
    return GBRS_OK;
}

#ifndef HMM_PIPE_MIN
// Samples from which a batch pass runs as two pipelined chromosome groups (emission of group 2 beside the sweeps of group 1).
// Parity-green and measured slower on one MI355X (256 samples 19.75-19.88 against 16.43-16.53 ms, 128: 13.14-13.18 against
// 9.21-9.31; profiles/r04_hmm_experiments.txt item 4), so never by default: GBRS_TUNING_HMM_PIPELINE=<samples> switches it on.
#define HMM_PIPE_MIN (1 << 30)
#endif
#ifndef HMM_PIPE_FIRST_PERCENT
#define HMM_PIPE_FIRST_PERCENT 30   // share of the genes in the group that goes first (the one with the longest chromosome)
#endif

int hmm_prepare_groups(gbrs_hmm *h) {
    if (h->n_groups) return GBRS_OK;
    const int nc = h->n_chrom;
    // two runs of consecutive chromosomes (a group's genes are one range of the gene axis); the cut that puts about
    // HMM_PIPE_FIRST_PERCENT of the genes beside the longest chromosome, and that group goes first
    int longest = 0;
    for (int c = 1; c < nc; ++c)
        if (h->chroms[c].n_genes > h->chroms[longest].n_genes) longest = c;
    int pct = HMM_PIPE_FIRST_PERCENT;
    if (const char *env = std::getenv("GBRS_TUNING_HMM_PIPE_FIRST"); env && std::atoi(env) > 0 && std::atoi(env) < 100) pct = std::atoi(env);
    const int64_t want = h->total_genes * pct / 100;
    int cut = 1;
    if (2 * (int64_t)h->chroms[longest].gene_off <= h->total_genes) {      // longest in the front half: first group = [0, cut)
        int64_t acc = 0;
        for (cut = 0; cut < nc - 1; ++cut) {
            acc += h->chroms[cut].n_genes;
            if (cut >= longest && acc >= want) { ++cut; break; }
        }
        cut = std::max(1, std::min(cut, nc - 1));
        h->grp_lo[0] = 0; h->grp_hi[0] = cut; h->grp_lo[1] = cut; h->grp_hi[1] = nc;
    } else {                                                               // first group = [cut, nc)
        int64_t acc = 0;
        for (cut = nc - 1; cut > 0; --cut) {
            acc += h->chroms[cut].n_genes;
            if (cut <= longest && acc >= want) break;
        }
        cut = std::max(1, std::min(cut, nc - 1));
        h->grp_lo[0] = cut; h->grp_hi[0] = nc; h->grp_lo[1] = 0; h->grp_hi[1] = cut;
    }
    std::vector<int32_t> order;
    for (int g = 0; g < 2; ++g) {
        std::vector<int32_t> mine;
        for (int c = h->grp_lo[g]; c < h->grp_hi[g]; ++c) mine.push_back(c);
        std::stable_sort(mine.begin(), mine.end(), [&](int a, int b) { return h->chroms[a].n_genes > h->chroms[b].n_genes; });
        h->grp_order_off[g] = (int)order.size();
        order.insert(order.end(), mine.begin(), mine.end());
        int mb = 0;
        for (int c : mine) mb = std::max(mb, std::min(h->chroms[c].n_genes, h->chroms[c].n_trans));
        h->grp_max_bp[g] = mb;
    }
    GBRS_TRY(h->d_order_grp.alloc(order.size()));
    GBRS_HIP_CHECK(hipMemcpy(h->d_order_grp.p, order.data(), order.size() * sizeof(int32_t), hipMemcpyHostToDevice));
    for (auto &st : h->stream_g) GBRS_HIP_CHECK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    for (hipEvent_t *e : {&h->gev_em[0], &h->gev_em[1], &h->gev_b[0], &h->gev_b[1], &h->gev_c[0], &h->gev_c[1], &h->gev_done[0], &h->gev_done[1]})
        GBRS_HIP_CHECK(hipEventCreateWithFlags(e, hipEventDisableTiming));
    GBRS_HIP_CHECK(hipEventCreate(&h->gev_start));
    h->n_groups = 2;
    return GBRS_OK;
}

// A large batch of 36-state samples as two pipelined chromosome groups (see gbrs_hmm above).  Same kernels, same
// arithmetic and stored quantities as the one-group pass of hmm_launch - only the launches are per group.
int hmm_launch_groups(gbrs_hmm *h) {
    constexpr int S = MF_S;
    GBRS_TRY(hmm_prepare_groups(h));
    h->delta_rows = RowMap{(int64_t)h->total_genes, 1};
    hipStream_t sa = h->stream;
    if (!h->amat_f.p) {
        GBRS_TRY(h->amat_f.alloc((size_t)h->total_trans * MF_BLK));
        GBRS_TRY(h->amat_b.alloc((size_t)h->total_trans * MF_BLK));
        hipLaunchKernelGGL(mfma_blocks_kernel, dim3(4096), dim3(256), 0, sa, h->total_trans, h->tprob.p, h->amat_f.p, h->amat_b.p);
    }
    h->logs_ready = false;
    h->free_backward = true;
    h->last_blocked = false;
    GBRS_HIP_CHECK(hipEventRecord(h->gev_start, sa));
    GBRS_HIP_CHECK(hipEventRecord(h->ev[1], sa));
    const int ns = h->n_samples;
    int bpl_min = HMM_BPL_MIN;
    if (const char *env = std::getenv("GBRS_TUNING_HMM_BPLANES"); env) bpl_min = std::atoi(env) > 0 ? std::atoi(env) : INT_MAX;
    for (int g = 0; g < 2; ++g) {
        hipStream_t s0 = g == 0 ? h->stream : h->stream_g[0], s1 = g == 0 ? h->stream_b : h->stream_g[1],
                    s2 = g == 0 ? h->stream_c : h->stream_g[2];
        const int c_lo = h->grp_lo[g], c_hi = h->grp_hi[g], ncg = c_hi - c_lo;
        if (ncg <= 0) continue;
        const int64_t gene_lo = h->chroms[c_lo].gene_off;
        const int64_t gene_hi = h->chroms[c_hi - 1].gene_off + h->chroms[c_hi - 1].n_genes;
        const int32_t *ord = h->d_order_grp.p + h->grp_order_off[g];
        if (g > 0) GBRS_HIP_CHECK(hipStreamWaitEvent(s0, h->gev_start, 0));
        // emission of the group's genes
        {
            const dim3 grid((unsigned)((gene_hi - gene_lo + EM_GENES - 1) / EM_GENES), (unsigned)((ns + EM_BATCH_SPB - 1) / EM_BATCH_SPB));
            hipLaunchKernelGGL(emission_batch_kernel<EM_LANES>, grid, dim3(64), 0, s0, h->total_genes, ns, EM_BATCH_SPB, h->expr.p,
                               h->avecs.p, h->has_avec.p, h->init_vec.p, h->em_thr, h->em_sigma, h->eprob.p, h->peprob.p,
                               gene_lo, gene_hi);
            GBRS_HIP_CHECK(hipEventRecord(h->gev_em[g], s0));
        }
        GBRS_HIP_CHECK(hipStreamWaitEvent(s1, h->gev_em[g], 0));
        GBRS_HIP_CHECK(hipStreamWaitEvent(s2, h->gev_em[g], 0));
        const dim3 mfma_grid((ns + 15) / 16, ncg);
        hipLaunchKernelGGL((alpha_mfma_kernel<HMM_NSET_M, 1>), mfma_grid, dim3(64), 0, s0, ns, RowMap{h->total_genes, 1}, h->d_chroms.p, ord,
                           h->amat_f.p, h->eprob.p, h->peprob.p, h->init_vec.p, h->xsum.p, h->invz.p);
        if (g == 0) GBRS_HIP_CHECK(hipEventRecord(h->ev[2], s0));
        hipLaunchKernelGGL((backward_mfma_kernel<HMM_NSET_M, 1>), mfma_grid, dim3(64), 0, s1, ns, RowMap{h->total_genes, 1}, h->d_chroms.p, ord,
                           h->amat_b.p, h->peprob.p, h->bhat.p, h->bscale.p);
        GBRS_HIP_CHECK(hipEventRecord(h->gev_b[g], s1));
        hipLaunchKernelGGL(delta_lanes_kernel, dim3((ns + DL_SAMPLES - 1) / DL_SAMPLES, ncg), dim3(64 * DL_WAVES), 0, s2, ns,
                           RowMap{h->total_genes, 1}, RowMap{h->total_genes, 1}, h->d_chroms.p, ord, h->tprob.p, h->eprob.p, h->init_vec.p, h->delta.p, h->last_state.p,
                           h->n_chrom);
        if (h->grp_max_bp[g] > 0) {
            if (ns >= bpl_min) {
                const int per_wg = std::min(64 * BPL_WAVES, ((ns + 63) / 64) * 64);
                hipLaunchKernelGGL((viterbi_bp_lanes_kernel<MF_S>), dim3(h->grp_max_bp[g], ncg, (ns + per_wg - 1) / per_wg), dim3(per_wg), 0,
                                   s2, ns, RowMap{h->total_genes, 1}, h->total_bp, h->d_chroms.p + c_lo, h->tprob.p, h->delta.p, h->bp.p);
            } else {
                hipLaunchKernelGGL(viterbi_bp_kernel, dim3(h->grp_max_bp[g], ncg), dim3(256), (size_t)S * (S + 1) * sizeof(double), s2,
                                   S, ns, h->total_genes, h->total_bp, h->d_chroms.p + c_lo, h->tprob.p, h->delta.p, h->bp.p);
            }
        }
        if (g == 0) GBRS_HIP_CHECK(hipEventRecord(h->ev_c1, s2));
        {
            const int bt_chunks = std::max(1, (h->grp_max_bp[g] + BT_B - 1) / BT_B);
            const dim3 bt_grid(bt_chunks, ncg, ns);
            hipLaunchKernelGGL(backtrace_maps_kernel, bt_grid, dim3(64), (size_t)BT_B * S * sizeof(uint16_t), s2, S, h->total_bp,
                               h->total_chunks, h->d_chroms.p + c_lo, h->bp.p, h->bt_exit.p);
            hipLaunchKernelGGL(backtrace_write_kernel, bt_grid, dim3(64), ((size_t)std::max(BT_B, bt_chunks) * S + BT_B) * sizeof(uint16_t),
                               s2, S, h->total_genes, h->total_bp, h->total_genes + h->n_chrom, h->total_chunks, h->n_chrom,
                               h->d_chroms.p, h->bp.p, h->bt_exit.p, h->last_state.p, h->states.p, h->calls.p, c_lo);
        }
        GBRS_HIP_CHECK(hipEventRecord(h->gev_c[g], s2));
        GBRS_HIP_CHECK(hipStreamWaitEvent(s0, h->gev_b[g], 0));
        launch_posterior_range(h, s0, gene_lo, gene_hi - gene_lo);
        if (g == 0) GBRS_HIP_CHECK(hipEventRecord(h->ev[3], s0));
        GBRS_HIP_CHECK(hipStreamWaitEvent(s0, h->gev_c[g], 0));
        GBRS_HIP_CHECK(hipEventRecord(h->gev_done[g], s0));
    }
    GBRS_HIP_CHECK(hipStreamWaitEvent(sa, h->gev_done[0], 0));
    GBRS_HIP_CHECK(hipStreamWaitEvent(sa, h->gev_done[1], 0));
    GBRS_HIP_CHECK(hipEventRecord(h->ev[4], sa));
    GBRS_HIP_CHECK(hipGetLastError());
    GBRS_HIP_CHECK(hipStreamSynchronize(sa));
    h->emission_pending = false;
    h->pe_ready = true;
    float ms = 0.f, ms2 = 0.f;
    // (first group's alpha / delta + backpointers; first group's backward + posterior; whole pass incl. both emissions)
    if (hipEventElapsedTime(&ms, h->ev[1], h->ev[2]) == hipSuccess && hipEventElapsedTime(&ms2, h->ev[1], h->ev_c1) == hipSuccess)
        h->t_fwd = std::max(ms, ms2);
    if (hipEventElapsedTime(&ms, h->ev[1], h->ev[3]) == hipSuccess) h->t_bwd = ms;
    h->t_bt = 0.0;
    if (hipEventElapsedTime(&ms, h->ev[1], h->ev[4]) == hipSuccess) h->t_run = ms;
    return GBRS_OK;
}

// SS_WAVE > 0: the single-wave chain kernels for that (even, <= 64) state count; otherwise KMAX / MAXT /
// EXACT select the quad chains (EXACT, S = 4*KMAX > 64) or the generic multi-wave kernels.
template <int SS_WAVE, int KMAX, int MAXT, bool EXACT>
int hmm_launch(gbrs_hmm *h) {
    if (h->emission_pending) {
        // a large 36-state batch whose emission gbrs_hmm_set_expression left to this run: two pipelined chromosome groups,
        // provided the batch takes the kernels that pass is made of (MFMA sweeps, samples-on-lanes delta chain)
        if constexpr (SS_WAVE == MF_S) {
            int mfma_min = HMM_MFMA_MIN, dl_min = HMM_DLANES_MIN;
            if (const char *env = std::getenv("GBRS_TUNING_HMM_MFMA"); env) mfma_min = std::atoi(env) > 0 ? std::atoi(env) : INT_MAX;
            if (const char *env = std::getenv("GBRS_TUNING_HMM_DLANES"); env) dl_min = std::atoi(env) > 0 ? std::atoi(env) : INT_MAX;
            const char *ng = std::getenv("GBRS_TUNING_HMM_MFMA_NG");
            if (h->n_samples >= mfma_min && h->n_samples >= dl_min && !(ng && std::atoi(ng) == 2)) return hmm_launch_groups(h);
        }
        GBRS_TRY(hmm_flush_emission(h));
    }
    const int S = h->S;
    const int threads = ((S * 4 + 63) / 64) * 64;
    const int64_t rows = h->total_genes * h->n_samples;
    const int bt_chunks = std::max(1, (h->max_bp_rows + BT_B - 1) / BT_B);
    const dim3 bt_grid(bt_chunks, h->n_chrom, h->n_samples);
    const size_t bt_maps_lds = (size_t)BT_B * S * sizeof(uint16_t);
    const size_t bt_write_lds = ((size_t)std::max(BT_B, bt_chunks) * S + BT_B) * sizeof(uint16_t);
    auto launch_backtrace = [&](hipStream_t st) {
        hipLaunchKernelGGL(backtrace_maps_kernel, bt_grid, dim3(64), bt_maps_lds, st, S, h->total_bp,
                           h->total_chunks, h->d_chroms.p, h->bp.p, h->bt_exit.p);
        hipLaunchKernelGGL(backtrace_write_kernel, bt_grid, dim3(64), bt_write_lds, st, S, h->total_genes,
                           h->total_bp, h->total_genes + h->n_chrom, h->total_chunks, h->n_chrom, h->d_chroms.p,
                           h->bp.p, h->bt_exit.p, h->last_state.p, h->states.p, h->calls.p, 0);
    };
    const dim3 unit_grid(h->n_chrom, h->n_samples);
    hipStream_t sa = h->stream, sb = h->stream_b, sc = h->stream_c;
    if (const char *env = std::getenv("GBRS_TUNING_HMM_SERIAL"); env && std::atoi(env)) sb = sc = sa;
    constexpr bool WAVE = SS_WAVE > 0;                // the single-wave recursions (tables in lane order)
    constexpr bool QUAD = !WAVE && EXACT && KMAX * 4 > 64 && KMAX % 2 == 0;   // S = 136: the quad chains (tables in lane order)
    h->logs_ready = false;
    h->free_backward = WAVE || QUAD;                  // those sweeps rescale on their own (beta_corr_kernel)
    GBRS_HIP_CHECK(hipEventRecord(h->ev[1], sa));
    if (!h->pe_ready) {                               // caller-supplied emissions (gbrs_hmm_set_eprob)
        hipLaunchKernelGGL(exp_emission_kernel, dim3((unsigned)((rows * S + 255) / 256)), dim3(256), 0, sa,
                           rows * S, h->eprob.p, h->peprob.p);
        h->pe_ready = true;
    }
    if constexpr (WAVE || QUAD) {
        // Three independent chains from here, each on its own stream (a sample's 40 chromosomes
        // occupy 40 CUs per chain):  A  alpha -> [join B] beta correction + outputs
        //                            B  free-running backward
        //                            C  delta -> backpointers -> backtrace
        std::function<void(hipStream_t)> launch_alpha, launch_back, launch_delta;
        if constexpr (WAVE) {
            constexpr int SS = SS_WAVE;
            // Few samples: one sample per wave (latency).  Many samples: HMM_SB samples share each wave's
            // transition registers (half the block loads per sample; measured best of 1-8 at 64 samples).
            const bool batched = h->n_samples >= HMM_BATCH_MIN;
            const dim3 wave_grid(batched ? (h->n_samples + HMM_SB - 1) / HMM_SB : h->n_samples, h->n_chrom);
            auto k_alpha = batched ? &forward_wave_kernel<SS, HMM_NSET_B, HMM_SB, 0, 18> : &forward_wave_kernel<SS, HMM_NSET, 1, 0, HMM_HOIST_A>;
            auto k_delta = batched ? &forward_wave_kernel<SS, HMM_NSET_B, HMM_SB, 1, 18> : &forward_wave_kernel<SS, HMM_NSET, 1, 1, HMM_HOIST_D>;
            auto k_back = batched ? &backward_wave_kernel<SS, HMM_NSET_B, HMM_SB> : &backward_wave_kernel<SS, HMM_NSET, 1>;
            // the blocked scan's chains (~40 steps each, ~1,000 wavefronts per kernel): two register sets, fewer hoisted reads
            auto kb_alpha = &forward_wave_kernel<SS, 2, 1, 0, HMM_HOIST_BLK>;
            auto kb_delta = &forward_wave_kernel<SS, 2, 1, 1, HMM_HOIST_BLK, true>;          // backpointers in the same pass over T
            auto kf_delta = &forward_wave_kernel<SS, HMM_NSET, 1, 1, HMM_HOIST_D, true>;       // the fallback chain behind the fix-up
            auto kb_back = &backward_wave_kernel<SS, 2, 1>;
            // GBRS_TUNING_HMM_MFMA = smallest batch that takes the MFMA sweeps (0: never) - the parity tests run them at 16
            int mfma_min = HMM_MFMA_MIN;
            if (const char *env = std::getenv("GBRS_TUNING_HMM_MFMA"); env) mfma_min = std::atoi(env) > 0 ? std::atoi(env) : INT_MAX;
            const bool mfma = SS == MF_S && h->n_samples >= mfma_min && h->total_trans > 0;
            // GBRS_TUNING_HMM_BLOCKED = largest batch that takes the blocked scan (0: never)
            int blocked_max = HMM_BLOCKED_MAX;
            if (const char *env = std::getenv("GBRS_TUNING_HMM_BLOCKED"); env) blocked_max = std::atoi(env);
            const bool blocked = SS == MF_S && !mfma && h->n_samples <= blocked_max && h->total_trans > 0;
            if (blocked) GBRS_TRY(hmm_prepare_blocks(h));
            h->last_blocked = blocked;
            h->last_delta_spec = false;
            if ((mfma || blocked) && !h->amat_f.p) {
                GBRS_TRY(h->amat_f.alloc((size_t)h->total_trans * MF_BLK));
                GBRS_TRY(h->amat_b.alloc((size_t)h->total_trans * MF_BLK));
                hipLaunchKernelGGL(mfma_blocks_kernel, dim3(4096), dim3(256), 0, sa, h->total_trans, h->tprob.p,
                                   h->amat_f.p, h->amat_b.p);
                GBRS_HIP_CHECK(hipEventRecord(h->ev[1], sa));        // one-off table work stays outside the run's timing
            }
            // sample groups of 16 per wavefront of the MFMA sweeps: 2 from HMM_MFMA_NG2_MIN samples on (GBRS_TUNING_HMM_MFMA_NG = 1 / 2 forces)
            int mfma_ng = h->n_samples >= HMM_MFMA_NG2_MIN ? 2 : 1;
            if (const char *env = std::getenv("GBRS_TUNING_HMM_MFMA_NG"); env && (std::atoi(env) == 1 || std::atoi(env) == 2)) mfma_ng = std::atoi(env);
            const dim3 mfma_grid((h->n_samples + 16 * mfma_ng - 1) / (16 * mfma_ng), h->n_chrom);
            // rows of the per-sample arrays as the batch kernels address them; GBRS_DIAG_HMM_INTERLEAVED=1 (diagnostic builds; timing only - the
            // other kernels keep [sample][gene], so the results are wrong): [gene][sample], a step's 16 rows contiguous
            RowMap chain_rows{h->total_genes, 1};
#if defined(GBRS_DIAG_BUILD)                     // never in the product library: the switch gives wrong results
            if (const char *env = std::getenv("GBRS_DIAG_HMM_INTERLEAVED"); env && std::atoi(env)) chain_rows = RowMap{1, h->n_samples};
#endif
            // GBRS_TUNING_HMM_XCD=1: the batch chain kernels on XCD-aware 1-D grids (RowMap::place)
            const int xcd_mask = [] { const char *env = std::getenv("GBRS_TUNING_HMM_XCD"); return env ? std::atoi(env) : HMM_XCD_GRIDS; }();   // 1: sweeps, 2: delta chain
            int xcd_span = HMM_XCD_SPAN;
            if (const char *env = std::getenv("GBRS_TUNING_HMM_XCD_SPAN"); env && (std::atoi(env) == 1 || std::atoi(env) == 2 || std::atoi(env) == 4)) xcd_span = std::atoi(env);
            auto xcd_grid = [&](unsigned groups, RowMap &rmap) {
                rmap.xcd_groups = (int32_t)groups;
                rmap.n_order = h->n_chrom;
                rmap.xcd_span = xcd_span;
                const unsigned sets = 8u / (unsigned)xcd_span, per = (groups + xcd_span - 1) / xcd_span;
                return dim3(8u * per * (unsigned)((h->n_chrom + sets - 1) / sets));
            };
            RowMap mfma_rows = chain_rows, dl_rows = chain_rows;
            const dim3 mfma_launch = (xcd_mask & 1) ? xcd_grid(mfma_grid.x, mfma_rows) : mfma_grid;
            const dim3 dl_grid((h->n_samples + DL_SAMPLES - 1) / DL_SAMPLES, h->n_chrom);
            const dim3 dl_launch = (xcd_mask & 2) ? xcd_grid(dl_grid.x, dl_rows) : dl_grid;
            // delta as [gene][sample] when both its writer and its reader are the samples-on-lanes kernels (a step's / a
            // gene's rows contiguous: the backpointer kernel reads one page per gene instead of one per lane); gbrs_hmm_get
            // copies a sample's rows out with a stride.  GBRS_TUNING_HMM_DELTA_ROWS=1 switches it on (measured: no gain).
            {
                int bplm = HMM_BPL_MIN;
                if (const char *env = std::getenv("GBRS_TUNING_HMM_BPLANES"); env) bplm = std::atoi(env) > 0 ? std::atoi(env) : INT_MAX;
                bool il = HMM_DELTA_INTERLEAVED != 0;
                if (const char *env = std::getenv("GBRS_TUNING_HMM_DELTA_ROWS"); env) il = std::atoi(env) != 0;
                const bool both = SS == MF_S && h->total_trans > 0 && h->n_samples >= bplm &&
                                  h->n_samples >= [] { int d = HMM_DLANES_MIN; if (const char *e = std::getenv("GBRS_TUNING_HMM_DLANES"); e) d = std::atoi(e) > 0 ? std::atoi(e) : INT_MAX; return d; }();
                h->delta_rows = il && both ? RowMap{1, h->n_samples} : RowMap{chain_rows.sample_stride, chain_rows.gene_stride};
            }
            launch_alpha = [=](hipStream_t st) {
                if (mfma) {
                    auto k = mfma_ng == 2 ? &alpha_mfma_kernel<HMM_NSET_M2, 2> : &alpha_mfma_kernel<HMM_NSET_M, 1>;
                    hipLaunchKernelGGL(k, mfma_launch, dim3(64), 0, st, h->n_samples, mfma_rows,
                                       h->d_chroms.p, h->d_order.p, h->amat_f.p, h->eprob.p, h->peprob.p, h->init_vec.p,
                                       h->xsum.p, h->invz.p);
                    return;
                }
                if (blocked) {
                    // the directly chained blocks on a stream of their own, beside the operators of the others; the combine
                    // starts from what those chains stored
                    auto chains = [&](hipStream_t q, int first, int count) {
                        if (count > 0)
                            hipLaunchKernelGGL(kb_alpha, dim3(h->n_samples, count), dim3(64), 0, q, h->n_samples, h->total_genes,
                                               h->d_vfwd.p, h->d_vorder[0].p + first, h->tprob_q.p, h->pprob.p, h->eprob.p, h->peprob.p,
                                               h->init_vec.p, h->xsum.p, h->invz.p, h->delta.p, h->last_state.p, h->inj_f.p, h->n_chrom,
                                               h->n_vb, (const int32_t *)nullptr, (uint16_t *)nullptr, (int64_t)0);
                    };
                    if (h->n_head[0]) {
                        (void)hipStreamWaitEvent(h->stream_h[0], h->ev_fork, 0);
                        chains(h->stream_h[0], 0, h->n_head[0]);
                        (void)hipEventRecord(h->ev_head[0], h->stream_h[0]);
                    }
                    hipLaunchKernelGGL((blockmat_mfma_kernel<0>), dim3(h->n_blk[0], h->n_samples), dim3(64), 0, st, h->total_genes,
                                       h->n_vb, h->d_ranges[0].p, h->amat_f.p, h->peprob.p, h->g_f.p, h->e_f.p);
                    (void)hipEventRecord(h->ev_ops[0], st);
                    if (h->n_head[0]) (void)hipStreamWaitEvent(st, h->ev_head[0], 0);
                    hipLaunchKernelGGL((combine_sumprod_kernel<0>), dim3(h->n_chrom, h->n_samples), dim3(64), 0, st, h->total_genes,
                                       h->n_vb, h->d_ranges[0].p, h->d_first_block[0].p, h->g_f.p, h->e_f.p, h->init_vec.p, h->eprob.p,
                                       h->peprob.p, h->xsum.p, h->inj_f.p);
                    chains(st, h->n_head[0], h->n_blk[0] - h->n_head[0]);
                    return;
                }
                hipLaunchKernelGGL(k_alpha, wave_grid, dim3(64), 0, st, h->n_samples, h->total_genes,
                                   h->d_chroms.p, h->d_order.p, h->tprob_q.p, h->pprob.p, h->eprob.p, h->peprob.p,
                                   h->init_vec.p, h->xsum.p, h->invz.p, h->delta.p, h->last_state.p, (const double *)nullptr,
                                   h->n_chrom, 0, (const int32_t *)nullptr, (uint16_t *)nullptr, (int64_t)0);
            };
            launch_back = [=](hipStream_t st) {
                if (mfma) {
                    auto k = mfma_ng == 2 ? &backward_mfma_kernel<HMM_NSET_M2, 2> : &backward_mfma_kernel<HMM_NSET_M, 1>;
                    hipLaunchKernelGGL(k, mfma_launch, dim3(64), 0, st, h->n_samples,
                                       mfma_rows, h->d_chroms.p, h->d_order.p, h->amat_b.p, h->peprob.p, h->bhat.p,
                                       h->bscale.p);
                    return;
                }
                if (blocked) {
                    auto chains = [&](hipStream_t q, int first, int count) {
                        if (count > 0)
                            hipLaunchKernelGGL(kb_back, dim3(h->n_samples, count), dim3(64), 0, q, h->n_samples, h->total_genes,
                                               h->d_vbwd.p, h->d_vorder[1].p + first, h->pprob_t.p, h->peprob.p, h->bhat.p, h->bscale.p,
                                               h->inj_b.p, h->n_vb);
                    };
                    if (h->n_head[1]) {
                        (void)hipStreamWaitEvent(h->stream_h[1], h->ev_fork, 0);
                        chains(h->stream_h[1], 0, h->n_head[1]);
                        (void)hipEventRecord(h->ev_head[1], h->stream_h[1]);
                    }
                    hipLaunchKernelGGL((blockmat_mfma_kernel<1>), dim3(h->n_blk[1], h->n_samples), dim3(64), 0, st, h->total_genes,
                                       h->n_vb, h->d_ranges[1].p, h->amat_b.p, h->peprob.p, h->g_b.p, h->e_b.p);
                    (void)hipEventRecord(h->ev_ops[1], st);
                    if (h->n_head[1]) (void)hipStreamWaitEvent(st, h->ev_head[1], 0);
                    hipLaunchKernelGGL((combine_sumprod_kernel<1>), dim3(h->n_chrom, h->n_samples), dim3(64), 0, st, h->total_genes,
                                       h->n_vb, h->d_ranges[1].p, h->d_first_block[1].p, h->g_b.p, h->e_b.p, h->init_vec.p, h->eprob.p,
                                       h->peprob.p, h->bhat.p, h->inj_b.p);
                    chains(st, h->n_head[1], h->n_blk[1] - h->n_head[1]);
                    return;
                }
                hipLaunchKernelGGL(k_back, wave_grid, dim3(64), 0, st, h->n_samples, h->total_genes,
                                   h->d_chroms.p, h->d_order.p, h->pprob_t.p, h->peprob.p, h->bhat.p, h->bscale.p,
                                   (const double *)nullptr, 0);
            };
            // GBRS_TUNING_HMM_DLANES = smallest batch that takes the samples-on-lanes delta chain (0: never)
            int dl_min = HMM_DLANES_MIN;
            if (const char *env = std::getenv("GBRS_TUNING_HMM_DLANES"); env) dl_min = std::atoi(env) > 0 ? std::atoi(env) : INT_MAX;
            const bool dlanes = SS == MF_S && h->n_samples >= dl_min && h->total_trans > 0;
            // GBRS_TUNING_HMM_DELTA_SPEC=0: the blocked scan's delta through max-plus block operators (round 3) instead of rank
            // convergence; GBRS_TUNING_HMM_DELTA_TOL=<absolute tolerance> (negative: no block ever converges - every chromosome
            // takes the fallback chain; the tests use it)
            bool delta_spec = blocked && HMM_DELTA_SPEC != 0;
            if (const char *env = std::getenv("GBRS_TUNING_HMM_DELTA_SPEC"); env) delta_spec = blocked && std::atoi(env) != 0;
            double delta_tol_abs = 1e-9, delta_tol_rel = 1e-13;
            if (const char *env = std::getenv("GBRS_TUNING_HMM_DELTA_TOL"); env) {
                delta_tol_abs = std::atof(env);
                if (delta_tol_abs < 0.0) delta_tol_rel = 0.0;
            }
            h->last_delta_spec = delta_spec;
            launch_delta = [=](hipStream_t st) {
                if (dlanes) {
                    hipLaunchKernelGGL(delta_lanes_kernel, dl_launch, dim3(64 * DL_WAVES), 0, st,
                                       h->n_samples, dl_rows, h->delta_rows, h->d_chroms.p, h->d_order.p, h->tprob.p, h->eprob.p,
                                       h->init_vec.p, h->delta.p, h->last_state.p, h->n_chrom);
                    return;
                }
                if (blocked) {
                    auto chains = [&](hipStream_t q, int first, int count) {
                        if (count > 0)
                            hipLaunchKernelGGL(kb_delta, dim3(h->n_samples, count), dim3(64), 0, q, h->n_samples, h->total_genes,
                                               h->d_vfwd.p, h->d_vorder[0].p + first, h->tprob_q.p, h->pprob.p, h->eprob.p, h->peprob.p,
                                               h->init_vec.p, h->xsum.p, h->invz.p, h->delta.p, h->last_state.p, h->inj_d.p, h->n_chrom,
                                               h->n_vb, (const int32_t *)nullptr, h->bp.p, h->total_bp);
                    };
                    if (delta_spec) {
                        // rank convergence instead of block operators (hmm_blocked.inc): guess -> chains in all blocks -> fix-up in
                        // all blocks -> the unblocked chain for the chromosomes whose flag a fix-up raised (idle otherwise)
                        const dim3 bgrid(h->n_blk[0], h->n_samples);
                        // GBRS_TUNING_HMM_DELTA_AFTER_OPS=1: the delta side (the short one) behind the two operator kernels instead
                        // of beside them - measured: the operators are no faster alone (backward side 0.556 against 0.563 ms) and
                        // the forward side gets longer (0.549 against 0.496): off.
                        if (const char *env = std::getenv("GBRS_TUNING_HMM_DELTA_AFTER_OPS"); env ? std::atoi(env) != 0 : HMM_DELTA_AFTER_OPS != 0) {
                            (void)hipStreamWaitEvent(st, h->ev_ops[0], 0);
                            (void)hipStreamWaitEvent(st, h->ev_ops[1], 0);
                        }
                        hipLaunchKernelGGL(delta_guess_kernel, bgrid, dim3(64), 0, st, h->total_genes, h->n_vb, h->n_chrom,
                                           h->d_ranges[0].p, h->eprob.p, h->inj_d.p, h->dspec_c.p, h->dspec_g.p, h->dspec_fail.p);
                        chains(st, 0, h->n_blk[0]);
                        hipLaunchKernelGGL(delta_fixup_kernel<SS>, bgrid, dim3(64), 0, st, h->total_genes, h->n_vb, h->n_chrom,
                                           h->d_ranges[0].p, h->tprob_q.p, h->eprob.p, h->delta.p, h->dspec_c.p, h->dspec_g.p,
                                           h->dspec_fail.p, delta_tol_abs, delta_tol_rel, h->bp.p, h->total_bp);
                        hipLaunchKernelGGL(kf_delta, wave_grid, dim3(64), 0, st, h->n_samples, h->total_genes,
                                           h->d_chroms.p, h->d_order.p, h->tprob_q.p, h->pprob.p, h->eprob.p, h->peprob.p,
                                           h->init_vec.p, h->xsum.p, h->invz.p, h->delta.p, h->last_state.p, (const double *)nullptr,
                                           h->n_chrom, 0, (const int32_t *)h->dspec_fail.p, h->bp.p, h->total_bp);
                        return;
                    }
                    if (h->n_head[0]) {
                        (void)hipStreamWaitEvent(h->stream_h[2], h->ev_fork, 0);
                        chains(h->stream_h[2], 0, h->n_head[0]);
                        (void)hipEventRecord(h->ev_head[2], h->stream_h[2]);
                    }
                    hipLaunchKernelGGL(blockmat_maxplus_kernel, dim3(h->n_blk[0], h->n_samples), dim3(64 * MP_WAVES), 0, st,
                                       h->total_genes, h->n_vb, h->d_ranges[0].p, h->tprob.p, h->eprob.p, h->g_d.p);
                    if (h->n_head[0]) (void)hipStreamWaitEvent(st, h->ev_head[2], 0);
                    hipLaunchKernelGGL(combine_maxplus_kernel, dim3(h->n_chrom, h->n_samples), dim3(64), 0, st, h->total_genes,
                                       h->n_vb, h->d_ranges[0].p, h->d_first_block[0].p, h->g_d.p, h->init_vec.p, h->eprob.p,
                                       h->delta.p, h->inj_d.p);
                    chains(st, h->n_head[0], h->n_blk[0] - h->n_head[0]);
                    return;
                }
                hipLaunchKernelGGL(k_delta, wave_grid, dim3(64), 0, st, h->n_samples, h->total_genes,
                                   h->d_chroms.p, h->d_order.p, h->tprob_q.p, h->pprob.p, h->eprob.p, h->peprob.p,
                                   h->init_vec.p, h->xsum.p, h->invz.p, h->delta.p, h->last_state.p, (const double *)nullptr,
                                   h->n_chrom, 0, (const int32_t *)nullptr, (uint16_t *)nullptr, (int64_t)0);
            };
        } else {
            const dim3 quad_grid(h->n_samples, h->n_chrom), quad_block(threads);
            launch_alpha = [=](hipStream_t st) {
                hipLaunchKernelGGL((group_chain_kernel<4, KMAX, 1, 1, 0>), quad_grid, quad_block, 0, st, h->total_genes,
                                   h->d_chroms.p, h->d_order.p, h->pprob.p, h->peprob.p, h->eprob.p, h->init_vec.p,
                                   h->xsum.p, h->invz.p, h->last_state.p);
            };
            launch_back = [=](hipStream_t st) {
                hipLaunchKernelGGL((group_chain_kernel<4, KMAX, 1, 1, 2>), quad_grid, quad_block, 0, st, h->total_genes,
                                   h->d_chroms.p, h->d_order.p, h->pprob_t.p, h->peprob.p, h->eprob.p, h->init_vec.p,
                                   h->bhat.p, h->bscale.p, h->last_state.p);
            };
            launch_delta = [=](hipStream_t st) {
                hipLaunchKernelGGL((group_chain_kernel<4, KMAX, 1, 1, 1>), quad_grid, quad_block, 0, st, h->total_genes,
                                   h->d_chroms.p, h->d_order.p, h->tprob_q.p, h->eprob.p, h->eprob.p, h->init_vec.p,
                                   h->delta.p, h->invz.p, h->last_state.p);
            };
        }
        const size_t bp_lds = (size_t)S * (S + 1) * sizeof(double);
        // GBRS_TUNING_HMM_BPLANES = smallest batch that takes the samples-on-lanes backpointer kernel (0: never)
        int bpl_min = HMM_BPL_MIN;
        if (const char *env = std::getenv("GBRS_TUNING_HMM_BPLANES"); env) bpl_min = std::atoi(env) > 0 ? std::atoi(env) : INT_MAX;
        GBRS_HIP_CHECK(hipEventRecord(h->ev_fork, sa));
        GBRS_HIP_CHECK(hipStreamWaitEvent(sb, h->ev_fork, 0));
        GBRS_HIP_CHECK(hipStreamWaitEvent(sc, h->ev_fork, 0));
        // GBRS_DIAG_HMM_SKIP=<letters of a, b, c, p, v> (diagnostic builds, -DGBRS_DIAG_BUILD; timing only: wrong results): leave the alpha / backward / delta chain, the
        // posterior, the backpointers + backtrace out of the pass
#if defined(GBRS_DIAG_BUILD)                         // never in the product library: the switch gives wrong results
        const char *skip = std::getenv("GBRS_DIAG_HMM_SKIP");
#else
        const char *skip = nullptr;
#endif
        auto skipped = [&](char c) { return skip && std::strchr(skip, c) != nullptr; };
        if (!skipped('a')) launch_alpha(sa);
        GBRS_HIP_CHECK(hipEventRecord(h->ev[2], sa));
        if (const char *env = std::getenv("GBRS_TUNING_HMM_BACK_AFTER"); env && std::atoi(env))
            GBRS_HIP_CHECK(hipStreamWaitEvent(sb, h->ev[2], 0));
        if (!skipped('b')) launch_back(sb);
        GBRS_HIP_CHECK(hipEventRecord(h->ev_b, sb));
        if (!skipped('c')) launch_delta(sc);
        // GBRS_TUNING_HMM_BP_AFTER=1: the backpointer kernel (one sample per lane: a cache line and a page per lane and load)
        // behind the sweeps instead of beside them
        if (const char *env = std::getenv("GBRS_TUNING_HMM_BP_AFTER"); env ? std::atoi(env) != 0 : HMM_BP_AFTER_SWEEPS != 0) {
            GBRS_HIP_CHECK(hipStreamWaitEvent(sc, h->ev[2], 0));
            GBRS_HIP_CHECK(hipStreamWaitEvent(sc, h->ev_b, 0));
        }
        if (h->max_bp_rows > 0 && !h->last_blocked && !skipped('v')) {     // the blocked scan's delta chains write the backpointers themselves
            if constexpr (QUAD)
                hipLaunchKernelGGL((viterbi_bp_quad_kernel<KMAX>), dim3(h->max_bp_rows, h->n_chrom), dim3(threads), 0,
                                   sc, h->n_samples, h->total_genes, h->total_bp, h->d_chroms.p, h->tprob_q.p,
                                   h->delta.p, h->bp.p);
            else if (WAVE && h->n_samples <= 4) {
                if constexpr (WAVE)
                    hipLaunchKernelGGL((viterbi_bp_wave_kernel<(WAVE ? SS_WAVE : 2)>), dim3((h->max_bp_rows + BPW_ROWS - 1) / BPW_ROWS, h->n_chrom),
                                       dim3(64), 0, sc, h->n_samples, h->total_genes, h->total_bp, h->d_chroms.p,
                                       h->tprob_q.p, h->delta.p, h->bp.p);
            } else if (WAVE && SS_WAVE == MF_S && h->n_samples >= bpl_min) {
                const int per_wg = std::min(64 * BPL_WAVES, ((h->n_samples + 63) / 64) * 64);
                hipLaunchKernelGGL((viterbi_bp_lanes_kernel<MF_S>), dim3(h->max_bp_rows, h->n_chrom, (h->n_samples + per_wg - 1) / per_wg),
                                   dim3(per_wg), 0, sc, h->n_samples, h->delta_rows, h->total_bp, h->d_chroms.p, h->tprob.p,
                                   h->delta.p, h->bp.p);
            } else
                hipLaunchKernelGGL(viterbi_bp_kernel, dim3(h->max_bp_rows, h->n_chrom), dim3(256),
                                   bp_lds, sc, S, h->n_samples, h->total_genes,
                                   h->total_bp, h->d_chroms.p, h->tprob.p, h->delta.p, h->bp.p);
        }
        GBRS_HIP_CHECK(hipEventRecord(h->ev_c1, sc));
        if (!skipped('v')) launch_backtrace(sc);
        if (h->last_delta_spec)                       // behind everything that reads the blocks' vectors as the fix-up left them
            hipLaunchKernelGGL(delta_apply_kernel, dim3(h->n_blk[0], h->n_samples), dim3(64), 0, sc, h->total_genes, h->n_vb,
                               h->n_chrom, h->d_ranges[0].p, h->d_first_block[0].p, h->dspec_c.p, h->dspec_g.p,
                               h->dspec_fail.p, h->delta.p);
        GBRS_HIP_CHECK(hipEventRecord(h->ev_c, sc));
        GBRS_HIP_CHECK(hipStreamWaitEvent(sa, h->ev_b, 0));
        if (!skipped('p')) launch_posterior(h, sa);   // the posterior is scale free: no beta correction needed
        GBRS_HIP_CHECK(hipEventRecord(h->ev[3], sa));
        GBRS_HIP_CHECK(hipStreamWaitEvent(sa, h->ev_c, 0));
        GBRS_HIP_CHECK(hipEventRecord(h->ev[4], sa));
        GBRS_HIP_CHECK(hipGetLastError());
        GBRS_HIP_CHECK(hipStreamSynchronize(sa));
        // the chains overlap: forward = the longer of alpha and delta + backpointers, backward =
        // backward sweep + correction + outputs (from the fork), backtrace = the backtrace kernel
        float ms = 0.f, ms2 = 0.f;
        if (hipEventElapsedTime(&ms, h->ev[1], h->ev[2]) == hipSuccess &&
            hipEventElapsedTime(&ms2, h->ev[1], h->ev_c1) == hipSuccess)
            h->t_fwd = std::max(ms, ms2);
        if (hipEventElapsedTime(&ms, h->ev_fork, h->ev[3]) == hipSuccess) h->t_bwd = ms;
        if (hipEventElapsedTime(&ms, h->ev_c1, h->ev_c) == hipSuccess) h->t_bt = ms;
        if (hipEventElapsedTime(&ms, h->ev[1], h->ev[4]) == hipSuccess) h->t_run = ms;
        return GBRS_OK;
    } else {
        hipLaunchKernelGGL((forward_viterbi_kernel<KMAX, MAXT, EXACT>), dim3(h->n_chrom, h->n_samples, 2), dim3(threads),
                           2 * S * sizeof(double), sa, S, h->total_genes, h->total_bp, h->d_chroms.p,
                           h->tprob.p, h->pprob.p, h->eprob.p, h->peprob.p, h->init_vec.p, h->xsum.p,
                           h->invz.p, h->delta.p, h->bp.p, h->last_state.p);
        GBRS_HIP_CHECK(hipEventRecord(h->ev[2], sa));
        hipLaunchKernelGGL((backward_kernel<KMAX, MAXT, EXACT>), unit_grid, dim3(threads),
                           2 * S * sizeof(double), sa, S, h->total_genes, h->d_chroms.p, h->pprob_t.p,
                           h->peprob.p, h->invz.p, h->bhat.p);
        launch_posterior(h, sa);
        GBRS_HIP_CHECK(hipEventRecord(h->ev[3], sa));
        launch_backtrace(sa);
        GBRS_HIP_CHECK(hipEventRecord(h->ev[4], sa));
        GBRS_HIP_CHECK(hipGetLastError());
        GBRS_HIP_CHECK(hipStreamSynchronize(sa));
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, h->ev[1], h->ev[2]) == hipSuccess) h->t_fwd = ms;
        if (hipEventElapsedTime(&ms, h->ev[2], h->ev[3]) == hipSuccess) h->t_bwd = ms;
        if (hipEventElapsedTime(&ms, h->ev[3], h->ev[4]) == hipSuccess) h->t_bt = ms;
        if (hipEventElapsedTime(&ms, h->ev[1], h->ev[4]) == hipSuccess) h->t_run = ms;
        return GBRS_OK;
    }
}

}  // namespace

extern "C" {

int gbrs_hmm_create(int num_haps, int n_chrom, const int32_t *n_genes, const int32_t *n_trans,
                    const double *const *tprob, int device, gbrs_hmm_t **out) {
    if (!out) return fail(GBRS_ERR_INVALID, "out is NULL");
    *out = nullptr;
    if (num_haps < 1 || num_haps > MAX_H) return fail(GBRS_ERR_INVALID, "num_haps must be in 1..%d", MAX_H);
    if (n_chrom < 1 || !n_genes || !n_trans || !tprob) return fail(GBRS_ERR_INVALID, "bad chromosome tables");
    GBRS_TRY(select_device(device));
    gbrs_hmm *h = new gbrs_hmm();
    struct Guard { gbrs_hmm *p; ~Guard() { if (p) gbrs_hmm_destroy(p); } } guard{h};
    h->device = device;
    h->H = num_haps;
    h->S = num_haps * (num_haps + 1) / 2;
    h->n_chrom = n_chrom;
    const int S = h->S;
    h->chroms.resize(n_chrom);
    for (int c = 0; c < n_chrom; ++c) {
        if (n_genes[c] < 1) return fail(GBRS_ERR_INVALID, "chromosome %d has no genes", c);
        // the backward sweep reads tprob[c][i] for i = 0 .. n-2 (gbrs_utils.py:541-549)
        if (n_trans[c] < n_genes[c] - 1)
            return fail(GBRS_ERR_INVALID, "index %d is out of bounds for axis 0 with size %d (tprob of chromosome %d)",
                        n_genes[c] - 2, n_trans[c], c);
        if (n_trans[c] > 0 && !tprob[c]) return fail(GBRS_ERR_INVALID, "tprob[%d] is NULL", c);
        ChromDesc &cd = h->chroms[c];
        cd.gene_off = h->total_genes;
        cd.trans_off = h->total_trans;
        cd.bp_off = h->total_bp;
        cd.chunk_off = h->total_chunks;
        h->total_chunks += (std::min(n_genes[c], n_trans[c]) + BT_B - 1) / BT_B;
        cd.n_genes = n_genes[c];
        cd.n_trans = n_trans[c];
        cd.inject = -1;
        cd.real_chrom = c;
        h->total_genes += n_genes[c];
        h->total_trans += n_trans[c];
        h->total_bp += std::min(n_genes[c], n_trans[c]);
        h->max_bp_rows = std::max(h->max_bp_rows, std::min(n_genes[c], n_trans[c]));
    }
    GBRS_HIP_CHECK(hipStreamCreateWithFlags(&h->stream, hipStreamDefault));
    GBRS_HIP_CHECK(hipStreamCreateWithFlags(&h->stream_b, hipStreamNonBlocking));
    {   // the Viterbi chain (delta -> backpointers -> backtrace) is the longest of the three: its stream goes first
        int lo_pri = 0, hi_pri = 0;
        GBRS_HIP_CHECK(hipDeviceGetStreamPriorityRange(&lo_pri, &hi_pri));
        GBRS_HIP_CHECK(hipStreamCreateWithPriority(&h->stream_c, hipStreamNonBlocking, hi_pri));
    }
    for (auto &e : h->ev) GBRS_HIP_CHECK(hipEventCreate(&e));
    for (hipEvent_t *e : {&h->ev_fork, &h->ev_b, &h->ev_c1, &h->ev_c}) GBRS_HIP_CHECK(hipEventCreate(e));
    for (hipEvent_t *e : {&h->ev_ops[0], &h->ev_ops[1]}) GBRS_HIP_CHECK(hipEventCreateWithFlags(e, hipEventDisableTiming));
    GBRS_TRY(h->d_chroms.alloc(n_chrom));
    GBRS_HIP_CHECK(hipMemcpy(h->d_chroms.p, h->chroms.data(), n_chrom * sizeof(ChromDesc), hipMemcpyHostToDevice));
    {
        std::vector<int32_t> order(n_chrom);
        for (int c = 0; c < n_chrom; ++c) order[c] = c;
        std::stable_sort(order.begin(), order.end(), [&](int32_t a, int32_t b) { return n_genes[a] > n_genes[b]; });
        GBRS_TRY(h->d_order.alloc(n_chrom));
        GBRS_HIP_CHECK(hipMemcpy(h->d_order.p, order.data(), n_chrom * sizeof(int32_t), hipMemcpyHostToDevice));
    }
    const size_t blk = (size_t)S * S;
    GBRS_TRY(h->tprob.alloc(std::max<size_t>(h->total_trans * blk, 1)));
    GBRS_TRY(h->pprob.alloc(std::max<size_t>(h->total_trans * blk, 1)));
    GBRS_TRY(h->pprob_t.alloc(std::max<size_t>(h->total_trans * blk, 1)));
    // state counts hmm_launch runs on the chain kernels keep exp(T), exp(T)^T and a copy of T in lane order
    const int lps = S == 136 ? 4 : (S == 36 || S == 28 || S == 10 || S == 6) ? 1 : 0;
    h->quad = lps != 0;
    if (h->quad) GBRS_TRY(h->tprob_q.alloc(std::max<size_t>(h->total_trans * blk, 1)));
    for (int c = 0; c < n_chrom; ++c)
        if (n_trans[c] > 0)
            GBRS_HIP_CHECK(hipMemcpy(h->tprob.p + h->chroms[c].trans_off * blk, tprob[c],
                                     (size_t)n_trans[c] * blk * sizeof(double), hipMemcpyHostToDevice));
    if (h->total_trans > 0) {
        if (h->quad)
            hipLaunchKernelGGL(lane_blocks_kernel, dim3(4096), dim3(256), 0, h->stream, S, lps, S / lps,
                               h->total_trans, h->tprob.p, h->pprob.p, h->pprob_t.p, h->tprob_q.p);
        else
            hipLaunchKernelGGL(exp_blocks_kernel, dim3(2048), dim3(256), 0, h->stream, S, h->total_trans,
                               h->tprob.p, h->pprob.p, h->pprob_t.p);
    }
    // init_vec (gbrs_utils.py:465-471): log(1/H^2) homozygous, log(2/H^2) heterozygous
    std::vector<double> iv;
    for (int a = 0; a < num_haps; ++a)
        for (int b = a; b < num_haps; ++b)
            iv.push_back(std::log((a == b ? 1.0 : 2.0) / (double)(num_haps * num_haps)));
    GBRS_TRY(h->init_vec.alloc(S));
    GBRS_HIP_CHECK(hipMemcpy(h->init_vec.p, iv.data(), S * sizeof(double), hipMemcpyHostToDevice));
    GBRS_HIP_CHECK(hipDeviceSynchronize());
    guard.p = nullptr;
    *out = h;
    return GBRS_OK;
}

int gbrs_hmm_set_expression(gbrs_hmm_t *h, int n_samples, const double *const *expr,
                            const double *const *avecs, const uint8_t *const *has_avec,
                            double expr_threshold, double sigma) {
    RoctxRange roctx_range("gbrs_hmm_set_expression");
    if (!h || !expr || n_samples < 1 || (avecs == nullptr) != (has_avec == nullptr))
        return fail(GBRS_ERR_INVALID, "bad argument");
    GBRS_TRY(select_device(h->device));
    GBRS_TRY(hmm_alloc_samples(h, n_samples));
    const int H = h->H;
    if (h->expr.n != (size_t)h->total_genes * n_samples * H) GBRS_TRY(h->expr.alloc((size_t)h->total_genes * n_samples * H));
    // The specificity blocks are sample independent (20 MB for 40k genes x 8 x 8): they are uploaded when
    // given and stay resident on the handle; later calls pass NULL tables and only move the expression rows.
    if (avecs) {
        if (!h->avecs.p) GBRS_TRY(h->avecs.alloc((size_t)h->total_genes * H * H));
        if (!h->has_avec.p) GBRS_TRY(h->has_avec.alloc(h->total_genes));
    } else if (!h->avecs.p) {
        return fail(GBRS_ERR_STATE, "no alignment specificity on the handle yet: pass avecs / has_avec once");
    }
    // Small batches: the expression rows are laid out in a pinned host image first and go over in ONE asynchronous copy on
    // the handle's stream, in front of the emission kernel (20 synchronous copies from pageable memory were 0.25 of a
    // single sample's 1.4 ms of wall time); large batches keep one strided copy per chromosome.
    const size_t expr_elems = (size_t)h->total_genes * n_samples * H;
    const bool staged = expr_elems * sizeof(double) <= ((size_t)8 << 20);       // (measured: 1 sample 0.35 -> 0.20 ms; from ~20 MB on the direct copies win)
    if (staged && h->expr_stage_n < expr_elems) {
        if (h->expr_stage) (void)hipHostFree(h->expr_stage);
        h->expr_stage = nullptr;
        h->expr_stage_n = 0;
        GBRS_HIP_CHECK(hipHostMalloc(reinterpret_cast<void **>(&h->expr_stage), expr_elems * sizeof(double), hipHostMallocDefault));
        h->expr_stage_n = expr_elems;
    }
    for (int c = 0; c < h->n_chrom; ++c) {
        const ChromDesc &cd = h->chroms[c];
        if (!expr[c] || (avecs && (!avecs[c] || !has_avec[c]))) return fail(GBRS_ERR_INVALID, "NULL table for chromosome %d", c);
        if (cd.n_genes == 0) continue;
        // the caller's [sample][gene][H] block into the genome-wide rows
        const size_t row = (size_t)cd.n_genes * H * sizeof(double);
        if (staged) {
            for (int sm = 0; sm < n_samples; ++sm)
                std::memcpy(h->expr_stage + ((size_t)sm * h->total_genes + cd.gene_off) * H,
                            expr[c] + (size_t)sm * cd.n_genes * H, row);
        } else {
            GBRS_HIP_CHECK(hipMemcpy2D(h->expr.p + (size_t)cd.gene_off * H, (size_t)h->total_genes * H * sizeof(double),
                                       expr[c], row, row, (size_t)n_samples, hipMemcpyHostToDevice));
        }
        if (avecs) {
            GBRS_HIP_CHECK(hipMemcpy(h->avecs.p + (size_t)cd.gene_off * H * H, avecs[c],
                                     (size_t)cd.n_genes * H * H * sizeof(double), hipMemcpyHostToDevice));
            GBRS_HIP_CHECK(hipMemcpy(h->has_avec.p + cd.gene_off, has_avec[c], cd.n_genes, hipMemcpyHostToDevice));
        }
    }
    const int64_t total = h->total_genes * n_samples;
    if (staged)
        GBRS_HIP_CHECK(hipMemcpyAsync(h->expr.p, h->expr_stage, expr_elems * sizeof(double), hipMemcpyHostToDevice, h->stream));
    GBRS_HIP_CHECK(hipEventRecord(h->ev[0], h->stream));
    const size_t em_lds = (size_t)EM_GENES * ((H * H + 1) + (H + 1) + 2 * (h->S + 1)) * sizeof(double);
    const int64_t em_blocks = ((h->total_genes + EM_GENES - 1) / EM_GENES) * n_samples;
    (void)total;
    // Large 36-state batches: the emission kernel is left to the run, which launches it per chromosome group in front of
    // that group's chains (hmm_launch_groups) - the second group's emission then runs beside the first group's chains.
    // GBRS_TUNING_HMM_PIPELINE = smallest batch that does so (0: never).
    int pipe_min = HMM_PIPE_MIN;
    if (const char *env = std::getenv("GBRS_TUNING_HMM_PIPELINE"); env) pipe_min = std::atoi(env) > 0 ? std::atoi(env) : INT_MAX;
    h->emission_pending = false;
    if (H == EM_LANES && h->S == MF_S && n_samples >= pipe_min && n_samples >= EM_BATCH_MIN && h->n_chrom >= 2 && h->total_trans > 0) {
        h->emission_pending = true;
        h->em_thr = expr_threshold;
        h->em_sigma = sigma;
    } else if (H == EM_LANES && n_samples >= EM_BATCH_MIN) {
        // the gene-only part of the model once per EM_BATCH_SPB samples instead of once per sample
        const dim3 grid((unsigned)((h->total_genes + EM_GENES - 1) / EM_GENES),
                        (unsigned)((n_samples + EM_BATCH_SPB - 1) / EM_BATCH_SPB));
        hipLaunchKernelGGL(emission_batch_kernel<EM_LANES>, grid, dim3(64), 0, h->stream, h->total_genes, n_samples,
                           EM_BATCH_SPB, h->expr.p, h->avecs.p, h->has_avec.p, h->init_vec.p, expr_threshold, sigma,
                           h->eprob.p, h->peprob.p, (int64_t)0, h->total_genes);
    } else {
        hipLaunchKernelGGL(emission_kernel, dim3((unsigned)em_blocks), dim3(64), em_lds, h->stream, H, h->S,
                           h->total_genes, n_samples, h->expr.p, h->avecs.p, h->has_avec.p, h->init_vec.p,
                           expr_threshold, sigma, h->eprob.p, h->peprob.p);
    }
    GBRS_HIP_CHECK(hipEventRecord(h->ev[1], h->stream));
    GBRS_HIP_CHECK(hipGetLastError());
    GBRS_HIP_CHECK(hipStreamSynchronize(h->stream));
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, h->ev[0], h->ev[1]) == hipSuccess) h->t_emis = ms;    // (a deferred emission is part of the run's time)
    h->have_eprob = true;
    h->pe_ready = !h->emission_pending;
    h->ran = false;
    return GBRS_OK;
}

int gbrs_hmm_set_eprob(gbrs_hmm_t *h, int n_samples, const double *const *eprob) {
    if (!h || !eprob || n_samples < 1) return fail(GBRS_ERR_INVALID, "bad argument");
    GBRS_TRY(select_device(h->device));
    GBRS_TRY(hmm_alloc_samples(h, n_samples));
    for (int c = 0; c < h->n_chrom; ++c) {
        const ChromDesc &cd = h->chroms[c];
        if (!eprob[c]) return fail(GBRS_ERR_INVALID, "eprob[%d] is NULL", c);
        for (int s = 0; s < n_samples; ++s)
            GBRS_HIP_CHECK(hipMemcpy(h->eprob.p + ((size_t)s * h->total_genes + cd.gene_off) * h->S,
                                     eprob[c] + (size_t)s * cd.n_genes * h->S,
                                     (size_t)cd.n_genes * h->S * sizeof(double), hipMemcpyHostToDevice));
    }
    h->have_eprob = true;
    h->emission_pending = false;
    h->pe_ready = false;
    h->ran = false;
    h->t_emis = 0;
    return GBRS_OK;
}

int gbrs_hmm_run(gbrs_hmm_t *h) {
    RoctxRange roctx_range("gbrs_hmm_run");
    if (!h) return fail(GBRS_ERR_INVALID, "handle is NULL");
    if (!h->have_eprob) return fail(GBRS_ERR_STATE, "no expression / emission data set");
    GBRS_TRY(select_device(h->device));
    const int S = h->S;
    int rc;
    // 4 lanes per state, KMAX = ceil(S / 4); the two production shapes (DO: H = 8, CC-style:
    // H = 16) divide evenly and get predicate-free instantiations
    if (S == 36) rc = hmm_launch<36, 9, 192, true>(h);            // 8 founders
    else if (S == 28) rc = hmm_launch<28, 12, 192, false>(h);     // 7
    else if (S == 10) rc = hmm_launch<10, 12, 192, false>(h);     // 4
    else if (S == 6) rc = hmm_launch<6, 12, 192, false>(h);       // 3
    else if (S == 136) rc = hmm_launch<0, 34, 576, true>(h);      // 16
    else if (S <= 48) rc = hmm_launch<0, 12, 192, false>(h);
    else if (S <= 64) rc = hmm_launch<0, 16, 256, false>(h);
    else rc = hmm_launch<0, 34, 576, false>(h);        // S <= 136 (MAX_H = 16)
    if (rc == GBRS_OK) h->ran = true;
    return rc;
}

int gbrs_hmm_get(gbrs_hmm_t *h, int sample, int chrom, double *gamma, int32_t *states, int32_t *calls,
                 double *alpha, double *beta, double *delta, double *scaler, double *eprob) {
    if (!h) return fail(GBRS_ERR_INVALID, "handle is NULL");
    if (sample < 0 || sample >= h->n_samples || chrom < 0 || chrom >= h->n_chrom)
        return fail(GBRS_ERR_INVALID, "sample/chromosome out of range");
    if (!h->ran && (gamma || states || calls || alpha || beta || delta || scaler))
        return fail(GBRS_ERR_STATE, "run() has not been called");
    GBRS_TRY(select_device(h->device));
    if (eprob && h->emission_pending) GBRS_TRY(hmm_flush_emission(h));      // asked for before any run made them
    if (alpha || beta || scaler) GBRS_TRY(hmm_make_logs(h));
    const ChromDesc &cd = h->chroms[chrom];
    const int S = h->S, n = cd.n_genes;
    const size_t goff = (size_t)sample * h->total_genes + cd.gene_off;
    std::vector<double> tmp((size_t)n * S);
    auto fetch_t = [&](const double *dev, double *dst) -> int {   // device [n][S] -> host [S][n]
        GBRS_HIP_CHECK(hipMemcpy(tmp.data(), dev + goff * S, tmp.size() * sizeof(double), hipMemcpyDeviceToHost));
        for (int i = 0; i < n; ++i)
            for (int s = 0; s < S; ++s) dst[(size_t)s * n + i] = tmp[(size_t)i * S + s];
        return GBRS_OK;
    };
    if (gamma) GBRS_TRY(fetch_t(h->gamma.p, gamma));
    if (alpha) GBRS_TRY(fetch_t(h->alpha.p, alpha));
    if (beta) GBRS_TRY(fetch_t(h->beta.p, beta));
    if (delta) {
        if (h->delta_rows.gene_stride > 1) {         // [gene][sample]: the sample's rows are gene_stride rows apart
            const RowMap &m = h->delta_rows;
            GBRS_HIP_CHECK(hipMemcpy2D(tmp.data(), (size_t)S * sizeof(double),
                                       h->delta.p + ((size_t)sample * m.sample_stride + (size_t)cd.gene_off * m.gene_stride) * S,
                                       (size_t)m.gene_stride * S * sizeof(double), (size_t)S * sizeof(double), (size_t)n,
                                       hipMemcpyDeviceToHost));
            for (int i = 0; i < n; ++i)
                for (int st = 0; st < S; ++st) delta[(size_t)st * n + i] = tmp[(size_t)i * S + st];
        } else {
            GBRS_TRY(fetch_t(h->delta.p, delta));
        }
    }
    if (scaler) GBRS_HIP_CHECK(hipMemcpy(scaler, h->scaler.p + goff, n * sizeof(double), hipMemcpyDeviceToHost));
    if (eprob) GBRS_HIP_CHECK(hipMemcpy(eprob, h->eprob.p + goff * S, (size_t)n * S * sizeof(double), hipMemcpyDeviceToHost));
    if (calls) GBRS_HIP_CHECK(hipMemcpy(calls, h->calls.p + goff, n * sizeof(int32_t), hipMemcpyDeviceToHost));
    if (states) {
        const int m = std::min(n, cd.n_trans);
        const size_t soff = (size_t)sample * (h->total_genes + h->n_chrom) + cd.gene_off + chrom;
        GBRS_HIP_CHECK(hipMemcpy(states, h->states.p + soff, (m + 1) * sizeof(int32_t), hipMemcpyDeviceToHost));
    }
    return GBRS_OK;
}

int gbrs_hmm_info(gbrs_hmm_t *h, gbrs_hmm_info_t *info) {
    if (!h || !info) return fail(GBRS_ERR_INVALID, "NULL argument");
    std::memset(info, 0, sizeof(*info));
    info->total_genes = h->total_genes;
    info->algorithmic_bytes = (uint64_t)h->total_genes * (16ull * h->S * h->S + 64ull * h->S);
    info->last_emission_ms = h->t_emis;
    info->last_forward_ms = h->t_fwd;
    info->last_backward_ms = h->t_bwd;
    info->last_backtrace_ms = h->t_bt;
    info->last_run_ms = h->t_run;
    info->num_states = h->S;
    info->n_samples = h->n_samples;
    if (h->last_delta_spec && h->n_blk[0] > 0) {
        // how the rank-convergence delta of the last run went: blocks fixed up, the longest fix-up, chains recomputed
        GBRS_TRY(select_device(h->device));
        std::vector<int32_t> g((size_t)h->n_vb * h->n_samples), f((size_t)h->n_samples * h->n_chrom);
        GBRS_HIP_CHECK(hipMemcpy(g.data(), h->dspec_g.p, g.size() * sizeof(int32_t), hipMemcpyDeviceToHost));
        GBRS_HIP_CHECK(hipMemcpy(f.data(), h->dspec_fail.p, f.size() * sizeof(int32_t), hipMemcpyDeviceToHost));
        int blocks = 0, longest = 0, fallbacks = 0;
        for (int s = 0; s < h->n_samples; ++s) {
            for (int v = 0; v < h->n_blk[0]; ++v) {
                const BlockRange &r = h->h_ranges0[v];
                if (r.lo == 0) continue;
                if (f[(size_t)s * h->n_chrom + r.chrom]) continue;
                ++blocks;
                longest = std::max(longest, g[(size_t)s * h->n_vb + v] - r.lo + 1);
            }
            for (int c = 0; c < h->n_chrom; ++c) fallbacks += f[(size_t)s * h->n_chrom + c] != 0;
        }
        info->last_delta_blocks = blocks;
        info->last_delta_longest_fixup = longest;
        info->last_delta_fallbacks = fallbacks;
    }
    return GBRS_OK;
}

int gbrs_interpolate(int S, int n_points, const double *x, const double *y, int n_grid,
                     const double *x_grid, double *out, int device) {
    if (S < 1 || n_points < 2 || n_grid < 0 || !x || !y || (n_grid && (!x_grid || !out)))
        return fail(GBRS_ERR_INVALID, "bad argument");
    for (int i = 0; i + 1 < n_points; ++i)
        if (!(x[i] <= x[i + 1])) return fail(GBRS_ERR_INVALID, "x must be ascending");
    for (int g = 0; g < n_grid; ++g) {
        if (x_grid[g] < x[0])
            return fail(GBRS_ERR_INVALID, "A value (%.17g) in x_new is below the interpolation range's minimum value (%.17g).",
                        x_grid[g], x[0]);
        if (x_grid[g] > x[n_points - 1])
            return fail(GBRS_ERR_INVALID, "A value (%.17g) in x_new is above the interpolation range's maximum value (%.17g).",
                        x_grid[g], x[n_points - 1]);
    }
    if (n_grid == 0) return GBRS_OK;
    GBRS_TRY(select_device(device));
    DevBuf<double> d_x, d_y, d_q, d_o;
    GBRS_TRY(d_x.alloc(n_points));
    GBRS_TRY(d_y.alloc((size_t)S * n_points));
    GBRS_TRY(d_q.alloc(n_grid));
    GBRS_TRY(d_o.alloc((size_t)S * n_grid));
    GBRS_HIP_CHECK(hipMemcpy(d_x.p, x, d_x.bytes(), hipMemcpyHostToDevice));
    GBRS_HIP_CHECK(hipMemcpy(d_y.p, y, d_y.bytes(), hipMemcpyHostToDevice));
    GBRS_HIP_CHECK(hipMemcpy(d_q.p, x_grid, d_q.bytes(), hipMemcpyHostToDevice));
    const int64_t total = (int64_t)S * n_grid;
    hipLaunchKernelGGL(interpolate_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, nullptr, S, n_points,
                       d_x.p, d_y.p, n_grid, d_q.p, d_o.p);
    GBRS_HIP_CHECK(hipGetLastError());
    GBRS_HIP_CHECK(hipMemcpy(out, d_o.p, d_o.bytes(), hipMemcpyDeviceToHost));
    return GBRS_OK;
}

int gbrs_genoprob_dosage(int num_haps, int64_t n_rows, const double *gprob, double *out, int device) {
    if (num_haps < 1 || num_haps > MAX_H || n_rows < 0 || (n_rows && (!gprob || !out)))
        return fail(GBRS_ERR_INVALID, "bad argument");
    if (n_rows == 0) return GBRS_OK;
    GBRS_TRY(select_device(device));
    const int S = num_haps * (num_haps + 1) / 2;
    DevBuf<double> d_p, d_o;
    GBRS_TRY(d_p.alloc((size_t)n_rows * S));
    GBRS_TRY(d_o.alloc((size_t)n_rows * num_haps));
    GBRS_HIP_CHECK(hipMemcpy(d_p.p, gprob, d_p.bytes(), hipMemcpyHostToDevice));
    const int64_t total = n_rows * num_haps;
    hipLaunchKernelGGL(dosage_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, nullptr, num_haps, S, n_rows,
                       d_p.p, d_o.p);
    GBRS_HIP_CHECK(hipGetLastError());
    GBRS_HIP_CHECK(hipMemcpy(out, d_o.p, d_o.bytes(), hipMemcpyDeviceToHost));
    return GBRS_OK;
}

int gbrs_hmm_destroy(gbrs_hmm_t *h) {
    if (!h) return GBRS_OK;
    (void)hipSetDevice(h->device);
    for (hipStream_t st : {h->stream, h->stream_b, h->stream_c, h->stream_h[0], h->stream_h[1], h->stream_h[2]})
        if (st) (void)hipStreamSynchronize(st);
    for (auto &e : h->ev)
        if (e) (void)hipEventDestroy(e);
    for (hipEvent_t e : {h->ev_fork, h->ev_b, h->ev_c1, h->ev_c, h->ev_ops[0], h->ev_ops[1]})
        if (e) (void)hipEventDestroy(e);
    for (hipStream_t st : {h->stream, h->stream_b, h->stream_c, h->stream_h[0], h->stream_h[1], h->stream_h[2]})
        if (st) (void)hipStreamDestroy(st);
    for (hipEvent_t e : h->ev_head)
        if (e) (void)hipEventDestroy(e);
    for (hipStream_t st : h->stream_g)
        if (st) { (void)hipStreamSynchronize(st); (void)hipStreamDestroy(st); }
    for (hipEvent_t e : {h->gev_em[0], h->gev_em[1], h->gev_b[0], h->gev_b[1], h->gev_c[0], h->gev_c[1], h->gev_done[0], h->gev_done[1], h->gev_start})
        if (e) (void)hipEventDestroy(e);
    if (h->expr_stage) (void)hipHostFree(h->expr_stage);
    delete h;
    return GBRS_OK;
}

}  // extern "C"

// gbrs_warm_up (common.hip): loads this file's code object
namespace gbrs {
__global__ void warm_hmm_kernel() {}
void warm_hmm(hipStream_t st) { hipLaunchKernelGGL(warm_hmm_kernel, dim3(1), dim3(64), 0, st); }
}  // namespace gbrs
