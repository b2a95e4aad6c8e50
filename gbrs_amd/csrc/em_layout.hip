// Builds the packed-row-tile device layout (em_layout.h) from the reference's CSC arrays, entirely
// on the device.  rocPRIM provides the radix sorts and scans (setup plumbing, run once per
// handle); every transformation kernel is written here.
//
// Pipeline
//   entries (h, l, r)  --sort by (r, l, h)-->  (row, locus) pairs with a haplotype mask
//   rows  --sort by (first locus, hash of locus list, hash of masks)-->  similar rows adjacent
//   [optional] identical adjacent rows merged into one weighted row (what `gbrs compress` does)
//   tiles cut where the running count of distinct locus lists or of words crosses a chunk border
//   per tile: locus dictionary (LDS bitonic sort + unique), rows padded so none straddles a
//   64-word batch, words emitted with dictionary-local indices
//   inverted index locus -> slots for the gather kernel
#include "em_layout.h"

#include <rocprim/rocprim.hpp>

#include <algorithm>

namespace gbrs {
namespace {

// ---- rocPRIM wrappers with one reusable temporary buffer ------------------------------------
struct Scratch {
    DevBuf<unsigned char> buf;
    int reserve(size_t bytes) {
        if (bytes <= buf.n) return GBRS_OK;
        return buf.alloc(bytes + (bytes >> 2) + 256);
    }
};

#define GBRS_PRIM(expr) GBRS_HIP_CHECK(expr)

template <typename T>
int exclusive_scan(Scratch &sc, const T *in, T *out, size_t n, hipStream_t s) {
    if (n == 0) return GBRS_OK;
    size_t bytes = 0;
    GBRS_PRIM(rocprim::exclusive_scan(nullptr, bytes, in, out, T(0), n, rocprim::plus<T>(), s));
    GBRS_TRY(sc.reserve(bytes));
    GBRS_PRIM(rocprim::exclusive_scan(sc.buf.p, bytes, in, out, T(0), n, rocprim::plus<T>(), s));
    return GBRS_OK;
}

template <typename T>
int inclusive_scan(Scratch &sc, const T *in, T *out, size_t n, hipStream_t s) {
    if (n == 0) return GBRS_OK;
    size_t bytes = 0;
    GBRS_PRIM(rocprim::inclusive_scan(nullptr, bytes, in, out, n, rocprim::plus<T>(), s));
    GBRS_TRY(sc.reserve(bytes));
    GBRS_PRIM(rocprim::inclusive_scan(sc.buf.p, bytes, in, out, n, rocprim::plus<T>(), s));
    return GBRS_OK;
}

int sort_keys64(Scratch &sc, const uint64_t *in, uint64_t *out, size_t n, unsigned end_bit, hipStream_t s) {
    if (n == 0) return GBRS_OK;
    size_t bytes = 0;
    GBRS_PRIM(rocprim::radix_sort_keys(nullptr, bytes, in, out, n, 0u, end_bit, s));
    GBRS_TRY(sc.reserve(bytes));
    GBRS_PRIM(rocprim::radix_sort_keys(sc.buf.p, bytes, in, out, n, 0u, end_bit, s));
    return GBRS_OK;
}

template <typename K>
int sort_pairs(Scratch &sc, const K *kin, K *kout, const uint32_t *vin, uint32_t *vout, size_t n,
               unsigned end_bit, hipStream_t s) {
    if (n == 0) return GBRS_OK;
    size_t bytes = 0;
    GBRS_PRIM(rocprim::radix_sort_pairs(nullptr, bytes, kin, kout, vin, vout, n, 0u, end_bit, s));
    GBRS_TRY(sc.reserve(bytes));
    GBRS_PRIM(rocprim::radix_sort_pairs(sc.buf.p, bytes, kin, kout, vin, vout, n, 0u, end_bit, s));
    return GBRS_OK;
}

template <typename T>
int fetch_last_plus(const T *scan_out, const T *in, size_t n, T &total, hipStream_t s) {
    // total of an exclusive scan = last output + last input
    T a = 0, b = 0;
    if (n) {
        GBRS_HIP_CHECK(hipMemcpyAsync(&a, scan_out + n - 1, sizeof(T), hipMemcpyDeviceToHost, s));
        GBRS_HIP_CHECK(hipMemcpyAsync(&b, in + n - 1, sizeof(T), hipMemcpyDeviceToHost, s));
        GBRS_HIP_CHECK(hipStreamSynchronize(s));
    }
    total = a + b;
    return GBRS_OK;
}

unsigned bits_for(uint64_t max_value) {
    unsigned b = 1;
    while (b < 64 && (max_value >> b)) ++b;
    return b;
}

inline unsigned grid_for(uint64_t n, unsigned block = 256) { return (unsigned)((n + block - 1) / block); }

// ---- kernels --------------------------------------------------------------------------------

struct BuildFlags {
    unsigned int duplicate;      // the same (row, locus, haplotype) stored twice
    unsigned int dict_overflow;  // a tile dictionary exceeded its capacity (internal error)
    unsigned int bad_row;        // row id >= R
    unsigned int n_long;         // rows with more than max_row_words(H) loci
};

// Every wavefront takes a contiguous span of KEY_SPAN entries: one full search for the column of its
// first entry, after that columns only move forward (a column holds some hundreds of entries, so most
// 64-entry steps stay inside the current one).  A full search per 64 entries made this the slowest
// kernel of the build: two chains of ~20 dependent loads for every 64 keys.
constexpr int KEY_ITERS = 32, KEY_SPAN = 64 * KEY_ITERS;
__global__ void __launch_bounds__(256)
make_keys_kernel(uint64_t n, uint32_t ncols, uint32_t L, uint64_t R, unsigned row_shift,
                 const uint64_t *__restrict__ col_ptr, const uint32_t *__restrict__ ent_row,
                 uint64_t *__restrict__ keys, BuildFlags *flags, uint32_t view_h = 0) {
    // view_h > 0 (em_layout.h, "half-loci"): haplotype h of locus l is keyed as haplotype h % view_h of locus
    // l * (H / view_h) + h / view_h - the same element of the locus-major vectors under a narrower haplotype count
    // key = row << row_shift | locus << 5 | haplotype; row_shift = 32, or 5 + the locus bits so that a
    // radix sort of the used bits has no all-zero digit to pass over
    const int lane = threadIdx.x & 63;
    const uint64_t base = ((uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * KEY_SPAN;
    if (base >= n) return;
    uint32_t c = 0;
    if (lane == 0) c = find_column(col_ptr, 0, ncols - 1, base);
    c = __shfl(c, 0, WAVE);
    bool bad = false;
    for (int it = 0; it < KEY_ITERS; ++it) {
        const uint64_t kb = base + (uint64_t)it * 64;
        if (kb >= n) break;
        c = find_column_from(col_ptr, c, ncols, kb);              // uniform: the column of the step's first entry
        const uint64_t k = kb + lane;
        if (k >= n) continue;
        const uint32_t mine = col_ptr[c + 1] > min(kb + 63, n - 1) ? c : find_column_from(col_ptr, c, ncols, k);
        uint32_t h = mine / L, l = mine - h * L;
        if (view_h) {
            l = l * (ncols / L / view_h) + h / view_h;
            h = h % view_h;
        }
        const uint32_t r = ent_row[k];
        bad |= r >= R;
        keys[k] = ((uint64_t)r << row_shift) | ((uint64_t)l << 5) | h;
    }
    if (bad) flags->bad_row = 1;
}

__global__ void pair_flag_kernel(uint64_t n, const uint64_t *__restrict__ keys, uint32_t *__restrict__ flag,
                                 BuildFlags *flags) {
    const uint64_t k = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    uint32_t f = 1;
    if (k > 0) {
        const uint64_t a = keys[k - 1], b = keys[k];
        if (a == b) flags->duplicate = 1;
        f = (a >> 5) != (b >> 5);
    }
    flag[k] = f;
}

__global__ void emit_pairs_kernel(uint64_t n, const uint64_t *__restrict__ keys, const uint32_t *__restrict__ flag,
                                  const uint32_t *__restrict__ pidx, unsigned row_shift,
                                  uint32_t *__restrict__ prow, uint32_t *__restrict__ ploc,
                                  uint32_t *__restrict__ pmask) {
    const uint64_t k = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n || !flag[k]) return;
    const uint64_t key = keys[k];
    uint32_t mask = 0;
    for (uint64_t j = k; j < n && (keys[j] >> 5) == (key >> 5); ++j) mask |= 1u << (uint32_t)(keys[j] & 31);
    const uint32_t p = pidx[k];
    prow[p] = (uint32_t)(key >> row_shift);
    ploc[p] = (uint32_t)((key >> 5) & ((1u << (row_shift - 5)) - 1u));
    pmask[p] = mask;
}

__global__ void row_flag_kernel(uint64_t np, const uint32_t *__restrict__ prow, uint32_t *__restrict__ flag) {
    const uint64_t p = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= np) return;
    flag[p] = (p == 0) || prow[p] != prow[p - 1];
}

__global__ void row_start_kernel(uint64_t np, uint64_t nrows, const uint32_t *__restrict__ flag,
                                 const uint32_t *__restrict__ ridx, const uint32_t *__restrict__ prow,
                                 uint32_t *__restrict__ rowstart, uint32_t *__restrict__ row_orig) {
    const uint64_t p = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p == 0) rowstart[nrows] = (uint32_t)np;
    if (p >= np || !flag[p]) return;
    rowstart[ridx[p]] = (uint32_t)p;
    row_orig[ridx[p]] = prow[p];
}

// ---- locus sets (em_layout.h) ---------------------------------------------------------------------
constexpr uint32_t NO_SET = 0xFFFFFFFFu;
constexpr uint32_t SET_MAX_LOCI = 1024;

__device__ __forceinline__ uint64_t mix64(uint64_t h, uint64_t v) {
    h = (h ^ v) * 0x9E3779B97F4A7C15ull;
    return h ^ (h >> 29);
}

// flag[r] = 1 when row r has >= 2 (row, locus) pairs that all carry one mask; key[r] = hash of its locus list
__global__ void set_candidate_kernel(uint64_t nrows, const uint32_t *__restrict__ rowstart, const uint32_t *__restrict__ ploc,
                                     const uint32_t *__restrict__ pmask, uint64_t *__restrict__ key, uint32_t *__restrict__ flag) {
    const uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= nrows) return;
    const uint32_t b = rowstart[r], e = rowstart[r + 1];
    bool ok = e - b >= 2 && e - b <= SET_MAX_LOCI;
    uint64_t h = e - b;
    if (ok) {
        const uint32_t m0 = pmask[b];
        for (uint32_t k = b; k < e; ++k) {
            ok &= pmask[k] == m0;
            h = mix64(h, ploc[k]);
        }
    }
    flag[r] = ok ? 1u : 0u;
    key[r] = h;
}

__global__ void set_compact_kernel(uint64_t nrows, const uint32_t *__restrict__ flag, const uint32_t *__restrict__ cidx,
                                   const uint64_t *__restrict__ key, uint64_t *__restrict__ ckey, uint32_t *__restrict__ crow) {
    const uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= nrows || !flag[r]) return;
    ckey[cidx[r]] = key[r];
    crow[cidx[r]] = (uint32_t)r;
}

// candidates sorted by hash: head[i] = 1 when candidate i's locus list differs from its predecessor's (exact comparison:
// a hash collision can only split a set in two, never join two)
__global__ void set_head_kernel(uint64_t nc, const uint64_t *__restrict__ skey, const uint32_t *__restrict__ srow,
                                const uint32_t *__restrict__ rowstart, const uint32_t *__restrict__ ploc, uint32_t *__restrict__ head) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nc) return;
    bool same = i > 0 && skey[i] == skey[i - 1];
    if (same) {
        const uint32_t a = srow[i], b = srow[i - 1];
        const uint32_t a0 = rowstart[a], b0 = rowstart[b], n = rowstart[a + 1] - a0;
        same = n == rowstart[b + 1] - b0;
        for (uint32_t k = 0; same && k < n; ++k) same = ploc[a0 + k] == ploc[b0 + k];
    }
    head[i] = same ? 0u : 1u;
}

__global__ void set_assign_kernel(uint64_t nc, const uint32_t *__restrict__ head, const uint32_t *__restrict__ hincl,
                                  const uint32_t *__restrict__ srow, const uint32_t *__restrict__ rowstart,
                                  uint32_t *__restrict__ set_of_row, uint32_t *__restrict__ set_len, uint32_t *__restrict__ set_rep) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nc) return;
    const uint32_t k = hincl[i] - 1, r = srow[i];
    set_of_row[r] = k;
    if (head[i]) {
        set_len[k] = rowstart[r + 1] - rowstart[r];
        set_rep[k] = r;
    }
}

__global__ void set_members_kernel(uint32_t n_sets, const uint32_t *__restrict__ set_ptr, const uint32_t *__restrict__ set_rep,
                                   const uint32_t *__restrict__ rowstart, const uint32_t *__restrict__ ploc,
                                   uint32_t *__restrict__ members) {
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n_sets) return;
    const uint32_t b = rowstart[set_rep[k]], o = set_ptr[k], n = set_ptr[k + 1] - o;
    for (uint32_t j = 0; j < n; ++j) members[o + j] = ploc[b + j];
}

__global__ void set_row_len_kernel(uint64_t nrows, const uint32_t *__restrict__ rowstart, const uint32_t *__restrict__ set_of_row,
                                   uint32_t *__restrict__ newlen) {
    const uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= nrows) return;
    newlen[r] = set_of_row[r] != NO_SET ? 1u : rowstart[r + 1] - rowstart[r];
}

__global__ void set_rewrite_kernel(uint64_t nrows, uint32_t L, const uint32_t *__restrict__ rowstart,
                                   const uint32_t *__restrict__ set_of_row, const uint32_t *__restrict__ newstart,
                                   const uint32_t *__restrict__ ploc, const uint32_t *__restrict__ pmask,
                                   uint32_t *__restrict__ ploc2, uint32_t *__restrict__ pmask2) {
    const uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= nrows) return;
    const uint32_t b = rowstart[r], e = rowstart[r + 1], o = newstart[r], k = set_of_row[r];
    if (k != NO_SET) {
        ploc2[o] = L + k;
        pmask2[o] = pmask[b];
    } else {
        for (uint32_t j = b; j < e; ++j) {
            ploc2[o + (j - b)] = ploc[j];
            pmask2[o + (j - b)] = pmask[j];
        }
    }
}

// ---- locus sets per mask group (round 4) ----------------------------------------------------------------------------
// A read that aligns to several isoforms of a gene with DIFFERENT haplotype masks is no whole-row set, but the loci of the row
// that share a mask are one: den = sum over the groups g of sum_h m_g,h * (theta[l1,h] + theta[l2,h] + ...), and every locus of a
// group receives the same count/den.  Rows are regrouped by (mask, locus); every group of >= 2 loci is a candidate, and a
// candidate becomes a set only when at least `min_rows` rows carry it: thin sets fill the tiles' dictionaries for nothing
// (profiles/r03_estep_experiments.txt item 16: 216 k sets, most of them carried by a handful of reads, made the
// multi-isoform sample 1.8x slower; the frequent ones are where the reads are).
constexpr uint32_t GROUP_ROW_MAX = 32;

// per row: pairs sorted by (mask, locus) into gloc / gmask (same offsets); nseg[r] = number of mask groups
__global__ void group_sort_kernel(uint64_t nrows, const uint32_t *__restrict__ rowstart, const uint32_t *__restrict__ ploc,
                                  const uint32_t *__restrict__ pmask, uint32_t *__restrict__ gloc, uint32_t *__restrict__ gmask,
                                  uint32_t *__restrict__ nseg) {
    const uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= nrows) return;
    const uint32_t b = rowstart[r], n = rowstart[r + 1] - b;
    if (n > GROUP_ROW_MAX) {                     // long rows stay as they are: every pair a group of its own
        for (uint32_t k = 0; k < n; ++k) { gloc[b + k] = ploc[b + k]; gmask[b + k] = pmask[b + k]; }
        nseg[r] = n;
        return;
    }
    uint64_t v[GROUP_ROW_MAX];
    for (uint32_t k = 0; k < n; ++k) v[k] = ((uint64_t)pmask[b + k] << 32) | ploc[b + k];
    for (uint32_t i = 1; i < n; ++i) {           // insertion sort (the loci arrive ascending: only the masks reorder)
        const uint64_t x = v[i];
        uint32_t j = i;
        while (j > 0 && v[j - 1] > x) { v[j] = v[j - 1]; --j; }
        v[j] = x;
    }
    uint32_t ns = 0;
    for (uint32_t k = 0; k < n; ++k) {
        gloc[b + k] = (uint32_t)v[k];
        gmask[b + k] = (uint32_t)(v[k] >> 32);
        ns += (k == 0 || (v[k] >> 32) != (v[k - 1] >> 32)) ? 1u : 0u;
    }
    nseg[r] = ns;
}

// per row: its groups as segments [seg_begin, seg_begin + seg_len) of gloc; a group of >= 2 loci is a candidate with the hash
// of its locus list as key
__global__ void group_segments_kernel(uint64_t nrows, const uint32_t *__restrict__ rowstart, const uint32_t *__restrict__ gloc,
                                      const uint32_t *__restrict__ gmask, const uint32_t *__restrict__ segoff,
                                      uint32_t *__restrict__ seg_begin, uint32_t *__restrict__ seg_len, uint64_t *__restrict__ key,
                                      uint32_t *__restrict__ cand) {
    const uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= nrows) return;
    const uint32_t b = rowstart[r], e = rowstart[r + 1];
    uint32_t s = segoff[r];
    const bool plain = e - b > GROUP_ROW_MAX;
    uint32_t k = b;
    while (k < e) {
        uint32_t k2 = k + 1;
        uint64_t h = 0;
        if (!plain) {
            h = mix64(0x51ed270b1ull, gloc[k]);
            while (k2 < e && gmask[k2] == gmask[k]) { h = mix64(h, gloc[k2]); ++k2; }
        }
        seg_begin[s] = k;
        seg_len[s] = k2 - k;
        key[s] = mix64(h, k2 - k);
        cand[s] = k2 - k >= 2 ? 1u : 0u;
        ++s;
        k = k2;
    }
}

__global__ void group_compact_kernel(uint64_t nseg, const uint32_t *__restrict__ cand, const uint32_t *__restrict__ cidx,
                                     const uint64_t *__restrict__ key, uint64_t *__restrict__ ckey, uint32_t *__restrict__ cseg) {
    const uint64_t s = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= nseg || !cand[s]) return;
    ckey[cidx[s]] = key[s];
    cseg[cidx[s]] = (uint32_t)s;
}

// candidates sorted by hash: head[i] = 1 when candidate i's locus list differs from its predecessor's (exact comparison)
__global__ void group_head_kernel(uint64_t nc, const uint64_t *__restrict__ skey, const uint32_t *__restrict__ sseg,
                                  const uint32_t *__restrict__ seg_begin, const uint32_t *__restrict__ seg_len,
                                  const uint32_t *__restrict__ gloc, uint32_t *__restrict__ head) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nc) return;
    bool same = i > 0 && skey[i] == skey[i - 1];
    if (same) {
        const uint32_t a = sseg[i], b = sseg[i - 1];
        const uint32_t a0 = seg_begin[a], b0 = seg_begin[b], n = seg_len[a];
        same = n == seg_len[b];
        for (uint32_t k = 0; same && k < n; ++k) same = gloc[a0 + k] == gloc[b0 + k];
    }
    head[i] = same ? 0u : 1u;
}

// rows per provisional set, and its first candidate (sorted order) as representative
__global__ void group_count_kernel(uint64_t nc, const uint32_t *__restrict__ head, const uint32_t *__restrict__ hincl,
                                   const uint32_t *__restrict__ sseg, uint32_t *__restrict__ count, uint32_t *__restrict__ rep) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nc) return;
    const uint32_t k = hincl[i] - 1;
    atomicAdd(&count[k], 1u);
    if (head[i]) rep[k] = sseg[i];
}

__global__ void group_keep_kernel(uint32_t nprov, uint32_t min_rows, const uint32_t *__restrict__ count, uint32_t *__restrict__ keep) {
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k < nprov) keep[k] = count[k] >= min_rows ? 1u : 0u;
}

// set_of_seg for the candidates of kept sets; length and representative segment of every kept set
__global__ void group_assign_kernel(uint64_t nc, const uint32_t *__restrict__ hincl, const uint32_t *__restrict__ sseg,
                                    const uint32_t *__restrict__ keep, const uint32_t *__restrict__ newid,
                                    const uint32_t *__restrict__ rep, const uint32_t *__restrict__ seg_len,
                                    uint32_t *__restrict__ set_of_seg, uint32_t *__restrict__ set_len, uint32_t *__restrict__ set_rep) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nc) return;
    const uint32_t k = hincl[i] - 1;
    if (!keep[k]) return;
    const uint32_t id = newid[k];
    set_of_seg[sseg[i]] = id;
    if (rep[k] == sseg[i]) {
        set_len[id] = seg_len[sseg[i]];
        set_rep[id] = sseg[i];
    }
}

__global__ void group_members_kernel(uint32_t n_sets, const uint32_t *__restrict__ set_ptr, const uint32_t *__restrict__ set_rep,
                                     const uint32_t *__restrict__ seg_begin, const uint32_t *__restrict__ gloc,
                                     uint32_t *__restrict__ members) {
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n_sets) return;
    const uint32_t b = seg_begin[set_rep[k]], o = set_ptr[k], n = set_ptr[k + 1] - o;
    for (uint32_t j = 0; j < n; ++j) members[o + j] = gloc[b + j];       // ascending: a group is sorted by locus
}

__global__ void group_row_len_kernel(uint64_t nrows, const uint32_t *__restrict__ segoff, const uint32_t *__restrict__ seg_len,
                                     const uint32_t *__restrict__ set_of_seg, uint32_t *__restrict__ newlen) {
    const uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= nrows) return;
    uint32_t n = 0;
    for (uint32_t s = segoff[r]; s < segoff[r + 1]; ++s) n += set_of_seg[s] != NO_SET ? 1u : seg_len[s];
    newlen[r] = n;
}

// the rows in their new form: a kept group becomes one pair on its set's id, the others keep their pairs; ids ascending
__global__ void group_rewrite_kernel(uint64_t nrows, uint32_t L, const uint32_t *__restrict__ segoff,
                                     const uint32_t *__restrict__ seg_begin, const uint32_t *__restrict__ seg_len,
                                     const uint32_t *__restrict__ set_of_seg, const uint32_t *__restrict__ newstart,
                                     const uint32_t *__restrict__ gloc, const uint32_t *__restrict__ gmask,
                                     uint32_t *__restrict__ ploc2, uint32_t *__restrict__ pmask2) {
    const uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= nrows) return;
    const uint32_t o = newstart[r], n = newstart[r + 1] - o;
    uint32_t w = o;
    for (uint32_t s = segoff[r]; s < segoff[r + 1]; ++s) {
        const uint32_t b = seg_begin[s], k = set_of_seg[s];
        if (k != NO_SET) {
            ploc2[w] = L + k;
            pmask2[w] = gmask[b];
            ++w;
        } else {
            for (uint32_t j = 0; j < seg_len[s]; ++j) { ploc2[w] = gloc[b + j]; pmask2[w] = gmask[b + j]; ++w; }
        }
    }
    if (n > GROUP_ROW_MAX) return;               // (a long row was copied in its own order)
    for (uint32_t i = 1; i < n; ++i) {           // back to ascending ids, as every later step expects of a row
        const uint32_t xl = ploc2[o + i], xm = pmask2[o + i];
        uint32_t j = i;
        while (j > 0 && ploc2[o + j - 1] > xl) { ploc2[o + j] = ploc2[o + j - 1]; pmask2[o + j] = pmask2[o + j - 1]; --j; }
        ploc2[o + j] = xl;
        pmask2[o + j] = xm;
    }
}


__device__ __forceinline__ uint32_t mix32(uint32_t h, uint32_t v) {
    h ^= v + 0x9e3779b9u + (h << 6) + (h >> 2);
    h *= 0x85ebca6bu;
    h ^= h >> 13;
    return h;
}

__global__ void row_key_kernel(uint64_t nrows, uint32_t max_row, unsigned loc_shift, const uint32_t *__restrict__ rowstart,
                               const uint32_t *__restrict__ ploc, const uint32_t *__restrict__ pmask,
                               uint64_t *__restrict__ key, uint32_t *__restrict__ ident, BuildFlags *flags) {
    const uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= nrows) return;
    ident[r] = (uint32_t)r;
    const uint32_t a = rowstart[r], b = rowstart[r + 1];
    if (b - a > max_row) {
        key[r] = ~0ull;
        atomicAdd(&flags->n_long, 1u);
        return;
    }
    uint32_t hl = 0x811c9dc5u, hm = 0x01000193u;
    for (uint32_t p = a; p < b; ++p) {
        hl = mix32(hl, ploc[p]);
        hm = mix32(hm, pmask[p]);
    }
    const uint64_t primary = (ploc[a] >> loc_shift) & 0xFFFFFFu;
    uint64_t k = (primary << 40) | ((uint64_t)(hl & 0xFFFFFFu) << 16) | (hm & 0xFFFFu);
    if (k == ~0ull) k -= 1;
    key[r] = k;
}

__device__ __forceinline__ bool same_loci(const uint32_t *__restrict__ rowstart, const uint32_t *__restrict__ ploc,
                                          uint32_t ra, uint32_t rb) {
    const uint32_t a = rowstart[ra], na = rowstart[ra + 1] - a;
    const uint32_t b = rowstart[rb], nb = rowstart[rb + 1] - b;
    if (na != nb) return false;
    for (uint32_t j = 0; j < na; ++j)
        if (ploc[a + j] != ploc[b + j]) return false;
    return true;
}

// head[i] = 1 when sorted row i starts a new (merged) row
__global__ void merge_flag_kernel(uint64_t n, int merge, const uint64_t *__restrict__ skey,
                                  const uint32_t *__restrict__ srow, const uint32_t *__restrict__ rowstart,
                                  const uint32_t *__restrict__ ploc, const uint32_t *__restrict__ pmask,
                                  uint32_t *__restrict__ head) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t f = 1;
    if (merge && i > 0 && skey[i] == skey[i - 1]) {
        const uint32_t ra = srow[i - 1], rb = srow[i];
        if (same_loci(rowstart, ploc, ra, rb)) {
            const uint32_t a = rowstart[ra], b = rowstart[rb], cnt = rowstart[ra + 1] - a;
            bool eq = true;
            for (uint32_t j = 0; j < cnt; ++j) eq &= pmask[a + j] == pmask[b + j];
            if (eq) f = 0;
        }
    }
    head[i] = f;
}

// per sorted row: merged ordinal m = incl[i] - 1; accumulate weights, record representative
__global__ void merged_rows_kernel(uint64_t n, const uint32_t *__restrict__ head, const uint32_t *__restrict__ hincl,
                                   const uint32_t *__restrict__ srow, const uint32_t *__restrict__ row_orig,
                                   const double *__restrict__ count, uint32_t *__restrict__ hrow,
                                   double *__restrict__ weight) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t m = hincl[i] - 1;
    if (head[i]) hrow[m] = srow[i];
    if (weight) atomicAdd(&weight[m], count ? count[row_orig[srow[i]]] : 1.0);
}

__global__ void row_len_kernel(uint64_t m_rows, const uint32_t *__restrict__ hrow, const uint32_t *__restrict__ rowstart,
                               const uint32_t *__restrict__ ploc, uint32_t *__restrict__ npm,
                               uint32_t *__restrict__ dnew) {
    const uint64_t m = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (m >= m_rows) return;
    const uint32_t r = hrow[m];
    const uint32_t cnt = rowstart[r + 1] - rowstart[r];
    npm[m] = cnt;
    const bool newlist = (m == 0) || !same_loci(rowstart, ploc, hrow[m - 1], r);
    dnew[m] = newlist ? cnt : 0;
}

__global__ void tile_flag_kernel(uint64_t m_rows, uint32_t tile_words, uint32_t dseg, const uint32_t *__restrict__ npm,
                                 const uint32_t *__restrict__ wordoff, const uint32_t *__restrict__ dincl,
                                 uint32_t *__restrict__ tflag) {
    const uint64_t m = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (m >= m_rows) return;
    uint32_t f = 1;
    if (m > 0) {
        const uint32_t dc = dincl[m] - npm[m], dp = dincl[m - 1] - npm[m - 1];
        f = (wordoff[m] / tile_words != wordoff[m - 1] / tile_words) || (dc / dseg != dp / dseg);
    }
    tflag[m] = f;
}

// interleave: rows of one locus list are dealt round-robin against the other lists of the tile so
// that the 64 words of a batch spread over as many dictionary entries as possible (LDS atomic
// conflicts in the E-step come from lanes that share a partial-sum copy AND a locus)
__global__ void group_flag_kernel(uint64_t m_rows, const uint32_t *__restrict__ dnew, const uint32_t *__restrict__ tflag,
                                  uint32_t *__restrict__ gflag, uint32_t *__restrict__ gpos) {
    const uint64_t m = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (m >= m_rows) return;
    const uint32_t f = (dnew[m] > 0) || tflag[m];
    gflag[m] = f;
    gpos[m] = f ? (uint32_t)m : 0u;
}

__global__ void interleave_key_kernel(uint64_t m_rows, const uint32_t *__restrict__ tincl, const uint32_t *__restrict__ gstart,
                                      const uint32_t *__restrict__ gord, uint64_t *__restrict__ key,
                                      uint32_t *__restrict__ ident) {
    const uint64_t m = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (m >= m_rows) return;
    const uint64_t rank = (uint32_t)m - gstart[m];
    key[m] = ((uint64_t)(tincl[m] - 1) << 40) | ((rank & 0x3FFFu) << 26) | (gord[m] & 0x3FFFFFFu);
    ident[m] = (uint32_t)m;
}

// stream order: inside a tile, rows by length (then in their sorted order), see tile_pad_kernel
__global__ void stream_key_kernel(uint64_t m_rows, const uint32_t *__restrict__ tincl, const uint32_t *__restrict__ npm,
                                  uint64_t *__restrict__ key, uint32_t *__restrict__ ident) {
    const uint64_t m = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (m >= m_rows) return;
    key[m] = ((uint64_t)(tincl[m] - 1) << 40) | ((uint64_t)min(npm[m], 63u) << 32) | (uint32_t)m;
    ident[m] = (uint32_t)m;
}

__global__ void permute_rows_kernel(uint64_t m_rows, const uint32_t *__restrict__ perm, const uint32_t *__restrict__ hrow,
                                    const uint32_t *__restrict__ npm, const double *__restrict__ weight,
                                    uint32_t *__restrict__ hrow2, uint32_t *__restrict__ npm2,
                                    double *__restrict__ weight2) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= m_rows) return;
    const uint32_t m = perm[i];
    hrow2[i] = hrow[m];
    npm2[i] = npm[m];
    if (weight) weight2[i] = weight[m];
}

__global__ void tile_start_kernel(uint64_t m_rows, uint64_t n_tiles, const uint32_t *__restrict__ tflag,
                                  const uint32_t *__restrict__ tincl, uint32_t *__restrict__ tile_row) {
    const uint64_t m = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (m == 0) tile_row[n_tiles] = (uint32_t)m_rows;
    if (m >= m_rows || !tflag[m]) return;
    tile_row[tincl[m] - 1] = (uint32_t)m;
}

// one thread per tile: padded offset of every row so that no row straddles a 64-word batch.
// streams: the tile's rows arrive grouped by length; a run of n rows of length w (w | 64) takes
// B = ceil(n / (64/w)) whole batches in which lane group g (w lanes) holds rows g*B .. g*B+B-1 of the
// run at batches 0 .. B-1: whichever batches a wavefront owns, each of its lanes walks a
// contiguous piece of the sorted rows.
__global__ void tile_pad_kernel(uint64_t n_tiles, int streams, const uint32_t *__restrict__ tile_row,
                                const uint32_t *__restrict__ npm, uint32_t *__restrict__ rowpad,
                                uint32_t *__restrict__ nbatch) {
    const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_tiles) return;
    uint32_t off = 0;
    const uint32_t end = tile_row[t + 1];
    uint32_t m = tile_row[t];
    while (m < end) {
        const uint32_t cnt = npm[m];
        if (streams && cnt <= 32u && (64u % cnt) == 0u) {
            uint32_t e = m + 1;
            while (e < end && npm[e] == cnt) ++e;
            const uint32_t n = e - m, G = 64u / cnt, B = (n + G - 1) / G;
            off = (off + 63u) & ~63u;
            for (uint32_t k = 0; k < n; ++k) rowpad[m + k] = off + (k % B) * 64u + (k / B) * cnt;
            off += B * 64u;
            m = e;
        } else {
            if ((off & 63u) + cnt > 64u) off = (off + 63u) & ~63u;
            rowpad[m] = off;
            off += cnt;
            ++m;
        }
    }
    nbatch[t] = (off + 63u) >> 6;
}

// Per-tile dictionaries = the distinct loci (and locus sets) of a tile's rows, ascending.  Built from ONE radix sort of
// (tile, id) keys over all the pairs of the layout (round 4; before: a bitonic sort per tile in LDS, which capped a tile at
// 16,384 words - and the E-step likes its tiles as large as one round of the chip's workgroups allows).
__global__ void tile_pair_keys_kernel(uint64_t m_rows, const uint32_t *__restrict__ tincl, const uint32_t *__restrict__ hrow,
                                      const uint32_t *__restrict__ rowstart, const uint32_t *__restrict__ ploc,
                                      const uint32_t *__restrict__ wordoff, uint64_t *__restrict__ keys) {
    const uint64_t m = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (m >= m_rows) return;
    const uint64_t t = tincl[m] - 1;
    const uint32_t r = hrow[m], p0 = rowstart[r], cnt = rowstart[r + 1] - p0, o = wordoff[m];
    for (uint32_t j = 0; j < cnt; ++j) keys[o + j] = (t << 32) | ploc[p0 + j];
}

__global__ void dict_flag_kernel(uint64_t n, const uint64_t *__restrict__ skeys, uint32_t *__restrict__ flag) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) flag[i] = (i == 0 || skeys[i] != skeys[i - 1]) ? 1u : 0u;
}

__global__ void dict_emit_kernel(uint64_t n, const uint64_t *__restrict__ skeys, const uint32_t *__restrict__ flag,
                                 const uint32_t *__restrict__ pos, uint32_t *__restrict__ dict, uint32_t *__restrict__ dict_base) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n || !flag[i]) return;
    dict[pos[i]] = (uint32_t)skeys[i];
    if (i == 0 || (skeys[i] >> 32) != (skeys[i - 1] >> 32)) dict_base[skeys[i] >> 32] = pos[i];
}

__global__ void tile_hdr_kernel(uint64_t n_tiles, uint32_t dcap, uint32_t n_slots, const uint32_t *__restrict__ batch_base,
                                const uint32_t *__restrict__ nbatch, const uint32_t *__restrict__ dict_base,
                                TileHdr *__restrict__ hdr, BuildFlags *flags) {
    const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_tiles) return;
    const uint32_t db = dict_base[t], d = (t + 1 < n_tiles ? dict_base[t + 1] : n_slots) - db;
    if (d > dcap) flags->dict_overflow = 1;
    hdr[t] = TileHdr{batch_base[t], nbatch[t], db, d};
}

// one thread per (merged) row: emit its words at the padded position
__global__ void emit_words_kernel(uint64_t m_rows, uint32_t H, const uint32_t *__restrict__ tincl,
                                  const TileHdr *__restrict__ hdr, const uint32_t *__restrict__ dict,
                                  const uint32_t *__restrict__ hrow, const uint32_t *__restrict__ rowstart,
                                  const uint32_t *__restrict__ ploc, const uint32_t *__restrict__ pmask,
                                  const uint32_t *__restrict__ rowpad, uint32_t *__restrict__ words,
                                  const double *__restrict__ row_weight, double *__restrict__ word_weight) {
    const uint64_t m = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (m >= m_rows) return;
    const TileHdr th = hdr[tincl[m] - 1];
    const uint32_t PB = pos_bits(H);
    const uint32_t r = hrow[m], p0 = rowstart[r], cnt = rowstart[r + 1] - p0;
    const uint32_t off = rowpad[m];
    const uint64_t base = (uint64_t)th.batch_base * 64 + off;
    const uint32_t *d = dict + th.dict_base;
    for (uint32_t j = 0; j < cnt; ++j) {
        const uint32_t l = ploc[p0 + j];
        uint32_t lo = 0, hi = th.dict_count;          // lower_bound; l is present by construction
        while (lo < hi) {
            const uint32_t mid = (lo + hi) >> 1;
            if (d[mid] < l) lo = mid + 1; else hi = mid;
        }
        words[base + j] = pmask[p0 + j] | (j << H) | ((cnt - 1 - j) << (H + PB)) | (lo << (H + 2 * PB));
        if (word_weight) word_weight[base + j] = row_weight[m];
    }
}

__global__ void iota_kernel(uint64_t n, uint32_t *__restrict__ v) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) v[i] = (uint32_t)i;
}

// destination entries of a slot: one for a locus, [header, one per member] for a locus set
__global__ void dest_count_kernel(uint64_t ns, uint32_t L, const uint32_t *__restrict__ dict, const uint32_t *__restrict__ set_ptr,
                                  uint32_t *__restrict__ cnt) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= ns) return;
    const uint32_t id = dict[i];
    cnt[i] = id < L ? 1u : 1u + (set_ptr[id - L + 1] - set_ptr[id - L]);
}

// eloc / eidx: (locus, own index) of every entry; slot_dest of a locus slot = its entry (routed below), of a set slot
// = SLOT_SET | offset of its header in dest_list, dest_list[header] = number of members
__global__ void dest_emit_kernel(uint64_t ns, uint32_t L, const uint32_t *__restrict__ dict, const uint32_t *__restrict__ set_ptr,
                                 const uint32_t *__restrict__ members, const uint32_t *__restrict__ eoff,
                                 uint32_t *__restrict__ eloc, uint32_t *__restrict__ eidx, uint32_t *__restrict__ slot_dest,
                                 uint32_t *__restrict__ dest_list) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= ns) return;
    const uint32_t id = dict[i], o = eoff[i];
    if (id < L) {
        eloc[o] = id;
        eidx[o] = o;
        slot_dest[i] = o;                    // provisional: the entry whose destination dest_route_kernel copies here
    } else {
        const uint32_t b = set_ptr[id - L], n = set_ptr[id - L + 1] - b;
        eloc[o] = L;                         // header
        eidx[o] = o;
        dest_list[o] = n;
        for (uint32_t j = 0; j < n; ++j) {
            eloc[o + 1 + j] = members[b + j];
            eidx[o + 1 + j] = o + 1 + j;
        }
        slot_dest[i] = SLOT_SET | o;
    }
}

// a locus slot takes its destination out of dest_list (where locus_class_kernel left the destination of every entry)
__global__ void dest_route_kernel(uint64_t ns, const uint32_t *__restrict__ dest_list, uint32_t *__restrict__ slot_dest) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= ns) return;
    const uint32_t d = slot_dest[i];
    if (!(d & SLOT_SET)) slot_dest[i] = dest_list[d];
}

// Last step of the build: the dictionary and the destinations in the form the E-step kernel reads (em_layout.h) - a
// two-member set carries its members and its two destinations in place (one round trip less in the tile's prologue and
// epilogue than the general set's lists), a larger set its index.
__global__ void encode_sets_kernel(uint64_t ns, uint32_t L, const uint32_t *__restrict__ set_ptr, const uint32_t *__restrict__ members,
                                   const uint32_t *__restrict__ dest_list, uint32_t *__restrict__ dict, uint32_t *__restrict__ dict_b,
                                   uint32_t *__restrict__ slot_dest, uint32_t *__restrict__ dest_b) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= ns) return;
    const uint32_t id = dict[i];
    dict_b[i] = 0;
    dest_b[i] = 0;
    if (id < L) return;
    const uint32_t k = id - L, b = set_ptr[k], n = set_ptr[k + 1] - b;
    if (n == 2) {
        const uint32_t off = slot_dest[i] & ~SLOT_SET;
        dict[i] = DICT_PAIR | members[b];
        dict_b[i] = members[b + 1];
        slot_dest[i] = SLOT_PAIR | dest_list[off + 1];      // (a destination is a row < 2^29 or SLOT_DIRECT | locus < 2^27)
        dest_b[i] = dest_list[off + 2];
    } else {
        dict[i] = DICT_SET | k;
    }
}

__global__ void slot_ptr_kernel(uint32_t L, uint64_t n_slots, const uint32_t *__restrict__ sorted_loc,
                                uint32_t *__restrict__ slot_ptr) {
    const uint32_t l = blockIdx.x * blockDim.x + threadIdx.x;
    if (l > L) return;
    uint64_t lo = 0, hi = n_slots;                   // first index with sorted_loc >= l
    while (lo < hi) {
        const uint64_t mid = (lo + hi) >> 1;
        if (sorted_loc[mid] < l) lo = mid + 1; else hi = mid;
    }
    slot_ptr[l] = (uint32_t)lo;
}

__global__ void locus_class_kernel(uint32_t L, const uint32_t *__restrict__ slot_ptr, const uint32_t *__restrict__ slot_list,
                                   uint8_t *__restrict__ cls, uint32_t *__restrict__ slot_dest) {
    const uint32_t l = blockIdx.x * blockDim.x + threadIdx.x;
    if (l >= L) return;
    const uint32_t k0 = slot_ptr[l], cnt = slot_ptr[l + 1] - k0;
    cls[l] = cnt == 0 ? 0 : (cnt == 1 ? 1 : (cnt <= (uint32_t)HEAVY_SLOTS ? 2 : 3));
    // Destination of every (tile, dictionary entry) sum: straight into A for a locus that lives in one
    // tile; otherwise row k of `partials`, k = the entry's rank in the inverted index, so that the
    // slots of a locus are consecutive rows and the gather streams them without an indirection.
    // (slot_list holds entry indices, slot_dest here is the per-entry destination array: dest_list)
    if (cnt == 1) slot_dest[slot_list[k0]] = SLOT_DIRECT | l;
    else
        for (uint32_t k = k0; k < k0 + cnt; ++k) slot_dest[slot_list[k]] = k;
}

__global__ void long_rows_kernel(uint64_t n_long, uint64_t first, const uint32_t *__restrict__ srow,
                                 const uint32_t *__restrict__ rowstart, const uint32_t *__restrict__ row_orig,
                                 const double *__restrict__ count, uint64_t *__restrict__ len,
                                 double *__restrict__ weight) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_long) return;
    const uint32_t r = srow[first + i];
    len[i] = rowstart[r + 1] - rowstart[r];
    weight[i] = count ? count[row_orig[r]] : 1.0;
}

__global__ void long_copy_kernel(uint64_t n_long, uint64_t first, const uint32_t *__restrict__ srow,
                                 const uint32_t *__restrict__ rowstart, const uint32_t *__restrict__ ploc,
                                 const uint32_t *__restrict__ pmask, const uint64_t *__restrict__ long_ptr,
                                 uint32_t *__restrict__ long_loc, uint32_t *__restrict__ long_mask) {
    const uint64_t i = blockIdx.x;
    if (i >= n_long) return;
    const uint32_t r = srow[first + i], p0 = rowstart[r], cnt = rowstart[r + 1] - p0;
    const uint64_t o = long_ptr[i];
    for (uint32_t j = threadIdx.x; j < cnt; j += blockDim.x) {
        long_loc[o + j] = ploc[p0 + j];
        long_mask[o + j] = pmask[p0 + j];
    }
}

}  // namespace

// ---- `--report-alignment-counts` (AlignmentPropertyMatrix.py:389-459) -------------------------
constexpr uint64_t COUNT_KEY_DROPPED = ~0ull;
__global__ void __launch_bounds__(256)
count_keys_kernel(uint64_t n, uint32_t ncols, uint32_t L, uint64_t R, const uint64_t *__restrict__ col_ptr,
                  const uint32_t *__restrict__ ent_row, const int32_t *__restrict__ locus_group,
                  uint64_t *__restrict__ keys, BuildFlags *flags) {
    const uint64_t k = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k - (threadIdx.x & 63) >= n) return;
    const bool live = k < n;
    const uint32_t c = entry_column(col_ptr, ncols, live ? k : n - 1, n);
    if (!live) return;
    const uint32_t h = c / L, l = c - h * L;
    const uint32_t r = ent_row[k];
    const int32_t lo = locus_group ? locus_group[l] : (int32_t)l;
    if (r >= R) flags->bad_row = 1;
    // an entry of a locus outside every group disappears from the bundled matrix
    // (AlignmentPropertyMatrix.py:155-188: the product with grp_conv_mat has no column for it); such
    // entries, like out-of-range row ids, get the all-ones key, which sorts last and is skipped below
    keys[k] = (r >= R || lo < 0) ? COUNT_KEY_DROPPED : (((uint64_t)r << 32) | ((uint64_t)(uint32_t)lo << 5) | h);
}

// per unique (row, locus', hap) entry: bump the row's entry counter; per unique (row, locus')
// pair: bump the row's locus counter
__global__ void count_rowstat_kernel(uint64_t n, const uint64_t *__restrict__ keys, uint32_t *__restrict__ nnz_row,
                                     uint32_t *__restrict__ nloc_row) {
    const uint64_t k = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    const uint64_t key = keys[k];
    if (key == COUNT_KEY_DROPPED) return;
    const bool new_entry = k == 0 || keys[k - 1] != key;
    const bool new_pair = k == 0 || (keys[k - 1] >> 5) != (key >> 5);
    const uint32_t r = (uint32_t)(key >> 32);
    if (new_entry) atomicAdd(&nnz_row[r], 1u);
    if (new_pair) atomicAdd(&nloc_row[r], 1u);
}

__global__ void count_accum_kernel(uint64_t n, uint32_t Lout, const uint64_t *__restrict__ keys,
                                   const uint32_t *__restrict__ nnz_row, const uint32_t *__restrict__ nloc_row,
                                   const double *__restrict__ count, double *__restrict__ aln,
                                   double *__restrict__ uniq, double *__restrict__ locus_uniq) {
    const uint64_t k = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    const uint64_t key = keys[k];
    if (key == COUNT_KEY_DROPPED) return;
    const bool new_entry = k == 0 || keys[k - 1] != key;
    if (!new_entry) return;
    const bool new_pair = k == 0 || (keys[k - 1] >> 5) != (key >> 5);
    const uint32_t r = (uint32_t)(key >> 32), l = (uint32_t)((key >> 5) & 0x7FFFFFFu), h = (uint32_t)(key & 31);
    const double w = count ? count[r] : 1.0;
    atomicAdd(&aln[(size_t)h * Lout + l], w);                       // integer-valued sums: exact
    if (nnz_row[r] == 1) atomicAdd(&uniq[(size_t)h * Lout + l], w);
    if (new_pair && nloc_row[r] == 1) atomicAdd(&locus_uniq[l], w);
}

struct CountsWork {
    Scratch sc;
    DevBuf<BuildFlags> d_flags;
    DevBuf<uint64_t> keys, keys2;
    DevBuf<uint32_t> nnz_row, nloc_row;
    uint64_t n = 0, rows = 0;
};
CountsWork *counts_work_new() { return new CountsWork(); }
void counts_work_free(CountsWork *w) { delete w; }

int alignment_counts_device(uint64_t R, uint32_t L, uint32_t H, uint64_t N, const uint32_t *ent_row,
                            const uint64_t *col_ptr, const double *count, const int32_t *locus_group,
                            uint32_t Lout, double *aln, double *uniq, double *locus_uniq, hipStream_t s,
                            CountsWork *work) {
    GBRS_HIP_CHECK(hipMemsetAsync(aln, 0, (size_t)H * Lout * 8, s));
    GBRS_HIP_CHECK(hipMemsetAsync(uniq, 0, (size_t)H * Lout * 8, s));
    GBRS_HIP_CHECK(hipMemsetAsync(locus_uniq, 0, (size_t)Lout * 8, s));
    if (N == 0) { GBRS_HIP_CHECK(hipStreamSynchronize(s)); return GBRS_OK; }
    CountsWork local;
    CountsWork &w = work ? *work : local;
    if (w.n != N || w.rows != R || !w.keys.p) {
        GBRS_TRY(w.d_flags.alloc(1));
        GBRS_TRY(w.keys.alloc(N)); GBRS_TRY(w.keys2.alloc(N)); GBRS_TRY(w.nnz_row.alloc(R)); GBRS_TRY(w.nloc_row.alloc(R));
        w.n = N;
        w.rows = R;
    }
    GBRS_HIP_CHECK(hipMemsetAsync(w.d_flags.p, 0, sizeof(BuildFlags), s));
    GBRS_HIP_CHECK(hipMemsetAsync(w.nnz_row.p, 0, w.nnz_row.bytes(), s));
    GBRS_HIP_CHECK(hipMemsetAsync(w.nloc_row.p, 0, w.nloc_row.bytes(), s));
    hipLaunchKernelGGL(count_keys_kernel, dim3(grid_for(N)), dim3(256), 0, s, N, H * L, L, R, col_ptr, ent_row,
                       locus_group, w.keys.p, w.d_flags.p);
    // the dropped key is all ones: within the compared bits it is larger than every real key (Lout < 2^27),
    // so dropped entries end up behind all real ones
    GBRS_TRY(sort_keys64(w.sc, w.keys.p, w.keys2.p, N, 32 + bits_for(R - 1), s));
    hipLaunchKernelGGL(count_rowstat_kernel, dim3(grid_for(N)), dim3(256), 0, s, N, w.keys2.p, w.nnz_row.p, w.nloc_row.p);
    hipLaunchKernelGGL(count_accum_kernel, dim3(grid_for(N)), dim3(256), 0, s, N, Lout, w.keys2.p, w.nnz_row.p, w.nloc_row.p,
                       count, aln, uniq, locus_uniq);
    BuildFlags hf{};
    GBRS_HIP_CHECK(hipMemcpyAsync(&hf, w.d_flags.p, sizeof(hf), hipMemcpyDeviceToHost, s));
    GBRS_HIP_CHECK(hipStreamSynchronize(s));
    GBRS_HIP_CHECK(hipGetLastError());
    if (hf.bad_row) return fail(GBRS_ERR_INVALID, "indices hold a row id >= num_rows");
    return GBRS_OK;
}

// ---- `gbrs compress` (gbrs/emase_utils.py:60-103): identical rows -> equivalence classes -------
__global__ void ec_min_row_kernel(uint64_t n, const uint32_t *__restrict__ hincl, const uint32_t *__restrict__ srow,
                                  const uint32_t *__restrict__ row_orig, const double *__restrict__ count,
                                  uint32_t *__restrict__ minrow, double *__restrict__ weight) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t g = hincl[i] - 1;
    const uint32_t r = row_orig[srow[i]];
    atomicMin(&minrow[g], r);
    atomicAdd(&weight[g], count ? count[r] : 1.0);
}

// rows without any alignment all share the empty key: one more class, at its first row's rank
__global__ void ec_empty_rows_kernel(uint64_t R, const uint32_t *__restrict__ nnz_row, const double *__restrict__ count,
                                     uint32_t *__restrict__ minrow_slot, double *__restrict__ weight_slot) {
    const uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= R || nnz_row[r] != 0) return;
    atomicMin(minrow_slot, (uint32_t)r);
    atomicAdd(weight_slot, count ? count[r] : 1.0);
}

__global__ void ec_row_nnz_kernel(uint64_t n, const uint32_t *__restrict__ ent_row, uint32_t *__restrict__ nnz_row) {
    const uint64_t k = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k < n) atomicAdd(&nnz_row[ent_row[k]], 1u);
}

__global__ void ec_rank_scatter_kernel(uint64_t n, const uint32_t *__restrict__ sorted_group, uint32_t *__restrict__ newid) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) newid[sorted_group[i]] = (uint32_t)i;
}

__global__ void ec_permute_count_kernel(uint64_t n, const uint32_t *__restrict__ newid, const double *__restrict__ w,
                                        double *__restrict__ out) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[newid[i]] = w[i];
}

__global__ void ec_count_entries_kernel(uint64_t n_short, const uint32_t *__restrict__ head, const uint32_t *__restrict__ srow,
                                        const uint32_t *__restrict__ rowstart, const uint32_t *__restrict__ pmask,
                                        uint32_t *__restrict__ nent) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_short) return;
    uint32_t c = 0;
    if (head[i]) {
        const uint32_t r = srow[i];
        for (uint32_t p = rowstart[r]; p < rowstart[r + 1]; ++p) c += __popc(pmask[p]);
    }
    nent[i] = c;
}

__global__ void ec_emit_entries_kernel(uint64_t n_short, const uint32_t *__restrict__ head, const uint32_t *__restrict__ hincl,
                                       const uint32_t *__restrict__ srow, const uint32_t *__restrict__ rowstart,
                                       const uint32_t *__restrict__ ploc, const uint32_t *__restrict__ pmask,
                                       const uint32_t *__restrict__ eoff, const uint32_t *__restrict__ newid,
                                       uint64_t *__restrict__ keys) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_short || !head[i]) return;
    const uint32_t r = srow[i];
    const uint64_t id = newid[hincl[i] - 1];
    uint32_t o = eoff[i];
    for (uint32_t p = rowstart[r]; p < rowstart[r + 1]; ++p) {
        uint32_t m = pmask[p];
        while (m) {
            const uint32_t h = __ffs(m) - 1;
            m &= m - 1;
            keys[o++] = ((uint64_t)h << 59) | ((uint64_t)ploc[p] << 32) | id;     // sort by (hap, locus, class)
        }
    }
}

__global__ void ec_split_kernel(uint64_t n, const uint64_t *__restrict__ keys, uint32_t *__restrict__ indices) {
    const uint64_t k = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k < n) indices[k] = (uint32_t)(keys[k] & 0xFFFFFFFFu);
}

__global__ void ec_indptr_kernel(uint32_t L, uint32_t H, uint64_t n, const uint64_t *__restrict__ keys,
                                 uint64_t *__restrict__ colptr) {
    const uint64_t c = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;   // column id h*L + l, plus the end
    if (c > (uint64_t)H * L) return;
    const uint64_t h = c / L, l = c - h * L;
    const uint64_t target = c == (uint64_t)H * L ? ~0ull : ((h << 59) | (l << 32));
    uint64_t lo = 0, hi = n;
    while (lo < hi) {
        const uint64_t mid = (lo + hi) >> 1;
        if (keys[mid] < target) lo = mid + 1; else hi = mid;
    }
    colptr[c] = lo;
}

int compress_device(CompressResult &out, uint64_t R, uint32_t L, uint32_t H, uint64_t N, const uint32_t *ent_row,
                    const uint64_t *col_ptr, const double *count, hipStream_t s) {
    if (H > 16 || N >= 0xFFFFFFFFull || L >= (1u << 27))
        return fail(GBRS_ERR_INVALID, "compress needs H <= 16, N < 2^32 entries and L < 2^27 loci");
    Scratch sc;
    DevBuf<BuildFlags> d_flags;
    GBRS_TRY(d_flags.alloc(1));
    GBRS_HIP_CHECK(hipMemsetAsync(d_flags.p, 0, sizeof(BuildFlags), s));
    BuildFlags hf{};
    out.num_ecs = 0;
    out.n_entries = 0;
    GBRS_TRY(out.col_ptr.alloc((size_t)H * L + 1));
    GBRS_HIP_CHECK(hipMemsetAsync(out.col_ptr.p, 0, out.col_ptr.bytes(), s));
    // rows without alignments
    DevBuf<uint32_t> nnz_row;
    GBRS_TRY(nnz_row.alloc(R));
    GBRS_HIP_CHECK(hipMemsetAsync(nnz_row.p, 0, nnz_row.bytes(), s));
    if (N) hipLaunchKernelGGL(ec_row_nnz_kernel, dim3(grid_for(N)), dim3(256), 0, s, N, ent_row, nnz_row.p);
    uint64_t P = 0, R1 = 0, M = 0;
    DevBuf<uint32_t> prow, ploc, pmask, rowstart, row_orig, srow, head, hincl;
    if (N) {
        DevBuf<uint64_t> keys, keys2;
        GBRS_TRY(keys.alloc(N)); GBRS_TRY(keys2.alloc(N));
        hipLaunchKernelGGL(make_keys_kernel, dim3(grid_for(N, 4 * KEY_SPAN)), dim3(256), 0, s, N, H * L, L, R, 32u, col_ptr, ent_row, keys.p,
                           d_flags.p);
        GBRS_TRY(sort_keys64(sc, keys.p, keys2.p, N, 32 + bits_for(R - 1), s));
        keys.release();
        DevBuf<uint32_t> pflag, pidx;
        GBRS_TRY(pflag.alloc(N)); GBRS_TRY(pidx.alloc(N));
        hipLaunchKernelGGL(pair_flag_kernel, dim3(grid_for(N)), dim3(256), 0, s, N, keys2.p, pflag.p, d_flags.p);
        GBRS_TRY(exclusive_scan(sc, pflag.p, pidx.p, N, s));
        uint32_t P32 = 0;
        GBRS_TRY(fetch_last_plus(pidx.p, pflag.p, N, P32, s));
        GBRS_HIP_CHECK(hipMemcpyAsync(&hf, d_flags.p, sizeof(hf), hipMemcpyDeviceToHost, s));
        GBRS_HIP_CHECK(hipStreamSynchronize(s));
        if (hf.bad_row) return fail(GBRS_ERR_INVALID, "indices hold a row id >= num_rows");
        if (hf.duplicate) return fail(GBRS_ERR_INVALID, "duplicate (row, locus, haplotype) entry: the CSC arrays must be canonical");
        P = P32;
        GBRS_TRY(prow.alloc(P)); GBRS_TRY(ploc.alloc(P)); GBRS_TRY(pmask.alloc(P));
        hipLaunchKernelGGL(emit_pairs_kernel, dim3(grid_for(N)), dim3(256), 0, s, N, keys2.p, pflag.p, pidx.p, 32u, prow.p, ploc.p,
                           pmask.p);
        GBRS_HIP_CHECK(hipStreamSynchronize(s));
        keys2.release(); pflag.release(); pidx.release();
        DevBuf<uint32_t> rflag, ridx;
        GBRS_TRY(rflag.alloc(P)); GBRS_TRY(ridx.alloc(P));
        hipLaunchKernelGGL(row_flag_kernel, dim3(grid_for(P)), dim3(256), 0, s, P, prow.p, rflag.p);
        GBRS_TRY(exclusive_scan(sc, rflag.p, ridx.p, P, s));
        uint32_t R32 = 0;
        GBRS_TRY(fetch_last_plus(ridx.p, rflag.p, P, R32, s));
        R1 = R32;
        GBRS_TRY(rowstart.alloc(R1 + 1)); GBRS_TRY(row_orig.alloc(R1));
        hipLaunchKernelGGL(row_start_kernel, dim3(grid_for(P)), dim3(256), 0, s, P, R1, rflag.p, ridx.p, prow.p, rowstart.p,
                           row_orig.p);
        GBRS_HIP_CHECK(hipStreamSynchronize(s));
        rflag.release(); ridx.release(); prow.release();
        DevBuf<uint64_t> rkey, skey;
        DevBuf<uint32_t> ident;
        GBRS_TRY(rkey.alloc(R1)); GBRS_TRY(skey.alloc(R1)); GBRS_TRY(ident.alloc(R1)); GBRS_TRY(srow.alloc(R1));
        const unsigned lbits = bits_for(L - 1);
        hipLaunchKernelGGL(row_key_kernel, dim3(grid_for(R1)), dim3(256), 0, s, R1, 0xFFFFFFFFu, lbits > 24 ? lbits - 24 : 0u,
                           rowstart.p, ploc.p, pmask.p, rkey.p, ident.p, d_flags.p);
        GBRS_TRY(sort_pairs<uint64_t>(sc, rkey.p, skey.p, ident.p, srow.p, R1, 64, s));
        GBRS_TRY(head.alloc(R1)); GBRS_TRY(hincl.alloc(R1));
        hipLaunchKernelGGL(merge_flag_kernel, dim3(grid_for(R1)), dim3(256), 0, s, R1, 1, skey.p, srow.p, rowstart.p, ploc.p,
                           pmask.p, head.p);
        GBRS_TRY(inclusive_scan(sc, head.p, hincl.p, R1, s));
        uint32_t m32 = 0;
        GBRS_HIP_CHECK(hipMemcpyAsync(&m32, hincl.p + R1 - 1, 4, hipMemcpyDeviceToHost, s));
        GBRS_HIP_CHECK(hipStreamSynchronize(s));
        M = m32;
    }
    // classes: M from rows with alignments (+1 for the empty key if any row is empty)
    DevBuf<uint32_t> minrow;
    DevBuf<double> weight;
    GBRS_TRY(minrow.alloc(M + 1));
    GBRS_TRY(weight.alloc(M + 1));
    GBRS_HIP_CHECK(hipMemsetAsync(minrow.p, 0xFF, minrow.bytes(), s));
    GBRS_HIP_CHECK(hipMemsetAsync(weight.p, 0, weight.bytes(), s));
    if (R1) hipLaunchKernelGGL(ec_min_row_kernel, dim3(grid_for(R1)), dim3(256), 0, s, R1, hincl.p, srow.p, row_orig.p, count,
                               minrow.p, weight.p);
    hipLaunchKernelGGL(ec_empty_rows_kernel, dim3(grid_for(R)), dim3(256), 0, s, R, nnz_row.p, count, minrow.p + M,
                       weight.p + M);
    uint32_t empty_first = 0xFFFFFFFFu;
    GBRS_HIP_CHECK(hipMemcpyAsync(&empty_first, minrow.p + M, 4, hipMemcpyDeviceToHost, s));
    GBRS_HIP_CHECK(hipStreamSynchronize(s));
    const uint64_t G = M + (empty_first != 0xFFFFFFFFu ? 1 : 0);
    out.num_ecs = G;
    if (G == 0) return GBRS_OK;
    // first-seen order (dict insertion order of the reference, :77, :95): rank classes by first row
    DevBuf<uint32_t> gid, gsorted, msorted, newid;
    GBRS_TRY(gid.alloc(G)); GBRS_TRY(gsorted.alloc(G)); GBRS_TRY(msorted.alloc(G)); GBRS_TRY(newid.alloc(G));
    hipLaunchKernelGGL(iota_kernel, dim3(grid_for(G)), dim3(256), 0, s, G, gid.p);
    GBRS_TRY(sort_pairs<uint32_t>(sc, minrow.p, msorted.p, gid.p, gsorted.p, G, 32, s));
    hipLaunchKernelGGL(ec_rank_scatter_kernel, dim3(grid_for(G)), dim3(256), 0, s, G, gsorted.p, newid.p);
    GBRS_TRY(out.count.alloc(G));
    {   // counts in the new order
        DevBuf<double> tmpw;
        GBRS_TRY(tmpw.alloc(G));
        GBRS_HIP_CHECK(hipMemcpyAsync(tmpw.p, weight.p, G * 8, hipMemcpyDeviceToDevice, s));
        // out.count[newid[g]] = weight[g]
        hipLaunchKernelGGL(ec_permute_count_kernel, dim3(grid_for(G)), dim3(256), 0, s, G, newid.p, tmpw.p, out.count.p);
        GBRS_HIP_CHECK(hipStreamSynchronize(s));
    }
    if (R1 == 0) return GBRS_OK;
    // entries of the representative rows, relabelled and ordered by (hap, locus, class)
    DevBuf<uint32_t> nent, eoff;
    GBRS_TRY(nent.alloc(R1)); GBRS_TRY(eoff.alloc(R1));
    hipLaunchKernelGGL(ec_count_entries_kernel, dim3(grid_for(R1)), dim3(256), 0, s, R1, head.p, srow.p, rowstart.p, pmask.p,
                       nent.p);
    GBRS_TRY(exclusive_scan(sc, nent.p, eoff.p, R1, s));
    uint32_t E32 = 0;
    GBRS_TRY(fetch_last_plus(eoff.p, nent.p, R1, E32, s));
    const uint64_t E = E32;
    out.n_entries = E;
    DevBuf<uint64_t> ekeys, ekeys2;
    GBRS_TRY(ekeys.alloc(E)); GBRS_TRY(ekeys2.alloc(E));
    hipLaunchKernelGGL(ec_emit_entries_kernel, dim3(grid_for(R1)), dim3(256), 0, s, R1, head.p, hincl.p, srow.p, rowstart.p,
                       ploc.p, pmask.p, eoff.p, newid.p, ekeys.p);
    GBRS_TRY(sort_keys64(sc, ekeys.p, ekeys2.p, E, 64, s));
    GBRS_TRY(out.indices.alloc(E));
    hipLaunchKernelGGL(ec_split_kernel, dim3(grid_for(E)), dim3(256), 0, s, E, ekeys2.p, out.indices.p);
    hipLaunchKernelGGL(ec_indptr_kernel, dim3(grid_for((uint64_t)H * L + 1)), dim3(256), 0, s, L, H, E, ekeys2.p,
                       out.col_ptr.p);
    GBRS_HIP_CHECK(hipStreamSynchronize(s));
    GBRS_HIP_CHECK(hipGetLastError());
    return GBRS_OK;
}

int build_tile_layout(TileLayout &out, uint64_t R, uint32_t L_in, uint32_t H, uint64_t N,
                      const uint32_t *ent_row, const uint64_t *col_ptr, const double *count,
                      bool merge, int row_order, bool deterministic, hipStream_t s, unsigned side_by_side, bool locus_sets,
                      uint32_t dict_cap, uint32_t view_factor) {
    uint32_t L = L_in;                     // grows by the number of locus sets in step 3b
    out.n_sets = 0;
    out.n_dest_rows = 0;
    const bool interleave = row_order == 1, streams = row_order == 2;
    if (H > 16) return fail(GBRS_ERR_INVALID, "the tiled layout packs the haplotype mask in 16 bits (H <= 16)");
    if (N >= 0xFFFFFFFFull || L >= (1u << 27))
        return fail(GBRS_ERR_INVALID, "the tiled layout needs N < 2^32 entries and L < 2^27 loci per handle");
    // (common.h) the temporaries are parked while the build runs - a hipMalloc that follows a large hipFree stalls on
    // some hosts - and go back in one pass when it ends; a one-shot process leaves them to the layout's destructor
    DeferFrees park_temporaries(out.retain_temporaries ? &out.retired : nullptr, &out.retired_bytes);
    StageTimer stg("layout");
    Scratch sc;
    DevBuf<BuildFlags> d_flags;
    GBRS_TRY(d_flags.alloc(1));
    GBRS_HIP_CHECK(hipMemsetAsync(d_flags.p, 0, sizeof(BuildFlags), s));
    BuildFlags hf{};
    auto read_flags = [&]() -> int {
        GBRS_HIP_CHECK(hipMemcpyAsync(&hf, d_flags.p, sizeof(hf), hipMemcpyDeviceToHost, s));
        GBRS_HIP_CHECK(hipStreamSynchronize(s));
        return GBRS_OK;
    };
    out.d_max = std::min<uint32_t>(1024, lds_theta_doubles(merge || count != nullptr, (int)H) / H);
    out.d_max = std::min<uint32_t>(out.d_max, dict_index_limit((int)H));     // what a word's index field can hold (256 at H = 16)
    out.deterministic = deterministic;
    if (deterministic) out.d_max = std::min<uint32_t>(out.d_max, det_dict_cap(H, merge || count != nullptr));
    if (dict_cap) out.d_max = std::min<uint32_t>(out.d_max, std::max<uint32_t>(dict_cap, (uint32_t)max_row_words(H) + 1));
    if (const char *env = std::getenv("GBRS_TUNING_DICT_CAP"); env && std::atoi(env) > max_row_words(H))
        out.d_max = std::min<uint32_t>(out.d_max, (uint32_t)std::atoi(env));
    if (out.d_max <= (uint32_t)max_row_words(H))
        return fail(GBRS_ERR_UNSUPPORTED, "the deterministic tile layout has no room for a row's loci at H = %u", H);
    const uint32_t dseg = out.d_max - max_row_words(H);
    out.weighted = merge || count != nullptr;
    out.n_pairs = out.n_rows = out.n_rows_in = out.n_long = out.n_tiles = out.n_batches = out.n_slots = 0;
    GBRS_TRY(out.slot_ptr.alloc((size_t)L + 1));
    GBRS_HIP_CHECK(hipMemsetAsync(out.slot_ptr.p, 0, out.slot_ptr.bytes(), s));
    GBRS_TRY(out.locus_class.alloc(L));
    GBRS_HIP_CHECK(hipMemsetAsync(out.locus_class.p, 0, out.locus_class.bytes(), s));
    if (N == 0) { GBRS_HIP_CHECK(hipStreamSynchronize(s)); return GBRS_OK; }

    // 1. entries -> sorted (row, locus, hap) keys
    DevBuf<uint64_t> keys, keys2;
    stg.mark("0 setup");
    GBRS_TRY(keys.alloc(N));
    GBRS_TRY(keys2.alloc(N));
    stg.mark("1a key buffers");
    const unsigned row_shift = 5 + bits_for(L - 1);           // <= 32 (L < 2^27 checked above)
    // (view_factor > 1: L_in and H are those of the half-locus view, the CSC columns those of the caller's L_in / factor loci
    // with H * factor haplotypes)
    hipLaunchKernelGGL(make_keys_kernel, dim3(grid_for(N, 4 * KEY_SPAN)), dim3(256), 0, s, N, H * L, L / view_factor, R, row_shift,
                       col_ptr, ent_row, keys.p, d_flags.p, view_factor > 1 ? H : 0u);
    stg.mark("1b make keys");
    GBRS_TRY(sort_keys64(sc, keys.p, keys2.p, N, row_shift + bits_for(R - 1), s));
    stg.mark("1c sort entries");
    keys.release();
    stg.mark("1d release");
    // 2. pairs
    DevBuf<uint32_t> pflag, pidx;
    GBRS_TRY(pflag.alloc(N));
    GBRS_TRY(pidx.alloc(N));
    stg.mark("2a alloc flags");
    hipLaunchKernelGGL(pair_flag_kernel, dim3(grid_for(N)), dim3(256), 0, s, N, keys2.p, pflag.p, d_flags.p);
    stg.mark("2b pair flags");
    GBRS_TRY(exclusive_scan(sc, pflag.p, pidx.p, N, s));
    stg.mark("2c scan");
    uint32_t P32 = 0;
    GBRS_TRY(fetch_last_plus(pidx.p, pflag.p, N, P32, s));
    GBRS_TRY(read_flags());
    stg.mark("2d fetch");
    if (hf.bad_row) return fail(GBRS_ERR_INVALID, "indices hold a row id >= num_rows");
    if (hf.duplicate) return fail(GBRS_ERR_INVALID, "duplicate (row, locus, haplotype) entry: the CSC arrays must be canonical");
    const uint64_t P = P32;
    out.n_pairs = P;
    DevBuf<uint32_t> prow, ploc, pmask;
    GBRS_TRY(prow.alloc(P));
    GBRS_TRY(ploc.alloc(P));
    GBRS_TRY(pmask.alloc(P));
    hipLaunchKernelGGL(emit_pairs_kernel, dim3(grid_for(N)), dim3(256), 0, s, N, keys2.p, pflag.p, pidx.p, row_shift,
                       prow.p, ploc.p, pmask.p);
    GBRS_HIP_CHECK(hipStreamSynchronize(s));
    stg.mark("2e emit pairs");
    keys2.release(); pflag.release(); pidx.release();
    stg.mark("2 pairs (release)");
    // 3. rows
    DevBuf<uint32_t> rflag, ridx;
    GBRS_TRY(rflag.alloc(P));
    GBRS_TRY(ridx.alloc(P));
    hipLaunchKernelGGL(row_flag_kernel, dim3(grid_for(P)), dim3(256), 0, s, P, prow.p, rflag.p);
    GBRS_TRY(exclusive_scan(sc, rflag.p, ridx.p, P, s));
    uint32_t R1 = 0;
    GBRS_TRY(fetch_last_plus(ridx.p, rflag.p, P, R1, s));
    out.n_rows_in = R1;
    DevBuf<uint32_t> rowstart, row_orig;
    GBRS_TRY(rowstart.alloc((size_t)R1 + 1));
    GBRS_TRY(row_orig.alloc(R1));
    hipLaunchKernelGGL(row_start_kernel, dim3(grid_for(P)), dim3(256), 0, s, P, (uint64_t)R1, rflag.p, ridx.p, prow.p,
                       rowstart.p, row_orig.p);
    GBRS_HIP_CHECK(hipStreamSynchronize(s));
    rflag.release(); ridx.release(); prow.release();
    stg.mark("3 rows");
    // 3b. locus sets, first form (round 3): a row whose pairs ALL carry one mask becomes one pair on the id of its locus set
    // (em_layout.h), every distinct set kept.  Round 4's step 3c below finds the same sets (a whole row is a row with one mask
    // group) and keeps the frequent ones only, which is what the samples want (C2: all 98,725 sets 0.0901 ms per iteration,
    // the 17-42 k sets carried by >= 128 / 32 reads 0.0857); this form stays reachable with GBRS_TUNING_LOCUS_SETS=1.
    const bool whole_row_sets_forced = [] { const char *e = std::getenv("GBRS_TUNING_LOCUS_SETS"); return e && std::atoi(e) == 1; }();
    if (locus_sets && R1 > 0 && whole_row_sets_forced) {
        DevBuf<uint64_t> key, ckey, skey2;
        DevBuf<uint32_t> flag, cidx, crow, srow2;
        GBRS_TRY(key.alloc(R1)); GBRS_TRY(flag.alloc(R1)); GBRS_TRY(cidx.alloc(R1));
        hipLaunchKernelGGL(set_candidate_kernel, dim3(grid_for(R1)), dim3(256), 0, s, (uint64_t)R1, rowstart.p, ploc.p, pmask.p,
                           key.p, flag.p);
        GBRS_TRY(exclusive_scan(sc, flag.p, cidx.p, R1, s));
        uint32_t C = 0;
        GBRS_TRY(fetch_last_plus(cidx.p, flag.p, R1, C, s));
        if (C > 0) {
            GBRS_TRY(ckey.alloc(C)); GBRS_TRY(skey2.alloc(C)); GBRS_TRY(crow.alloc(C)); GBRS_TRY(srow2.alloc(C));
            hipLaunchKernelGGL(set_compact_kernel, dim3(grid_for(R1)), dim3(256), 0, s, (uint64_t)R1, flag.p, cidx.p, key.p,
                               ckey.p, crow.p);
            GBRS_TRY(sort_pairs<uint64_t>(sc, ckey.p, skey2.p, crow.p, srow2.p, C, 64, s));
            key.release(); ckey.release(); crow.release();
            DevBuf<uint32_t> head2, hincl2, set_of_row;
            GBRS_TRY(head2.alloc(C)); GBRS_TRY(hincl2.alloc(C)); GBRS_TRY(set_of_row.alloc(R1));
            hipLaunchKernelGGL(set_head_kernel, dim3(grid_for(C)), dim3(256), 0, s, (uint64_t)C, skey2.p, srow2.p, rowstart.p,
                               ploc.p, head2.p);
            GBRS_TRY(inclusive_scan(sc, head2.p, hincl2.p, C, s));
            uint32_t V = 0;
            GBRS_HIP_CHECK(hipMemcpyAsync(&V, hincl2.p + C - 1, 4, hipMemcpyDeviceToHost, s));
            GBRS_HIP_CHECK(hipStreamSynchronize(s));
            // (ids of loci and sets share the 27 bits of a row key: a sample with that many distinct sets keeps its plain rows)
            const bool ids_fit = (uint64_t)L_in + V < (1u << 27);
            if (!ids_fit) V = 0;
            DevBuf<uint32_t> set_len, set_rep;
            if (ids_fit) {
            GBRS_TRY(set_len.alloc(V)); GBRS_TRY(set_rep.alloc(V));
            GBRS_HIP_CHECK(hipMemsetAsync(set_of_row.p, 0xFF, set_of_row.bytes(), s));
            hipLaunchKernelGGL(set_assign_kernel, dim3(grid_for(C)), dim3(256), 0, s, (uint64_t)C, head2.p, hincl2.p, srow2.p,
                               rowstart.p, set_of_row.p, set_len.p, set_rep.p);
            GBRS_TRY(out.set_ptr.alloc((size_t)V + 1));
            GBRS_TRY(exclusive_scan(sc, set_len.p, out.set_ptr.p, V, s));
            uint32_t n_members = 0;
            GBRS_TRY(fetch_last_plus(out.set_ptr.p, set_len.p, V, n_members, s));
            GBRS_HIP_CHECK(hipMemcpyAsync(out.set_ptr.p + V, &n_members, 4, hipMemcpyHostToDevice, s));
            GBRS_TRY(out.set_members.alloc(n_members));
            hipLaunchKernelGGL(set_members_kernel, dim3(grid_for(V)), dim3(256), 0, s, V, out.set_ptr.p, set_rep.p, rowstart.p,
                               ploc.p, out.set_members.p);
            // the rows in their new form
            DevBuf<uint32_t> newlen, rowstart2, ploc2, pmask2;
            GBRS_TRY(newlen.alloc(R1)); GBRS_TRY(rowstart2.alloc((size_t)R1 + 1));
            hipLaunchKernelGGL(set_row_len_kernel, dim3(grid_for(R1)), dim3(256), 0, s, (uint64_t)R1, rowstart.p, set_of_row.p,
                               newlen.p);
            GBRS_TRY(exclusive_scan(sc, newlen.p, rowstart2.p, R1, s));
            uint32_t P2 = 0;
            GBRS_TRY(fetch_last_plus(rowstart2.p, newlen.p, R1, P2, s));
            GBRS_HIP_CHECK(hipMemcpyAsync(rowstart2.p + R1, &P2, 4, hipMemcpyHostToDevice, s));
            GBRS_TRY(ploc2.alloc(P2)); GBRS_TRY(pmask2.alloc(P2));
            hipLaunchKernelGGL(set_rewrite_kernel, dim3(grid_for(R1)), dim3(256), 0, s, (uint64_t)R1, L_in, rowstart.p,
                               set_of_row.p, rowstart2.p, ploc.p, pmask.p, ploc2.p, pmask2.p);
            GBRS_HIP_CHECK(hipStreamSynchronize(s));
            GBRS_HIP_CHECK(hipGetLastError());
            // Worth it?  A set entry costs its tile a longer prologue and epilogue (its members' theta summed, its sums
            // stored once per member), which pays when the words it saves are many and every dictionary entry serves many
            // words.  Measured (profiles/r03_estep_experiments.txt item 8): C2, 23 % fewer words at 137 words per id:
            // E-step -11 %; the 16-haplotype shard (same saving, 62 words per id) +8 %; multi-isoform reads (9 % fewer
            // words) +8 %.  GBRS_TUNING_LOCUS_SETS=1 / 0 forces the choice.
            // (and no more sets than twice the loci: many thin sets fill the tiles' dictionaries, item 16 of the same file)
            bool use = (uint64_t)P2 * 100 <= (uint64_t)P * 85 && (uint64_t)P2 >= 100ull * ((uint64_t)L_in + V) && (uint64_t)V <= 2ull * L_in;
            if (const char *env = std::getenv("GBRS_TUNING_LOCUS_SETS"); env) use = std::atoi(env) != 0;
            if (use) {
                rowstart.swap(rowstart2); ploc.swap(ploc2); pmask.swap(pmask2);
                out.n_sets = V;
                out.n_pairs = P2;
                L = L_in + V;              // ids of the rows' pairs, the sort keys and the dictionaries from here on
            } else {
                out.set_ptr.release();
                out.set_members.release();
            }
            }   // ids_fit
        }
        stg.mark("3b locus sets");
    }
    // 3c. locus sets per mask group, when the rows are no whole-row sets (reads over several isoforms with differing masks):
    // the loci of a row that share a mask become one pair on a set id, for the sets that enough rows carry
    if (locus_sets && R1 > 0 && out.n_sets == 0) {
        const char *genv = std::getenv("GBRS_TUNING_GROUP_SETS");
        const bool forced_on = genv && std::atoi(genv) == 1, forced_off = genv && std::atoi(genv) == 0;
        // rows a set must be carried by.  Iteration time, one box each: multi-isoform sample (2.35 words per read, no sets 0.1970 ms):
        // 64 rows (39 k sets) 0.199, 128 (21 k) 0.188, 160: 0.186, 192 (13.8 k) 0.184, 256 (10 k) 0.183, 384 / 512: 0.184;
        // C2 (whole-row sets, all 98.7 k of them 0.0901 ms): 32 rows (42.5 k) 0.0858, 128 (17 k) 0.0857, 512 (4.3 k) 0.0881
        uint32_t min_rows = 192;
        if (const char *e = std::getenv("GBRS_TUNING_SET_MIN_ROWS"); e && std::atoi(e) > 0) min_rows = (uint32_t)std::atoi(e);
        DevBuf<uint32_t> gloc, gmask, nseg, segoff;
        if (!forced_off) {
            GBRS_TRY(gloc.alloc(P)); GBRS_TRY(gmask.alloc(P)); GBRS_TRY(nseg.alloc(R1)); GBRS_TRY(segoff.alloc((size_t)R1 + 1));
            hipLaunchKernelGGL(group_sort_kernel, dim3(grid_for(R1)), dim3(256), 0, s, (uint64_t)R1, rowstart.p, ploc.p, pmask.p,
                               gloc.p, gmask.p, nseg.p);
            GBRS_TRY(exclusive_scan(sc, nseg.p, segoff.p, R1, s));
            uint32_t S = 0;
            GBRS_TRY(fetch_last_plus(segoff.p, nseg.p, R1, S, s));
            GBRS_HIP_CHECK(hipMemcpyAsync(segoff.p + R1, &S, 4, hipMemcpyHostToDevice, s));
            DevBuf<uint32_t> seg_begin, seg_len, cand, cidx;
            DevBuf<uint64_t> key;
            GBRS_TRY(seg_begin.alloc(S)); GBRS_TRY(seg_len.alloc(S)); GBRS_TRY(cand.alloc(S)); GBRS_TRY(cidx.alloc(S));
            GBRS_TRY(key.alloc(S));
            hipLaunchKernelGGL(group_segments_kernel, dim3(grid_for(R1)), dim3(256), 0, s, (uint64_t)R1, rowstart.p, gloc.p, gmask.p,
                               segoff.p, seg_begin.p, seg_len.p, key.p, cand.p);
            GBRS_TRY(exclusive_scan(sc, cand.p, cidx.p, S, s));
            uint32_t C = 0;
            GBRS_TRY(fetch_last_plus(cidx.p, cand.p, S, C, s));
            if (C > 0 && S < P) {
                DevBuf<uint64_t> ckey, skey2;
                DevBuf<uint32_t> cseg, sseg, head2, hincl2;
                GBRS_TRY(ckey.alloc(C)); GBRS_TRY(skey2.alloc(C)); GBRS_TRY(cseg.alloc(C)); GBRS_TRY(sseg.alloc(C));
                hipLaunchKernelGGL(group_compact_kernel, dim3(grid_for(S)), dim3(256), 0, s, (uint64_t)S, cand.p, cidx.p, key.p,
                                   ckey.p, cseg.p);
                GBRS_TRY(sort_pairs<uint64_t>(sc, ckey.p, skey2.p, cseg.p, sseg.p, C, 64, s));
                key.release(); ckey.release(); cseg.release(); cand.release(); cidx.release();
                GBRS_TRY(head2.alloc(C)); GBRS_TRY(hincl2.alloc(C));
                hipLaunchKernelGGL(group_head_kernel, dim3(grid_for(C)), dim3(256), 0, s, (uint64_t)C, skey2.p, sseg.p, seg_begin.p,
                                   seg_len.p, gloc.p, head2.p);
                GBRS_TRY(inclusive_scan(sc, head2.p, hincl2.p, C, s));
                uint32_t Vp = 0;
                GBRS_HIP_CHECK(hipMemcpyAsync(&Vp, hincl2.p + C - 1, 4, hipMemcpyDeviceToHost, s));
                GBRS_HIP_CHECK(hipStreamSynchronize(s));
                DevBuf<uint32_t> cnt, rep, keep, newid;
                GBRS_TRY(cnt.alloc(Vp)); GBRS_TRY(rep.alloc(Vp)); GBRS_TRY(keep.alloc(Vp)); GBRS_TRY(newid.alloc(Vp));
                GBRS_HIP_CHECK(hipMemsetAsync(cnt.p, 0, cnt.bytes(), s));
                hipLaunchKernelGGL(group_count_kernel, dim3(grid_for(C)), dim3(256), 0, s, (uint64_t)C, head2.p, hincl2.p, sseg.p,
                                   cnt.p, rep.p);
                // the frequent sets only, and no more of them than half the loci: the threshold doubles until they fit
                uint32_t V = 0;
                for (int round = 0; round < 24; ++round) {
                    hipLaunchKernelGGL(group_keep_kernel, dim3(grid_for(Vp)), dim3(256), 0, s, Vp, min_rows, cnt.p, keep.p);
                    GBRS_TRY(exclusive_scan(sc, keep.p, newid.p, Vp, s));
                    GBRS_TRY(fetch_last_plus(newid.p, keep.p, Vp, V, s));
                    if ((uint64_t)V * 2 <= (uint64_t)L_in || forced_on) break;
                    min_rows *= 2;
                }
                if (V > 0 && (uint64_t)L_in + V < (1u << 27)) {
                    DevBuf<uint32_t> set_of_seg, set_len, set_rep, newlen, rowstart2, ploc2, pmask2;
                    GBRS_TRY(set_of_seg.alloc(S)); GBRS_TRY(set_len.alloc(V)); GBRS_TRY(set_rep.alloc(V));
                    GBRS_HIP_CHECK(hipMemsetAsync(set_of_seg.p, 0xFF, set_of_seg.bytes(), s));
                    hipLaunchKernelGGL(group_assign_kernel, dim3(grid_for(C)), dim3(256), 0, s, (uint64_t)C, hincl2.p, sseg.p, keep.p,
                                       newid.p, rep.p, seg_len.p, set_of_seg.p, set_len.p, set_rep.p);
                    GBRS_TRY(out.set_ptr.alloc((size_t)V + 1));
                    GBRS_TRY(exclusive_scan(sc, set_len.p, out.set_ptr.p, V, s));
                    uint32_t n_members = 0;
                    GBRS_TRY(fetch_last_plus(out.set_ptr.p, set_len.p, V, n_members, s));
                    GBRS_HIP_CHECK(hipMemcpyAsync(out.set_ptr.p + V, &n_members, 4, hipMemcpyHostToDevice, s));
                    GBRS_TRY(out.set_members.alloc(n_members));
                    hipLaunchKernelGGL(group_members_kernel, dim3(grid_for(V)), dim3(256), 0, s, V, out.set_ptr.p, set_rep.p,
                                       seg_begin.p, gloc.p, out.set_members.p);
                    GBRS_TRY(newlen.alloc(R1)); GBRS_TRY(rowstart2.alloc((size_t)R1 + 1));
                    hipLaunchKernelGGL(group_row_len_kernel, dim3(grid_for(R1)), dim3(256), 0, s, (uint64_t)R1, segoff.p, seg_len.p,
                                       set_of_seg.p, newlen.p);
                    GBRS_TRY(exclusive_scan(sc, newlen.p, rowstart2.p, R1, s));
                    uint32_t P2 = 0;
                    GBRS_TRY(fetch_last_plus(rowstart2.p, newlen.p, R1, P2, s));
                    GBRS_HIP_CHECK(hipMemcpyAsync(rowstart2.p + R1, &P2, 4, hipMemcpyHostToDevice, s));
                    GBRS_TRY(ploc2.alloc(P2)); GBRS_TRY(pmask2.alloc(P2));
                    hipLaunchKernelGGL(group_rewrite_kernel, dim3(grid_for(R1)), dim3(256), 0, s, (uint64_t)R1, L_in, segoff.p,
                                       seg_begin.p, seg_len.p, set_of_seg.p, rowstart2.p, gloc.p, gmask.p, ploc2.p, pmask2.p);
                    GBRS_HIP_CHECK(hipStreamSynchronize(s));
                    GBRS_HIP_CHECK(hipGetLastError());
                    // worth it when the frequent sets take a twentieth of the words away (GBRS_TUNING_GROUP_SETS=1 / 0 forces the choice)
                    const bool use = forced_on || (uint64_t)P2 * 100 <= (uint64_t)P * 95;
                    if (use) {
                        rowstart.swap(rowstart2); ploc.swap(ploc2); pmask.swap(pmask2);
                        out.n_sets = V;
                        out.n_pairs = P2;
                        L = L_in + V;
                    } else {
                        out.set_ptr.release();
                        out.set_members.release();
                    }
                }
            }
        }
        stg.mark("3c mask-group sets");
    }
    // 4. order rows so that similar rows are adjacent
    DevBuf<uint64_t> rkey, skey;
    DevBuf<uint32_t> ident, srow;
    GBRS_TRY(rkey.alloc(R1)); GBRS_TRY(skey.alloc(R1)); GBRS_TRY(ident.alloc(R1)); GBRS_TRY(srow.alloc(R1));
    const unsigned lbits = bits_for(L - 1);
    hipLaunchKernelGGL(row_key_kernel, dim3(grid_for(R1)), dim3(256), 0, s, (uint64_t)R1, (uint32_t)max_row_words(H),
                       lbits > 24 ? lbits - 24 : 0u,
                       rowstart.p, ploc.p, pmask.p, rkey.p, ident.p, d_flags.p);
    GBRS_TRY(sort_pairs<uint64_t>(sc, rkey.p, skey.p, ident.p, srow.p, R1, 64, s));
    GBRS_TRY(read_flags());
    rkey.release(); ident.release();
    const uint64_t n_long = hf.n_long, n_short = R1 - n_long;
    out.n_long = n_long;
    stg.mark("4 order rows");
    // 5. optional merge of identical adjacent rows + weights
    DevBuf<uint32_t> head, hincl, hrow;
    uint64_t M = n_short;
    if (n_short) {
        GBRS_TRY(head.alloc(n_short)); GBRS_TRY(hincl.alloc(n_short));
        hipLaunchKernelGGL(merge_flag_kernel, dim3(grid_for(n_short)), dim3(256), 0, s, n_short, merge ? 1 : 0, skey.p,
                           srow.p, rowstart.p, ploc.p, pmask.p, head.p);
        GBRS_TRY(inclusive_scan(sc, head.p, hincl.p, n_short, s));
        uint32_t m32 = 0;
        GBRS_HIP_CHECK(hipMemcpyAsync(&m32, hincl.p + n_short - 1, 4, hipMemcpyDeviceToHost, s));
        GBRS_HIP_CHECK(hipStreamSynchronize(s));
        M = m32;
        GBRS_TRY(hrow.alloc(M));
        if (out.weighted) {
            GBRS_TRY(out.row_weight.alloc(M));
            GBRS_HIP_CHECK(hipMemsetAsync(out.row_weight.p, 0, out.row_weight.bytes(), s));
        }
        hipLaunchKernelGGL(merged_rows_kernel, dim3(grid_for(n_short)), dim3(256), 0, s, n_short, head.p, hincl.p, srow.p,
                           row_orig.p, count, hrow.p, out.weighted ? out.row_weight.p : nullptr);
        GBRS_HIP_CHECK(hipStreamSynchronize(s));
        head.release(); hincl.release();
    }
    skey.release();
    out.n_rows = M;
    out.all_one_word = false;
    stg.mark("5 merge");
    // 6. long rows keep their pair form
    if (n_long) {
        DevBuf<uint64_t> llen;
        GBRS_TRY(llen.alloc(n_long));
        GBRS_TRY(out.long_ptr.alloc(n_long + 1));
        GBRS_TRY(out.long_weight.alloc(n_long));
        hipLaunchKernelGGL(long_rows_kernel, dim3(grid_for(n_long)), dim3(256), 0, s, n_long, n_short, srow.p, rowstart.p,
                           row_orig.p, count, llen.p, out.long_weight.p);
        GBRS_TRY(exclusive_scan(sc, llen.p, out.long_ptr.p, n_long, s));
        uint64_t tot = 0;
        GBRS_TRY(fetch_last_plus(out.long_ptr.p, llen.p, n_long, tot, s));
        GBRS_HIP_CHECK(hipMemcpyAsync(out.long_ptr.p + n_long, &tot, 8, hipMemcpyHostToDevice, s));
        GBRS_TRY(out.long_loc.alloc(tot));
        GBRS_TRY(out.long_mask.alloc(tot));
        hipLaunchKernelGGL(long_copy_kernel, dim3((unsigned)n_long), dim3(64), 0, s, n_long, n_short, srow.p, rowstart.p,
                           ploc.p, pmask.p, out.long_ptr.p, out.long_loc.p, out.long_mask.p);
        GBRS_TRY(out.acc_extra.alloc((size_t)L_in * H));
        GBRS_HIP_CHECK(hipStreamSynchronize(s));
    }
    srow.release(); row_orig.release();
    if (M == 0) { GBRS_HIP_CHECK(hipStreamSynchronize(s)); return GBRS_OK; }
    stg.mark("6 long rows");
    // 7. tiles
    DevBuf<uint32_t> npm, dnew, wordoff, dincl, tflag, tincl;
    GBRS_TRY(npm.alloc(M)); GBRS_TRY(dnew.alloc(M)); GBRS_TRY(wordoff.alloc(M)); GBRS_TRY(dincl.alloc(M));
    GBRS_TRY(tflag.alloc(M)); GBRS_TRY(tincl.alloc(M));
    hipLaunchKernelGGL(row_len_kernel, dim3(grid_for(M)), dim3(256), 0, s, M, hrow.p, rowstart.p, ploc.p, npm.p, dnew.p);
    GBRS_TRY(exclusive_scan(sc, npm.p, wordoff.p, M, s));
    GBRS_TRY(inclusive_scan(sc, dnew.p, dincl.p, M, s));
    // tile size: as large as still leaves TILE_ROUNDS_MIN rounds of the chip's resident E-step workgroups (3 or 2 per CU),
    // between TILE_WORDS and TILE_WORDS_MAX (em_layout.h); GBRS_TUNING_TILE_WORDS overrides
    uint32_t tile_words = TILE_WORDS;
    {
        uint32_t total_words = 0;
        GBRS_TRY(fetch_last_plus(wordoff.p, npm.p, M, total_words, s));
        out.all_one_word = total_words == M && n_long == 0;
        int dev = 0, n_cu = 0;
        GBRS_HIP_CHECK(hipGetDevice(&dev));
        GBRS_HIP_CHECK(hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev));
        // (handles that run side by side - the locus ranges of one sample, GBRS_EM_SIDE_BY_SIDE - fill the rounds together)
        const unsigned per_cu = (out.weighted || H > 8) ? 2u : 3u;      // resident E-step workgroups per CU (tile_estep_kernel's launch bounds)
        const uint64_t fit = (uint64_t)total_words * side_by_side / ((uint64_t)TILE_ROUNDS_MIN * per_cu * (uint64_t)std::max(n_cu, 1));
        // (weighted rows - merged reads, EC counts - stay at 16,320: the merged C2 sample reads 0.0345 ms there, 0.0357 at 20,900)
        const uint64_t cap = out.weighted ? std::min<uint64_t>(TILE_WORDS_MAX, 16320) : (uint64_t)TILE_WORDS_MAX;
        tile_words = (uint32_t)std::min<uint64_t>(std::max<uint64_t>(fit, TILE_WORDS), cap) & ~63u;
        if (const char *env = std::getenv("GBRS_TUNING_TILE_WORDS"); env && std::atoi(env) >= 64)
            tile_words = (uint32_t)std::min(std::atoi(env), GBRS_TILE_CAP - 64);
    }
    hipLaunchKernelGGL(tile_flag_kernel, dim3(grid_for(M)), dim3(256), 0, s, M, tile_words, dseg, npm.p,
                       wordoff.p, dincl.p, tflag.p);
    GBRS_TRY(inclusive_scan(sc, tflag.p, tincl.p, M, s));
    uint32_t T32 = 0;
    GBRS_HIP_CHECK(hipMemcpyAsync(&T32, tincl.p + M - 1, 4, hipMemcpyDeviceToHost, s));
    GBRS_HIP_CHECK(hipStreamSynchronize(s));
    const uint64_t T = T32;
    out.n_tiles = T;
    dincl.release();
    DevBuf<uint32_t> tile_row;
    GBRS_TRY(tile_row.alloc(T + 1));
    hipLaunchKernelGGL(tile_start_kernel, dim3(grid_for(M)), dim3(256), 0, s, M, T, tflag.p, tincl.p, tile_row.p);
    stg.mark("7 tiles");
    // 7b. interleave the locus lists inside each tile (tile membership and sizes are unchanged)
    if (interleave || streams) {
        DevBuf<uint32_t> gflag, gpos, gstart, gord, ident, perm, hrow2, npm2;
        DevBuf<uint64_t> ikey, ikey2;
        DevBuf<double> weight2;
        if (interleave) {
            GBRS_TRY(gflag.alloc(M)); GBRS_TRY(gpos.alloc(M)); GBRS_TRY(gstart.alloc(M)); GBRS_TRY(gord.alloc(M));
            hipLaunchKernelGGL(group_flag_kernel, dim3(grid_for(M)), dim3(256), 0, s, M, dnew.p, tflag.p, gflag.p, gpos.p);
            {
                size_t bytes = 0;
                GBRS_PRIM(rocprim::inclusive_scan(nullptr, bytes, gpos.p, gstart.p, (size_t)M, rocprim::maximum<uint32_t>(), s));
                GBRS_TRY(sc.reserve(bytes));
                GBRS_PRIM(rocprim::inclusive_scan(sc.buf.p, bytes, gpos.p, gstart.p, (size_t)M, rocprim::maximum<uint32_t>(), s));
            }
            GBRS_TRY(inclusive_scan(sc, gflag.p, gord.p, M, s));
            GBRS_HIP_CHECK(hipStreamSynchronize(s));
            gflag.release(); gpos.release();
        }
        GBRS_TRY(ikey.alloc(M)); GBRS_TRY(ikey2.alloc(M)); GBRS_TRY(ident.alloc(M)); GBRS_TRY(perm.alloc(M));
        stg.mark("7b-a alloc");
        if (interleave)
            hipLaunchKernelGGL(interleave_key_kernel, dim3(grid_for(M)), dim3(256), 0, s, M, tincl.p, gstart.p, gord.p,
                               ikey.p, ident.p);
        else
            hipLaunchKernelGGL(stream_key_kernel, dim3(grid_for(M)), dim3(256), 0, s, M, tincl.p, npm.p, ikey.p, ident.p);
        stg.mark("7b-b keys");
        GBRS_TRY(sort_pairs<uint64_t>(sc, ikey.p, ikey2.p, ident.p, perm.p, M, 40 + bits_for(T), s));
        GBRS_HIP_CHECK(hipStreamSynchronize(s));
        stg.mark("7b-c sort");
        ikey.release(); ikey2.release(); ident.release(); gstart.release(); gord.release();
        stg.mark("7b-d release");
        GBRS_TRY(hrow2.alloc(M)); GBRS_TRY(npm2.alloc(M));
        if (out.weighted) GBRS_TRY(weight2.alloc(M));
        hipLaunchKernelGGL(permute_rows_kernel, dim3(grid_for(M)), dim3(256), 0, s, M, perm.p, hrow.p, npm.p,
                           out.weighted ? out.row_weight.p : (const double *)nullptr, hrow2.p, npm2.p,
                           out.weighted ? weight2.p : (double *)nullptr);
        GBRS_HIP_CHECK(hipMemcpyAsync(hrow.p, hrow2.p, M * 4, hipMemcpyDeviceToDevice, s));
        GBRS_HIP_CHECK(hipMemcpyAsync(npm.p, npm2.p, M * 4, hipMemcpyDeviceToDevice, s));
        if (out.weighted)
            GBRS_HIP_CHECK(hipMemcpyAsync(out.row_weight.p, weight2.p, M * 8, hipMemcpyDeviceToDevice, s));
        GBRS_TRY(exclusive_scan(sc, npm.p, wordoff.p, M, s));
        GBRS_HIP_CHECK(hipStreamSynchronize(s));
    }
    dnew.release();
    tflag.release();
    stg.mark("7b row order");
    // 8. padding so that no row straddles a batch, batch offsets
    DevBuf<uint32_t> rowpad, nbatch, batch_base;
    GBRS_TRY(rowpad.alloc(M)); GBRS_TRY(nbatch.alloc(T)); GBRS_TRY(batch_base.alloc(T));
    hipLaunchKernelGGL(tile_pad_kernel, dim3(grid_for(T, 64)), dim3(64), 0, s, T, streams ? 1 : 0, tile_row.p, npm.p, rowpad.p,
                       nbatch.p);
    GBRS_TRY(exclusive_scan(sc, nbatch.p, batch_base.p, T, s));
    uint32_t NB = 0;
    GBRS_TRY(fetch_last_plus(batch_base.p, nbatch.p, T, NB, s));
    out.n_batches = NB;
    stg.mark("8 padding");
    // 9. per-tile dictionaries: one sort of (tile, id) keys over all the pairs (see tile_pair_keys_kernel)
    const uint32_t dcap = out.d_max;
    uint32_t NS = 0;
    DevBuf<uint32_t> dict_base;
    GBRS_TRY(dict_base.alloc(T));
    {
        uint32_t W = 0;
        GBRS_TRY(fetch_last_plus(wordoff.p, npm.p, M, W, s));
        DevBuf<uint64_t> dkey, dkey2;
        DevBuf<uint32_t> dflag, dpos;
        GBRS_TRY(dkey.alloc(W)); GBRS_TRY(dkey2.alloc(W)); GBRS_TRY(dflag.alloc(W)); GBRS_TRY(dpos.alloc(W));
        hipLaunchKernelGGL(tile_pair_keys_kernel, dim3(grid_for(M)), dim3(256), 0, s, M, tincl.p, hrow.p, rowstart.p, ploc.p,
                           wordoff.p, dkey.p);
        GBRS_TRY(sort_keys64(sc, dkey.p, dkey2.p, W, 32 + bits_for(T), s));
        hipLaunchKernelGGL(dict_flag_kernel, dim3(grid_for(W)), dim3(256), 0, s, (uint64_t)W, dkey2.p, dflag.p);
        GBRS_TRY(exclusive_scan(sc, dflag.p, dpos.p, W, s));
        GBRS_TRY(fetch_last_plus(dpos.p, dflag.p, W, NS, s));
        out.n_slots = NS;
        GBRS_TRY(out.tiles.alloc(T));
        GBRS_TRY(out.dict.alloc(std::max<uint32_t>(NS, 1)));
        hipLaunchKernelGGL(dict_emit_kernel, dim3(grid_for(W)), dim3(256), 0, s, (uint64_t)W, dkey2.p, dflag.p, dpos.p, out.dict.p,
                           dict_base.p);
        hipLaunchKernelGGL(tile_hdr_kernel, dim3(grid_for(T)), dim3(256), 0, s, T, dcap, NS, batch_base.p, nbatch.p, dict_base.p,
                           out.tiles.p, d_flags.p);
        GBRS_TRY(read_flags());
        if (hf.dict_overflow) return fail(GBRS_ERR_INVALID, "internal error: a tile dictionary overflowed its capacity");
    }
    stg.mark("9 dictionaries");
    // 10. words
    GBRS_TRY(out.words.alloc((size_t)NB * 64));
    GBRS_HIP_CHECK(hipMemsetAsync(out.words.p, 0, out.words.bytes(), s));
    if (out.weighted) {
        GBRS_TRY(out.word_weight.alloc((size_t)NB * 64));
        GBRS_HIP_CHECK(hipMemsetAsync(out.word_weight.p, 0, out.word_weight.bytes(), s));
    }
    hipLaunchKernelGGL(emit_words_kernel, dim3(grid_for(M)), dim3(256), 0, s, M, H, tincl.p, out.tiles.p, out.dict.p,
                       hrow.p, rowstart.p, ploc.p, pmask.p, rowpad.p, out.words.p,
                       out.weighted ? out.row_weight.p : (const double *)nullptr,
                       out.weighted ? out.word_weight.p : (double *)nullptr);
    GBRS_HIP_CHECK(hipStreamSynchronize(s));
    dict_base.release(); batch_base.release(); nbatch.release();
    rowpad.release(); tincl.release(); npm.release(); wordoff.release(); tile_row.release();
    hrow.release(); rowstart.release(); ploc.release(); pmask.release();
    out.row_weight.release();
    // Launch order of the tiles: the ones with the most batches first, so that the launch's last round - when most
    // of the chip has run out of tiles - is made of the short ones (tiles that end at the dictionary limit have
    // fewer words; C2: E-step 0.0998-0.1008 -> 0.0960-0.0962 ms).  Nothing but the E-step reads the header array by
    // position.  GBRS_TUNING_TILE_ORDER=0 keeps the locus order.
    const char *order_env = std::getenv("GBRS_TUNING_TILE_ORDER");
    if (!(order_env && std::atoi(order_env) == 0) && T > 1) {
        std::vector<TileHdr> hdr(T);
        GBRS_HIP_CHECK(hipMemcpy(hdr.data(), out.tiles.p, T * sizeof(TileHdr), hipMemcpyDeviceToHost));
        std::stable_sort(hdr.begin(), hdr.end(), [](const TileHdr &a, const TileHdr &b) { return a.n_batches > b.n_batches; });
        GBRS_HIP_CHECK(hipMemcpy(out.tiles.p, hdr.data(), T * sizeof(TileHdr), hipMemcpyHostToDevice));
    }
    stg.mark("10 words");
    // 11. inverted index locus -> destination rows.  A slot (tile, dictionary entry) of a locus delivers one row of sums
    // to that locus; the slot of a locus set delivers the same row to every member locus.  The rows are numbered by
    // locus (ascending slot inside a locus: the radix sort is stable), so that the rows of a locus are consecutive in
    // `partials` and the gather streams them without an indirection.
    GBRS_TRY(out.slot_dest.alloc(std::max<uint32_t>(NS, 1)));
    uint32_t NE = 0, n_rows_real = 0;
    if (NS) {
        DevBuf<uint32_t> ecnt, eoff;
        GBRS_TRY(ecnt.alloc(NS)); GBRS_TRY(eoff.alloc(NS));
        hipLaunchKernelGGL(dest_count_kernel, dim3(grid_for(NS)), dim3(256), 0, s, (uint64_t)NS, L_in, out.dict.p,
                           out.n_sets ? out.set_ptr.p : (const uint32_t *)nullptr, ecnt.p);
        GBRS_TRY(exclusive_scan(sc, ecnt.p, eoff.p, NS, s));
        GBRS_TRY(fetch_last_plus(eoff.p, ecnt.p, NS, NE, s));
        // (SLOT_PAIR is the lowest flag bit of a destination: checked here, before anything is sized by NE or indexed with it)
        if (NE >= SLOT_PAIR) return fail(GBRS_ERR_INVALID, "the tiled layout needs fewer than 2^29 destination rows");
        DevBuf<uint32_t> eloc, eidx, sloc;
        GBRS_TRY(eloc.alloc(NE)); GBRS_TRY(eidx.alloc(NE)); GBRS_TRY(sloc.alloc(NE));
        GBRS_TRY(out.slot_list.alloc(NE));
        GBRS_TRY(out.dest_list.alloc(NE));
        hipLaunchKernelGGL(dest_emit_kernel, dim3(grid_for(NS)), dim3(256), 0, s, (uint64_t)NS, L_in, out.dict.p,
                           out.n_sets ? out.set_ptr.p : (const uint32_t *)nullptr, out.set_members.p, eoff.p, eloc.p, eidx.p,
                           out.slot_dest.p, out.dest_list.p);
        // (the header entry of a set slot carries locus id L_in: behind every real locus, never looked at)
        GBRS_TRY(sort_pairs<uint32_t>(sc, eloc.p, sloc.p, eidx.p, out.slot_list.p, NE, bits_for(L_in), s));
        hipLaunchKernelGGL(slot_ptr_kernel, dim3(grid_for((uint64_t)L_in + 1)), dim3(256), 0, s, L_in, (uint64_t)NE, sloc.p,
                           out.slot_ptr.p);
        // the rows that receive sums: the entries of real loci (a set slot's header entry sorts behind them)
        GBRS_HIP_CHECK(hipMemcpyAsync(&n_rows_real, out.slot_ptr.p + L_in, 4, hipMemcpyDeviceToHost, s));
        GBRS_HIP_CHECK(hipStreamSynchronize(s));
    } else {
        GBRS_TRY(out.slot_list.alloc(1));
        GBRS_TRY(out.dest_list.alloc(1));
    }
    out.n_dest_rows = n_rows_real;
    GBRS_TRY(out.partials.alloc(std::max<size_t>((size_t)n_rows_real * H, 1)));
    hipLaunchKernelGGL(locus_class_kernel, dim3(grid_for(L_in)), dim3(256), 0, s, L_in, out.slot_ptr.p, out.slot_list.p,
                       out.locus_class.p, out.dest_list.p);
    if (NS) hipLaunchKernelGGL(dest_route_kernel, dim3(grid_for(NS)), dim3(256), 0, s, (uint64_t)NS, out.dest_list.p, out.slot_dest.p);
    if (out.n_sets && NS) {
        GBRS_TRY(out.dict_b.alloc(NS));
        GBRS_TRY(out.dest_b.alloc(NS));
        hipLaunchKernelGGL(encode_sets_kernel, dim3(grid_for(NS)), dim3(256), 0, s, (uint64_t)NS, L_in, out.set_ptr.p,
                           out.set_members.p, out.dest_list.p, out.dict.p, out.dict_b.p, out.slot_dest.p, out.dest_b.p);
    }
    stg.mark("11 inverted index");
    // 12. loci with many slots get a whole wave in the gather kernel
    {
        std::vector<uint32_t> sp((size_t)L_in + 1), heavy, lightv;
        GBRS_HIP_CHECK(hipMemcpyAsync(sp.data(), out.slot_ptr.p, sp.size() * 4, hipMemcpyDeviceToHost, s));
        GBRS_HIP_CHECK(hipStreamSynchronize(s));
        for (uint32_t l = 0; l < L_in; ++l)
            if (sp[l + 1] - sp[l] > (uint32_t)HEAVY_SLOTS) heavy.push_back(l);
            else if (sp[l + 1] - sp[l] >= 2) lightv.push_back(l);
        out.n_heavy = heavy.size();
        out.n_light = lightv.size();
        for (int which = 0; which < 2; ++which) {
            const std::vector<uint32_t> &loci = which == 0 ? heavy : lightv;
            DevBuf<uint32_t> &dst = which == 0 ? out.heavy_range : out.light_range;
            std::vector<uint32_t> rng;
            rng.reserve(loci.size() * 3);
            for (uint32_t l : loci) {
                rng.push_back(l);
                rng.push_back(sp[l]);
                rng.push_back(sp[l + 1]);
            }
            GBRS_TRY(dst.alloc(std::max<size_t>(rng.size(), 3)));
            if (!rng.empty()) GBRS_HIP_CHECK(hipMemcpyAsync(dst.p, rng.data(), rng.size() * 4, hipMemcpyHostToDevice, s));
            GBRS_HIP_CHECK(hipStreamSynchronize(s));          // rng goes out of scope
        }
        GBRS_TRY(out.light_loci.alloc(std::max<size_t>(lightv.size(), 1)));
        if (!lightv.empty())
            GBRS_HIP_CHECK(hipMemcpyAsync(out.light_loci.p, lightv.data(), lightv.size() * 4, hipMemcpyHostToDevice, s));
        GBRS_TRY(out.heavy_loci.alloc(std::max<size_t>(heavy.size(), 1)));
        if (!heavy.empty())
            GBRS_HIP_CHECK(hipMemcpyAsync(out.heavy_loci.p, heavy.data(), heavy.size() * 4, hipMemcpyHostToDevice, s));
        GBRS_HIP_CHECK(hipStreamSynchronize(s));
    }
    stg.mark("12 heavy / light lists");
    GBRS_HIP_CHECK(hipGetLastError());
return GBRS_OK;
}

}  // namespace gbrs

// gbrs_warm_up (common.hip): loads this file's code object
namespace gbrs {
__global__ void warm_layout_kernel() {}
void warm_layout(hipStream_t st) { hipLaunchKernelGGL(warm_layout_kernel, dim3(1), dim3(64), 0, st); }
}  // namespace gbrs
