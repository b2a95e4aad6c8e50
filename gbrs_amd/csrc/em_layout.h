// Device layout of the alignment incidence tensor for the tiled E-step ("packed row tiles").
// Built once per handle by build_tile_layout() (em_layout.hip) from the reference's CSC arrays.
//
//   word   = one (row, locus) pair of a read: [ local locus index | rem | pos | hap mask ]
//            bits [0,H) mask; PB bits pos = words of the row before this one; PB bits rem = words
//            of the row after it; the rest = index into the tile's dictionary.  PB = 5 (rows of up
//            to 32 loci) for H <= 8, 4 (16 loci) for H <= 16.
//   batch  = 64 consecutive words = what one wavefront takes per step; rows never straddle a
//            batch (zero words pad the tail), so a batch is self-contained.
//   tile   = a run of batches processed by one workgroup with one locus dictionary of at most
//            D_MAX loci, so that theta and the partial sums of the tile live in LDS.
//   slot   = one (tile, dictionary entry): the tile's partial sum for that locus, H doubles in
//            `partials`.  Slots are numbered by locus (`slot_dest` maps a tile's dictionary entry to
//            its row; `slot_ptr[l] .. slot_ptr[l+1]` are the rows of locus l), so the gather streams
//            consecutive rows in fixed order (no float atomics in global memory, bit-reproducible
//            across launches up to LDS atomic order inside a tile).  `slot_list` (build only) is the
//            inverted index in the tile-major numbering.
#pragma once
#include "common.h"

#include <vector>

namespace gbrs {

#ifndef GBRS_TILE_THREADS
#define GBRS_TILE_THREADS 512
#endif
constexpr int TILE_THREADS = GBRS_TILE_THREADS;   // 8 waves per workgroup
constexpr int TILE_WAVES = TILE_THREADS / 64;
#ifndef GBRS_TILE_CAP
#define GBRS_TILE_CAP 32768                    // most words (incl. padding) a tile may hold.  Round 4: the dictionaries come from one
#endif                                         // global sort (the per-tile LDS sort of rounds 1-3 held 16,384), so the cap is a choice:
                                               // C2, same box: 16,320 words 0.0751-0.0753 ms, 20,800: 0.0745-0.0747, 26,048: 0.0730-0.0735,
                                               // 32,704: 0.0729-0.0742, 52,096 (one round): 0.0749-0.0750, 65,472: 0.0747-0.0749
// Unpadded words per tile.  Larger tiles spread the tile prologue / epilogue (dictionary, theta gather, flush of the
// sums) over more words, but a launch needs several rounds of tiles on the chip's resident workgroups to hide
// its tail.  Measured on the C2 sample (52.4 M words; profiles/r02_estep_experiments.txt items 10, 12, 13): in locus
// order 5,800 words 0.1101 ms, 8,128: 0.1020, 10,000-12,000: 0.0994-0.0999, 16,320 (5.1 rounds): 0.1015; with the
// tiles launched largest first 11,008: 0.0949, 13,000: 0.0937, 14,500: 0.0933, 16,320: 0.0927; the merged-rows form
// of the same sample (10.7 M words) is slower with tiles above 8,128 words (1.5 rounds instead of 1.7).
// With the largest tiles first even the merged-rows form prefers the largest tiles (8,128: 0.0465-0.0469 ms per iteration,
// 11,008: 0.0464, 16,320: 0.0449 at 1.2 rounds; 6,016: 0.0476, 4,096: 0.0501, 2,816: 0.0528), so what is left of the
// rule is that a launch should not have fewer tiles than the chip has places for them: build_tile_layout takes the
// largest size between TILE_WORDS and TILE_WORDS_MAX that leaves TILE_ROUNDS_MIN round(s).
#ifndef GBRS_TILE_WORDS
#define GBRS_TILE_WORDS 2048
#endif
constexpr int TILE_WORDS = GBRS_TILE_WORDS, TILE_WORDS_MAX = GBRS_TILE_CAP - 64, TILE_ROUNDS_MIN = 1;
static_assert(TILE_WORDS <= TILE_WORDS_MAX, "a tile's padded words must fit the dictionary sort");
// rows with more distinct loci than this go to the long-row path
// (more than 8 haplotypes: 3 bits - rows of up to 8 loci in the tiles - leave 10 bits of dictionary index beside a 16-bit
// mask; with 4 + 4 bits the index had 8, and 256 loci per tile is less than the 16-haplotype kernel's LDS holds)
__host__ __device__ constexpr int pos_bits(int H) { return H <= 8 ? 5 : 3; }
__host__ __device__ constexpr int max_row_words(int H) { return 1 << pos_bits(H); }
#ifndef GBRS_LDS_DOUBLES
#define GBRS_LDS_DOUBLES 3072
#endif
#ifndef GBRS_LDS_DOUBLES_WEIGHTED
#define GBRS_LDS_DOUBLES_WEIGHTED 4608         // round 4: 4,096 -> 4,608 (74 KB per workgroup, two per CU): merged C2 rows 0.0441 -> 0.0435 ms
#endif
// theta of the tile (D_MAX * H doubles) and, 64 doubles larger, its privatised partial sums.  Unweighted
// layouts use 3,072 doubles (49 KB per workgroup with the sums: three workgroups per CU, which the
// 72-register unweighted kernel fills); the weighted kernels need ~120 registers, two workgroups per CU
// is all they can have, so their tiles take the larger dictionaries.
// 16 haplotypes, unweighted: the kernel needs ~126 registers, two workgroups per CU is all it can have, and a locus takes 16
// doubles - with 3,072 doubles a tile held 192 loci and the config-5 shard's tiles 9 k words, half of them prologue and
// epilogue; 4,800 doubles (78 KB per workgroup with the sums, two per CU) hold 300 loci.
#ifndef GBRS_LDS_DOUBLES_H16
#define GBRS_LDS_DOUBLES_H16 4800
#endif
// a word is [dictionary index | words left in the row | position in the row | haplotype mask]: 32 - H - 2 pos_bits(H) bits of index
__host__ __device__ constexpr uint32_t dict_index_limit(int H) { return 1u << (32 - H - 2 * pos_bits(H)); }
__host__ __device__ constexpr int lds_theta_doubles(bool weighted, int H) {
    return weighted ? GBRS_LDS_DOUBLES_WEIGHTED : (H == 16 ? GBRS_LDS_DOUBLES_H16 : GBRS_LDS_DOUBLES);
}
__host__ __device__ constexpr int lds_acc_doubles(bool weighted, int H) { return lds_theta_doubles(weighted, H) + 64; }
// deterministic mode (GBRS_EM_DETERMINISTIC): every wavefront of a tile owns a private copy of the tile's
// sums, so a tile may reference at most this many loci
__host__ __device__ constexpr uint32_t det_dict_cap(int H, bool weighted) { return (lds_acc_doubles(weighted, H) / TILE_WAVES - 1) / H; }
constexpr uint32_t SLOT_DIRECT = 0x80000000u;
constexpr uint32_t SLOT_SET = 0x40000000u;    // slot_dest of a locus-set entry: offset of its destination list in dest_list
constexpr uint32_t SLOT_PAIR = 0x20000000u;   // slot_dest of a two-member set: first destination here, second in dest_b
// dictionary entries as the E-step kernel reads them (the build works on plain ids and encodes them at the end):
// a locus id; DICT_PAIR | first member of a two-member set (second member in dict_b); DICT_SET | set index (set_ptr)
constexpr uint32_t DICT_PAIR = 0x80000000u, DICT_SET = 0x40000000u;
constexpr int HEAVY_SLOTS = 16;                // loci with more slots get a whole wave in the gather (measured 1, 4, 16, 64)

struct TileHdr {
    uint32_t batch_base;   // first batch of the tile in `words`
    uint32_t n_batches;
    uint32_t dict_base;    // first slot / dictionary entry of the tile
    uint32_t dict_count;   // D
};

// what the E-step kernel needs to know about locus sets (null set_ptr: the layout has none)
struct SetArgs {
    uint32_t n_loci;                 // number of loci (the encoded dictionary does not need it; kept for checks)
    const uint32_t *set_ptr, *set_members, *dest_list;
    const uint32_t *dict_b, *dest_b; // per slot: second member / second destination of a two-member set
};

struct TileLayout {
    // sizes
    uint64_t n_pairs = 0;        // (row, locus) pairs = unpadded words
    uint64_t n_rows_in = 0;      // rows with at least one alignment
    uint64_t n_rows = 0;         // rows in the layout (after optional merging), short rows only
    uint64_t n_long = 0;         // rows with more than MAX_ROW_WORDS loci
    uint64_t n_tiles = 0, n_batches = 0, n_slots = 0, n_heavy = 0, n_light = 0;
    uint32_t d_max = 0;          // dictionary capacity used when cutting tiles
    bool weighted = false;       // per-row weights present (count given or rows merged)
    bool deterministic = false;  // dictionaries capped at det_dict_cap(H): one private sum copy per wavefront
    bool all_one_word = false;   // every row of the layout is a single word (and there are no long rows)

    DevBuf<uint32_t> words;          // n_batches * 64
    DevBuf<TileHdr> tiles;           // n_tiles
    DevBuf<uint32_t> dict;           // n_slots: global locus of each slot
    DevBuf<double> word_weight;      // n_batches * 64 (weighted): the row's weight on each of its words
    DevBuf<double> row_weight;       // n_rows (weighted; build-time only)
    DevBuf<uint32_t> slot_ptr;       // L + 1
    DevBuf<uint32_t> slot_list;      // n_slots, grouped by locus, ascending slot inside a locus
    DevBuf<uint32_t> heavy_loci;     // loci with more than HEAVY_SLOTS slots
    DevBuf<uint32_t> light_loci;     // loci with 2..HEAVY_SLOTS slots
    // (locus, first slot row, end slot row) of every heavy / light locus, 3 words each: what the fused gather + M-step
    // launch reads first, so that its slot rows are one dependent load away instead of three (list -> slot_ptr -> rows)
    DevBuf<uint32_t> heavy_range, light_range;
    DevBuf<uint32_t> slot_dest;      // n_slots: where the tile epilogue stores a slot: its own index in
                                     // `partials`, or SLOT_DIRECT | locus when the locus has this one slot only
    DevBuf<uint8_t> locus_class;     // L: 0 no slot, 1 one slot (sum goes straight to acc), 2 few, 3 heavy
    DevBuf<double> partials;         // n_slots * H
    // long rows (kept in pair form)
    DevBuf<uint64_t> long_ptr;       // n_long + 1 offsets into long_loc / long_mask
    DevBuf<uint32_t> long_loc, long_mask;
    DevBuf<double> long_weight;      // n_long
    DevBuf<double> acc_extra;        // L*H, global-atomic target of the long-row kernel
    // Locus sets (GBRS_EM_NO_LOCUS_SETS switches them off).  A read whose alignments to several loci all carry the same
    // haplotype mask contributes  sum_h m_h * (theta[l1,h] + theta[l2,h] + ...)  to its denominator and the same v to every
    // one of those loci: the set {l1, l2, ...} behaves like one locus with theta = the sum of its members'.  The build
    // gives every distinct such set an id n_loci + k, the read becomes ONE word on that id, and the tiles' dictionaries
    // hold set ids beside locus ids.  Nothing outside the tiles sees a set: a tile's prologue sums the members' theta
    // for a set entry (set_ptr / set_members), its epilogue stores a set entry's sums once per member - into a slot row
    // of that member locus (dest_list), so the gather and the M-step add them up like any other slot of the locus.
    uint32_t n_sets = 0;
    uint64_t n_dest_rows = 0;                // rows of `partials` + direct stores: one per (slot, member locus)
    DevBuf<uint32_t> set_ptr, set_members;   // n_sets + 1 offsets; member loci of every set, ascending
    DevBuf<uint32_t> dest_list;              // per set slot: [n, dest_1 .. dest_n] at slot_dest[slot] & ~SLOT_SET
    DevBuf<uint32_t> dict_b, dest_b;         // n_slots: second member / second destination of the two-member sets
    // GBRS_EM_ONE_SHOT: the build temporaries stay allocated until the layout goes (common.h, DeferFrees): a
    // process that handles one sample and exits never pays for returning them.  Without the flag they are freed in
    // one pass when the build ends.
    bool retain_temporaries = false;
    std::vector<void *> retired;
    size_t retired_bytes = 0;
    TileLayout() = default;
    TileLayout(const TileLayout &) = delete;
    TileLayout &operator=(const TileLayout &) = delete;
    ~TileLayout() {
        for (void *p : retired) (void)hipFree(p);
    }
};

// ent_row / col_ptr: the concatenated CSC arrays already on the device (column c = h*L + l).
// count: device pointer or nullptr.  Returns GBRS_OK or a status with the message set.
int build_tile_layout(TileLayout &out, uint64_t R, uint32_t L, uint32_t H, uint64_t N,
                      const uint32_t *ent_row, const uint64_t *col_ptr, const double *count,
                      bool merge_identical_rows, int row_order /* 0 sorted, 1 interleaved, 2 streams */,
                      bool deterministic, hipStream_t stream, unsigned side_by_side = 1 /* handles sharing the device */,
                      bool locus_sets = false, uint32_t dict_cap = 0 /* > 0: at most this many loci per tile dictionary */,
                      uint32_t view_factor = 1);
// view_factor = 2 (round 4, 16 haplotypes): the layout is built over HALF-LOCI - locus l's haplotypes 0-7 are "locus" 2l,
// its haplotypes 8-15 "locus" 2l + 1, L and H passed here are 2 L and 8.  The locus-major vectors (theta, A, lengths:
// element l * 16 + h) are element for element the half-locus view's (2l + h / 8) * 8 + h % 8, so nothing outside the layout
// moves; a read that aligns to both halves of a locus has two words in its row, and the tiles run on the 8-haplotype
// E-step kernel (theta row and sums in registers, 0/1 doubles from the LDS tables, six waves per SIMD) instead of the
// 16-haplotype one (128 registers, four waves per SIMD, theta from LDS for every word).

// `gbrs compress`: equivalence classes of identical rows, in first-seen order.
struct CompressResult {
    uint64_t num_ecs = 0, n_entries = 0;
    DevBuf<uint64_t> col_ptr;     // H*L + 1 offsets into indices (column c = h*L + l)
    DevBuf<uint32_t> indices;     // class ids, ascending inside a column
    DevBuf<double> count;         // num_ecs
};
int compress_device(CompressResult &out, uint64_t R, uint32_t L, uint32_t H, uint64_t N, const uint32_t *ent_row,
                    const uint64_t *col_ptr, const double *count, hipStream_t s);

// `--report-alignment-counts`: aln/uniq are (H x Lout) row-major, locus_uniq is Lout, all DEVICE
// buffers; locus_group (device, nullable) maps locus -> output column (gene level).
// device workspace of alignment_counts_device, kept between calls on the same alignments (gbrs_counts_*)
struct CountsWork;
CountsWork *counts_work_new();
void counts_work_free(CountsWork *w);
int alignment_counts_device(uint64_t R, uint32_t L, uint32_t H, uint64_t N, const uint32_t *ent_row,
                            const uint64_t *col_ptr, const double *count, const int32_t *locus_group,
                            uint32_t Lout, double *aln, double *uniq, double *locus_uniq, hipStream_t s,
                            CountsWork *work = nullptr);

// shared by em.hip and em_layout.hip -----------------------------------------------------------
__device__ __forceinline__ uint32_t find_column(const uint64_t *__restrict__ col_ptr, uint32_t lo,
                                                uint32_t hi, uint64_t k) {
    // largest c in [lo, hi] with col_ptr[c] <= k   (col_ptr non-decreasing, col_ptr[lo] <= k)
    while (lo < hi) {
        const uint32_t mid = lo + ((hi - lo + 1) >> 1);
        if (col_ptr[mid] <= k) lo = mid; else hi = mid - 1;
    }
    return lo;
}

// the same, searching upwards from a column c0 known to be at or before the answer (col_ptr[c0] <= k):
// doubling steps, then bisection; one or two probes when the answer is c0 or close to it
__device__ __forceinline__ uint32_t find_column_from(const uint64_t *__restrict__ col_ptr, uint32_t c0,
                                                     uint32_t ncols, uint64_t k) {
    uint32_t lo = c0, step = 1;
    while (lo + step < ncols && col_ptr[lo + step] <= k) {
        lo += step;
        step <<= 1;
    }
    const uint32_t hi = min(lo + step, ncols) - 1;        // col_ptr[hi + 1] > k or hi is the last column
    return find_column(col_ptr, lo, hi, k);
}

__device__ __forceinline__ uint32_t entry_column(const uint64_t *__restrict__ col_ptr, uint32_t ncols,
                                                 uint64_t k, uint64_t n) {
    // the wave's entries are consecutive: search the wave's first and last entry, then only
    // inside that window (usually a single column)
    const uint64_t kbase = k - (threadIdx.x & 63);
    const uint64_t klast = min(kbase + 63, n - 1);
    uint32_t c_first = 0, c_last = 0;
    if ((threadIdx.x & 63) == 0) {
        c_first = find_column(col_ptr, 0, ncols - 1, kbase);
        c_last = find_column(col_ptr, c_first, ncols - 1, klast);
    }
    c_first = __shfl(c_first, 0, WAVE);
    c_last = __shfl(c_last, 0, WAVE);
    if (c_first == c_last) return c_first;
    return find_column(col_ptr, c_first, c_last, k);
}

}  // namespace gbrs
