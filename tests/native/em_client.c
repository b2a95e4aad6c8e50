/* A plain C99 client of include/gbrs_hip.h: what a non-Python host (cgo, JNI, a C tool) would write to run the
 * EM behind `gbrs quantify`.  tests/test_c_client.py compiles it with gcc (-std=c99 -pedantic: the header must be
 * C, not C++), links it against gbrs_amd/libgbrs_hip.so and compares what it writes with the reference's golden
 * values.
 *
 * Input file (little endian):  u64 R, u32 L, u32 H, u32 has_count, u32 has_len, f64 pseudocount, f64 tol,
 *   u32 max_iters, u32 pad;  per haplotype: u32 indptr[L+1], u32 nnz, u32 indices[nnz];
 *   f64 count[R] if has_count; f64 eff_len[H*L] if has_len.
 * Output file: i32 n_iters, i32 pad, f64 theta[H*L], f64 expected_counts[H*L].
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include "gbrs_hip.h"

static void *xread(FILE *f, size_t n) {
    void *p = malloc(n ? n : 1);
    if (!p || fread(p, 1, n, f) != n) {
        fprintf(stderr, "em_client: short read\n");
        exit(2);
    }
    return p;
}

#define CHECK(call)                                                              \
    do {                                                                         \
        const int st_ = (call);                                                  \
        if (st_ != GBRS_OK) {                                                    \
            fprintf(stderr, "em_client: %s -> %d: %s\n", #call, st_, gbrs_last_error()); \
            return 3;                                                            \
        }                                                                        \
    } while (0)

int main(int argc, char **argv) {
    if (argc != 3) {
        fprintf(stderr, "usage: em_client IN OUT\n");
        return 1;
    }
    if (gbrs_abi_version() != GBRS_ABI_VERSION) {
        fprintf(stderr, "em_client: library ABI %d, header %d\n", gbrs_abi_version(), GBRS_ABI_VERSION);
        return 1;
    }
    FILE *f = fopen(argv[1], "rb");
    if (!f) return 2;
    uint64_t R;
    uint32_t hdr[4], tail[2];
    double pt[2];
    if (fread(&R, 8, 1, f) != 1 || fread(hdr, 4, 4, f) != 4 || fread(pt, 8, 2, f) != 2 || fread(tail, 4, 2, f) != 2) return 2;
    const uint32_t L = hdr[0], H = hdr[1], has_count = hdr[2], has_len = hdr[3];
    const uint32_t **indptr = (const uint32_t **)malloc(H * sizeof(*indptr));
    const uint32_t **indices = (const uint32_t **)malloc(H * sizeof(*indices));
    uint32_t h;
    for (h = 0; h < H; ++h) {
        uint32_t nnz;
        indptr[h] = (const uint32_t *)xread(f, ((size_t)L + 1) * 4);
        if (fread(&nnz, 4, 1, f) != 1) return 2;
        indices[h] = (const uint32_t *)xread(f, (size_t)nnz * 4);
    }
    const double *count = has_count ? (const double *)xread(f, (size_t)R * 8) : NULL;
    const double *eff_len = has_len ? (const double *)xread(f, (size_t)H * L * 8) : NULL;
    fclose(f);

    gbrs_em_t *em = NULL;
    int n_iters = 0;
    double err_hist[1024], stamps[1024];
    double *theta = (double *)malloc((size_t)H * L * 8), *counts = (double *)malloc((size_t)H * L * 8);
    gbrs_em_info_t info;
    CHECK(gbrs_em_create(R, L, H, indptr, indices, count, eff_len, 0, 0u, &em));
    CHECK(gbrs_em_info(em, &info));
    CHECK(gbrs_em_prepare(em, pt[0]));
    CHECK(gbrs_em_run(em, 4, pt[1], (int)tail[0], &n_iters, err_hist, 1024, stamps));
    CHECK(gbrs_em_get(em, theta, counts));
    CHECK(gbrs_em_destroy(em));
    printf("em_client: %d iterations, last err_sum %.6g, %llu device words\n", n_iters,
           n_iters > 0 && n_iters <= 1024 ? err_hist[n_iters - 1] : 0.0, (unsigned long long)info.num_device_words);

    f = fopen(argv[2], "wb");
    if (!f) return 2;
    {
        const int32_t head[2] = {n_iters, 0};
        fwrite(head, 4, 2, f);
        fwrite(theta, 8, (size_t)H * L, f);
        fwrite(counts, 8, (size_t)H * L, f);
    }
    fclose(f);
    return 0;
}
