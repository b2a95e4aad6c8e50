/* A plain C99 client of the HMM half of include/gbrs_hip.h (what `gbrs reconstruct` runs per sample): create with
 * the transition tables, set_expression, run, get per chromosome.  tests/test_c_client.py compiles it with
 * gcc -std=c99 -pedantic -Werror, links it against gbrs_amd/libgbrs_hip.so and compares its output with the
 * reference's golden values.
 *
 * Input (little endian): i32 H, i32 n_chrom, f64 expr_threshold, f64 sigma; per chromosome: i32 n_genes, i32 n_trans,
 *   f64 tprob[n_trans][S][S], f64 expr[n_genes][H], f64 avecs[n_genes][H][H], u8 has_avec[n_genes] (padded to 8 bytes).
 * Output: per chromosome: f64 gamma[S][n_genes], i32 calls[n_genes], i32 n_states_path, i32 states[n_states_path].
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include "gbrs_hip.h"

static void *xread(FILE *f, size_t n) {
    void *p = malloc(n ? n : 1);
    if (!p || fread(p, 1, n, f) != n) {
        fprintf(stderr, "hmm_client: short read\n");
        exit(2);
    }
    return p;
}

#define CHECK(call)                                                              \
    do {                                                                         \
        const int st_ = (call);                                                  \
        if (st_ != GBRS_OK) {                                                    \
            fprintf(stderr, "hmm_client: %s -> %d: %s\n", #call, st_, gbrs_last_error()); \
            return 3;                                                            \
        }                                                                        \
    } while (0)

int main(int argc, char **argv) {
    if (argc != 3) return 1;
    FILE *f = fopen(argv[1], "rb");
    if (!f) return 2;
    int32_t head[2];
    double par[2];
    if (fread(head, 4, 2, f) != 2 || fread(par, 8, 2, f) != 2) return 2;
    const int H = head[0], C = head[1], S = H * (H + 1) / 2;
    int32_t *n_genes = (int32_t *)malloc(C * sizeof(int32_t)), *n_trans = (int32_t *)malloc(C * sizeof(int32_t));
    const double **tprob = (const double **)malloc(C * sizeof(*tprob));
    const double **expr = (const double **)malloc(C * sizeof(*expr));
    const double **avecs = (const double **)malloc(C * sizeof(*avecs));
    const uint8_t **has = (const uint8_t **)malloc(C * sizeof(*has));
    int c;
    for (c = 0; c < C; ++c) {
        int32_t nn[2];
        if (fread(nn, 4, 2, f) != 2) return 2;
        n_genes[c] = nn[0];
        n_trans[c] = nn[1];
        tprob[c] = (const double *)xread(f, (size_t)nn[1] * S * S * 8);
        expr[c] = (const double *)xread(f, (size_t)nn[0] * H * 8);
        avecs[c] = (const double *)xread(f, (size_t)nn[0] * H * H * 8);
        has[c] = (const uint8_t *)xread(f, ((size_t)nn[0] + 7) / 8 * 8);
    }
    fclose(f);

    gbrs_hmm_t *hmm = NULL;
    gbrs_hmm_info_t info;
    CHECK(gbrs_hmm_create(H, C, n_genes, n_trans, tprob, 0, &hmm));
    CHECK(gbrs_hmm_set_expression(hmm, 1, expr, avecs, has, par[0], par[1]));
    CHECK(gbrs_hmm_run(hmm));
    CHECK(gbrs_hmm_info(hmm, &info));
    printf("hmm_client: %llu genes, %d states, run %.3f ms\n", (unsigned long long)info.total_genes, (int)info.num_states,
           info.last_run_ms);
    f = fopen(argv[2], "wb");
    if (!f) return 2;
    for (c = 0; c < C; ++c) {
        const int n = n_genes[c];
        const int32_t path = (n < n_trans[c] ? n : n_trans[c]) + 1;
        double *gamma = (double *)malloc((size_t)S * (n ? n : 1) * 8);
        int32_t *calls = (int32_t *)malloc((size_t)(n ? n : 1) * 4), *states = (int32_t *)malloc((size_t)path * 4);
        CHECK(gbrs_hmm_get(hmm, 0, c, gamma, states, calls, NULL, NULL, NULL, NULL, NULL));
        fwrite(gamma, 8, (size_t)S * n, f);
        fwrite(calls, 4, (size_t)n, f);
        fwrite(&path, 4, 1, f);
        fwrite(states, 4, (size_t)path, f);
        free(gamma); free(calls); free(states);
    }
    fclose(f);
    CHECK(gbrs_hmm_destroy(hmm));
    return 0;
}
