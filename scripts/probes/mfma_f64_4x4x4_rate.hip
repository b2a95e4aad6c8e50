// Probe: issue rate of v_mfma_f64_4x4x4_4b_f64 (four independent 4x4x4 products per instruction, 512 flops) for one wavefront and
// for one wavefront per SIMD - next to v_mfma_f64_16x16x4 (2,048 flops, 64 cycles on this chip: mfma_f64_rate.hip): a 36-state
// operator is nine 4-blocks in every dimension, so the small shape would waste nothing where the 16x16x4 tiles pad 36 to 48 twice.
// GPU box: hipcc --offload-arch=gfx950 -O3 -o /tmp/p scripts/probes/mfma_f64_4x4x4_rate.hip && /tmp/p
#include <hip/hip_runtime.h>
#include <cstdio>
template <int CHAINS>
__global__ void __launch_bounds__(64) k(double *out, int iters) {
    double a = 1.0 + threadIdx.x * 1e-9, b = 1.0 - threadIdx.x * 1e-9;
    double d[CHAINS];
    for (int c = 0; c < CHAINS; ++c) d[c] = 0.0;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int r = 0; r < 9; ++r)
#pragma unroll
            for (int c = 0; c < CHAINS; ++c) d[c] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, d[c], 0, 0, 0);
    }
    double s = 0;
    for (int c = 0; c < CHAINS; ++c) s += d[c];
    out[blockIdx.x * 64 + threadIdx.x] = s;
}
template <int CHAINS>
void run(const char *name, int blocks) {
    double *out;
    hipMalloc(&out, blocks * 64 * sizeof(double));
    const int iters = 4000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<CHAINS><<<blocks, 64>>>(out, 10);
    hipEventRecord(e0); k<CHAINS><<<blocks, 64>>>(out, iters); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double n = (double)iters * 9 * CHAINS;
    printf("%s, %d wavefront(s): %.2f ns per v_mfma_f64_4x4x4 (512 flops)\n", name, blocks, ms * 1e6 / n);
    hipFree(out);
}
int main() {
    run<1>("1 dependent chain", 1);
    run<4>("4 chains round robin", 1);
    run<8>("8 chains round robin", 1);
    run<8>("8 chains round robin", 1024);
    return 0;
}
