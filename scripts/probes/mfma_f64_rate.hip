// Probe: issue rate of v_mfma_f64_16x16x4_f64 for one wavefront (GPU box: hipcc --offload-arch=gfx950 -O3 -o /tmp/p scripts/probes/mfma_f64_rate.hip && /tmp/p)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));
template <int CHAINS>
__global__ void __launch_bounds__(64) k(double *out, int iters, long long *cycles) {
    double a = 1.0 + threadIdx.x * 1e-9, b = 1.0 - threadIdx.x * 1e-9;
    d4 d[CHAINS];
    for (int c = 0; c < CHAINS; ++c) d[c] = d4{0, 0, 0, 0};
    long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int r = 0; r < 9; ++r)
#pragma unroll
            for (int c = 0; c < CHAINS; ++c) d[c] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, d[c], 0, 0, 0);
    }
    long long t1 = __builtin_readcyclecounter();
    double s = 0;
    for (int c = 0; c < CHAINS; ++c) s += d[c][0] + d[c][1] + d[c][2] + d[c][3];
    out[blockIdx.x * 64 + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cycles = t1 - t0;
}
template <int CHAINS>
void run(const char *name, int blocks) {
    double *out; long long *cyc, h;
    hipMalloc(&out, blocks * 64 * sizeof(double)); hipMalloc(&cyc, 8);
    const int iters = 2000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<CHAINS><<<blocks, 64>>>(out, 10, cyc);
    hipEventRecord(e0); k<CHAINS><<<blocks, 64>>>(out, iters, cyc); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
    const double n = (double)iters * 9 * CHAINS;
    printf("%s blocks %d: %.1f ns per MFMA (wall), %.1f shader-clock ticks per MFMA (s_memtime units)\n", name, blocks, ms * 1e6 / n, (double)h / n);
}
int main() {
    run<1>("1 dependent chain", 1);
    run<3>("3 chains round robin", 1);
    run<3>("3 chains round robin", 1024);
    run<3>("3 chains round robin", 4096);
    return 0;
}
