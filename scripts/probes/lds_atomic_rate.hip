// Probe: cost of ds_add_f64 (LDS float64 atomic add) per wavefront instruction as a function of the active lanes and of
// address collisions.  GPU box: hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics -o /tmp/p scripts/probes/lds_atomic_rate.hip && /tmp/p
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void __launch_bounds__(512) k(double *out, int iters, int active, int spread, int waves, long long *cycles) {
    __shared__ double acc[4096];
    for (int i = threadIdx.x; i < 4096; i += blockDim.x) acc[i] = 0.0;
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    long long t0 = 0, t1 = 0;
    if (wave < waves) {
        // spread = 1: every lane its own address (8 consecutive doubles per lane); spread = 0: all lanes one address
        double *p = acc + (spread ? (wave * 512 + lane * 8) : wave * 8);
        t0 = __builtin_readcyclecounter();
        if (lane < active) {
            for (int i = 0; i < iters; ++i) {
#pragma unroll
                for (int h = 0; h < 8; ++h) atomicAdd(p + h, 1.0);
            }
        }
        __builtin_amdgcn_s_waitcnt(0);
        t1 = __builtin_readcyclecounter();
    }
    __syncthreads();
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc[threadIdx.x];
    if (threadIdx.x == 0 && blockIdx.x == 0) *cycles = t1 - t0;
}
int main() {
    double *out; long long *cyc, h;
    hipMalloc(&out, 512 * 8 * 1024); hipMalloc(&cyc, 8);
    const int iters = 2000;
    for (int waves : {1, 8})
        for (int spread : {1, 0})
            for (int active : {64, 32, 16, 8, 4, 1}) {
                k<<<1, 512>>>(out, 10, active, spread, waves, cyc);
                hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
                hipEventRecord(e0); k<<<1, 512>>>(out, iters, active, spread, waves, cyc); hipEventRecord(e1); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1);
                hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
                printf("waves %d  %s  active lanes %2d: %.1f ns per ds_add_f64 wave-instruction (wave 0 view: %.1f counter ticks)\n", waves,
                       spread ? "distinct addresses" : "one address      ", active, ms * 1e6 / (iters * 8.0), (double)h / (iters * 8.0));
            }
    return 0;
}
