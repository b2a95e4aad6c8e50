// What shader clock does a kernel see while something else keeps the chip busy?  One wavefront runs a fixed number of
// dependent scalar additions (a constant number of shader cycles) between two reads of the constant-rate 100 MHz counter
// (s_memrealtime), over and over for a few seconds; the host prints the distribution of the achieved rate over time.  The
// probe's workgroup is 16 wavefronts of 128 VGPRs each -- the whole register file of one compute unit -- so no other
// kernel's wavefronts share its CU: 15 of them wait at the closing barrier, one counts.  Run it beside a workload:   ./clock_probe 3 > probe.txt &  <workload> ; wait     (a second argument names a file for the raw series)
// hipcc --offload-arch=gfx950 -O3 -o clock_probe clock_probe.hip
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

__global__ void __launch_bounds__(1024) probe_kernel(unsigned long long *out, int n_samples, int inner) {
    asm volatile("v_mov_b32 v127, 0" ::: "v127");             // allocate 128 VGPRs per wavefront
    __builtin_amdgcn_s_setprio(3);
    if (threadIdx.x >= 64) n_samples = 0;
    for (int i = 0; i < n_samples; ++i) {
        const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
        unsigned x = (unsigned)i;
        for (int k = 0; k < inner; ++k)
            asm volatile(".rept 1024\n\ts_add_u32 %0, %0, 1\n\t.endr"   // 4 KB of straight-line code per taken branch: what the
                         : "+s"(x)                                    // neighbouring CU does to the shared instruction cache
                         :                                            // stays out of the measurement
                         : "scc");
        const unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
        if (threadIdx.x == 0) out[i] = ((t1 - t0) << 32) | (x & 0xffffu);
    }
    __syncthreads();
}

int main(int argc, char **argv) {
    const double seconds = argc > 1 ? atof(argv[1]) : 3.0;
    const int inner = 32;                                     // 32,768 dependent scalar adds per sample
    unsigned long long *d = nullptr;
    // calibrate: how long is one sample on the idle chip
    hipMalloc(&d, sizeof(unsigned long long) * (1 << 20));
    hipLaunchKernelGGL(probe_kernel, dim3(1), dim3(1024), 0, 0, d, 64, inner);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(64);
    hipMemcpy(h.data(), d, 64 * sizeof(unsigned long long), hipMemcpyDeviceToHost);
    std::vector<double> t;
    for (int i = 16; i < 64; ++i) t.push_back((double)(h[i] >> 32) * 10e-9);       // seconds per sample (100 MHz ticks)
    std::sort(t.begin(), t.end());
    const double idle = t[t.size() / 2];
    const int n = (int)std::min<double>((1 << 20), seconds / idle);
    std::printf("# idle chip: %.2f us per sample of %d dependent s_add_u32 = %.0f M adds/s; %d samples follow\n", idle * 1e6, inner * 1024,
                inner * 1024 / idle / 1e6, n);
    std::fflush(stdout);
    hipLaunchKernelGGL(probe_kernel, dim3(1), dim3(1024), 0, 0, d, n, inner);
    hipDeviceSynchronize();
    h.resize(n);
    hipMemcpy(h.data(), d, n * sizeof(unsigned long long), hipMemcpyDeviceToHost);
    if (argc > 2) {                                           // raw series: time [ms], rate relative to the idle chip
        FILE *f = std::fopen(argv[2], "w");
        double at = 0.0;
        for (int i = 0; i < n; ++i) {
            const double dt = (double)(h[i] >> 32) * 10e-9;
            at += dt;
            std::fprintf(f, "%.3f %.3f\n", at * 1e3, idle / dt);
        }
        std::fclose(f);
    }
    // rate relative to the idle chip, in 50 ms buckets of the probe's own time
    double clock_t = 0.0, bucket_end = 0.05, acc = 0.0;
    int cnt = 0;
    double lo = 1e9;
    for (int i = 0; i < n; ++i) {
        const double dt = (double)(h[i] >> 32) * 10e-9;
        clock_t += dt;
        acc += idle / dt;
        ++cnt;
        lo = std::min(lo, idle / dt);
        if (clock_t >= bucket_end) {
            std::printf("t = %5.2f s   mean rate %.3f of idle   slowest sample %.3f\n", bucket_end, acc / cnt, lo);
            acc = 0.0; cnt = 0; lo = 1e9; bucket_end += 0.05;
        }
    }
    return 0;
}
