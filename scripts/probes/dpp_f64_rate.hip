// Probe: issue rate of the 64-bit DPP forms (v_fmac_f64_dpp, v_mov_b64_dpp, row_newbcast) against plain v_fma_f64 and
// 32-bit DPP moves, one wavefront, four independent chains.
// GPU box: hipcc --offload-arch=gfx950 -O3 -o /tmp/p scripts/probes/dpp_f64_rate.hip && /tmp/p
#include <hip/hip_runtime.h>
#include <cstdio>
template <int MODE>
__global__ void __launch_bounds__(64) k(double *out, int iters) {
    double a = 1.0 + threadIdx.x * 1e-9, t = 1e-9, s0 = 0, s1 = 0, s2 = 0, s3 = 0;
    asm volatile("" : "+v"(a), "+v"(t));
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            if (MODE == 0) {          // plain fma
                asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(s0) : "v"(a), "v"(t));
                asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(s1) : "v"(a), "v"(t));
                asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(s2) : "v"(a), "v"(t));
                asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(s3) : "v"(a), "v"(t));
            } else if (MODE == 1) {   // fmac with a DPP row broadcast
                asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:3 row_mask:0xf bank_mask:0xf" : "+v"(s0) : "v"(a), "v"(t));
                asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:5 row_mask:0xf bank_mask:0xf" : "+v"(s1) : "v"(a), "v"(t));
                asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:7 row_mask:0xf bank_mask:0xf" : "+v"(s2) : "v"(a), "v"(t));
                asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:9 row_mask:0xf bank_mask:0xf" : "+v"(s3) : "v"(a), "v"(t));
            } else if (MODE == 2) {   // 64-bit DPP move
                asm volatile("v_mov_b64_dpp %0, %1 row_newbcast:3 row_mask:0xf bank_mask:0xf" : "+v"(s0) : "v"(a));
                asm volatile("v_mov_b64_dpp %0, %1 row_newbcast:5 row_mask:0xf bank_mask:0xf" : "+v"(s1) : "v"(a));
                asm volatile("v_mov_b64_dpp %0, %1 row_newbcast:7 row_mask:0xf bank_mask:0xf" : "+v"(s2) : "v"(a));
                asm volatile("v_mov_b64_dpp %0, %1 row_newbcast:9 row_mask:0xf bank_mask:0xf" : "+v"(s3) : "v"(a));
            } else {                  // two 32-bit DPP moves (one double)
                int lo = __double2loint(a), hi = __double2hiint(a), x0, x1, x2, x3;
                asm volatile("v_mov_b32_dpp %0, %1 row_newbcast:3 row_mask:0xf bank_mask:0xf" : "=v"(x0) : "v"(lo));
                asm volatile("v_mov_b32_dpp %0, %1 row_newbcast:3 row_mask:0xf bank_mask:0xf" : "=v"(x1) : "v"(hi));
                asm volatile("v_mov_b32_dpp %0, %1 row_newbcast:5 row_mask:0xf bank_mask:0xf" : "=v"(x2) : "v"(lo));
                asm volatile("v_mov_b32_dpp %0, %1 row_newbcast:5 row_mask:0xf bank_mask:0xf" : "=v"(x3) : "v"(hi));
                s0 += __hiloint2double(x1, x0) * 0.0;
                s1 += __hiloint2double(x3, x2) * 0.0;
            }
        }
    }
    out[blockIdx.x * 64 + threadIdx.x] = s0 + s1 + s2 + s3;
}
template <int MODE>
void run(const char *name, double per_iter) {
    double *out;
    hipMalloc(&out, 64 * sizeof(double));
    const int iters = 20000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<MODE><<<1, 64>>>(out, 10);
    hipEventRecord(e0); k<MODE><<<1, 64>>>(out, iters); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-44s %.2f ns per instruction (one wavefront)\n", name, ms * 1e6 / ((double)iters * per_iter));
}
int main() {
    run<0>("v_fma_f64", 64);
    run<1>("v_fmac_f64_dpp row_newbcast", 64);
    run<2>("v_mov_b64_dpp row_newbcast", 64);
    run<3>("v_mov_b32_dpp x2 + v_fma_f64 (per instruction)", 96);      // 16 x (4 moves + 2 fmas)
    return 0;
}
