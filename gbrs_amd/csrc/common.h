// Shared host/device helpers for libgbrs_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <utility>
#include <vector>

#include "../../include/gbrs_hip.h"

namespace gbrs {

void set_error(const char *fmt, ...);
int fail(int status, const char *fmt, ...);

#define GBRS_HIP_CHECK(expr)                                                              \
    do {                                                                                  \
        hipError_t e__ = (expr);                                                          \
        if (e__ != hipSuccess)                                                            \
            return ::gbrs::fail(GBRS_ERR_HIP, "%s failed: %s (%s:%d)", #expr,            \
                                hipGetErrorString(e__), __FILE__, __LINE__);              \
    } while (0)

#define GBRS_TRY(expr)                                                                    \
    do {                                                                                  \
        int s__ = (expr);                                                                 \
        if (s__ != GBRS_OK) return s__;                                                   \
    } while (0)

int select_device(int device);

// RAII device buffer (raw hipMalloc.  A stream-ordered pool - hipMallocAsync with an unbounded release
// threshold - was measured for the layout builder's temporaries: no gain over hipMalloc once a first
// build has run, and a second build next to a live handle took 0.4 s in the pool, so it is not used.)
// hipFree, or - inside a DeferFrees scope on this thread - a note to free the block when the scope ends.  On some
// hosts of this pool a hipMalloc that follows a large hipFree stalls for 0.1-0.2 s (measured in the layout build:
// 0.06 -> 111 ms and 0.3 -> 217 ms for two allocations that follow a release), so a one-off build that walks through
// several multi-gigabyte temporaries keeps them until it is done instead of handing them back one by one (up to
// 96 GB of them; beyond that blocks are freed as they go).
void deferred_free(void *p, size_t bytes);
// hipFree everything parked on this thread right now (a hipMalloc just failed); returns the bytes given back
size_t deferred_flush();
struct DeferFrees {
    // sink: where the collected blocks go when the scope ends (their new owner frees them later, e.g. with the
    // handle: GBRS_EM_ONE_SHOT); nullptr = one batched hipFree pass there and then.  sink_bytes (nullable) is
    // increased by the bytes handed over.
    explicit DeferFrees(std::vector<void *> *sink = nullptr, size_t *sink_bytes = nullptr);
    ~DeferFrees();
    std::vector<void *> *sink;
    size_t *sink_bytes;
    DeferFrees(const DeferFrees &) = delete;
    DeferFrees &operator=(const DeferFrees &) = delete;
    bool outer;
};

template <typename T>
struct DevBuf {
    T *p = nullptr;
    size_t n = 0;
    DevBuf() = default;
    DevBuf(const DevBuf &) = delete;
    DevBuf &operator=(const DevBuf &) = delete;
    ~DevBuf() { release(); }
    void release() {
        if (p) deferred_free(p, n * sizeof(T));
        p = nullptr;
        n = 0;
    }
    int alloc(size_t count) {
        release();
        n = count;
        if (count == 0) return GBRS_OK;
        hipError_t e = hipMalloc(reinterpret_cast<void **>(&p), count * sizeof(T));
        if (e != hipSuccess && deferred_flush() > 0) {      // blocks parked by a DeferFrees scope: give them back, retry
            (void)hipGetLastError();
            e = hipMalloc(reinterpret_cast<void **>(&p), count * sizeof(T));
        }
        if (e != hipSuccess) {
            p = nullptr;
            return fail(GBRS_ERR_HIP, "hipMalloc(%zu bytes) failed: %s", count * sizeof(T),
                        hipGetErrorString(e));
        }
        return GBRS_OK;
    }
    size_t bytes() const { return n * sizeof(T); }
    void swap(DevBuf &o) {
        std::swap(p, o.p);
        std::swap(n, o.n);
    }
};

// roctx ranges around the library's calls (rocprofv3 --marker-trace), when GBRS_ROCTX=1: the marker library is
// looked up at run time (librocprofiler-sdk-roctx.so, else libroctx64.so); without the variable or the library the
// ranges cost one predictable branch.
void roctx_push(const char *name);
void roctx_pop();
struct RoctxRange {
    explicit RoctxRange(const char *name) { roctx_push(name); }
    ~RoctxRange() { roctx_pop(); }
    RoctxRange(const RoctxRange &) = delete;
    RoctxRange &operator=(const RoctxRange &) = delete;
};

// Wall-clock checkpoints of the one-off build steps, printed to stderr when GBRS_TUNING_BUILD_TIMES=1.
struct StageTimer {
    bool on;
    std::chrono::steady_clock::time_point t0, last;
    const char *what;
    explicit StageTimer(const char *w) : on(false), what(w) {
        const char *e = std::getenv("GBRS_TUNING_BUILD_TIMES");
        on = e && std::atoi(e) != 0;
        t0 = last = std::chrono::steady_clock::now();
    }
    void mark(const char *stage) {
        if (!on) return;
        (void)hipDeviceSynchronize();
        const auto now = std::chrono::steady_clock::now();
        std::fprintf(stderr, "[%s] %-28s %8.2f ms (total %8.2f)\n", what, stage,
                     std::chrono::duration<double, std::milli>(now - last).count(),
                     std::chrono::duration<double, std::milli>(now - t0).count());
        last = now;
    }
};

constexpr int WAVE = 64;

// ---- wave-level helpers (64-wide wavefronts) ---------------------------------------------
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, WAVE);
    return v;  // valid in lane 0
}

__device__ __forceinline__ double wave_sum_all(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, WAVE);
    return v;  // same value in every lane (butterfly: fixed association)
}

// Block-wide sum, fixed association order (deterministic).  blockDim.x multiple of 64, <= 1024.
__device__ __forceinline__ double block_sum(double v, double *lds /* >= 16 doubles */) {
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    v = wave_sum(v);
    __syncthreads();
    if (lane == 0) lds[wid] = v;
    __syncthreads();
    double r = 0.0;
    if (wid == 0) {
        const int nw = (blockDim.x + 63) >> 6;
        r = (lane < nw) ? lds[lane] : 0.0;
        r = wave_sum(r);
    }
    return r;  // valid in thread 0
}

}  // namespace gbrs
