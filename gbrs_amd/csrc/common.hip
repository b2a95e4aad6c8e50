// Error channel and device selection shared by the EM and HMM translation units.
#include "common.h"

#include <cstdlib>
#include <dlfcn.h>
#include <vector>

namespace gbrs {

namespace {
struct Roctx {
    int (*push)(const char *) = nullptr;
    int (*pop)() = nullptr;
    Roctx() {
        const char *e = std::getenv("GBRS_ROCTX");
        if (!e || !std::atoi(e)) return;
        for (const char *lib : {"librocprofiler-sdk-roctx.so", "librocprofiler-sdk-roctx.so.1", "libroctx64.so", "libroctx64.so.4"}) {
            if (void *h = dlopen(lib, RTLD_NOW | RTLD_GLOBAL)) {
                push = reinterpret_cast<int (*)(const char *)>(dlsym(h, "roctxRangePushA"));
                pop = reinterpret_cast<int (*)()>(dlsym(h, "roctxRangePop"));
                if (push && pop) return;
                push = nullptr;
                pop = nullptr;
            }
        }
    }
};
const Roctx &roctx() {
    static const Roctx r;
    return r;
}
}  // namespace
void roctx_push(const char *name) {
    if (roctx().push) (void)roctx().push(name);
}
void roctx_pop() {
    if (roctx().pop) (void)roctx().pop();
}

static thread_local char g_err[1024] = "";

void set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

namespace {
thread_local std::vector<void *> *g_deferred = nullptr;
thread_local size_t g_deferred_bytes = 0;
constexpr size_t DEFER_CAP = (size_t)96 << 30;
}
void deferred_free(void *p, size_t bytes) {
    if (g_deferred && g_deferred_bytes + bytes <= DEFER_CAP) {
        g_deferred->push_back(p);
        g_deferred_bytes += bytes;
    } else {
        (void)hipFree(p);
    }
}
size_t deferred_flush() {
    if (!g_deferred || g_deferred->empty()) return 0;
    (void)hipDeviceSynchronize();                   // nothing in flight may still read a parked block
    for (void *p : *g_deferred) (void)hipFree(p);
    g_deferred->clear();
    const size_t b = g_deferred_bytes;
    g_deferred_bytes = 0;
    return b;
}
DeferFrees::DeferFrees(std::vector<void *> *sink_, size_t *sink_bytes_)
    : sink(sink_), sink_bytes(sink_bytes_), outer(g_deferred == nullptr) {
    static const bool off = [] { const char *e = std::getenv("GBRS_TUNING_EAGER_FREE"); return e && std::atoi(e) != 0; }();
    if (outer && !off) g_deferred = new std::vector<void *>();
    else outer = false;
}
DeferFrees::~DeferFrees() {
    if (!outer) return;
    std::vector<void *> *v = g_deferred;
    g_deferred = nullptr;
    if (sink) {
        sink->insert(sink->end(), v->begin(), v->end());
        if (sink_bytes) *sink_bytes += g_deferred_bytes;
    } else {
        for (void *p : *v) (void)hipFree(p);
    }
    g_deferred_bytes = 0;
    delete v;
}

int fail(int status, const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return status;
}

int select_device(int device) {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0)
        return fail(GBRS_ERR_NO_DEVICE, "no HIP device visible (%s); libgbrs_hip has no CPU fallback",
                    e == hipSuccess ? "device count is 0" : hipGetErrorString(e));
    if (device < 0 || device >= n) return fail(GBRS_ERR_INVALID, "device %d out of range (0..%d)", device, n - 1);
    GBRS_HIP_CHECK(hipSetDevice(device));
    return GBRS_OK;
}

}  // namespace gbrs

// One trivial launch per translation unit: every .hip file carries its own code object, which the runtime loads on
// the first launch of one of its kernels (~20 ms for the largest).
namespace gbrs { void warm_em(hipStream_t); void warm_layout(hipStream_t); void warm_hmm(hipStream_t); }

extern "C" {

const char *gbrs_last_error(void) { return gbrs::g_err; }
int gbrs_abi_version(void) { return GBRS_ABI_VERSION; }

int gbrs_warm_up(int device) {
    GBRS_TRY(gbrs::select_device(device));
    int *d = nullptr, h = 0;
    GBRS_HIP_CHECK(hipMalloc(&d, sizeof(int)));
    GBRS_HIP_CHECK(hipMemset(d, 0, sizeof(int)));
    gbrs::warm_em(nullptr);
    gbrs::warm_layout(nullptr);
    gbrs::warm_hmm(nullptr);
    GBRS_HIP_CHECK(hipMemcpy(&h, d, sizeof(int), hipMemcpyDeviceToHost));     // also sets up the staging path of small copies
    GBRS_HIP_CHECK(hipFree(d));
    return GBRS_OK;
}

int gbrs_device_count(void) {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) return gbrs::fail(GBRS_ERR_NO_DEVICE, "hipGetDeviceCount: %s", hipGetErrorString(e));
    return n;
}

}
