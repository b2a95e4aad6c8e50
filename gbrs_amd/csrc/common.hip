// Error channel and device selection shared by the EM and HMM translation units.
#include "common.h"

namespace gbrs {

static thread_local char g_err[1024] = "";

void set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
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

extern "C" {

const char *gbrs_last_error(void) { return gbrs::g_err; }
int gbrs_abi_version(void) { return GBRS_ABI_VERSION; }

int gbrs_device_count(void) {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) return gbrs::fail(GBRS_ERR_NO_DEVICE, "hipGetDeviceCount: %s", hipGetErrorString(e));
    return n;
}

}
