// pm_api.hip — version, error strings, per-thread HIP error slot.
#include "pm_common.h"

namespace pm {
thread_local int g_last_hip_error = 0;
}

namespace pm {
// one wave; lane 0 stamps, the wave sleeps between samples (64 x 64 cycles at a time) so that it takes no issue slots worth naming
__global__ __launch_bounds__(64) void clock_probe_kernel(unsigned long long *__restrict__ out, int n, unsigned long long period) {
    if (threadIdx.x != 0) return;
    unsigned long long next = __builtin_amdgcn_s_memrealtime();
    for (int k = 0; k < n; ++k) {                       // ends after n samples: every iteration waits for a deadline that arrives
        unsigned long long rt;
        while ((rt = __builtin_amdgcn_s_memrealtime()) < next) __builtin_amdgcn_s_sleep(64);
        const unsigned long long ct = __builtin_amdgcn_s_memtime();
        out[2 * k] = ct;
        out[2 * k + 1] = rt;
        next += period;
    }
}
}  // namespace pm

extern "C" {

int pm_clock_probe(unsigned long long *samples, int n_samples, unsigned long long period_ticks, void *stream) {
    if (!samples || n_samples <= 0 || n_samples > 65536 || period_ticks == 0) return PM_ERR_INVALID_ARG;
    if ((double)n_samples * (double)period_ticks > 5.0e8) return PM_ERR_INVALID_ARG;      // 5 s of the 100 MHz counter
    pm::clock_probe_kernel<<<1, 64, 0, (hipStream_t)stream>>>(samples, n_samples, period_ticks);
    return pm::launch_status();
}

// ---- device memory outside any caching allocator (platymatch_amd/device_memory.py; a C caller's malloc / free) ----------------
namespace pm {
struct OnDevice {                                     // the calling thread's current device is put back on every path
    int prev = -1;
    hipError_t err;
    explicit OnDevice(int device) {
        err = hipGetDevice(&prev);
        if (err == hipSuccess && prev != device) err = hipSetDevice(device);
    }
    ~OnDevice() {
        if (prev >= 0) (void)hipSetDevice(prev);
    }
};
static int hip_failure(hipError_t e) {
    (void)hipGetLastError();                          // (a failed allocation leaves its code behind: the next launch_status must not see it)
    g_last_hip_error = (int)e;
    return e == hipErrorOutOfMemory ? PM_ERR_NO_MEMORY : PM_ERR_LAUNCH;
}
}  // namespace pm

int pm_device_alloc(int device, size_t bytes, void **out) {
    if (!out || bytes == 0 || device < 0) return PM_ERR_INVALID_ARG;
    *out = nullptr;
    pm::OnDevice here(device);
    if (here.err != hipSuccess) return pm::hip_failure(here.err);
    void *p = nullptr;
    const hipError_t e = hipMalloc(&p, bytes);
    if (e != hipSuccess) return pm::hip_failure(e);
    *out = p;
    return PM_OK;
}

int pm_device_free(int device, void *ptr) {
    if (device < 0 || !ptr) return PM_ERR_INVALID_ARG;
    pm::OnDevice here(device);
    if (here.err != hipSuccess) return pm::hip_failure(here.err);
    const hipError_t e = hipFree(ptr);                // (waits for the device's outstanding work: a block is never pulled from under a kernel)
    return e == hipSuccess ? PM_OK : pm::hip_failure(e);
}

int pm_device_memory(int device, size_t *free_bytes, size_t *total_bytes) {
    if (device < 0 || !free_bytes || !total_bytes) return PM_ERR_INVALID_ARG;
    pm::OnDevice here(device);
    if (here.err != hipSuccess) return pm::hip_failure(here.err);
    const hipError_t e = hipMemGetInfo(free_bytes, total_bytes);
    return e == hipSuccess ? PM_OK : pm::hip_failure(e);
}

int pm_version(void) { return PM_ABI_VERSION; }

const char *pm_error_string(int code) {
    switch (code) {
        case PM_OK: return "ok";
        case PM_ERR_INVALID_ARG: return "invalid argument (null pointer, non-positive size, misaligned descriptor or bad enum)";
        case PM_ERR_WORKSPACE: return "workspace missing or smaller than pm_*_workspace() reports";
        case PM_ERR_LAUNCH: return "HIP reported an error while enqueuing work (see pm_last_hip_error)";
        case PM_ERR_UNSUPPORTED: return "not implemented on the device path";
        case PM_ERR_NO_MEMORY: return "the device has no block of that size left (pm_device_alloc)";
        default: return "unknown error code";
    }
}

int pm_last_hip_error(void) { return pm::g_last_hip_error; }

}  // extern "C"
