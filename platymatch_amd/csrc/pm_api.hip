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

int pm_version(void) { return PM_ABI_VERSION; }

const char *pm_error_string(int code) {
    switch (code) {
        case PM_OK: return "ok";
        case PM_ERR_INVALID_ARG: return "invalid argument (null pointer, non-positive size, misaligned descriptor or bad enum)";
        case PM_ERR_WORKSPACE: return "workspace missing or smaller than pm_*_workspace() reports";
        case PM_ERR_LAUNCH: return "HIP reported an error while enqueuing work (see pm_last_hip_error)";
        case PM_ERR_UNSUPPORTED: return "not implemented on the device path";
        default: return "unknown error code";
    }
}

int pm_last_hip_error(void) { return pm::g_last_hip_error; }

}  // extern "C"
