// pm_api.hip — version, error strings, per-thread HIP error slot.
#include "pm_common.h"

namespace pm {
thread_local int g_last_hip_error = 0;
}

extern "C" {

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
