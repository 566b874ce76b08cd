// pm_common.h — shared device helpers for the gfx950 kernels (wave64, float64).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/platymatch_hip.h"

#define PM_WAVE 64

// Every arithmetic statement in these kernels is one IEEE rounding: the translation units are
// compiled with -ffp-contract=off and fused operations are written explicitly (__builtin_fma).

namespace pm {

extern thread_local int g_last_hip_error;

inline int launch_status() {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        g_last_hip_error = (int)e;
        return PM_ERR_LAUNCH;
    }
    return PM_OK;
}

inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

// ---- deterministic reductions (fixed tree: lane butterfly, then waves in order) ------------------
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, PM_WAVE);
    return v;  // valid in lane 0
}

// Block sum of one double per thread; result valid in thread 0.  `scratch` holds >= blockDim/64 doubles.
__device__ __forceinline__ double block_sum(double v, double *scratch) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    v = wave_sum(v);
    __syncthreads();  // scratch may still be read from a previous call
    if (lane == 0) scratch[wave] = v;
    __syncthreads();
    double r = 0.0;
    if (threadIdx.x == 0)
        for (int w = 0; w < nw; ++w) r += scratch[w];
    return r;
}

// 1/d refined to full float64 accuracy, then n/d correctly rounded, for finite positive d and
// finite n >= 0 well inside the exponent range (no scaling/fix-up needed): the LLVM AMDGPU f64
// division sequence (v_rcp_f64 + 2 Newton steps + residual correction) without div_scale/div_fixup.
__device__ __forceinline__ double div_pos(double n, double d) {
    double r = __builtin_amdgcn_rcp(d);
    double e = __builtin_fma(-d, r, 1.0);
    r = __builtin_fma(e, r, r);
    e = __builtin_fma(-d, r, 1.0);
    r = __builtin_fma(e, r, r);
    double q = n * r;
    double res = __builtin_fma(-d, q, n);
    return __builtin_fma(res, r, q);
}

}  // namespace pm
