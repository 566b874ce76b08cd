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

// n / d, correctly rounded, for finite positive d and finite n >= 0 well inside the exponent range (no
// scaling / fix-up needed).  v_rcp_f64 is good to 2^-24.4 on gfx950 (tools/microbench/rcp_accuracy.hip); one
// cubic step r <- r + r*(e + e*e), e = 1 - d*r, leaves a truncation error of e^3 < 2^-73, i.e. the reciprocal
// rounded to nearest in all but ~2^-20 of cases and within 0.5000005 ulp otherwise; the quotient then gets the
// usual exact-residual correction (Markstein).  This is LLVM's AMDGPU f64 division with its two Newton steps
// fused into one (one fma fewer) and without div_scale/div_fixup; it is checked bit for bit against IEEE
// division on 1.3e11 operands, a third of them built to sit within 2^-53 ulp of a rounding midpoint
// (tools/microbench/div_check.hip, also run by the gpu test-suite), and by every chi-square parity test.
__device__ __forceinline__ double div_pos(double n, double d) {
    double r = __builtin_amdgcn_rcp(d);
    const double e = __builtin_fma(-d, r, 1.0);
    const double t = __builtin_fma(e, e, e);
    r = __builtin_fma(r, t, r);
    const double q = n * r;
    const double res = __builtin_fma(-d, q, n);
    return __builtin_fma(res, r, q);
}

}  // namespace pm
