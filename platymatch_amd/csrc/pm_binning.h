// pm_binning.h — log-spherical bin index of one neighbour, by comparison only.
//
// Reproduces get_shape_context's per-neighbour body and get_bin_index
// (reference shape_context.py:25-35, 46-58) for the only binning the reference
// ever uses (r_inner 1/8, r_outer 2, 5 x 6 x 12 bins):
//   r_index     first edge of np.logspace(log10(1/8), log10(2), 5) with r < edge, else 4
//   theta_index arccos(z_/r_) // (pi/6)      -> thresholds on c = z_/r_      (pm_bin_tables.h)
//   phi_index   atan2(y_,x_) mod 2pi // (pi/6) -> exact half-plane sign tests (pm_bin_tables.h)
// No acos/atan2 is evaluated: every step of the two floor-divisions is a
// precomputed float64 threshold, so the result is the same on every device.
// Pure functions, compiled for the GPU by hipcc and for the host by the CPU
// test harness (tests/csrc/binning_host.c), which checks them against the oracle.
#pragma once

#if defined(__HIPCC__)
#define PM_HD __host__ __device__ __forceinline__
#define PM_TABLE_QUAL static __device__ constexpr
#else
#define PM_HD static inline
#define PM_TABLE_QUAL static const
#endif

#include "pm_bin_tables.h"

#define PM_NBINS 360
#define PM_DROP (-1)   // neighbour not counted: NaN index, or index >= 360 (shape_context.py:39-40)

// np.logspace(np.log10(1/8), np.log10(2), 5) as numpy 2.2.6 evaluates it (shape_context.py:24):
// edges 1 and 2 are one ulp above 1/4 and 1/2.  Edge 4 (= 2) never matters: r >= 1 is ring 4 either way.
#define PM_REDGE0 0x1.0000000000000p-3
#define PM_REDGE1 0x1.0000000000001p-2
#define PM_REDGE2 0x1.0000000000001p-1
#define PM_REDGE3 0x1.0000000000000p+0

// sign of CH*y - SH*x with CH+CL, SH+SL double-double constants: +1 / -1 (0 counts as +1).
PM_HD int pm_halfplane_ge(double ch, double cl, double sh, double sl, double x, double y) {
    double p1 = ch * y, p2 = sh * x;
    double d = p1 - p2;
    double mag = __builtin_fabs(p1) + __builtin_fabs(p2);
    if (__builtin_fabs(d) > mag * 0x1p-48) return d > 0.0;   // rounding + lo parts are < 2^-50 * mag
    // near the boundary: error-free products and difference, then the lo parts
    double e1 = __builtin_fma(ch, y, -p1), e2 = __builtin_fma(sh, x, -p2);
    double s = p1 - p2;
    double bb = s - p1;
    double t = (p1 - (s - bb)) + (-p2 - bb);            // two_sum(p1, -p2) = s + t exactly
    double rest = t + ((e1 - e2) + (cl * y - sl * x));
    return (s + rest) >= 0.0;
}

// phi_index (0..12) for y != 0 handled by sign tests; exact zeros follow atan2's signed-zero rules.
PM_HD int pm_phi_index(double x, double y) {
    if (y == 0.0) {
        // atan2(+-0, x>0 or +0) = +-0 -> 0 ; atan2(+-0, x<0 or -0) = +-pi -> phi = pi_d
        int neg = (x < 0.0) || (x == 0.0 && __builtin_signbit(x));
        return neg ? PM_IDX_PI : 0;
    }
    // keep the sign tests' products away from underflow/overflow (power-of-two scaling keeps the angle)
    double big = __builtin_fmax(__builtin_fabs(x), __builtin_fabs(y));
    if (big < 0x1p-900) { x *= 0x1p+200; y *= 0x1p+200; }
    else if (big > 0x1p+900) { x *= 0x1p-200; y *= 0x1p-200; }
    int idx;
    if (y > 0.0) {
        idx = 0;
#pragma unroll
        for (int m = 0; m < PM_PHI_UPPER; ++m)
            idx += pm_halfplane_ge(PM_PHI[m][0], PM_PHI[m][1], PM_PHI[m][2], PM_PHI[m][3], x, y);
    } else {
        idx = PM_PHI_UPPER;
#pragma unroll
        for (int m = PM_PHI_UPPER; m < 12; ++m)
            idx += pm_halfplane_ge(PM_PHI[m][0], PM_PHI[m][1], PM_PHI[m][2], PM_PHI[m][3], x, y);
    }
    return idx;
}

// Bin of a neighbour given its frame coordinates (x_, y_, z_), r_ = ||.|| and r = r_/mean_dist,
// both already computed by the caller in float64 (shape_context.py:29-30).
PM_HD int pm_bin_index(double x_, double y_, double z_, double r_, double r) {
    double c = z_ / r_;                                   // :31 argument of arccos
    if (!(__builtin_fabs(c) <= 1.0)) return PM_DROP;      // arccos -> NaN (r_ = 0, NaN input, |c| > 1 by rounding)
    if (x_ != x_ || y_ != y_) return PM_DROP;             // atan2 -> NaN
    int th = (c <= PM_CTH[0]) + (c <= PM_CTH[1]) + (c <= PM_CTH[2]) + (c <= PM_CTH[3]) + (c <= PM_CTH[4]) + (c <= PM_CTH[5]);
    int ring = 4;                                         // :49 default, also for NaN r
    if (r < PM_REDGE0) ring = 0;
    else if (r < PM_REDGE1) ring = 1;
    else if (r < PM_REDGE2) ring = 2;
    else if (r < PM_REDGE3) ring = 3;
    int idx = ring * 72 + th * 12 + pm_phi_index(x_, y_);
    return idx < PM_NBINS ? idx : PM_DROP;
}
