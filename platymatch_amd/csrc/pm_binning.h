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

// ---- all frames of get_unary at once (the shape-context kernel's inner step) ------------------------------
// Frames 2..4 see the same neighbour as (-x,-y), (x,-y), (-x,y) (shape_context.py:172-175, 180-181): same ring,
// same theta, and — whenever (x, y) is clear of every sector edge — phi sectors that are fixed permutations of
// frame 1's.  pm_phi_index4 classifies frame 1 with two multiplications (|y| against tan30*|x|, tan60*|x|), accepts
// the result only if the point is at least 2^-40 (relative) away from all twelve rays — far more than the
// ~1e-15 by which the exact thresholds and this arithmetic can differ — and otherwise falls back to the exact
// sign tests per frame.  Ring thresholds on r_ replace the division r_/mean_dist (pm_ring_thresholds).

PM_HD double pm_next_up(double t) {       // t > 0 finite
    long long b; __builtin_memcpy(&b, &t, 8); b += 1; __builtin_memcpy(&t, &b, 8); return t;
}
PM_HD double pm_next_down(double t) {     // t > 0
    long long b; __builtin_memcpy(&b, &t, 8); b -= 1; __builtin_memcpy(&t, &b, 8); return t;
}

// rho[k] = smallest float64 r_ with fl(r_ / md) >= edge k, so that  #{k : r_ >= rho[k]}  is the reference's
// r_index (first edge with r < edge, else 4; shape_context.py:49-56) for every r_, bit for bit.
PM_HD void pm_ring_thresholds(double md, double rho[4]) {
    const double e[4] = {PM_REDGE0, PM_REDGE1, PM_REDGE2, PM_REDGE3};
    for (int k = 0; k < 4; ++k) {
        const double t0 = e[k] * md;
        // mean distance 0 or NaN (r = inf / NaN: `r < edge` never holds -> ring 4), or e*md underflowed: every r_ passes
        if (!(md > 0.0) || !(t0 > 0.0)) { rho[k] = -1.0; continue; }
        // e*md overflowed or md = inf (r = 0 for every finite r_ -> ring 0): nothing passes
        if (!(t0 < 0x1p+1020)) { rho[k] = t0; continue; }
        double t = t0;
        for (int it = 0; it < 8; ++it) { const double p = pm_next_down(t); if (p / md >= e[k]) t = p; else break; }
        for (int it = 0; it < 8; ++it) { if (t / md >= e[k]) break; t = pm_next_up(t); }
        rho[k] = t;
    }
}

#define PM_TAN30 0x1.279a74590331cp-1
#define PM_TAN60 0x1.bb67ae8584caap+0

// phi sectors of (x,y), (-x,-y), (x,-y), (-x,y).  Exact in all cases; fast when clear of the edges.
PM_HD void pm_phi_index4(double x, double y, int nframes, int p[4]) {
    const double ax = __builtin_fabs(x), ay = __builtin_fabs(y);
    const double sum = ax + ay;
    const double m = sum * 0x1p-40;
    const double d1 = ay - PM_TAN30 * ax, d2 = ay - PM_TAN60 * ax;
    const int safe = (__builtin_fabs(d1) > m) & (__builtin_fabs(d2) > m) & (ax > m) & (ay > m) & (sum > 0x1p-900) & (sum < 0x1p+900);
    if (safe) {
        const int k = (d1 > 0.0) + (d2 > 0.0);
        const int p0 = (y > 0.0) ? ((x > 0.0) ? k : 5 - k) : ((x > 0.0) ? 11 - k : 6 + k);
        p[0] = p0;
        p[1] = (p0 < 6) ? p0 + 6 : p0 - 6;
        p[2] = 11 - p0;
        p[3] = (p0 < 6) ? 5 - p0 : 17 - p0;
    } else {
        p[0] = pm_phi_index(x, y);
        p[1] = pm_phi_index(-x, -y);
        p[2] = (nframes > 2) ? pm_phi_index(x, -y) : 0;
        p[3] = (nframes > 2) ? pm_phi_index(-x, y) : 0;
    }
}

// Bins of one neighbour (frame coordinates x_, y_, z_) in frames 1..nframes: out[f] = bin or PM_DROP.
//
// The reference takes r_ = sqrt(s), s = (x_^2 + y_^2) + z_^2 (:29), compares r_/mean_dist with the ring edges and
// c = z_/r_ (the argument of arccos, :31) with the theta steps: a square root and a division per neighbour, ~40 % of
// this function's instructions.  Both comparisons are monotone and can be made on s instead:
//     r_ >= rho   <=>  s >= rho^2                 c <= T  <=>  z_|z_| <= T|T| s      (u|u| is increasing, s = r_^2 > 0)
// up to the roundings of sqrt, the division and the products (each < 2^-51 relative).  A decision is therefore taken on
// s only if it is clear by 2^-40 (relative to s) of EVERY ring and theta step, and s is finite, normal and non-zero;
// otherwise — one neighbour in ~1e9 for generic data — the reference's own expressions decide.
PM_HD void pm_bin_index4(double x_, double y_, double z_, const double rho[4], int nframes, int out[4]) {
    const double s = (x_ * x_ + y_ * y_) + z_ * z_;
    const double zz = z_ * __builtin_fabs(z_), m = s * 0x1p-40;
    int safe = (s > 0x1p-900) & (s < 0x1p+900);           // also false for NaN coordinates
    int th = 0, ring = 0;
#pragma unroll
    for (int k = 0; k < 6; ++k) {
        const double dd = zz - (PM_CTH[k] * __builtin_fabs(PM_CTH[k])) * s;
        safe &= (__builtin_fabs(dd) > m);
        th += (dd <= 0.0);
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const double dd = s - ((rho[k] < 0.0) ? -1.0 : rho[k] * rho[k]);      // rho < 0: every r_ passes (pm_ring_thresholds)
        safe &= (__builtin_fabs(dd) > m);
        ring += (dd >= 0.0);
    }
    if (!safe) {
        const double r_ = __builtin_sqrt(s);                               // :29
        const double c = z_ / r_;                                          // :31 argument of arccos
        if (!(__builtin_fabs(c) <= 1.0) || x_ != x_ || y_ != y_) {         // arccos / atan2 -> NaN: not counted
            out[0] = out[1] = out[2] = out[3] = PM_DROP;
            return;
        }
        th = (c <= PM_CTH[0]) + (c <= PM_CTH[1]) + (c <= PM_CTH[2]) + (c <= PM_CTH[3]) + (c <= PM_CTH[4]) + (c <= PM_CTH[5]);
        ring = (r_ >= rho[0]) + (r_ >= rho[1]) + (r_ >= rho[2]) + (r_ >= rho[3]);
    }
    const int base = ring * 72 + th * 12;
    int p[4];
    pm_phi_index4(x_, y_, nframes, p);
    for (int f = 0; f < 4; ++f) {
        const int idx = base + p[f];
        out[f] = (f < nframes && idx < PM_NBINS) ? idx : PM_DROP;
    }
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

// ---- single-precision pre-classification (the tile kernel's inner step, pm_shape_context.hip) -------------------------
// Which bin a neighbour falls into is a set of sign decisions: which side of four spheres (rings), of four cones and a plane
// (theta) and of twelve half-planes (phi sectors) it lies on.  A decision taken in float32 is THE decision — the one the
// float64 expressions above reach, which are the oracle's — whenever the float32 quantity is clear of its threshold by more
// than float32 arithmetic can be off.  With v the float64 difference vector rounded to float32 (f), the frame rounded to
// float32, L = |f|_1 and s = |v|^2:
//   linear quantities (x_, y_, z_, y_ - tan(30|60) x_)   are off by < 2^-19.9 L   -> accepted when clear by mL = 2^-17 L
//   quadratic ones (4 z_^2 - s, 4 z_^2 - 3 s)            are off by < 2^-15.9 s   -> accepted when clear by mQ = 2^-14 s
//   w = s * 64 / md^2 (ring = position of w among 1, 4, 16, 64) is off by < 2^-17.9 w -> accepted when w's mantissa is clear
//   of a power of four by 2^-15
// (the thresholds' own representation — cos(k pi/6) rounded, np.logspace's ulp-off edges, fl(2 pi/12) steps — moves them
// by < 2^-50, far inside every margin).  Frames 2..4 are then the fixed phi permutations of frame 1 (x_, y_ clear of zero).
// About one neighbour in 4 000 is NOT clear: -1 is returned and the caller decides it with pm_bin_index4 in float64.
// theta from t = z_|z_| / s:  thresholds cos^2(30) = 3/4, cos^2(60) = 1/4, 0 and their negatives.
#define PM_TAN30F 0x1.279a74p-1f
#define PM_TAN60F 0x1.bb67aep+0f

// min_L: the float32 result is taken only for neighbours with |f|_1 > min_L.  The tile kernel passes 2^17 x the ABSOLUTE noise the
// reference's inv()-based local coordinates carry for this query (PM_GUARD_REF x |p|_1, pm_shape_context.hip): a neighbour clear
// of a boundary by 2^-17 L (relative to ITS OWN length) is then also clear of the reference's noise band, so whatever float32
// decides lies outside the edge guard's reach; nearer neighbours (1e-5 away at coordinates of ~500) go to the float64 path, where
// the guard counts them.  0 = no such floor (the host harness's default).
PM_HD int pm_bin_fast32(float f0, float f1, float f2, const float fr[9], float k64, float min_L) {
    const float L = (__builtin_fabsf(f0) + __builtin_fabsf(f1)) + __builtin_fabsf(f2);
    const float vx = __builtin_fmaf(fr[2], f2, __builtin_fmaf(fr[1], f1, fr[0] * f0));
    const float vy = __builtin_fmaf(fr[5], f2, __builtin_fmaf(fr[4], f1, fr[3] * f0));
    const float vz = __builtin_fmaf(fr[8], f2, __builtin_fmaf(fr[7], f1, fr[6] * f0));
    const float q2 = vz * vz;
    const float s = __builtin_fmaf(vx, vx, __builtin_fmaf(vy, vy, q2));
    const float a = 4.0f * q2;
    const float dd1 = a - s, dd3 = __builtin_fmaf(-3.0f, s, a);
    const float mQ = s * 0x1p-14f, mL = L * 0x1p-17f;
    const float ax = __builtin_fabsf(vx), ay = __builtin_fabsf(vy);
    const float d1 = __builtin_fmaf(-PM_TAN30F, ax, ay), d2 = __builtin_fmaf(-PM_TAN60F, ax, ay);
    const float w = s * k64;
    unsigned int wb;
    __builtin_memcpy(&wb, &w, 4);
    const unsigned int e = wb >> 23;                       // (w >= 0 or NaN: no sign bit to strip unless NaN, rejected below)
    const int E = (int)e - 127;                            // floor(log2 w)
    const unsigned int frac = wb & 0x7fffffu;
    const unsigned int dist = (E & 1) ? (0x7fffffu - frac) : frac;     // distance (in ulps of w) to the nearest power of FOUR boundary
    // clear of every boundary?  (minima first: five comparisons instead of nine — on the GPU every comparison result is a lane
    // mask in scalar registers and every `and` of two masks a scalar instruction, and a CU has ONE scalar unit for its four
    // SIMDs.  fminf drops a NaN operand, but a NaN or infinite coordinate makes s — hence w — NaN or infinite: e = 255 fails.)
    const float nearQ = __builtin_fminf(__builtin_fabsf(dd1), __builtin_fabsf(dd3));
    const float nearL = __builtin_fminf(__builtin_fminf(__builtin_fabsf(vz), __builtin_fabsf(d1)),
                                        __builtin_fminf(__builtin_fabsf(d2), __builtin_fminf(ax, ay)));
    const int safe = (nearQ > mQ) & (nearL > mL) & ((e - 40u) < 176u) & (dist >= 256u) & (L > min_L);
    int ring = (E >> 1) + 1;                               // #{k : w >= 4^k, k = 0..3}
    ring = ring < 0 ? 0 : (ring > 4 ? 4 : ring);
    const int c = (dd1 > 0.0f) + (dd3 > 0.0f);
    const int th = (vz > 0.0f) ? 2 - c : 3 + c;
    const int k = (d1 > 0.0f) + (d2 > 0.0f);
    const int u = (vx > 0.0f) ? k : 5 - k;                 // sector within the upper half plane, 0..5
    const int p0 = (vy > 0.0f) ? u : 11 - u;               // (computed for every lane, selected below: no divergent branch)
    return safe ? ring * 72 + th * 12 + p0 : -1;
}

// phi sector of frame f (1..3) given frame 1's (the permutations of get_unary's four frames, shape_context.py:172-181) and back
PM_HD int pm_phi_perm(int f, int p0) {
    return f == 0 ? p0 : f == 1 ? (p0 < 6 ? p0 + 6 : p0 - 6) : f == 2 ? 11 - p0 : (p0 < 6 ? 5 - p0 : 17 - p0);
}
// bin of frame f for a neighbour that frame 1 puts into `bin` (every permutation above is an involution or the half turn,
// which is its own inverse: the same map reads a frame-f bin back to frame 1's)
PM_HD int pm_bin_perm(int f, int bin) {
    const int shell = bin / 12, p0 = bin - shell * 12;
    return shell * 12 + pm_phi_perm(f, p0);
}
