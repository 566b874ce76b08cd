// pm_solve.h — small dense solves done by one lane in registers.
// Least-squares affine [fixed;1] ~ A [moving;1] (get_affine_transform, find_transform.py:4-17) from
// moment sums: with centred moments C_mm = sum (m-mbar)(m-mbar)^T and C_fm = sum (f-fbar)(m-mbar)^T,
// the linear part is L = C_fm C_mm^-1 and the translation t = fbar - L mbar.
#pragma once
#include "pm_common.h"

namespace pm {

// Conditioning thresholds below which a fit is reported as degenerate (callers then refit with the reference's own
// pinv on the host, which returns the minimum-norm answer for rank-deficient input: find_transform.py:17).
//   PM_DEGENERATE_MOMENTS: det(Cmm) / (trace(Cmm)/3)^3 of the centred 3x3 moment matrix (1 for an isotropic cloud, 0 for a
//                          planar one); the normal-equation solve loses about eps / ratio relative accuracy.  1e-5 (round 3; 1e-8
//                          before): a fit worse than ~1e-11 is not worth its speed — an ICP run whose cloud flattens (far more
//                          moving than fixed points) amplified a 1e-9 difference to 1e-3 within a few iterations, where the
//                          reference's pinv chain (what the caller reruns with) stays on track.  Clouds flatter than ~500:500:1
//                          take that slower path.
//   PM_DEGENERATE_SIMPLEX: |det D| / (|d1| |d2| |d3|) of the edge matrix of a 4-point sample (Hadamard ratio).
#define PM_DEGENERATE_MOMENTS 1e-5
#define PM_DEGENERATE_MOMENTS_SAMPLE 1e-6
#define PM_DEGENERATE_SIMPLEX 1e-6

// L (3x3 row-major) = Cfm . inverse(Cmm), Cmm symmetric given as {00,01,02,11,12,22}.
// Cofactor inverse; a singular Cmm (coplanar points) yields inf/NaN.  Returns det / (trace/3)^3 (NaN-propagating), the
// conditioning measure callers compare with PM_DEGENERATE_MOMENTS.
__device__ __forceinline__ double solve_sym3(const double cmm[6], const double cfm[9], double L[9]) {
    const double a = cmm[0], b = cmm[1], c = cmm[2], d = cmm[3], e = cmm[4], f = cmm[5];
    const double i00 = d * f - e * e, i01 = c * e - b * f, i02 = b * e - c * d;
    const double i11 = a * f - c * c, i12 = b * c - a * e, i22 = a * d - b * b;
    const double det = (a * i00 + b * i01) + c * i02;
    const double rdet = 1.0 / det;       // one division: this runs in a single lane on the critical path of every ICP iteration
    const double inv[9] = {i00 * rdet, i01 * rdet, i02 * rdet, i01 * rdet, i11 * rdet, i12 * rdet, i02 * rdet, i12 * rdet, i22 * rdet};
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int q = 0; q < 3; ++q)
            L[3 * r + q] = (cfm[3 * r] * inv[q] + cfm[3 * r + 1] * inv[3 + q]) + cfm[3 * r + 2] * inv[6 + q];
    const double tr3 = ((a + d) + f) * (1.0 / 3.0);
    return det / ((tr3 * tr3) * tr3);
}

// The affine through exactly four pairs (get_affine_transform on a square system, find_transform.py:4-17): with edge
// matrices Dm = [m1-m0, m2-m0, m3-m0], Df likewise (columns), L = Df Dm^-1 and t = f0 - L m0.  Conditioned like Dm
// itself (the moment form squares it).  m, f: [4][3].  Returns the Hadamard ratio of Dm (0 for coplanar samples).
__device__ __forceinline__ double affine_from_4(const double m[4][3], const double f[4][3], double A[16]) {
    double dm[3][3], df[3][3];          // [edge][coordinate]
#pragma unroll
    for (int e = 0; e < 3; ++e)
#pragma unroll
        for (int c = 0; c < 3; ++c) { dm[e][c] = m[e + 1][c] - m[0][c]; df[e][c] = f[e + 1][c] - f[0][c]; }
    // cross products = rows of det * Dm^-1 (Dm has the edges as columns: Dm[c][e] = dm[e][c])
    const double c0[3] = {dm[1][1] * dm[2][2] - dm[1][2] * dm[2][1], dm[1][2] * dm[2][0] - dm[1][0] * dm[2][2], dm[1][0] * dm[2][1] - dm[1][1] * dm[2][0]};
    const double c1[3] = {dm[2][1] * dm[0][2] - dm[2][2] * dm[0][1], dm[2][2] * dm[0][0] - dm[2][0] * dm[0][2], dm[2][0] * dm[0][1] - dm[2][1] * dm[0][0]};
    const double c2[3] = {dm[0][1] * dm[1][2] - dm[0][2] * dm[1][1], dm[0][2] * dm[1][0] - dm[0][0] * dm[1][2], dm[0][0] * dm[1][1] - dm[0][1] * dm[1][0]};
    const double det = (dm[0][0] * c0[0] + dm[0][1] * c0[1]) + dm[0][2] * c0[2];
    const double rdet = 1.0 / det;
    // inverse rows: inv[e][c] = cross_e[c] / det  (inv . Dm = I: row e dotted with edge e' is delta)
    double L[9];
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = 0; c < 3; ++c)
            L[3 * r + c] = ((df[0][r] * c0[c] + df[1][r] * c1[c]) + df[2][r] * c2[c]) * rdet;
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        A[4 * r] = L[3 * r]; A[4 * r + 1] = L[3 * r + 1]; A[4 * r + 2] = L[3 * r + 2];
        A[4 * r + 3] = f[0][r] - ((L[3 * r] * m[0][0] + L[3 * r + 1] * m[0][1]) + L[3 * r + 2] * m[0][2]);
    }
    A[12] = 0.0; A[13] = 0.0; A[14] = 0.0; A[15] = 1.0;
    const double n0 = (dm[0][0] * dm[0][0] + dm[0][1] * dm[0][1]) + dm[0][2] * dm[0][2];
    const double n1 = (dm[1][0] * dm[1][0] + dm[1][1] * dm[1][1]) + dm[1][2] * dm[1][2];
    const double n2 = (dm[2][0] * dm[2][0] + dm[2][1] * dm[2][1]) + dm[2][2] * dm[2][2];
    return fabs(det) / __builtin_sqrt((n0 * n1) * n2);
}

// Least-squares affine from CENTRED moments (cmm, cfm about the means mb, fb; absolute coordinates): L = Cfm Cmm^-1,
// t = fb - L mb.  Returns the conditioning ratio of solve_sym3.
__device__ __forceinline__ double affine_from_centred(const double cmm[6], const double cfm[9], const double mb[3], const double fb[3],
                                                      double A[16]) {
    double L[9];
    const double ratio = solve_sym3(cmm, cfm, L);
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        A[4 * r] = L[3 * r]; A[4 * r + 1] = L[3 * r + 1]; A[4 * r + 2] = L[3 * r + 2];
        A[4 * r + 3] = fb[r] - ((L[3 * r] * mb[0] + L[3 * r + 1] * mb[1]) + L[3 * r + 2] * mb[2]);
    }
    A[12] = 0.0; A[13] = 0.0; A[14] = 0.0; A[15] = 1.0;
    return ratio;
}

// ---- the fixed reduction tree of the ICP moment sums ---------------------------------------------------------------------
// Every sum over the moving points (22 moment slots + the residual) is taken in ONE order wherever it is computed — the
// stand-alone kernels of pm_transform.hip, the fused iteration kernel of pm_icp_grid.hip, any number of ranks:
//   leaf   = the PM_TREE_LEAF (8) consecutive points 8b .. 8b+7, added one after the other starting from 0.0;
//   group  = the PM_TREE_GROUP (64) consecutive leaves 64g .. 64g+63 (512 points), added in leaf order;
//   total  = the groups, added in group order.
// Points past the end of the cloud contribute +0.0.  Same additions in the same order => same bits.
#define PM_TREE_LEAF 8
#define PM_TREE_GROUP 64
#define PM_TREE_POINTS (PM_TREE_LEAF * PM_TREE_GROUP)
#define PM_NMOMENTS 22          // slots 1..22 of the PM_ICP_NSUMS layout (slot 0 is the count)

// The 22 per-point terms: a = moving - origin_m, f = matched fixed - origin_f.
__device__ __forceinline__ void moment_terms(double a0, double a1, double a2, double f0, double f1, double f2, double s[PM_NMOMENTS]) {
    s[0] = a0; s[1] = a1; s[2] = a2;
    s[3] = f0; s[4] = f1; s[5] = f2;
    s[6] = a0 * a0; s[7] = a0 * a1; s[8] = a0 * a2; s[9] = a1 * a1; s[10] = a1 * a2; s[11] = a2 * a2;
    s[12] = f0 * a0; s[13] = f0 * a1; s[14] = f0 * a2;
    s[15] = f1 * a0; s[16] = f1 * a1; s[17] = f1 * a2;
    s[18] = f2 * a0; s[19] = f2 * a1; s[20] = f2 * a2;
    s[21] = (f0 * f0 + f1 * f1) + f2 * f2;
}

// One row of apply_affine_transform (apply_transform.py:14-17: np.matmul(A, [p; 1])) in np.matmul's own arithmetic: BLAS dgemm's
// x86-64 kernels run the k = 0..3 products of an output element as one chain of fused multiply-adds, starting from the rounded
// first product.  Every kernel that moves a point uses this form, so that a cloud carries the reference's bits wherever the
// 4 x 4 does (the RANSAC winner; pinv fits of a degenerate ICP step, whose next fit amplifies the cloud's last bits by 1e11).
__device__ __forceinline__ double affine_row(const double *A4, double x, double y, double z) {
    double acc = A4[0] * x;
    acc = __builtin_fma(A4[1], y, acc);
    acc = __builtin_fma(A4[2], z, acc);
    return __builtin_fma(A4[3], 1.0, acc);
}

// A_icp <- A_est . A_icp (perform_icp.py:25, np.matmul), all four rows, in np.matmul's arithmetic (dgemm's multiply-add chain, as
// affine_row).  It matters when a degenerate run's pinv fits carry entries of 1e10 that cancel in this product: the plain
// multiply-and-add form left the composed 4x4 1e-3 from the reference's although every cloud along the way had its bits.
__device__ __forceinline__ void compose_affine(const double A[16], double *A_icp16) {
    double C[16];
    for (int r = 0; r < 4; ++r)
        for (int c = 0; c < 4; ++c) {
            double t = A[4 * r] * A_icp16[c];
            for (int k = 1; k < 4; ++k) t = __builtin_fma(A[4 * r + k], A_icp16[4 * k + c], t);
            C[4 * r + c] = t;
        }
    for (int k = 0; k < 16; ++k) A_icp16[k] = C[k];
}

// sums: PM_ICP_NSUMS layout about origin6 = {origin_m(3), origin_f(3)} -> A (4x4 row-major).
// Returns the conditioning ratio of solve_sym3 (compare with PM_DEGENERATE_MOMENTS).
__device__ __forceinline__ double affine_from_sums(const double *sums, const double *origin6, double A[16]) {
    const double n = sums[0], rn = 1.0 / n;
    const double mb[3] = {sums[1] * rn, sums[2] * rn, sums[3] * rn};
    const double fb[3] = {sums[4] * rn, sums[5] * rn, sums[6] * rn};
    double cmm[6], cfm[9];
    cmm[0] = sums[7] - n * mb[0] * mb[0];
    cmm[1] = sums[8] - n * mb[0] * mb[1];
    cmm[2] = sums[9] - n * mb[0] * mb[2];
    cmm[3] = sums[10] - n * mb[1] * mb[1];
    cmm[4] = sums[11] - n * mb[1] * mb[2];
    cmm[5] = sums[12] - n * mb[2] * mb[2];
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int q = 0; q < 3; ++q) cfm[3 * r + q] = sums[13 + 3 * r + q] - n * fb[r] * mb[q];
    double L[9];
    const double ratio = solve_sym3(cmm, cfm, L);
    const double ma[3] = {mb[0] + origin6[0], mb[1] + origin6[1], mb[2] + origin6[2]};
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        A[4 * r] = L[3 * r]; A[4 * r + 1] = L[3 * r + 1]; A[4 * r + 2] = L[3 * r + 2];
        A[4 * r + 3] = (fb[r] + origin6[3 + r]) - ((L[3 * r] * ma[0] + L[3 * r + 1] * ma[1]) + L[3 * r + 2] * ma[2]);
    }
    A[12] = 0.0; A[13] = 0.0; A[14] = 0.0; A[15] = 1.0;
    return ratio;
}

}  // namespace pm
