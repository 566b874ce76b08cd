// pm_solve.h — small dense solves done by one lane in registers.
// Least-squares affine [fixed;1] ~ A [moving;1] (get_affine_transform, find_transform.py:4-17) from
// moment sums: with centred moments C_mm = sum (m-mbar)(m-mbar)^T and C_fm = sum (f-fbar)(m-mbar)^T,
// the linear part is L = C_fm C_mm^-1 and the translation t = fbar - L mbar.
#pragma once
#include "pm_common.h"

namespace pm {

// L (3x3 row-major) = Cfm . inverse(Cmm), Cmm symmetric given as {00,01,02,11,12,22}.
// Cofactor inverse; a singular Cmm (coplanar points) yields inf/NaN, which callers treat as "no fit".
__device__ __forceinline__ void solve_sym3(const double cmm[6], const double cfm[9], double L[9]) {
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
}

// sums: PM_ICP_NSUMS layout about origin6 = {origin_m(3), origin_f(3)} -> A (4x4 row-major).
__device__ __forceinline__ void affine_from_sums(const double *sums, const double *origin6, double A[16]) {
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
    solve_sym3(cmm, cfm, L);
    const double ma[3] = {mb[0] + origin6[0], mb[1] + origin6[1], mb[2] + origin6[2]};
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        A[4 * r] = L[3 * r]; A[4 * r + 1] = L[3 * r + 1]; A[4 * r + 2] = L[3 * r + 2];
        A[4 * r + 3] = (fb[r] + origin6[3 + r]) - ((L[3 * r] * ma[0] + L[3 * r + 1] * ma[1]) + L[3 * r + 2] * ma[2]);
    }
    A[12] = 0.0; A[13] = 0.0; A[14] = 0.0; A[15] = 1.0;
}

}  // namespace pm
