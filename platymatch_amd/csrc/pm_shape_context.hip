// pm_shape_context.hip — per-nucleus 3D shape-context histograms.
// Reference: get_unary (shape_context.py:144-188) with get_Y (:6-8), transform (:61-84),
// get_shape_context (:10-42) and get_bin_index (:46-58).
//
// One workgroup per queried point.  Its local frame is wave-uniform; the cloud streams through
// in coalesced structure-of-arrays loads (the 3 x N layout of the reference), each lane bins one
// neighbour per step for all frames, and counts go to per-wave LDS histograms with ds_add_u32.
// `transform` builds T = B.inv(A) mapping p_i -> 0, p_i + x,y,z -> e1,e2,e3; for the orthonormal
// frame that is [x y z]^T (p_j - p_i), evaluated here directly.
// Frames 2..4 differ from frame 1 only in the signs of (x, y) (:172-175, :180-181), so ring and theta are
// shared; the phi sectors of all frames come from one classification when the neighbour is clear of every
// sector edge and from exact per-frame sign tests otherwise (pm_bin_index4, pm_binning.h).
#include "pm_common.h"
#include "pm_binning.h"

namespace pm {

constexpr int SC_THREADS = 256;
constexpr int SC_WAVES = SC_THREADS / 64;

template <int NF>
__global__ __launch_bounds__(SC_THREADS) void shape_context_kernel(
    const double *__restrict__ xyz, int n, int row0, const double *__restrict__ centroid3,
    const double *__restrict__ x0_3, const double *__restrict__ mean_dist1, int32_t *__restrict__ counts,
    int32_t *__restrict__ totals, double *__restrict__ hist, int nrows) {
    __shared__ unsigned int h[SC_WAVES][NF][PM_NBINS];
    __shared__ unsigned int tot_s[NF];
    const int tid = threadIdx.x, wave = tid >> 6;
    const int row = blockIdx.x;       // row within this block of rows
    const int i = row0 + row;         // queried point
    for (int k = tid; k < SC_WAVES * NF * PM_NBINS; k += SC_THREADS) (&h[0][0][0])[k] = 0u;
    if (tid < NF) tot_s[tid] = 0u;

    const double *P0 = xyz, *P1 = xyz + (size_t)n, *P2 = xyz + 2 * (size_t)n;
    const double p0 = P0[i], p1 = P1[i], p2 = P2[i];
    const double md = mean_dist1[0];
    // local frame (shape_context.py:169-175); every lane computes the same values
    double w0 = p0 - centroid3[0], w1 = p1 - centroid3[1], w2 = p2 - centroid3[2];
    double nw = __builtin_sqrt((w0 * w0 + w1 * w1) + w2 * w2);
    const double z0 = w0 / nw, z1 = w1 / nw, z2 = w2 / nw;
    const double a0 = x0_3[0], a1 = x0_3[1], a2 = x0_3[2];
    double d = (a0 * z0 + a1 * z1) + a2 * z2;
    double x0 = a0 - z0 * d, x1 = a1 - z1 * d, x2 = a2 - z2 * d;
    double nx = __builtin_sqrt((x0 * x0 + x1 * x1) + x2 * x2);
    x0 /= nx; x1 /= nx; x2 /= nx;
    double y0 = z1 * x2 - z2 * x1, y1 = z2 * x0 - z0 * x2, y2 = z0 * x1 - z1 * x0;   // get_Y
    double ny = __builtin_sqrt((y0 * y0 + y1 * y1) + y2 * y2);
    y0 /= ny; y1 /= ny; y2 /= ny;
    __syncthreads();

    double rho[4];
    pm_ring_thresholds(md, rho);       // r_/md < edge  <=>  r_ < rho (exact), computed once per workgroup

    for (int j = tid; j < n; j += SC_THREADS) {
        if (j == i) continue;                                  // np.delete (:168)
        const double v0 = P0[j] - p0, v1 = P1[j] - p1, v2 = P2[j] - p2;
        const double vx = (x0 * v0 + x1 * v1) + x2 * v2;
        const double vy = (y0 * v0 + y1 * v1) + y2 * v2;
        const double vz = (z0 * v0 + z1 * v1) + z2 * v2;
        int b[4];
        pm_bin_index4(vx, vy, vz, rho, NF, b);
#pragma unroll
        for (int f = 0; f < NF; ++f)
            if (b[f] != PM_DROP) atomicAdd(&h[wave][f][b[f]], 1u);
    }
    __syncthreads();
    // fold the per-wave histograms, total per frame
    for (int k = tid; k < NF * PM_NBINS; k += SC_THREADS) {
        const int f = k / PM_NBINS, bin = k - f * PM_NBINS;
        unsigned int c = 0;
#pragma unroll
        for (int w = 0; w < SC_WAVES; ++w) c += h[w][f][bin];
        h[0][f][bin] = c;
        if (c) atomicAdd(&tot_s[f], c);
    }
    __syncthreads();
    for (int k = tid; k < NF * PM_NBINS; k += SC_THREADS) {
        const int f = k / PM_NBINS, bin = k - f * PM_NBINS;
        const unsigned int c = h[0][f][bin];
        const size_t o = ((size_t)f * nrows + row) * PM_NBINS + bin;
        if (counts) counts[o] = (int32_t)c;
        if (hist) hist[o] = (double)c / (double)tot_s[f];    // sc / sc.sum() (:41); 0/0 = NaN as in the reference
    }
    if (totals && tid < NF) totals[(size_t)tid * nrows + row] = (int32_t)tot_s[tid];
}

// get_shape_context on an explicit, already transformed neighbour list (one workgroup)
__global__ __launch_bounds__(SC_THREADS) void neighbors_kernel(const double *__restrict__ nb, int n, double md,
                                                               int32_t *__restrict__ counts, int32_t *__restrict__ total,
                                                               double *__restrict__ hist) {
    __shared__ unsigned int h[PM_NBINS];
    __shared__ unsigned int tot;
    for (int k = threadIdx.x; k < PM_NBINS; k += SC_THREADS) h[k] = 0u;
    if (threadIdx.x == 0) tot = 0u;
    __syncthreads();
    for (int j = threadIdx.x; j < n; j += SC_THREADS) {
        const double vx = nb[3 * (size_t)j], vy = nb[3 * (size_t)j + 1], vz = nb[3 * (size_t)j + 2];
        const double r_ = __builtin_sqrt((vx * vx + vy * vy) + vz * vz);
        const int b = pm_bin_index(vx, vy, vz, r_, r_ / md);
        if (b != PM_DROP) { atomicAdd(&h[b], 1u); atomicAdd(&tot, 1u); }
    }
    __syncthreads();
    for (int k = threadIdx.x; k < PM_NBINS; k += SC_THREADS) {
        if (counts) counts[k] = (int32_t)h[k];
        if (hist) hist[k] = (double)h[k] / (double)tot;
    }
    if (total && threadIdx.x == 0) total[0] = (int32_t)tot;
}

}  // namespace pm

extern "C" int pm_shape_context_neighbors(const double *nb, int n, double mean_dist, int32_t *counts, int32_t *total,
                                          double *hist, void *stream) {
    if (!nb || n <= 0 || (!counts && !hist)) return PM_ERR_INVALID_ARG;
    pm::neighbors_kernel<<<1, pm::SC_THREADS, 0, (hipStream_t)stream>>>(nb, n, mean_dist, counts, total, hist);
    return pm::launch_status();
}

extern "C" int pm_shape_context(const double *xyz, int n, int row0, int nrows, const double *centroid3,
                                const double *x0_3, const double *mean_dist1, int n_frames, int32_t *counts,
                                int32_t *totals, double *hist, void *stream) {
    if (!xyz || !centroid3 || !x0_3 || !mean_dist1 || n <= 0 || row0 < 0 || nrows < 0 || row0 + nrows > n)
        return PM_ERR_INVALID_ARG;
    if (n_frames != 2 && n_frames != 4) return PM_ERR_INVALID_ARG;
    if (!counts && !totals && !hist) return PM_ERR_INVALID_ARG;
    if (nrows == 0) return PM_OK;
    hipStream_t s = (hipStream_t)stream;
    if (n_frames == 2)
        pm::shape_context_kernel<2><<<nrows, pm::SC_THREADS, 0, s>>>(xyz, n, row0, centroid3, x0_3, mean_dist1, counts, totals, hist, nrows);
    else
        pm::shape_context_kernel<4><<<nrows, pm::SC_THREADS, 0, s>>>(xyz, n, row0, centroid3, x0_3, mean_dist1, counts, totals, hist, nrows);
    return pm::launch_status();
}
