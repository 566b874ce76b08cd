// pm_stats.hip — cloud statistics: centroid, mean pairwise distance, first PCA axis.
// Reference: utils/utils.py:48-56 (get_centroid), :58-75 (get_mean_distance),
// shape_context.py:162-165 (PCA(3).fit(detections).components_[0]).
#include "pm_common.h"

namespace pm {

constexpr int STAT_THREADS = 1024;
constexpr int MD_TILE = 256;

// ---- centroid: one block, fixed reduction tree --------------------------------------------------
__global__ __launch_bounds__(STAT_THREADS) void centroid_kernel(const double *__restrict__ xyz, int n,
                                                                double *__restrict__ out3) {
    __shared__ double scratch[STAT_THREADS / 64];
    for (int c = 0; c < 3; ++c) {
        const double *row = xyz + (size_t)c * n;
        double s = 0.0;
        for (int i = threadIdx.x; i < n; i += STAT_THREADS) s += row[i];
        double tot = block_sum(s, scratch);
        if (threadIdx.x == 0) out3[c] = tot / (double)n;
    }
}

// ---- mean pairwise distance: upper-triangle tiles, then an ordered sum of the tile partials -------
// One thread owns point i of tile bi and walks tile bj (staged in LDS, broadcast reads).
// Tile rows row_offset, row_offset + row_stride, ... (ranks of a sharded run interleave the rows: row bi holds T - bi tiles).
__global__ __launch_bounds__(MD_TILE) void mean_distance_tiles(const double *__restrict__ xyz, int n,
                                                               double *__restrict__ partial, int row_offset, int row_stride) {
    const int bi = row_offset + blockIdx.y * row_stride, bj = blockIdx.x, T = gridDim.x;
    if (bj < bi) return;  // partial[] for these is never read
    __shared__ double tj[3][MD_TILE];
    __shared__ double scratch[MD_TILE / 64];
    const int tid = threadIdx.x;
    const int gi = bi * MD_TILE + tid, gj0 = bj * MD_TILE;
    for (int c = 0; c < 3; ++c) tj[c][tid] = (gj0 + tid < n) ? xyz[(size_t)c * n + gj0 + tid] : 0.0;
    double p0 = 0, p1 = 0, p2 = 0;
    if (gi < n) { p0 = xyz[gi]; p1 = xyz[(size_t)n + gi]; p2 = xyz[2 * (size_t)n + gi]; }
    __syncthreads();
    double s = 0.0;
    const int jn = min(MD_TILE, n - gj0);
    if (gi < n) {
        for (int j = 0; j < jn; ++j) {
            double d0 = p0 - tj[0][j], d1 = p1 - tj[1][j], d2 = p2 - tj[2][j];
            double d = __builtin_sqrt((d0 * d0 + d1 * d1) + d2 * d2);
            s += (gj0 + j > gi) ? d : 0.0;
        }
    }
    double tot = block_sum(s, scratch);
    if (tid == 0) partial[(size_t)bi * T + bj] = tot;
}

__global__ __launch_bounds__(STAT_THREADS) void mean_distance_final(const double *__restrict__ partial, int T, int n,
                                                                    double *__restrict__ out1) {
    __shared__ double scratch[STAT_THREADS / 64];
    double s = 0.0;
    const int total = T * T;
    for (int t = threadIdx.x; t < total; t += STAT_THREADS) {
        int bi = t / T, bj = t - bi * T;
        if (bj >= bi) s += partial[t];
    }
    double tot = block_sum(s, scratch);
    if (threadIdx.x == 0) out1[0] = tot / (0.5 * (double)n * (double)(n - 1));
}

// ---- PCA axis: centred covariance in one block, 3x3 symmetric eigen-solve by cyclic Jacobi --------
__device__ void jacobi3(double a[3][3], double v[3][3]) {
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) v[i][j] = (i == j) ? 1.0 : 0.0;
    for (int sweep = 0; sweep < 32; ++sweep) {
        double off = a[0][1] * a[0][1] + a[0][2] * a[0][2] + a[1][2] * a[1][2];
        double diag = a[0][0] * a[0][0] + a[1][1] * a[1][1] + a[2][2] * a[2][2];
        if (off <= diag * 1e-40 || off == 0.0) break;
        for (int p = 0; p < 2; ++p)
            for (int q = p + 1; q < 3; ++q) {
                double apq = a[p][q];
                if (apq == 0.0) continue;
                double theta = (a[q][q] - a[p][p]) / (2.0 * apq);
                double t = ((theta >= 0.0) ? 1.0 : -1.0) / (__builtin_fabs(theta) + __builtin_sqrt(theta * theta + 1.0));
                double c = 1.0 / __builtin_sqrt(t * t + 1.0), s = t * c;
                for (int k = 0; k < 3; ++k) {  // A <- A J
                    double akp = a[k][p], akq = a[k][q];
                    a[k][p] = c * akp - s * akq;
                    a[k][q] = s * akp + c * akq;
                }
                for (int k = 0; k < 3; ++k) {  // A <- J^T A
                    double apk = a[p][k], aqk = a[q][k];
                    a[p][k] = c * apk - s * aqk;
                    a[q][k] = s * apk + c * aqk;
                }
                for (int k = 0; k < 3; ++k) {  // V <- V J
                    double vkp = v[k][p], vkq = v[k][q];
                    v[k][p] = c * vkp - s * vkq;
                    v[k][q] = s * vkp + c * vkq;
                }
            }
    }
}

// ncomp = 1: first axis (out[3]); ncomp = 3: all axes by decreasing variance (out[9], row-major)
__global__ __launch_bounds__(STAT_THREADS) void pca_axis_kernel(const double *__restrict__ xyz, int n,
                                                                double *__restrict__ out3, int ncomp) {
    __shared__ double scratch[STAT_THREADS / 64];
    __shared__ double mean[3];
    for (int c = 0; c < 3; ++c) {
        const double *row = xyz + (size_t)c * n;
        double s = 0.0;
        for (int i = threadIdx.x; i < n; i += STAT_THREADS) s += row[i];
        double tot = block_sum(s, scratch);
        if (threadIdx.x == 0) mean[c] = tot / (double)n;
    }
    __syncthreads();
    const double m0 = mean[0], m1 = mean[1], m2 = mean[2];
    double c00 = 0, c01 = 0, c02 = 0, c11 = 0, c12 = 0, c22 = 0;
    for (int i = threadIdx.x; i < n; i += STAT_THREADS) {
        double a = xyz[i] - m0, b = xyz[(size_t)n + i] - m1, c = xyz[2 * (size_t)n + i] - m2;
        c00 += a * a; c01 += a * b; c02 += a * c; c11 += b * b; c12 += b * c; c22 += c * c;
    }
    c00 = block_sum(c00, scratch); c01 = block_sum(c01, scratch); c02 = block_sum(c02, scratch);
    c11 = block_sum(c11, scratch); c12 = block_sum(c12, scratch); c22 = block_sum(c22, scratch);
    if (threadIdx.x == 0) {
        const double inv = 1.0 / (double)(n - 1);
        double a[3][3] = {{c00 * inv, c01 * inv, c02 * inv}, {c01 * inv, c11 * inv, c12 * inv}, {c02 * inv, c12 * inv, c22 * inv}};
        double v[3][3];
        jacobi3(a, v);
        // eigenvalues on the diagonal of a, eigenvectors in the columns of v; order by decreasing eigenvalue
        int order[3] = {0, 1, 2};
        for (int i = 0; i < 2; ++i)
            for (int j = i + 1; j < 3; ++j)
                if (a[order[j]][order[j]] > a[order[i]][order[i]]) { int t = order[i]; order[i] = order[j]; order[j] = t; }
        for (int c = 0; c < ncomp; ++c) {
            const int col = order[c];
            double e0 = v[0][col], e1 = v[1][col], e2 = v[2][col];
            double nrm = __builtin_sqrt((e0 * e0 + e1 * e1) + e2 * e2);
            e0 /= nrm; e1 /= nrm; e2 /= nrm;
            // svd_flip(u_based_decision=False): the entry of largest magnitude is made positive
            double big = e0;
            if (__builtin_fabs(e1) > __builtin_fabs(big)) big = e1;
            if (__builtin_fabs(e2) > __builtin_fabs(big)) big = e2;
            if (big < 0.0) { e0 = -e0; e1 = -e1; e2 = -e2; }
            out3[3 * c] = e0; out3[3 * c + 1] = e1; out3[3 * c + 2] = e2;
        }
    }
}

}  // namespace pm

extern "C" {

size_t pm_centroid_workspace(int) { return 0; }

int pm_centroid(const double *xyz, int n, double *out3, void *, size_t, void *stream) {
    if (!xyz || !out3 || n <= 0) return PM_ERR_INVALID_ARG;
    pm::centroid_kernel<<<1, pm::STAT_THREADS, 0, (hipStream_t)stream>>>(xyz, n, out3);
    return pm::launch_status();
}

size_t pm_mean_distance_workspace(int n) {
    if (n <= 0) return 0;
    size_t T = ((size_t)n + pm::MD_TILE - 1) / pm::MD_TILE;
    return T * T * sizeof(double);
}

int pm_mean_distance(const double *xyz, int n, double *out1, void *ws, size_t ws_bytes, void *stream) {
    if (!xyz || !out1 || n < 2) return PM_ERR_INVALID_ARG;
    if (!ws || ws_bytes < pm_mean_distance_workspace(n)) return PM_ERR_WORKSPACE;
    const int T = (n + pm::MD_TILE - 1) / pm::MD_TILE;
    hipStream_t s = (hipStream_t)stream;
    pm::mean_distance_tiles<<<dim3(T, T), pm::MD_TILE, 0, s>>>(xyz, n, (double *)ws, 0, 1);
    pm::mean_distance_final<<<1, pm::STAT_THREADS, 0, s>>>((const double *)ws, T, n, out1);
    return pm::launch_status();
}

int pm_mean_distance_rows(const double *xyz, int n, int row_offset, int row_stride, double *partials, size_t partial_bytes,
                          void *stream) {
    if (!xyz || !partials || n < 2 || row_offset < 0 || row_stride < 1 || row_offset >= row_stride) return PM_ERR_INVALID_ARG;
    if (partial_bytes < pm_mean_distance_workspace(n)) return PM_ERR_WORKSPACE;
    const int T = (n + pm::MD_TILE - 1) / pm::MD_TILE;
    hipStream_t s = (hipStream_t)stream;
    if (hipMemsetAsync(partials, 0, pm_mean_distance_workspace(n), s) != hipSuccess) return pm::launch_status();
    if (row_offset < T) {
        const int rows = (T - row_offset + row_stride - 1) / row_stride;
        pm::mean_distance_tiles<<<dim3(T, rows), pm::MD_TILE, 0, s>>>(xyz, n, partials, row_offset, row_stride);
    }
    return pm::launch_status();
}

int pm_mean_distance_finish(const double *partials, int n, double *out1, void *stream) {
    if (!partials || !out1 || n < 2) return PM_ERR_INVALID_ARG;
    const int T = (n + pm::MD_TILE - 1) / pm::MD_TILE;
    pm::mean_distance_final<<<1, pm::STAT_THREADS, 0, (hipStream_t)stream>>>(partials, T, n, out1);
    return pm::launch_status();
}

size_t pm_pca_axis_workspace(int) { return 0; }

int pm_pca_axis(const double *xyz, int n, double *out3, void *, size_t, void *stream) {
    if (!xyz || !out3 || n < 2) return PM_ERR_INVALID_ARG;
    pm::pca_axis_kernel<<<1, pm::STAT_THREADS, 0, (hipStream_t)stream>>>(xyz, n, out3, 1);
    return pm::launch_status();
}

int pm_pca_components(const double *xyz, int n, double *out9, void *stream) {
    if (!xyz || !out9 || n < 2) return PM_ERR_INVALID_ARG;
    pm::pca_axis_kernel<<<1, pm::STAT_THREADS, 0, (hipStream_t)stream>>>(xyz, n, out9, 3);
    return pm::launch_status();
}

}  // extern "C"
