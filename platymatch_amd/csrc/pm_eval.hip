// pm_eval.hip — rows "next" of SURVEY.md §8f: pairwise Euclidean distances for the evaluation metrics and
// per-label moments of a label image.
// Reference: EvaluateMetrics._calculate_metrics (_dock_widget.py:1030-1080: scipy cdist + linear_sum_assignment)
// and the label-image branch of EstimateTransform._click_run (_dock_widget.py:497-521: per label np.where + np.mean).
#include "pm_common.h"

namespace pm {

// ---- cdist: 8 bytes written per pair, nothing else to do -> HBM-write bound ----------------------------------
// Workgroup tile: 32 rows x 512 columns.  A lane owns two adjacent columns (one 16-byte store per row: the
// store width that reaches the write roofline on gfx950) and keeps their b-points in registers; the 32 a-points of
// the tile are wave-uniform (scalar loads).  scipy's euclidean kernel sums (u_i - v_i)^2 in coordinate order and
// takes one sqrt: ((d0*d0 + d1*d1) + d2*d2), reproduced operation for operation.
constexpr int CD_ROWS = 32;
constexpr int CD_COLS = 512;

__global__ __launch_bounds__(256) void cdist_kernel(const double *__restrict__ a, int n, const double *__restrict__ b, int m,
                                                    double *__restrict__ out, size_t ld) {
    const int j = blockIdx.x * CD_COLS + 2 * threadIdx.x;
    const int i0 = blockIdx.y * CD_ROWS;
    if (j >= m) return;
    const bool two = (j + 1 < m) && ((ld & 1) == 0) && ((((uintptr_t)out) & 15) == 0);   // 16-byte store allowed
    const int j1 = min(j + 1, m - 1);
    const double b0 = b[j], b1 = b[(size_t)m + j], b2 = b[2 * (size_t)m + j];
    const double c0 = b[j1], c1 = b[(size_t)m + j1], c2 = b[2 * (size_t)m + j1];
    const int rows = min(CD_ROWS, n - i0);
    for (int r = 0; r < rows; ++r) {
        const int i = i0 + r;                                  // wave-uniform
        const double p0 = a[i], p1 = a[(size_t)n + i], p2 = a[2 * (size_t)n + i];
        double d0 = p0 - b0, d1 = p1 - b1, d2 = p2 - b2;
        const double u = __builtin_sqrt((d0 * d0 + d1 * d1) + d2 * d2);
        d0 = p0 - c0; d1 = p1 - c1; d2 = p2 - c2;
        const double v = __builtin_sqrt((d0 * d0 + d1 * d1) + d2 * d2);
        double *dst = out + (size_t)i * ld + j;
        if (two) {
            *reinterpret_cast<double2 *>(dst) = make_double2(u, v);
        } else {
            dst[0] = u;
            if (j + 1 < m) dst[1] = v;
        }
    }
}

// ---- label moments -----------------------------------------------------------------------------------------------
// A workgroup owns a brick of LM_TZ x LM_TY rows x LM_TX chunks of LM_RUN consecutive voxels; one thread walks one chunk
// and hands a (count, count*z, count*y, sum x) record per run of equal labels to a small open-addressing table in LDS keyed
// by the label (ds atomics); at the end the table's occupied slots go to the global accumulators — one set of global
// atomics per label and brick instead of one per run (nuclei are blobs: a brick sees a handful of labels, each in dozens of
// runs).  A run that finds no slot within LM_PROBES probes goes to the global accumulators directly, so the table's size
// only affects speed.  All accumulators are 64-bit integers: the sums are exact and do not depend on the order of the atomics.
constexpr int LM_RUN = 16;          // voxels per thread (four 16-byte loads when the row is aligned)
constexpr int LM_TX = 8, LM_TY = 8, LM_TZ = 4;      // chunks along x, rows, slices per workgroup (= 256 threads)
constexpr int LM_SLOTS = 128;
constexpr int LM_PROBES = 8;

struct LabelTable {
    int key[LM_SLOTS];
    unsigned long long acc[LM_SLOTS][4];
};

__device__ __forceinline__ void label_flush(LabelTable &tab, int n_labels, unsigned long long *__restrict__ counts,
                                            unsigned long long *__restrict__ sums, int label, unsigned long long cnt,
                                            unsigned long long sz, unsigned long long sy, unsigned long long sx) {
    unsigned int h = ((unsigned int)label * 2654435761u) >> 25;           // 7 bits
    for (int probe = 0; probe < LM_PROBES; ++probe, h = (h + 1) & (LM_SLOTS - 1)) {
        const int old = atomicCAS(&tab.key[h], 0, label);
        if (old == 0 || old == label) {
            atomicAdd(&tab.acc[h][0], cnt);
            atomicAdd(&tab.acc[h][1], sz);
            atomicAdd(&tab.acc[h][2], sy);
            atomicAdd(&tab.acc[h][3], sx);
            return;
        }
    }
    atomicAdd(&counts[label], cnt);
    atomicAdd(&sums[label], sz);
    atomicAdd(&sums[(size_t)n_labels + label], sy);
    atomicAdd(&sums[2 * (size_t)n_labels + label], sx);
}

__global__ __launch_bounds__(256) void label_moments_kernel(const int32_t *__restrict__ labels, int nz, int ny, int nx,
                                                            int n_labels, unsigned long long *__restrict__ counts,
                                                            unsigned long long *__restrict__ sums, int *__restrict__ bad) {
    __shared__ LabelTable tab;
    const int tid = threadIdx.x;
    for (int k = tid; k < LM_SLOTS; k += 256) {
        tab.key[k] = 0;
        tab.acc[k][0] = tab.acc[k][1] = tab.acc[k][2] = tab.acc[k][3] = 0ull;
    }
    __syncthreads();
    const int chunks = (nx + LM_RUN - 1) / LM_RUN;
    const int bx = (chunks + LM_TX - 1) / LM_TX, by = (ny + LM_TY - 1) / LM_TY;
    const int tile_x = blockIdx.x % bx, tile_y = (blockIdx.x / bx) % by, tile_z = blockIdx.x / (bx * by);
    const int chunk = tile_x * LM_TX + (tid % LM_TX);
    const int y = tile_y * LM_TY + (tid / LM_TX) % LM_TY;
    const int z = tile_z * LM_TZ + tid / (LM_TX * LM_TY);
    if (chunk < chunks && y < ny && z < nz) {
        const int x0 = chunk * LM_RUN, x1 = min(nx, x0 + LM_RUN);
        const int32_t *p = labels + ((long long)z * ny + y) * (long long)nx;
        int lab[LM_RUN];
        if (x1 - x0 == LM_RUN && (((uintptr_t)(p + x0)) & 15) == 0) {
#pragma unroll
            for (int q = 0; q < LM_RUN / 4; ++q) {
                const int4 v = *reinterpret_cast<const int4 *>(p + x0 + 4 * q);
                lab[4 * q] = v.x; lab[4 * q + 1] = v.y; lab[4 * q + 2] = v.z; lab[4 * q + 3] = v.w;
            }
        } else {
#pragma unroll
            for (int q = 0; q < LM_RUN; ++q) lab[q] = (x0 + q < x1) ? p[x0 + q] : 0;
        }
        int cur = 0;
        unsigned long long cnt = 0, sx = 0;
#pragma unroll
        for (int q = 0; q < LM_RUN; ++q) {
            int l = lab[q];
            if (l != cur) {
                if (cur > 0 && cnt) label_flush(tab, n_labels, counts, sums, cur, cnt, cnt * (unsigned long long)z, cnt * (unsigned long long)y, sx);
                cnt = 0; sx = 0;
                if (l < 0 || l >= n_labels) { atomicOr(bad, 1); l = 0; }
                cur = l;
            }
            if (cur > 0) { ++cnt; sx += (unsigned long long)(x0 + q); }
        }
        if (cur > 0 && cnt) label_flush(tab, n_labels, counts, sums, cur, cnt, cnt * (unsigned long long)z, cnt * (unsigned long long)y, sx);
    }
    __syncthreads();
    for (int k = tid; k < LM_SLOTS; k += 256) {
        const int label = tab.key[k];
        if (label > 0) {
            atomicAdd(&counts[label], tab.acc[k][0]);
            atomicAdd(&sums[label], tab.acc[k][1]);
            atomicAdd(&sums[(size_t)n_labels + label], tab.acc[k][2]);
            atomicAdd(&sums[2 * (size_t)n_labels + label], tab.acc[k][3]);
        }
    }
}

// ---- row arg-min of stacked cost matrices --------------------------------------------------------------------------
// np.argmin(U, axis=1): first index of the row minimum; a NaN anywhere in the row wins (NumPy's `mp < min ||
// isnan(mp)` scan stops at the first NaN).  One wave per row, 16-byte loads when the row is 16-byte aligned, a
// butterfly of (value, index) pairs at the end.  8 bytes read per entry, nothing else: HBM-read bound.
__global__ __launch_bounds__(256) void row_argmin_kernel(const double *__restrict__ U, long long total_rows, int rows, int cols,
                                                         size_t ld, size_t matrix_stride, int32_t *__restrict__ idx,
                                                         double *__restrict__ val) {
    const int lane = threadIdx.x & 63;
    const long long r = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= total_rows) return;                               // whole wave leaves together
    const int h = (int)(r / rows), i = (int)(r % rows);
    const double *row = U + (size_t)h * matrix_stride + (size_t)i * ld;
    double best = __builtin_inf();
    int bi = 0x7fffffff, first_nan = 0x7fffffff;
    auto take = [&](double x, int j) {
        if (x != x) first_nan = min(first_nan, j);
        else if (x < best || bi == 0x7fffffff) { best = x; bi = j; }
    };
    if ((((uintptr_t)row) & 15) == 0) {
        const int pairs = cols >> 1;
        const double2 *row2 = reinterpret_cast<const double2 *>(row);
        for (int p = lane; p < pairs; p += 64) {
            const double2 v = row2[p];
            take(v.x, 2 * p);
            take(v.y, 2 * p + 1);
        }
        if ((cols & 1) && lane == 0) take(row[cols - 1], cols - 1);
    } else {
        for (int j = lane; j < cols; j += 64) take(row[j], j);
    }
    for (int off = 32; off; off >>= 1) {
        const double ov = __shfl_xor(best, off);
        const int oi = __shfl_xor(bi, off);
        const int on = __shfl_xor(first_nan, off);
        first_nan = min(first_nan, on);
        if (oi != 0x7fffffff && (bi == 0x7fffffff || ov < best || (ov == best && oi < bi))) { best = ov; bi = oi; }
    }
    if (lane == 0) {
        const bool nan = first_nan != 0x7fffffff;
        idx[r] = nan ? first_nan : bi;
        if (val) val[r] = nan ? __builtin_nan("") : best;
    }
}

}  // namespace pm

extern "C" {

int pm_row_argmin(const double *U, int n_mat, int rows, int cols, size_t ld, size_t matrix_stride, int32_t *idx,
                  double *val, void *stream) {
    if (!U || !idx || n_mat <= 0 || rows <= 0 || cols <= 0 || ld < (size_t)cols) return PM_ERR_INVALID_ARG;
    if (n_mat > 1 && matrix_stride < (size_t)(rows - 1) * ld + (size_t)cols) return PM_ERR_INVALID_ARG;
    const long long total = (long long)n_mat * rows;
    const long long blocks = (total + 3) / 4;
    if (blocks > 0xffffffffLL / 256) return PM_ERR_INVALID_ARG;      // (a launch holds fewer than 2^32 work-items)
    pm::row_argmin_kernel<<<(unsigned int)blocks, 256, 0, (hipStream_t)stream>>>(U, total, rows, cols, ld, matrix_stride, idx, val);
    return pm::launch_status();
}

int pm_cdist(const double *a, int n, const double *b, int m, double *out, size_t ld, void *stream) {
    if (!a || !b || !out || n <= 0 || m <= 0 || ld < (size_t)m) return PM_ERR_INVALID_ARG;
    dim3 grid((m + pm::CD_COLS - 1) / pm::CD_COLS, (n + pm::CD_ROWS - 1) / pm::CD_ROWS);
    if (grid.y > 65535u * 1024u) return PM_ERR_INVALID_ARG;
    pm::cdist_kernel<<<grid, 256, 0, (hipStream_t)stream>>>(a, n, b, m, out, ld);
    return pm::launch_status();
}

int pm_label_moments(const int32_t *labels, int nz, int ny, int nx, int n_labels, unsigned long long *counts,
                     unsigned long long *sums3, void *stream) {
    if (!labels || !counts || !sums3 || nz <= 0 || ny <= 0 || nx <= 0 || n_labels < 2) return PM_ERR_INVALID_ARG;
    hipStream_t s = (hipStream_t)stream;
    // counts[0] doubles as the out-of-range flag word (label 0 is background and never accumulated)
    if (hipMemsetAsync(counts, 0, sizeof(unsigned long long) * (size_t)n_labels, s) != hipSuccess) return pm::launch_status();
    if (hipMemsetAsync(sums3, 0, sizeof(unsigned long long) * 3 * (size_t)n_labels, s) != hipSuccess) return pm::launch_status();
    const long long chunks = (nx + pm::LM_RUN - 1) / pm::LM_RUN;
    const long long blocks = ((chunks + pm::LM_TX - 1) / pm::LM_TX) * ((ny + pm::LM_TY - 1) / pm::LM_TY) * ((nz + pm::LM_TZ - 1) / pm::LM_TZ);
    if (blocks > 0xffffffffLL / 256) return PM_ERR_INVALID_ARG;      // (a launch holds fewer than 2^32 work-items)
    pm::label_moments_kernel<<<(unsigned int)blocks, 256, 0, s>>>(labels, nz, ny, nx, n_labels, counts, sums3, (int *)counts);
    return pm::launch_status();
}

}  // extern "C"
