// pm_icp.hip — ICP correspondence search and the ICP loop.
// Reference: perform_icp (perform_icp.py:7-26).  Per iteration the reference materialises
// scipy.spatial.distance_matrix(moving.T, fixed.T) — sum(|x-y|**2, axis=-1)**0.5, N x M float64 — and
// takes np.argmin(axis=1) (first index on ties).  Here the matrix is never formed: each lane keeps
// one moving point in registers, the fixed cloud streams through LDS as wave-uniform broadcasts, and
// a running (distance, index) pair lives in registers.
//
// Exact argmin over the ROUNDED square roots without taking a square root per pair: candidate j
// (later than every candidate this lane has seen) replaces the current best R only if
// fl(sqrt(s_j)) < R, i.e. s_j < thr with thr = the smallest float64 whose rounded root is R.
// sqrt and thr are recomputed only when a lane's best changes (O(log M) times per lane).
// Squared distances use the reference's operation order ((d0*d0 + d1*d1) + d2*d2, d = fixed - moving),
// one rounding each, so indices match np.argmin bit for bit.
#include "pm_common.h"

namespace pm {

int accumulate(const double *, int, const double *, int, const int32_t *, const double *, double *, double *, hipStream_t);
int update(const double *, const double *, const double *, double *, int, const double *, int, const int32_t *, double *,
           double *, double *, double *, double *, hipStream_t);

constexpr int NN_THREADS = 256;
constexpr int NN_WAVES = 4;
constexpr int NN_CHUNK = 1024;                 // fixed points staged per step
constexpr int NN_SUB = NN_CHUNK / NN_WAVES;    // per wave

__device__ __forceinline__ double next_below(double t) {   // t > 0
    return __longlong_as_double(__double_as_longlong(t) - 1);
}

__global__ __launch_bounds__(NN_THREADS) void nn_kernel(const double *__restrict__ mov, int n,
                                                        const double *__restrict__ fix, int m, int slice_len,
                                                        int32_t *__restrict__ out_idx, double *__restrict__ out_dist) {
    __shared__ double F[3][NN_CHUNK];
    __shared__ double mR[NN_WAVES][64];
    __shared__ int mI[NN_WAVES][64];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i = blockIdx.x * 64 + lane;
    const int ic = min(i, n - 1);
    const double p0 = mov[ic], p1 = mov[(size_t)n + ic], p2 = mov[2 * (size_t)n + ic];
    const int jb = blockIdx.y * slice_len, je = min(m, jb + slice_len);

    double bestR = INFINITY, thr = INFINITY;
    int bestI = 0x7fffffff;
    for (int c0 = jb; c0 < je; c0 += NN_CHUNK) {
        __syncthreads();
        for (int e = tid; e < NN_CHUNK; e += NN_THREADS) {
            const int j = c0 + e;
            const bool ok = j < je;
            F[0][e] = ok ? fix[j] : INFINITY;      // padding: infinitely far, never wins
            F[1][e] = ok ? fix[(size_t)m + j] : 0.0;
            F[2][e] = ok ? fix[2 * (size_t)m + j] : 0.0;
        }
        __syncthreads();
        const int e0 = wave * NN_SUB;
#pragma unroll 4
        for (int e = e0; e < e0 + NN_SUB; ++e) {
            const double d0 = F[0][e] - p0, d1 = F[1][e] - p1, d2 = F[2][e] - p2;
            const double s = (d0 * d0 + d1 * d1) + d2 * d2;
            if (s < thr) {
                const double R = __builtin_sqrt(s);
                bestR = R;
                bestI = c0 + e;
                double t = s;
                for (int it = 0; it < 8 && t > 0.0; ++it) {
                    const double tp = next_below(t);
                    if (__builtin_sqrt(tp) == R) t = tp; else break;
                }
                thr = t;
            }
        }
    }
    // merge the four waves (disjoint index ranges): smaller distance, then smaller index
    mR[wave][lane] = bestR;
    mI[wave][lane] = bestI;
    __syncthreads();
    if (wave == 0 && i < n) {
        double R = mR[0][lane];
        int I = mI[0][lane];
#pragma unroll
        for (int w = 1; w < NN_WAVES; ++w) {
            const double Rw = mR[w][lane];
            const int Iw = mI[w][lane];
            if (Rw < R || (Rw == R && Iw < I)) { R = Rw; I = Iw; }
        }
        out_idx[(size_t)blockIdx.y * n + i] = I;
        out_dist[(size_t)blockIdx.y * n + i] = R;
    }
}

__global__ __launch_bounds__(256) void nn_merge_kernel(const int32_t *__restrict__ pidx, const double *__restrict__ pdist,
                                                       int n, int slices, int32_t *__restrict__ nn, double *__restrict__ dist) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    double R = pdist[i];
    int I = pidx[i];
    for (int s = 1; s < slices; ++s) {
        const double Rs = pdist[(size_t)s * n + i];
        const int Is = pidx[(size_t)s * n + i];
        if (Rs < R || (Rs == R && Is < I)) { R = Rs; I = Is; }
    }
    nn[i] = I;
    if (dist) dist[i] = R;
}

inline int nn_slices(int n, int m) {
    const int row_blocks = (n + 63) / 64;
    int want = (1024 + row_blocks - 1) / row_blocks;            // aim for >= 1024 workgroups (4 per CU)
    int max_slices = (m + NN_CHUNK - 1) / NN_CHUNK;
    int s = want < 1 ? 1 : want;
    if (s > max_slices) s = max_slices;
    if (s > 64) s = 64;
    return s < 1 ? 1 : s;
}

inline size_t nn_ws_bytes(int n, int m) {
    const size_t s = (size_t)nn_slices(n, m);
    return align_up(s * n * sizeof(double), 256) + align_up(s * n * sizeof(int32_t), 256);
}

int nn_search(const double *mov, int n, const double *fix, int m, int32_t *nn, double *dist, void *ws, hipStream_t s) {
    const int slices = nn_slices(n, m);
    int slice_len = (m + slices - 1) / slices;
    slice_len = (slice_len + NN_CHUNK - 1) / NN_CHUNK * NN_CHUNK;
    double *pdist = (double *)ws;
    int32_t *pidx = (int32_t *)((char *)ws + align_up((size_t)slices * n * sizeof(double), 256));
    nn_kernel<<<dim3((n + 63) / 64, slices), NN_THREADS, 0, s>>>(mov, n, fix, m, slice_len, pidx, pdist);
    nn_merge_kernel<<<(n + 255) / 256, 256, 0, s>>>(pidx, pdist, n, slices, nn, dist);
    return launch_status();
}

__global__ void icp_init_kernel(const double *__restrict__ fix, int m, double *__restrict__ origin6, double *__restrict__ A16) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        for (int c = 0; c < 3; ++c) { origin6[c] = fix[(size_t)c * m]; origin6[3 + c] = fix[(size_t)c * m]; }
        for (int k = 0; k < 16; ++k) A16[k] = (k % 5 == 0) ? 1.0 : 0.0;
    }
}

struct IcpWs {
    size_t nn_ws, nn, acc_ws, sums, origin, total;
};

inline IcpWs icp_layout(int n, int m) {
    IcpWs w;
    size_t o = 0;
    w.nn_ws = o; o += align_up(nn_ws_bytes(n, m), 256);
    w.nn = o; o += align_up((size_t)n * sizeof(int32_t), 256);
    w.acc_ws = o; o += align_up(pm_icp_accumulate_workspace(n), 256);
    w.sums = o; o += 256;
    w.origin = o; o += 256;
    w.total = o;
    return w;
}

}  // namespace pm

extern "C" {

size_t pm_icp_nn_workspace(int n, int m) { return (n > 0 && m > 0) ? pm::nn_ws_bytes(n, m) : 0; }

int pm_icp_nn(const double *mov, int n, const double *fix, int m, int32_t *nn, double *dist, void *ws, size_t ws_bytes,
              void *stream) {
    if (!mov || !fix || !nn || n <= 0 || m <= 0) return PM_ERR_INVALID_ARG;
    if (!ws || ws_bytes < pm_icp_nn_workspace(n, m)) return PM_ERR_WORKSPACE;
    return pm::nn_search(mov, n, fix, m, nn, dist, ws, (hipStream_t)stream);
}

size_t pm_icp_workspace(int n, int m) { return (n > 0 && m > 0) ? pm::icp_layout(n, m).total : 0; }

int pm_icp(double *mov, int n, const double *fix, int m, int iters, double *A_icp16, double *residuals, int32_t *nn_all,
           void *ws, size_t ws_bytes, void *stream) {
    if (!mov || !fix || !A_icp16 || n <= 0 || m <= 0 || iters < 0) return PM_ERR_INVALID_ARG;
    if (!ws || ws_bytes < pm_icp_workspace(n, m)) return PM_ERR_WORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    const pm::IcpWs L = pm::icp_layout(n, m);
    char *base = (char *)ws;
    int32_t *nn_buf = (int32_t *)(base + L.nn);
    double *acc_ws = (double *)(base + L.acc_ws);
    double *sums = (double *)(base + L.sums);
    double *origin = (double *)(base + L.origin);
    pm::icp_init_kernel<<<1, 64, 0, s>>>(fix, m, origin, A_icp16);
    for (int it = 0; it < iters; ++it) {
        int32_t *nn = nn_all ? nn_all + (size_t)it * n : nn_buf;
        int rc = pm::nn_search(mov, n, fix, m, nn, nullptr, base + L.nn_ws, s);
        if (rc != PM_OK) return rc;
        rc = pm::accumulate(mov, n, fix, m, nn, origin, sums, acc_ws, s);
        if (rc != PM_OK) return rc;
        rc = pm::update(sums, origin, nullptr, mov, n, fix, m, nn, A_icp16, nullptr, nullptr, residuals ? residuals + it : nullptr,
                        acc_ws, s);
        if (rc != PM_OK) return rc;
    }
    return pm::launch_status();
}

}  // extern "C"
