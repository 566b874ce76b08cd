// pm_transform.hip — affine application and least-squares affine fitting.
// Reference: apply_affine_transform (apply_transform.py:3-17), get_affine_transform
// (find_transform.py:4-17: [fixed;1] . pinv([moving;1])), and the refit/apply/compose steps of
// perform_icp (perform_icp.py:18, 23-25) with get_error (utils/utils.py:77-88).
//
// For a full-rank cloud fixed . pinv(moving) is the least-squares solution of
// [fixed;1] ~ A [moving;1].  It is computed here from 22 sums (count, first moments, second
// moments about a caller-chosen origin) by a 3x3 symmetric solve on centred moments plus a
// translation: O(N) streaming, deterministic reduction order, no N-sized pseudo-inverse.
#include "pm_common.h"
#include "pm_solve.h"

namespace pm {

constexpr int TF_THREADS = 256;
constexpr int TF_PER_THREAD = 2;
constexpr int TF_BLOCK_PTS = TF_THREADS * TF_PER_THREAD;
constexpr int NS = 22;   // accumulated slots (PM_ICP_NSUMS - count - pad)
constexpr int TF_RES_RING = 64;   // iterations of the fused ICP loop whose residual partials are reduced by one launch

__global__ __launch_bounds__(TF_THREADS) void apply_affine_kernel(const double *__restrict__ A, const double *in, int n,
                                                                  double *out) {
    const int i = blockIdx.x * TF_THREADS + threadIdx.x;
    if (i >= n) return;
    const double x = in[i], y = in[(size_t)n + i], z = in[2 * (size_t)n + i];
#pragma unroll
    for (int r = 0; r < 3; ++r) out[(size_t)r * n + i] = ((A[4 * r] * x + A[4 * r + 1] * y) + A[4 * r + 2] * z) + A[4 * r + 3];
}

// Ordered sum over the blocks of one slot of the per-block partials: the same additions in the same order wherever
// it is called from (accumulate_final, or every block of update_kernel in the fused loop).  Loads are issued eight
// at a time so that the chain is one add per partial, not one memory round trip.
// Ordered sum over the blocks of one slot of the per-block partials: t = (((0 + p[0]) + p[1]) + ...), the same additions
// in the same order wherever it is called from.  Serial form (one thread per slot, loads issued eight at a time):
__device__ __forceinline__ double ordered_partial_sum(const double *__restrict__ partial, int nblocks, int stride, int k) {
    double t = 0.0;
    int b = 0;
    for (; b + 8 <= nblocks; b += 8) {
        double v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = partial[(size_t)(b + u) * stride + k];
#pragma unroll
        for (int u = 0; u < 8; ++u) t += v[u];
    }
    for (; b < nblocks; ++b) t += partial[(size_t)b * stride + k];
    return t;
}

// Workgroup form of the same sums: all threads fetch a chunk of partials into LDS in one round trip, then thread k adds
// slot k's values in block order.  Result for slot k in out_s[k] (k < nslots), valid after the call's final barrier.
// stage: TF_STAGE doubles of LDS.
constexpr int TF_STAGE = 2816;     // 128 blocks x 22 slots
__device__ __forceinline__ void ordered_partial_sums_block(const double *__restrict__ partial, int nblocks, int nslots,
                                                           double *__restrict__ stage, double *__restrict__ out_s) {
    const int per = TF_STAGE / nslots;                   // blocks per chunk
    double t = 0.0;
    for (int b0 = 0; b0 < nblocks; b0 += per) {
        const int cnt = min(per, nblocks - b0) * nslots;
        __syncthreads();
        for (int e = threadIdx.x; e < cnt; e += blockDim.x) stage[e] = partial[(size_t)b0 * nslots + e];
        __syncthreads();
        if ((int)threadIdx.x < nslots)
            for (int e = threadIdx.x; e < cnt; e += nslots) t += stage[e];
    }
    if ((int)threadIdx.x < nslots) out_s[threadIdx.x] = t;
    __syncthreads();
}

// per-block partial sums, slots 1..22 of the PM_ICP_NSUMS layout (slot 0, the count, is known).
// Fixed reduction tree: lane butterfly per wave (no barrier), then the block's waves in order.
__global__ __launch_bounds__(TF_THREADS) void accumulate_kernel(const double *__restrict__ mov, int n,
                                                                const double *__restrict__ fix, int m,
                                                                const int32_t *__restrict__ nn,
                                                                const double *__restrict__ origin6,
                                                                double *__restrict__ partial) {
    __shared__ double wsum[TF_THREADS / 64][NS];
    const double om0 = origin6[0], om1 = origin6[1], om2 = origin6[2];
    const double of0 = origin6[3], of1 = origin6[4], of2 = origin6[5];
    double s[NS];
#pragma unroll
    for (int k = 0; k < NS; ++k) s[k] = 0.0;
    const int base = blockIdx.x * TF_BLOCK_PTS;
#pragma unroll
    for (int u = 0; u < TF_PER_THREAD; ++u) {
        const int i = base + u * TF_THREADS + threadIdx.x;
        if (i < n) {
            const int j = nn ? nn[i] : i;
            const double a0 = mov[i] - om0, a1 = mov[(size_t)n + i] - om1, a2 = mov[2 * (size_t)n + i] - om2;
            const double f0 = fix[j] - of0, f1 = fix[(size_t)m + j] - of1, f2 = fix[2 * (size_t)m + j] - of2;
            s[0] += a0; s[1] += a1; s[2] += a2;
            s[3] += f0; s[4] += f1; s[5] += f2;
            s[6] += a0 * a0; s[7] += a0 * a1; s[8] += a0 * a2; s[9] += a1 * a1; s[10] += a1 * a2; s[11] += a2 * a2;
            s[12] += f0 * a0; s[13] += f0 * a1; s[14] += f0 * a2;
            s[15] += f1 * a0; s[16] += f1 * a1; s[17] += f1 * a2;
            s[18] += f2 * a0; s[19] += f2 * a1; s[20] += f2 * a2;
            s[21] += (f0 * f0 + f1 * f1) + f2 * f2;
        }
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < NS; ++k) {
        const double t = wave_sum(s[k]);
        if (lane == 0) wsum[wave][k] = t;
    }
    __syncthreads();
    if (threadIdx.x < NS) {
        double t = 0.0;
#pragma unroll
        for (int w = 0; w < TF_THREADS / 64; ++w) t += wsum[w][threadIdx.x];
        partial[(size_t)blockIdx.x * NS + threadIdx.x] = t;
    }
}

// ordered sum of the block partials -> sums[22]
__global__ __launch_bounds__(64) void accumulate_final(const double *__restrict__ partial, int nblocks, int n,
                                                       double *__restrict__ sums) {
    const int k = threadIdx.x;
    if (k == 0) sums[0] = (double)n;
    if (k < NS) sums[1 + k] = ordered_partial_sum(partial, nblocks, NS, k);
    if (k == NS) sums[1 + NS] = 0.0;
}

__global__ void solve_kernel(const double *__restrict__ sums, const double *__restrict__ origin6, double *__restrict__ A16,
                             int32_t *__restrict__ status) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        double A[16];
        const double ratio = affine_from_sums(sums, origin6, A);
        for (int k = 0; k < 16; ++k) A16[k] = A[k];
        if (status) status[0] = (ratio > PM_DEGENERATE_MOMENTS) ? 0 : 1;
    }
}

// mov <- A_est . mov ; residual partial per block ; block 0 composes A_icp <- A_est . A_icp.
// A_est comes from `A_given`, or is solved from `sums`, or — fused ICP loop — from the per-block partials of
// accumulate_kernel, which every block then sums itself in the fixed order (acc_partial != nullptr).
__global__ __launch_bounds__(TF_THREADS) void update_kernel(const double *__restrict__ sums, const double *__restrict__ acc_partial,
                                                            const double *__restrict__ origin6,
                                                            const double *__restrict__ A_given, double *mov, int n, const double *__restrict__ fix, int m,
                                                            const int32_t *__restrict__ nn, double *A_icp16, double *A_est16,
                                                            double *__restrict__ partial, int32_t *status) {
    __shared__ double scratch[TF_THREADS / 64];
    __shared__ double As[16];
    __shared__ double sums_s[PM_ICP_NSUMS];
    __shared__ double stage[TF_STAGE];
    if (acc_partial) {
        if (threadIdx.x == 0) { sums_s[0] = (double)n; sums_s[1 + NS] = 0.0; }
        ordered_partial_sums_block(acc_partial, gridDim.x, NS, stage, sums_s + 1);
    }
    if (threadIdx.x == 0) {
        double A[16];
        double ratio = 1.0;
        if (A_given) {
            for (int k = 0; k < 16; ++k) A[k] = A_given[k];
        } else {
            ratio = affine_from_sums(acc_partial ? sums_s : sums, origin6, A);   // identical operations in every block -> identical A_est
        }
        for (int k = 0; k < 16; ++k) As[k] = A[k];
        if (blockIdx.x == 0) {
            // a (nearly) planar moving cloud: the normal equations are singular where the reference's pinv is not; sticky flag,
            // the host mirror then reruns the refinement with pinv fits (perform_icp.py:18, find_transform.py:17)
            if (status && !(ratio > PM_DEGENERATE_MOMENTS)) status[0] = 1;
            if (A_est16)
                for (int k = 0; k < 16; ++k) A_est16[k] = A[k];
            if (A_icp16) {                   // perform_icp.py:25, np.matmul(A_est, A_icp)
                double C[16];
                for (int r = 0; r < 4; ++r)
                    for (int c = 0; c < 4; ++c) {
                        double t = 0.0;
                        for (int k = 0; k < 4; ++k) t += A[4 * r + k] * A_icp16[4 * k + c];
                        C[4 * r + c] = t;
                    }
                for (int k = 0; k < 16; ++k) A_icp16[k] = C[k];
            }
        }
    }
    __syncthreads();
    double res = 0.0;
    const int base = blockIdx.x * TF_BLOCK_PTS;
#pragma unroll
    for (int u = 0; u < TF_PER_THREAD; ++u) {
        const int i = base + u * TF_THREADS + threadIdx.x;
        if (i < n) {
            const double x = mov[i], y = mov[(size_t)n + i], z = mov[2 * (size_t)n + i];
            const double q0 = ((As[0] * x + As[1] * y) + As[2] * z) + As[3];
            const double q1 = ((As[4] * x + As[5] * y) + As[6] * z) + As[7];
            const double q2 = ((As[8] * x + As[9] * y) + As[10] * z) + As[11];
            mov[i] = q0; mov[(size_t)n + i] = q1; mov[2 * (size_t)n + i] = q2;
            const int j = nn ? nn[i] : i;
            const double d0 = q0 - fix[j], d1 = q1 - fix[(size_t)m + j], d2 = q2 - fix[2 * (size_t)m + j];
            res += __builtin_sqrt((d0 * d0 + d1 * d1) + d2 * d2);     // get_error: mean of column norms
        }
    }
    double t = block_sum(res, scratch);
    if (threadIdx.x == 0) partial[blockIdx.x] = t;
}

// mean residuals of `count` iterations at once: workgroup r adds row r of partial[count][nblocks] in block order
// (the additions of residual_final) -> mean_out[r].  The fused ICP loop calls it once per TF_RES_RING iterations.
__global__ __launch_bounds__(64) void residual_flush(const double *__restrict__ partial, int nblocks, int n, double *__restrict__ mean_out) {
    if (threadIdx.x == 0) mean_out[blockIdx.x] = ordered_partial_sum(partial + (size_t)blockIdx.x * nblocks, nblocks, 1, 0) / (double)n;
}

__global__ void residual_final(const double *__restrict__ partial, int nblocks, int n, double *__restrict__ parts2,
                               double *__restrict__ mean_out) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        const double t = ordered_partial_sum(partial, nblocks, 1, 0);
        if (parts2) { parts2[0] = t; parts2[1] = (double)n; }
        if (mean_out) mean_out[0] = t / (double)n;
    }
}

// get_error (utils/utils.py:77-88): mean over columns of ||a - b||
__global__ __launch_bounds__(TF_THREADS) void error_kernel(const double *__restrict__ a, const double *__restrict__ b, int n,
                                                           double *__restrict__ partial) {
    __shared__ double scratch[TF_THREADS / 64];
    double res = 0.0;
    const int base = blockIdx.x * TF_BLOCK_PTS;
#pragma unroll
    for (int u = 0; u < TF_PER_THREAD; ++u) {
        const int i = base + u * TF_THREADS + threadIdx.x;
        if (i < n) {
            const double d0 = a[i] - b[i], d1 = a[(size_t)n + i] - b[(size_t)n + i], d2 = a[2 * (size_t)n + i] - b[2 * (size_t)n + i];
            res += __builtin_sqrt((d0 * d0 + d1 * d1) + d2 * d2);
        }
    }
    double t = block_sum(res, scratch);
    if (threadIdx.x == 0) partial[blockIdx.x] = t;
}

int tf_blocks(int n) { return (n + TF_BLOCK_PTS - 1) / TF_BLOCK_PTS; }

// internal entry points shared with pm_icp.hip
int accumulate(const double *mov, int n, const double *fix, int m, const int32_t *nn, const double *origin6,
               double *sums, double *ws, hipStream_t s) {
    const int nb = tf_blocks(n);
    accumulate_kernel<<<nb, TF_THREADS, 0, s>>>(mov, n, fix, m, nn, origin6, ws);
    accumulate_final<<<1, 64, 0, s>>>(ws, nb, n, sums);
    return launch_status();
}

int update(const double *sums, const double *origin6, const double *A_given, double *mov, int n, const double *fix, int m,
           const int32_t *nn, double *A_icp16, double *A_est16, double *parts2, double *mean_out, double *ws, int32_t *status,
           hipStream_t s) {
    const int nb = tf_blocks(n);
    update_kernel<<<nb, TF_THREADS, 0, s>>>(sums, nullptr, origin6, A_given, mov, n, fix, m, nn, A_icp16, A_est16, ws, status);
    residual_final<<<1, 64, 0, s>>>(ws, nb, n, parts2, mean_out);
    return launch_status();
}

// One refit + apply of the fused ICP loop in two launches: per-block moment partials, then a kernel whose every block
// adds those partials in block order, solves and applies.  Same additions in the same order as accumulate() + update(),
// hence the same bits.  The residual partials of iteration `it` go to row it % TF_RES_RING of res_ring; the caller
// turns the rows into mean residuals with residual_rows().  acc_ws: tf_blocks(n) * NS doubles.
int refit_apply(double *mov, int n, const double *fix, int m, const int32_t *nn, const double *origin6, double *A_icp16,
                double *acc_ws, double *res_ring, int it, int32_t *status, hipStream_t s) {
    const int nb = tf_blocks(n);
    accumulate_kernel<<<nb, TF_THREADS, 0, s>>>(mov, n, fix, m, nn, origin6, acc_ws);
    update_kernel<<<nb, TF_THREADS, 0, s>>>(nullptr, acc_ws, origin6, nullptr, mov, n, fix, m, nn, A_icp16, nullptr,
                                            res_ring + (size_t)(it % TF_RES_RING) * nb, status);
    return launch_status();
}

int residual_rows(const double *res_ring, int n, int count, double *mean_out, hipStream_t s) {
    residual_flush<<<count, 64, 0, s>>>(res_ring, tf_blocks(n), n, mean_out);
    return launch_status();
}

size_t residual_ring_bytes(int n) { return (size_t)TF_RES_RING * tf_blocks(n) * sizeof(double); }

__global__ void origin_kernel(const double *__restrict__ mov, int n, const double *__restrict__ fix, int m,
                              double *__restrict__ origin6) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        for (int c = 0; c < 3; ++c) {
            origin6[c] = mov[(size_t)c * n];
            origin6[3 + c] = fix[(size_t)c * m];
        }
    }
}

}  // namespace pm

extern "C" {

int pm_apply_affine(const double *A16, const double *in, int n, double *out, void *stream) {
    if (!A16 || !in || !out || n <= 0) return PM_ERR_INVALID_ARG;
    pm::apply_affine_kernel<<<(n + pm::TF_THREADS - 1) / pm::TF_THREADS, pm::TF_THREADS, 0, (hipStream_t)stream>>>(A16, in, n, out);
    return pm::launch_status();
}

size_t pm_icp_accumulate_workspace(int n) { return n > 0 ? (size_t)pm::tf_blocks(n) * pm::NS * sizeof(double) : 0; }

int pm_icp_accumulate(const double *mov, int n, const double *fix, int m, const int32_t *nn, const double *origin6,
                      double *sums, void *ws, size_t ws_bytes, void *stream) {
    if (!mov || !fix || !origin6 || !sums || n <= 0 || m <= 0) return PM_ERR_INVALID_ARG;
    if (!ws || ws_bytes < pm_icp_accumulate_workspace(n)) return PM_ERR_WORKSPACE;
    return pm::accumulate(mov, n, fix, m, nn, origin6, sums, (double *)ws, (hipStream_t)stream);
}

size_t pm_icp_update_workspace(int n) { return n > 0 ? (size_t)pm::tf_blocks(n) * sizeof(double) : 0; }

int pm_icp_update(const double *sums, const double *origin6, double *mov, int n, const double *fix, int m,
                  const int32_t *nn, double *A_icp16, double *A_est16, double *residual_parts2, int32_t *status1, void *ws,
                  size_t ws_bytes, void *stream) {
    if (!sums || !origin6 || !mov || !fix || n <= 0 || m <= 0) return PM_ERR_INVALID_ARG;
    if (!ws || ws_bytes < pm_icp_update_workspace(n)) return PM_ERR_WORKSPACE;
    return pm::update(sums, origin6, nullptr, mov, n, fix, m, nn, A_icp16, A_est16, residual_parts2, nullptr, (double *)ws,
                      status1, (hipStream_t)stream);
}

int pm_icp_apply(const double *A_est16, double *mov, int n, const double *fix, int m, const int32_t *nn, double *A_icp16,
                 double *residual_parts2, void *ws, size_t ws_bytes, void *stream) {
    if (!A_est16 || !mov || !fix || n <= 0 || m <= 0) return PM_ERR_INVALID_ARG;
    if (!ws || ws_bytes < pm_icp_update_workspace(n)) return PM_ERR_WORKSPACE;
    return pm::update(nullptr, nullptr, A_est16, mov, n, fix, m, nn, A_icp16, nullptr, residual_parts2, nullptr, (double *)ws,
                      nullptr, (hipStream_t)stream);
}

size_t pm_get_error_workspace(int n) { return pm_icp_update_workspace(n); }

int pm_get_error(const double *a, const double *b, int n, double *out1, void *ws, size_t ws_bytes, void *stream) {
    if (!a || !b || !out1 || n <= 0) return PM_ERR_INVALID_ARG;
    if (!ws || ws_bytes < pm_get_error_workspace(n)) return PM_ERR_WORKSPACE;
    const int nb = pm::tf_blocks(n);
    pm::error_kernel<<<nb, pm::TF_THREADS, 0, (hipStream_t)stream>>>(a, b, n, (double *)ws);
    pm::residual_final<<<1, 64, 0, (hipStream_t)stream>>>((const double *)ws, nb, n, nullptr, out1);
    return pm::launch_status();
}

size_t pm_fit_affine_workspace(int n) {
    return n > 0 ? pm::align_up(pm_icp_accumulate_workspace(n), 256) + 256 + 256 : 0;  // partials | sums[22] | origin[6]
}

int pm_fit_affine(const double *mov, int n, const double *fix, int n_fix, const int32_t *nn, double *A_out16, int32_t *status1,
                  void *ws, size_t ws_bytes, void *stream) {
    if (!mov || !fix || !A_out16 || n <= 0 || n_fix <= 0) return PM_ERR_INVALID_ARG;
    if (!nn && n_fix < n) return PM_ERR_INVALID_ARG;
    if (!ws || ws_bytes < pm_fit_affine_workspace(n)) return PM_ERR_WORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    char *base = (char *)ws;
    double *partial = (double *)base;
    double *sums = (double *)(base + pm::align_up(pm_icp_accumulate_workspace(n), 256));
    double *origin = sums + 32;
    pm::origin_kernel<<<1, 64, 0, s>>>(mov, n, fix, n_fix, origin);
    int rc = pm::accumulate(mov, n, fix, n_fix, nn, origin, sums, partial, s);
    if (rc != PM_OK) return rc;
    pm::solve_kernel<<<1, 64, 0, s>>>(sums, origin, A_out16, status1);
    return pm::launch_status();
}

}  // extern "C"
