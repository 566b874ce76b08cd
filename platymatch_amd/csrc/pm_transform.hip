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
constexpr int TF_BLOCK_PTS = PM_TREE_POINTS;   // one workgroup of the reduction kernels = one group of the tree (512 points)
constexpr int TF_RED_THREADS = PM_TREE_POINTS; // ... one thread per point
constexpr int NS = PM_NMOMENTS;                // accumulated slots (PM_ICP_NSUMS - count - pad)

__global__ __launch_bounds__(TF_THREADS) void apply_affine_kernel(const double *__restrict__ A, const double *in, int n,
                                                                  double *out) {
    const int i = blockIdx.x * TF_THREADS + threadIdx.x;
    if (i >= n) return;
    const double x = in[i], y = in[(size_t)n + i], z = in[2 * (size_t)n + i];
    // np.matmul's arithmetic (pm_solve.h: affine_row) — found necessary by tests/probes/soak_parity.py: on lattice-like data
    // (voxel coordinates) a moved point can sit exactly midway between two fixed points, and the last bit of the cloud ICP
    // starts from decides its first correspondence
#pragma unroll
    for (int r = 0; r < 3; ++r) out[(size_t)r * n + i] = affine_row(A + 4 * r, x, y, z);
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

// Group partial sums (slots 1..22 of the PM_ICP_NSUMS layout; slot 0, the count, is known): one workgroup = one group of
// the reduction tree (pm_solve.h), one thread per point.  Terms go to LDS, half of the slots at a time; leaves are added
// serially (8 points), then the 64 leaves serially.
__global__ __launch_bounds__(TF_RED_THREADS) void accumulate_kernel(const double *__restrict__ mov, int n,
                                                                    const double *__restrict__ fix, int m,
                                                                    const int32_t *__restrict__ nn,
                                                                    const double *__restrict__ origin6,
                                                                    double *__restrict__ partial) {
    constexpr int HALF = NS / 2;
    __shared__ double term[HALF][TF_RED_THREADS];
    __shared__ double leaf[HALF][PM_TREE_GROUP];
    const int tid = threadIdx.x;
    const int i = blockIdx.x * TF_BLOCK_PTS + tid;
    double s[NS];
#pragma unroll
    for (int k = 0; k < NS; ++k) s[k] = 0.0;
    if (i < n) {
        const int j = nn ? nn[i] : i;
        moment_terms(mov[i] - origin6[0], mov[(size_t)n + i] - origin6[1], mov[2 * (size_t)n + i] - origin6[2],
                     fix[j] - origin6[3], fix[(size_t)m + j] - origin6[4], fix[2 * (size_t)m + j] - origin6[5], s);
    }
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        __syncthreads();
#pragma unroll
        for (int k = 0; k < HALF; ++k) term[k][tid] = s[half * HALF + k];
        __syncthreads();
        for (int item = tid; item < HALF * PM_TREE_GROUP; item += TF_RED_THREADS) {
            const int k = item / PM_TREE_GROUP, b = item % PM_TREE_GROUP;
            double acc = 0.0;
#pragma unroll
            for (int q = 0; q < PM_TREE_LEAF; ++q) acc += term[k][PM_TREE_LEAF * b + q];
            leaf[k][b] = acc;
        }
        __syncthreads();
        if (tid < HALF) {
            double acc = 0.0;
            for (int b = 0; b < PM_TREE_GROUP; ++b) acc += leaf[tid][b];
            partial[(size_t)blockIdx.x * NS + half * HALF + tid] = acc;
        }
    }
}

// The residual's group partial by the same tree: r = this thread's point's term (0.0 past the end); valid in thread 0.
__device__ __forceinline__ double tree_group_sum(double r, double *__restrict__ term512, double *__restrict__ leaf64) {
    const int tid = threadIdx.x;
    __syncthreads();
    term512[tid] = r;
    __syncthreads();
    if (tid < PM_TREE_GROUP) {
        double acc = 0.0;
#pragma unroll
        for (int q = 0; q < PM_TREE_LEAF; ++q) acc += term512[PM_TREE_LEAF * tid + q];
        leaf64[tid] = acc;
    }
    __syncthreads();
    double acc = 0.0;
    if (tid == 0)
        for (int b = 0; b < PM_TREE_GROUP; ++b) acc += leaf64[b];
    return acc;
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

// mov <- A_est . mov (rows 0-2 of A_est); residual partial per group of the tree; workgroup 0 composes A_icp <- A_est . A_icp.
// A_est comes from `A_given`, or is solved from `sums` (every workgroup solves the same 24 numbers: identical A_est).
__global__ __launch_bounds__(TF_RED_THREADS) void update_kernel(const double *__restrict__ sums, const double *__restrict__ origin6,
                                                                const double *__restrict__ A_given, double *mov, int n,
                                                                const double *__restrict__ fix, int m, const int32_t *__restrict__ nn,
                                                                double *A_icp16, double *A_est16, double *__restrict__ partial,
                                                                int32_t *status) {
    __shared__ double As[16];
    __shared__ double term[TF_RED_THREADS];
    __shared__ double leaf[PM_TREE_GROUP];
    if (threadIdx.x == 0) {
        double A[16];
        double ratio = 1.0;
        if (A_given) {
            for (int k = 0; k < 16; ++k) A[k] = A_given[k];
        } else {
            ratio = affine_from_sums(sums, origin6, A);
        }
        for (int k = 0; k < 16; ++k) As[k] = A[k];
        if (blockIdx.x == 0) {
            // a (nearly) planar moving cloud: the normal equations are singular where the reference's pinv is not; sticky flag,
            // the host mirror then reruns the refinement with pinv fits (perform_icp.py:18, find_transform.py:17)
            if (status && !(ratio > PM_DEGENERATE_MOMENTS)) status[0] = 1;
            if (A_est16)
                for (int k = 0; k < 16; ++k) A_est16[k] = A[k];
            if (A_icp16) compose_affine(A, A_icp16);                      // perform_icp.py:25
        }
    }
    __syncthreads();
    double res = 0.0;
    const int i = blockIdx.x * TF_BLOCK_PTS + threadIdx.x;
    if (i < n) {
        const double x = mov[i], y = mov[(size_t)n + i], z = mov[2 * (size_t)n + i];
        const double q0 = affine_row(As, x, y, z), q1 = affine_row(As + 4, x, y, z), q2 = affine_row(As + 8, x, y, z);
        mov[i] = q0; mov[(size_t)n + i] = q1; mov[2 * (size_t)n + i] = q2;
        const int j = nn ? nn[i] : i;
        const double d0 = q0 - fix[j], d1 = q1 - fix[(size_t)m + j], d2 = q2 - fix[2 * (size_t)m + j];
        res = __builtin_sqrt((d0 * d0 + d1 * d1) + d2 * d2);          // get_error: mean of column norms
    }
    const double t = tree_group_sum(res, term, leaf);
    if (threadIdx.x == 0) partial[blockIdx.x] = t;
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
__global__ __launch_bounds__(TF_RED_THREADS) void error_kernel(const double *__restrict__ a, const double *__restrict__ b, int n,
                                                               double *__restrict__ partial) {
    __shared__ double term[TF_RED_THREADS];
    __shared__ double leaf[PM_TREE_GROUP];
    double res = 0.0;
    const int i = blockIdx.x * TF_BLOCK_PTS + threadIdx.x;
    if (i < n) {
        const double d0 = a[i] - b[i], d1 = a[(size_t)n + i] - b[(size_t)n + i], d2 = a[2 * (size_t)n + i] - b[2 * (size_t)n + i];
        res = __builtin_sqrt((d0 * d0 + d1 * d1) + d2 * d2);
    }
    const double t = tree_group_sum(res, term, leaf);
    if (threadIdx.x == 0) partial[blockIdx.x] = t;
}

int tf_blocks(int n) { return (n + TF_BLOCK_PTS - 1) / TF_BLOCK_PTS; }

// internal entry points shared with pm_icp.hip
int accumulate(const double *mov, int n, const double *fix, int m, const int32_t *nn, const double *origin6,
               double *sums, double *ws, hipStream_t s) {
    const int nb = tf_blocks(n);
    accumulate_kernel<<<nb, TF_RED_THREADS, 0, s>>>(mov, n, fix, m, nn, origin6, ws);
    accumulate_final<<<1, 64, 0, s>>>(ws, nb, n, sums);
    return launch_status();
}

int update(const double *sums, const double *origin6, const double *A_given, double *mov, int n, const double *fix, int m,
           const int32_t *nn, double *A_icp16, double *A_est16, double *parts2, double *mean_out, double *ws, int32_t *status,
           hipStream_t s) {
    const int nb = tf_blocks(n);
    update_kernel<<<nb, TF_RED_THREADS, 0, s>>>(sums, origin6, A_given, mov, n, fix, m, nn, A_icp16, A_est16, ws, status);
    residual_final<<<1, 64, 0, s>>>(ws, nb, n, parts2, mean_out);
    return launch_status();
}

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
    pm::error_kernel<<<nb, pm::TF_RED_THREADS, 0, (hipStream_t)stream>>>(a, b, n, (double *)ws);
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
