// pm_icp.hip — ICP correspondence search and the ICP loop.
// Reference: perform_icp (perform_icp.py:7-26).  Per iteration the reference materialises
// scipy.spatial.distance_matrix(moving.T, fixed.T) — sum(|x-y|**2, axis=-1)**0.5, N x M float64 — and
// takes np.argmin(axis=1) (first index on ties).  Here the matrix is never formed: each lane keeps
// two moving points in registers, the fixed cloud streams through the scalar unit as wave-uniform loads,
// and a running (squared distance, index) pair per point lives in registers.
//
// Exact argmin over the ROUNDED square roots without taking a square root per pair: candidate j (later
// than every candidate this lane has seen) replaces the current best only if fl(sqrt(s_j)) < fl(sqrt(S_best));
// that is certain when s_j is below S_best by more than a few ulps and is checked with two square roots in
// the (rare) remaining sliver — see NnBest.
// Squared distances use the reference's operation order ((d0*d0 + d1*d1) + d2*d2, d = fixed - moving),
// one rounding each, so indices match np.argmin bit for bit.
#include <atomic>
#include "pm_common.h"

namespace pm {

// apply + residual with a given 4x4 (pm_transform.hip): the tail of the fused loop
int update(const double *sums, const double *origin6, const double *A_given, double *mov, int n, const double *fix, int m,
           const int32_t *nn, double *A_icp16, double *A_est16, double *parts2, double *mean_out, double *ws, int32_t *status,
           hipStream_t s);
// uniform-grid search (pm_icp_grid.hip): same results as the brute-force kernels below, O(N) instead of O(N*M) per iteration
size_t grid_ws_bytes(int m);
int grid_build(const double *fix, int m, void *ws, hipStream_t s);
int grid_query(const double *mov, int n, int m, const void *ws, int32_t *nn, double *dist, hipStream_t s);
// one whole ICP iteration per launch (pm_icp_grid.hip)
size_t iter_leaf_bytes(int n);
size_t iter_group_bytes(int n);
size_t iter_counter_bytes(int n);
int icp_iteration(bool first, double *mov, int n, const double *fix, int m, const void *grid_ws, const int32_t *nn_prev, int32_t *nn_out,
                  const double *origin6, double *leaf_partial, double *group_partial, unsigned int *counters, double *A_est,
                  double *A_icp, double *res_prev_out, int32_t *status, hipStream_t s);
// iterations first_it .. iters-1 in ONE launch (persistent workgroups; pm_icp_grid.hip)
bool icp_loop_fits(int n);
int icp_loop(double *mov, int n, const double *fix, int m, const void *grid_ws, const int32_t *nn_prev, int32_t *nn_last, int32_t *nn_all,
             const double *origin6, double *leaf_partial, double *group_partial, unsigned int *counters, double *A_est, double *A_icp,
             double *residuals, int32_t *status, int first_it, int iters, hipStream_t s);

constexpr int NN_THREADS = 256;
constexpr int NN_TILE = 128;                   // moving points per wave (two per lane)
constexpr int NN_UNROLL = 4;                   // fixed points per loop trip

__device__ __forceinline__ double next_below(double t) {   // t > 0
    return __longlong_as_double(__double_as_longlong(t) - 1);
}

// fixed cloud 3 x m (SoA) -> packed {x, y, z, 0} records, padded with infinitely distant points up to m_pad,
// so that one wave-uniform 32-byte (scalar) load fetches a whole point and slices need no tail handling
__global__ __launch_bounds__(256) void nn_pack_kernel(const double *__restrict__ fix, int m, int m_pad, double4 *__restrict__ packed) {
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j >= m_pad) return;
    double4 v;
    if (j < m) { v.x = fix[j]; v.y = fix[(size_t)m + j]; v.z = fix[2 * (size_t)m + j]; v.w = 0.0; }
    else { v.x = INFINITY; v.y = 0.0; v.z = 0.0; v.w = 0.0; }
    packed[j] = v;
}

// Running best of one moving point.  S is the squared distance of the current best candidate, I its index,
// lim = S * (1 - 2^-48): a later candidate with s < lim has a strictly smaller ROUNDED root (the real roots
// differ by > 8 ulp), so it wins without any square root being taken; s in [lim, S) is the only case in which
// two different squared distances may round to the same root — decided exactly there, and rare.
struct NnBest {
    double S, lim;
    int I;
};

__device__ __forceinline__ void nn_offer(NnBest &b, double s, int j) {
    if (s < b.S) {
        if (s < b.lim || __builtin_sqrt(s) < __builtin_sqrt(b.S)) {   // np.argmin keeps the FIRST index of equal roots
            b.S = s;
            b.I = j;
            b.lim = s * 0x1.ffffffffffffp-1;                           // 1 - 2^-48
        }
    }
}

// One wave = 128 moving points (lane l owns l and l + 64) against one slice of the fixed cloud.
// The fixed point is the same for every lane: it arrives by wave-uniform loads (scalar unit, SGPR operands),
// so the vector unit only sees 3 sub + 3 mul + 2 add + 1 compare per pair and LDS is not used at all.
__global__ __launch_bounds__(NN_THREADS) void nn_kernel(const double *__restrict__ mov, int n,
                                                        const double4 *__restrict__ fixp, int slice_len,
                                                        int32_t *__restrict__ out_idx, double *__restrict__ out_dist) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int base = (blockIdx.x * 4 + wave) * NN_TILE;
    if (base >= n) return;
    const int i0 = base + lane, i1 = base + 64 + lane;
    const int c0 = min(i0, n - 1), c1 = min(i1, n - 1);
    const double a0 = mov[c0], a1 = mov[(size_t)n + c0], a2 = mov[2 * (size_t)n + c0];
    const double b0 = mov[c1], b1 = mov[(size_t)n + c1], b2 = mov[2 * (size_t)n + c1];
    const int jb = blockIdx.y * slice_len, je = jb + slice_len;

    NnBest A = {INFINITY, INFINITY, 0}, B = {INFINITY, INFINITY, 0};   // index 0 if nothing ever wins (all-NaN row: np.argmin gives 0)
    double4 cur[NN_UNROLL];
#pragma unroll
    for (int u = 0; u < NN_UNROLL; ++u) cur[u] = fixp[jb + u];
    for (int j = jb; j < je; j += NN_UNROLL) {
        // software prefetch: the next trip's scalar loads are in flight while this trip's arithmetic runs
        double4 nxt[NN_UNROLL];
        const int jn = (j + NN_UNROLL < je) ? j + NN_UNROLL : jb;
#pragma unroll
        for (int u = 0; u < NN_UNROLL; ++u) nxt[u] = fixp[jn + u];
        double sa[NN_UNROLL], sb[NN_UNROLL];
        unsigned long long hit = 0;
#pragma unroll
        for (int u = 0; u < NN_UNROLL; ++u) {
            const double4 f = cur[u];
            double d0 = f.x - a0, d1 = f.y - a1, d2 = f.z - a2;
            sa[u] = (d0 * d0 + d1 * d1) + d2 * d2;
            d0 = f.x - b0; d1 = f.y - b1; d2 = f.z - b2;
            sb[u] = (d0 * d0 + d1 * d1) + d2 * d2;
            hit |= __builtin_amdgcn_ballot_w64(sa[u] < A.S) | __builtin_amdgcn_ballot_w64(sb[u] < B.S);
        }
        if (hit) {                                  // wave-uniform branch; candidates are offered in index order
#pragma unroll
            for (int u = 0; u < NN_UNROLL; ++u) {
                nn_offer(A, sa[u], j + u);
                nn_offer(B, sb[u], j + u);
            }
        }
#pragma unroll
        for (int u = 0; u < NN_UNROLL; ++u) cur[u] = nxt[u];
    }
    if (i0 < n) { out_idx[(size_t)blockIdx.y * n + i0] = A.I; out_dist[(size_t)blockIdx.y * n + i0] = __builtin_sqrt(A.S); }
    if (i1 < n) { out_idx[(size_t)blockIdx.y * n + i1] = B.I; out_dist[(size_t)blockIdx.y * n + i1] = __builtin_sqrt(B.S); }
}

// slices cover disjoint, increasing index ranges: smaller distance wins, then smaller index
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

struct NnPlan {
    int slices, slice_len, m_pad;
    size_t off_idx, off_pack, total;
};

inline NnPlan nn_plan(int n, int m) {
    NnPlan p;
    const int tiles = (n + NN_TILE - 1) / NN_TILE;
    int s = 8192 / tiles;                                         // fill 8 waves per SIMD (1024 SIMDs) without spilling into a ninth
    const int max_slices = (m + 63) / 64;                         // at least 64 fixed points per slice
    if (s > max_slices) s = max_slices;
    if (s > 256) s = 256;
    if (s < 1) s = 1;
    int len = (m + s - 1) / s;
    len = (len + NN_UNROLL - 1) / NN_UNROLL * NN_UNROLL;
    s = (m + len - 1) / len;                                      // drop slices made empty by the rounding
    p.slices = s;
    p.slice_len = len;
    p.m_pad = s * len;
    size_t o = align_up((size_t)s * n * sizeof(double), 256);
    p.off_idx = o; o += align_up((size_t)s * n * sizeof(int32_t), 256);
    p.off_pack = o; o += align_up((size_t)p.m_pad * sizeof(double4), 256);
    p.total = o;
    return p;
}

inline size_t nn_ws_bytes(int n, int m) { return nn_plan(n, m).total; }

// `packed_ready`: the packed copy of `fix` in the workspace is still valid (same fix, same n, m) — the ICP loop packs once
int nn_search(const double *mov, int n, const double *fix, int m, int32_t *nn, double *dist, void *ws, hipStream_t s,
              bool packed_ready = false) {
    const NnPlan p = nn_plan(n, m);
    double *pdist = (double *)ws;
    int32_t *pidx = (int32_t *)((char *)ws + p.off_idx);
    double4 *packed = (double4 *)((char *)ws + p.off_pack);
    if (!packed_ready) nn_pack_kernel<<<(p.m_pad + 255) / 256, 256, 0, s>>>(fix, m, p.m_pad, packed);
    nn_kernel<<<dim3((n + 4 * NN_TILE - 1) / (4 * NN_TILE), p.slices), NN_THREADS, 0, s>>>(mov, n, packed, p.slice_len, pidx, pdist);
    nn_merge_kernel<<<(n + 255) / 256, 256, 0, s>>>(pidx, pdist, n, p.slices, nn, dist);
    return launch_status();
}

__global__ void icp_init_kernel(const double *__restrict__ fix, int m, double *__restrict__ origin6, double *__restrict__ A16,
                                int32_t *__restrict__ status) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        if (status) status[0] = 0;
        for (int c = 0; c < 3; ++c) { origin6[c] = fix[(size_t)c * m]; origin6[3 + c] = fix[(size_t)c * m]; }
        for (int k = 0; k < 16; ++k) A16[k] = (k % 5 == 0) ? 1.0 : 0.0;
    }
}

struct IcpWs {
    size_t grid, nn, leaf, group, counters, update_ws, small, total;
};

inline IcpWs icp_layout(int n, int m) {
    IcpWs w;
    size_t o = 0;
    w.grid = o; o += align_up(grid_ws_bytes(m), 256);
    w.nn = o; o += align_up(2 * (size_t)n * sizeof(int32_t), 256);            // matches of this and of the previous iteration
    w.leaf = o; o += align_up(iter_leaf_bytes(n), 256);
    w.group = o; o += align_up(iter_group_bytes(n), 256);
    w.counters = o; o += align_up(iter_counter_bytes(n), 256);
    w.update_ws = o; o += align_up(pm_icp_update_workspace(n), 256);
    w.small = o; o += 512;                                                       // origin[6] | A_est[16]
    w.total = o;
    return w;
}

}  // namespace pm

extern "C" {

size_t pm_icp_nn_brute_workspace(int n, int m) { return (n > 0 && m > 0) ? pm::nn_ws_bytes(n, m) : 0; }

int pm_icp_nn_brute(const double *mov, int n, const double *fix, int m, int32_t *nn, double *dist, void *ws, size_t ws_bytes,
                    void *stream) {
    if (!mov || !fix || !nn || n <= 0 || m <= 0) return PM_ERR_INVALID_ARG;
    if (!ws || ws_bytes < pm_icp_nn_brute_workspace(n, m)) return PM_ERR_WORKSPACE;
    return pm::nn_search(mov, n, fix, m, nn, dist, ws, (hipStream_t)stream);
}

size_t pm_icp_grid_workspace(int m) { return m > 0 ? pm::grid_ws_bytes(m) : 0; }

int pm_icp_grid_build(const double *fix, int m, void *grid, size_t grid_bytes, void *stream) {
    if (!fix || m <= 0) return PM_ERR_INVALID_ARG;
    if (!grid || grid_bytes < pm_icp_grid_workspace(m)) return PM_ERR_WORKSPACE;
    return pm::grid_build(fix, m, grid, (hipStream_t)stream);
}

int pm_icp_grid_nn(const double *mov, int n, int m, const void *grid, size_t grid_bytes, int32_t *nn, double *dist,
                   void *stream) {
    if (!mov || !nn || n <= 0 || m <= 0) return PM_ERR_INVALID_ARG;
    if (!grid || grid_bytes < pm_icp_grid_workspace(m)) return PM_ERR_WORKSPACE;
    return pm::grid_query(mov, n, m, grid, nn, dist, (hipStream_t)stream);
}

size_t pm_icp_nn_workspace(int n, int m) { return (n > 0 && m > 0) ? pm::grid_ws_bytes(m) : 0; }

int pm_icp_nn(const double *mov, int n, const double *fix, int m, int32_t *nn, double *dist, void *ws, size_t ws_bytes,
              void *stream) {
    if (!mov || !fix || !nn || n <= 0 || m <= 0) return PM_ERR_INVALID_ARG;
    if (!ws || ws_bytes < pm_icp_nn_workspace(n, m)) return PM_ERR_WORKSPACE;
    int rc = pm::grid_build(fix, m, ws, (hipStream_t)stream);
    if (rc != PM_OK) return rc;
    return pm::grid_query(mov, n, m, ws, nn, dist, (hipStream_t)stream);
}

size_t pm_icp_workspace(int n, int m) { return (n > 0 && m > 0) ? pm::icp_layout(n, m).total : 0; }

static int icp_run(bool one_launch, double *mov, int n, const double *fix, int m, int iters, double *A_icp16, double *residuals,
                   int32_t *nn_all, int32_t *status1, void *ws, size_t ws_bytes, void *stream) {
    if (!mov || !fix || !A_icp16 || n <= 0 || m <= 0 || iters < 0) return PM_ERR_INVALID_ARG;
    if (!ws || ws_bytes < pm_icp_workspace(n, m)) return PM_ERR_WORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    const pm::IcpWs L = pm::icp_layout(n, m);
    char *base = (char *)ws;
    int32_t *nn_buf = (int32_t *)(base + L.nn);
    double *origin = (double *)(base + L.small), *A_est = origin + 8;
    unsigned int *counters = (unsigned int *)(base + L.counters);
    pm::icp_init_kernel<<<1, 64, 0, s>>>(fix, m, origin, A_icp16, status1);
    if (iters == 0) return pm::launch_status();
    if (hipMemsetAsync(counters, 0, pm::iter_counter_bytes(n), s) != hipSuccess) return pm::launch_status();
    int rc = pm::grid_build(fix, m, base + L.grid, s);             // the fixed cloud never changes: bin it once
    if (rc != PM_OK) return rc;
    const int32_t *nn_prev = nullptr;
    if (one_launch && iters >= 3 && pm::icp_loop_fits(n)) {
        // iteration 0 has nothing to bound its search with (32 lanes per point walk the rings: a launch of its own); iterations
        // 1 .. iters-1 run in ONE launch of persistent workgroups
        int32_t *nn0 = nn_all ? nn_all : nn_buf;
        rc = pm::icp_iteration(true, mov, n, fix, m, base + L.grid, nullptr, nn0, origin, (double *)(base + L.leaf),
                               (double *)(base + L.group), counters, A_est, A_icp16, nullptr, status1, s);
        if (rc != PM_OK) return rc;
        rc = pm::icp_loop(mov, n, fix, m, base + L.grid, nn0, nn_buf + n, nn_all, origin, (double *)(base + L.leaf),
                          (double *)(base + L.group), counters, A_est, A_icp16, residuals, status1, 1, iters, s);
        if (rc != PM_OK) return rc;
        nn_prev = nn_buf + n;
    } else {
        // ONE launch per iteration: iteration `it` applies the transform fitted by iteration it-1 on the way in
        for (int it = 0; it < iters; ++it) {
            int32_t *nn = nn_all ? nn_all + (size_t)it * n : nn_buf + (size_t)(it & 1) * n;
            rc = pm::icp_iteration(it == 0, mov, n, fix, m, base + L.grid, nn_prev, nn, origin, (double *)(base + L.leaf),
                                   (double *)(base + L.group), counters, A_est, A_icp16, (residuals && it > 0) ? residuals + (it - 1) : nullptr,
                                   status1, s);
            if (rc != PM_OK) return rc;
            nn_prev = nn;
        }
    }
    // the last fitted transform still has to be applied (perform_icp.py:23) and its residual taken (:24)
    return pm::update(nullptr, nullptr, A_est, mov, n, fix, m, nn_prev, nullptr, nullptr, nullptr,
                      residuals ? residuals + (iters - 1) : nullptr, (double *)(base + L.update_ws), nullptr, s);
}

int pm_icp(double *mov, int n, const double *fix, int m, int iters, double *A_icp16, double *residuals, int32_t *nn_all,
           int32_t *status1, void *ws, size_t ws_bytes, void *stream) {
    return icp_run(false, mov, n, fix, m, iters, A_icp16, residuals, nn_all, status1, ws, ws_bytes, stream);
}

// One persistent grid in flight per device (two half-resident ones can starve each other): a flag per device, taken here and
// handed back by a host function the stream runs once everything this call enqueued has finished.  A second call on the same
// device while the flag is held is refused with PM_ERR_UNSUPPORTED (the caller takes pm_icp, or waits) — the library's own
// enforcement of the contract in the header; the Python mirror's per-device lock merely avoids ever seeing the refusal.
static std::atomic<int> g_one_launch_busy[64];
static void one_launch_done(void *flag) { ((std::atomic<int> *)flag)->store(0, std::memory_order_release); }

int pm_icp_one_launch(double *mov, int n, const double *fix, int m, int iters, double *A_icp16, double *residuals, int32_t *nn_all,
                      int32_t *status1, void *ws, size_t ws_bytes, void *stream) {
    if (!mov || !fix || !A_icp16 || n <= 0 || m <= 0 || iters < 0) return PM_ERR_INVALID_ARG;
    if (iters < 3 || !pm::icp_loop_fits(n))               // the launch-per-iteration path: nothing persistent, nothing to reserve
        return icp_run(false, mov, n, fix, m, iters, A_icp16, residuals, nn_all, status1, ws, ws_bytes, stream);
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return PM_ERR_INVALID_ARG;
    std::atomic<int> &busy = g_one_launch_busy[dev];
    int expected = 0;
    if (!busy.compare_exchange_strong(expected, 1, std::memory_order_acquire)) return PM_ERR_UNSUPPORTED;
    const int rc = icp_run(true, mov, n, fix, m, iters, A_icp16, residuals, nn_all, status1, ws, ws_bytes, stream);
    if (hipLaunchHostFunc((hipStream_t)stream, one_launch_done, &busy) != hipSuccess) {
        (void)hipGetLastError();
        (void)hipStreamSynchronize((hipStream_t)stream);   // no callback: drain here, then hand the flag back
        busy.store(0, std::memory_order_release);
    }
    return rc;
}

}  // extern "C"
