// pm_lsap_dev.hip — DEVICE side of the assignment solve: everything that touches the dense N x M cost matrix.
// Reference call sites: scipy.optimize.linear_sum_assignment(U_h) for the eight matrices, _dock_widget.py:604-611.
// The matrices (up to 8 x 3.2 GB at 20 000 nuclei) stay in HBM; the host solves a sparse core (pm_lsap_core.cpp) and these
// two kernels connect it with the dense problem, one streaming pass each (HBM-read bound: 8 B per entry, read once):
//   row_select_kernel   per row, a handful of entries with the smallest cost - v[col]: the core's initial edges (v = 0)
//                       and, with the solver's column duals, the PRICING step (which entries violate dual feasibility);
//   certificate_kernel  the optimality certificate of a finished solve against every entry of the matrix, plus the list of
//                       entries within eps of tight that decides uniqueness (pm_lsap_unique).
#include <algorithm>
#include "pm_common.h"

namespace pm {

constexpr int LS_THREADS = 256;

// One workgroup per row.  Thread t keeps the entry minimising red = cost - v[col] over its columns t, t + 256, ... (first
// one on ties); the 256 thread minima are ranked by (red, col) and the k best written in rank order.  The row's overall
// minimum is always among them (rank 0); the rest are a cheap, deterministic sample of the row's small entries — pricing
// repairs whatever the sample missed, so it need not be the exact k smallest.  Slots beyond the row's width get col = -1.
// (T: the matrix's element type — float64, or float32 for the filter matrices of lsap.FilteredMatrix, which only SELECT entries.)
template <bool HAS_V, typename T = double>
__global__ __launch_bounds__(LS_THREADS) void row_select_kernel(const T *__restrict__ U, int nc, size_t ld,
                                                                const double *__restrict__ v, int k, int32_t *__restrict__ out_col,
                                                                double *__restrict__ out_cost, int32_t *__restrict__ nonfinite) {
    __shared__ double s_red[LS_THREADS];
    __shared__ int s_col[LS_THREADS];
    const int tid = threadIdx.x;
    const T *row = U + (size_t)blockIdx.x * ld;
    double best = INFINITY, best_cost = INFINITY;
    int best_col = 0x7fffffff;
    bool bad = false;
    for (int j = tid; j < nc; j += LS_THREADS) {
        const double c = (double)row[j];
        bad |= !(fabs(c) < INFINITY);                       // NaN or +-inf: the caller takes SciPy's own path for such matrices
        const double red = HAS_V ? c - v[j] : c;
        if (red < best) { best = red; best_cost = c; best_col = j; }
    }
    s_red[tid] = best;
    s_col[tid] = best_col;
    __syncthreads();
    int rank = 0;
    for (int t = 0; t < LS_THREADS; ++t) {
        const double r = s_red[t];                          // same address in every lane: an LDS broadcast
        const int c = s_col[t];
        rank += (r < best || (r == best && (c < best_col || (c == best_col && t < tid)))) ? 1 : 0;   // empty threads tie: thread order
    }
    if (rank < k) {
        out_col[(size_t)blockIdx.x * k + rank] = (best_col == 0x7fffffff) ? -1 : best_col;
        out_cost[(size_t)blockIdx.x * k + rank] = best_cost;
    }
    if (bad) nonfinite[0] = 1;
}

// Bids of a list of rows (the Jacobi form of Jonker & Volgenant's augmenting row reduction, run on the DENSE rows): for
// row i = rows[b]: j1 = the column minimising red = U[i][j] - v[j] (lowest column on ties), u1 = that minimum, u2 = the
// second smallest red of the row (= u1 if two columns tie).  One workgroup per listed row; every thread keeps its two
// smallest over its interleaved columns, then the (min, column, second) triples are merged — the merge is exact: the second
// smallest of a union is the smaller of the two seconds and the larger of the two firsts.
struct Bid {
    double m1, m2;
    int c1;
};
__device__ __forceinline__ Bid bid_merge(const Bid &a, const Bid &b) {
    Bid r;
    const bool a_first = a.m1 < b.m1 || (a.m1 == b.m1 && a.c1 <= b.c1);
    if (a_first) { r.m1 = a.m1; r.c1 = a.c1; r.m2 = fmin(a.m2, b.m1); }
    else { r.m1 = b.m1; r.c1 = b.c1; r.m2 = fmin(b.m2, a.m1); }
    return r;
}

__global__ __launch_bounds__(LS_THREADS) void bid_kernel(const double *__restrict__ U, int nc, size_t ld, const double *__restrict__ v,
                                                         const int32_t *__restrict__ rows, int32_t *__restrict__ out_j1,
                                                         double *__restrict__ out_u1, double *__restrict__ out_u2) {
    __shared__ double s_m1[LS_THREADS / 64], s_m2[LS_THREADS / 64];
    __shared__ int s_c1[LS_THREADS / 64];
    const int tid = threadIdx.x;
    const double *row = U + (size_t)rows[blockIdx.x] * ld;
    Bid b = {INFINITY, INFINITY, 0x7fffffff};
    for (int j = tid; j < nc; j += LS_THREADS) {
        const double red = row[j] - v[j];
        if (red < b.m1) { b.m2 = b.m1; b.m1 = red; b.c1 = j; }
        else if (red < b.m2) b.m2 = red;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        Bid o;
        o.m1 = __shfl_xor(b.m1, off, 64);
        o.m2 = __shfl_xor(b.m2, off, 64);
        o.c1 = __shfl_xor(b.c1, off, 64);
        b = bid_merge(b, o);
    }
    if ((tid & 63) == 0) { s_m1[tid >> 6] = b.m1; s_m2[tid >> 6] = b.m2; s_c1[tid >> 6] = b.c1; }
    __syncthreads();
    if (tid == 0) {
        Bid t = {s_m1[0], s_m2[0], s_c1[0]};
        for (int w = 1; w < LS_THREADS / 64; ++w) {
            const Bid o = {s_m1[w], s_m2[w], s_c1[w]};
            t = bid_merge(t, o);
        }
        out_j1[blockIdx.x] = (t.c1 == 0x7fffffff) ? -1 : t.c1;
        out_u1[blockIdx.x] = t.m1;
        out_u2[blockIdx.x] = t.m2;
    }
}

// Column minima v[j] = min_i U[i][j] (the column reduction that starts a square solve from good duals): thread <-> column,
// rows streamed in slabs so that the launch has enough workgroups; slab minima combined by a second tiny kernel.
constexpr int CM_ROWS = 256;          // rows per slab

template <typename T = double>
__global__ __launch_bounds__(LS_THREADS) void col_min_slab_kernel(const T *__restrict__ U, int nr, int nc, size_t ld,
                                                                  double *__restrict__ slab_min) {
    const int j = blockIdx.x * LS_THREADS + threadIdx.x;
    if (j >= nc) return;
    const int r0 = blockIdx.y * CM_ROWS, r1 = min(nr, r0 + CM_ROWS);
    double best = INFINITY;
    for (int i = r0; i < r1; ++i) {
        const double c = (double)U[(size_t)i * ld + j];
        best = c < best ? c : best;
    }
    slab_min[(size_t)blockIdx.y * nc + j] = best;
}

__global__ __launch_bounds__(LS_THREADS) void col_min_final_kernel(const double *__restrict__ slab_min, int slabs, int nc,
                                                                   double *__restrict__ v) {
    const int j = blockIdx.x * LS_THREADS + threadIdx.x;
    if (j >= nc) return;
    double best = INFINITY;
    for (int s = 0; s < slabs; ++s) {
        const double c = slab_min[(size_t)s * nc + j];
        best = c < best ? c : best;
    }
    v[j] = best;
}

// Optimality certificate of (u, v, col4row) for the nr x nc matrix U — the conditions of LP duality, entry by entry:
//   dual feasibility          (U[i][j] - v[j]) - u[i] >= -delta          for every entry            -> summary[0] counts failures
//   complementary slackness   |(U[i][j] - v[j]) - u[i]| <= delta         for j = col4row[i]          -> summary[2] counts failures
// and the non-matching entries with reduced cost <= eps (the "tight" edges uniqueness is decided on): appended to
// tight[cap][2] (with their reduced costs in tight_red[cap]) in arbitrary order, summary[1] = their number (may exceed
// cap: then the list is incomplete).  A row's tight entries are staged in LDS and appended with ONE global reservation per
// row (not one atomic per entry: ~2.5e9 atomics on one address for a degenerate 50k matrix, and an int32 count that wraps
// negative past 2^31 entries); the count saturates: once it exceeds cap no workgroup adds to it any more, and a row with
// more than CERT_STAGE tight entries reports cap + 1 (list incomplete) instead of a partial list.
// stats[0] = largest |reduced cost| on a matched entry, stats[1] = largest violation (positive number), as float64 bit
// patterns (non-negative doubles order like their bit patterns, so an integer atomic max does it).
constexpr int CERT_STAGE = 1024;      // tight entries of one row staged in LDS (12 KB)

template <typename T = double>
__global__ __launch_bounds__(LS_THREADS) void certificate_kernel(const T *__restrict__ U, int nc, size_t ld,
                                                                 const double *__restrict__ u, const double *__restrict__ v,
                                                                 const int32_t *__restrict__ col4row, double delta, double eps,
                                                                 int32_t *__restrict__ summary, unsigned long long *__restrict__ stats,
                                                                 int32_t *__restrict__ tight, double *__restrict__ tight_red, int cap,
                                                                 double *__restrict__ row_slack, double *__restrict__ row_neg) {
    __shared__ int s_cnt[2];
    __shared__ unsigned long long s_max[2];
    __shared__ int s_nt, s_base;
    __shared__ int32_t s_tj[CERT_STAGE];
    __shared__ double s_tr[CERT_STAGE];
    const int tid = threadIdx.x, i = blockIdx.x;
    if (tid < 2) { s_cnt[tid] = 0; s_max[tid] = 0ull; }
    if (tid == 0) { s_nt = 0; s_base = -1; }
    __syncthreads();
    const T *row = U + (size_t)i * ld;
    const double ui = u[i];
    const int jm = col4row[i];
    int viol = 0, loose = 0;
    double worst = 0.0, slack = 0.0;
    for (int j = tid; j < nc; j += LS_THREADS) {
        const double red = ((double)row[j] - v[j]) - ui;
        if (j == jm) {
            if (!(fabs(red) <= delta)) ++loose;
            slack = fmax(slack, fabs(red));                  // fmax drops NaN; `loose` has counted it
        } else if (!(red >= -delta)) {
            ++viol;
            worst = fmax(worst, -red);
        } else if (red <= eps) {
            worst = fmax(worst, -red);                       // a negative reduced cost inside the tolerance still counts in the bound
            const int at = atomicAdd(&s_nt, 1);              // LDS
            if (at < CERT_STAGE) { s_tj[at] = j; s_tr[at] = red; }
        }
    }
    if (viol) atomicAdd(&s_cnt[0], viol);
    if (loose) atomicAdd(&s_cnt[1], loose);
    if (slack > 0.0) atomicMax(&s_max[0], (unsigned long long)__double_as_longlong(slack));
    if (worst > 0.0) atomicMax(&s_max[1], (unsigned long long)__double_as_longlong(worst));
    __syncthreads();
    if (tid == 0) {
        if (s_cnt[0]) atomicAdd(&summary[0], s_cnt[0]);
        if (s_cnt[1]) atomicAdd(&summary[2], s_cnt[1]);
        if (s_max[0]) atomicMax(&stats[0], s_max[0]);
        if (s_max[1]) atomicMax(&stats[1], s_max[1]);
        // per row: |reduced cost| of its matched entry, and its most negative reduced cost — the host adds them up: any other
        // assignment costs at least (the new entries' reduced costs) - sum(row_slack) - sum(row_neg) more than this one
        if (row_slack) row_slack[i] = __longlong_as_double((long long)s_max[0]);
        if (row_neg) row_neg[i] = __longlong_as_double((long long)s_max[1]);
        const int nt = s_nt;
        if (nt > 0) {
            // saturating count: nothing is added once the list has overflowed (the host only asks "more than cap?"), so the
            // int32 cannot wrap: at most (resident workgroups) x CERT_STAGE is added beyond cap
            const int seen = __hip_atomic_load(&summary[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (nt > CERT_STAGE) {
                if (seen <= cap) atomicMax(&summary[1], cap < 0x7fffffff ? cap + 1 : cap);   // a row beyond the staging area: list incomplete
            } else if (seen <= cap) {
                const int base = atomicAdd(&summary[1], nt);
                if (base >= 0 && base < cap) s_base = base;
            }
        }
    }
    __syncthreads();
    const int base = s_base;
    if (base >= 0) {
        const int nt = s_nt;                                 // <= CERT_STAGE here
        for (int k = tid; k < nt; k += LS_THREADS) {
            const long long at = (long long)base + k;
            if (at < cap) { tight[2 * (size_t)at] = i; tight[2 * (size_t)at + 1] = s_tj[k]; tight_red[at] = s_tr[k]; }
        }
    }
}


// out[r] = U[r][col0 + r]: the entries a block of rows starting at global row col0 contributes to the matrix's diagonal
__global__ __launch_bounds__(256) void diagonal_kernel(const double *__restrict__ U, int n, size_t ld, int col0, double *__restrict__ out) {
    const int r = blockIdx.x * 256 + threadIdx.x;
    if (r < n) out[r] = U[(size_t)r * ld + col0 + r];
}

// dst[j][i] = src[i][j]: 64 x 64 tiles through LDS (pitch 65: the column-wise reads of the tile hit distinct banks), both
// sides coalesced.  The solver wants the short side of a matrix as rows (SciPy transposes likewise); with more moving than
// fixed nuclei that is the transpose of what the cost kernel writes.
constexpr int TR_TILE = 64;
__global__ __launch_bounds__(256) void transpose_kernel(const double *__restrict__ src, int rows, int cols, size_t ld_src,
                                                        double *__restrict__ dst, size_t ld_dst) {
    __shared__ double tile[TR_TILE][TR_TILE + 1];
    const int j0 = blockIdx.x * TR_TILE, i0 = blockIdx.y * TR_TILE;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;          // 64 x 4 threads
    for (int r = ty; r < TR_TILE; r += 4) {
        const int i = i0 + r, j = j0 + tx;
        if (i < rows && j < cols) tile[r][tx] = src[(size_t)i * ld_src + j];
    }
    __syncthreads();
    for (int r = ty; r < TR_TILE; r += 4) {
        const int j = j0 + r, i = i0 + tx;
        if (j < cols && i < rows) dst[(size_t)j * ld_dst + i] = tile[tx][r];
    }
}

}  // namespace pm

extern "C" {

int pm_lsap_row_select(const double *U, int nr, int nc, size_t ld, const double *v, int k, int32_t *out_col, double *out_cost,
                       int32_t *nonfinite1, void *stream) {
    if (!U || !out_col || !out_cost || !nonfinite1 || nr <= 0 || nc <= 0 || ld < (size_t)nc || k <= 0 || k > pm::LS_THREADS)
        return PM_ERR_INVALID_ARG;
    hipStream_t s = (hipStream_t)stream;
    if (hipMemsetAsync(nonfinite1, 0, sizeof(int32_t), s) != hipSuccess) return pm::launch_status();
    if (v) pm::row_select_kernel<true><<<nr, pm::LS_THREADS, 0, s>>>(U, nc, ld, v, k, out_col, out_cost, nonfinite1);
    else pm::row_select_kernel<false><<<nr, pm::LS_THREADS, 0, s>>>(U, nc, ld, nullptr, k, out_col, out_cost, nonfinite1);
    return pm::launch_status();
}

int pm_lsap_bid(const double *U, int nr, int nc, size_t ld, const double *v, const int32_t *rows, int n_rows, int32_t *out_j1,
                double *out_u1, double *out_u2, void *stream) {
    if (!U || !v || !rows || !out_j1 || !out_u1 || !out_u2 || nr <= 0 || nc <= 0 || ld < (size_t)nc || n_rows <= 0)
        return PM_ERR_INVALID_ARG;
    pm::bid_kernel<<<n_rows, pm::LS_THREADS, 0, (hipStream_t)stream>>>(U, nc, ld, v, rows, out_j1, out_u1, out_u2);
    return pm::launch_status();
}

size_t pm_lsap_col_min_workspace(int nr, int nc) {
    return (nr > 0 && nc > 0) ? (size_t)((nr + pm::CM_ROWS - 1) / pm::CM_ROWS) * nc * sizeof(double) : 0;
}

int pm_lsap_col_min(const double *U, int nr, int nc, size_t ld, double *v, void *ws, size_t ws_bytes, void *stream) {
    if (!U || !v || nr <= 0 || nc <= 0 || ld < (size_t)nc) return PM_ERR_INVALID_ARG;
    if (!ws || ws_bytes < pm_lsap_col_min_workspace(nr, nc)) return PM_ERR_WORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    const int slabs = (nr + pm::CM_ROWS - 1) / pm::CM_ROWS, cb = (nc + pm::LS_THREADS - 1) / pm::LS_THREADS;
    pm::col_min_slab_kernel<double><<<dim3(cb, slabs), pm::LS_THREADS, 0, s>>>(U, nr, nc, ld, (double *)ws);
    pm::col_min_final_kernel<<<cb, pm::LS_THREADS, 0, s>>>((const double *)ws, slabs, nc, v);
    return pm::launch_status();
}

int pm_lsap_certificate(const double *U, int nr, int nc, size_t ld, const double *u, const double *v, const int32_t *col4row,
                        double delta, double eps, int32_t *summary4, double *stats2, int32_t *tight, double *tight_red, int cap,
                        double *row_slack, double *row_neg, void *stream) {
    if (!U || !u || !v || !col4row || !summary4 || !stats2 || !tight || !tight_red || nr <= 0 || nc < nr || ld < (size_t)nc || cap <= 0 ||
        !(delta >= 0.0) || !(eps >= 0.0))
        return PM_ERR_INVALID_ARG;
    hipStream_t s = (hipStream_t)stream;
    if (hipMemsetAsync(summary4, 0, 4 * sizeof(int32_t), s) != hipSuccess) return pm::launch_status();
    if (hipMemsetAsync(stats2, 0, 2 * sizeof(double), s) != hipSuccess) return pm::launch_status();
    pm::certificate_kernel<double><<<nr, pm::LS_THREADS, 0, s>>>(U, nc, ld, u, v, col4row, delta, eps, summary4, (unsigned long long *)stats2,
                                                                  tight, tight_red, cap, row_slack, row_neg);
    return pm::launch_status();
}

// The same three passes over a FLOAT32 matrix (a filter matrix: lsap.FilteredMatrix) — reduced costs are formed in float64 from the
// converted entry; out_cost receives the converted entries.
int pm_lsap_row_select_f32(const float *U, int nr, int nc, size_t ld, const double *v, int k, int32_t *out_col, double *out_cost,
                           int32_t *nonfinite1, void *stream) {
    if (!U || !out_col || !out_cost || !nonfinite1 || nr <= 0 || nc <= 0 || ld < (size_t)nc || k <= 0 || k > pm::LS_THREADS)
        return PM_ERR_INVALID_ARG;
    hipStream_t s = (hipStream_t)stream;
    if (hipMemsetAsync(nonfinite1, 0, sizeof(int32_t), s) != hipSuccess) return pm::launch_status();
    if (v) pm::row_select_kernel<true, float><<<nr, pm::LS_THREADS, 0, s>>>(U, nc, ld, v, k, out_col, out_cost, nonfinite1);
    else pm::row_select_kernel<false, float><<<nr, pm::LS_THREADS, 0, s>>>(U, nc, ld, nullptr, k, out_col, out_cost, nonfinite1);
    return pm::launch_status();
}

int pm_lsap_col_min_f32(const float *U, int nr, int nc, size_t ld, double *v, void *ws, size_t ws_bytes, void *stream) {
    if (!U || !v || nr <= 0 || nc <= 0 || ld < (size_t)nc) return PM_ERR_INVALID_ARG;
    if (!ws || ws_bytes < pm_lsap_col_min_workspace(nr, nc)) return PM_ERR_WORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    const int slabs = (nr + pm::CM_ROWS - 1) / pm::CM_ROWS, cb = (nc + pm::LS_THREADS - 1) / pm::LS_THREADS;
    pm::col_min_slab_kernel<float><<<dim3(cb, slabs), pm::LS_THREADS, 0, s>>>(U, nr, nc, ld, (double *)ws);
    pm::col_min_final_kernel<<<cb, pm::LS_THREADS, 0, s>>>((const double *)ws, slabs, nc, v);
    return pm::launch_status();
}

int pm_lsap_certificate_f32(const float *U, int nr, int nc, size_t ld, const double *u, const double *v, const int32_t *col4row,
                            double delta, double eps, int32_t *summary4, double *stats2, int32_t *tight, double *tight_red, int cap,
                            double *row_slack, double *row_neg, void *stream) {
    if (!U || !u || !v || !col4row || !summary4 || !stats2 || !tight || !tight_red || nr <= 0 || nc < nr || ld < (size_t)nc || cap <= 0 ||
        !(delta >= 0.0) || !(eps >= 0.0))
        return PM_ERR_INVALID_ARG;
    hipStream_t s = (hipStream_t)stream;
    if (hipMemsetAsync(summary4, 0, 4 * sizeof(int32_t), s) != hipSuccess) return pm::launch_status();
    if (hipMemsetAsync(stats2, 0, 2 * sizeof(double), s) != hipSuccess) return pm::launch_status();
    pm::certificate_kernel<float><<<nr, pm::LS_THREADS, 0, s>>>(U, nc, ld, u, v, col4row, delta, eps, summary4, (unsigned long long *)stats2,
                                                                 tight, tight_red, cap, row_slack, row_neg);
    return pm::launch_status();
}

int pm_lsap_diagonal(const double *U, int nr, int nc, size_t ld, int col0, double *out, void *stream) {
    if (!U || !out || nr <= 0 || nc <= 0 || ld < (size_t)nc || col0 < 0) return PM_ERR_INVALID_ARG;
    const int n = std::min(nr, nc - col0);
    if (n <= 0) return PM_OK;
    pm::diagonal_kernel<<<(n + 255) / 256, 256, 0, (hipStream_t)stream>>>(U, n, ld, col0, out);
    return pm::launch_status();
}

int pm_transpose_f64(const double *src, int rows, int cols, size_t ld_src, double *dst, size_t ld_dst, void *stream) {
    if (!src || !dst || rows <= 0 || cols <= 0 || ld_src < (size_t)cols || ld_dst < (size_t)rows) return PM_ERR_INVALID_ARG;
    const dim3 grid((cols + pm::TR_TILE - 1) / pm::TR_TILE, (rows + pm::TR_TILE - 1) / pm::TR_TILE);
    if (grid.y > 65535u) return PM_ERR_INVALID_ARG;
    pm::transpose_kernel<<<grid, 256, 0, (hipStream_t)stream>>>(src, rows, cols, ld_src, dst, ld_dst);
    return pm::launch_status();
}

}  // extern "C"
