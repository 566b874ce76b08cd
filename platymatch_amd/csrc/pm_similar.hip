// pm_similar.hip — the O(N) part of get_similar_transform (reference find_transform.py:21-99) and of the Similar-mode ICP
// iteration (perform_icp.py:14-25) on the device, in NumPy's own arithmetic.
//
// The reference's fit takes ROW 0 of np.linalg.eig's eigenvector matrix as its quaternion (find_transform.py:60-66), so its
// result hangs on LAPACK's eigenvector signs, and those flip when the 4 x 4 matrix N changes in the last bit.
// The 4 x 4 eigen-decomposition therefore stays on the host, in NumPy, on bit-identical input — and everything that is O(N)
// is computed here with the operations NumPy performs, in NumPy's order, so that the seventeen numbers handed to the host are
// the reference's bits:
//   com_source    np.mean(moving, 1)      rows of a C-ordered 3 x N array: chunked pairwise sums (pm_pairwise.h) / N
//   com_target    np.mean(fixed, 1)       in ICP `fixed` is fixed[:, nn] — Fortran-ordered: the columns are added one after
//                                         the other (a serial chain per row) / N
//   Sxx .. Szz    np.sum(Yx * Px) ...     products of the centred coordinates, chunked pairwise sums            (:43-53)
//   D, Sp         sum of Y[:, i] . Y[:, i] one point after the other; the 3-vector dot product is BLAS ddot, whose x86 kernels
//                                         accumulate with fused multiply-adds: fma(z, z, fma(y, y, x * x))       (:86-91)
// and, for the ICP loop, the application of the fitted 4 x 4 — np.matmul(A, [moving; 1]) is BLAS dgemm, a chain of fused
// multiply-adds over k = 0 .. 3 starting from zero (apply_transform.py:14-17) — and the residual np.mean(np.linalg.norm(moved -
// matched, axis=0)) (utils.py:77-88).  The fused forms are what OpenBLAS's x86-64 kernels compute (checked against NumPy on the
// host of the build container and of the GPU box by the tests); on a CPU whose BLAS rounds differently the reference itself
// gives other bits there.
#include "pm_common.h"
#include "pm_pairwise.h"

namespace pm {

constexpr int SM_THREADS = 256;
constexpr int SM_CHAIN_CHUNK = 1024;

struct SimPlan {             // written by sim_plan_kernel
    int leaves, chunks, bad, pad;
};

struct SimWs {
    SimPlan *plan;
    int *off, *chunk_first;
    double *leafsum;         // [12][leaf_cap]
    double *norms;           // [n]
    double *scratch;         // [32]
    int leaf_cap, chunk_cap;
};

inline size_t sim_layout(int n, char *base, SimWs *w) {
    size_t o = 0;
    auto take = [&](size_t bytes) { size_t at = o; o = align_up(o + bytes, 256); return at; };
    const int leaf_cap = pm_pw_leaf_cap(n), chunk_cap = pm_pw_chunk_cap(n);
    const size_t o_plan = take(sizeof(SimPlan));
    const size_t o_off = take(sizeof(int) * ((size_t)leaf_cap + 1));
    const size_t o_cf = take(sizeof(int) * ((size_t)chunk_cap + 1));
    const size_t o_leaf = take(sizeof(double) * 12 * (size_t)leaf_cap);
    const size_t o_norm = take(sizeof(double) * (size_t)n);
    const size_t o_scr = take(sizeof(double) * 32);
    if (w) {
        w->plan = (SimPlan *)(base + o_plan);
        w->off = (int *)(base + o_off);
        w->chunk_first = (int *)(base + o_cf);
        w->leafsum = (double *)(base + o_leaf);
        w->norms = (double *)(base + o_norm);
        w->scratch = (double *)(base + o_scr);
        w->leaf_cap = leaf_cap;
        w->chunk_cap = chunk_cap;
    }
    return o;
}

__global__ void sim_plan_kernel(int n, int *off, int leaf_cap, int *chunk_first, int chunk_cap, SimPlan *plan) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    const int leaves = pm_pw_plan(n, off, leaf_cap, chunk_first, chunk_cap);
    plan->leaves = leaves < 0 ? 0 : leaves;
    plan->chunks = (n + PM_PW_CHUNK - 1) / PM_PW_CHUNK;
    plan->bad = leaves < 0;
    plan->pad = 0;
}

// The element values a leaf sum runs over.  SET 0: the three rows of `mov` (com_source); SET 1: the nine products of the centred
// coordinates; SET 2: one vector (the residual norms); SET 3: the three rows of fixed[:, nn] (com_target of a C-ordered
// `fixed`, i.e. get_similar_transform called directly rather than from the ICP loop).
template <int SET> struct SetSize { static constexpr int Q = SET == 0 ? 3 : SET == 1 ? 9 : SET == 2 ? 1 : 3; };

struct SimSrc {
    const double *mov; int n;
    const double *fix; int m;
    const int32_t *nn;
    const double *cs, *ct;       // centroids (SET 1)
    const double *vec;           // SET 2
};

template <int SET>
__device__ __forceinline__ void sim_values(const SimSrc &s, int i, double v[SetSize<SET>::Q]) {
    if (SET == 0) {
        v[0] = s.mov[i]; v[1] = s.mov[(size_t)s.n + i]; v[2] = s.mov[2 * (size_t)s.n + i];
    } else if (SET == 3) {
        const int j = s.nn ? s.nn[i] : i;
        v[0] = s.fix[j]; v[1] = s.fix[(size_t)s.m + j]; v[2] = s.fix[2 * (size_t)s.m + j];
    } else if (SET == 2) {
        v[0] = s.vec[i];
    } else {
        const int j = s.nn ? s.nn[i] : i;
        const double Px = s.mov[i] - s.cs[0], Py = s.mov[(size_t)s.n + i] - s.cs[1], Pz = s.mov[2 * (size_t)s.n + i] - s.cs[2];      // :32
        const double Yx = s.fix[j] - s.ct[0], Yy = s.fix[(size_t)s.m + j] - s.ct[1], Yz = s.fix[2 * (size_t)s.m + j] - s.ct[2];   // :31
        v[0] = Yx * Px; v[1] = Px * Yy; v[2] = Px * Yz;        // Sxx, Sxy, Sxz (:43-45)
        v[3] = Py * Yx; v[4] = Py * Yy; v[5] = Py * Yz;        // Syx, Syy, Syz
        v[6] = Pz * Yx; v[7] = Pz * Yy; v[8] = Pz * Yz;        // Szx, Szy, Szz
    }
}

// Eight lanes per leaf: lane j keeps NumPy's partial sum r[j] (elements j, 8 + j, 16 + j, ... of the leaf), lane 0 combines them
// as ((r0 + r1) + (r2 + r3)) + ((r4 + r5) + (r6 + r7)) and adds the trailing len % 8 elements; a leaf shorter than 8 is added up
// one element after the other by lane 0.  leafsum[q][k] = sum of quantity q over leaf k.
template <int SET>
__global__ __launch_bounds__(SM_THREADS) void sim_leaf_kernel(const SimSrc s, const int *__restrict__ off, const SimPlan *__restrict__ plan,
                                                              double *__restrict__ leafsum, int leaf_cap) {
    constexpr int Q = SetSize<SET>::Q;
    const int tid = blockIdx.x * SM_THREADS + threadIdx.x;
    const int k = tid >> 3, j = tid & 7;
    const int leaves = plan->leaves;
    if (leaves <= 0) return;                             // plan overflow (plan->bad; sim_combine reports NaN): uniform exit, nothing to read
    const int kc = min(k, leaves - 1);                   // surplus groups shadow the last leaf (no divergent exit before shuffles)
    const int o = off[kc], len = off[kc + 1] - o;
    double r[Q];
#pragma unroll
    for (int q = 0; q < Q; ++q) r[q] = 0.0;
    double res[Q];
    if (len < 8) {
#pragma unroll
        for (int q = 0; q < Q; ++q) res[q] = 0.0;
        if (j == 0)
            for (int i = 0; i < len; ++i) {
                double v[Q];
                sim_values<SET>(s, o + i, v);
#pragma unroll
                for (int q = 0; q < Q; ++q) res[q] += v[q];
            }
    } else {
        const int top = len - (len % 8);
        {
            double v[Q];
            sim_values<SET>(s, o + j, v);
#pragma unroll
            for (int q = 0; q < Q; ++q) r[q] = v[q];
        }
        for (int i = 8; i < top; i += 8) {
            double v[Q];
            sim_values<SET>(s, o + i + j, v);
#pragma unroll
            for (int q = 0; q < Q; ++q) r[q] += v[q];
        }
#pragma unroll
        for (int q = 0; q < Q; ++q) {
            const int base = (threadIdx.x & 63) & ~7;
            const double r0 = __shfl(r[q], base + 0, 64), r1 = __shfl(r[q], base + 1, 64), r2 = __shfl(r[q], base + 2, 64),
                         r3 = __shfl(r[q], base + 3, 64), r4 = __shfl(r[q], base + 4, 64), r5 = __shfl(r[q], base + 5, 64),
                         r6 = __shfl(r[q], base + 6, 64), r7 = __shfl(r[q], base + 7, 64);
            res[q] = ((r0 + r1) + (r2 + r3)) + ((r4 + r5) + (r6 + r7));
        }
        if (j == 0)
            for (int i = top; i < len; ++i) {
                double v[Q];
                sim_values<SET>(s, o + i, v);
#pragma unroll
                for (int q = 0; q < Q; ++q) res[q] += v[q];
            }
    }
    if (j == 0 && k < leaves) {
#pragma unroll
        for (int q = 0; q < Q; ++q) leafsum[(size_t)q * leaf_cap + k] = res[q];
    }
}

// lane q adds up quantity q's leaf sums in NumPy's order; out[q] = total (* scale: 1, or 1 / n for a mean — a true division)
__global__ void sim_combine_kernel(const double *__restrict__ leafsum, int leaf_cap, const int *__restrict__ off,
                                   const int *__restrict__ chunk_first, const SimPlan *__restrict__ plan, int n, int Q, int divide_by_n,
                                   double *__restrict__ out) {
    const int q = threadIdx.x;
    if (q >= Q) return;
    if (plan->bad) { out[q] = __builtin_nan(""); return; }     // the leaf plan did not fit its capacities: say so loudly, not garbage with PM_OK
    const double total = pm_pw_combine(leafsum + (size_t)q * leaf_cap, off, chunk_first, plan->chunks, n);
    out[q] = divide_by_n ? total / (double)n : total;
}

// Serial chains: KIND 0 = the three rows of fixed[:, nn] (com_target of the Fortran-ordered matches: np.mean adds the columns one
// after the other), KIND 1 = D and Sp (find_transform.py:86-91), KIND 2 = the three rows of `mov` (a Fortran-ordered moving cloud).
// One workgroup: waves 1..3 stage the next SM_CHAIN_CHUNK terms in LDS while lanes 0..chains-1 of wave 0 add the current ones.
template <int KIND>
__global__ __launch_bounds__(SM_THREADS) void sim_chain_kernel(const SimSrc s, int divide_by_n, double *__restrict__ out) {
    constexpr int C = KIND == 1 ? 2 : 3;
    __shared__ double buf[2][C][SM_CHAIN_CHUNK];
    const int tid = threadIdx.x, n = s.n;
    auto stage = [&](int b, int i0, int first_thread, int nthreads) {
        for (int e = tid - first_thread; e < SM_CHAIN_CHUNK; e += nthreads) {
            const int i = i0 + e;
            if (e < 0 || i >= n) continue;
            if (KIND == 0) {
                const int j = s.nn ? s.nn[i] : i;
                buf[b][0][e] = s.fix[j]; buf[b][1][e] = s.fix[(size_t)s.m + j]; buf[b][2 % C][e] = s.fix[2 * (size_t)s.m + j];
            } else if (KIND == 2) {
                buf[b][0][e] = s.mov[i]; buf[b][1][e] = s.mov[(size_t)n + i]; buf[b][2 % C][e] = s.mov[2 * (size_t)n + i];
            } else {
                const int j = s.nn ? s.nn[i] : i;
                const double Px = s.mov[i] - s.cs[0], Py = s.mov[(size_t)n + i] - s.cs[1], Pz = s.mov[2 * (size_t)n + i] - s.cs[2];
                const double Yx = s.fix[j] - s.ct[0], Yy = s.fix[(size_t)s.m + j] - s.ct[1], Yz = s.fix[2 * (size_t)s.m + j] - s.ct[2];
                buf[b][0][e] = __builtin_fma(Yz, Yz, __builtin_fma(Yy, Yy, Yx * Yx));     // ddot(Y[:, i], Y[:, i]), :90
                buf[b][1][e] = __builtin_fma(Pz, Pz, __builtin_fma(Py, Py, Px * Px));     // ddot(P[:, i], P[:, i]), :91
            }
        }
    };
    stage(0, 0, 0, SM_THREADS);
    __syncthreads();
    double acc = 0.0;
    int b = 0;
    for (int i0 = 0; i0 < n; i0 += SM_CHAIN_CHUNK, b ^= 1) {
        if (tid >= 64) stage(b ^ 1, i0 + SM_CHAIN_CHUNK, 64, SM_THREADS - 64);
        else if (tid < C) {
            const int cnt = min(SM_CHAIN_CHUNK, n - i0);
            const double *t = buf[b][tid];
            int e = 0;
            for (; e + 8 <= cnt; e += 8) {               // (loads batched; the additions stay one after the other)
                const double t0 = t[e], t1 = t[e + 1], t2 = t[e + 2], t3 = t[e + 3], t4 = t[e + 4], t5 = t[e + 5], t6 = t[e + 6], t7 = t[e + 7];
                acc += t0; acc += t1; acc += t2; acc += t3; acc += t4; acc += t5; acc += t6; acc += t7;
            }
            for (; e < cnt; ++e) acc += t[e];
        }
        __syncthreads();
    }
    if (tid < C) out[tid] = divide_by_n ? acc / (double)n : acc;
}

// moved = (A . [moving; 1])[:3] as dgemm computes it, in place; norms[i] = np.linalg.norm(moved[:, i] - fixed[:, nn[i]])
__global__ __launch_bounds__(SM_THREADS) void sim_apply_kernel(const double *__restrict__ A, double *__restrict__ mov, int n,
                                                               const double *__restrict__ fix, int m, const int32_t *__restrict__ nn,
                                                               double *__restrict__ norms) {
    const int i = blockIdx.x * SM_THREADS + threadIdx.x;
    if (i >= n) return;
    const double x = mov[i], y = mov[(size_t)n + i], z = mov[2 * (size_t)n + i];
    double p[3];
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        double acc = A[4 * r] * x;                           // fma(a, b, 0) = the rounded product
        acc = __builtin_fma(A[4 * r + 1], y, acc);
        acc = __builtin_fma(A[4 * r + 2], z, acc);
        acc = __builtin_fma(A[4 * r + 3], 1.0, acc);
        p[r] = acc;
    }
    mov[i] = p[0]; mov[(size_t)n + i] = p[1]; mov[2 * (size_t)n + i] = p[2];
    if (norms) {
        const int j = nn ? nn[i] : i;
        const double d0 = p[0] - fix[j], d1 = p[1] - fix[(size_t)m + j], d2 = p[2] - fix[2 * (size_t)m + j];
        norms[i] = __builtin_sqrt((d0 * d0 + d1 * d1) + d2 * d2);
    }
}

}  // namespace pm

extern "C" {

size_t pm_similar_workspace(int n) { return n > 0 ? pm::sim_layout(n, nullptr, nullptr) : 0; }

int pm_similar_moments(const double *mov, int n, const double *fix, int m, const int32_t *nn, int mov_sequential, int fix_sequential,
                       double *out17, void *ws, size_t ws_bytes, void *stream) {
    if (!mov || !fix || !out17 || n <= 0 || m <= 0) return PM_ERR_INVALID_ARG;
    if (!nn && n != m) return PM_ERR_INVALID_ARG;
    if (!ws || ((uintptr_t)ws & 255) || ws_bytes < pm_similar_workspace(n)) return PM_ERR_WORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    pm::SimWs w;
    pm::sim_layout(n, (char *)ws, &w);
    pm::SimSrc src;
    src.mov = mov; src.n = n; src.fix = fix; src.m = m; src.nn = nn;
    src.cs = out17; src.ct = out17 + 3; src.vec = nullptr;
    pm::sim_plan_kernel<<<1, 1, 0, s>>>(n, w.off, w.leaf_cap, w.chunk_first, w.chunk_cap, w.plan);
    const int groups = (w.leaf_cap * 8 + pm::SM_THREADS - 1) / pm::SM_THREADS;       // (an upper bound: the kernel reads the leaf count)
    // com_source (:28), com_target (:27)
    if (mov_sequential) pm::sim_chain_kernel<2><<<1, pm::SM_THREADS, 0, s>>>(src, 1, out17);
    else {
        pm::sim_leaf_kernel<0><<<groups, pm::SM_THREADS, 0, s>>>(src, w.off, w.plan, w.leafsum, w.leaf_cap);
        pm::sim_combine_kernel<<<1, 64, 0, s>>>(w.leafsum, w.leaf_cap, w.off, w.chunk_first, w.plan, n, 3, 1, out17);
    }
    if (fix_sequential) pm::sim_chain_kernel<0><<<1, pm::SM_THREADS, 0, s>>>(src, 1, out17 + 3);
    else {
        pm::sim_leaf_kernel<3><<<groups, pm::SM_THREADS, 0, s>>>(src, w.off, w.plan, w.leafsum, w.leaf_cap);
        pm::sim_combine_kernel<<<1, 64, 0, s>>>(w.leafsum, w.leaf_cap, w.off, w.chunk_first, w.plan, n, 3, 1, out17 + 3);
    }
    // the nine sums (:43-53), D and Sp (:86-91)
    pm::sim_leaf_kernel<1><<<groups, pm::SM_THREADS, 0, s>>>(src, w.off, w.plan, w.leafsum, w.leaf_cap);
    pm::sim_combine_kernel<<<1, 64, 0, s>>>(w.leafsum, w.leaf_cap, w.off, w.chunk_first, w.plan, n, 9, 0, out17 + 6);
    pm::sim_chain_kernel<1><<<1, pm::SM_THREADS, 0, s>>>(src, 0, out17 + 15);
    return pm::launch_status();
}

int pm_similar_apply(const double *A16, double *mov, int n, const double *fix, int m, const int32_t *nn, double *residual1,
                     void *ws, size_t ws_bytes, void *stream) {
    if (!A16 || !mov || n <= 0) return PM_ERR_INVALID_ARG;
    if (residual1 && (!fix || m <= 0 || (!nn && n != m))) return PM_ERR_INVALID_ARG;
    if (residual1 && (!ws || ((uintptr_t)ws & 255) || ws_bytes < pm_similar_workspace(n))) return PM_ERR_WORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    pm::SimWs w;
    if (residual1) pm::sim_layout(n, (char *)ws, &w);
    pm::sim_apply_kernel<<<(n + pm::SM_THREADS - 1) / pm::SM_THREADS, pm::SM_THREADS, 0, s>>>(A16, mov, n, fix, m, nn, residual1 ? w.norms : nullptr);
    if (residual1) {
        pm::SimSrc src;
        src.mov = mov; src.n = n; src.fix = fix; src.m = m; src.nn = nn; src.cs = nullptr; src.ct = nullptr; src.vec = w.norms;
        pm::sim_plan_kernel<<<1, 1, 0, s>>>(n, w.off, w.leaf_cap, w.chunk_first, w.chunk_cap, w.plan);
        const int groups = (w.leaf_cap * 8 + pm::SM_THREADS - 1) / pm::SM_THREADS;
        pm::sim_leaf_kernel<2><<<groups, pm::SM_THREADS, 0, s>>>(src, w.off, w.plan, w.leafsum, w.leaf_cap);
        pm::sim_combine_kernel<<<1, 64, 0, s>>>(w.leafsum, w.leaf_cap, w.off, w.chunk_first, w.plan, n, 1, 1, residual1);
    }
    return pm::launch_status();
}

}  // extern "C"
