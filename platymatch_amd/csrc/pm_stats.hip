// pm_stats.hip — cloud statistics: centroid, mean pairwise distance, first PCA axis.
// Reference: utils/utils.py:48-56 (get_centroid), :58-75 (get_mean_distance),
// shape_context.py:162-165 (PCA(3).fit(detections).components_[0]).
#include "pm_common.h"
#include "pm_pairwise.h"

namespace pm {

// the partial pieces below plan their leaves into 160 offsets / 4 chunk slots: a piece has at most PM_PW_CHUNK elements, i.e. at most
// PM_PW_CHUNK / 57 + 2 leaves in one chunk, so pm_pw_plan cannot overflow there (its return value is the leaf count, never -1)
static_assert(PM_PW_CHUNK / 57 + 3 <= 160, "leaf capacity of a piece");


constexpr int STAT_THREADS = 1024;

// ---- centroid in NumPy's own summation order (round 3) ------------------------------------------------------------------
// get_centroid (utils/utils.py:48-56) is np.mean(detections[:3, :], 1) for the 3 x N layout the widget passes: each row goes
// through np.add.reduce in pieces of 8 192 elements, a piece summed pairwise (csrc/pm_pairwise.h), the pieces added first to
// last, divided by N — restated here so that the centroid (hence every local frame's z axis) has the reference's bits.
// Wave r owns coordinate row r; in a full piece lane l adds up leaf l (128 consecutive elements, eight interleaved partial
// sums) and the 64 leaf sums meet in a balanced shuffle tree; the partial last piece takes the plan / leaf / combine route.
// sequential != 0: the N x 3 layout (transposed=True) — np.mean over axis 0 adds the points one after the other.
__global__ __launch_bounds__(192) void centroid_kernel(const double *__restrict__ xyz, int n, int sequential, double *__restrict__ out3) {
    __shared__ int s_off[3][160];
    __shared__ int s_cf[3][4];
    __shared__ double s_leaf[3][160];
    const int lane = threadIdx.x & 63, r = threadIdx.x >> 6;
    const double *row = xyz + (size_t)r * n;
    double acc = 0.0;
    if (sequential) {
        if (lane == 0) {
            int i = 0;
            for (; i + 8 <= n; i += 8) {
                const double t0 = row[i], t1 = row[i + 1], t2 = row[i + 2], t3 = row[i + 3], t4 = row[i + 4], t5 = row[i + 5], t6 = row[i + 6], t7 = row[i + 7];
                acc += t0; acc += t1; acc += t2; acc += t3; acc += t4; acc += t5; acc += t6; acc += t7;
            }
            for (; i < n; ++i) acc += row[i];
            out3[r] = acc / (double)n;
        }
        return;
    }
    for (int c0 = 0; c0 < n; c0 += PM_PW_CHUNK) {
        const int len = min(PM_PW_CHUNK, n - c0);
        double piece;
        if (len == PM_PW_CHUNK) {
            double keep = pm_pw_leaf_sum(row + c0 + lane * PM_PW_LEAF, PM_PW_LEAF);
#pragma unroll
            for (int st = 1; st < 64; st <<= 1) keep = keep + __shfl_down(keep, st, 64);
            piece = keep;
        } else {
            if (lane == 0) s_cf[r][3] = pm_pw_plan(len, s_off[r], 160, s_cf[r], 4);
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            const int leaves = s_cf[r][3];
            for (int lf = lane; lf < leaves; lf += 64) s_leaf[r][lf] = pm_pw_leaf_sum(row + c0 + s_off[r][lf], s_off[r][lf + 1] - s_off[r][lf]);
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
            piece = (lane == 0) ? pm_pw_combine(s_leaf[r], s_off[r], s_cf[r], 1, len) : 0.0;
        }
        acc = (c0 == 0) ? piece : acc + piece;           // (meaningful in lane 0)
    }
    if (lane == 0) out3[r] = acc / (double)n;
}

// ---- mean pairwise distance, in the reference's own arithmetic (round 3) --------------------------------------------------
// get_mean_distance (utils/utils.py:58-75) appends np.linalg.norm(p_i - p_j) for i < j in lexicographic order to a list and
// takes np.average of it.  Restated exactly:
//   element   np.linalg.norm of a 3-vector is sqrt(x.dot(x)), BLAS ddot, whose x86-64 kernels accumulate with fused
//             multiply-adds: sqrt(fma(d2, d2, fma(d1, d1, d0 * d0)))   (verified against NumPy on the reference's fixtures);
//   mean      np.add.reduce over the P = N(N-1)/2 elements in pieces of 8 192 (np.getbufsize()), every piece summed pairwise
//             (csrc/pm_pairwise.h), the piece sums added first to last, divided by P.
// The value equals the reference's bit for bit (tests: all twelve fixture clouds), so the ring radii — mean distance times the
// logspace edges — are the reference's by construction.  A wave owns a piece: its lanes hold 64 consecutive elements (loads of
// p_j contiguous within a row of the pair list, p_i mostly wave-uniform), a 128-element leaf is two such steps, NumPy's eight
// interleaved partial sums run as chains through the lanes of a 16-lane row (DPP moves; layout at the loop below), the 64 leaf
// sums of a full piece meet in a balanced tree.  The last, partial piece takes the general plan / leaf / combine route of
// pm_pairwise.h.  Piece sums go to `partial[]`; a second launch adds them one after the other (staged through LDS) and divides.
constexpr int MDX_WAVES = 4;
constexpr int MDX_LEAF_PITCH = 136;       // doubles between the leaves of a block in LDS (128 + 8: see mean_distance_chunks)

template <int CTRL>
__device__ __forceinline__ double dpp_f64(double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(lo, lo, CTRL, 0xf, 0xf, false);
    hi = __builtin_amdgcn_update_dpp(hi, hi, CTRL, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double readlane_f64(double v, int lane) {
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), lane), __builtin_amdgcn_readlane(__double2loint(v), lane));
}

__device__ __forceinline__ long long md_row_start(long long i, long long n) { return i * (n - 1) - i * (i - 1) / 2; }

__device__ __forceinline__ void md_locate(long long e, int n, int &i, int &j) {
    const double t = 2.0 * n - 1.0;
    long long ii = (long long)((t - __builtin_sqrt(__builtin_fmax(t * t - 8.0 * (double)e, 0.0))) * 0.5);
    ii = ii < 0 ? 0 : (ii > n - 2 ? n - 2 : ii);
    while (ii < n - 2 && md_row_start(ii + 1, n) <= e) ++ii;
    while (ii > 0 && md_row_start(ii, n) > e) --ii;
    i = (int)ii;
    j = (int)(ii + 1 + (e - md_row_start(ii, n)));
}

__device__ __forceinline__ double md_dist(const double *__restrict__ P0, const double *__restrict__ P1, const double *__restrict__ P2,
                                          int i, int j) {
    const double d0 = P0[i] - P0[j], d1 = P1[i] - P1[j], d2 = P2[i] - P2[j];
    return __builtin_sqrt(__builtin_fma(d2, d2, __builtin_fma(d1, d1, d0 * d0)));
}

__device__ __forceinline__ void md_advance(int &i, int &j, int by, int n) {
    j += by;
    while (j >= n && i < n - 2) { const int over = j - n; ++i; j = i + 1 + over; }
}

// a block that crosses rows of the pair list: every lane follows its own pair
__device__ __forceinline__ void md_block_across_rows(const double *__restrict__ P0, const double *__restrict__ P1,
                                                               const double *__restrict__ P2, int n, long long eb, int lane, double *blk) {
    int i, j;
    md_locate(eb + lane, n, i, j);
    for (int t = 0; t < 16; ++t) {
        blk[(t >> 1) * MDX_LEAF_PITCH + (t & 1) * 64 + lane] = md_dist(P0, P1, P2, i, j);
        md_advance(i, j, 64, n);
    }
}

// The partial last piece: plan its leaves (lane 0), leaf sums by all lanes, tree by lane 0 -> the piece sum (valid in lane 0).
// A kernel of its own (mean_distance_partial_piece, one wave) since round 4's second session: inside mean_distance_chunks its
// plan, unrolled leaf sums and index arithmetic raised that kernel to 191 vector registers — two waves per SIMD for a loop that
// needs more to hide its square-root chains — for the sake of ONE piece in 152 588 at 50 000 points.
__device__ __forceinline__ double md_partial_piece(const double *__restrict__ P0, const double *__restrict__ P1,
                                                             const double *__restrict__ P2, int n, long long e0, int len, int lane,
                                                             double *s_leaf_w, int *s_off_w, int *s_cf_w) {
    double chunk_sum;
    if (lane == 0) {
        const int leaves = pm_pw_plan(len, s_off_w, 160, s_cf_w, 4);
        s_cf_w[3] = leaves;
    }
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    const int leaves = s_cf_w[3];
    for (int lf = lane; lf < leaves; lf += 64) {
        const int o = s_off_w[lf], ll = s_off_w[lf + 1] - o;
        int i, j;
        md_locate(e0 + o, n, i, j);
        double res;
        if (ll < 8) {
            res = 0.0;
            for (int t = 0; t < ll; ++t) { res += md_dist(P0, P1, P2, i, j); md_advance(i, j, 1, n); }
        } else {
            double r[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) { r[u] = md_dist(P0, P1, P2, i, j); md_advance(i, j, 1, n); }
            int t = 8;
            for (; t < ll - (ll % 8); t += 8) {
#pragma unroll
                for (int u = 0; u < 8; ++u) { r[u] += md_dist(P0, P1, P2, i, j); md_advance(i, j, 1, n); }
            }
            res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
            for (; t < ll; ++t) { res += md_dist(P0, P1, P2, i, j); md_advance(i, j, 1, n); }
        }
        s_leaf_w[lf] = res;
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    chunk_sum = 0.0;
    if (lane == 0) chunk_sum = pm_pw_combine(s_leaf_w, s_off_w, s_cf_w, 1, len);
    return chunk_sum;
}

__global__ __launch_bounds__(MDX_WAVES * 64) void mean_distance_chunks(const double *__restrict__ xyz, int n, long long P, int nchunks,
                                                                       int first, int stride, double *__restrict__ partial) {
    __shared__ double s_blk[MDX_WAVES][8 * MDX_LEAF_PITCH];          // a wave's block of 1 024 distances (eight leaves)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const double *P0 = xyz, *P1 = xyz + (size_t)n, *P2 = xyz + 2 * (size_t)n;
    const long long waves_total = (long long)gridDim.x * MDX_WAVES;
    // (this rank's pieces first, first + stride, ...; the last piece of the list — the only one that can be partial — goes first)
    const long long mine = (nchunks - 1 - first) / stride + 1;
    for (long long k = (long long)blockIdx.x * MDX_WAVES + wave; k < mine; k += waves_total) {
        const int c = first + (int)(mine - 1 - k) * stride;
        const long long e0 = (long long)c * PM_PW_CHUNK;
        const int len = (int)((P - e0 < PM_PW_CHUNK) ? P - e0 : PM_PW_CHUNK);
        double chunk_sum;
        if (len == PM_PW_CHUNK) {
            // Round 4.  A piece is 64 leaves of 128 elements; NumPy runs eight interleaved partial sums through a leaf (r[k] takes
            // elements k, k + 8, ..., k + 120, one after the other) and combines them ((r0 + r1) + (r2 + r3)) + ((r4 + r5) + (r6 + r7)).
            // The piece is taken in eight blocks of 1 024 elements (eight leaves).  Per block:
            //   1. the 1 024 distances are computed with the lanes in ELEMENT order (coalesced loads of p_j) and written to this
            //      wave's LDS block; a block that lies inside one row of the pair list — all but ~4 % at 50 000 points — has ONE p_i
            //      (wave-uniform, scalar registers) and p_j = p[jb + 64 t + lane]: no per-lane index arithmetic at all;
            //   2. lane (leaf l = lane / 8, partial sum k = lane % 8) adds its sixteen elements l * 128 + k + 8 t in order from LDS
            //      (leaves 136 doubles apart: the 64 lanes of a read fall on all 32 bank pairs twice — the natural rate of 8-byte
            //      reads) — no cross-lane chain (round 3 chained the partial sums through the lanes by DPP moves: ~25 vector
            //      instructions per element; this is one write, one read and one add);
            //   3. the eight partial sums of a leaf meet by three row-shift adds, in NumPy's bracketing.
            // The 64 leaf sums of the piece are then added by the balanced tree below, as before.  Same operations on the same
            // operands in the same order as round 3's kernel, hence the same bits (tests/test_gpu_parity.py::test_statistics*).
            double *blk = s_blk[wave];
            const int l = lane >> 3, kk = lane & 7;
            double keep = 0.0;
            for (int it = 0; it < 8; ++it) {
                const long long eb = e0 + (long long)it * 1024;
                int ib, jb;
                md_locate(eb, n, ib, jb);                            // (every lane computes the same pair: wave-uniform)
                ib = __builtin_amdgcn_readfirstlane(ib);
                jb = __builtin_amdgcn_readfirstlane(jb);
                if (jb + 1024 <= n) {                                // the whole block pairs p_ib with p_jb .. p_jb+1023
                    const double a0 = P0[ib], a1 = P1[ib], a2 = P2[ib];
                    const double *q0 = P0 + jb + lane, *q1 = P1 + jb + lane, *q2 = P2 + jb + lane;
#pragma unroll 4
                    for (int t = 0; t < 16; ++t) {
                        const double d0 = a0 - q0[64 * t], d1 = a1 - q1[64 * t], d2 = a2 - q2[64 * t];
                        blk[(t >> 1) * MDX_LEAF_PITCH + (t & 1) * 64 + lane] = __builtin_sqrt(__builtin_fma(d2, d2, __builtin_fma(d1, d1, d0 * d0)));
                    }
                } else {                                             // the block crosses rows of the pair list: every lane follows its own pair
                    md_block_across_rows(P0, P1, P2, n, eb, lane, blk);
                }
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                __builtin_amdgcn_wave_barrier();
                const double *lf = blk + l * MDX_LEAF_PITCH + kk;
                double r = lf[0];
#pragma unroll
                for (int t = 1; t < 16; ++t) r = r + lf[8 * t];
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                __builtin_amdgcn_wave_barrier();                     // (the next block's writes come after these reads)
                const double s1 = r + dpp_f64<0x101>(r);             // row_shl:1 — lane k reads lane k + 1: r[k] + r[k + 1] (k even)
                const double s2 = s1 + dpp_f64<0x102>(s1);           // (r0 + r1) + (r2 + r3) in k = 0, (r4 + r5) + (r6 + r7) in k = 4
                const double s3 = s2 + dpp_f64<0x104>(s2);           // the leaf sum, in lane 8 l
#pragma unroll
                for (int w = 0; w < 8; ++w) {
                    const double ls = readlane_f64(s3, 8 * w);
                    if (lane == it * 8 + w) keep = ls;
                }
            }
#pragma unroll
            for (int st = 1; st < 64; st <<= 1) keep = keep + __shfl_down(keep, st, 64);   // balanced tree: left + right
            chunk_sum = keep;                                        // valid in lane 0
        } else {
            continue;                                                // the one partial piece of a cloud: mean_distance_partial_piece
        }
        if (lane == 0) partial[c] = chunk_sum;
    }
}

// The last piece of the pair list when it is not a full one (one wave; launched beside mean_distance_chunks by the rank that owns it).
__global__ __launch_bounds__(64) void mean_distance_partial_piece(const double *__restrict__ xyz, int n, long long P, int nchunks,
                                                                  double *__restrict__ partial) {
    __shared__ int s_off[160];
    __shared__ int s_cf[4];
    __shared__ double s_leaf[160];
    const int c = nchunks - 1;
    const long long e0 = (long long)c * PM_PW_CHUNK;
    const int len = (int)(P - e0);
    const double s = md_partial_piece(xyz, xyz + (size_t)n, xyz + 2 * (size_t)n, n, e0, len, (int)threadIdx.x, s_leaf, s_off, s_cf);
    if (threadIdx.x == 0) partial[c] = s;
}

// the piece sums, one after the other (np.add.reduce across its buffer-sized pieces), divided by P
__global__ __launch_bounds__(256) void mean_distance_final(const double *__restrict__ partial, int nchunks, long long P,
                                                           double *__restrict__ out1) {
    constexpr int CH = 2048;
    __shared__ double buf[2][CH];
    const int tid = threadIdx.x;
    for (int e = tid; e < CH; e += 256) buf[0][e] = (e < nchunks) ? partial[e] : 0.0;
    __syncthreads();
    double acc = 0.0;
    int b = 0;
    for (int c0 = 0; c0 < nchunks; c0 += CH, b ^= 1) {
        if (tid >= 64) {
            for (int e = tid - 64; e < CH; e += 192) buf[b ^ 1][e] = (c0 + CH + e < nchunks) ? partial[c0 + CH + e] : 0.0;
        } else if (tid == 0) {
            const int cnt = min(CH, nchunks - c0);
            const double *t = buf[b];
            int e = 0;
            if (c0 == 0) { acc = t[0]; e = 1; }
            // batches of 32 LDS reads, then their 32 additions one after the other: 12 cycles per addition, within a third of
            // the dependent float64 add's latency (software-pipelined variants measured slower: 0.96 / 1.13 ms against 0.79 ms
            // for the 152 588 piece sums of a 50 000-point cloud)
            for (; e + 32 <= cnt; e += 32) {
                double u[32];
#pragma unroll
                for (int k = 0; k < 32; ++k) u[k] = t[e + k];
#pragma unroll
                for (int k = 0; k < 32; ++k) acc += u[k];
            }
            for (; e < cnt; ++e) acc += t[e];
        }
        __syncthreads();
    }
    if (tid == 0) out1[0] = acc / (double)P;
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
    pm::centroid_kernel<<<1, 192, 0, (hipStream_t)stream>>>(xyz, n, 0, out3);
    return pm::launch_status();
}

int pm_centroid_sequential(const double *xyz, int n, double *out3, void *stream) {
    if (!xyz || !out3 || n <= 0) return PM_ERR_INVALID_ARG;
    pm::centroid_kernel<<<1, 192, 0, (hipStream_t)stream>>>(xyz, n, 1, out3);
    return pm::launch_status();
}

static long long md_pairs(int n) { return (long long)n * (n - 1) / 2; }
static int md_chunks(int n) { return (int)((md_pairs(n) + PM_PW_CHUNK - 1) / PM_PW_CHUNK); }

size_t pm_mean_distance_workspace(int n) {
    if (n < 2) return 0;
    return ((size_t)md_chunks(n) * sizeof(double) + 255) / 256 * 256;
}

static int md_launch(const double *xyz, int n, int first, int stride, double *partial, hipStream_t s) {
    const int nchunks = md_chunks(n);
    if (first >= nchunks) return PM_OK;
    const long long mine = (nchunks - 1 - first) / stride + 1;
    const long long want = (mine + pm::MDX_WAVES - 1) / pm::MDX_WAVES;
    const int blocks = (int)(want < 8192 ? want : 8192);             // (waves stride over the pieces beyond that)
    const long long P = md_pairs(n);
    if (P % PM_PW_CHUNK != 0 && (nchunks - 1 - first) % stride == 0)    // this rank holds the last piece and it is a partial one
        pm::mean_distance_partial_piece<<<1, 64, 0, s>>>(xyz, n, P, nchunks, partial);
    // (loads batched 2, 4, 8 or 16 at a time measure the same, 2.71-2.77 ms with the serial finish at 50 000 points: profiles/r04_stats_timing.txt)
    pm::mean_distance_chunks<<<blocks, pm::MDX_WAVES * 64, 0, s>>>(xyz, n, P, nchunks, first, stride, partial);
    return pm::launch_status();
}

int pm_mean_distance(const double *xyz, int n, double *out1, void *ws, size_t ws_bytes, void *stream) {
    if (!xyz || !out1 || n < 2) return PM_ERR_INVALID_ARG;
    if (n > 2000000) return PM_ERR_UNSUPPORTED;                      // (the piece count must fit an int: N(N-1)/2 / 8192)
    if (!ws || ws_bytes < pm_mean_distance_workspace(n)) return PM_ERR_WORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    const int rc = md_launch(xyz, n, 0, 1, (double *)ws, s);
    if (rc != PM_OK) return rc;
    pm::mean_distance_final<<<1, 256, 0, s>>>((const double *)ws, md_chunks(n), md_pairs(n), out1);
    return pm::launch_status();
}

int pm_mean_distance_rows(const double *xyz, int n, int row_offset, int row_stride, double *partials, size_t partial_bytes,
                          void *stream) {
    if (!xyz || !partials || n < 2 || row_offset < 0 || row_stride < 1 || row_offset >= row_stride) return PM_ERR_INVALID_ARG;
    if (n > 2000000) return PM_ERR_UNSUPPORTED;
    if (partial_bytes < pm_mean_distance_workspace(n)) return PM_ERR_WORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    if (hipMemsetAsync(partials, 0, pm_mean_distance_workspace(n), s) != hipSuccess) return pm::launch_status();
    return md_launch(xyz, n, row_offset, row_stride, partials, s);
}

int pm_mean_distance_finish(const double *partials, int n, double *out1, void *stream) {
    if (!partials || !out1 || n < 2) return PM_ERR_INVALID_ARG;
    if (n > 2000000) return PM_ERR_UNSUPPORTED;
    pm::mean_distance_final<<<1, 256, 0, (hipStream_t)stream>>>(partials, md_chunks(n), md_pairs(n), out1);
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
