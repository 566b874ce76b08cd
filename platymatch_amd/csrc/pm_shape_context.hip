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
//
// Round 3: the default path is the TILE kernel below (sc_tile_kernel): a workgroup owns SC_Q consecutive queried points, a
// lane keeps one neighbour's coordinates in registers and walks the tile's queries (their frames are wave-uniform: scalar
// loads from the frame table the prepare kernel wrote, SGPR operands), the neighbour is pre-classified in float32
// (pm_bin_fast32: accepted only when clear of every bin boundary by more than float32 can be off, ~3 999 of 4 000) and
// decided by the float64 expressions otherwise; ONE histogram per query (frame 1) is kept in LDS — frames 2..4 are its phi
// permutations, written out as such.  A neighbour whose exact per-frame bins are NOT that permutation (it sits on a sector
// edge or a pole of the frame) marks the tile, and the one-workgroup-per-point kernel (shape_context_kernel, the round-1
// design, four explicit histograms) recomputes the marked tiles: same results in every case, by construction.
#include <algorithm>
#include "pm_common.h"
#include "pm_binning.h"

namespace pm {

constexpr int SC_THREADS = 256;
constexpr int SC_WAVES = SC_THREADS / 64;
constexpr int SC_Q = 16;              // queried points per workgroup of the tile kernel (LDS: 16 x 1 440 B)
constexpr int SC_FR = 10;             // doubles per row of frames64
// Edge guard: the mean pairwise distance and the PCA axis agree with the reference's to 1e-14 / 1e-12 (by test — observed
// 4e-16 / 9e-14 on the reference's fixtures —, not by
// construction: other summation orders, another eigen-solver).  A neighbour whose distance lies within PM_GUARD_RING (relative) of a
// ring radius, or whose azimuth lies within PM_GUARD_ANGLE / sin(angle(axis, z)) of a sector edge, could be binned differently by
// the reference.  They are COUNTED (two counters per launch) so that "identical histograms" is a checked statement per call.
// Round 3, second half: the reference's OWN local coordinates are noisy.  transform() (shape_context.py:61-84) builds them as
// B . inv(A) . [p; 1] with np.linalg.inv of a 4 x 4 whose entries are the query's world coordinates: LAPACK's rounding leaves
// an absolute error of up to 4e-14 x (|d_x| + |d_y| + |d_z|) on every local coordinate (measured against 200-bit arithmetic,
// tests/golden/soak_oracle_vs_reference.py and DESIGN.md §5: 1.5e-11 for coordinates around 200), which no restatement can
// follow bit for bit.  A neighbour that lies within that noise of a bin boundary — on lattice-like data: exactly ON it — is
// binned by the reference as its LAPACK build happens to round.  The guard therefore also counts every neighbour within
// PM_GUARD_REF x |d|_1 (absolute, 4x the measured maximum) of a ring radius (counter 0), of a sector plane or of a polar cone
// (counter 1): guard = 0 then means that the histograms are the reference's whatever its linear algebra library rounds like.
#define PM_GUARD_RING 4e-14
#define PM_GUARD_ANGLE 1e-12
#define PM_GUARD_REF 1.6e-13

// What the prepare kernel leaves in the workspace for a launch over rows [row0, row0 + nrows):
//   ScParams            ring thresholds, 64 / md^2 in float32, whether the float32 pre-classification may be used
//   frames32 [nrows][16 dwords]  per queried point ONE 64-byte record {x, y, z of its frame as 9 floats, pad, its coordinates as
//                                3 doubles}: the tile kernel fetches a query with a single s_load_dwordx16
//   frames64 [nrows][10] double  the same frame in float64: what the exact path projects with (the oracle's bits); [9] = |x0 - z (x0.z)|,
//                                the sine of the angle between the PCA axis and z (how much an error of the axis turns x and y)
//   redo     [tiles]    int32    1 = recompute this tile with the general kernel
struct alignas(64) ScQuery {          // what the tile kernel needs of one queried point: 16 dwords, one scalar load
    float fr[9];
    float min_L;                       // 2^17 x PM_GUARD_REF x |p|_1: float32's verdict is only taken for neighbours longer than this (pm_bin_fast32)
    double p[3];
};
static_assert(sizeof(ScQuery) == 64, "one s_load_dwordx16");

struct ScParams {
    double rho[4];
    float k64;
    int fast_ok;
    int pad[2];
};

__device__ __forceinline__ void local_frame(const double *__restrict__ xyz, int n, int i, const double *__restrict__ centroid3,
                                            const double *__restrict__ x0_3, double fr[SC_FR]) {
    // shape_context.py:169-175; one rounding per written operation, in the oracle's order
    const double p0 = xyz[i], p1 = xyz[(size_t)n + i], p2 = xyz[2 * (size_t)n + i];
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
    fr[0] = x0; fr[1] = x1; fr[2] = x2; fr[3] = y0; fr[4] = y1; fr[5] = y2; fr[6] = z0; fr[7] = z1; fr[8] = z2;
    fr[9] = nx / __builtin_sqrt((a0 * a0 + a1 * a1) + a2 * a2);        // sin(angle(axis, z)) (the axis is a unit vector up to rounding)
}

__global__ __launch_bounds__(256) void sc_prepare_kernel(const double *__restrict__ xyz, int n, int row0, int nrows,
                                                         const double *__restrict__ centroid3, const double *__restrict__ x0_3,
                                                         const double *__restrict__ mean_dist1, ScParams *__restrict__ prm,
                                                         float *__restrict__ frames32, double *__restrict__ frames64) {
    const int r = blockIdx.x * 256 + threadIdx.x;
    if (r == 0) {
        const double md = mean_dist1[0];
        double rho[4];
        pm_ring_thresholds(md, rho);
#pragma unroll
        for (int k = 0; k < 4; ++k) prm->rho[k] = rho[k];
        const double k64 = 64.0 / (md * md);
        const int ok = (md > 0.0) && (k64 >= 0x1p-30) && (k64 <= 0x1p+30);       // false for NaN
        prm->k64 = ok ? (float)k64 : 0.0f;
        prm->fast_ok = ok;
        prm->pad[0] = prm->pad[1] = 0;
    }
    if (r >= nrows) return;
    double fr[SC_FR];
    local_frame(xyz, n, row0 + r, centroid3, x0_3, fr);
    ScQuery rec;
#pragma unroll
    for (int k = 0; k < 9; ++k) { frames64[(size_t)r * SC_FR + k] = fr[k]; rec.fr[k] = (float)fr[k]; }
    frames64[(size_t)r * SC_FR + 9] = fr[9];
    const int i = row0 + r;
    rec.p[0] = xyz[i]; rec.p[1] = xyz[(size_t)n + i]; rec.p[2] = xyz[2 * (size_t)n + i];
    rec.min_L = (float)(PM_GUARD_REF * 0x1p+17 * ((__builtin_fabs(rec.p[0]) + __builtin_fabs(rec.p[1])) + __builtin_fabs(rec.p[2]))) * 1.0001f;
    ((ScQuery *)frames32)[r] = rec;
}

// One neighbour of one query, decided in float64 with the oracle's operations; -> frame 1's bin if the frames' bins are the
// permutations the tile kernel's output assumes (or every frame drops it: -1), -2 if they are not (the tile must be redone).
template <int NF>
__device__ __noinline__ int sc_exact_bin(double v0, double v1, double v2, const double *__restrict__ fr,
                                         const double *__restrict__ rho_g, unsigned int *__restrict__ guard, double d_l1,
                                         bool is_self) {
    const double rho[4] = {rho_g[0], rho_g[1], rho_g[2], rho_g[3]};
    const double vx = (fr[0] * v0 + fr[1] * v1) + fr[2] * v2;
    const double vy = (fr[3] * v0 + fr[4] * v1) + fr[5] * v2;
    const double vz = (fr[6] * v0 + fr[7] * v1) + fr[8] * v2;
    {   // edge guard: only neighbours that float32 could not clear come here, and nothing nearer than 2^-17 escapes them
        const double s = (vx * vx + vy * vy) + vz * vz;
        if (s == 0.0 && !is_self) {
            // a DUPLICATE of the queried point: here (and in exact arithmetic) its direction is 0/0 and it is not counted, as
            // arccos(0/0) = NaN is not in the reference — but the reference's inv()-based coordinates of it are ~1e-12, not 0,
            // and it lands in some bin of the innermost ring, whichever way the noise points
            atomicAdd(&guard[0], 1u);
            atomicAdd(&guard[1], 1u);
        }
        if (s > 0.0 && s < INFINITY) {
            const double r_ = __builtin_sqrt(s);
            const double ref_noise = PM_GUARD_REF * d_l1;                          // what the reference's inv() leaves on a local coordinate
            bool ring_near = false;
#pragma unroll
            for (int k = 0; k < 4; ++k)
                ring_near = ring_near || (rho[k] > 0.0 && __builtin_fabs(r_ - rho[k]) <= __builtin_fmax(PM_GUARD_RING * rho[k], ref_noise));
            const double ax = __builtin_fabs(vx), ay = __builtin_fabs(vy), az = __builtin_fabs(vz);
            const double pl = __builtin_sqrt(ax * ax + ay * ay);                 // in-plane radius: azimuthal distances are relative to it
            // distance (in-plane length) to the nearest of the twelve sector edges: the axes, the 30 and the 60 degree rays of each quadrant
            const double near = __builtin_fmin(__builtin_fmin(ax, ay),
                                               __builtin_fmin(__builtin_fabs(ay - PM_TAN30 * ax) * 0.8660254037844386,
                                                              __builtin_fabs(ay - PM_TAN60 * ax) * 0.5));
            const double g_ang = PM_GUARD_ANGLE / __builtin_fmax(fr[9], 1e-6);
            bool angle_near = !(near > __builtin_fmax(g_ang * pl, ref_noise));     // (also true on the frame's z axis, pl = 0)
            // ... and to the nearest polar cone (theta = 30, 60, 90, 120, 150 degrees; by symmetry in |z|: the plane z = 0 and two cones)
            const double cone = __builtin_fmin(az, __builtin_fmin(__builtin_fabs(az * 0.5 - pl * 0.8660254037844386),
                                                                  __builtin_fabs(az * 0.8660254037844386 - pl * 0.5)));
            angle_near = angle_near || !(cone > ref_noise);
            if (ring_near) atomicAdd(&guard[0], 1u);
            if (angle_near) atomicAdd(&guard[1], 1u);
        }
    }
    int b[4];
    pm_bin_index4(vx, vy, vz, rho, NF, b);
    if (b[0] == PM_DROP) {
        bool all = true;
#pragma unroll
        for (int f = 1; f < NF; ++f) all = all && (b[f] == PM_DROP);
        return all ? -1 : -2;
    }
    bool conform = true;
#pragma unroll
    for (int f = 1; f < NF; ++f) conform = conform && (b[f] == pm_bin_perm(f, b[0]));
    return conform ? b[0] : -2;
}

// grid = (tiles, segments): workgroup (t, g) bins the neighbours [g * seg_len, (g + 1) * seg_len) for the 16 rows of tile t and
// adds its histogram into cnt1 (integer atomics: exact, order-free).  Splitting the neighbours keeps the work units small
// against the chip (a 50 000-row launch has 3 125 tiles for 1 536 resident workgroups — 2.03 per slot, i.e. a third round
// for 53 of them — and a rank's 6 250-row block of an 8-GPU run would fill a quarter of the slots).
template <int NF>
__global__ __launch_bounds__(SC_THREADS) void sc_tile_kernel(
    const double *__restrict__ xyz, int n, int row0, int nrows, int seg_len, const ScParams *__restrict__ prm,
    const float *__restrict__ frames32, const double *__restrict__ frames64, unsigned int *__restrict__ cnt1,
    int32_t *__restrict__ redo, unsigned int *__restrict__ guard) {
    __shared__ unsigned int h[SC_Q][PM_NBINS];
    __shared__ int s_redo;
    const int tid = threadIdx.x;
    const int q0 = blockIdx.x * SC_Q;                      // first row of the tile (within the row block)
    const int nq = min(SC_Q, nrows - q0);
    const int j_begin = blockIdx.y * seg_len, j_end = min(n, j_begin + seg_len);
    for (int k = tid; k < SC_Q * PM_NBINS; k += SC_THREADS) (&h[0][0])[k] = 0u;
    if (tid == 0) s_redo = 0;
    __syncthreads();

    const double *P0 = xyz, *P1 = xyz + (size_t)n, *P2 = xyz + 2 * (size_t)n;
    const float k64 = prm->k64;
    const bool fast_ok = prm->fast_ok != 0;

    for (int j0 = j_begin; j0 < j_end; j0 += SC_THREADS) {
        const int j = j0 + tid;
        const bool valid = j < j_end;
        const int jj = valid ? j : j_end - 1;
        const double pj0 = P0[jj], pj1 = P1[jj], pj2 = P2[jj];
        // the tile's queries one after the other (wave-uniform): each query's point and float32 frame arrive by scalar loads,
        // fetched one query ahead so that their latency hides behind the previous query's arithmetic
        // the tile's queries one after the other (wave-uniform): a query's record arrives by scalar loads, fetched one query
        // ahead (two register sets used in turn, no copies) so that its latency hides behind the previous query's arithmetic
        const ScQuery *qrec = (const ScQuery *)frames32 + q0;
        auto one_query = [&](const ScQuery &r, int q) {
            const double v0 = pj0 - r.p[0], v1 = pj1 - r.p[1], v2 = pj2 - r.p[2];   // np.delete (:168): the point itself gives v = 0, dropped below as NaN
            const float fr[9] = {r.fr[0], r.fr[1], r.fr[2], r.fr[3], r.fr[4], r.fr[5], r.fr[6], r.fr[7], r.fr[8]};
            // the reference's inv() leaves PM_GUARD_REF x |p|_1 on every local coordinate of this query: float32's verdict is taken
            // only for neighbours whose own 2^-17 clearance exceeds it (r.min_L, from the prepare kernel; pm_bin_fast32)
            int bin = fast_ok ? pm_bin_fast32((float)v0, (float)v1, (float)v2, fr, k64, r.min_L) : -1;
            if (bin < 0 && valid) {
                // not clear of a boundary in float32 (or the pair of the point with itself / a duplicate: v = 0 -> NaN -> not
                // counted, exactly as arccos(0/0) in the reference): the float64 expressions decide
                bin = sc_exact_bin<NF>(v0, v1, v2, frames64 + (size_t)(q0 + q) * SC_FR, prm->rho, guard,
                                       (__builtin_fabs(r.p[0]) + __builtin_fabs(r.p[1])) + __builtin_fabs(r.p[2]), j == row0 + q0 + q);
                if (bin == -2) s_redo = 1;
            }
            if (valid && bin >= 0) atomicAdd(&h[q][bin], 1u);
        };
        ScQuery ra = qrec[0];
        for (int q = 0; q < nq; q += 2) {
            const ScQuery rb = qrec[min(q + 1, nq - 1)];
            one_query(ra, q);
            ra = qrec[min(q + 2, nq - 1)];
            if (q + 1 < nq) one_query(rb, q + 1);
        }
    }
    __syncthreads();
    if (s_redo) {                                          // some neighbour sits on a sector edge or pole: general kernel
        if (tid == 0) redo[blockIdx.x] = 1;                // (zeroed by the prepare kernel; only ever set)
        return;
    }
    for (int k = tid; k < nq * PM_NBINS; k += SC_THREADS) {
        const unsigned int c = (&h[0][0])[k];
        if (c) atomicAdd(&cnt1[(size_t)q0 * PM_NBINS + k], c);
    }
}

// Frame 1's counts -> the outputs of all frames: totals, counts, normalised histograms (one workgroup per row).
template <int NF>
__global__ __launch_bounds__(128) void sc_finish_kernel(const unsigned int *__restrict__ cnt1, int nrows, const int32_t *__restrict__ redo,
                                                        int32_t *__restrict__ counts, int32_t *__restrict__ totals,
                                                        double *__restrict__ hist) {
    __shared__ unsigned int c_s[PM_NBINS];
    __shared__ unsigned int part[2];
    const int row = blockIdx.x, tid = threadIdx.x;
    if (redo[row / SC_Q]) return;                          // the general kernel writes this tile's rows
    unsigned int mine = 0;
    for (int k = tid; k < PM_NBINS; k += 128) {
        const unsigned int c = cnt1[(size_t)row * PM_NBINS + k];
        c_s[k] = c;
        mine += c;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) mine += __shfl_down(mine, off, PM_WAVE);
    if ((tid & 63) == 0) part[tid >> 6] = mine;
    __syncthreads();
    // every frame counts the same neighbours (a permutation inside each (ring, theta) shell): one total
    const unsigned int tot = part[0] + part[1];
#pragma unroll
    for (int f = 0; f < NF; ++f) {
        for (int bin = tid; bin < PM_NBINS; bin += 128) {
            const unsigned int c = c_s[pm_bin_perm(f, bin)];                       // frame f's bin <- frame 1's (self-inverse maps)
            const size_t o = ((size_t)f * nrows + row) * PM_NBINS + bin;
            if (counts) counts[o] = (int32_t)c;
            if (hist) hist[o] = (double)c / (double)tot;                           // sc / sc.sum() (:41); 0/0 = NaN as in the reference
        }
        if (totals && tid == 0) totals[(size_t)f * nrows + row] = (int32_t)tot;
    }
}

template <int NF>
__global__ __launch_bounds__(SC_THREADS) void shape_context_kernel(
    const double *__restrict__ xyz, int n, int row0, const double *__restrict__ centroid3,
    const double *__restrict__ x0_3, const double *__restrict__ mean_dist1, int32_t *__restrict__ counts,
    int32_t *__restrict__ totals, double *__restrict__ hist, int nrows, const int32_t *__restrict__ redo) {
    __shared__ unsigned int h[SC_WAVES][NF][PM_NBINS];
    __shared__ unsigned int tot_s[NF];
    const int tid = threadIdx.x, wave = tid >> 6;
    const int row = blockIdx.x;       // row within this block of rows
    if (redo && !redo[row / SC_Q]) return;                 // (behind the tile kernel: only the tiles it marked)
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
        const double r_ = __builtin_sqrt(__builtin_fma(vz, vz, __builtin_fma(vy, vy, vx * vx)));   // np.linalg.norm of three numbers: BLAS ddot's fused chain (as pm_stats.hip)
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

// get_shape_context with the CALLER's binning (shape_context.py:10-42 called with explicit r_inner, r_outer, n_rbins,
// n_thetabins, n_phibins — get_unary never does, the stand-alone function's signature allows it).  The reference evaluates
// arccos / arctan2 and two float floor divisions per neighbour (:31-35, :51-52).  Here, per neighbour:
//   ring   the reference's own loop: first edge of np.logspace(...) with r_/mean_dist < edge, else n_rbins - 1 (IEEE division and
//          comparisons: the same bits on any machine);
//   theta  theta_index = #{k : c <= cos_steps[k]}, c = z_/r_ — cos_steps[k] is the largest float64 c whose
//          np.arccos(c) // (pi/n_thetabins) reaches k + 1, found on the HOST with NumPy's own functions (estimate_transform/binning.py):
//          exact for every c, no arccos here;
//   phi    atan2 of the device library, wrapped as the reference does (:32-35), compared with phi_steps[m] (the smallest float64
//          phi whose phi // (2 pi/n_phibins) reaches m + 1, again from NumPy).  The device's atan2 may differ from the host
//          libm's in the last bits, so a neighbour within 2^-46 (16 ulps of 2 pi) of a step is NOT counted here: its index goes
//          to `unsure` and the host decides it with the reference's own NumPy calls (lattice data: neighbours exactly on a
//          sector plane; generic data: none).
// Bin = ring * n_thetabins * n_phibins + theta_index * n_phibins + phi_index, counted if < n_bins (theta = pi and phi = 2 pi spill into
// the next shell or ring exactly as the reference's float index does); NaN coordinates, r_ = 0, |c| > 1 are not counted.
#define PM_TWO_PI 0x1.921fb54442d18p+2          // 2 * np.pi
#define PM_PHI_UNSURE 0x1p-46
__global__ __launch_bounds__(256) void neighbors_binned_kernel(const double *__restrict__ nb, int n, double md,
                                                               const double *__restrict__ r_edges, int n_r,
                                                               const double *__restrict__ cos_steps, int k_t,
                                                               const double *__restrict__ phi_steps, int k_p, int n_t, int n_p,
                                                               long n_bins, int32_t *__restrict__ counts, int32_t *__restrict__ total,
                                                               int32_t *__restrict__ unsure, int32_t *__restrict__ n_unsure) {
    for (long j = (long)blockIdx.x * blockDim.x + threadIdx.x; j < n; j += (long)gridDim.x * blockDim.x) {
        const double vx = nb[3 * j], vy = nb[3 * j + 1], vz = nb[3 * j + 2];
        const double r_ = __builtin_sqrt(__builtin_fma(vz, vz, __builtin_fma(vy, vy, vx * vx)));      // :29 (np.linalg.norm of three numbers: ddot's fused chain)
        const double r = r_ / md;                                              // :30
        const double c = vz / r_;                                              // :31 argument of arccos
        if (!(__builtin_fabs(c) <= 1.0) || vx != vx || vy != vy) continue;     // arccos / arctan2 -> NaN: never counted
        int ring = n_r - 1;                                                    // :49
        for (int k = 0; k < n_r; ++k)
            if (r < r_edges[k]) { ring = k; break; }                           // :53-56
        int th = 0;
        for (int k = 0; k < k_t; ++k) th += (c <= cos_steps[k]);
        double phi = atan2(vy, vx);                                            // :32
        if (phi < 0.0) phi = PM_TWO_PI + phi;                                  // :33
        int ph = 0;
        bool sure = true;
        for (int m = 0; m < k_p; ++m) {
            const double d = phi - phi_steps[m];
            ph += (d >= 0.0);
            sure = sure && (__builtin_fabs(d) > PM_PHI_UNSURE);
        }
        if (!sure) {
            unsure[atomicAdd(n_unsure, 1)] = (int32_t)j;
            continue;
        }
        const long idx = ((long)ring * n_t + th) * n_p + ph;
        if (idx < n_bins) {
            atomicAdd(&counts[idx], 1);
            atomicAdd(total, 1);
        }
    }
}

}  // namespace pm

extern "C" int pm_shape_context_neighbors_binned(const double *nb, int n, double mean_dist, const double *r_edges, int n_rbins,
                                                 const double *cos_steps, int n_cos_steps, const double *phi_steps, int n_phi_steps,
                                                 int n_thetabins, int n_phibins, int32_t *counts, int32_t *total, int32_t *unsure,
                                                 int32_t *n_unsure, void *stream) {
    if (!nb || n <= 0 || !r_edges || n_rbins < 1 || n_thetabins < 1 || n_phibins < 1 || n_cos_steps < 0 || n_phi_steps < 0 ||
        (n_cos_steps > 0 && !cos_steps) || (n_phi_steps > 0 && !phi_steps) || !counts || !total || !unsure || !n_unsure)
        return PM_ERR_INVALID_ARG;
    const long n_bins = (long)n_rbins * n_thetabins * n_phibins;
    if (n_bins > (1L << 24) || n_cos_steps > 4096 || n_phi_steps > 4096 || n_rbins > 4096) return PM_ERR_UNSUPPORTED;
    hipStream_t st = (hipStream_t)stream;
    if (hipMemsetAsync(counts, 0, sizeof(int32_t) * (size_t)n_bins, st) != hipSuccess || hipMemsetAsync(total, 0, sizeof(int32_t), st) != hipSuccess ||
        hipMemsetAsync(n_unsure, 0, sizeof(int32_t), st) != hipSuccess) {
        pm::launch_status();                  // records the HIP error for pm_last_hip_error
        return PM_ERR_LAUNCH;
    }
    const int blocks = (int)std::min<long>(((long)n + 255) / 256, 2048);
    pm::neighbors_binned_kernel<<<blocks, 256, 0, st>>>(nb, n, mean_dist, r_edges, n_rbins, cos_steps, n_cos_steps, phi_steps, n_phi_steps,
                                                        n_thetabins, n_phibins, n_bins, counts, total, unsure, n_unsure);
    return pm::launch_status();
}

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
        pm::shape_context_kernel<2><<<nrows, pm::SC_THREADS, 0, s>>>(xyz, n, row0, centroid3, x0_3, mean_dist1, counts, totals, hist, nrows, nullptr);
    else
        pm::shape_context_kernel<4><<<nrows, pm::SC_THREADS, 0, s>>>(xyz, n, row0, centroid3, x0_3, mean_dist1, counts, totals, hist, nrows, nullptr);
    return pm::launch_status();
}

namespace pm {
struct ScWorkspace {
    ScParams *prm;
    float *frames32;
    double *frames64;
    int32_t *redo;
    unsigned int *cnt1;
    unsigned int *guard;               // [2]
    size_t zero_from, zero_bytes;      // [redo | guard | cnt1]: zeroed before every launch
};
inline size_t sc_ws_layout(int nrows, char *base, ScWorkspace *w) {
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t at = off; off = align_up(off + bytes, 256); return at; };
    const size_t o_prm = take(sizeof(ScParams));
    const size_t o_f32 = take((size_t)nrows * sizeof(ScQuery));
    const size_t o_f64 = take((size_t)nrows * SC_FR * sizeof(double));
    const size_t o_redo = take((size_t)((nrows + SC_Q - 1) / SC_Q) * sizeof(int32_t));
    const size_t o_guard = take(2 * sizeof(unsigned int));
    const size_t o_cnt = take((size_t)nrows * PM_NBINS * sizeof(unsigned int));
    if (w) {
        w->prm = (ScParams *)(base + o_prm);
        w->frames32 = (float *)(base + o_f32);
        w->frames64 = (double *)(base + o_f64);
        w->redo = (int32_t *)(base + o_redo);
        w->cnt1 = (unsigned int *)(base + o_cnt);
        w->guard = (unsigned int *)(base + o_guard);
        w->zero_from = o_redo;
        w->zero_bytes = off - o_redo;
    }
    return off;
}
}  // namespace pm

extern "C" size_t pm_shape_context_workspace(int nrows) {
    return nrows <= 0 ? 0 : pm::sc_ws_layout(nrows, nullptr, nullptr);
}

extern "C" int pm_shape_context_tiled(const double *xyz, int n, int row0, int nrows, const double *centroid3,
                                      const double *x0_3, const double *mean_dist1, int n_frames, int32_t *counts,
                                      int32_t *totals, double *hist, uint32_t *edge_guard2, void *workspace, size_t workspace_bytes,
                                      void *stream) {
    if (!xyz || !centroid3 || !x0_3 || !mean_dist1 || n <= 0 || row0 < 0 || nrows < 0 || row0 + nrows > n)
        return PM_ERR_INVALID_ARG;
    if (n_frames != 2 && n_frames != 4) return PM_ERR_INVALID_ARG;
    if (!counts && !totals && !hist) return PM_ERR_INVALID_ARG;
    if (nrows == 0) return PM_OK;
    if (!workspace || ((uintptr_t)workspace & 255) || workspace_bytes < pm_shape_context_workspace(nrows)) return PM_ERR_WORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    pm::ScWorkspace w;
    pm::sc_ws_layout(nrows, (char *)workspace, &w);
    const int tiles = (nrows + pm::SC_Q - 1) / pm::SC_Q;
    // neighbour segments: about ten work units per resident workgroup (256 CUs x 6), each at least 2 048 neighbours long
    int segs = (10 * 1536 + tiles - 1) / tiles;
    segs = std::max(1, std::min(segs, (n + 2047) / 2048));
    int seg_len = (n + segs - 1) / segs;
    seg_len = (seg_len + pm::SC_THREADS - 1) / pm::SC_THREADS * pm::SC_THREADS;
    segs = (n + seg_len - 1) / seg_len;
    if (hipMemsetAsync((char *)workspace + w.zero_from, 0, w.zero_bytes, s) != hipSuccess) return PM_ERR_LAUNCH;
    pm::sc_prepare_kernel<<<(nrows + 255) / 256, 256, 0, s>>>(xyz, n, row0, nrows, centroid3, x0_3, mean_dist1, w.prm, w.frames32, w.frames64);
    const dim3 grid((unsigned)tiles, (unsigned)segs);
    if (n_frames == 2) {
        pm::sc_tile_kernel<2><<<grid, pm::SC_THREADS, 0, s>>>(xyz, n, row0, nrows, seg_len, w.prm, w.frames32, w.frames64, w.cnt1, w.redo, w.guard);
        pm::sc_finish_kernel<2><<<nrows, 128, 0, s>>>(w.cnt1, nrows, w.redo, counts, totals, hist);
        pm::shape_context_kernel<2><<<nrows, pm::SC_THREADS, 0, s>>>(xyz, n, row0, centroid3, x0_3, mean_dist1, counts, totals, hist, nrows, w.redo);
    } else {
        pm::sc_tile_kernel<4><<<grid, pm::SC_THREADS, 0, s>>>(xyz, n, row0, nrows, seg_len, w.prm, w.frames32, w.frames64, w.cnt1, w.redo, w.guard);
        pm::sc_finish_kernel<4><<<nrows, 128, 0, s>>>(w.cnt1, nrows, w.redo, counts, totals, hist);
        pm::shape_context_kernel<4><<<nrows, pm::SC_THREADS, 0, s>>>(xyz, n, row0, centroid3, x0_3, mean_dist1, counts, totals, hist, nrows, w.redo);
    }
    if (edge_guard2 && hipMemcpyAsync(edge_guard2, w.guard, 2 * sizeof(uint32_t), hipMemcpyDeviceToDevice, s) != hipSuccess) return PM_ERR_LAUNCH;
    return pm::launch_status();
}
