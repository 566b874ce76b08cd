// pm_icp_grid.hip — exact nearest-neighbour search of perform_icp's correspondence step on a uniform grid.
// Reference: perform_icp.py:15-16 — scipy distance_matrix (N x M) + np.argmin(axis=1).
//
// The reference (and pm_icp_nn_brute) look at all N*M pairs.  The fixed cloud does not change during ICP, so it is
// binned ONCE per call into a uniform grid (counting sort: cell index, per-cell counts, exclusive scan, scatter) and
// every iteration then inspects only the cells around each moving point: rings of growing Chebyshev radius r around
// its home cell until the best distance found is smaller than r*h, the least distance any point in a farther ring can
// have (h = cell edge; home cells of points outside the grid are clamped, which only makes unexplored points farther).
// The result is IDENTICAL to the brute-force arg-min — same squared-distance arithmetic ((d0*d0 + d1*d1) + d2*d2,
// d = fixed - moving), same comparison of ROUNDED square roots, ties to the lowest original index — because every
// point that could tie or win lies within the explored radius; candidate order does not matter since ties are broken
// on the stored original index explicitly.  Work per iteration drops from N*M to ~N * (points in ~27 cells).
#include <algorithm>
#include <atomic>
#include "pm_common.h"
#include "pm_solve.h"

namespace pm {

constexpr int GR_MAX_DIM = 512;
// Cells per point.  Nucleus clouds are blobs, not uniform boxes: the bounding box is mostly empty and the core is
// ~30x denser than the box average, so the grid is sized for ~1/8 point per cell on average (~4 in the core).
#ifndef PM_GR_CELLS_PER_POINT
#define PM_GR_CELLS_PER_POINT 8
#endif
constexpr int GR_CELLS_PER_POINT = PM_GR_CELLS_PER_POINT;    // measured on the 50k blob: 1 -> 234, 2 -> 146, 4 -> 97, 8 -> 77, 16 -> 75 us per ICP iteration
constexpr int GR_RING_CAP = 6;            // beyond this ring radius a group scans the whole cloud instead (sparse outliers)
constexpr int GR_BATCH = 4;               // cells per lane whose ranges are fetched together
#ifndef GR_INFLIGHT
#define GR_INFLIGHT 2                     // candidate records a lane requests before it compares them
#endif
#ifndef GR_FLAT
#define GR_FLAT 4                         // candidate records in flight in the flattened scan of the bounded search
#endif
constexpr int GR_BALL_RUNS = 36;          // largest box (in runs of x-adjacent cells) the bounded search of the fused ICP iteration takes on
// Lanes per moving point once the previous match bounds the search (a handful of cells): 8, or 4 for large clouds so that
// all workgroups of a launch are resident at once (measured per iteration at 50k: 16 lanes 43.0 us, 8 -> 30.3, 4 -> 25.8;
// at 20k 8 -> 19.5, 4 -> 20.6; at 5k 8 -> 14.4, 4 -> 15.5).
#ifndef PM_GR_FEW_FROM
#define PM_GR_FEW_FROM 32768
#endif
constexpr int GR_ITER_FEW_LANES_FROM = PM_GR_FEW_FROM;
constexpr int GR_LANES = 32;              // lanes per moving point (measured on the 50k blob: 4 -> 73, 8 -> 67, 16 -> 63, 32 -> 56 us per ICP iteration)

struct GridHeader {          // lives at the start of the workspace, written by grid_plan_kernel
    double lo[3];            // bounding box minimum
    double h;                // cell edge
    double inv_h;
    double reach_scale;      // < 1: slack for the rounding of the point -> cell map (grows with |coordinate| / h)
    int g[3];                // cells per axis
    int ncells;
};

inline int grid_max_cells(int m) {
    long c = (long)m * GR_CELLS_PER_POINT + 1;
    if (c > (1L << 23)) c = 1L << 23;
    return (int)c;
}

struct GridWs {
    size_t header, start, cursor, chunk, pts, total;
};

inline GridWs grid_layout(int m) {
    GridWs w;
    size_t o = 0;
    w.header = o; o += 256;
    w.start = o; o += align_up(((size_t)grid_max_cells(m) + 1) * sizeof(int), 256);
    w.cursor = o; o += align_up(((size_t)grid_max_cells(m) + 1) * sizeof(int), 256);
    w.chunk = o; o += align_up(((size_t)grid_max_cells(m) / 8192 + 2) * sizeof(int), 256);      // one int per scan chunk
    w.pts = o; o += align_up((size_t)m * sizeof(double4), 256);
    w.total = o;
    return w;
}

__device__ __forceinline__ int cell_coord(double x, double lo, double inv_h, int g) {
    const double t = (x - lo) * inv_h;
    int c = (t >= 0.0) ? ((t < (double)g) ? (int)t : g - 1) : 0;     // clamps points outside the box; NaN -> 0
    return c;
}

// one workgroup: bounding box of the fixed cloud, cell edge and grid dimensions (at most max_cells cells)
__global__ __launch_bounds__(1024) void grid_plan_kernel(const double *__restrict__ fix, int m, int max_cells, GridHeader *hd) {
    __shared__ double smin[3][16], smax[3][16];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    {
        // bounding box: the three coordinate rows in ONE sweep, four points per thread and step (twelve independent loads in flight:
        // one coordinate after the other, one load per step, this single workgroup spent ~150 dependent round trips, 40-77 us)
        double lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
        for (int j0 = threadIdx.x; j0 < m; j0 += 4 * 1024) {
            double v[4][3];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int j = min(j0 + u * 1024, m - 1);              // (a clamped repeat changes no minimum or maximum)
#pragma unroll
                for (int c = 0; c < 3; ++c) v[u][c] = fix[(size_t)c * m + j];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    lo[c] = v[u][c] < lo[c] ? v[u][c] : lo[c];
                    hi[c] = v[u][c] > hi[c] ? v[u][c] : hi[c];
                }
        }
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            double a = lo[c], b = hi[c];
            for (int off = 32; off > 0; off >>= 1) {
                const double a2 = __shfl_down(a, off, 64), b2 = __shfl_down(b, off, 64);
                a = a2 < a ? a2 : a;
                b = b2 > b ? b2 : b;
            }
            if (lane == 0) { smin[c][wave] = a; smax[c][wave] = b; }
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        double lo[3], len[3];
        for (int c = 0; c < 3; ++c) {
            double a = smin[c][0], b = smax[c][0];
            for (int w = 1; w < 16; ++w) { a = smin[c][w] < a ? smin[c][w] : a; b = smax[c][w] > b ? smax[c][w] : b; }
            if (!(a <= b)) { a = 0.0; b = 0.0; }                 // NaN-only input: degenerate box
            lo[c] = a;
            len[c] = b - a;
        }
        // cell edge from the volume (area / length for flat clouds) and the target occupancy
        double vol = 1.0;
        int dims = 0;
        for (int c = 0; c < 3; ++c)
            if (len[c] > 0.0) { vol *= len[c]; ++dims; }
        const double target = (double)max_cells - 1.0;
        double h = 1.0;
        if (dims > 0 && target >= 1.0) h = pow(vol / target, 1.0 / dims);
        if (dims > 0 && !(h > 0.0)) h = fmax(fmax(len[0], len[1]), len[2]);
        int g[3];
        for (int it = 0; it < 64; ++it) {
            long prod = 1;
            for (int c = 0; c < 3; ++c) {
                double q = len[c] / h;
                g[c] = (q < (double)(GR_MAX_DIM - 1)) ? (int)q + 1 : GR_MAX_DIM;
                if (g[c] < 1) g[c] = 1;
                prod *= g[c];
            }
            if (prod <= max_cells) break;
            h *= 1.2599210498948732;                             // 2^(1/3): halves the cell count
        }
        if ((long)g[0] * g[1] * g[2] > max_cells) { g[0] = g[1] = g[2] = 1; }
        for (int c = 0; c < 3; ++c) { hd->lo[c] = lo[c]; hd->g[c] = g[c]; }
        hd->h = h;
        hd->inv_h = 1.0 / h;
        // A point's cell coordinate (x - lo) / h carries a rounding error of a few ulps of |x| / h cells; the
        // "unexplored points are farther than r*h" bound is therefore taken with that much slack (plus 1e-6).
        double maxabs = 0.0;
        for (int c = 0; c < 3; ++c) maxabs = fmax(maxabs, fmax(fabs(lo[c]), fabs(lo[c] + len[c])));
        hd->reach_scale = 1.0 - (1e-6 + 16.0 * maxabs * 0x1p-52 / h);
        hd->ncells = g[0] * g[1] * g[2];
    }
}

__device__ __forceinline__ int cell_of(const GridHeader &hd, double x, double y, double z) {
    const int cx = cell_coord(x, hd.lo[0], hd.inv_h, hd.g[0]);
    const int cy = cell_coord(y, hd.lo[1], hd.inv_h, hd.g[1]);
    const int cz = cell_coord(z, hd.lo[2], hd.inv_h, hd.g[2]);
    return (cz * hd.g[1] + cy) * hd.g[0] + cx;                    // x fastest: a run of x-cells is contiguous in memory
}

__global__ __launch_bounds__(256) void grid_count_kernel(const double *__restrict__ fix, int m, const GridHeader *__restrict__ hd,
                                                         int *__restrict__ count) {
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j >= m) return;
    atomicAdd(&count[cell_of(*hd, fix[j], fix[(size_t)m + j], fix[2 * (size_t)m + j])], 1);
}

// exclusive scan of count[0..ncells) into start[0..ncells] (cursor <- start) in three small launches: chunk sums, scan of the
// chunk sums (one workgroup), chunk-local scan plus offset.  Chunk = 1024 threads x SCAN_PER cells.
constexpr int SCAN_PER = 8;
constexpr int SCAN_CHUNK = 1024 * SCAN_PER;

__device__ __forceinline__ int block_exclusive_scan_1024(int v, int *wsum, int *total) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int incl = v;
    for (int off = 1; off < 64; off <<= 1) {
        const int t = __shfl_up(incl, off, 64);
        if (lane >= off) incl += t;
    }
    __syncthreads();
    if (lane == 63) wsum[wave] = incl;
    __syncthreads();
    int prefix = 0, all = 0;
    for (int w = 0; w < 16; ++w) {
        const int t = wsum[w];
        if (w < wave) prefix += t;
        all += t;
    }
    *total = all;
    return prefix + incl - v;
}

__global__ __launch_bounds__(1024) void grid_scan_sums_kernel(const GridHeader *__restrict__ hd, const int *__restrict__ count,
                                                              int *__restrict__ chunk_sum) {
    __shared__ int wsum[16];
    const int n = hd->ncells;
    const int base = blockIdx.x * SCAN_CHUNK + threadIdx.x * SCAN_PER;
    int t = 0;
    if (blockIdx.x * SCAN_CHUNK < n) {
#pragma unroll
        for (int q = 0; q < SCAN_PER; ++q) t += (base + q < n) ? count[base + q] : 0;
    }
    int all;
    block_exclusive_scan_1024(t, wsum, &all);
    if (threadIdx.x == 0) chunk_sum[blockIdx.x] = all;
}

__global__ __launch_bounds__(1024) void grid_scan_offsets_kernel(int *__restrict__ chunk_sum, int nchunks) {
    __shared__ int wsum[16];
    __shared__ int carry;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (int base = 0; base < nchunks; base += 1024) {
        const int i = base + threadIdx.x;
        const int v = (i < nchunks) ? chunk_sum[i] : 0;
        int all;
        const int excl = block_exclusive_scan_1024(v, wsum, &all);
        const int c = carry;
        if (i < nchunks) chunk_sum[i] = c + excl;
        __syncthreads();
        if (threadIdx.x == 0) carry = c + all;
        __syncthreads();
    }
}

__global__ __launch_bounds__(1024) void grid_scan_apply_kernel(const GridHeader *__restrict__ hd, const int *__restrict__ chunk_off,
                                                               int *__restrict__ start, int *__restrict__ cursor) {
    __shared__ int wsum[16];
    const int n = hd->ncells;
    if (blockIdx.x * SCAN_CHUNK > n) return;                 // (the chunk holding index n writes start[n])
    const int base = blockIdx.x * SCAN_CHUNK + threadIdx.x * SCAN_PER;
    int c[SCAN_PER], t = 0;
#pragma unroll
    for (int q = 0; q < SCAN_PER; ++q) { c[q] = (base + q < n) ? start[base + q] : 0; t += c[q]; }   // counts were accumulated in `start`
    int all;
    int run = chunk_off[blockIdx.x] + block_exclusive_scan_1024(t, wsum, &all);
#pragma unroll
    for (int q = 0; q < SCAN_PER; ++q) {
        if (base + q <= n) {
            start[base + q] = run;                           // index n receives the grand total
            if (base + q < n) cursor[base + q] = run;
        }
        run += c[q];
    }
}

__global__ __launch_bounds__(256) void grid_scatter_kernel(const double *__restrict__ fix, int m, const GridHeader *__restrict__ hd,
                                                           int *__restrict__ cursor, double4 *__restrict__ pts) {
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j >= m) return;
    const double x = fix[j], y = fix[(size_t)m + j], z = fix[2 * (size_t)m + j];
    const int pos = atomicAdd(&cursor[cell_of(*hd, x, y, z)], 1);
    double4 v;
    v.x = x; v.y = y; v.z = z;
    v.w = __longlong_as_double((long long)j);                    // original index rides along (exact bit pattern)
    pts[pos] = v;
}

// exact "a is a better match than b": smaller rounded root, then smaller original index
__device__ __forceinline__ bool grid_better(double aS, int aI, double bS, int bI) {
    if (aS < bS * 0x1.ffffffffffffp-1) return true;               // clearly smaller (roots >= 2 ulp apart)
    if (!(aS <= bS * 0x1.000000000001p+0)) return false;          // clearly larger, or NaN
    if (aS == bS) return aI < bI;
    const double ra = __builtin_sqrt(aS), rb = __builtin_sqrt(bS);
    return (ra < rb) || (ra == rb && aI < bI);
}

// L lanes (a power of two <= 64) cooperate on one moving point: lane l takes the cells l, l + L, ... of the current pass
// (the 3 x 3 x 3 cube around the home cell first, then shells of growing Chebyshev radius), GR_BATCH cells at a time —
// the cell ranges of a batch are fetched together (independent loads, one round trip), then their points are compared —
// and the group's candidates are merged with the exact comparator.
// GridSearch holds one lane's view of one query; `bS`/`bI` may be pre-loaded with a known candidate (exact: a candidate
// offered twice changes nothing).
template <int L, bool TRACK = false>
struct GridSearch {
    const GridHeader &hd;
    const int *__restrict__ start;
    const double4 *__restrict__ pts;
    const double p0, p1, p2;
    const int sub;
    double bS;
    int bI;
    double bX, bY, bZ;               // TRACK: coordinates of the best candidate (what the caller would otherwise gather again)

    __device__ __forceinline__ GridSearch(const GridHeader &h, const int *st, const double4 *pt, double x, double y, double z, int s)
        : hd(h), start(st), pts(pt), p0(x), p1(y), p2(z), sub(s), bS(INFINITY), bI(0x7fffffff), bX(0.0), bY(0.0), bZ(0.0) {}
    __device__ __forceinline__ GridSearch(const GridHeader &h, const int *st, const double4 *pt, double x, double y, double z, int s,
                                          double S, int I, double X, double Y, double Z)
        : hd(h), start(st), pts(pt), p0(x), p1(y), p2(z), sub(s), bS(S), bI(I), bX(X), bY(Y), bZ(Z) {}
    // bI = 0x7fffffff is "nothing yet": loses every index tie; replaced by 0 at the end if nothing ever matched

    // candidates q0, q0 + step, ... < q1, GR_INFLIGHT loads in flight at a time
    __device__ __forceinline__ void scan(int q0, int q1, int step) {
        for (int q = q0; q < q1; q += GR_INFLIGHT * step) {
            double4 f[GR_INFLIGHT];
#pragma unroll
            for (int u = 0; u < GR_INFLIGHT; ++u) f[u] = pts[(q + u * step < q1) ? q + u * step : q];
#pragma unroll
            for (int u = 0; u < GR_INFLIGHT; ++u) {
                const double d0 = f[u].x - p0, d1 = f[u].y - p1, d2 = f[u].z - p2;
                const double s = (d0 * d0 + d1 * d1) + d2 * d2;
                const int j = (int)__double_as_longlong(f[u].w);
                if (grid_better(s, j, bS, bI)) {                       // a clamped repeat of the last candidate changes nothing
                    bS = s; bI = j;
                    if (TRACK) { bX = f[u].x; bY = f[u].y; bZ = f[u].z; }
                }
            }
        }
    }

    // cells of the cube of radius r around the home cell; shell_only: those at Chebyshev distance exactly r
    __device__ __forceinline__ void pass(int cx, int cy, int cz, int r, bool shell_only) {
        const int w = 2 * r + 1, cube = w * w * w;
        for (int k0 = sub; k0 < cube; k0 += L * GR_BATCH) {
            int q0[GR_BATCH], q1[GR_BATCH];
#pragma unroll
            for (int u = 0; u < GR_BATCH; ++u) {
                const int k = k0 + u * L;
                const int dz = k / (w * w) - r, rem = k % (w * w), dy = rem / w - r, dx = rem % w - r;
                const int x = cx + dx, y = cy + dy, z = cz + dz;
                const bool take = k < cube && (!shell_only || max(max(abs(dx), abs(dy)), abs(dz)) == r)
                                  && x >= 0 && y >= 0 && z >= 0 && x < hd.g[0] && y < hd.g[1] && z < hd.g[2];
                const int c = take ? (z * hd.g[1] + y) * hd.g[0] + x : 0;
                q0[u] = start[c];
                q1[u] = take ? start[c + 1] : q0[u];
            }
#pragma unroll
            for (int u = 0; u < GR_BATCH; ++u) scan(q0[u], q1[u], 1);
        }
    }

    __device__ __forceinline__ void merge() {          // butterfly over the group: afterwards every lane holds the winner
#pragma unroll
        for (int off = L / 2; off > 0; off >>= 1) {
            const double oS = __shfl_xor(bS, off, 64);
            const int oI = __shfl_xor(bI, off, 64);
            const bool take = grid_better(oS, oI, bS, bI);
            if (TRACK) {
                const double oX = __shfl_xor(bX, off, 64), oY = __shfl_xor(bY, off, 64), oZ = __shfl_xor(bZ, off, 64);
                if (take) { bX = oX; bY = oY; bZ = oZ; }
            }
            if (take) { bS = oS; bI = oI; }
        }
    }

    // The exact nearest neighbour by rings of cells around the home cell (every lane of the group ends with the winner).
    __device__ __forceinline__ void rings() {
        const int cx = cell_coord(p0, hd.lo[0], hd.inv_h, hd.g[0]);
        const int cy = cell_coord(p1, hd.lo[1], hd.inv_h, hd.g[1]);
        const int cz = cell_coord(p2, hd.lo[2], hd.inv_h, hd.g[2]);
        const int rmax = max(max(max(cx, hd.g[0] - 1 - cx), max(cy, hd.g[1] - 1 - cy)), max(cz, hd.g[2] - 1 - cz));
        // rings 0 and 1 together, the 3 x 3 x 3 cube around the home cell.  Cells adjacent in x are adjacent in memory, and so
        // are their points: the cube is nine runs of (up to) three cells.  Three lanes share a run — the same two range
        // loads, then interleaved points, so that neighbouring lanes touch neighbouring 32-byte records.
        if (L >= 32) {
            if (sub < 27) {
                const int run = sub / 3, part = sub - 3 * run;
                const int y = cy + run % 3 - 1, z = cz + run / 3 - 1;
                if (y >= 0 && z >= 0 && y < hd.g[1] && z < hd.g[2]) {
                    const int row = (z * hd.g[1] + y) * hd.g[0];
                    scan(start[row + max(cx - 1, 0)] + part, start[row + min(cx + 1, hd.g[0] - 1) + 1], 3);
                }
            }
        } else {
            pass(cx, cy, cz, 1, false);
        }
        merge();
        int r = 1;
        // everything not yet visited is farther than r*h (minus the rounding slack of the cell map)
        // (reach_scale <= 0 for absurd coordinate / cell-size ratios simply widens the search to the full scan)
        while (!(hd.reach_scale > 0.0 && bS < (r * hd.h * hd.reach_scale) * (r * hd.h * hd.reach_scale)) && r < rmax) {
            ++r;
            if (r > GR_RING_CAP) {                                  // sparse neighbourhood: scan every point (always exact)
                const int m_all = start[hd.ncells];
                for (int q = sub; q < m_all; q += L) {
                    const double4 f = pts[q];
                    const double d0 = f.x - p0, d1 = f.y - p1, d2 = f.z - p2;
                    const double s = (d0 * d0 + d1 * d1) + d2 * d2;
                    const int j = (int)__double_as_longlong(f.w);
                    if (grid_better(s, j, bS, bI)) {
                        bS = s; bI = j;
                        if (TRACK) { bX = f.x; bY = f.y; bZ = f.z; }
                    }
                }
                merge();
                break;
            }
            pass(cx, cy, cz, r, true);
            merge();
        }
    }

    // candidates of up to three index ranges [a_b, e_b) (stride `step` each) as ONE sequence, GR_FLAT records in flight at a time: a
    // lane that owns several runs of cells pays one round trip per GR_FLAT candidates instead of one (or more) per run
    __device__ __forceinline__ void scan3(int a0, int e0, int a1, int e1, int a2, int e2, int lg) {     // stride 1 << lg
        const int step = 1 << lg;
        const int c0 = max(0, (e0 - a0 + step - 1) >> lg), c1 = max(0, (e1 - a1 + step - 1) >> lg), c2 = max(0, (e2 - a2 + step - 1) >> lg);
        const int total = c0 + c1 + c2;
        for (int t = 0; t < total; t += GR_FLAT) {
            double4 f[GR_FLAT];
#pragma unroll
            for (int u = 0; u < GR_FLAT; ++u) {
                const int tt = min(t + u, total - 1);                 // a clamped repeat of the last candidate changes nothing
                const int q = tt < c0 ? a0 + (tt << lg) : (tt < c0 + c1 ? a1 + ((tt - c0) << lg) : a2 + ((tt - c0 - c1) << lg));
                f[u] = pts[q];
            }
#pragma unroll
            for (int u = 0; u < GR_FLAT; ++u) {
                const double d0 = f[u].x - p0, d1 = f[u].y - p1, d2 = f[u].z - p2;
                const double s = (d0 * d0 + d1 * d1) + d2 * d2;
                const int j = (int)__double_as_longlong(f[u].w);
                if (grid_better(s, j, bS, bI)) {
                    bS = s; bI = j;
                    if (TRACK) { bX = f[u].x; bY = f[u].y; bZ = f[u].z; }
                }
            }
        }
    }

    // The same answer when a candidate at squared distance bS is already known (ICP: the previous iteration's match): every
    // point that could beat or tie it lies within sqrt(bS) of the query, hence in the cells the box query +- rad covers
    // (the point -> cell map is monotone per axis, so no slack beyond rad >= the true distance is needed).  The box is
    // ny * nz runs of x-adjacent cells; lanes share the runs (several lanes per run while there are fewer runs than lanes).
    // A lane's runs are taken three at a time: their six range ends in one round trip, then their candidates as one sequence
    // (round 3: the slowest point of a launch — a large box, nine runs per lane — set the pace of every iteration).
    // Returns false (nothing scanned) if the box has more than GR_BALL_RUNS runs: rings() is the better plan then.
    __device__ __forceinline__ bool ball(double rad) {
        const int x0 = cell_coord(p0 - rad, hd.lo[0], hd.inv_h, hd.g[0]), x1 = cell_coord(p0 + rad, hd.lo[0], hd.inv_h, hd.g[0]);
        const int y0 = cell_coord(p1 - rad, hd.lo[1], hd.inv_h, hd.g[1]), y1 = cell_coord(p1 + rad, hd.lo[1], hd.inv_h, hd.g[1]);
        const int z0 = cell_coord(p2 - rad, hd.lo[2], hd.inv_h, hd.g[2]), z1 = cell_coord(p2 + rad, hd.lo[2], hd.inv_h, hd.g[2]);
        const int ny = y1 - y0 + 1, nruns = ny * (z1 - z0 + 1);
        if (nruns > GR_BALL_RUNS || nruns < 1) return false;
        // lanes per run: the largest power of two <= L / nruns (no integer division in this loop; counters of the 50 000-point
        // launch, profiles/r03_icp_iter_pmc.txt: ~870 vector instructions per wave, waves waiting 53 % of their cycles, issue
        // stalls 24 %, the VALU busy 21 % of the launch — neither more loads in flight, nor more lanes per point, nor a finer
        // grid moved the 6 us this phase takes: profiles/r03_icp_stamps_per_launch_variants.txt, r03_icp_grid_resolution_sweep.txt)
        int lg = 0;
        while ((2 << lg) * nruns <= L) ++lg;
        const int part = sub & ((1 << lg) - 1), stride = L >> lg;
        const unsigned int inv_ny = (65536u + (unsigned int)ny - 1u) / (unsigned int)ny;      // floor(r / ny) = (r * inv_ny) >> 16 for r < 64
        for (int run = sub >> lg; run < nruns; run += 3 * stride) {
            int a[3], e[3];
#pragma unroll
            for (int b = 0; b < 3; ++b) {
                const int r = run + b * stride;
                const bool have = r < nruns;
                const int rr = have ? r : run;
                const int rz = (int)(((unsigned int)rr * inv_ny) >> 16), ry = rr - rz * ny;
                const int row = ((z0 + rz) * hd.g[1] + (y0 + ry)) * hd.g[0];
                a[b] = start[row + x0] + part;
                e[b] = have ? start[row + x1 + 1] : a[b];           // (an absent run: empty range)
            }
            scan3(a[0], e[0], a[1], e[1], a[2], e[2], lg);
        }
        merge();
        return true;
    }
};

template <int L>
__global__ __launch_bounds__(256) void grid_nn_kernel(const double *__restrict__ mov, int n, const GridHeader *__restrict__ hdp,
                                                      const int *__restrict__ start, const double4 *__restrict__ pts,
                                                      int32_t *__restrict__ nn, double *__restrict__ dist) {
    const int sub = threadIdx.x & (L - 1);
    const int i = (int)(((long)blockIdx.x * 256 + threadIdx.x) / L);
    const int ic = min(i, n - 1);                                 // surplus groups shadow the last point (no divergent exit before shuffles)
    const GridHeader hd = *hdp;
    GridSearch<L> q(hd, start, pts, mov[ic], mov[(size_t)n + ic], mov[2 * (size_t)n + ic], sub);
    q.rings();
    if (sub == 0 && i < n) {
        nn[i] = (q.bI == 0x7fffffff) ? 0 : q.bI;                 // all-NaN row: np.argmin answers 0
        if (dist) dist[i] = __builtin_sqrt(q.bS);
    }
}

// ---- one whole ICP iteration in one launch (perform_icp.py:14-25) -----------------------------------------------------------
// Workgroup = 8 moving points x 32 lanes = one LEAF of the reduction tree (pm_solve.h).  Per point:
//   (not FIRST) apply the previous iteration's A_est (:23), store the moved point, residual against the previous match (:24);
//   nearest fixed point (:15-16): the previous match bounds the search (GridSearch::ball), rings otherwise;
//   the 22 moment terms of (point, match) (:18's least squares as normal equations) and the residual term -> leaf partial.
// The workgroup that completes a group of 64 leaves adds them in leaf order; the one that completes the last group adds the
// groups in order, solves the 4 x 4 (A_est for the next launch), composes A_icp (:25) and writes the mean residual — the
// "last one out" pattern: a counter per group, no workgroup ever waits for another (nothing can hang).
constexpr int IT_SLOTS = PM_NMOMENTS + 1;          // 22 moments + residual
constexpr int IT_STRIDE = 24;
constexpr int IT_FINAL_CHUNK = 2 * PM_TREE_GROUP;   // group partials the last workgroup fetches in one round trip (128: clouds up to 65 536 points)

struct IterArgs {
    double *mov; int n;
    const GridHeader *hd; const int *start; const double4 *pts;
    const double *fix; int m;
    const int32_t *nn_prev; int32_t *nn_out;
    const double *A_prev;            // 16: the transform fitted by the previous launch (unused if FIRST)
    const double *origin6;
    double *leaf_partial;            // [leaves][IT_STRIDE]
    double *group_partial;           // [groups][IT_STRIDE]
    unsigned int *counters;          // [1 + groups], zero on entry, zero again on exit
    double *A_est;                   // 16 out
    double *A_icp;                   // 16 in/out
    double *res_prev_out;            // mean residual of the previous iteration (may be null)
    int32_t *status;
    int leaves, groups;
};

__device__ __forceinline__ double coherent_load(const double *p) {
    return __longlong_as_double((long long)__hip_atomic_load((const unsigned long long *)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
}
__device__ __forceinline__ void coherent_store(double *p, double v) {
    __hip_atomic_store((unsigned long long *)p, (unsigned long long)__double_as_longlong(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Diagnostic build only (-DPM_ICP_STAMPS, tools/icp_stamps.py): lane 0 of every workgroup stamps the constant 100 MHz clock at
// the phase boundaries of the iteration into a buffer no other code reads.  The product build contains none of this.
#ifdef PM_ICP_STAMPS
__device__ unsigned long long *g_icp_stamps = nullptr;       // [workgroups][8]
#define PM_STAMP(k) do { if (g_icp_stamps && threadIdx.x == 0) g_icp_stamps[(size_t)blockIdx.x * 8 + (k)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define PM_STAMP(k) do { } while (0)
#endif

template <int L, bool FIRST>
__global__ __launch_bounds__(256) void icp_iter_kernel(const IterArgs a) {
    constexpr int PTS = 256 / L;                        // moving points per workgroup
    constexpr int LPB = PTS / PM_TREE_LEAF;             // leaves of the reduction tree per workgroup
    static_assert(PTS % PM_TREE_LEAF == 0 && LPB >= 1 && PM_TREE_GROUP % LPB == 0, "a workgroup must hold whole leaves of one group");
    __shared__ double term[PTS][IT_STRIDE];
    __shared__ double s_stage[IT_FINAL_CHUNK * IT_STRIDE];      // 24 KB: the partials of one level, fetched by all threads at once
    __shared__ double totals[IT_STRIDE + 1];
    __shared__ int s_flag;
    const int tid = threadIdx.x, sub = tid & (L - 1), slot = tid / L;
    const int n = a.n, m = a.m;
    PM_STAMP(0);
    const int i = blockIdx.x * PTS + slot;
    const int ic = min(i, n - 1);
    const GridHeader hd = *a.hd;
    double p0 = a.mov[ic], p1 = a.mov[(size_t)n + ic], p2 = a.mov[2 * (size_t)n + ic];
    double res_prev = 0.0, S0 = INFINITY;
    double iS = INFINITY, iX = 0.0, iY = 0.0, iZ = 0.0;         // the candidate the search starts from: the previous match
    int iI = 0x7fffffff;
    if (!FIRST) {
        const double x = p0, y = p1, z = p2;
        p0 = affine_row(a.A_prev, x, y, z);
        p1 = affine_row(a.A_prev + 4, x, y, z);
        p2 = affine_row(a.A_prev + 8, x, y, z);
        const int j_prev = a.nn_prev[ic];
        const double f0 = a.fix[j_prev], f1 = a.fix[(size_t)m + j_prev], f2 = a.fix[2 * (size_t)m + j_prev];
        const double d0 = p0 - f0, d1 = p1 - f1, d2 = p2 - f2;
        S0 = (d0 * d0 + d1 * d1) + d2 * d2;              // == the search's (f - p) form: negation is exact
        res_prev = __builtin_sqrt(S0);
        if (sub == 0 && i < n) { a.mov[i] = p0; a.mov[(size_t)n + i] = p1; a.mov[2 * (size_t)n + i] = p2; }
        if (S0 < INFINITY) { iS = S0; iI = j_prev; iX = f0; iY = f1; iZ = f2; }     // (NaN fails the comparison)
    }
#ifdef PM_ICP_STAMPS
    if (__builtin_isnan(S0 + p0)) __builtin_amdgcn_s_sleep(1);      // (forces the loads above to have landed before the stamp)
#endif
    PM_STAMP(1);
    GridSearch<L, true> q(hd, a.start, a.pts, p0, p1, p2, sub, iS, iI, iX, iY, iZ);
    bool done = false;
    if (!FIRST && S0 < INFINITY) done = q.ball(res_prev * (1.0 + 0x1p-20) + 0x1p-1000);
    if (!done) q.rings();
#ifdef PM_ICP_STAMPS
    if (q.bI == -5) __builtin_amdgcn_s_sleep(1);
#endif
    PM_STAMP(2);
    if (sub == 0) {
        double s[PM_NMOMENTS];
        if (i < n) {
            const bool none = q.bI == 0x7fffffff;        // all-NaN row: np.argmin answers 0
            const int j = none ? 0 : q.bI;
            a.nn_out[i] = j;
            const double f0 = none ? a.fix[0] : q.bX, f1 = none ? a.fix[(size_t)m] : q.bY, f2 = none ? a.fix[2 * (size_t)m] : q.bZ;
            moment_terms(p0 - a.origin6[0], p1 - a.origin6[1], p2 - a.origin6[2], f0 - a.origin6[3], f1 - a.origin6[4], f2 - a.origin6[5], s);
        } else {
#pragma unroll
            for (int k = 0; k < PM_NMOMENTS; ++k) s[k] = 0.0;
        }
#pragma unroll
        for (int k = 0; k < PM_NMOMENTS; ++k) term[slot][k] = s[k];
        term[slot][PM_NMOMENTS] = (i < n) ? res_prev : 0.0;
    }
    __syncthreads();
    // Hand-off discipline (MI355X_MICROARCH.md, "Valid forms": write-through stores, drained, one lane signals behind the
    // workgroup barrier; the workgroup whose add came last reads with L1-bypassing loads — no fence, no waiting):
    const int leaf0 = blockIdx.x * LPB;
    const int mine = min(LPB, a.leaves - leaf0);        // leaves of this workgroup that exist
    for (int e = tid; e < LPB * IT_SLOTS; e += 256) {
        const int lf = e / IT_SLOTS, k = e - lf * IT_SLOTS;
        if (lf < mine) {
            double acc = 0.0;
#pragma unroll
            for (int pnt = 0; pnt < PM_TREE_LEAF; ++pnt) acc += term[lf * PM_TREE_LEAF + pnt][k];
            coherent_store(&a.leaf_partial[(size_t)(leaf0 + lf) * IT_STRIDE + k], acc);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    PM_STAMP(3);
    // ---- last one out: leaves -> group -> total -> solve
    const int g = leaf0 / PM_TREE_GROUP;
    __syncthreads();
    if (tid == 0) {
        const unsigned int size = (unsigned int)min(PM_TREE_GROUP, a.leaves - g * PM_TREE_GROUP);
        s_flag = (atomicAdd(&a.counters[1 + g], (unsigned int)mine) + (unsigned int)mine == size) ? 1 : 0;
    }
    __syncthreads();
    PM_STAMP(4);
    if (!s_flag) return;
    // The workgroup whose add completed group g adds its (up to) 64 leaves in leaf order.  Round 3: ALL 256 threads fetch the
    // partials — every load of the level in flight at once, ONE memory round trip (these are L1-bypassing loads of lines
    // other XCDs wrote through: ~0.7 us each; 23 lanes fetching eight at a time took eight round trips here and thirteen
    // for the 98 groups of a 50 000-point cloud) — into LDS, then 23 lanes add in the fixed order.
    {
        const int size = min(PM_TREE_GROUP, a.leaves - g * PM_TREE_GROUP);
        const double *lp = a.leaf_partial + (size_t)g * PM_TREE_GROUP * IT_STRIDE;
        const int total = size * IT_STRIDE;
        constexpr int PER = (PM_TREE_GROUP * IT_STRIDE + 255) / 256;       // 6
        double v[PER];
#pragma unroll
        for (int u = 0; u < PER; ++u) v[u] = coherent_load(lp + min(tid + u * 256, total - 1));
#pragma unroll
        for (int u = 0; u < PER; ++u)
            if (tid + u * 256 < total) s_stage[tid + u * 256] = v[u];
        __syncthreads();
        if (tid < IT_SLOTS) {
            double acc = 0.0;
            for (int b = 0; b < size; ++b) acc += s_stage[b * IT_STRIDE + tid];
            coherent_store(&a.group_partial[(size_t)g * IT_STRIDE + tid], acc);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    PM_STAMP(5);
    __syncthreads();
    if (tid == 0) s_flag = (atomicAdd(&a.counters[0], 1u) == (unsigned int)a.groups - 1u) ? 1 : 0;
    __syncthreads();
    PM_STAMP(6);
    if (!s_flag) return;
    // ... and the one that completed the last group adds the groups in group order, 64 at a time through the same staging area
    {
        double acc = 0.0;
        for (int g0 = 0; g0 < a.groups; g0 += IT_FINAL_CHUNK) {
            const int size = min(IT_FINAL_CHUNK, a.groups - g0);
            const double *gp = a.group_partial + (size_t)g0 * IT_STRIDE;
            const int total = size * IT_STRIDE;
            constexpr int PER = (IT_FINAL_CHUNK * IT_STRIDE + 255) / 256;  // 12
            double v[PER];
#pragma unroll
            for (int u = 0; u < PER; ++u) v[u] = coherent_load(gp + min(tid + u * 256, total - 1));
            __syncthreads();                             // (the previous chunk has been added up)
#pragma unroll
            for (int u = 0; u < PER; ++u)
                if (tid + u * 256 < total) s_stage[tid + u * 256] = v[u];
            __syncthreads();
            if (tid < IT_SLOTS)
                for (int b = 0; b < size; ++b) acc += s_stage[b * IT_STRIDE + tid];
        }
        if (tid < IT_SLOTS) totals[1 + tid] = acc;
    }
    for (int c = tid; c < 1 + a.groups; c += 256)                            // everyone has passed: ready for the next launch
        __hip_atomic_store(&a.counters[c], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    if (tid == 0) {
        totals[0] = (double)n;
        double sums[PM_ICP_NSUMS];
        for (int k = 0; k < PM_ICP_NSUMS - 1; ++k) sums[k] = totals[k];
        sums[PM_ICP_NSUMS - 1] = 0.0;
        double A[16];
        const double ratio = affine_from_sums(sums, a.origin6, A);
        if (a.status && !(ratio > PM_DEGENERATE_MOMENTS)) a.status[0] = 1;
        for (int k = 0; k < 16; ++k) a.A_est[k] = A[k];
        compose_affine(A, a.A_icp);
        if (!FIRST && a.res_prev_out) a.res_prev_out[0] = totals[1 + PM_NMOMENTS] / (double)n;
    }
    PM_STAMP(7);
}

#ifdef PM_ICP_STAMPS
extern "C" int pm_debug_set_icp_stamps(unsigned long long *buf) {
    return hipMemcpyToSymbol(HIP_SYMBOL(g_icp_stamps), &buf, sizeof(buf)) == hipSuccess ? 0 : -1;
}
#endif


// ---- the whole loop in one launch (round 3) -------------------------------------------------------------------------------
// icp_iter_kernel pays, per iteration, for a kernel boundary (which also empties every XCD's L2: the grid tables and the fixed
// records come back from the Infinity Cache each time), for reloading the moving points and for gathering the previous
// matches.  icp_loop_kernel runs iterations first_it .. iters-1 in ONE launch: a workgroup keeps its points, their matches
// and the current 4 x 4 in registers / LDS for the whole loop, the read-only search tables stay warm in L2, and the only
// inter-workgroup traffic per iteration is the reduction tree of icp_iter_kernel (same leaves, groups and order: same bits)
// plus the publication of the fitted transform:
//   every workgroup   leaf partials (write-through) -> drain -> add `mine` to its group's counter;
//   last of a group   fetch the group's leaves (all threads, one round trip) -> sum in leaf order -> store -> add to counter 0;
//   last of all       fetch the groups -> sum in group order -> solve -> A_est, A_icp, residual (write-through) -> drain ->
//                     generation word = round;
//   every workgroup   one lane polls the generation word (L1-bypassing load + s_sleep), barrier, all reload A_est (bypassing).
// Counters are never reset: in round r a group's counter ends at r x (its leaves), counter 0 at r x groups.  The grid must be
// co-resident (icp_loop_capacity, with a margin of one workgroup per CU) and the caller must not have another such launch
// in flight on the device (two half-resident persistent grids can starve each other); a wait longer than spin_ticks (2 s) sets
// status 2 and leaves the loop — every workgroup then does — so the grid always drains.
#ifndef PM_ICP_POLL_SLEEP
#define PM_ICP_POLL_SLEEP 16
#endif
struct LoopArgs {
    IterArgs a;
    int first_it, iters;
    double *residuals;               // [iters] (may be null): slot it-1 is written by iteration it
    int32_t *nn_all;                 // [iters][n] (may be null)
    int32_t *nn_last;                // [n]: the matches of the last iteration (for the trailing update kernel)
    unsigned int *gen;               // generation word, zero on entry
    unsigned long long spin_ticks;   // how long a workgroup waits for a round (100 MHz ticks) before it reports status 2 and leaves
};

// The rare half of a round, kept out of line so that its registers (the staged fetches, the solve) do not weigh on the loop
// every workgroup runs: called by the workgroup that completed group g; the one that also completes the last group
// publishes the round.  s_stage / totals / flags: the caller's LDS.
__device__ __forceinline__ void loop_round_tail(const LoopArgs &la, int g, unsigned int gsize, unsigned int round, int it,
                                             double *s_stage, double *totals, int *s_last_of_all, double *C_s) {
    const IterArgs &a = la.a;
    const int tid = threadIdx.x, n = a.n;
    {
            const double *lp = a.leaf_partial + (size_t)g * PM_TREE_GROUP * IT_STRIDE;
            const int total = (int)gsize * IT_STRIDE;
            constexpr int PER = (PM_TREE_GROUP * IT_STRIDE + 255) / 256;
            double v[PER];
#pragma unroll
            for (int u = 0; u < PER; ++u) v[u] = coherent_load(lp + min(tid + u * 256, total - 1));
#pragma unroll
            for (int u = 0; u < PER; ++u)
                if (tid + u * 256 < total) s_stage[tid + u * 256] = v[u];
            __syncthreads();
            if (tid < IT_SLOTS) {
                double acc = 0.0;
                for (int b = 0; b < (int)gsize; ++b) acc += s_stage[b * IT_STRIDE + tid];
                coherent_store(&a.group_partial[(size_t)g * IT_STRIDE + tid], acc);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (tid == 0) *s_last_of_all = (atomicAdd(&a.counters[0], 1u) + 1u == (unsigned int)a.groups * round) ? 1 : 0;
            __syncthreads();
            if (*s_last_of_all) {                         // ... and the last group: total, solve, publish
                double acc = 0.0;
                for (int g0 = 0; g0 < a.groups; g0 += IT_FINAL_CHUNK) {
                    const int size = min(IT_FINAL_CHUNK, a.groups - g0);
                    const double *gp = a.group_partial + (size_t)g0 * IT_STRIDE;
                    const int tot = size * IT_STRIDE;
                    constexpr int PERF = (IT_FINAL_CHUNK * IT_STRIDE + 255) / 256;
                    double w[PERF];
#pragma unroll
                    for (int u = 0; u < PERF; ++u) w[u] = coherent_load(gp + min(tid + u * 256, tot - 1));
                    __syncthreads();
#pragma unroll
                    for (int u = 0; u < PERF; ++u)
                        if (tid + u * 256 < tot) s_stage[tid + u * 256] = w[u];
                    __syncthreads();
                    if (tid < IT_SLOTS)
                        for (int b = 0; b < size; ++b) acc += s_stage[b * IT_STRIDE + tid];
                }
                if (tid < IT_SLOTS) totals[1 + tid] = acc;
                // A_icp was composed by another workgroup last round: sixteen lanes fetch it at once (bypassing loads issued one
                // after the other by a single lane would be sixteen round trips), one lane solves and composes in LDS, sixteen
                // lanes publish
                // C_s: [0..16) A_icp, [16..32) A_est — the caller's term table, idle since the leaf sums (s_stage is still being read)
                if (tid >= 64 && tid < 80) C_s[tid - 64] = coherent_load(&a.A_icp[tid - 64]);
                __syncthreads();
                if (tid == 0) {
                    totals[0] = (double)n;
                    double sums[PM_ICP_NSUMS];
                    for (int k = 0; k < PM_ICP_NSUMS - 1; ++k) sums[k] = totals[k];
                    sums[PM_ICP_NSUMS - 1] = 0.0;
                    double A[16], C[16];
                    const double ratio = affine_from_sums(sums, a.origin6, A);
                    if (a.status && !(ratio > PM_DEGENERATE_MOMENTS)) __hip_atomic_store(a.status, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    for (int k = 0; k < 16; ++k) C[k] = C_s[k];
                    compose_affine(A, C);
                    for (int k = 0; k < 16; ++k) { C_s[k] = C[k]; C_s[16 + k] = A[k]; }
                    if (la.residuals) la.residuals[it - 1] = totals[1 + PM_NMOMENTS] / (double)n;
                }
                __syncthreads();
                if (tid < 16) coherent_store(&a.A_icp[tid], C_s[tid]);
                else if (tid < 32) coherent_store(&a.A_est[tid - 16], C_s[tid]);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // (wave 0 stored everything: its drain covers the publication)
                if (tid == 0) __hip_atomic_store(la.gen, round, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
    }
}

template <int L>
__global__ __launch_bounds__(256, 4) void icp_loop_kernel(const LoopArgs la) {      // 4 workgroups per CU: <= 128 VGPRs
    constexpr int PTS = 256 / L;
    constexpr int LPB = PTS / PM_TREE_LEAF;
    static_assert(PTS % PM_TREE_LEAF == 0 && LPB >= 1 && PM_TREE_GROUP % LPB == 0, "a workgroup must hold whole leaves of one group");
    __shared__ double term[PTS][IT_STRIDE];
    __shared__ double s_stage[IT_FINAL_CHUNK * IT_STRIDE];
    __shared__ double totals[IT_STRIDE + 1];
    __shared__ double A_s[16];
    __shared__ int s_last_of_group, s_last_of_all, s_ok;   // three words: each is written once per round, between barriers that
                                                           // every reader has passed (a shared flag reused across uniform branches would race)
    const IterArgs &a = la.a;
    const int tid = threadIdx.x, sub = tid & (L - 1), slot = tid / L;
    const int n = a.n, m = a.m;
    const int i = blockIdx.x * PTS + slot;
    const int ic = min(i, n - 1);
    const GridHeader hd = *a.hd;
    double p0 = a.mov[ic], p1 = a.mov[(size_t)n + ic], p2 = a.mov[2 * (size_t)n + ic];
    // the match of the iteration before this launch: its index, and its coordinates gathered once — afterwards both ride along
    int jm = a.nn_prev[ic];
    double m0 = a.fix[jm], m1 = a.fix[(size_t)m + jm], m2 = a.fix[2 * (size_t)m + jm];
    if (tid < 16) A_s[tid] = a.A_prev[tid];             // fitted by the previous launch (kernel boundary: plain load)
    const int leaf0 = blockIdx.x * LPB;
    const int mine = min(LPB, a.leaves - leaf0);
    const int g = leaf0 / PM_TREE_GROUP;
    const unsigned int gsize = (unsigned int)min(PM_TREE_GROUP, a.leaves - g * PM_TREE_GROUP);
    const double o0 = a.origin6[0], o1 = a.origin6[1], o2 = a.origin6[2], o3 = a.origin6[3], o4 = a.origin6[4], o5 = a.origin6[5];
    __syncthreads();
    unsigned int round = 0;
    for (int it = la.first_it; it < la.iters; ++it) {
        ++round;
#ifdef PM_ICP_STAMPS
        const bool stamp_it = it == la.iters - 1;
#define PM_LSTAMP(k) do { if (stamp_it) PM_STAMP(k); } while (0)
#else
#define PM_LSTAMP(k) do { } while (0)
#endif
        PM_LSTAMP(0);
        // apply the transform fitted by the previous iteration (:23) and take the residual against the previous match (:24)
        {
            const double x = p0, y = p1, z = p2;
            p0 = affine_row(A_s, x, y, z);
            p1 = affine_row(A_s + 4, x, y, z);
            p2 = affine_row(A_s + 8, x, y, z);
        }
        const double e0 = p0 - m0, e1 = p1 - m1, e2 = p2 - m2;
        const double S0 = (e0 * e0 + e1 * e1) + e2 * e2;
        const double res_prev = __builtin_sqrt(S0);
        const bool known = S0 < INFINITY;                // (NaN fails the comparison)
        GridSearch<L, true> q(hd, a.start, a.pts, p0, p1, p2, sub, known ? S0 : INFINITY, known ? jm : 0x7fffffff,
                              known ? m0 : 0.0, known ? m1 : 0.0, known ? m2 : 0.0);
        bool done = false;
        if (known) done = q.ball(res_prev * (1.0 + 0x1p-20) + 0x1p-1000);
        if (!done) q.rings();
#ifdef PM_ICP_STAMPS
        if (q.bI == -5) __builtin_amdgcn_s_sleep(1);
#endif
        PM_LSTAMP(2);
        const bool none = q.bI == 0x7fffffff;            // all-NaN row: np.argmin answers 0
        jm = none ? 0 : q.bI;
        if (none) { m0 = a.fix[0]; m1 = a.fix[(size_t)m]; m2 = a.fix[2 * (size_t)m]; }
        else { m0 = q.bX; m1 = q.bY; m2 = q.bZ; }
        if (sub == 0) {
            double s[PM_NMOMENTS];
            if (i < n) {
                if (la.nn_all) la.nn_all[(size_t)it * n + i] = jm;
                moment_terms(p0 - o0, p1 - o1, p2 - o2, m0 - o3, m1 - o4, m2 - o5, s);
            } else {
#pragma unroll
                for (int k = 0; k < PM_NMOMENTS; ++k) s[k] = 0.0;
            }
#pragma unroll
            for (int k = 0; k < PM_NMOMENTS; ++k) term[slot][k] = s[k];
            term[slot][PM_NMOMENTS] = (i < n) ? res_prev : 0.0;
        }
        __syncthreads();
        if (tid < LPB * IT_SLOTS) {
            const int lf = tid / IT_SLOTS, k = tid - lf * IT_SLOTS;
            if (lf < mine) {
                double acc = 0.0;
#pragma unroll
                for (int pnt = 0; pnt < PM_TREE_LEAF; ++pnt) acc += term[lf * PM_TREE_LEAF + pnt][k];
                coherent_store(&a.leaf_partial[(size_t)(leaf0 + lf) * IT_STRIDE + k], acc);
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        PM_LSTAMP(3);
        __syncthreads();
        if (tid == 0) s_last_of_group = (atomicAdd(&a.counters[1 + g], (unsigned int)mine) + (unsigned int)mine == gsize * round) ? 1 : 0;
        __syncthreads();
        PM_LSTAMP(4);
        if (s_last_of_group) loop_round_tail(la, g, gsize, round, it, s_stage, totals, &s_last_of_all, &term[0][0]);   // this workgroup completed its group
        if (s_last_of_group) PM_LSTAMP(5);
        // everyone: wait for this round's transform
        if (tid == 0) {
            // (a foreign kernel on another stream may hold CUs that workgroups of this grid are still waiting for: the wait is
            // bounded by wall-clock time, not by a poll count, and the poll is paced — one L1-bypassing load per ~microsecond)
            const unsigned long long t_start = __builtin_amdgcn_s_memrealtime();
            int ok = 1;
            while (__hip_atomic_load(la.gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < round) {
                __builtin_amdgcn_s_sleep(PM_ICP_POLL_SLEEP);
                if (__builtin_amdgcn_s_memrealtime() - t_start > la.spin_ticks) { ok = 0; break; }
            }
            s_ok = ok;
        }
        __syncthreads();
        if (!s_ok) {                                   // the grid was not co-resident after all: report and drain
            if (tid == 0 && a.status) __hip_atomic_store(a.status, 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            break;
        }
        PM_LSTAMP(6);
        if (tid < 16) A_s[tid] = coherent_load(&a.A_est[tid]);
        __syncthreads();
        PM_LSTAMP(7);
    }
    // positions after every fit but the last, and the last matches: what the trailing update kernel applies the last fit to
    if (sub == 0 && i < n) {
        a.mov[i] = p0; a.mov[(size_t)n + i] = p1; a.mov[2 * (size_t)n + i] = p2;
        la.nn_last[i] = jm;
    }
}

size_t grid_ws_bytes(int m) { return grid_layout(m).total; }

int grid_build(const double *fix, int m, void *ws, hipStream_t s) {
    const GridWs L = grid_layout(m);
    char *base = (char *)ws;
    GridHeader *hd = (GridHeader *)(base + L.header);
    int *start = (int *)(base + L.start), *cursor = (int *)(base + L.cursor);
    const int max_cells = grid_max_cells(m);
    if (hipMemsetAsync(start, 0, ((size_t)max_cells + 1) * sizeof(int), s) != hipSuccess) return launch_status();
    grid_plan_kernel<<<1, 1024, 0, s>>>(fix, m, max_cells, hd);
    grid_count_kernel<<<(m + 255) / 256, 256, 0, s>>>(fix, m, hd, start);
    const int nchunks = max_cells / SCAN_CHUNK + 1;                          // covers index ncells <= max_cells too
    int *chunk = (int *)(base + L.chunk);
    grid_scan_sums_kernel<<<nchunks, 1024, 0, s>>>(hd, start, chunk);
    grid_scan_offsets_kernel<<<1, 1024, 0, s>>>(chunk, nchunks);
    grid_scan_apply_kernel<<<nchunks, 1024, 0, s>>>(hd, chunk, start, cursor);
    grid_scatter_kernel<<<(m + 255) / 256, 256, 0, s>>>(fix, m, hd, cursor, (double4 *)(base + L.pts));
    return launch_status();
}

int grid_query(const double *mov, int n, int m, const void *ws, int32_t *nn, double *dist, hipStream_t s) {
    const GridWs L = grid_layout(m);
    const char *base = (const char *)ws;
    const GridHeader *hd = (const GridHeader *)(base + L.header);
    const int *start = (const int *)(base + L.start);
    const double4 *pts = (const double4 *)(base + L.pts);
    const unsigned int blocks = (unsigned int)(((long)n * GR_LANES + 255) / 256);
    grid_nn_kernel<GR_LANES><<<blocks, 256, 0, s>>>(mov, n, hd, start, pts, nn, dist);
    return launch_status();
}

// one fused iteration: launch geometry and workspace sizes
int iter_leaves(int n) { return (n + PM_TREE_LEAF - 1) / PM_TREE_LEAF; }
int iter_groups(int n) { return (iter_leaves(n) + PM_TREE_GROUP - 1) / PM_TREE_GROUP; }
size_t iter_leaf_bytes(int n) { return (size_t)iter_leaves(n) * IT_STRIDE * sizeof(double); }
size_t iter_group_bytes(int n) { return (size_t)iter_groups(n) * IT_STRIDE * sizeof(double); }
size_t iter_gen_offset(int n) { return align_up((size_t)(1 + iter_groups(n)) * sizeof(unsigned int), 1024) + 1024; }   // (bytes from the counters' start)
size_t iter_counter_bytes(int n) { return iter_gen_offset(n) + 1024; }   // [total | groups...] ... [generation word: on a line no counter shares, so that polls do not queue with the arrival atomics]

// How many workgroups of the loop kernel the device holds at once; 0 if the query fails.  The occupancy query can be one high
// when the SGPR count is what binds (MI355X_MICROARCH.md, "Residency and cooperative launch": admitted per CU =
// min(query, 8, floor(800 / (ceil(sgpr/16)*16 + 16)))); at the architectural maximum of SGPRs that bound is 5, so the query is
// capped at 5 whatever the compiler allocated (the kernel is built for 4: __launch_bounds__(256, 4) and 37 KB of LDS) — and ONE
// workgroup per CU of that is kept back as margin for foreign kernels' workgroups already resident when the grid arrives.
template <int L>
static int loop_capacity_of(int dev) {
    int cus = 0, per = 0;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per, icp_loop_kernel<L>, 256, 0) != hipSuccess) return 0;
    per = std::min(per, 5) - 1;
    return per > 0 ? per * cus : 0;
}

int icp_loop_lanes(int n) { return n >= GR_ITER_FEW_LANES_FROM ? 4 : 8; }

bool icp_loop_fits(int n) {
    // cached per DEVICE (CU count and partition mode differ between devices; a thread may switch): [device][lanes == 4]
    constexpr int MAX_DEV = 64;
    static std::atomic<int> cap[MAX_DEV][2];               // 0 = not asked yet (stored as capacity + 1)
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= MAX_DEV) return false;
    const int lanes = icp_loop_lanes(n);
    std::atomic<int> &slot = cap[dev][lanes == 4];
    int c = slot.load(std::memory_order_relaxed);
    if (c == 0) {
        c = (lanes == 4 ? loop_capacity_of<4>(dev) : loop_capacity_of<8>(dev)) + 1;
        slot.store(c, std::memory_order_relaxed);
    }
    const long blocks = ((long)n * lanes + 255) / 256;
    return blocks <= c - 1;
}

// iterations first_it .. iters-1 in one launch (first_it >= 1: the launch before fitted A_est and left its matches in nn_prev)
int icp_loop(double *mov, int n, const double *fix, int m, const void *grid_ws, const int32_t *nn_prev, int32_t *nn_last, int32_t *nn_all,
             const double *origin6, double *leaf_partial, double *group_partial, unsigned int *counters, double *A_est, double *A_icp,
             double *residuals, int32_t *status, int first_it, int iters, hipStream_t s) {
    const GridWs Lw = grid_layout(m);
    const char *base = (const char *)grid_ws;
    LoopArgs la;
    IterArgs &a = la.a;
    a.mov = mov; a.n = n;
    a.hd = (const GridHeader *)(base + Lw.header); a.start = (const int *)(base + Lw.start); a.pts = (const double4 *)(base + Lw.pts);
    a.fix = fix; a.m = m;
    a.nn_prev = nn_prev; a.nn_out = nullptr;
    a.A_prev = A_est; a.origin6 = origin6;
    a.leaf_partial = leaf_partial; a.group_partial = group_partial; a.counters = counters;
    a.A_est = A_est; a.A_icp = A_icp; a.res_prev_out = nullptr; a.status = status;
    a.leaves = iter_leaves(n); a.groups = iter_groups(n);
    la.first_it = first_it; la.iters = iters; la.residuals = residuals; la.nn_all = nn_all; la.nn_last = nn_last;
    la.gen = (unsigned int *)((char *)counters + iter_gen_offset(n));
    la.spin_ticks = 200000000ull;                        // 2 s: only ever reached if the grid cannot become co-resident in that time
    if (icp_loop_lanes(n) == 4) icp_loop_kernel<4><<<(n + 63) / 64, 256, 0, s>>>(la);
    else icp_loop_kernel<8><<<(n + 31) / 32, 256, 0, s>>>(la);
    return launch_status();
}

int icp_iteration(bool first, double *mov, int n, const double *fix, int m, const void *grid_ws, const int32_t *nn_prev, int32_t *nn_out,
                  const double *origin6, double *leaf_partial, double *group_partial, unsigned int *counters, double *A_est,
                  double *A_icp, double *res_prev_out, int32_t *status, hipStream_t s) {
    const GridWs Lw = grid_layout(m);
    const char *base = (const char *)grid_ws;
    IterArgs a;
    a.mov = mov; a.n = n;
    a.hd = (const GridHeader *)(base + Lw.header); a.start = (const int *)(base + Lw.start); a.pts = (const double4 *)(base + Lw.pts);
    a.fix = fix; a.m = m;
    a.nn_prev = nn_prev; a.nn_out = nn_out;
    a.A_prev = A_est; a.origin6 = origin6;
    a.leaf_partial = leaf_partial; a.group_partial = group_partial; a.counters = counters;
    a.A_est = A_est; a.A_icp = A_icp; a.res_prev_out = res_prev_out; a.status = status;
    a.leaves = iter_leaves(n); a.groups = iter_groups(n);
    // first iteration: nothing bounds the search, 32 lanes per point walk the rings; afterwards the previous match does
    // and 8 or 4 lanes per point (4 or 8 leaves per workgroup) are plenty for the few cells left
    if (first) icp_iter_kernel<GR_LANES, true><<<(n + 256 / GR_LANES - 1) / (256 / GR_LANES), 256, 0, s>>>(a);
#ifdef PM_GR_TRY_LANES      // (tuning builds: PM_EXTRA_DEFINES, tools/icp_stamps.py)
    else if (n >= GR_ITER_FEW_LANES_FROM) icp_iter_kernel<PM_GR_TRY_LANES, false><<<(n + 256 / PM_GR_TRY_LANES - 1) / (256 / PM_GR_TRY_LANES), 256, 0, s>>>(a);
#else
    else if (n >= GR_ITER_FEW_LANES_FROM) icp_iter_kernel<4, false><<<(n + 63) / 64, 256, 0, s>>>(a);
#endif
    else icp_iter_kernel<8, false><<<(n + 31) / 32, 256, 0, s>>>(a);
    return launch_status();
}

}  // namespace pm
