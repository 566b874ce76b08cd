/*
 * pm_oracle.c — CPU restatement of the PlatyMatch estimate_transform hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  This file is the checker the HIP path is compared
 * against.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
 * leg may build, load or call it.  Nothing under platymatch_amd/ imports it,
 * links it or falls back to it.
 *
 * Parity status: PINNED.  Every function below is checked against outputs of
 * the unmodified reference run in the build container (the .npz fixtures under tests/golden/,
 * produced by tests/golden/gen_golden.py) by tests/test_oracle_golden.py.
 *
 * Plain scalar C, IEEE-754 binary64, one rounding per written operation:
 * build with -ffp-contract=off and without -ffast-math (oracle/Makefile).
 * Each function cites the reference lines it follows (paths relative to the
 * reference checkout, juglab/PlatyMatch setup.py version 0.0.4).
 *
 * Third-party arithmetic on the path that is not in the reference tree
 * (requirements.txt pins no versions; versions of the build container):
 *   numpy 2.2.6   floor_divide on float64 (npy_divmod), np.linalg.norm, np.cross
 *   scipy 1.15.3  spatial.distance_matrix (minkowski p=2), optimize.linear_sum_assignment
 * Their published algorithms are restated where used.
 *
 * Two places where the reference's own bits depend on its C / linear-algebra libraries, and what this file does there
 * (measured by tests/golden/soak_oracle_vs_reference.py):
 *   x ** 2 on NumPy float64 scalars (get_unary_distance, shape_context.py:95) is libm pow(x, 2.0); glibc >= 2.28 does not
 *     round it correctly (one operand in ~2 000 differs from x * x by an ulp).  Squared here by multiplication: 10 of
 *     1 035 624 cost entries differ from this container's reference by one ulp.
 *   transform() (shape_context.py:61-84) builds local coordinates through np.linalg.inv of a 4 x 4: LAPACK's rounding leaves
 *     up to 4e-14 x |d|_1 on each.  Projected directly here; histograms are the reference's unless a neighbour lies within
 *     that noise of a bin boundary or duplicates the queried point (generic clouds: never in 577 random pairs).
 */
#include <math.h>
#include <omp.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define PM_NBINS 360
#ifndef M_PI
#define M_PI 0x1.921fb54442d18p+1 /* np.pi */
#endif

/* r_edges = np.logspace(np.log10(1/8), np.log10(2), 5)   (shape_context.py:24)
 * The exact float64 values numpy 2.2.6 produces (tests/golden/micro.npz:r_edges);
 * note edges 1 and 2 are one ulp above 1/4 and 1/2. */
static const double R_EDGES[5] = {0x1.0000000000000p-3, 0x1.0000000000001p-2,
                                  0x1.0000000000001p-1, 0x1.0000000000000p+0,
                                  0x1.0000000000000p+1};

/* numpy float64 `//`: npy_divmod (numpy/_core/src/npymath/npy_math_internal.h.src),
 * used at shape_context.py:51-52. */
static double np_floor_divide(double a, double b) {
    if (b == 0.0) return a / b;
    double mod = fmod(a, b);
    double div = (a - mod) / b;
    if (mod) { /* true for NaN too, as in the C original */
        if ((b < 0) != (mod < 0)) { mod += b; div -= 1.0; }
    } else {
        mod = copysign(0.0, b);
    }
    double fd;
    if (div) {
        fd = floor(div);
        if (div - fd > 0.5) fd += 1.0;
    } else {
        fd = copysign(0.0, a / b);
    }
    return fd;
}

/* np.linalg.norm of a 3-vector: sqrt(x.dot(x)) (numpy/linalg/_linalg.py norm, ord=None). */
static double norm3(double a, double b, double c) { return sqrt((a * a + b * b) + c * c); }
static double dot3(const double *a, const double *b) { return (a[0] * b[0] + a[1] * b[1]) + a[2] * b[2]; }

/* One neighbour, already expressed in the local frame: the loop body of
 * get_shape_context (shape_context.py:25-35) followed by get_bin_index
 * (shape_context.py:46-58).  Returns the float bin index (may be NaN, may be >= 360). */
static double bin_index_r(double x_, double y_, double z_, double r_, double mean_dist) {
    double r = r_ / mean_dist;                     /* :30 */
    double theta = acos(z_ / r_);                  /* :31 */
    double at = atan2(y_, x_);                     /* :32-35 */
    double phi = (at < 0) ? (2 * M_PI + at) : at;
    double r_index = 5 - 1;                        /* :49 */
    double theta_index = np_floor_divide(theta, M_PI / 6);      /* :51 */
    double phi_index = np_floor_divide(phi, 2 * M_PI / 12);     /* :52 */
    for (int k = 0; k < 5; ++k)                    /* :53-56 */
        if (r < R_EDGES[k]) { r_index = k; break; }
    return r_index * 6 * 12 + theta_index * 12 + phi_index;     /* :57 */
}

/* Inside get_unary the neighbours come out of transform()'s 4x4 inverse (:61-84), which this restatement replaces by a direct
 * projection (see sc_one_point): :29's norm is taken in the plain order there — the two differ by far less than that step's own
 * noise (the product's edge guard counts the neighbours it could matter for). */
static double bin_index(double x_, double y_, double z_, double mean_dist) {
    return bin_index_r(x_, y_, z_, norm3(x_, y_, z_), mean_dist);      /* :29 */
}

/* get_shape_context + get_bin_index on an explicit neighbour list (n x 3, row-major).  Here the neighbours ARE the reference's
 * input, so :29 is restated to the bit: np.linalg.norm of a 3-vector is sqrt(x.dot(x)) with BLAS ddot, whose x86-64 kernels
 * accumulate with fused multiply-adds — sqrt(fma(z, z, fma(y, y, x * x))) (20 000 of 20 000 random vectors on this host; the
 * plain order matches 17 905; tests/test_oracle_golden.py pins it against NumPy itself). */
int pmo_bin_index_projected(const double *nb, int n, double mean_dist, double *idx_out) {     /* get_unary's inner step as sc_one_point takes it */
    for (int i = 0; i < n; ++i) idx_out[i] = bin_index(nb[3 * i], nb[3 * i + 1], nb[3 * i + 2], mean_dist);
    return 0;
}
int pmo_bin_index(const double *nb, int n, double mean_dist, double *idx_out) {
    for (int i = 0; i < n; ++i) {
        const double x_ = nb[3 * i], y_ = nb[3 * i + 1], z_ = nb[3 * i + 2];
        idx_out[i] = bin_index_r(x_, y_, z_, sqrt(fma(z_, z_, fma(y_, y_, x_ * x_))), mean_dist);
    }
    return 0;
}

/* Threads for the row loops below (rows are independent: every output element is produced by one thread, so results do
 * not depend on the count).  Default 1; bench.py's cpu_baseline leg raises it to the host's cores for its all-cores figure. */
static int g_threads = 1;
int pmo_set_threads(int t) {
    g_threads = t > 0 ? t : 1;
    return g_threads;
}
int pmo_max_threads(void) { return omp_get_num_procs(); }

/* get_mean_distance (utils/utils.py:58-75): np.average of the list [np.linalg.norm(p_i - p_j) for i < j] — restated exactly
 * (round 3; bit-identical to the reference on every fixture, tests/test_oracle_golden.py):
 *   np.linalg.norm of a 3-vector = sqrt(x.dot(x)); the dot product is BLAS ddot, whose x86-64 kernels accumulate with fused
 *   multiply-adds (established against NumPy on the build host: tests/test_oracle_golden.py pins it through the fixtures);
 *   np.average -> np.add.reduce in pieces of 8 192 elements (np.getbufsize()), each piece summed pairwise (numpy/_core/src/
 *   umath/loops_utils.h.src: blocks of <= 128 with eight interleaved partial sums), the pieces added first to last; / count. */
static double np_pairwise(const double *a, long n) {
    if (n < 8) {
        double res = 0.0;
        for (long i = 0; i < n; ++i) res += a[i];
        return res;
    }
    if (n <= 128) {
        double r[8];
        for (int j = 0; j < 8; ++j) r[j] = a[j];
        long i;
        for (i = 8; i < n - (n % 8); i += 8)
            for (int j = 0; j < 8; ++j) r[j] += a[i + j];
        double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
        for (; i < n; ++i) res += a[i];
        return res;
    }
    long half = n / 2;
    half -= half % 8;
    return np_pairwise(a, half) + np_pairwise(a + half, n - half);
}

int pmo_mean_distance(const double *xyz, int n, double *out) {
    const double *X = xyz, *Y = xyz + n, *Z = xyz + 2 * (size_t)n;
    const long long P = (long long)n * (n - 1) / 2;
    if (P <= 0) { *out = 0.0 / 0.0; return 0; }                 /* np.average([]) */
    const long long nchunks = (P + 8191) / 8192;
    double *piece = (double *)malloc(sizeof(double) * (size_t)nchunks);
    if (!piece) return -1;
#pragma omp parallel num_threads(g_threads)
    {
        double *buf = (double *)malloc(sizeof(double) * 8192);
#pragma omp for schedule(dynamic, 16)
        for (long long c = 0; c < nchunks; ++c) {
            const long long e0 = c * 8192;
            const long len = (long)((P - e0 < 8192) ? P - e0 : 8192);
            /* element e0 -> (i, j): row i starts at i (n - 1) - i (i - 1) / 2 */
            long long i = (long long)(((2.0 * n - 1.0) - sqrt(fmax((2.0 * n - 1.0) * (2.0 * n - 1.0) - 8.0 * (double)e0, 0.0))) * 0.5);
            if (i < 0) i = 0;
            if (i > n - 2) i = n - 2;
            while (i < n - 2 && (i + 1) * (long long)(n - 1) - (i + 1) * i / 2 <= e0) ++i;
            while (i > 0 && i * (long long)(n - 1) - i * (i - 1) / 2 > e0) --i;
            long long j = i + 1 + (e0 - (i * (long long)(n - 1) - i * (i - 1) / 2));
            for (long t = 0; t < len; ++t) {
                const double d0 = X[i] - X[j], d1 = Y[i] - Y[j], d2 = Z[i] - Z[j];
                buf[t] = sqrt(fma(d2, d2, fma(d1, d1, d0 * d0)));
                if (++j >= n) { ++i; j = i + 1; }
            }
            piece[c] = np_pairwise(buf, len);
        }
        free(buf);
    }
    double total = piece[0];
    for (long long c = 1; c < nchunks; ++c) total += piece[c];   /* the same value for any thread count */
    free(piece);
    *out = total / (double)P;
    return 0;
}

/* get_unary (shape_context.py:144-188) for one cloud, as integer histograms.
 *   xyz       3 x n, row-major (the reference's 3 x N layout, transposed=False)
 *   centroid  3, x0 3 (first PCA axis, shape_context.py:162-165), mean_dist
 *   n_frames  2 ('moving') or 4 ('fixed')
 *   counts    [n_frames][n][360] int32: index.count(i) (shape_context.py:39-40)
 *   totals    [n_frames][n]      int32: sc.sum() before normalisation (:41)
 * The local frame follows :169-175 and :180-181; `transform` (:61-84) maps
 * p_i -> 0 and p_i+x,y,z -> e1,e2,e3, i.e. neighbour -> [x y z]^T (p_j - p_i):
 * this restatement projects directly (measured identical
 * integer histograms). */
static void sc_one_point(const double *xyz, int n, int i, const double *centroid, const double *x0, double mean_dist,
                         int n_frames, int32_t *counts, int32_t *totals, size_t frame_stride_counts, size_t frame_stride_totals) {
    const double *P0 = xyz, *P1 = xyz + n, *P2 = xyz + 2 * (size_t)n;
    double p[3] = {P0[i], P1[i], P2[i]};
    double w[3] = {p[0] - centroid[0], p[1] - centroid[1], p[2] - centroid[2]};
    double nw = norm3(w[0], w[1], w[2]);
    double z[3] = {w[0] / nw, w[1] / nw, w[2] / nw};                  /* :169 */
    double d = dot3(x0, z);
    double x[3] = {x0[0] - z[0] * d, x0[1] - z[1] * d, x0[2] - z[2] * d}; /* :170 */
    double nx = norm3(x[0], x[1], x[2]);
    x[0] /= nx; x[1] /= nx; x[2] /= nx;                                /* :171 */
    double y[3] = {z[1] * x[2] - z[2] * x[1], z[2] * x[0] - z[0] * x[2], z[0] * x[1] - z[1] * x[0]}; /* get_Y :6-8 */
    double ny = norm3(y[0], y[1], y[2]);
    y[0] /= ny; y[1] /= ny; y[2] /= ny;
    /* frame 2: x0 -> -x0 gives (-x, -y) exactly (:172-175); frame 3: (x, -y); frame 4: (-x, +y) (:180-181) */
    static const double SX[4] = {1, -1, 1, -1}, SY[4] = {1, -1, -1, 1};
    for (int j = 0; j < n; ++j) {
        if (j == i) continue;                                          /* np.delete :168 */
        double v[3] = {P0[j] - p[0], P1[j] - p[1], P2[j] - p[2]};
        double vx = dot3(x, v), vy = dot3(y, v), vz = dot3(z, v);
        for (int f = 0; f < n_frames; ++f) {
            double idx = bin_index(SX[f] * vx, SY[f] * vy, vz, mean_dist);
            if (idx >= 0 && idx < PM_NBINS && idx == floor(idx)) {
                counts[(size_t)f * frame_stride_counts + (int)idx] += 1;
                totals[(size_t)f * frame_stride_totals] += 1;
            }
        }
    }
}

int pmo_shape_context_counts(const double *xyz, int n, const double *centroid, const double *x0,
                             double mean_dist, int n_frames, int32_t *counts, int32_t *totals) {
    memset(counts, 0, sizeof(int32_t) * (size_t)n_frames * n * PM_NBINS);
    memset(totals, 0, sizeof(int32_t) * (size_t)n_frames * n);
#pragma omp parallel for schedule(dynamic, 4) num_threads(g_threads)
    for (int i = 0; i < n; ++i)
        sc_one_point(xyz, n, i, centroid, x0, mean_dist, n_frames, counts + (size_t)i * PM_NBINS, totals + i,
                     (size_t)n * PM_NBINS, (size_t)n);
    return 0;
}

/* The same for a list of query points only (against the whole cloud): counts [n_frames][n_rows][360], totals
 * [n_frames][n_rows].  For checks at sizes where all N rows would take hours on one core. */
int pmo_shape_context_rows(const double *xyz, int n, const int32_t *rows, int n_rows, const double *centroid, const double *x0,
                           double mean_dist, int n_frames, int32_t *counts, int32_t *totals) {
    memset(counts, 0, sizeof(int32_t) * (size_t)n_frames * n_rows * PM_NBINS);
    memset(totals, 0, sizeof(int32_t) * (size_t)n_frames * n_rows);
    for (int r = 0; r < n_rows; ++r)
        if (rows[r] < 0 || rows[r] >= n) return -1;
#pragma omp parallel for schedule(dynamic, 1) num_threads(g_threads)
    for (int r = 0; r < n_rows; ++r) {
        sc_one_point(xyz, n, rows[r], centroid, x0, mean_dist, n_frames, counts + (size_t)r * PM_NBINS, totals + r,
                     (size_t)n_rows * PM_NBINS, (size_t)n_rows);
    }
    return 0;
}

/* get_unary_distance (shape_context.py:88-99) for every (i, j): the widget's
 * N x M double loops (_dock_widget.py:547-602).  Sequential sum over the 360
 * bins in index order, bins with a == b skipped. */
int pmo_chi2(const double *a, int n, const double *b, int m, double *out) {
#pragma omp parallel for schedule(dynamic, 1) num_threads(g_threads)
    for (int i = 0; i < n; ++i) {
        const double *ai = a + (size_t)i * PM_NBINS;
        for (int j = 0; j < m; ++j) {
            const double *bj = b + (size_t)j * PM_NBINS;
            double dist = 0.0;
            for (int k = 0; k < PM_NBINS; ++k) {
                if (ai[k] != bj[k]) {
                    double df = ai[k] - bj[k];
                    dist = dist + (df * df) / (ai[k] + bj[k]);
                }
            }
            out[(size_t)i * m + j] = 0.5 * dist;
        }
    }
    return 0;
}

/* perform_icp's correspondence step (perform_icp.py:15-16):
 * scipy.spatial.distance_matrix = sum(|x-y|**2, axis=-1)**0.5 (scipy/spatial/_kdtree.py
 * minkowski_distance, p=2), then np.argmin(axis=1) — first index on ties. */
int pmo_nn_argmin(const double *mov, int n, const double *fix, int m, int32_t *idx, double *dist) {
#pragma omp parallel for schedule(dynamic, 16) num_threads(g_threads)
    for (int i = 0; i < n; ++i) {
        double best = INFINITY;
        int bj = 0;
        for (int j = 0; j < m; ++j) {
            double d0 = fix[j] - mov[i], d1 = fix[m + j] - mov[n + i], d2 = fix[2 * (size_t)m + j] - mov[2 * (size_t)n + i];
            double d = sqrt((d0 * d0 + d1 * d1) + d2 * d2);
            if (d < best) { best = d; bj = j; }
        }
        idx[i] = bj;
        if (dist) dist[i] = best;
    }
    return 0;
}

/* do_ransac's scoring loop (shape_context.py:130-135) for a batch of candidate
 * transforms: predicted = A . [moving; 1] (apply_affine_transform, apply_transform.py:13-16),
 * inliers = #{ ||fixed - predicted|| <= error }. */
int pmo_ransac_score(const double *mov, const double *fix, int n, const double *A, int trials,
                     double error, int32_t *inliers) {
#pragma omp parallel for schedule(dynamic, 8) num_threads(g_threads)
    for (int t = 0; t < trials; ++t) {
        const double *a = A + 16 * (size_t)t;
        int cnt = 0;
        for (int i = 0; i < n; ++i) {
            double m0 = mov[i], m1 = mov[n + i], m2 = mov[2 * (size_t)n + i];
            /* :128 apply_affine_transform = np.matmul: BLAS dgemm, a chain of fused multiply-adds over k = 0..3;
             * :133 np.linalg.norm of a 3-vector = sqrt(x.dot(x)): BLAS ddot, the same kind of chain (as in pmo_mean_distance) */
            double p0 = fma(a[3], 1.0, fma(a[2], m2, fma(a[1], m1, a[0] * m0)));
            double p1 = fma(a[7], 1.0, fma(a[6], m2, fma(a[5], m1, a[4] * m0)));
            double p2 = fma(a[11], 1.0, fma(a[10], m2, fma(a[9], m1, a[8] * m0)));
            double e0 = fix[i] - p0, e1 = fix[n + i] - p1, e2 = fix[2 * (size_t)n + i] - p2;
            double d = sqrt(fma(e2, e2, fma(e1, e1, e0 * e0)));
            if (d <= error) ++cnt;
        }
        inliers[t] = cnt;
    }
    return 0;
}
