/* Host build of the product's binning header, for the CPU test-suite:
 * tests/test_binning_host.py compares it with the oracle's libm-based bin index. */
#include <stdint.h>
#include <math.h>
#include "pm_binning.h"

/* nb: n x 3 frame coordinates; out: n int32 (PM_DROP = -1 for not counted) */
int pmt_bin_index(const double *nb, int n, double mean_dist, int32_t *out) {
    for (int i = 0; i < n; ++i) {
        double x = nb[3 * i], y = nb[3 * i + 1], z = nb[3 * i + 2];
        double r_ = sqrt(fma(z, z, fma(y, y, x * x)));     /* the neighbour-list kernels' form: np.linalg.norm's fused chain (shape_context.py:29) */
        out[i] = pm_bin_index(x, y, z, r_, r_ / mean_dist);
    }
    return 0;
}
int pmt_phi_index(const double *xy, int n, int32_t *out) {
    for (int i = 0; i < n; ++i) out[i] = pm_phi_index(xy[2 * i], xy[2 * i + 1]);
    return 0;
}

/* the kernel's per-neighbour step: all four frames at once; out: n x 4 int32 */
int pmt_bin_index4(const double *nb, int n, double mean_dist, int nframes, int32_t *out) {
    double rho[4];
    pm_ring_thresholds(mean_dist, rho);
    for (int i = 0; i < n; ++i) {
        int b[4];
        pm_bin_index4(nb[3 * i], nb[3 * i + 1], nb[3 * i + 2], rho, nframes, b);
        for (int f = 0; f < 4; ++f) out[4 * i + f] = b[f];
    }
    return 0;
}
int pmt_ring_thresholds(double md, double *rho) { pm_ring_thresholds(md, rho); return 0; }

/* the tile kernel's float32 pre-classification: v = n x 3 float64 difference vectors (world coordinates), fr = the frame
 * (x, y, z: 9 float64), md = mean distance; out[i] = pm_bin_fast32's answer (-1: not decided in float32).
 * proj (n x 3, may be NULL) receives the frame coordinates in the oracle's operation order, for the float64 comparison. */
int pmt_bin_fast32(const double *v, int n, const double *fr, double md, int32_t *out, double *proj) {
    float fr32[9];
    for (int k = 0; k < 9; ++k) fr32[k] = (float)fr[k];
    const double k64d = 64.0 / (md * md);
    const int ok = (md > 0.0) && (k64d >= 0x1p-30) && (k64d <= 0x1p+30);
    const float k64 = ok ? (float)k64d : 0.0f;
    for (int i = 0; i < n; ++i) {
        const double v0 = v[3 * i], v1 = v[3 * i + 1], v2 = v[3 * i + 2];
        out[i] = ok ? pm_bin_fast32((float)v0, (float)v1, (float)v2, fr32, k64, 0.0f) : -1;
        if (proj) {
            proj[3 * i] = (fr[0] * v0 + fr[1] * v1) + fr[2] * v2;
            proj[3 * i + 1] = (fr[3] * v0 + fr[4] * v1) + fr[5] * v2;
            proj[3 * i + 2] = (fr[6] * v0 + fr[7] * v1) + fr[8] * v2;
        }
    }
    return 0;
}
int pmt_bin_perm(int f, int bin) { return pm_bin_perm(f, bin); }
