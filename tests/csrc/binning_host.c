/* Host build of the product's binning header, for the CPU test-suite:
 * tests/test_binning_host.py compares it with the oracle's libm-based bin index. */
#include <stdint.h>
#include <math.h>
#include "pm_binning.h"

/* nb: n x 3 frame coordinates; out: n int32 (PM_DROP = -1 for not counted) */
int pmt_bin_index(const double *nb, int n, double mean_dist, int32_t *out) {
    for (int i = 0; i < n; ++i) {
        double x = nb[3 * i], y = nb[3 * i + 1], z = nb[3 * i + 2];
        double r_ = sqrt((x * x + y * y) + z * z);
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
