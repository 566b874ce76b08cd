// rcp_accuracy.hip — largest relative error of v_rcp_f64 / v_rsq_f64 on gfx950 over a dense sweep of mantissas.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void sweep(double *out_max, unsigned long long stride, unsigned long long count) {
    unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
    double worst = 0.0, worst_rsq = 0.0;
    for (unsigned long long k = i; k < count; k += (unsigned long long)gridDim.x * blockDim.x) {
        unsigned long long bits = 0x3FF0000000000000ull + k * stride;      // d in [1, 2)
        double d = __longlong_as_double(bits);
        double y = __builtin_amdgcn_rcp(d);
        double e = __builtin_fabs(__builtin_fma(-d, y, 1.0));            // |1 - d*y| = relative error of y
        worst = e > worst ? e : worst;
        double r = __builtin_amdgcn_rsq(d);
        double e2 = __builtin_fabs(__builtin_fma(-d * r, r, 1.0));       // ~ 2 * relative error of r
        worst_rsq = e2 > worst_rsq ? e2 : worst_rsq;
    }
    for (int off = 32; off > 0; off >>= 1) {
        double o = __shfl_down(worst, off, 64); worst = o > worst ? o : worst;
        double o2 = __shfl_down(worst_rsq, off, 64); worst_rsq = o2 > worst_rsq ? o2 : worst_rsq;
    }
    if ((threadIdx.x & 63) == 0) { out_max[2 * (i >> 6)] = worst; out_max[2 * (i >> 6) + 1] = worst_rsq; }
}
int main() {
    const int blocks = 2048, threads = 256;
    double *d_out; hipMalloc(&d_out, sizeof(double) * 2 * blocks * threads / 64);
    const unsigned long long count = 1ull << 34, stride = (1ull << 52) / count;   // 1.7e10 evenly spaced mantissas
    sweep<<<blocks, threads>>>(d_out, stride, count);
    hipDeviceSynchronize();
    static double h[2 * 2048 * 256 / 64];
    hipMemcpy(h, d_out, sizeof(h), hipMemcpyDeviceToHost);
    double w = 0, w2 = 0;
    for (int i = 0; i < blocks * threads / 64; ++i) { if (h[2 * i] > w) w = h[2 * i]; if (h[2 * i + 1] > w2) w2 = h[2 * i + 1]; }
    printf("v_rcp_f64 max relative error over 2^34 mantissas: %.3e = 2^%.2f\n", w, log2(w));
    printf("v_rsq_f64 max |1 - d r^2| over 2^34 mantissas:    %.3e = 2^%.2f\n", w2, log2(w2));
    return 0;
}
