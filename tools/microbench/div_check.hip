// div_check.hip — is a shortened float64 division sequence still correctly rounded?
// Compares, bit for bit, against the compiler's IEEE division (n / d):
//   seq A: v_rcp_f64 + two Newton steps + residual correction  (pm::div_pos as shipped)
//   seq B: v_rcp_f64 + ONE cubic step (y1 = y0 + y0*(e + e*e)) + residual correction  (one fma fewer)
// on (1) chi-square operands built from random histogram counts, (2) random doubles, (3) quotients
// constructed to lie within ~2^-53 ulp of a rounding midpoint — the hard cases of division.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

__device__ __forceinline__ double div_a(double n, double d) {
    double r = __builtin_amdgcn_rcp(d);
    double e = __builtin_fma(-d, r, 1.0); r = __builtin_fma(e, r, r);
    e = __builtin_fma(-d, r, 1.0); r = __builtin_fma(e, r, r);
    double q = n * r; double res = __builtin_fma(-d, q, n);
    return __builtin_fma(res, r, q);
}
__device__ __forceinline__ double div_b(double n, double d) {
    double r = __builtin_amdgcn_rcp(d);
    double e = __builtin_fma(-d, r, 1.0);
    double t = __builtin_fma(e, e, e);
    r = __builtin_fma(r, t, r);
    double q = n * r; double res = __builtin_fma(-d, q, n);
    return __builtin_fma(res, r, q);
}
__device__ __forceinline__ double div_c(double n, double d) {   // control: ONE Newton step only (2^-48 reciprocal) — must fail
    double r = __builtin_amdgcn_rcp(d);
    double e = __builtin_fma(-d, r, 1.0); r = __builtin_fma(e, r, r);
    double q = n * r; double res = __builtin_fma(-d, q, n);
    return __builtin_fma(res, r, q);
}
// seq D: as B, but from a seed that is NOT v_rcp_f64's own output: the reciprocal perturbed by a random relative error of up to
// +-2^-23 (what a seed recovered from a batched inversion — one rcp of a product, then multiplications — could look like)
__device__ __forceinline__ double div_d(double n, double d, unsigned long long bits) {
    double r = __builtin_amdgcn_rcp(d);
    const double delta = (double)((long long)(bits >> 11) - (1ll << 52)) * 0x1p-75;     // uniform in [-2^-23, 2^-23)
    r = __builtin_fma(r, delta, r);
    double e = __builtin_fma(-d, r, 1.0);
    double t = __builtin_fma(e, e, e);
    r = __builtin_fma(r, t, r);
    double q = n * r; double res = __builtin_fma(-d, q, n);
    return __builtin_fma(res, r, q);
}
__device__ __forceinline__ unsigned long long rng(unsigned long long &s) {
    s ^= s >> 12; s ^= s << 25; s ^= s >> 27; return s * 0x2545F4914F6CDD1Dull;
}

__global__ void check(unsigned long long *bad, int mode, int iters) {
    unsigned long long s = 0x9E3779B97F4A7C15ull ^ ((unsigned long long)(blockIdx.x * blockDim.x + threadIdx.x) * 0xD1B54A32D192ED03ull + mode);
    unsigned long long badA = 0, badB = 0, badC = 0, badD = 0;
    for (int it = 0; it < iters; ++it) {
        double n, d;
        if (mode == 0) {                    // histogram-like operands
            unsigned long long r1 = rng(s), r2 = rng(s);
            double Ta = (double)(100 + (r1 & 0xFFFF)), Tb = (double)(100 + ((r1 >> 16) & 0xFFFF));
            double ca = (double)((r2 & 0xFFF) % (unsigned)Ta), cb = (double)(((r2 >> 12) & 0xFFF) % (unsigned)Tb);
            double a = ca / Ta, b = cb / Tb;
            if (b == 0.0) b = 1e-300;
            double df = a - b; n = df * df; d = a + b;
        } else if (mode == 1) {             // random doubles, exponents spread over ~80 binades
            unsigned long long r1 = rng(s), r2 = rng(s);
            n = __longlong_as_double(((1023ull - 40 + (r1 >> 58)) << 52) | (r1 & 0xFFFFFFFFFFFFFull));
            d = __longlong_as_double(((1023ull - 20 + (r2 >> 59)) << 52) | (r2 & 0xFFFFFFFFFFFFFull));
        } else {                            // n/d within ~rho * 2^-54 ulp of a midpoint between two doubles
            // odd divisor d, small odd rho: choose w = 2m+1 with d*w = +-rho (mod 2^54); then n = (d*w -+ rho) / 2^54 is an
            // integer and n/d = (m + 1/2) * 2^-53 -+ rho / (2^54 d): a hair from the midpoint of two neighbouring doubles
            const unsigned long long M54 = (1ull << 54) - 1;
            unsigned long long di = (1ull << 52) | (rng(s) & 0xFFFFFFFFFFFFFull) | 1ull;
            unsigned long long inv = di;                       // Newton for the inverse modulo 2^64
            for (int k = 0; k < 6; ++k) inv *= 2 - di * inv;
            unsigned long long r = rng(s);
            unsigned long long rho = ((r & 0x3FF) << 1) | 1ull;
            const bool above = (r >> 63) != 0;
            unsigned long long w = ((above ? (0 - rho) : rho) * inv) & M54;
            if (w < (1ull << 53)) w = ((above ? (0 - (rho + 2)) : (rho + 2)) * inv) & M54, rho += 2;
            if (w < (1ull << 53)) { --it; continue; }         // 2m+1 out of range for this divisor: draw again
            unsigned __int128 P = (unsigned __int128)di * w;
            unsigned long long ni = (unsigned long long)((above ? P + rho : P - rho) >> 54);
            n = (double)ni; d = (double)di;
        }
        double ref = n / d;
        badA += (__double_as_longlong(div_a(n, d)) != __double_as_longlong(ref));
        badB += (__double_as_longlong(div_b(n, d)) != __double_as_longlong(ref));
        badC += (__double_as_longlong(div_c(n, d)) != __double_as_longlong(ref));
        badD += (__double_as_longlong(div_d(n, d, rng(s))) != __double_as_longlong(ref));
    }
    atomicAdd(&bad[3 * mode], badA);
    atomicAdd(&bad[3 * mode + 1], badB);
    atomicAdd(&bad[3 * mode + 2], badC);
    atomicAdd(&bad[9 + mode], badD);
}

int main(int argc, char **argv) {
    unsigned long long *bad; hipMalloc(&bad, 12 * sizeof(unsigned long long)); hipMemset(bad, 0, 96);
    const int blocks = 4096, threads = 256, iters = argc > 1 ? atoi(argv[1]) : 40000;    // default 4.2e10 divisions per mode
    for (int mode = 0; mode < 3; ++mode) check<<<blocks, threads>>>(bad, mode, iters);
    hipDeviceSynchronize();
    unsigned long long h[12]; hipMemcpy(h, bad, 96, hipMemcpyDeviceToHost);
    const char *names[3] = {"histogram operands", "random doubles", "near-midpoint quotients"};
    for (int m = 0; m < 3; ++m)
        printf("%-26s %.2e divisions: mismatches vs IEEE '/'  seqA(2 Newton) %llu   seqB(1 cubic) %llu   control(1 Newton) %llu   seqD(1 cubic, seed off by up to 2^-23) %llu\n", names[m],
               (double)blocks * threads * iters, h[3 * m], h[3 * m + 1], h[3 * m + 2], h[9 + m]);
    return 0;
}
