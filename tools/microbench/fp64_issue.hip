// fp64_issue.hip — issue cost of the float64 instructions the chi-square kernel is made of (gfx950).
// Each kernel runs ITER x 8 independent instructions per wave; every SIMD of every CU gets `waves` waves.
// Build: hipcc -O3 --offload-arch=gfx950 fp64_issue.hip -o fp64_issue ; run: ./fp64_issue
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define ITER 4096

#define KERNEL(name, BODY)                                                                      \
    __global__ void name(double *out, double seed) {                                            \
        double a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, \
               a7 = a0 + 7;                                                                     \
        double c = 1.0000001, d = 0.9999999;                                                    \
        unsigned long long t0 = __builtin_amdgcn_s_memtime();                                   \
        for (int i = 0; i < ITER; ++i) { BODY }                                                 \
        unsigned long long t1 = __builtin_amdgcn_s_memtime();                                   \
        out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + (double)(t1 - t0); \
        if (threadIdx.x == 0 && blockIdx.x == 0) out[gridDim.x * blockDim.x] = (double)(t1 - t0); \
    }

#define OP8(INS) \
    asm volatile(INS " %0, %0, %8, %9\n" INS " %1, %1, %8, %9\n" INS " %2, %2, %8, %9\n" INS " %3, %3, %8, %9\n" \
                 INS " %4, %4, %8, %9\n" INS " %5, %5, %8, %9\n" INS " %6, %6, %8, %9\n" INS " %7, %7, %8, %9\n" \
                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c), "v"(d));
#define OP8_2(INS) \
    asm volatile(INS " %0, %0, %8\n" INS " %1, %1, %8\n" INS " %2, %2, %8\n" INS " %3, %3, %8\n" \
                 INS " %4, %4, %8\n" INS " %5, %5, %8\n" INS " %6, %6, %8\n" INS " %7, %7, %8\n" \
                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c));
#define OP8_1(INS) \
    asm volatile(INS " %0, %0\n" INS " %1, %1\n" INS " %2, %2\n" INS " %3, %3\n" \
                 INS " %4, %4\n" INS " %5, %5\n" INS " %6, %6\n" INS " %7, %7\n" \
                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));

KERNEL(k_fma, OP8("v_fma_f64"))
KERNEL(k_mul, OP8_2("v_mul_f64"))
KERNEL(k_add, OP8_2("v_add_f64"))
KERNEL(k_rcp, OP8_1("v_rcp_f64"))
KERNEL(k_rsq, OP8_1("v_rsq_f64"))
KERNEL(k_sqrt, OP8_1("v_sqrt_f64"))
KERNEL(k_max, OP8_2("v_max_f64"))

__global__ void k_rcp32(double *out, double seed) {   // cvt_f32_f64 + rcp_f32 + cvt_f64_f32 per value
    double a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < ITER; ++i) {
#define R32(x) { float f; asm volatile("v_cvt_f32_f64 %0, %1\n v_rcp_f32 %0, %0\n v_cvt_f64_f32 %1, %0\n" : "=&v"(f), "+v"(x)); }
        R32(a0) R32(a1) R32(a2) R32(a3) R32(a4) R32(a5) R32(a6) R32(a7)
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
    if (threadIdx.x == 0 && blockIdx.x == 0) out[gridDim.x * blockDim.x] = (double)(t1 - t0);
}

template <typename F>
void run(const char *name, F kern, int waves_per_simd, int instr_per_iter, double *out) {
    const int blocks = 256 * waves_per_simd;   // 256 threads = 4 waves = one per SIMD; blocks per CU = waves_per_simd
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    kern<<<blocks, 256>>>(out, 1.5);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int r = 0; r < 5; ++r) kern<<<blocks, 256>>>(out, 1.5);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double cyc; hipMemcpy(&cyc, out + (size_t)blocks * 256, 8, hipMemcpyDeviceToHost);
    double wave_instr_per_simd = (double)ITER * instr_per_iter * waves_per_simd;   // per launch
    double ns_per_instr = ms / 5 * 1e6 / wave_instr_per_simd;
    printf("%-10s waves/SIMD %d: %.2f ns per wave-instr per SIMD (wall); one wave: %.2f memtime ticks per instr\n", name,
           waves_per_simd, ns_per_instr, cyc / ((double)ITER * instr_per_iter));
}

int main() {
    double *out; hipMalloc(&out, sizeof(double) * (256 * 8 * 256 + 16));
    for (int w : {1, 2, 4}) {
        run("fma_f64", k_fma, w, 8, out);
        run("mul_f64", k_mul, w, 8, out);
        run("add_f64", k_add, w, 8, out);
        run("max_f64", k_max, w, 8, out);
        run("rcp_f64", k_rcp, w, 8, out);
        run("rsq_f64", k_rsq, w, 8, out);
        run("sqrt_f64", k_sqrt, w, 8, out);
        run("rcp32+2cvt", k_rcp32, w, 8, out);
    }
    return 0;
}
