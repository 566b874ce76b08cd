// pm_chi2.hip — chi-square descriptor-distance cost matrices.
// Reference: get_unary_distance (shape_context.py:88-99) evaluated for every (i, j) by the widget's
// eight N x M double loops (_dock_widget.py:547-602).
//
// Exactness contract: U[i][j] is bit-identical to the reference's scalar loop — float64, bins
// summed in index order k = 0..359, one rounding per operation, bins with a == b contributing
// nothing.  (a-b)^2/(a+b) is +0 when a == b != 0, and adding +0 to the non-negative running sum
// leaves its bits unchanged, so those bins are not branched on; a == b == 0 would be 0/0, so
// zeros of the B operand are replaced by 1e-300 while the tile is staged: with b = 1e-300,
// (a-b)^2/(a+b) equals a^2/a bit for bit when a != 0 (1e-300 is below half an ulp of any
// descriptor value >= 1/N), and (1e-300)^2 underflows to +0 when a == 0.
//
// Work decomposition (float64 VALU bound: 360 correctly rounded divisions per pair and matrix):
//   workgroup = 256 threads = 4 waves, tile = 16 rows (i) x 64 columns (j)
//   lane <-> column j, so the eight row-major stores of a wave are 512 contiguous bytes each
//   wave w owns rows 4w..4w+3 and NFA*NFB running sums per row (all statically indexed registers)
//   the 360 bins stream through LDS in 30 stages of 12 (one (r,theta) shell of 12 phi sectors):
//   B rows are read per lane (row pitch 112 B = 7 x 16 B, odd, so ds_read_b128 is conflict
//   free), A rows are wave-uniform broadcasts.
//   Tiles are numbered with i fastest and the block id is remapped so that each XCD walks a
//   contiguous run of tiles: concurrent workgroups of an XCD share one B tile in its L2.
#include "pm_common.h"

namespace pm {

constexpr int CH_THREADS = 256;
constexpr int CH_TI = 16;                 // rows per tile
constexpr int CH_TJ = 64;                 // columns per tile (one per lane)
constexpr int CH_RI = CH_TI / 4;          // rows per wave
constexpr int CH_K = 12;                  // bins per stage
constexpr int CH_STAGES = PM_NBINS / CH_K;
constexpr int CH_BPITCH = CH_K + 2;       // doubles; 112 B row pitch
constexpr double CH_TINY = 1e-300;

template <int NFA, int NFB>
struct Chi2Args {
    const double *a[NFA];
    const double *b[NFB];
};

template <int NFA, int NFB>
__global__ __launch_bounds__(CH_THREADS) void chi2_kernel(Chi2Args<NFA, NFB> args, int nA, int nB,
                                                          double *__restrict__ out, size_t ld, size_t mstride,
                                                          int nTi, unsigned int nblocks) {
    __shared__ __attribute__((aligned(16))) double A_s[NFA][CH_TI][CH_K];
    __shared__ __attribute__((aligned(16))) double B_s[NFB][CH_TJ][CH_BPITCH];

    // XCD-aware tile order: blocks b, b+8, b+16.. (one XCD) take consecutive tiles
    unsigned int bid = blockIdx.x;
    const unsigned int full = nblocks / 8u * 8u;
    if (bid < full) bid = (bid % 8u) * (full / 8u) + bid / 8u;
    const int ti = bid % (unsigned int)nTi, tj = bid / (unsigned int)nTi;
    const int i0 = ti * CH_TI, j0 = tj * CH_TJ;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

    double acc[CH_RI][NFA * NFB];
#pragma unroll
    for (int r = 0; r < CH_RI; ++r)
#pragma unroll
        for (int h = 0; h < NFA * NFB; ++h) acc[r][h] = 0.0;

    for (int g = 0; g < CH_STAGES; ++g) {
        __syncthreads();
        // stage A: NFA x 16 rows x 12 bins
        for (int e = tid; e < NFA * CH_TI * CH_K; e += CH_THREADS) {
            const int f = e / (CH_TI * CH_K), rem = e - f * (CH_TI * CH_K);
            const int r = rem / CH_K, k = rem - r * CH_K;
            const int gi = min(i0 + r, nA - 1);
            A_s[f][r][k] = args.a[f][(size_t)gi * PM_NBINS + g * CH_K + k];
        }
        // stage B: NFB x 64 rows x 12 bins as 16-byte pieces, zeros -> 1e-300
        for (int e = tid; e < NFB * CH_TJ * (CH_K / 2); e += CH_THREADS) {
            const int f = e / (CH_TJ * (CH_K / 2)), rem = e - f * (CH_TJ * (CH_K / 2));
            const int j = rem / (CH_K / 2), kk = rem - j * (CH_K / 2);
            const int gj = min(j0 + j, nB - 1);
            double2 v = *reinterpret_cast<const double2 *>(args.b[f] + (size_t)gj * PM_NBINS + g * CH_K + 2 * kk);
            v.x = (v.x == 0.0) ? CH_TINY : v.x;
            v.y = (v.y == 0.0) ? CH_TINY : v.y;
            *reinterpret_cast<double2 *>(&B_s[f][j][2 * kk]) = v;
        }
        __syncthreads();
#pragma unroll
        for (int fb = 0; fb < NFB; ++fb) {
            double b[CH_K];
#pragma unroll
            for (int k = 0; k < CH_K; k += 2) {
                double2 v = *reinterpret_cast<const double2 *>(&B_s[fb][lane][k]);
                b[k] = v.x; b[k + 1] = v.y;
            }
#pragma unroll
            for (int r = 0; r < CH_RI; ++r) {
#pragma unroll
                for (int fa = 0; fa < NFA; ++fa) {
                    const double *arow = &A_s[fa][wave * CH_RI + r][0];
                    double s = acc[r][fa * NFB + fb];
#pragma unroll
                    for (int k = 0; k < CH_K; ++k) {
                        const double a = arow[k];
                        const double df = a - b[k];
                        s = s + div_pos(df * df, a + b[k]);
                    }
                    acc[r][fa * NFB + fb] = s;
                }
            }
        }
    }
    const int gj = j0 + lane;
    if (gj < nB) {
#pragma unroll
        for (int r = 0; r < CH_RI; ++r) {
            const int gi = i0 + wave * CH_RI + r;
            if (gi < nA) {
#pragma unroll
                for (int h = 0; h < NFA * NFB; ++h) out[(size_t)h * mstride + (size_t)gi * ld + gj] = 0.5 * acc[r][h];
            }
        }
    }
}

template <int NFA, int NFB>
int chi2_launch(const Chi2Args<NFA, NFB> &args, int nA, int nB, double *out, size_t ld, size_t mstride, hipStream_t s) {
    const long nTi = ((long)nA + CH_TI - 1) / CH_TI, nTj = ((long)nB + CH_TJ - 1) / CH_TJ;
    const long nblocks = nTi * nTj;
    if (nblocks > 0x7fffffffL) return PM_ERR_INVALID_ARG;
    chi2_kernel<NFA, NFB><<<(unsigned int)nblocks, CH_THREADS, 0, s>>>(args, nA, nB, out, ld, mstride, (int)nTi, (unsigned int)nblocks);
    return launch_status();
}

}  // namespace pm

extern "C" int pm_chi2_cost(const double *scA, int nA, const double *scB, int nB, double *out, size_t ld, void *stream) {
    if (!scA || !scB || !out || nA <= 0 || nB <= 0 || ld < (size_t)nB) return PM_ERR_INVALID_ARG;
    if (((uintptr_t)scB & 15) != 0) return PM_ERR_INVALID_ARG;  // 16-byte loads of descriptor rows
    pm::Chi2Args<1, 1> args;
    args.a[0] = scA;
    args.b[0] = scB;
    return pm::chi2_launch<1, 1>(args, nA, nB, out, ld, 0, (hipStream_t)stream);
}

extern "C" int pm_chi2_cost8(const double *sc_m1, const double *sc_m2, int nM, const double *sc_f1, const double *sc_f2,
                             const double *sc_f3, const double *sc_f4, int nF, double *out, size_t ld,
                             size_t matrix_stride, void *stream) {
    if (!sc_m1 || !sc_m2 || !sc_f1 || !sc_f2 || !sc_f3 || !sc_f4 || !out) return PM_ERR_INVALID_ARG;
    if (nM <= 0 || nF <= 0 || ld < (size_t)nF || matrix_stride < (size_t)nM * ld) return PM_ERR_INVALID_ARG;
    if ((((uintptr_t)sc_f1 | (uintptr_t)sc_f2 | (uintptr_t)sc_f3 | (uintptr_t)sc_f4) & 15) != 0) return PM_ERR_INVALID_ARG;
    pm::Chi2Args<2, 4> args;
    args.a[0] = sc_m1; args.a[1] = sc_m2;
    args.b[0] = sc_f1; args.b[1] = sc_f2; args.b[2] = sc_f3; args.b[3] = sc_f4;
    return pm::chi2_launch<2, 4>(args, nM, nF, out, ld, matrix_stride, (hipStream_t)stream);
}
