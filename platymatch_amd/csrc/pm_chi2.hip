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
//   The block id is remapped so that each XCD walks a contiguous run of tiles, ordered in groups of 8
//   column tiles (tile_of): concurrent workgroups of an XCD share a few A and B tiles in its L2.
#include <algorithm>
#include <cstdlib>
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

// Tile order within an XCD's run of tiles: column tiles are taken in groups of CH_GJ; inside a group the row
// tile advances slowest.  The ~100 workgroups an XCD has in flight then cover ~12 row tiles x 8 column tiles:
// the 8 B tiles of the group stay hot in that XCD's L2 for the whole group and every A tile is fetched once per
// group instead of once per column tile (measured before this ordering: ~120 GB of A re-reads per 50k build).
constexpr int CH_GJ = 8;
__device__ __forceinline__ void tile_of(unsigned int lid, int nTi, unsigned int nblocks, int &ti, int &tj) {
    const unsigned int nTj = nblocks / (unsigned int)nTi;
    const unsigned int per_group = CH_GJ * (unsigned int)nTi;
    const unsigned int g = lid / per_group, rem = lid - g * per_group;
    const unsigned int gcols = min((unsigned int)CH_GJ, nTj - g * CH_GJ);
    ti = (int)(rem / gcols);
    tj = (int)(g * CH_GJ + rem % gcols);
}

// A HIP launch holds fewer than 2^32 work-items per grid dimension (gridDim.x * blockDim.x); beyond that this runtime neither
// refuses the launch nor runs all of it — 140 000 x 140 000 nuclei are 19.1 M tiles of 16 x 64 = 4.9e9 threads, and the filter
// matrix came back mostly unwritten (round 5: every row "violated" in every pricing round, tools/solve_trace.py).  The tile
// launchers therefore cut the tile rows into BANDS of at most CH_MAX_BLOCKS tiles, one launch per band on offset pointers.
// (PM_CHI2_MAX_BLOCKS in the environment lowers the cap: the tests force many bands on a small problem with it.)
constexpr long CH_MAX_BLOCKS = 0xffffffffL / CH_THREADS;
inline long max_blocks() {
    const char *e = getenv("PM_CHI2_MAX_BLOCKS");          // (read per launch: a test switches it within one process)
    const long v = e ? atol(e) : 0;
    return (v > 0 && v < CH_MAX_BLOCKS) ? v : CH_MAX_BLOCKS;
}
inline long band_tile_rows(long nTj) {
    const long cap = max_blocks();
    return nTj > cap ? 0 : cap / nTj;
}      // 0: one tile row alone is too wide

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
    int ti, tj;
    tile_of(bid, nTi, nblocks, ti, tj);
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
    const long band = band_tile_rows(nTj);
    if (band == 0) return PM_ERR_INVALID_ARG;
    for (long t0 = 0; t0 < nTi; t0 += band) {
        const long tiles = std::min(band, nTi - t0), r0 = t0 * CH_TI;
        const int rows = (int)std::min((long)nA - r0, tiles * CH_TI);
        Chi2Args<NFA, NFB> part = args;
        for (int f = 0; f < NFA; ++f) part.a[f] = args.a[f] + (size_t)r0 * PM_NBINS;
        const unsigned int nblocks = (unsigned int)(tiles * nTj);
        chi2_kernel<NFA, NFB><<<nblocks, CH_THREADS, 0, s>>>(part, rows, nB, out + (size_t)r0 * ld, ld, mstride, (int)tiles, nblocks);
    }
    return launch_status();
}

// ---- the eight hypothesis matrices from the frame-1 descriptors alone ----------------------------
// get_unary's frames 2..4 re-express every neighbour with (x, y) -> (-x, -y), (x, -y), (-x, y)
// (shape_context.py:172-175, 180-181): phi -> phi + pi, -phi, pi - phi.  Away from exact sector edges
// that permutes the 12 phi sectors of each (r, theta) shell:
//     sc2[q] = sc1[(q + 6) % 12]     sc3[q] = sc1[11 - q]     sc4[q] = sc1[(5 - q) mod 12]
// so U_ab = sum_q f(sc_a^m[q], sc_b^f[q]) only ever pairs A[p] (moving frame 1) with one of
// B[p], B[p+6], B[11-p], B[5-p] (fixed frame 1): four sets of terms, each summed in two orders —
// natural p = 0..11 (U11, U12, U13, U14) and rolled p = 6..11,0..5 (U22, U21, U24, U23).  Terms are
// computed once and added into both running sums in the reference's order, so all eight matrices stay
// bit-identical to the general kernel at half its divisions.  Whether the permutation relation holds
// for the given descriptor arrays is CHECKED bit for bit by symmetry_check_kernel, never assumed.
// TSEL = -1: all four pairings -> eight matrices in the widget's order; TSEL = t: pairing t alone -> its two matrices
// (natural order first, rolled order second): the cost build of ONE hypothesis and its twin, for clouds whose eight
// matrices do not fit in HBM together (each pairing's terms are exactly a quarter of the eight-matrix launch).
//
// Term table (TL > 0).  A descriptor value is count / total (get_shape_context: sc / sc.sum(), shape_context.py:40-43), so a
// term (a-b)^2/(a+b) is a function of two small integers and the two totals.  counts_* below recover the counts from the
// doubles and VERIFY, bit for bit, that every value is fl(count / total) — nothing is assumed; one failure and the launch
// computes every term as before.  Where all counts of a shell (both clouds) are below TL, the shell's 48 terms per pair
// come out of a TL x TL table in LDS, filled at the start of the workgroup by the SAME operations on the SAME operands
// (fl(ca/totA), fl(cb/totB) or 1e-300, div_pos) — identical bits by construction — at one ds_read_b64 + one address add per
// term instead of a 13-slot division.  The moving count is wave-uniform (a scalar row offset), the fixed count per lane,
// so a wave reads one table row: lanes with equal counts broadcast, counts 32 apart conflict.  Shells with larger counts
// (the outer ones of a large cloud) are computed.
constexpr int CH_NSHELL = CH_STAGES;
constexpr int CH_TL = 94;              // 94 x 94 doubles + the staging tiles = 79 392 B: two workgroups per CU (160 KiB), as the registers allow
struct SymMeta {
    unsigned long long minbits[2];     // smallest positive descriptor value of each cloud (bit pattern; +inf if none)
    double tot[2];                     // its reciprocal rounded to an integer: the candidate total
    int bad;                           // some value is not fl(count / total): no table
    int pad;
    int maxc[2][CH_NSHELL];            // largest count per (r, theta) shell, saturated at 255
};

// RELAX (round 4, opt-in, never the default): the same eight matrices WITHOUT the bit-identity contract — an experiment on what
// the exact formulation costs.  (a-b)^2/(a+b) = (a+b) - 4ab/(a+b), so
//     U = 0.5 sum_k (a_k - b_k)^2/(a_k + b_k) = 0.5 (sum a + sum b) - 2 sum_k a_k b_k / (a_k + b_k)
// and with the order of summation free the natural- and rolled-order twins coincide: FOUR running sums per row instead of
// eight, each term an add, a multiply, v_rcp_f64 + ONE Newton step (relative error 2^-48.8, no residual correction) and one
// fused multiply-add into the sum: 6 instructions / 9 issue slots per term against 12 / 15.  Row sums (sum a, sum b) come from
// a small pre-kernel (relaxed_rowsum_kernel).  Per-entry error against the exact value: <= PM_CHI2_RELAX_DELTA (absolute; the
// terms' 2e-15 relative, the 360 additions' roundings and the row sums'), which the caller's uniqueness certificate has to
// cover (lsap.certify(min_eps=...)).
#define PM_CHI2_RELAX_DELTA 2e-13

__device__ __forceinline__ double relaxed_recip(double s) {
    double r = __builtin_amdgcn_rcp(s);
    const double e = __builtin_fma(-s, r, 1.0);
    return __builtin_fma(r, e, r);
}

template <int SY_RI, int MINW, int TSEL = -1, int TL = 0, bool RELAX = false>   // rows per wave; min waves per SIMD for the register allocator; pairing; table size; relaxed arithmetic
__global__ __launch_bounds__(CH_THREADS, MINW) void chi2_sym_kernel(const double *__restrict__ scA, int nA,
                                                                 const double *__restrict__ scB, int nB,
                                                                 double *__restrict__ out, size_t ld, size_t mstride,
                                                                 int nTi, unsigned int nblocks,
                                                                 const unsigned char *__restrict__ cntA = nullptr,
                                                                 const unsigned char *__restrict__ cntB = nullptr,
                                                                 const SymMeta *__restrict__ meta = nullptr,
                                                                 const double *__restrict__ sumA = nullptr,
                                                                 const double *__restrict__ sumB = nullptr) {
    constexpr int SY_TI = 4 * SY_RI;      // rows per tile
    __shared__ __attribute__((aligned(16))) double A_s[SY_TI][CH_K];
    __shared__ __attribute__((aligned(16))) double B_s[CH_TJ][CH_BPITCH];
    __shared__ __attribute__((aligned(16))) double tab[TL > 0 ? TL * TL : 1];

    unsigned int bid = blockIdx.x;
    const unsigned int full = nblocks / 8u * 8u;
    if (bid < full) bid = (bid % 8u) * (full / 8u) + bid / 8u;
    int ti, tj;
    tile_of(bid, nTi, nblocks, ti, tj);
    const int i0 = ti * SY_TI, j0 = tj * CH_TJ;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

    constexpr int NH = (TSEL < 0) ? 8 : 2;
    constexpr int NACC = RELAX ? NH / 2 : NH;      // relaxed: one sum per pairing (the twins coincide)
    double acc[SY_RI][NACC];
#pragma unroll
    for (int r = 0; r < SY_RI; ++r)
#pragma unroll
        for (int h = 0; h < NACC; ++h) acc[r][h] = 0.0;

    unsigned int tabmask = 0;             // shells served from the table (uniform)
    if constexpr (TL > 0) {
        if (meta->bad == 0) {
            for (int g = 0; g < CH_STAGES; ++g)
                if (meta->maxc[0][g] < TL && meta->maxc[1][g] < TL) tabmask |= 1u << g;
        }
        tabmask = __builtin_amdgcn_readfirstlane(tabmask);
        if (tabmask) {
            const double totA = meta->tot[0], totB = meta->tot[1];
            for (int e = tid; e < TL * TL; e += CH_THREADS) {
                const int ca = e / TL, cb = e - ca * TL;
                const double a = (double)ca / totA, b = cb ? (double)cb / totB : CH_TINY;
                const double df = a - b;
                tab[e] = RELAX ? (a * b) * relaxed_recip(a + b) : div_pos(df * df, a + b);
            }
            __syncthreads();
        }
    }

    // counts of the next tabled shell, fetched one tabled shell ahead (an L2 round trip is about as long as a tabled shell):
    // the lane's twelve fixed counts (3 dwords) and, per row of the wave, the twelve moving counts (3 dwords, wave-uniform)
    unsigned int nb[3] = {0, 0, 0}, na[SY_RI][3];
#pragma unroll
    for (int r = 0; r < SY_RI; ++r) na[r][0] = na[r][1] = na[r][2] = 0;
    const unsigned int *pb0 = nullptr, *pa0[SY_RI];
    if constexpr (TL > 0) {
        pb0 = reinterpret_cast<const unsigned int *>(cntB + (size_t)min(j0 + lane, nB - 1) * PM_NBINS);
#pragma unroll
        for (int r = 0; r < SY_RI; ++r)
            pa0[r] = reinterpret_cast<const unsigned int *>(cntA + (size_t)min(i0 + wave * SY_RI + r, nA - 1) * PM_NBINS);
        if (tabmask) {
            const int g0 = __builtin_ctz(tabmask);
#pragma unroll
            for (int k = 0; k < 3; ++k) nb[k] = pb0[g0 * 3 + k];
#pragma unroll
            for (int r = 0; r < SY_RI; ++r)
#pragma unroll
                for (int k = 0; k < 3; ++k) na[r][k] = pa0[r][g0 * 3 + k];
        }
    }

    // PREFETCH (the exact kernels at two waves per SIMD): a thread's share of a computed stage's staging — one double of the tile's
    // moving rows (threads 0..191), one or two double2 of its fixed rows — travels through registers one computed stage ahead
    constexpr bool PF = !RELAX && MINW == 2 && SY_RI == 4;
    static_assert(!PF || (SY_TI * CH_K <= CH_THREADS && CH_TJ * (CH_K / 2) <= 2 * CH_THREADS), "one A value and at most two B pairs per thread");
    double pf_a = 0.0;
    double2 pf_b0 = {0.0, 0.0}, pf_b1 = {0.0, 0.0};
    const int pf_ia = (tid < SY_TI * CH_K) ? tid : -1;                                   // flat index into A_s[SY_TI][CH_K]
    const int pf_j0 = tid / (CH_K / 2), pf_k0 = 2 * (tid - pf_j0 * (CH_K / 2));
    const int pf_e1 = tid + CH_THREADS;
    const int pf_j1 = (pf_e1 < CH_TJ * (CH_K / 2)) ? pf_e1 / (CH_K / 2) : -1;
    const int pf_k1 = (pf_j1 >= 0) ? 2 * (pf_e1 - pf_j1 * (CH_K / 2)) : 0;
    auto prefetch = [&](int gs) {
        if (pf_ia >= 0) pf_a = scA[(size_t)min(i0 + pf_ia / CH_K, nA - 1) * PM_NBINS + gs * CH_K + (pf_ia % CH_K)];
        pf_b0 = *reinterpret_cast<const double2 *>(scB + (size_t)min(j0 + pf_j0, nB - 1) * PM_NBINS + gs * CH_K + pf_k0);
        if (pf_j1 >= 0) pf_b1 = *reinterpret_cast<const double2 *>(scB + (size_t)min(j0 + pf_j1, nB - 1) * PM_NBINS + gs * CH_K + pf_k1);
    };
    if constexpr (PF) {
        int g0 = 0;
        while (g0 < CH_STAGES && ((tabmask >> g0) & 1u)) ++g0;
        if (g0 < CH_STAGES) prefetch(g0);
    }

    for (int g = 0; g < CH_STAGES; ++g) {
        if constexpr (TL > 0) {
            if ((tabmask >> g) & 1u) {
                // this lane's twelve fixed counts of the shell, as byte offsets into a table row
                const unsigned int wb[3] = {nb[0], nb[1], nb[2]};
                unsigned int wav[SY_RI][3];
#pragma unroll
                for (int r = 0; r < SY_RI; ++r)
#pragma unroll
                    for (int k = 0; k < 3; ++k) wav[r][k] = na[r][k];
                const unsigned int later = tabmask >> g >> 1;
                if (later) {
                    const int gn = g + 1 + __builtin_ctz(later);
#pragma unroll
                    for (int k = 0; k < 3; ++k) nb[k] = pb0[gn * 3 + k];
#pragma unroll
                    for (int r = 0; r < SY_RI; ++r)
#pragma unroll
                        for (int k = 0; k < 3; ++k) na[r][k] = pa0[r][gn * 3 + k];
                }
                unsigned int cb8[CH_K];
#pragma unroll
                for (int k = 0; k < CH_K; ++k) cb8[k] = ((wb[k >> 2] >> (8 * (k & 3))) & 255u) << 3;
                const char *tbase = reinterpret_cast<const char *>(tab);
                unsigned int wa[SY_RI][3];
#pragma unroll
                for (int r = 0; r < SY_RI; ++r)
#pragma unroll
                    for (int k = 0; k < 3; ++k) wa[r][k] = __builtin_amdgcn_readfirstlane(wav[r][k]);
                // (row, pairing) groups of twelve lookups, software-pipelined by hand: the reads of group i + 1 are issued
                // before the 24 additions of group i, so that the LDS round trip hides behind them (the scheduler, left alone,
                // keeps four reads in flight and waits on each)
                constexpr int NT = (TSEL < 0) ? 4 : 1, NG = SY_RI * NT;
                double Tc[CH_K], Tn[CH_K];
                auto lookups = [&](int i, double (&T)[CH_K]) {
                    const int r = i / NT, t = (TSEL < 0) ? i % NT : TSEL;
#pragma unroll
                    for (int p = 0; p < CH_K; ++p) {
                        const int q = (t == 0) ? p : (t == 1) ? (p + 6) % 12 : (t == 2) ? 11 - p : (17 - p) % 12;
                        const unsigned int row = ((wa[r][p >> 2] >> (8 * (p & 3))) & 255u) * (unsigned int)(TL * 8);   // scalar
                        T[p] = *reinterpret_cast<const double *>(tbase + row + cb8[q]);
                    }
                };
                lookups(0, Tc);
#pragma unroll
                for (int i = 0; i < NG; ++i) {
                    if (i + 1 < NG) lookups(i + 1, Tn);
                    __builtin_amdgcn_sched_barrier(0);
                    const int r = i / NT, t = (TSEL < 0) ? i % NT : TSEL;
                    const int hn = (TSEL < 0) ? t : 0, hr = (TSEL >= 0) ? 1 : (t == 0) ? 5 : (t == 1) ? 4 : (t == 2) ? 7 : 6;
                    if constexpr (RELAX) {
                        double sn = acc[r][hn];
#pragma unroll
                        for (int p = 0; p < CH_K; ++p) sn = sn + Tc[p];
                        acc[r][hn] = sn;
                    } else {
                        double sn = acc[r][hn], sr = acc[r][hr];
#pragma unroll
                        for (int p = 0; p < CH_K; ++p) sn = sn + Tc[p];
#pragma unroll
                        for (int p = 0; p < CH_K; ++p) sr = sr + Tc[(p + 6) % 12];
                        acc[r][hn] = sn;
                        acc[r][hr] = sr;
                    }
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int p = 0; p < CH_K; ++p) Tc[p] = Tn[p];
                }
                continue;                 // uniform over the workgroup: nothing staged, no barrier
            }
        }
        if constexpr (PF) {
            // this stage's values were fetched into registers while the previous computed stage ran (or before the loop): park them
            // in LDS between two barriers, then start the NEXT computed stage's fetch — its L2 / HBM round trip runs under this
            // stage's ~10 000 cycles of arithmetic instead of in front of them (round 5: both workgroups of a CU used to meet their
            // load phases with nothing to issue; same values, same operations, same bits)
            __syncthreads();
            if (pf_ia >= 0) A_s[pf_ia / CH_K][pf_ia % CH_K] = pf_a;
            {
                double2 v = pf_b0;
                v.x = (v.x == 0.0) ? CH_TINY : v.x;
                v.y = (v.y == 0.0) ? CH_TINY : v.y;
                *reinterpret_cast<double2 *>(&B_s[pf_j0][pf_k0]) = v;
                if (pf_j1 >= 0) {
                    v = pf_b1;
                    v.x = (v.x == 0.0) ? CH_TINY : v.x;
                    v.y = (v.y == 0.0) ? CH_TINY : v.y;
                    *reinterpret_cast<double2 *>(&B_s[pf_j1][pf_k1]) = v;
                }
            }
            __syncthreads();
            int gn = g + 1;
            while (gn < CH_STAGES && ((tabmask >> gn) & 1u)) ++gn;
            if (gn < CH_STAGES) prefetch(gn);
        } else {
            __syncthreads();
            for (int e = tid; e < SY_TI * CH_K; e += CH_THREADS) {
                const int r = e / CH_K, k = e - r * CH_K;
                A_s[r][k] = scA[(size_t)min(i0 + r, nA - 1) * PM_NBINS + g * CH_K + k];
            }
            for (int e = tid; e < CH_TJ * (CH_K / 2); e += CH_THREADS) {
                const int j = e / (CH_K / 2), kk = e - j * (CH_K / 2);
                double2 v = *reinterpret_cast<const double2 *>(scB + (size_t)min(j0 + j, nB - 1) * PM_NBINS + g * CH_K + 2 * kk);
                v.x = (v.x == 0.0) ? CH_TINY : v.x;
                v.y = (v.y == 0.0) ? CH_TINY : v.y;
                *reinterpret_cast<double2 *>(&B_s[j][2 * kk]) = v;
            }
            __syncthreads();
        }
        double b[CH_K];
#pragma unroll
        for (int k = 0; k < CH_K; k += 2) {
            double2 v = *reinterpret_cast<const double2 *>(&B_s[lane][k]);
            b[k] = v.x; b[k + 1] = v.y;
        }
#pragma unroll
        for (int r = 0; r < SY_RI; ++r) {
            double a[CH_K];
#pragma unroll
            for (int k = 0; k < CH_K; k += 2) {
                double2 v = *reinterpret_cast<const double2 *>(&A_s[wave * SY_RI + r][k]);
                a[k] = v.x; a[k + 1] = v.y;
            }
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                if (TSEL >= 0 && t != TSEL) continue;
                // pairing t: B index for A index p; natural-order matrix, rolled-order matrix (widget numbering 0..7)
                const int hn = (TSEL < 0) ? t : 0, hr = (TSEL >= 0) ? 1 : (t == 0) ? 5 : (t == 1) ? 4 : (t == 2) ? 7 : 6;
                if constexpr (RELAX) {
                    // two independent chains per (row, pairing) so that the reciprocals' latency overlaps
                    double s0 = acc[r][hn], s1 = 0.0;
#pragma unroll
                    for (int p = 0; p < CH_K; p += 2) {
                        const int q0 = (t == 0) ? p : (t == 1) ? (p + 6) % 12 : (t == 2) ? 11 - p : (17 - p) % 12;
                        const int q1 = (t == 0) ? p + 1 : (t == 1) ? (p + 7) % 12 : (t == 2) ? 10 - p : (16 - p) % 12;
                        s0 = __builtin_fma(a[p] * b[q0], relaxed_recip(a[p] + b[q0]), s0);
                        s1 = __builtin_fma(a[p + 1] * b[q1], relaxed_recip(a[p + 1] + b[q1]), s1);
                    }
                    acc[r][hn] = s0 + s1;
                    continue;
                }
                double T[CH_K];
#pragma unroll
                for (int p = 0; p < CH_K; ++p) {
                    const int q = (t == 0) ? p : (t == 1) ? (p + 6) % 12 : (t == 2) ? 11 - p : (17 - p) % 12;
                    const double df = a[p] - b[q];
                    T[p] = div_pos(df * df, a[p] + b[q]);
                }
                double sn = acc[r][hn], sr = acc[r][hr];
#pragma unroll
                for (int p = 0; p < CH_K; ++p) sn = sn + T[p];
#pragma unroll
                for (int p = 0; p < CH_K; ++p) sr = sr + T[(p + 6) % 12];
                acc[r][hn] = sn;
                acc[r][hr] = sr;
            }
        }
    }
    const int gj = j0 + lane;
    if (gj < nB) {
#pragma unroll
        for (int r = 0; r < SY_RI; ++r) {
            const int gi = i0 + wave * SY_RI + r;
            if (gi < nA) {
                if constexpr (RELAX) {
                    const double half = 0.5 * (sumA[gi] + sumB[gj]);
#pragma unroll
                    for (int t = 0; t < NACC; ++t) {
                        const double u = __builtin_fma(-2.0, acc[r][t], half);
                        const int hr = (TSEL >= 0) ? 1 : (t == 0) ? 5 : (t == 1) ? 4 : (t == 2) ? 7 : 6;       // the twin: the same number
                        out[(size_t)t * mstride + (size_t)gi * ld + gj] = u;
                        out[(size_t)hr * mstride + (size_t)gi * ld + gj] = u;
                    }
                } else {
#pragma unroll
                    for (int h = 0; h < NH; ++h) out[(size_t)h * mstride + (size_t)gi * ld + gj] = 0.5 * acc[r][h];
                }
            }
        }
    }
}

// sum over a descriptor row's 360 bins, bin order (relaxed cost build: 0.5 (sum a + sum b) - 2 sum ab/(a+b))
__global__ __launch_bounds__(256) void relaxed_rowsum_kernel(const double *__restrict__ sc, int n, double *__restrict__ out) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const double *row = sc + (size_t)i * PM_NBINS;
    double s = 0.0;
    for (int k = 0; k < PM_NBINS; ++k) s += row[k];
    out[i] = s;
}

// Listed entries of ONE pairing's two matrices, exact: entry e = (rows[e], cols[e]) of the natural-order matrix (U11, U12, U13,
// U14 for pairing 0..3) and of its rolled-order twin (U22, U21, U24, U23) — the same operations in the same order as
// chi2_sym_kernel's computed path (and hence its table path), so the values carry the exact build's bits.  One thread per
// entry: what the relaxed mode's certificate evaluates the matched and the near-tight entries with (a few N of them).
__global__ __launch_bounds__(256) void entries_sym_kernel(const double *__restrict__ scA, int nA, const double *__restrict__ scB, int nB,
                                                          int t, const int32_t *__restrict__ rows, const int32_t *__restrict__ cols,
                                                          int n_entries, double *__restrict__ out_nat, double *__restrict__ out_rol) {
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= n_entries) return;
    const int i = rows[e], j = cols[e];
    if ((unsigned int)i >= (unsigned int)nA || (unsigned int)j >= (unsigned int)nB) {
        out_nat[e] = out_rol[e] = __builtin_nan("");
        return;
    }
    const double *a = scA + (size_t)i * PM_NBINS, *b = scB + (size_t)j * PM_NBINS;
    double sn = 0.0, sr = 0.0;
    for (int g = 0; g < CH_STAGES; ++g) {
        double T[CH_K];
#pragma unroll
        for (int p = 0; p < CH_K; ++p) {
            const int q = (t == 0) ? p : (t == 1) ? (p + 6) % 12 : (t == 2) ? 11 - p : (17 - p) % 12;
            double bq = b[g * CH_K + q];
            bq = (bq == 0.0) ? CH_TINY : bq;
            const double ap = a[g * CH_K + p];
            const double df = ap - bq;
            T[p] = div_pos(df * df, ap + bq);
        }
#pragma unroll
        for (int p = 0; p < CH_K; ++p) sn = sn + T[p];
#pragma unroll
        for (int p = 0; p < CH_K; ++p) sr = sr + T[(p + 6) % 12];
    }
    out_nat[e] = 0.5 * sn;
    out_rol[e] = 0.5 * sr;
}

// ---- FILTER build (round 4, opt-in): the four pairings' matrices in packed FLOAT32 arithmetic, written as float64 -------------
// U~ = 0.5 (sum a + sum b) - 2 sum_k a_k b_k / (a_k + b_k) with the terms in float32 (v_pk_add_f32 / v_pk_mul_f32 / v_pk_fma_f32 on two
// bins at a time, v_rcp_f32 per bin: 5.5 issue slots per term against the exact build's ~18), every shell computed, a shell's two
// float32 chains added into a float64 sum per (row, pairing).  NOT the reference's values: every entry is within PM_CHI2_FILTER_DELTA of
// the exact cost — per term 8 x 2^-24 relative (a, b rounded to float32: 2 each; a + b, a b: 1 each; v_rcp_f32: 2), i.e. <= 2.4e-7 on
// the sum (<= 0.5); the six fused adds and the final add of a shell's chains 7 x 2^-24 of that shell's sum, <= 2.1e-7 over all shells;
// doubled by the factor 2: < 1e-6; stored as float32 (pm_chi2_filter4_f32) another 2^-24 of an entry <= 1: PM_CHI2_FILTER_DELTA covers both.  It serves as a FILTER only (lsap.FilteredMatrix): it says which entries can matter, their exact
// costs come from entries_sym_kernel.  out4 + t * mstride = the matrix of pairing t (both U11/U22-type twins: one matrix).
#define PM_CHI2_FILTER_DELTA 1.1e-6
typedef float pm_f2 __attribute__((ext_vector_type(2)));

template <int TSEL, typename OUT>      // -1: the four pairings -> out + t * mstride; t: pairing t alone -> out.  OUT: float64 or float32 storage
__global__ __launch_bounds__(CH_THREADS, 2) void filter4_kernel(const double *__restrict__ scA, int nA, const double *__restrict__ scB, int nB,
                                                                OUT *__restrict__ out, size_t ld, size_t mstride, int nTi,
                                                                unsigned int nblocks, const double *__restrict__ sumA,
                                                                const double *__restrict__ sumB) {
    constexpr int RI = 4, TI = 4 * RI;
    __shared__ __attribute__((aligned(16))) float A_s[TI][CH_K];
    __shared__ __attribute__((aligned(16))) float B_s[CH_TJ][CH_K + 2];
    unsigned int bid = blockIdx.x;
    const unsigned int full = nblocks / 8u * 8u;
    if (bid < full) bid = (bid % 8u) * (full / 8u) + bid / 8u;
    int ti, tj;
    tile_of(bid, nTi, nblocks, ti, tj);
    const int i0 = ti * TI, j0 = tj * CH_TJ;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    constexpr int NT = TSEL < 0 ? 4 : 1;
    double acc[RI][NT];
#pragma unroll
    for (int r = 0; r < RI; ++r)
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[r][t] = 0.0;
    for (int g = 0; g < CH_STAGES; ++g) {
        __syncthreads();
        if (tid < TI * CH_K) {
            const int r = tid / CH_K, k = tid - r * CH_K;
            A_s[r][k] = (float)scA[(size_t)min(i0 + r, nA - 1) * PM_NBINS + g * CH_K + k];
        }
        for (int e = tid; e < CH_TJ * CH_K; e += CH_THREADS) {
            const int j = e / CH_K, k = e - j * CH_K;
            const float v = (float)scB[(size_t)min(j0 + j, nB - 1) * PM_NBINS + g * CH_K + k];
            B_s[j][k] = (v == 0.f) ? 1e-30f : v;
        }
        __syncthreads();
        pm_f2 b[6], br[6];                            // the lane's twelve fixed bins as pairs, and the same pairs swapped
#pragma unroll
        for (int k = 0; k < 6; ++k) {
            b[k] = *reinterpret_cast<const pm_f2 *>(&B_s[lane][2 * k]);
            br[k] = (pm_f2){b[k].y, b[k].x};
        }
#pragma unroll
        for (int r = 0; r < RI; ++r) {
            pm_f2 a[6], s4[NT];
#pragma unroll
            for (int k = 0; k < 6; ++k) a[k] = *reinterpret_cast<const pm_f2 *>(&A_s[wave * RI + r][2 * k]);
#pragma unroll
            for (int t = 0; t < NT; ++t) s4[t] = (pm_f2){0.f, 0.f};
#pragma unroll
            for (int k = 0; k < 6; ++k) {
                // moving bins (2k, 2k+1) meet fixed bins: pairing 0 the same; 1: +6; 2: (11-2k, 10-2k) = pair 5-k swapped;
                // 3: (17-2k, 16-2k) mod 12 = pair (8-k) mod 6 swapped
                const pm_f2 q[4] = {b[k], b[(k + 3) % 6], br[5 - k], br[(8 - k) % 6]};
#pragma unroll
                for (int tt = 0; tt < NT; ++tt) {
                    const int t = TSEL < 0 ? tt : TSEL;
                    const pm_f2 sm = a[k] + q[t];
                    const pm_f2 pr = a[k] * q[t];
                    const pm_f2 rc = (pm_f2){__builtin_amdgcn_rcpf(sm.x), __builtin_amdgcn_rcpf(sm.y)};
                    s4[tt] = __builtin_elementwise_fma(pr, rc, s4[tt]);
                }
            }
#pragma unroll
            for (int t = 0; t < NT; ++t) acc[r][t] += (double)(s4[t].x + s4[t].y);
        }
    }
    const int gj = j0 + lane;
    if (gj < nB) {
#pragma unroll
        for (int r = 0; r < RI; ++r) {
            const int gi = i0 + wave * RI + r;
            if (gi < nA) {
                const double half = 0.5 * (sumA[gi] + sumB[gj]);
#pragma unroll
                for (int t = 0; t < NT; ++t) out[(size_t)t * mstride + (size_t)gi * ld + gj] = (OUT)__builtin_fma(-2.0, acc[r][t], half);
            }
        }
    }
}

// flag[0] |= 1 unless, bit for bit, sc2 = roll6(sc1), sc3 = reverse(sc1), sc4 = (5-q)(sc1) within every shell.
// which: 0 = moving (frame 2 only), 1 = fixed (frames 2, 3, 4).  One thread per (row, bin).
__global__ __launch_bounds__(256) void symmetry_check_kernel(const unsigned long long *__restrict__ s1,
                                                             const unsigned long long *__restrict__ s2,
                                                             const unsigned long long *__restrict__ s3,
                                                             const unsigned long long *__restrict__ s4, int n,
                                                             int *__restrict__ flag) {
    const size_t e = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= (size_t)n * PM_NBINS) return;
    const int k = (int)(e % PM_NBINS), q = k % 12;
    const size_t base = e - q;
    bool bad = s2[e] != s1[base + (q + 6) % 12];
    if (s3) bad = bad || s3[e] != s1[base + 11 - q] || s4[e] != s1[base + (17 - q) % 12];
    if (bad) atomicOr(flag, 1);
}

}  // namespace pm

extern "C" int pm_chi2_symmetry_check(const double *sc_m1, const double *sc_m2, int nM, const double *sc_f1,
                                      const double *sc_f2, const double *sc_f3, const double *sc_f4, int nF, int32_t *flag1,
                                      void *stream) {
    if (!sc_m1 || !sc_m2 || !sc_f1 || !sc_f2 || !sc_f3 || !sc_f4 || !flag1 || nM <= 0 || nF <= 0) return PM_ERR_INVALID_ARG;
    hipStream_t s = (hipStream_t)stream;
    if (hipMemsetAsync(flag1, 0, sizeof(int32_t), s) != hipSuccess) return pm::launch_status();
    typedef const unsigned long long *U;
    const size_t em = (size_t)nM * PM_NBINS, ef = (size_t)nF * PM_NBINS;
    pm::symmetry_check_kernel<<<(unsigned int)((em + 255) / 256), 256, 0, s>>>((U)sc_m1, (U)sc_m2, nullptr, nullptr, nM, flag1);
    pm::symmetry_check_kernel<<<(unsigned int)((ef + 255) / 256), 256, 0, s>>>((U)sc_f1, (U)sc_f2, (U)sc_f3, (U)sc_f4, nF, flag1);
    return pm::launch_status();
}

namespace pm {
template <int RI, int MINW, int TSEL = -1, int TL = 0, bool RELAX = false>
int chi2_sym_launch(const double *sc_m1, int nM, const double *sc_f1, int nF, double *out, size_t ld, size_t mstride, hipStream_t s,
                    const unsigned char *cntA = nullptr, const unsigned char *cntB = nullptr, const SymMeta *meta = nullptr,
                    const double *sumA = nullptr, const double *sumB = nullptr) {
    const long nTi = ((long)nM + 4 * RI - 1) / (4 * RI), nTj = ((long)nF + CH_TJ - 1) / CH_TJ;
    const long band = band_tile_rows(nTj);
    if (band == 0) return PM_ERR_INVALID_ARG;
    if (TL > 0 && (!cntA || !cntB || !meta)) return PM_ERR_INVALID_ARG;
    if (RELAX && (!sumA || !sumB)) return PM_ERR_INVALID_ARG;
    for (long t0 = 0; t0 < nTi; t0 += band) {              // (one band up to ~130 000 x 130 000; see CH_MAX_BLOCKS)
        const long tiles = std::min(band, nTi - t0), r0 = t0 * 4 * RI;
        const int rows = (int)std::min((long)nM - r0, tiles * 4 * RI);
        const unsigned int nblocks = (unsigned int)(tiles * nTj);
        chi2_sym_kernel<RI, MINW, TSEL, TL, RELAX><<<nblocks, CH_THREADS, 0, s>>>(
            sc_m1 + (size_t)r0 * PM_NBINS, rows, sc_f1, nF, out + (size_t)r0 * ld, ld, mstride, (int)tiles, nblocks,
            cntA ? cntA + (size_t)r0 * PM_NBINS : nullptr, cntB, meta, sumA ? sumA + r0 : nullptr, sumB);
    }
    return launch_status();
}

// ---- counts behind the descriptor values (for the term table) -------------------------------------
__global__ void sym_meta_init_kernel(SymMeta *m) {
    const int t = threadIdx.x;
    if (t < 2) {
        m->minbits[t] = 0x7ff0000000000000ull;
        m->tot[t] = 0.0;
    }
    if (t == 0) m->bad = 0, m->pad = 0;
    if (t < CH_NSHELL) m->maxc[0][t] = m->maxc[1][t] = 0;
}

// smallest positive value (count 1 of some bin, in any cloud with a singly occupied bin: 1 / total); anything that is not a
// finite value >= 0 rules the table out
__global__ __launch_bounds__(256) void counts_min_kernel(const double *__restrict__ sc, size_t n_entries, SymMeta *m, int which) {
    unsigned long long best = 0x7ff0000000000000ull;
    bool bad = false;
    for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < n_entries; e += (size_t)gridDim.x * 256) {
        const double v = sc[e];
        if (!(v >= 0.0) || v > 1.0) bad = true;
        else if (v > 0.0) best = min(best, (unsigned long long)__double_as_longlong(v));   // positive doubles order like their bits
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) best = min(best, (unsigned long long)__shfl_down((unsigned long long)best, off, PM_WAVE));
    if ((threadIdx.x & 63) == 0) atomicMin(&m->minbits[which], best);
    if (bad) atomicOr(&m->bad, 1);
}

__global__ void counts_total_kernel(SymMeta *m) {
    const int w = threadIdx.x;
    if (w >= 2) return;
    const double v = __longlong_as_double((long long)m->minbits[w]);
    double tot = 0.0;
    if (v > 0.0 && v <= 1.0) tot = rint(1.0 / v);
    if (!(tot >= 1.0 && tot <= 2147483647.0)) {
        tot = 1.0;
        atomicOr(&m->bad, 1);
    }
    m->tot[w] = tot;
}

// count = rint(value * total), accepted only if fl(count / total) IS the value; per-shell maxima for the kernel's choice.
// One thread per (row, shell): its twelve values are 96 contiguous bytes, its twelve counts three dwords, and the shell's maximum
// costs ONE LDS atomic per thread (round 4, second session; one thread and one LDS atomic per VALUE before: 0.88 ms per 50 000-row
// cloud, the 64 lanes of a wave queuing on five or six addresses).
__global__ __launch_bounds__(256) void counts_extract_kernel(const double *__restrict__ sc, size_t n_groups, unsigned char *__restrict__ cnt,
                                                             SymMeta *m, int which) {
    __shared__ int smax[CH_NSHELL];
    if (threadIdx.x < CH_NSHELL) smax[threadIdx.x] = 0;
    __syncthreads();
    const size_t g = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (g < n_groups) {
        const double tot = m->tot[which];
        const double2 *src = reinterpret_cast<const double2 *>(sc + g * CH_K);
        double v[CH_K];
#pragma unroll
        for (int k = 0; k < CH_K / 2; ++k) { const double2 t = src[k]; v[2 * k] = t.x; v[2 * k + 1] = t.y; }
        unsigned int packed[CH_K / 4] = {0u, 0u, 0u};
        int top = 0;
        bool all_ok = true;
#pragma unroll
        for (int k = 0; k < CH_K; ++k) {
            const double c = rint(v[k] * tot);
            const bool ok = c >= 0.0 && c <= 2147483647.0 && c / tot == v[k];
            all_ok = all_ok && ok;
            const int cc = ok ? (int)min(c, 255.0) : 255;
            packed[k >> 2] |= (unsigned int)cc << (8 * (k & 3));
            top = max(top, cc);
        }
        if (!all_ok) atomicOr(&m->bad, 1);
        unsigned int *dst = reinterpret_cast<unsigned int *>(cnt + g * CH_K);
        dst[0] = packed[0]; dst[1] = packed[1]; dst[2] = packed[2];
        atomicMax(&smax[(int)(g % CH_NSHELL)], top);
    }
    __syncthreads();
    if (threadIdx.x < CH_NSHELL && smax[threadIdx.x] > 0) atomicMax(&m->maxc[which][threadIdx.x], smax[threadIdx.x]);
}

constexpr size_t SYM_META_BYTES = 512;
static_assert(sizeof(SymMeta) <= SYM_META_BYTES, "workspace header");

struct SymWs {
    SymMeta *meta;
    unsigned char *cntA, *cntB;
};

int sym_prepare(const double *sc_m1, int nM, const double *sc_f1, int nF, void *ws, size_t ws_bytes, hipStream_t s, SymWs &w) {
    if (!ws || ((uintptr_t)ws & 15) != 0 || ws_bytes < pm_chi2_sym_workspace_bytes(nM, nF)) return PM_ERR_WORKSPACE;
    char *base = (char *)ws;
    w.meta = (SymMeta *)base;
    w.cntA = (unsigned char *)(base + SYM_META_BYTES);
    w.cntB = w.cntA + align_up((size_t)nM * PM_NBINS, 16);
    const size_t eA = (size_t)nM * PM_NBINS, eB = (size_t)nF * PM_NBINS;
    sym_meta_init_kernel<<<1, 64, 0, s>>>(w.meta);
    counts_min_kernel<<<(unsigned int)min((eA + 255) / 256, (size_t)4096), 256, 0, s>>>(sc_m1, eA, w.meta, 0);
    counts_min_kernel<<<(unsigned int)min((eB + 255) / 256, (size_t)4096), 256, 0, s>>>(sc_f1, eB, w.meta, 1);
    counts_total_kernel<<<1, 64, 0, s>>>(w.meta);
    const size_t gA = eA / CH_K, gB = eB / CH_K;              // (row, shell) groups: 360 = 30 x 12
    counts_extract_kernel<<<(unsigned int)((gA + 255) / 256), 256, 0, s>>>(sc_m1, gA, w.cntA, w.meta, 0);
    counts_extract_kernel<<<(unsigned int)((gB + 255) / 256), 256, 0, s>>>(sc_f1, gB, w.cntB, w.meta, 1);
    return launch_status();
}
}  // namespace pm

extern "C" size_t pm_chi2_sym_workspace_bytes(int nM, int nF) {
    if (nM <= 0 || nF <= 0) return 0;
    return pm::SYM_META_BYTES + pm::align_up((size_t)nM * PM_NBINS, 16) + pm::align_up((size_t)nF * PM_NBINS, 16);
}

// variant: 0 = 64 x 64 table, 1 = CH_TL x CH_TL (the product's), 2 = no table (the plain kernel; the workspace is still prepared)
extern "C" int pm_chi2_cost8_sym_ws_variant(const double *sc_m1, int nM, const double *sc_f1, int nF, double *out, size_t ld,
                                            size_t matrix_stride, void *ws, size_t ws_bytes, int variant, void *stream) {
    if (!sc_m1 || !sc_f1 || !out || nM <= 0 || nF <= 0 || ld < (size_t)nF || matrix_stride < (size_t)nM * ld)
        return PM_ERR_INVALID_ARG;
    if (((uintptr_t)sc_f1 & 15) != 0 || ((uintptr_t)sc_m1 & 15) != 0) return PM_ERR_INVALID_ARG;
    hipStream_t s = (hipStream_t)stream;
    pm::SymWs w;
    const int rc = pm::sym_prepare(sc_m1, nM, sc_f1, nF, ws, ws_bytes, s, w);
    if (rc != PM_OK) return rc;
    switch (variant) {
        case 0: return pm::chi2_sym_launch<4, 2, -1, 64>(sc_m1, nM, sc_f1, nF, out, ld, matrix_stride, s, w.cntA, w.cntB, w.meta);
        case 1: return pm::chi2_sym_launch<4, 2, -1, pm::CH_TL>(sc_m1, nM, sc_f1, nF, out, ld, matrix_stride, s, w.cntA, w.cntB, w.meta);
        case 2: return pm::chi2_sym_launch<4, 2>(sc_m1, nM, sc_f1, nF, out, ld, matrix_stride, s);
        case 3: return pm::chi2_sym_launch<4, 3, -1, 64>(sc_m1, nM, sc_f1, nF, out, ld, matrix_stride, s, w.cntA, w.cntB, w.meta);
        case 4: return pm::chi2_sym_launch<4, 3, -1, 48>(sc_m1, nM, sc_f1, nF, out, ld, matrix_stride, s, w.cntA, w.cntB, w.meta);
        case 5:      // the 88-table kernel with the table ruled out: what its computed shells cost at its occupancy
            if (hipMemsetAsync(&w.meta->bad, 1, 1, s) != hipSuccess) return pm::launch_status();
            return pm::chi2_sym_launch<4, 2, -1, pm::CH_TL>(sc_m1, nM, sc_f1, nF, out, ld, matrix_stride, s, w.cntA, w.cntB, w.meta);
        default: return PM_ERR_INVALID_ARG;
    }
}

// which shells the last *_ws launch on this workspace served from its table (a report for benchmarks and tests; synchronises)
extern "C" int pm_chi2_sym_table_info(const void *ws, int32_t *tabled30, int32_t *table_size, void *stream) {
    if (!ws || !tabled30 || !table_size) return PM_ERR_INVALID_ARG;
    pm::SymMeta m;
    hipStream_t s = (hipStream_t)stream;
    if (hipMemcpyAsync(&m, ws, sizeof(m), hipMemcpyDeviceToHost, s) != hipSuccess || hipStreamSynchronize(s) != hipSuccess)
    {
        pm::launch_status();
        return PM_ERR_LAUNCH;
    }
    for (int g = 0; g < pm::CH_NSHELL; ++g)
        tabled30[g] = (m.bad == 0 && m.maxc[0][g] < pm::CH_TL && m.maxc[1][g] < pm::CH_TL) ? 1 : 0;
    *table_size = pm::CH_TL;
    return PM_OK;
}

// The relaxed build (opt-in experiment; see RELAX above).  Workspace: pm_chi2_relaxed_workspace_bytes = the term-table workspace +
// the two row-sum vectors.  variant 0: every shell computed; 1: shells with small counts from a (relaxed) term table.
extern "C" size_t pm_chi2_relaxed_workspace_bytes(int nM, int nF) {
    if (nM <= 0 || nF <= 0) return 0;
    return pm::align_up(pm_chi2_sym_workspace_bytes(nM, nF), 256) + pm::align_up((size_t)nM * 8, 256) + pm::align_up((size_t)nF * 8, 256);
}

extern "C" double pm_chi2_relaxed_delta(void) { return PM_CHI2_RELAX_DELTA; }

extern "C" int pm_chi2_cost8_relaxed(const double *sc_m1, int nM, const double *sc_f1, int nF, double *out, size_t ld,
                                     size_t matrix_stride, void *ws, size_t ws_bytes, int variant, void *stream) {
    if (!sc_m1 || !sc_f1 || !out || nM <= 0 || nF <= 0 || ld < (size_t)nF || matrix_stride < (size_t)nM * ld)
        return PM_ERR_INVALID_ARG;
    if (((uintptr_t)sc_f1 & 15) != 0 || ((uintptr_t)sc_m1 & 15) != 0) return PM_ERR_INVALID_ARG;
    if (!ws || ((uintptr_t)ws & 15) != 0 || ws_bytes < pm_chi2_relaxed_workspace_bytes(nM, nF)) return PM_ERR_WORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    const size_t tab_bytes = pm::align_up(pm_chi2_sym_workspace_bytes(nM, nF), 256);
    double *sumA = (double *)((char *)ws + tab_bytes);
    double *sumB = (double *)((char *)sumA + pm::align_up((size_t)nM * 8, 256));
    pm::relaxed_rowsum_kernel<<<(nM + 255) / 256, 256, 0, s>>>(sc_m1, nM, sumA);
    pm::relaxed_rowsum_kernel<<<(nF + 255) / 256, 256, 0, s>>>(sc_f1, nF, sumB);
    if (variant == 0)
        return pm::chi2_sym_launch<4, 2, -1, 0, true>(sc_m1, nM, sc_f1, nF, out, ld, matrix_stride, s, nullptr, nullptr, nullptr, sumA, sumB);
    if (variant != 1 && variant != 2) return PM_ERR_INVALID_ARG;
    pm::SymWs w;
    const int rc = pm::sym_prepare(sc_m1, nM, sc_f1, nF, ws, tab_bytes, s, w);
    if (rc != PM_OK) return rc;
    if (variant == 2)
        return pm::chi2_sym_launch<4, 3, -1, 64, true>(sc_m1, nM, sc_f1, nF, out, ld, matrix_stride, s, w.cntA, w.cntB, w.meta, sumA, sumB);
    return pm::chi2_sym_launch<4, 2, -1, pm::CH_TL, true>(sc_m1, nM, sc_f1, nF, out, ld, matrix_stride, s, w.cntA, w.cntB, w.meta, sumA, sumB);
}

// The filter build (see filter4_kernel).  Workspace: pm_chi2_filter_workspace_bytes = the two row-sum vectors.
extern "C" size_t pm_chi2_filter_workspace_bytes(int nM, int nF) {
    if (nM <= 0 || nF <= 0) return 0;
    return pm::align_up((size_t)nM * 8, 256) + pm::align_up((size_t)nF * 8, 256);
}

extern "C" double pm_chi2_filter_delta(void) { return PM_CHI2_FILTER_DELTA; }

namespace pm {
template <typename OUT>
static int filter_launch(const double *sc_m1, int nM, const double *sc_f1, int nF, int pairing, OUT *out, size_t ld, size_t matrix_stride,
                         void *ws, size_t ws_bytes, void *stream) {
    if (!sc_m1 || !sc_f1 || !out || nM <= 0 || nF <= 0 || ld < (size_t)nF || (pairing < 0 && matrix_stride < (size_t)nM * ld))
        return PM_ERR_INVALID_ARG;
    if (!ws || ((uintptr_t)ws & 15) != 0 || ws_bytes < pm_chi2_filter_workspace_bytes(nM, nF)) return PM_ERR_WORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    double *sumA = (double *)ws;
    double *sumB = (double *)((char *)ws + align_up((size_t)nM * 8, 256));
    relaxed_rowsum_kernel<<<(nM + 255) / 256, 256, 0, s>>>(sc_m1, nM, sumA);
    relaxed_rowsum_kernel<<<(nF + 255) / 256, 256, 0, s>>>(sc_f1, nF, sumB);
    const long nTi = ((long)nM + 15) / 16, nTj = ((long)nF + CH_TJ - 1) / CH_TJ;
    const long band = band_tile_rows(nTj);
    if (band == 0 || pairing < -1 || pairing > 3) return PM_ERR_INVALID_ARG;
    for (long t0 = 0; t0 < nTi; t0 += band) {              // (one band up to ~130 000 x 130 000; see CH_MAX_BLOCKS)
        const long tiles = std::min(band, nTi - t0), r0 = t0 * 16;
        const int rows = (int)std::min((long)nM - r0, tiles * 16);
        const unsigned int grid = (unsigned int)(tiles * nTj);
        const double *a = sc_m1 + (size_t)r0 * PM_NBINS, *sa = sumA + r0;
        OUT *o = out + (size_t)r0 * ld;
        switch (pairing) {
            case -1: filter4_kernel<-1, OUT><<<grid, CH_THREADS, 0, s>>>(a, rows, sc_f1, nF, o, ld, matrix_stride, (int)tiles, grid, sa, sumB); break;
            case 0: filter4_kernel<0, OUT><<<grid, CH_THREADS, 0, s>>>(a, rows, sc_f1, nF, o, ld, 0, (int)tiles, grid, sa, sumB); break;
            case 1: filter4_kernel<1, OUT><<<grid, CH_THREADS, 0, s>>>(a, rows, sc_f1, nF, o, ld, 0, (int)tiles, grid, sa, sumB); break;
            case 2: filter4_kernel<2, OUT><<<grid, CH_THREADS, 0, s>>>(a, rows, sc_f1, nF, o, ld, 0, (int)tiles, grid, sa, sumB); break;
            default: filter4_kernel<3, OUT><<<grid, CH_THREADS, 0, s>>>(a, rows, sc_f1, nF, o, ld, 0, (int)tiles, grid, sa, sumB); break;
        }
    }
    return launch_status();
}
}  // namespace pm

extern "C" int pm_chi2_filter4(const double *sc_m1, int nM, const double *sc_f1, int nF, double *out4, size_t ld, size_t matrix_stride,
                               void *ws, size_t ws_bytes, void *stream) {
    return pm::filter_launch(sc_m1, nM, sc_f1, nF, -1, out4, ld, matrix_stride, ws, ws_bytes, stream);
}

// the same matrices stored as float32: half the memory and half the traffic of the solver's dense passes (pm_lsap_*_f32)
extern "C" int pm_chi2_filter4_f32(const double *sc_m1, int nM, const double *sc_f1, int nF, float *out4, size_t ld, size_t matrix_stride,
                                   void *ws, size_t ws_bytes, void *stream) {
    return pm::filter_launch(sc_m1, nM, sc_f1, nF, -1, out4, ld, matrix_stride, ws, ws_bytes, stream);
}

extern "C" int pm_chi2_filter_pair_f32(const double *sc_m1, int nM, const double *sc_f1, int nF, int pairing, float *out1, size_t ld,
                                       void *ws, size_t ws_bytes, void *stream) {
    if (pairing < 0 || pairing > 3) return PM_ERR_INVALID_ARG;
    return pm::filter_launch(sc_m1, nM, sc_f1, nF, pairing, out1, ld, 0, ws, ws_bytes, stream);
}

// one pairing's filter matrix alone (a quarter of the launch): for clouds whose four filter matrices do not fit in HBM together
extern "C" int pm_chi2_filter_pair(const double *sc_m1, int nM, const double *sc_f1, int nF, int pairing, double *out1, size_t ld,
                                   void *ws, size_t ws_bytes, void *stream) {
    if (pairing < 0 || pairing > 3) return PM_ERR_INVALID_ARG;
    return pm::filter_launch(sc_m1, nM, sc_f1, nF, pairing, out1, ld, 0, ws, ws_bytes, stream);
}

extern "C" int pm_chi2_entries_sym(const double *sc_m1, int nM, const double *sc_f1, int nF, int pairing, const int32_t *rows,
                                   const int32_t *cols, int n_entries, double *out_natural, double *out_rolled, void *stream) {
    if (!sc_m1 || !sc_f1 || nM <= 0 || nF <= 0 || pairing < 0 || pairing > 3 || n_entries < 0) return PM_ERR_INVALID_ARG;
    if (n_entries == 0) return PM_OK;
    if (!rows || !cols || !out_natural || !out_rolled) return PM_ERR_INVALID_ARG;
    pm::entries_sym_kernel<<<(unsigned int)((n_entries + 255) / 256), 256, 0, (hipStream_t)stream>>>(sc_m1, nM, sc_f1, nF, pairing, rows, cols,
                                                                                                 n_entries, out_natural, out_rolled);
    return pm::launch_status();
}

extern "C" int pm_chi2_cost8_sym_ws(const double *sc_m1, int nM, const double *sc_f1, int nF, double *out, size_t ld,
                                    size_t matrix_stride, void *ws, size_t ws_bytes, void *stream) {
    return pm_chi2_cost8_sym_ws_variant(sc_m1, nM, sc_f1, nF, out, ld, matrix_stride, ws, ws_bytes, 1, stream);
}

extern "C" int pm_chi2_cost_pair_sym_ws(const double *sc_m1, int nM, const double *sc_f1, int nF, int pairing, double *out2, size_t ld,
                                        size_t matrix_stride, void *ws, size_t ws_bytes, void *stream) {
    if (!sc_m1 || !sc_f1 || !out2 || nM <= 0 || nF <= 0 || ld < (size_t)nF || matrix_stride < (size_t)nM * ld)
        return PM_ERR_INVALID_ARG;
    if (((uintptr_t)sc_f1 & 15) != 0 || ((uintptr_t)sc_m1 & 15) != 0) return PM_ERR_INVALID_ARG;
    if (pairing < 0 || pairing > 3) return PM_ERR_INVALID_ARG;
    hipStream_t s = (hipStream_t)stream;
    pm::SymWs w;
    const int rc = pm::sym_prepare(sc_m1, nM, sc_f1, nF, ws, ws_bytes, s, w);
    if (rc != PM_OK) return rc;
    switch (pairing) {
        case 0: return pm::chi2_sym_launch<4, 2, 0, pm::CH_TL>(sc_m1, nM, sc_f1, nF, out2, ld, matrix_stride, s, w.cntA, w.cntB, w.meta);
        case 1: return pm::chi2_sym_launch<4, 2, 1, pm::CH_TL>(sc_m1, nM, sc_f1, nF, out2, ld, matrix_stride, s, w.cntA, w.cntB, w.meta);
        case 2: return pm::chi2_sym_launch<4, 2, 2, pm::CH_TL>(sc_m1, nM, sc_f1, nF, out2, ld, matrix_stride, s, w.cntA, w.cntB, w.meta);
        default: return pm::chi2_sym_launch<4, 2, 3, pm::CH_TL>(sc_m1, nM, sc_f1, nF, out2, ld, matrix_stride, s, w.cntA, w.cntB, w.meta);
    }
}

extern "C" int pm_chi2_cost_pair_sym(const double *sc_m1, int nM, const double *sc_f1, int nF, int pairing, double *out2, size_t ld,
                                     size_t matrix_stride, void *stream) {
    if (!sc_m1 || !sc_f1 || !out2 || nM <= 0 || nF <= 0 || ld < (size_t)nF || matrix_stride < (size_t)nM * ld)
        return PM_ERR_INVALID_ARG;
    if (((uintptr_t)sc_f1 & 15) != 0 || ((uintptr_t)sc_m1 & 15) != 0) return PM_ERR_INVALID_ARG;
    hipStream_t s = (hipStream_t)stream;
    switch (pairing) {
        case 0: return pm::chi2_sym_launch<4, 2, 0>(sc_m1, nM, sc_f1, nF, out2, ld, matrix_stride, s);
        case 1: return pm::chi2_sym_launch<4, 2, 1>(sc_m1, nM, sc_f1, nF, out2, ld, matrix_stride, s);
        case 2: return pm::chi2_sym_launch<4, 2, 2>(sc_m1, nM, sc_f1, nF, out2, ld, matrix_stride, s);
        case 3: return pm::chi2_sym_launch<4, 2, 3>(sc_m1, nM, sc_f1, nF, out2, ld, matrix_stride, s);
        default: return PM_ERR_INVALID_ARG;
    }
}

// tuning hook (not part of the public ABI): same result from differently shaped launches
extern "C" int pm_chi2_cost8_sym_variant(const double *sc_m1, int nM, const double *sc_f1, int nF, double *out, size_t ld,
                                         size_t matrix_stride, int variant, void *stream) {
    if (!sc_m1 || !sc_f1 || !out || nM <= 0 || nF <= 0 || ld < (size_t)nF || matrix_stride < (size_t)nM * ld)
        return PM_ERR_INVALID_ARG;
    if (((uintptr_t)sc_f1 & 15) != 0 || ((uintptr_t)sc_m1 & 15) != 0) return PM_ERR_INVALID_ARG;
    hipStream_t s = (hipStream_t)stream;
    switch (variant) {
        case 0: return pm::chi2_sym_launch<4, 2>(sc_m1, nM, sc_f1, nF, out, ld, matrix_stride, s);
        case 1: return pm::chi2_sym_launch<4, 3>(sc_m1, nM, sc_f1, nF, out, ld, matrix_stride, s);
        case 2: return pm::chi2_sym_launch<2, 3>(sc_m1, nM, sc_f1, nF, out, ld, matrix_stride, s);
        case 3: return pm::chi2_sym_launch<2, 4>(sc_m1, nM, sc_f1, nF, out, ld, matrix_stride, s);
        case 4: return pm::chi2_sym_launch<1, 4>(sc_m1, nM, sc_f1, nF, out, ld, matrix_stride, s);
        case 5: return pm::chi2_sym_launch<8, 2>(sc_m1, nM, sc_f1, nF, out, ld, matrix_stride, s);
        case 6: return pm::chi2_sym_launch<4, 4>(sc_m1, nM, sc_f1, nF, out, ld, matrix_stride, s);
        default: return PM_ERR_INVALID_ARG;
    }
}

extern "C" int pm_chi2_cost8_sym(const double *sc_m1, int nM, const double *sc_f1, int nF, double *out, size_t ld,
                                 size_t matrix_stride, void *stream) {
    return pm_chi2_cost8_sym_variant(sc_m1, nM, sc_f1, nF, out, ld, matrix_stride, 0, stream);
}

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
