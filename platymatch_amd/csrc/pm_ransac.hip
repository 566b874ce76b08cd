// pm_ransac.hip — RANSAC trial scoring.
// Reference: do_ransac (shape_context.py:103-139), called on the matched pair lists
// moving[:, row_indices], fixed[:, col_indices] (_dock_widget.py:622-675).
//
// The reference runs `trials` sequential iterations of {draw 4 pairs, fit, apply to all, count
// inliers}.  Trials are independent given their index sets, so the host draws every set first (same
// RNG calls, same order) and one launch evaluates them all: lane <-> trial (its 4x4 stays in
// registers), the matched pairs stream through LDS as wave-uniform broadcasts, the four waves of a
// workgroup split each staged chunk and their counts are added at the end.
// The fit of a trial's min_samples pairs (find_transform.py:4-17): four pairs -> the interpolating affine from the
// sample's edge matrices; more -> least squares from moments centred on the sample means (pm_solve.h).  Samples that
// are (nearly) rank deficient — coplanar or repeated points, where the reference's pinv returns a minimum-norm
// answer — are flagged, score zero here, and are refitted by the host mirror with the reference's own expression.
#include "pm_common.h"
#include "pm_solve.h"

namespace pm {

// ---- index sets drawn on the device (unseeded runs) -----------------------------------------------------------------
// The reference draws a trial's pairs with np.random.choice(n, k, replace=False) from NumPy's GLOBAL generator
// (shape_context.py:122) and never seeds it: what it asks for is "k distinct pairs, every k-subset equally
// likely", not a particular stream.  NumPy's call permutes all n indices per trial — 8 x 8 000 full shuffles per
// registration, the step that bounded a registration below 50 000 nuclei when restated on the host (pm_host_rng.cpp, kept
// for SEEDED runs, where the reference's exact sets are reproduced).  Here each trial owns a counter-based stream
// (Philox-4x32-10, Salmon et al. SC'11: key = the caller's 64-bit seed, counter = (trial, block, run, 0)) and picks its k
// indices by Floyd's algorithm: for j = n-k .. n-1: t = uniform{0..j}; take t unless already taken, else j — every k-subset
// with probability 1 / C(n, k), k draws, no array.  Uniform integers by Lemire's multiply-and-reject (exactly uniform).
struct Philox {
    uint32_t k0, k1, c0, c1, c2, c3;
    uint32_t w[4];
    int have;
    __device__ __forceinline__ Philox(uint64_t seed, uint32_t trial, uint32_t run) : k0((uint32_t)seed), k1((uint32_t)(seed >> 32)),
                                                                                   c0(trial), c1(0u), c2(run), c3(0u), have(0) {}
    __device__ __forceinline__ void block() {
        uint32_t x0 = c0, x1 = c1, x2 = c2, x3 = c3, a = k0, b = k1;
#pragma unroll
        for (int r = 0; r < 10; ++r) {
            const uint64_t p0 = (uint64_t)0xD2511F53u * x0, p1 = (uint64_t)0xCD9E8D57u * x2;
            const uint32_t y0 = (uint32_t)(p1 >> 32) ^ x1 ^ a, y1 = (uint32_t)p1, y2 = (uint32_t)(p0 >> 32) ^ x3 ^ b, y3 = (uint32_t)p0;
            x0 = y0; x1 = y1; x2 = y2; x3 = y3;
            a += 0x9E3779B9u; b += 0xBB67AE85u;
        }
        w[0] = x0; w[1] = x1; w[2] = x2; w[3] = x3;
        ++c1;                                              // next block of this trial's stream
        have = 4;
    }
    __device__ __forceinline__ uint32_t next() {
        if (have == 0) block();
        const uint32_t v = have == 4 ? w[0] : have == 3 ? w[1] : have == 2 ? w[2] : w[3];    // (no dynamic register indexing)
        --have;
        return v;
    }
    // uniform integer in [0, range), range >= 1 (Lemire 2019: unbiased, one multiply, a division only on the rare retry path)
    __device__ __forceinline__ uint32_t below(uint32_t range) {
        uint64_t m = (uint64_t)next() * range;
        uint32_t lo = (uint32_t)m;
        if (lo < range) {
            const uint32_t thresh = (0u - range) % range;
            while (lo < thresh) { m = (uint64_t)next() * range; lo = (uint32_t)m; }
        }
        return (uint32_t)(m >> 32);
    }
};

// k distinct indices of [0, n) for one trial, written to out[0..k) (global memory of this thread's own row).
__device__ __forceinline__ void draw_subset(uint64_t seed, uint32_t trial, uint32_t run, int n, int k, int32_t *out) {
    Philox g(seed, trial, run);
    for (int q = 0; q < k; ++q) {
        const int j = n - k + q;
        int t = (int)g.below((uint32_t)j + 1u);
        for (int p = 0; p < q; ++p)
            if (out[p] == t) { t = j; break; }             // j itself cannot have been taken: earlier picks are < j
        out[q] = t;
    }
}

__global__ __launch_bounds__(256) void draw_kernel(int n, int k, int trials, uint64_t seed, uint32_t run, int32_t *__restrict__ samples) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t < trials) draw_subset(seed, (uint32_t)t, run, n, k, samples + (size_t)t * k);
}

constexpr int RS_THREADS = 256;
constexpr int RS_WAVES = 4;
constexpr int RS_CHUNK = 512;
constexpr int RS_SUB = RS_CHUNK / RS_WAVES;

template <bool FIT, bool DRAW = false>
__global__ __launch_bounds__(RS_THREADS) void ransac_kernel(const double *__restrict__ mov, int n_mov,
                                                            const double *__restrict__ fix, int n_fix,
                                                            const int32_t *__restrict__ rows, const int32_t *__restrict__ cols,
                                                            int n, int32_t *samples, uint64_t seed, uint32_t run, int k,
                                                            const double *__restrict__ A_in, int trials, double error,
                                                            double *__restrict__ A_out, int32_t *__restrict__ inliers,
                                                            int32_t *__restrict__ degenerate) {
    __shared__ double Pm[3][RS_CHUNK];
    __shared__ double Pf[3][RS_CHUNK];
    __shared__ int cnt_s[RS_WAVES][64];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int t = blockIdx.x * 64 + lane;
    const int tc = min(t, trials - 1);

    double A[16];
    bool flagged = false;
    if (FIT) {
        if (DRAW) {
            // this trial's index set is drawn here, in front of its fit: lane `lane` of wave 0 draws it into the caller's samples
            // array (kept: degenerate samples are refitted on the host, and the winner's set is reported); the same lane of
            // the other three waves — which fit the same trial and score another quarter of each chunk — reads it from there
            // behind the workgroup barrier (release/acquire at workgroup scope: one CU, one L1)
            if (wave == 0 && t < trials) draw_subset(seed, (uint32_t)t, run, n, k, samples + (size_t)t * k);
            __syncthreads();
        }
        auto pair_of = [&](int q, double a[3], double f[3]) {
            const int s = samples[(size_t)tc * k + q];
            const int im = rows ? rows[s] : s, jf = cols ? cols[s] : s;
#pragma unroll
            for (int c = 0; c < 3; ++c) { a[c] = mov[(size_t)c * n_mov + im]; f[c] = fix[(size_t)c * n_fix + jf]; }
        };
        if (k == 4) {
            // four pairs: the interpolating affine from the edge matrices (conditioned like the sample itself)
            double m4[4][3], f4[4][3];
#pragma unroll
            for (int q = 0; q < 4; ++q) pair_of(q, m4[q], f4[q]);
            const double had = affine_from_4(m4, f4, A);
            flagged = !(had > PM_DEGENERATE_SIMPLEX);                      // also true for NaN
        } else {
            // k > 4 pairs: least squares from moments centred on the sample means (two passes over the k pairs)
            double mb[3] = {0.0, 0.0, 0.0}, fb[3] = {0.0, 0.0, 0.0};
            for (int q = 0; q < k; ++q) {
                double a[3], f[3];
                pair_of(q, a, f);
#pragma unroll
                for (int c = 0; c < 3; ++c) { mb[c] += a[c]; fb[c] += f[c]; }
            }
            const double rk = 1.0 / (double)k;
#pragma unroll
            for (int c = 0; c < 3; ++c) { mb[c] *= rk; fb[c] *= rk; }
            double cmm[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0}, cfm[9];
#pragma unroll
            for (int c = 0; c < 9; ++c) cfm[c] = 0.0;
            for (int q = 0; q < k; ++q) {
                double a[3], f[3];
                pair_of(q, a, f);
#pragma unroll
                for (int c = 0; c < 3; ++c) { a[c] -= mb[c]; f[c] -= fb[c]; }
                cmm[0] += a[0] * a[0]; cmm[1] += a[0] * a[1]; cmm[2] += a[0] * a[2];
                cmm[3] += a[1] * a[1]; cmm[4] += a[1] * a[2]; cmm[5] += a[2] * a[2];
#pragma unroll
                for (int r = 0; r < 3; ++r)
#pragma unroll
                    for (int c = 0; c < 3; ++c) cfm[3 * r + c] += f[r] * a[c];
            }
            const double ratio = affine_from_centred(cmm, cfm, mb, fb, A);
            flagged = !(ratio > PM_DEGENERATE_MOMENTS_SAMPLE);
        }
        if (flagged) {                  // never an inlier below; the host refits such trials with the reference's pinv
#pragma unroll
            for (int q = 0; q < 16; ++q) A[q] = NAN;
        }
    } else {
#pragma unroll
        for (int q = 0; q < 16; ++q) A[q] = A_in[(size_t)tc * 16 + q];
    }

    int cnt = 0;
    for (int c0 = 0; c0 < n; c0 += RS_CHUNK) {
        __syncthreads();
        for (int e = tid; e < RS_CHUNK; e += RS_THREADS) {
            const int k = c0 + e;
            if (k < n) {
                const int im = rows ? rows[k] : k, jf = cols ? cols[k] : k;
#pragma unroll
                for (int c = 0; c < 3; ++c) { Pm[c][e] = mov[(size_t)c * n_mov + im]; Pf[c][e] = fix[(size_t)c * n_fix + jf]; }
            } else {
#pragma unroll
                for (int c = 0; c < 3; ++c) { Pm[c][e] = 0.0; Pf[c][e] = NAN; }   // NaN distance: never an inlier
            }
        }
        __syncthreads();
        const int e0 = wave * RS_SUB;
#pragma unroll 4
        for (int e = e0; e < e0 + RS_SUB; ++e) {
            const double x = Pm[0][e], y = Pm[1][e], z = Pm[2][e];
            const double d0 = Pf[0][e] - affine_row(A, x, y, z);
            const double d1 = Pf[1][e] - affine_row(A + 4, x, y, z);
            const double d2 = Pf[2][e] - affine_row(A + 8, x, y, z);
            const double d = __builtin_sqrt(__builtin_fma(d2, d2, __builtin_fma(d1, d1, d0 * d0)));   // np.linalg.norm of a 3-vector (:133) = sqrt(x.dot(x)): BLAS ddot's multiply-add chain
            cnt += (d <= error) ? 1 : 0;                                        // :134
        }
    }
    cnt_s[wave][lane] = cnt;
    __syncthreads();
    if (wave == 0 && t < trials) {
        inliers[t] = cnt_s[0][lane] + cnt_s[1][lane] + cnt_s[2][lane] + cnt_s[3][lane];
        if (FIT && degenerate) degenerate[t] = flagged ? 1 : 0;
        if (FIT && A_out) {
#pragma unroll
            for (int k = 0; k < 16; ++k) A_out[(size_t)t * 16 + k] = A[k];
        }
    }
}

}  // namespace pm

extern "C" {

int pm_ransac_affine(const double *mov, int n_mov, const double *fix, int n_fix, const int32_t *rows, const int32_t *cols,
                     int n, const int32_t *samples, int min_samples, int trials, double error, double *A_out, int32_t *inliers,
                     int32_t *degenerate, void *stream) {
    if (!mov || !fix || !samples || !inliers || n_mov <= 0 || n_fix <= 0 || n <= 0 || trials <= 0) return PM_ERR_INVALID_ARG;
    if (min_samples < 4) return PM_ERR_UNSUPPORTED;      // fewer than four pairs: rank deficient by construction (host pinv)
    if (min_samples > n) return PM_ERR_INVALID_ARG;
    pm::ransac_kernel<true><<<(trials + 63) / 64, pm::RS_THREADS, 0, (hipStream_t)stream>>>(
        mov, n_mov, fix, n_fix, rows, cols, n, const_cast<int32_t *>(samples), 0ull, 0u, min_samples, nullptr, trials, error, A_out,
        inliers, degenerate);
    return pm::launch_status();
}

int pm_ransac_draw(int n, int min_samples, int trials, uint64_t seed, uint32_t run, int32_t *samples, void *stream) {
    if (!samples || n <= 0 || trials <= 0 || min_samples <= 0) return PM_ERR_INVALID_ARG;
    if (min_samples > n) return PM_ERR_INVALID_ARG;
    pm::draw_kernel<<<(trials + 255) / 256, 256, 0, (hipStream_t)stream>>>(n, min_samples, trials, seed, run, samples);
    return pm::launch_status();
}

int pm_ransac_affine_draw(const double *mov, int n_mov, const double *fix, int n_fix, const int32_t *rows, const int32_t *cols,
                          int n, int min_samples, int trials, uint64_t seed, uint32_t run, double error, int32_t *samples_out,
                          double *A_out, int32_t *inliers, int32_t *degenerate, void *stream) {
    if (!mov || !fix || !samples_out || !inliers || n_mov <= 0 || n_fix <= 0 || n <= 0 || trials <= 0) return PM_ERR_INVALID_ARG;
    if (min_samples < 4) return PM_ERR_UNSUPPORTED;
    if (min_samples > n) return PM_ERR_INVALID_ARG;
    pm::ransac_kernel<true, true><<<(trials + 63) / 64, pm::RS_THREADS, 0, (hipStream_t)stream>>>(
        mov, n_mov, fix, n_fix, rows, cols, n, samples_out, seed, run, min_samples, nullptr, trials, error, A_out, inliers, degenerate);
    return pm::launch_status();
}

int pm_ransac_score(const double *mov, int n_mov, const double *fix, int n_fix, const int32_t *rows, const int32_t *cols,
                    int n, const double *A_in, int trials, double error, int32_t *inliers, void *stream) {
    if (!mov || !fix || !A_in || !inliers || n_mov <= 0 || n_fix <= 0 || n <= 0 || trials <= 0) return PM_ERR_INVALID_ARG;
    pm::ransac_kernel<false><<<(trials + 63) / 64, pm::RS_THREADS, 0, (hipStream_t)stream>>>(
        mov, n_mov, fix, n_fix, rows, cols, n, nullptr, 0ull, 0u, 0, A_in, trials, error, nullptr, inliers, nullptr);
    return pm::launch_status();
}

}  // extern "C"
