// pm_host_rng.cpp — HOST-side reproduction of the random index sets do_ransac draws.
//
// Reference: `np.random.choice(n, min_samples, replace=False)` once per RANSAC trial on NumPy's global legacy
// RandomState (shape_context.py:122).  Third-party code absent from the reference tree (NumPy 2.2.6):
//   RandomState.choice(replace=False, p=None)  ->  permutation(n)[:size]         (numpy/random/mtrand.pyx)
//   permutation(n)  ->  arange(n) shuffled in place: for i = n-1 .. 1: j = random_interval(i); swap(a[i], a[j])
//   random_interval(max): smallest all-ones mask >= max; draw 32-bit MT19937 outputs & mask until <= max
//                                                                     (numpy/random/src/distributions, legacy path)
//   MT19937: Matsumoto & Nishimura's reference generator (state 624 words + position), standard tempering.
// Every trial costs a Python call and a fresh arange in NumPy (~0.5 ms at n = 5000, i.e. 4 s per 8 x 8000 trials);
// the loop below consumes exactly the same generator outputs in C and hands back the advanced state, so the global
// stream continues as if NumPy had drawn.  Parity is anchored on NumPy itself (tests/test_host_logic.py).
#include <cstdint>
#include <vector>
#if defined(__x86_64__)
#include <immintrin.h>
#endif

#include "../../include/platymatch_hip.h"

namespace {

// MT19937 with the 624 tempered outputs of a state block produced in one (vectorisable) sweep.  The consumer's vector loop
// takes 128 (or 64) outputs at a time and should not stop at block boundaries: the outputs live in a window buf[CARRY + 624];
// when fewer than CARRY are left they are moved in front of the next block, which is tempered right behind them.
constexpr int CARRY = 128;
struct MT {
    uint32_t key[624];             // the state behind the block in buf[CARRY ..]
    uint32_t prev[624];            // the state behind the block before it (whose last outputs may sit in buf[.. CARRY))
    uint32_t buf[CARRY + 624];

    // (both loops vectorise: the recurrence reaches back 227 words; clones are picked at load time by the CPU's features)
    __attribute__((target_clones("avx512f", "avx2", "default"))) void twist() {
        const uint32_t UPPER = 0x80000000u, LOWER = 0x7fffffffu, MATRIX = 0x9908b0dfu;
        int i;
        uint32_t y;
        for (i = 0; i < 624 - 397; i++) {
            y = (key[i] & UPPER) | (key[i + 1] & LOWER);
            key[i] = key[i + 397] ^ (y >> 1) ^ (-(int32_t)(y & 1) & MATRIX);
        }
        for (; i < 623; i++) {
            y = (key[i] & UPPER) | (key[i + 1] & LOWER);
            key[i] = key[i + (397 - 624)] ^ (y >> 1) ^ (-(int32_t)(y & 1) & MATRIX);
        }
        y = (key[623] & UPPER) | (key[0] & LOWER);
        key[623] = key[396] ^ (y >> 1) ^ (-(int32_t)(y & 1) & MATRIX);
    }
    __attribute__((target_clones("avx512f", "avx2", "default"))) void temper() {
        uint32_t *__restrict dst = buf + CARRY;
        for (int i = 0; i < 624; i++) {
            uint32_t y = key[i];
            y ^= (y >> 11);
            y ^= (y << 7) & 0x9d2c5680u;
            y ^= (y << 15) & 0xefc60000u;
            y ^= (y >> 18);
            dst[i] = y;
        }
    }
    // the caller's state, `pos` outputs of its block consumed -> index of the next output in buf
    int first(const uint32_t *state, int pos) {
        __builtin_memcpy(key, state, sizeof(key));
        __builtin_memcpy(prev, state, sizeof(prev));
        temper();
        return CARRY + pos;
    }
    // fewer than CARRY outputs are left (buf[p .. CARRY + 624)): move them in front of the next block -> new index
    int refill(int p) {
        const int left = CARRY + 624 - p;
        __builtin_memmove(buf + CARRY - left, buf + p, sizeof(uint32_t) * (size_t)left);
        __builtin_memcpy(prev, key, sizeof(prev));
        twist();
        temper();
        return CARRY - left;
    }
    // the state to hand back when the next output would be buf[p]
    void finish(int p, uint32_t *key_out, int *pos_out) const {
        if (p >= CARRY) {
            __builtin_memcpy(key_out, key, sizeof(key));
            *pos_out = p - CARRY;
        } else {                                     // still inside the outputs carried over from the block before
            __builtin_memcpy(key_out, prev, sizeof(prev));
            *pos_out = 624 - (CARRY - p);
        }
    }
};

constexpr int CHUNK = 64;

// The first k entries of the shuffled arange(n), given every swap partner of the Fisher-Yates pass
// `for i = n-1 .. 1: swap(a[i], a[j_i])` — WITHOUT performing the swaps.  The partners arrive in draw order:
// w[t] = j_i for i = n-1-t (t = 0 .. n-2).  Follow position q backwards through the swaps (last swap first, i.e. i
// ascending): the element that ends at q sat, before swap i = q, at j_q; from then on only a later swap whose partner IS the
// tracked position moves it (to that swap's i).  So: pos = j_q, then one scan over i = q+1 .. n-1 for partners equal to pos
// (about ln(n/q) hits), each hit setting pos = i.  The element is `pos` itself (the array starts as arange).
// follow one tracked position through swaps i_from .. i_to-1 (partners at w[n-1-i]); returns where it ends up
__attribute__((target_clones("avx512f", "avx2", "default"))) int32_t trace_range(const int32_t *__restrict w, long n, int32_t pos,
                                                                                 long i_from, long i_to) {
    long t = n - 1 - i_from;                      // walk t downwards = i upwards
    const long t_end = n - 1 - i_to;              // exclusive
    while (t > t_end) {
        long stop = t;                            // skip blocks of 64 that do not contain pos
        for (; stop - 64 >= t_end; stop -= 64) {
            int hit = 0;
            for (int u = 0; u < 64; ++u) hit |= (w[stop - u] == pos);
            if (hit) break;
        }
        const long e = stop - 64 >= t_end ? stop - 64 : t_end;    // exclusive lower end of the block to look into
        long h = stop;
        for (; h > e; --h)
            if (w[h] == pos) break;
        if (h > e) { pos = (int32_t)(n - 1 - h); t = h - 1; }
        else t = e;
    }
    return pos;
}

// The same for up to 8 positions in ONE pass with AVX-512: every vector of 16 partners is compared with all tracked positions
// at once (hits are rare: ~ln(n/q) per position and shuffle), and a vector that holds a hit is replayed lane by lane in swap
// order.  Per 64 partners: 4 loads + 4k compares against one branch, where the per-position scan above spends a reduction and
// a branch per position (measured on the box's EPYC: 0.04 ns per shuffled element and position there).
#if defined(__x86_64__)
// partners w[t_hi], w[t_hi - 1], ... (count of them) one by one, in swap order
__attribute__((target("avx512f,avx512bw,avx512vl"))) static inline void replay_avx512(const int32_t *__restrict w, long n, int k, int32_t *pos,
                                                                                   __m512i *vp, long t_hi, int count) {
    for (int c = 0; c < count; ++c) {
        const int32_t j = w[t_hi - c];
        const int32_t at = (int32_t)(n - 1 - (t_hi - c));
        for (int q = 0; q < k; ++q)
            if (pos[q] == j) { pos[q] = at; vp[q] = _mm512_set1_epi32(at); }
    }
}

__attribute__((target("avx512f,avx512bw,avx512vl"))) static void first_entries_avx512(const int32_t *__restrict w, long n, int k,
                                                                                   int32_t *__restrict out) {
    int32_t pos[8];
    for (int q = 0; q < k; ++q) pos[q] = (q >= 1 && q < n) ? w[n - 1 - q] : q;       // position after its own swap i = q
    // swaps i = 1 .. k: position q takes part from i = q + 1 on
    long i = 1;
    for (; i <= k && i < n; ++i) {
        const int32_t j = w[n - 1 - i];
        for (int q = 0; q < k && q < i; ++q)
            if (pos[q] == j) pos[q] = (int32_t)i;
    }
    // from here on every position takes part: swap i has its partner at w[n - 1 - i]; i ascending = address descending
    __m512i vp[8];
    for (int q = 0; q < k; ++q) vp[q] = _mm512_set1_epi32(pos[q]);
    long t = n - 1 - i;                                   // index of the next partner to look at (t >= 0 while i <= n - 1)
    while (t >= 63) {
        const __m512i a = _mm512_loadu_si512((const void *)(w + t - 15)), b = _mm512_loadu_si512((const void *)(w + t - 31));
        const __m512i c = _mm512_loadu_si512((const void *)(w + t - 47)), d = _mm512_loadu_si512((const void *)(w + t - 63));
        __mmask16 hit = 0;
        for (int q = 0; q < k; ++q)
            hit |= _mm512_cmpeq_epi32_mask(a, vp[q]) | _mm512_cmpeq_epi32_mask(b, vp[q]) | _mm512_cmpeq_epi32_mask(c, vp[q]) |
                   _mm512_cmpeq_epi32_mask(d, vp[q]);
        if (hit) {                                        // rare: find the group(s) of 16, in swap order, against the positions as they move
            const __m512i grp[4] = {a, b, c, d};
            for (int g = 0; g < 4; ++g) {
                __mmask16 hg = 0;
                for (int q = 0; q < k; ++q) hg |= _mm512_cmpeq_epi32_mask(grp[g], vp[q]);
                if (hg) replay_avx512(w, n, k, pos, vp, t - 16 * g, 16);
            }
        }
        t -= 64;
    }
    if (t >= 0) replay_avx512(w, n, k, pos, vp, t, (int)(t + 1));
    for (int q = 0; q < k; ++q) out[q] = pos[q];
}
#endif

void first_entries(const int32_t *__restrict w, long n, int k, int32_t *__restrict out) {
#if defined(__x86_64__)
    static const bool wide = __builtin_cpu_supports("avx512f") && __builtin_cpu_supports("avx512bw") && __builtin_cpu_supports("avx512vl");
    if (wide && k <= 8) {
        first_entries_avx512(w, n, k, out);
        return;
    }
#endif
    // the k traces advance together through chunks of the partner array that stay in L1 (one pass over memory for all k)
    constexpr long CHUNK_SWAPS = 4096;
    for (int q = 0; q < k; ++q) out[q] = (q >= 1 && q < n) ? w[n - 1 - q] : q;       // position after its own swap i = q
    for (long i0 = 1; i0 < n; i0 += CHUNK_SWAPS) {
        const long i1 = i0 + CHUNK_SWAPS < n ? i0 + CHUNK_SWAPS : n;
        for (int q = 0; q < k; ++q) {
            const long from = i0 > q ? i0 : (long)q + 1;
            if (from < i1) out[q] = trace_range(w, n, out[q], from, i1);
        }
    }
}

// One mask run of the rejection sampling, vectorised: draws v = rnd & mask are accepted iff v <= i, where i drops by one per
// acceptance.  For a block of 16 draws starting at i0 the acceptance pattern is the fixed point of a_t = [v_t <= i0 - #{s < t:
// a_s}]; the pattern that assumes every earlier draw accepted (threshold i0 - t) is a subset of it, the pattern computed from
// that subset's counts a superset: when they coincide — always, unless a draw falls in the 16-wide band below i0 — it is the
// answer and the accepted values are compress-stored.  Otherwise, and near the ends of runs and generator blocks, the scalar
// loop decides.  Returns the number of draws consumed; *pi is advanced, accepted partners appended at w[*pw ...].
#if defined(__x86_64__)
__attribute__((target("avx512f,avx512bw,avx512vl,bmi2,popcnt"))) static long consume_avx512(const uint32_t *rnd, long avail, uint32_t mask,
                                                                                         long lo, long *pi, int32_t *w, long *pw) {
    long used = 0, i = *pi, wp = *pw;
    const __m512i vmask = _mm512_set1_epi32((int)mask);
    const __m512i lane = _mm512_setr_epi32(0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15);
    for (;;) {
        // 128 draws at once: two groups of 64 whose masks do not depend on each other (the second one's band is twice as wide:
        // its draws see i somewhere in [i0 - 127, i0]); what runs from one step to the next through i is then one chain of
        // broadcast -> compare -> mask -> count per 128 draws instead of per 64.
        if (avail - used >= 128 && i - 128 >= lo) {
            const __m512i hi = _mm512_set1_epi32((int)i), belowA = _mm512_set1_epi32((int)(i - 63)), belowB = _mm512_set1_epi32((int)(i - 127));
            __m512i v[8];
#pragma GCC unroll 8
            for (int g = 0; g < 8; ++g) v[g] = _mm512_and_si512(_mm512_loadu_si512((const void *)(rnd + used + 16 * g)), vmask);
            uint64_t candA = 0, accA = 0, candB = 0, accB = 0;
#pragma GCC unroll 4
            for (int g = 0; g < 4; ++g) {
                candA |= (uint64_t)_mm512_cmple_epu32_mask(v[g], hi) << (16 * g);
                accA |= (uint64_t)_mm512_cmple_epu32_mask(v[g], belowA) << (16 * g);
                candB |= (uint64_t)_mm512_cmple_epu32_mask(v[4 + g], hi) << (16 * g);
                accB |= (uint64_t)_mm512_cmple_epu32_mask(v[4 + g], belowB) << (16 * g);
            }
            uint64_t band = candA & ~accA;
            while (band) {                                  // in draw order: the earlier ones are decided by now
                const int t = __builtin_ctzll(band);
                band &= band - 1;
                if ((long)(rnd[used + t] & mask) <= i - (long)__builtin_popcountll(accA & ((1ull << t) - 1))) accA |= 1ull << t;
            }
            const long totalA = __builtin_popcountll(accA);
            band = candB & ~accB;
            while (band) {
                const int t = __builtin_ctzll(band);
                band &= band - 1;
                if ((long)(rnd[used + 64 + t] & mask) <= i - totalA - (long)__builtin_popcountll(accB & ((1ull << t) - 1))) accB |= 1ull << t;
            }
            long at = wp;
#pragma GCC unroll 4
            for (int g = 0; g < 4; ++g) {
                const __mmask16 m = (__mmask16)(accA >> (16 * g));
                _mm512_storeu_si512((void *)(w + at), _mm512_maskz_compress_epi32(m, v[g]));
                at += __builtin_popcount((unsigned)m);
            }
#pragma GCC unroll 4
            for (int g = 0; g < 4; ++g) {
                const __mmask16 m = (__mmask16)(accB >> (16 * g));
                _mm512_storeu_si512((void *)(w + at), _mm512_maskz_compress_epi32(m, v[4 + g]));
                at += __builtin_popcount((unsigned)m);
            }
            i -= at - wp;
            wp = at;
            used += 128;
            continue;
        }
        // 64 draws at once, free of the count-to-threshold dependency: over these draws i stays within [i0 - 63, i0], so a value
        // <= i0 - 63 is accepted and one > i0 rejected whatever the counts are.  Only a value inside that band needs the number
        // of acceptances before it; those few (a draw falls in the band with probability 63 / 2^b) are settled one by one, in
        // draw order, from the masks.
        if (avail - used >= 64 && i - 64 >= lo) {
            const __m512i hi = _mm512_set1_epi32((int)i), below = _mm512_set1_epi32((int)(i - 63));
            const __m512i v0 = _mm512_and_si512(_mm512_loadu_si512((const void *)(rnd + used)), vmask);
            const __m512i v1 = _mm512_and_si512(_mm512_loadu_si512((const void *)(rnd + used + 16)), vmask);
            const __m512i v2 = _mm512_and_si512(_mm512_loadu_si512((const void *)(rnd + used + 32)), vmask);
            const __m512i v3 = _mm512_and_si512(_mm512_loadu_si512((const void *)(rnd + used + 48)), vmask);
            const uint64_t cand = (uint64_t)_mm512_cmple_epu32_mask(v0, hi) | ((uint64_t)_mm512_cmple_epu32_mask(v1, hi) << 16) |
                                  ((uint64_t)_mm512_cmple_epu32_mask(v2, hi) << 32) | ((uint64_t)_mm512_cmple_epu32_mask(v3, hi) << 48);
            uint64_t acc = (uint64_t)_mm512_cmple_epu32_mask(v0, below) | ((uint64_t)_mm512_cmple_epu32_mask(v1, below) << 16) |
                           ((uint64_t)_mm512_cmple_epu32_mask(v2, below) << 32) | ((uint64_t)_mm512_cmple_epu32_mask(v3, below) << 48);
            uint64_t band = cand & ~acc;
            while (band) {                                  // in draw order: the earlier ones are decided by now
                const int t = __builtin_ctzll(band);
                band &= band - 1;
                const long before = __builtin_popcountll(acc & ((1ull << t) - 1));
                if ((long)(rnd[used + t] & mask) <= i - before) acc |= 1ull << t;
            }
            // (compress in registers + a full-width store: what lies beyond the accepted values is overwritten by the next store;
            // w has 64 entries of slack)
            const __mmask16 m0 = (__mmask16)acc, m1 = (__mmask16)(acc >> 16), m2 = (__mmask16)(acc >> 32), m3 = (__mmask16)(acc >> 48);
            const int c0 = __builtin_popcount((unsigned)m0), c1 = __builtin_popcount((unsigned)m1), c2 = __builtin_popcount((unsigned)m2);
            _mm512_storeu_si512((void *)(w + wp), _mm512_maskz_compress_epi32(m0, v0));
            _mm512_storeu_si512((void *)(w + wp + c0), _mm512_maskz_compress_epi32(m1, v1));
            _mm512_storeu_si512((void *)(w + wp + c0 + c1), _mm512_maskz_compress_epi32(m2, v2));
            _mm512_storeu_si512((void *)(w + wp + c0 + c1 + c2), _mm512_maskz_compress_epi32(m3, v3));
            const int total = (int)__builtin_popcountll(acc);
            wp += total;
            i -= total;
            used += 64;
            continue;
        }
        if (i - 128 >= lo) break;                            // short of outputs, not near the end of a mask run: the caller tops up
        // the last < 64 of a mask run, 16 draws at a time in the same way (band: the 15 values below i0)
        if (!(avail - used >= 16 && i - 16 >= lo)) break;
        const __m512i v = _mm512_and_si512(_mm512_loadu_si512((const void *)(rnd + used)), vmask);
        const unsigned cand = _mm512_cmple_epu32_mask(v, _mm512_set1_epi32((int)i));
        unsigned acc = _mm512_cmple_epu32_mask(v, _mm512_set1_epi32((int)(i - 15)));
        unsigned band = cand & ~acc;
        while (band) {
            const int t = __builtin_ctz(band);
            band &= band - 1;
            const long before = __builtin_popcount(acc & ((1u << t) - 1));
            if ((long)(rnd[used + t] & mask) <= i - before) acc |= 1u << t;
        }
        _mm512_storeu_si512((void *)(w + wp), _mm512_maskz_compress_epi32((__mmask16)acc, v));
        const int total = __builtin_popcount(acc);
        wp += total;
        i -= total;
        used += 16;
    }
    *pi = i;
    *pw = wp;
    return used;
}
static bool have_avx512() {
    static const bool ok = __builtin_cpu_supports("avx512f") && __builtin_cpu_supports("avx512bw") && __builtin_cpu_supports("avx512vl");
    return ok;
}
#else
static long consume_avx512(const uint32_t *, long, uint32_t, long, long *, int32_t *, long *) { return 0; }
static bool have_avx512() { return false; }
#endif

}  // namespace

// One trial's rejection sampling: consume generator outputs (block by block through `next_block`) until the n - 1 swap partners
// of a shuffle are decided; w receives them in draw order.  rnd / p: the current block and the position in it (updated).
// One trial's rejection sampling: consume generator outputs until the n - 1 swap partners of a shuffle are decided; w
// receives them in draw order.  p: index of the next output in mt.buf (updated).
static inline void draw_partners(long n, int32_t *__restrict w, MT &mt, int &p, bool simd) {
    // Fisher-Yates from the top, in runs of i that share one rejection mask (2^b - 1 for i in [2^(b-1), 2^b)): a draw
    // is accepted if it is <= the current i, which then drops by one.  Accepted values are appended to w without a
    // data-dependent branch (a rejected draw is overwritten by the next one); no array is permuted.
    constexpr int END = CARRY + 624;
    const uint32_t *__restrict rnd = mt.buf;
    long i = n - 1, wp = 0;
    while (i >= 1) {
        const uint32_t mask = 0xffffffffu >> __builtin_clz((uint32_t)i);   // smallest all-ones mask >= i
        const long lo = (long)(mask >> 1) + 1;                             // last i that uses this mask
        while (i >= lo) {
            if (END - p < CARRY) p = mt.refill(p);
            if (simd) p += (int)consume_avx512(rnd + p, END - p, mask, lo, &i, w, &wp);
            // scalar: up to 16 draws (the end of a mask run)
            for (int q = 0; q < 16 && p < END && i >= lo; ++q) {
                const uint32_t v = rnd[p++] & mask;
                w[wp] = (int32_t)v;
                const long take = (v <= (uint32_t)i);
                wp += take;
                i -= take;
            }
        }
    }
}

static int legacy_choice_serial(uint32_t *key, int *pos, long n, int k, long trials, int32_t *out) {
    MT mt;
    int p = mt.first(key, *pos);                        // outputs pos..623 of the block the caller's state is in
    std::vector<int32_t> partner((size_t)n + CHUNK);
    int32_t *__restrict w = partner.data();
    const bool simd = have_avx512();
    for (long t = 0; t < trials; ++t) {
        draw_partners(n, w, mt, p, simd);
        first_entries(w, n, k, out + t * k);
    }
    mt.finish(p, key, pos);
    return PM_OK;
}

// (A three-stage pipeline on three threads — generator | rejection sampling | trace, single-producer/single-consumer rings —
// was measured and dropped: 1.4 ns per shuffled element against 0.52-0.93 ns on one thread of the MI355X box's EPYC 9575F, and
// likewise slower on a Xeon: handing 4 bytes per generator output from core to core costs more than producing them.)

// key[624], *pos: the MT19937 part of np.random.get_state() (updated in place).  out: trials x k int32, the first k
// entries of each of `trials` successive permutation(n) calls.  n <= 2^31.
extern "C" int pm_legacy_choice(uint32_t *key, int *pos, long n, int k, long trials, int32_t *out) {
    if (!key || !pos || !out || n <= 0 || k <= 0 || k > n || trials < 0 || n > 0x7fffffffL || *pos < 0 || *pos > 624)
        return PM_ERR_INVALID_ARG;
    try {
        return legacy_choice_serial(key, pos, n, k, trials, out);
    } catch (...) {
        return PM_ERR_WORKSPACE;
    }
}
