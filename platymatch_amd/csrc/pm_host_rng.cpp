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

#include "../../include/platymatch_hip.h"

namespace {

// MT19937 with the 624 tempered outputs of a state block produced in one (vectorisable) sweep.  `pos` counts the
// outputs of the current block already consumed, exactly NumPy's `pos`.
struct MT {
    uint32_t *key;
    int pos;
    uint32_t out[624];

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
        for (int i = 0; i < 624; i++) {
            uint32_t y = key[i];
            y ^= (y >> 11);
            y ^= (y << 7) & 0x9d2c5680u;
            y ^= (y << 15) & 0xefc60000u;
            y ^= (y >> 18);
            out[i] = y;
        }
    }
    void refill() {       // all outputs of the current block are consumed: advance the state by one block
        twist();
        temper();
        pos = 0;
    }
};

constexpr int CHUNK = 64;

// The first k entries of the shuffled arange(n), given every swap partner j[i] (i = 1 .. n-1; j[i] <= i) of the
// Fisher-Yates pass `for i = n-1 .. 1: swap(a[i], a[j[i]])` — WITHOUT performing the swaps.  Follow position q backwards
// through the swaps (last swap first, i.e. i ascending): the element that ends at q sat, before swap i = q, at j[q]; from
// then on only a later swap whose partner IS the tracked position moves it (to that swap's i).  So: pos = j[q], then one
// ascending scan of j[q+1 ..] for entries equal to pos (about ln(n/q) hits), each hit setting pos = i.  The element is `pos`
// itself (the array starts as arange).  k scans run together; the compare loop vectorises.
__attribute__((target_clones("avx512f", "avx2", "default"))) void first_entries(const int32_t *__restrict j, long n, int k,
                                                                                 int32_t *__restrict out) {
    for (int q = 0; q < k; ++q) {
        int32_t pos = (q >= 1) ? j[q] : 0;
        long i = q + 1;
        while (i < n) {
            // skip ahead in blocks while no entry equals pos
            long stop = i;
            for (; stop + 64 <= n; stop += 64) {
                int hit = 0;
                for (int u = 0; u < 64; ++u) hit |= (j[stop + u] == pos);
                if (hit) break;
            }
            long e = stop + 64 < n ? stop + 64 : n;
            long h = stop;
            for (; h < e; ++h)
                if (j[h] == pos) break;
            if (h < e) { pos = (int32_t)h; i = h + 1; }
            else i = e;
        }
        out[q] = pos;
    }
}

}  // namespace

// key[624], *pos: the MT19937 part of np.random.get_state() (updated in place).  out: trials x k int32, the first k
// entries of each of `trials` successive permutation(n) calls.  n <= 2^31.
extern "C" int pm_legacy_choice(uint32_t *key, int *pos, long n, int k, long trials, int32_t *out) {
    if (!key || !pos || !out || n <= 0 || k <= 0 || k > n || trials < 0 || n > 0x7fffffffL || *pos < 0 || *pos > 624)
        return PM_ERR_INVALID_ARG;
    MT mt;
    mt.key = key;
    mt.pos = *pos;
    mt.temper();                                        // outputs pos..623 of the block the caller's state is in
    std::vector<int32_t> partner((size_t)n + CHUNK);
    int32_t *__restrict jv = partner.data();
    const uint32_t *__restrict rnd = mt.out;           // locals the compiler can keep apart from the stores into jv[]
    int p = mt.pos;
    for (long t = 0; t < trials; ++t) {
        // Fisher-Yates from the top, in runs of i that share one rejection mask (2^b - 1 for i in [2^(b-1), 2^b)): a draw
        // is accepted if it is <= the current i, which then drops by one.  Accepted values are stored at jv[i] without a
        // data-dependent branch (a rejected draw is overwritten by the next one); no array is permuted.
        jv[0] = 0;
        long i = n - 1;
        while (i >= 1) {
            const uint32_t mask = 0xffffffffu >> __builtin_clz((uint32_t)i);   // smallest all-ones mask >= i
            const long lo = (long)(mask >> 1) + 1;                             // last i that uses this mask
            while (i >= lo) {
                if (p == 624) {
                    mt.refill();
                    p = 0;
                }
                // as many draws as this generator block and this mask run allow, at most
                long room = 624 - p;
                for (long q = 0; q < room && i >= lo; ++q) {
                    const uint32_t v = rnd[p++] & mask;
                    jv[i] = (int32_t)v;
                    i -= (v <= (uint32_t)i);
                }
            }
        }
        first_entries(jv, n, k, out + t * k);
    }
    mt.pos = p;
    *pos = mt.pos;
    return PM_OK;
}
