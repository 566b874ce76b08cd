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

struct MT {
    uint32_t *key;
    int pos;
    void refill() {
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
        pos = 0;
    }
    inline uint32_t next() {
        if (pos == 624) refill();
        uint32_t y = key[pos++];
        y ^= (y >> 11);
        y ^= (y << 7) & 0x9d2c5680u;
        y ^= (y << 15) & 0xefc60000u;
        y ^= (y >> 18);
        return y;
    }
};

}  // namespace

// key[624], *pos: the MT19937 part of np.random.get_state() (updated in place).  out: trials x k int32, the first k
// entries of each of `trials` successive permutation(n) calls.  n <= 2^31.
extern "C" int pm_legacy_choice(uint32_t *key, int *pos, long n, int k, long trials, int32_t *out) {
    if (!key || !pos || !out || n <= 0 || k <= 0 || k > n || trials < 0 || n > 0x7fffffffL || *pos < 0 || *pos > 624)
        return PM_ERR_INVALID_ARG;
    MT mt{key, *pos};
    std::vector<int32_t> a((size_t)n);
    uint32_t mask_for_top = (uint32_t)(n - 1);
    mask_for_top |= mask_for_top >> 1; mask_for_top |= mask_for_top >> 2; mask_for_top |= mask_for_top >> 4;
    mask_for_top |= mask_for_top >> 8; mask_for_top |= mask_for_top >> 16;
    for (long t = 0; t < trials; ++t) {
        for (long i = 0; i < n; ++i) a[(size_t)i] = (int32_t)i;
        uint32_t mask = mask_for_top;
        for (long i = n - 1; i >= 1; --i) {
            const uint32_t max = (uint32_t)i;
            while ((mask >> 1) >= max) mask >>= 1;          // smallest all-ones mask >= i (i only decreases)
            uint32_t j;
            while ((j = (mt.next() & mask)) > max) {}
            const int32_t tmp = a[(size_t)i];
            a[(size_t)i] = a[j];
            a[j] = tmp;
        }
        for (int q = 0; q < k; ++q) out[t * k + q] = a[(size_t)q];
    }
    *pos = mt.pos;
    return PM_OK;
}
