// Host-side: the view pairs RANSAC triangulation tests, drawn exactly as CPython's random.sample(views, 2)
// draws them from the Mersenne Twister state it is handed, so that a caller who seeds Python's `random` gets the
// reference's hypotheses without one interpreter call per draw (0.7 million of them for 4096 frames).
//
// CPython (3.9 - 3.12, Lib/random.py and Modules/_randommodule.c), restated:
//   getrandbits(k), k <= 32     = genrand_uint32() >> (32 - k)            (MT19937 with the standard tempering)
//   _randbelow(n)               : k = n.bit_length(); r = getrandbits(k); while r >= n: r = getrandbits(k)
//   sample(population, 2), n = len(population):
//     n <= 21 : pool = list(population); j = _randbelow(n); first = pool[j]; pool[j] = pool[n-1];
//               j = _randbelow(n-1); second = pool[j]
//     n  > 21 : first = _randbelow(n); second = _randbelow(n) repeated until it differs from first
// No GPU work happens here.
#include <stdint.h>

#include "mvx_common.h"

namespace {

struct Twister {
    uint32_t* mt;  // 624 words, caller-owned
    int pos;

    uint32_t next() {
        constexpr int N = 624, M = 397;
        constexpr uint32_t kUpper = 0x80000000u, kLower = 0x7fffffffu, kMatrix = 0x9908b0dfu;
        if (pos >= N) {
            int kk = 0;
            for (; kk < N - M; ++kk) {
                const uint32_t y = (mt[kk] & kUpper) | (mt[kk + 1] & kLower);
                mt[kk] = mt[kk + M] ^ (y >> 1) ^ ((y & 1u) ? kMatrix : 0u);
            }
            for (; kk < N - 1; ++kk) {
                const uint32_t y = (mt[kk] & kUpper) | (mt[kk + 1] & kLower);
                mt[kk] = mt[kk + (M - N)] ^ (y >> 1) ^ ((y & 1u) ? kMatrix : 0u);
            }
            const uint32_t y = (mt[N - 1] & kUpper) | (mt[0] & kLower);
            mt[N - 1] = mt[M - 1] ^ (y >> 1) ^ ((y & 1u) ? kMatrix : 0u);
            pos = 0;
        }
        uint32_t y = mt[pos++];
        y ^= y >> 11;
        y ^= (y << 7) & 0x9d2c5680u;
        y ^= (y << 15) & 0xefc60000u;
        y ^= y >> 18;
        return y;
    }

    uint32_t below(uint32_t n) {  // n >= 1
        int bits = 0;
        for (uint32_t t = n; t != 0; t >>= 1) ++bits;
        uint32_t r = next() >> (32 - bits);
        while (r >= n) r = next() >> (32 - bits);
        return r;
    }
};

}  // namespace

extern "C" int mvx_draw_view_pairs(uint32_t* mt_state, int32_t* mt_pos, int n_views, int64_t n_draws,
                                   uint8_t* pairs) {
    using namespace mvx;
    MVX_REQUIRE(mt_state != nullptr && mt_pos != nullptr && *mt_pos >= 0 && *mt_pos <= 624 && n_views >= 2 &&
                    n_views <= 255 && n_draws >= 0 && (n_draws == 0 || pairs != nullptr),
                MVX_ERR_INVALID_ARGUMENT, "mvx_draw_view_pairs: bad argument (n_views=%d in 2..255, n_draws=%lld)",
                n_views, (long long)n_draws);
    Twister tw{mt_state, *mt_pos};
    const uint32_t n = static_cast<uint32_t>(n_views);
    for (int64_t i = 0; i < n_draws; ++i) {
        uint32_t a, b;
        if (n <= 21) {
            const uint32_t j = tw.below(n);
            a = j;
            const uint32_t k = tw.below(n - 1);
            // the pool held view k at position k, except that position j was refilled with the last view
            b = (k == j) ? n - 1 : k;
        } else {
            a = tw.below(n);
            do {
                b = tw.below(n);
            } while (b == a);
        }
        pairs[2 * i] = static_cast<uint8_t>(a < b ? a : b);
        pairs[2 * i + 1] = static_cast<uint8_t>(a < b ? b : a);
    }
    *mt_pos = tw.pos;
    return MVX_OK;
}
