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
#include <string.h>

#include "mvx_common.h"

namespace {

constexpr int kN = 624, kM = 397;

// The generator's next 624 state words (the standard MT19937 recurrence, in place).
void regenerate(uint32_t* mt) {
    constexpr uint32_t kUpper = 0x80000000u, kLower = 0x7fffffffu, kMatrix = 0x9908b0dfu;
    int kk = 0;
    for (; kk < kN - kM; ++kk) {
        const uint32_t y = (mt[kk] & kUpper) | (mt[kk + 1] & kLower);
        mt[kk] = mt[kk + kM] ^ (y >> 1) ^ ((0u - (y & 1u)) & kMatrix);
    }
    for (; kk < kN - 1; ++kk) {
        const uint32_t y = (mt[kk] & kUpper) | (mt[kk + 1] & kLower);
        mt[kk] = mt[kk + (kM - kN)] ^ (y >> 1) ^ ((0u - (y & 1u)) & kMatrix);
    }
    const uint32_t y = (mt[kN - 1] & kUpper) | (mt[0] & kLower);
    mt[kN - 1] = mt[kM - 1] ^ (y >> 1) ^ ((0u - (y & 1u)) & kMatrix);
}

// Output words of a whole block at once (a loop without dependences: the compiler vectorises it).
void temper(const uint32_t* __restrict__ mt, uint32_t* __restrict__ out) {
    for (int i = 0; i < kN; ++i) {
        uint32_t y = mt[i];
        y ^= y >> 11;
        y ^= (y << 7) & 0x9d2c5680u;
        y ^= (y << 15) & 0xefc60000u;
        y ^= y >> 18;
        out[i] = y;
    }
}

struct Twister {
    uint32_t* mt;  // 624 words, caller-owned
    int pos;       // words of the current block already used (624: the next word starts a new block)
    uint32_t out[kN];

    Twister(uint32_t* state, int position) : mt(state), pos(position) { temper(mt, out); }
    void refill() {
        regenerate(mt);
        temper(mt, out);
        pos = 0;
    }
    uint32_t next() {
        if (pos >= kN) refill();
        return out[pos++];
    }
    uint32_t below(uint32_t n) {  // n >= 1
        int bits = 0;
        for (uint32_t t = n; t != 0; t >>= 1) ++bits;
        uint32_t r = next() >> (32 - bits);
        while (r >= n) r = next() >> (32 - bits);
        return r;
    }
};

int bit_length(uint32_t n) {
    int bits = 0;
    for (; n != 0; n >>= 1) ++bits;
    return bits;
}

}  // namespace

extern "C" int mvx_draw_view_pairs(uint32_t* mt_state, int32_t* mt_pos, int n_views, int64_t n_draws,
                                   uint8_t* pairs) {
    using namespace mvx;
    MVX_REQUIRE(mt_state != nullptr && mt_pos != nullptr && *mt_pos >= 0 && *mt_pos <= 624 && n_views >= 2 &&
                    n_views <= 255 && n_draws >= 0 && (n_draws == 0 || pairs != nullptr),
                MVX_ERR_INVALID_ARGUMENT, "mvx_draw_view_pairs: bad argument (n_views=%d in 2..255, n_draws=%lld)",
                n_views, (long long)n_draws);
    Twister tw(mt_state, *mt_pos);
    const uint32_t n = static_cast<uint32_t>(n_views);
    if (n <= 21) {
        // Every output word is tried against the limit of the draw in progress (n for a sample's first index, n - 1
        // for its second) and kept or dropped without a branch: both candidates are formed from the word, the phase
        // picks one, and an accepted value advances the write position and flips the phase.  (With a branch per word
        // the rejections, which come at random, cost a misprediction every other word: 5 ns per word against 1.7.)
        // An accepted value is written straight into `pairs`; the pool logic and the ordering of each pair follow in
        // a second sweep.
        const int sh0 = 32 - bit_length(n), sh1 = 32 - bit_length(n - 1);
        const uint32_t lim0 = n, lim1 = n - 1;
        const int64_t target = 2 * n_draws;
        int64_t cnt = 0;
        uint32_t second = 0;  // all ones while a sample's second index is being drawn
        while (cnt < target) {
            if (tw.pos >= kN) tw.refill();
            const uint32_t* w = tw.out;
            int i = tw.pos;
            // a whole block can be swallowed when even all-accepted it cannot overshoot the target
            if (target - cnt >= kN - i) {
                for (; i < kN; ++i) {
                    // selects written as mask arithmetic: compilers turn `second ? x : y` here into a branch
                    const uint32_t r0 = w[i] >> sh0, r1 = w[i] >> sh1;
                    const uint32_t ok0 = 0u - static_cast<uint32_t>(r0 < lim0), ok1 = 0u - static_cast<uint32_t>(r1 < lim1);
                    const uint32_t r = r0 ^ ((r0 ^ r1) & second);
                    const uint32_t ok = ok0 ^ ((ok0 ^ ok1) & second);  // all ones: accepted
                    pairs[cnt] = static_cast<uint8_t>(r);
                    cnt += ok & 1u;
                    second ^= ok;
                }
            } else {
                for (; i < kN && cnt < target; ++i) {
                    const uint32_t r = second ? (w[i] >> sh1) : (w[i] >> sh0);
                    if (r < (second ? lim1 : lim0)) {
                        pairs[cnt++] = static_cast<uint8_t>(r);
                        second = ~second;
                    }
                }
            }
            tw.pos = i;
        }
        for (int64_t d = 0; d < n_draws; ++d) {
            const uint32_t j = pairs[2 * d], k = pairs[2 * d + 1];
            // the pool held view k at position k, except that position j was refilled with the last view
            const uint32_t a = j, b = (k == j) ? n - 1 : k;
            pairs[2 * d] = static_cast<uint8_t>(a < b ? a : b);
            pairs[2 * d + 1] = static_cast<uint8_t>(a < b ? b : a);
        }
    } else {
        for (int64_t i = 0; i < n_draws; ++i) {
            const uint32_t a = tw.below(n);
            uint32_t b;
            do {
                b = tw.below(n);
            } while (b == a);
            pairs[2 * i] = static_cast<uint8_t>(a < b ? a : b);
            pairs[2 * i + 1] = static_cast<uint8_t>(a < b ? b : a);
        }
    }
    *mt_pos = tw.pos;
    return MVX_OK;
}
