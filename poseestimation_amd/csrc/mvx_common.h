// Shared host/device helpers of the mvx kernels (gfx950, wave64).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/mvx.h"

namespace mvx {

// ---- host side: error reporting -------------------------------------------------------------
void set_error(const char* fmt, ...);
unsigned kernel_selection();  // mask set by mvx_set_kernel_selection (MVX_SELECT_* bits)

#define MVX_REQUIRE(cond, code, ...)      \
    do {                                  \
        if (!(cond)) {                    \
            mvx::set_error(__VA_ARGS__);  \
            return (code);                \
        }                                 \
    } while (0)

#define MVX_CHECK_LAUNCH(name)                                                            \
    do {                                                                                  \
        hipError_t e_ = hipGetLastError();                                                \
        if (e_ != hipSuccess) {                                                           \
            mvx::set_error("%s: launch failed: %s", (name), hipGetErrorString(e_));       \
            return MVX_ERR_HIP;                                                           \
        }                                                                                 \
    } while (0)

// Dynamic LDS to request (and not use) so that at most about `bytes_in_flight` of maps of `hw` floats are resident
// per CU: HBM sustains more when the chip-wide window of open pages is small.  Never fewer than two workgroups per
// CU (one cannot hide its own latency: 4.3 against 5.7 TB/s on 112 x 112 maps); 0 = no cap (8 or more fit).
inline size_t residency_cap_lds(int hw, int64_t bytes_in_flight) {
    int per_cu = static_cast<int>((bytes_in_flight + static_cast<int64_t>(hw) * 2) / (static_cast<int64_t>(hw) * 4));
    if (per_cu < 2) per_cu = 2;
    return per_cu >= 8 ? 0 : (static_cast<size_t>(160) * 1024 / per_cu - 1024) / 1024 * 1024;
}

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

// ---- device side: wave64 reductions through DPP ----------------------------------------------
// quad_perm [1,0,3,2] = 0xB1, quad_perm [2,3,0,1] = 0x4E, row_half_mirror = 0x141, row_mirror = 0x140.
// After the four steps every lane holds the reduction of its 16-lane row; the four rows are then
// combined through readlane (wave-uniform result).  Every lane of these patterns has a valid source; bound_ctrl is
// set only so that the compiler need not materialise the "old" operand (a v_mov of zero per DPP move).
template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
template <int CTRL>
__device__ __forceinline__ double dpp_mov(double v) {
    unsigned long long b = __builtin_bit_cast(unsigned long long, v);
    int lo = static_cast<int>(b), hi = static_cast<int>(b >> 32);
    lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xF, 0xF, true);
    hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xF, 0xF, true);
    b = (static_cast<unsigned long long>(static_cast<unsigned>(hi)) << 32) | static_cast<unsigned>(lo);
    return __builtin_bit_cast(double, b);
}
__device__ __forceinline__ float read_lane(float v, int lane) {
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), lane));
}
__device__ __forceinline__ double read_lane(double v, int lane) {
    unsigned long long b = __builtin_bit_cast(unsigned long long, v);
    unsigned lo = static_cast<unsigned>(__builtin_amdgcn_readlane(static_cast<int>(b), lane));
    unsigned hi = static_cast<unsigned>(__builtin_amdgcn_readlane(static_cast<int>(b >> 32), lane));
    return __builtin_bit_cast(double, (static_cast<unsigned long long>(hi) << 32) | lo);
}

// Division of small non-negative numbers by a runtime constant without the float-reciprocal sequence the compiler
// emits for it: q = (mulhi(n, magic) + n) >> shift, exact for n < 2^31 (magic and shift from fast_divisor()).
struct FastDiv {
    unsigned magic, shift, divisor;
};
inline FastDiv fast_divisor(unsigned d) {
    FastDiv f;
    f.divisor = d;
    f.shift = 0;
    while ((1u << f.shift) < d) ++f.shift;
    f.magic = static_cast<unsigned>((((1ull << f.shift) - d) << 32) / d + 1);
    return f;
}
__device__ __forceinline__ unsigned fast_div(unsigned n, FastDiv f) { return (__umulhi(n, f.magic) + n) >> f.shift; }

__device__ __forceinline__ float wave_max(float v) {
    v = fmaxf(v, dpp_mov<0xB1>(v));
    v = fmaxf(v, dpp_mov<0x4E>(v));
    v = fmaxf(v, dpp_mov<0x141>(v));
    v = fmaxf(v, dpp_mov<0x140>(v));
    return fmaxf(fmaxf(read_lane(v, 0), read_lane(v, 16)), fmaxf(read_lane(v, 32), read_lane(v, 48)));
}
template <int CTRL>
__device__ __forceinline__ int dpp_movi(int v) {
    return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xF, 0xF, true);
}
__device__ __forceinline__ int wave_min_i(int v) {
    v = min(v, dpp_movi<0xB1>(v));
    v = min(v, dpp_movi<0x4E>(v));
    v = min(v, dpp_movi<0x141>(v));
    v = min(v, dpp_movi<0x140>(v));
    return min(min(__builtin_amdgcn_readlane(v, 0), __builtin_amdgcn_readlane(v, 16)),
               min(__builtin_amdgcn_readlane(v, 32), __builtin_amdgcn_readlane(v, 48)));
}
__device__ __forceinline__ int wave_max_i(int v) {
    v = max(v, dpp_movi<0xB1>(v));
    v = max(v, dpp_movi<0x4E>(v));
    v = max(v, dpp_movi<0x141>(v));
    v = max(v, dpp_movi<0x140>(v));
    return max(max(__builtin_amdgcn_readlane(v, 0), __builtin_amdgcn_readlane(v, 16)),
               max(__builtin_amdgcn_readlane(v, 32), __builtin_amdgcn_readlane(v, 48)));
}
__device__ __forceinline__ double wave_sum(double v) {
    v += dpp_mov<0xB1>(v);
    v += dpp_mov<0x4E>(v);
    v += dpp_mov<0x141>(v);
    v += dpp_mov<0x140>(v);
    return (read_lane(v, 0) + read_lane(v, 16)) + (read_lane(v, 32) + read_lane(v, 48));
}

// exp(d) for d <= 0 through v_exp_f32 (2^x): relative error about 1e-7, which the soft-argmax
// tolerances leave ample room for (the sums it feeds are accumulated in float64).
__device__ __forceinline__ float exp_neg(float d) { return __builtin_amdgcn_exp2f(d * 1.44269504088896340736f); }

// max(t, 0) that keeps a NaN a NaN, as torch's relu does (fmaxf would return 0 for it).
__device__ __forceinline__ float relu_keep_nan(float t) { return t > 0.0f ? t : (t != t ? t : 0.0f); }

// Blocks b and b+8 share an XCD (and its L2).  Gives each XCD a contiguous range of work items so that
// neighbouring items, which read the same inputs, meet in one L2.  Bijective for every grid size.
__device__ __forceinline__ unsigned xcd_contiguous(unsigned id, unsigned n) {
    const unsigned xcd = id & 7u, k = id >> 3, q = n >> 3, r = n & 7u;
    const unsigned base = xcd < r ? xcd * (q + 1u) : r * (q + 1u) + (xcd - r) * q;
    return base + k;
}

}  // namespace mvx
