// Soft-argmax over 2-D heatmaps (one workgroup of 256 threads per heatmap).
//
// HBM traffic per heatmap: one read of h*w floats, 8 bytes of coordinates, and (optionally) one write of
// h*w floats.  The register kernel keeps the whole map in VGPRs between the max, the exp-sum and the
// normalised write-back, so each element crosses HBM exactly once in each direction (any map of up to 16384
// elements whose width is a multiple of 4); the LDS kernel does the same through a shared-memory tile for
// the other shapes; maps that do not fit LDS are re-read from global memory (L2) per pass.
//
// Arithmetic: t = float32(x * multiplier); m = max t (exact); e = exp(t - m) in float32;
// S = sum e, Sx = sum e*col, Sy = sum e*row in float64; x = Sx/S, y = Sy/S rounded once to float32.
#include <stdlib.h>

#include "mvx_common.h"

namespace mvx {

namespace {

constexpr int kThreads = 256;  // block size of the any-shape kernel

template <int WAVES>
struct BlockReduce {
    float m[WAVES];
    double s[WAVES][3];
};

// Block-wide max of one float per thread; every thread gets the result.
template <int WAVES>
__device__ __forceinline__ float block_max(float v, BlockReduce<WAVES>& red) {
    v = wave_max(v);
    if ((threadIdx.x & 63) == 0) red.m[threadIdx.x >> 6] = v;
    __syncthreads();
    float m = red.m[0];
#pragma unroll
    for (int i = 1; i < WAVES; ++i) m = fmaxf(m, red.m[i]);
    return m;
}

// Block-wide sums of three doubles per thread; every thread gets the results.
template <int WAVES>
__device__ __forceinline__ void block_sum3(double& a, double& b, double& c, BlockReduce<WAVES>& red) {
    a = wave_sum(a);
    b = wave_sum(b);
    c = wave_sum(c);
    if ((threadIdx.x & 63) == 0) {
        red.s[threadIdx.x >> 6][0] = a;
        red.s[threadIdx.x >> 6][1] = b;
        red.s[threadIdx.x >> 6][2] = c;
    }
    __syncthreads();
    a = red.s[0][0];
    b = red.s[0][1];
    c = red.s[0][2];
#pragma unroll
    for (int i = 1; i < WAVES; ++i) {
        a += red.s[i][0];
        b += red.s[i][1];
        c += red.s[i][2];
    }
}

// Adds the four elements of one float4 (same row, columns col..col+3) to the float64 sums.
__device__ __forceinline__ void accumulate4(const float4& e, int row, int col, double& s, double& sx, double& sy) {
    const double e0 = e.x, e1 = e.y, e2 = e.z, e3 = e.w;
    const double s4 = (e0 + e1) + (e2 + e3);
    s += s4;
    sy = fma(s4, static_cast<double>(row), sy);
    sx = fma(s4, static_cast<double>(col), sx);
    sx += fma(3.0, e3, fma(2.0, e2, e1));
}

typedef float vec4f __attribute__((ext_vector_type(4)));
// Streaming (non-temporal) 16-byte accesses: every byte is touched once, so nothing is worth keeping in cache.
__device__ __forceinline__ float4 load_f4(const float4* p, bool nt) {
    if (nt) {
        const vec4f r = __builtin_nontemporal_load(reinterpret_cast<const vec4f*>(p));
        return make_float4(r.x, r.y, r.z, r.w);
    }
    return *p;
}
__device__ __forceinline__ void store_f4(float4* p, const float4& v, bool nt) {
    if (nt) {
        vec4f r = {v.x, v.y, v.z, v.w};
        __builtin_nontemporal_store(r, reinterpret_cast<vec4f*>(p));
    } else {
        *p = v;
    }
}

// w % 4 == 0 and h*w <= THREADS*4*NV: the map lives in NV float4 registers per thread.  EXACT: h*w is exactly
// THREADS*4*NV (no guards); otherwise float4 slots past the end of the map hold -inf, which neither the maximum
// nor the sums see (exp gives 0, max(x, 0) gives 0), and are not written back.
template <int THREADS, int NV, bool EXACT>
__global__ __launch_bounds__(THREADS) void soft_argmax_2d_reg(const float* __restrict__ hm, int hw, int w, float mult,
                                                               int softmax, float* __restrict__ coords,
                                                               float* __restrict__ out) {
    constexpr int kThreads = THREADS;
    __shared__ BlockReduce<THREADS / 64> red;
    const int tid = threadIdx.x;
    const size_t map = blockIdx.x;
    const int n4 = hw >> 2;
    const float4* src = reinterpret_cast<const float4*>(hm + map * static_cast<size_t>(hw));

    float4 v[NV];
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        v[k] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        if (EXACT || k * kThreads + tid < n4) v[k] = load_f4(src + k * kThreads + tid, true);
    }
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        v[k].x *= mult;
        v[k].y *= mult;
        v[k].z *= mult;
        v[k].w *= mult;
        if (!EXACT && k * kThreads + tid >= n4) v[k] = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
    }

    float m = 0.0f;
    if (softmax) {
        m = fmaxf(fmaxf(v[0].x, v[0].y), fmaxf(v[0].z, v[0].w));
#pragma unroll
        for (int k = 1; k < NV; ++k) m = fmaxf(m, fmaxf(fmaxf(v[k].x, v[k].y), fmaxf(v[k].z, v[k].w)));
        m = block_max(m, red);
    }

    // position of this thread's first float4 and the (uniform) step between its float4s
    int row = (tid * 4) / w, col = (tid * 4) % w;
    const int row_step = (kThreads * 4) / w, col_step = (kThreads * 4) % w;
    double s = 0.0, sx = 0.0, sy = 0.0;
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        if (softmax) {
            v[k].x = exp_neg(v[k].x - m);
            v[k].y = exp_neg(v[k].y - m);
            v[k].z = exp_neg(v[k].z - m);
            v[k].w = exp_neg(v[k].w - m);
        } else {
            v[k].x = relu_keep_nan(v[k].x);
            v[k].y = relu_keep_nan(v[k].y);
            v[k].z = relu_keep_nan(v[k].z);
            v[k].w = relu_keep_nan(v[k].w);
        }
        if (EXACT || k * kThreads + tid < n4) accumulate4(v[k], row, col, s, sx, sy);
        row += row_step;
        col += col_step;
        if (col >= w) {
            col -= w;
            row += 1;
        }
    }
    block_sum3(s, sx, sy, red);

    if (tid == 0) {
        coords[map * 2 + 0] = static_cast<float>(sx / s);
        coords[map * 2 + 1] = static_cast<float>(sy / s);
    }
    if (out != nullptr) {
        const float inv = softmax ? static_cast<float>(1.0 / s) : 1.0f;
        float4* dst = reinterpret_cast<float4*>(out + map * static_cast<size_t>(hw));
#pragma unroll
        for (int k = 0; k < NV; ++k) {
            float4 p = v[k];
            p.x *= inv;
            p.y *= inv;
            p.z *= inv;
            p.w *= inv;
            if (EXACT || k * kThreads + tid < n4) store_f4(dst + k * kThreads + tid, p, true);
        }
    }
}

// Any h, w.  STAGED: the scaled map is staged once in dynamic LDS and the later passes read the tile;
// otherwise every pass re-reads global memory.
template <bool STAGED>
__global__ __launch_bounds__(kThreads) void soft_argmax_2d_any(const float* __restrict__ hm, int h, int w, float mult,
                                                                int softmax, float* __restrict__ coords,
                                                                float* __restrict__ out) {
    extern __shared__ __align__(16) float tile[];
    __shared__ BlockReduce<kThreads / 64> red;
    const int tid = threadIdx.x;
    const size_t map = blockIdx.x;
    const int hw = h * w;
    const float* src = hm + map * static_cast<size_t>(hw);

    float m = -INFINITY;
    for (int i = tid; i < hw; i += kThreads) {
        const float t = src[i] * mult;
        if (STAGED) tile[i] = t;
        m = fmaxf(m, t);
    }
    if (softmax) {
        m = block_max(m, red);  // also orders the tile writes before the reads below
    } else {
        m = 0.0f;
        __syncthreads();
    }

    double s = 0.0, sx = 0.0, sy = 0.0;
    for (int i = tid; i < hw; i += kThreads) {
        const float t = STAGED ? tile[i] : src[i] * mult;
        const float e = softmax ? exp_neg(t - m) : relu_keep_nan(t);
        if (STAGED) tile[i] = e;
        const int row = i / w, col = i - row * w;
        const double ed = e;
        s += ed;
        sx = fma(ed, static_cast<double>(col), sx);
        sy = fma(ed, static_cast<double>(row), sy);
    }
    block_sum3(s, sx, sy, red);

    if (tid == 0) {
        coords[map * 2 + 0] = static_cast<float>(sx / s);
        coords[map * 2 + 1] = static_cast<float>(sy / s);
    }
    if (out != nullptr) {
        const float inv = softmax ? static_cast<float>(1.0 / s) : 1.0f;
        float* dst = out + map * static_cast<size_t>(hw);
        for (int i = tid; i < hw; i += kThreads) {
            float e;
            if (STAGED) {
                e = tile[i];
            } else {
                const float t = src[i] * mult;
                e = softmax ? exp_neg(t - m) : relu_keep_nan(t);
            }
            dst[i] = e * inv;
        }
    }
}

}  // namespace

}  // namespace mvx

extern "C" int mvx_soft_argmax_2d(const float* heatmaps, int64_t n_maps, int h, int w, float multiplier, int softmax,
                                  float* coords, float* heatmaps_out, void* stream) {
    using namespace mvx;
    MVX_REQUIRE(n_maps >= 0 && h > 0 && w > 0, MVX_ERR_INVALID_ARGUMENT,
                "mvx_soft_argmax_2d: bad shape n_maps=%lld h=%d w=%d", (long long)n_maps, h, w);
    if (n_maps == 0) return MVX_OK;
    MVX_REQUIRE(heatmaps != nullptr && coords != nullptr, MVX_ERR_INVALID_ARGUMENT,
                "mvx_soft_argmax_2d: null heatmaps or coords pointer");
    MVX_REQUIRE(n_maps <= 0x7fffffffLL && static_cast<int64_t>(h) * w <= (1 << 24), MVX_ERR_UNSUPPORTED,
                "mvx_soft_argmax_2d: n_maps=%lld or h*w=%lld beyond the launch limits", (long long)n_maps,
                (long long)h * w);
    hipStream_t st = static_cast<hipStream_t>(stream);
    const dim3 grid(static_cast<unsigned>(n_maps)), block(kThreads);
    const int hw = h * w;
    const bool vec_ok = (w % 4 == 0) && aligned16(heatmaps) && (heatmaps_out == nullptr || aligned16(heatmaps_out));
    // Every map is read once and written once.  With the write-back, HBM sustains more when few maps are in
    // flight per CU (about 72 KB, i.e. 2 resident workgroups for 96x96 maps, measured best on MI355X: the reads
    // and writes of the whole chip then stay inside a window of a few MB); LDS the kernel does not use is
    // requested only to cap the residency.  Without the write-back the stream is read-only and twice that
    // (4 workgroups) is best.
    const size_t cap_lds = residency_cap_lds(hw, (heatmaps_out != nullptr ? 72 : 144) * 1024);
    // (256, 384, 576 and 768 threads per map were measured within 1 % of each other at the same residency.)
#define MVX_LAUNCH_REG(NV, EXACT)                                                                                   \
    hipLaunchKernelGGL((soft_argmax_2d_reg<256, NV, EXACT>), grid, block, cap_lds, st, heatmaps, hw, w, multiplier,   \
                       softmax, coords, heatmaps_out)
    const int n4 = hw / 4;
    if (vec_ok && n4 <= 256 * 16) {
        // the map in registers: exact sizes (64 x 64, 96 x 96, 128 x 128) without guards, others in the smallest
        // guarded instantiation that holds them
        if (n4 == 256 * 9) MVX_LAUNCH_REG(9, true);
        else if (n4 == 256 * 4) MVX_LAUNCH_REG(4, true);
        else if (n4 == 256 * 16) MVX_LAUNCH_REG(16, true);
        else if (n4 <= 256 * 1) MVX_LAUNCH_REG(1, false);
        else if (n4 <= 256 * 2) MVX_LAUNCH_REG(2, false);
        else if (n4 <= 256 * 3) MVX_LAUNCH_REG(3, false);
        else if (n4 <= 256 * 4) MVX_LAUNCH_REG(4, false);
        else if (n4 <= 256 * 6) MVX_LAUNCH_REG(6, false);
        else if (n4 <= 256 * 9) MVX_LAUNCH_REG(9, false);
        else if (n4 <= 256 * 12) MVX_LAUNCH_REG(12, false);
        else MVX_LAUNCH_REG(16, false);
#undef MVX_LAUNCH_REG
    } else if (static_cast<size_t>(hw) * sizeof(float) <= 64 * 1024) {
        hipLaunchKernelGGL(soft_argmax_2d_any<true>, grid, block, static_cast<size_t>(hw) * sizeof(float), st,
                           heatmaps, h, w, multiplier, softmax, coords, heatmaps_out);
    } else {
        hipLaunchKernelGGL(soft_argmax_2d_any<false>, grid, block, 0, st, heatmaps, h, w, multiplier, softmax,
                           coords, heatmaps_out);
    }
    MVX_CHECK_LAUNCH("mvx_soft_argmax_2d");
    return MVX_OK;
}
