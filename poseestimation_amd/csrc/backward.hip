// Backward kernels of the two volumetric operations: what torch autograd computes when the reference trains through
// unproject_heatmaps (mvn/utils/op.py:99-166: grid_sample per view, zeroing of voxels behind the camera, aggregation
// over views) and integrate_tensor_3d / integrate_tensor_3d_with_coordinates (mvn/utils/op.py:50-96: softmax or ReLU,
// then the expectation of the coordinates); the reference's training step is train.py:281-290.
//
// Unprojection.  A lane projects its voxel with the forward kernels' own routine (unproject_taps.h: the same taps,
// the same weights) and adds weight x gradient x (derivative of the view aggregation) to the four taps of every
// channel of a channel-last gradient [frame][view][h][w][Cp].
//   unproject_backward_maps  one 1024-thread workgroup per (frame, view, 2 channels) keeps that slice of the gradient
//                            -- the whole h x w map in float64, 145 KB for 96 x 96 -- in LDS, walks all voxels of the
//                            frame adding into it with LDS atomics, and writes the slice once, rounded to float32, with
//                            plain stores.  No global atomics, no clearing pass.  Larger maps are cut into bands of
//                            rows, one workgroup each.  float64 because that is the fast LDS atomic on gfx950: ds_add_f64
//                            retires 4 to 7 lanes per clock per CU, ds_add_f32 one lane per 3 clocks whatever the
//                            addresses (tools/lds_atomic_rate.hip) -- and the sums come out exact to float32 rounding.
//                            Rows of the slice are padded by one element: the voxels of a wave lie along one grid
//                            axis and project onto a line of pixels, and with rows of 2 W doubles a vertical line
//                            would fall into one pair of LDS banks.
//   unproject_view_factors   max and softmax only, before the kernel above: the derivative of the aggregate with
//                            respect to each view's sample (for max: which view holds the maximum), written once per
//                            (view, channel, voxel) into the workspace
//                            (the size of the per-view volumes the forward never materialises: V C N floats per frame,
//                            134 MB for the model's shape), so that the slices sample nothing.  Each of the 16 slices
//                            of a view sampling for itself uses 8 of the 128 bytes of every pixel it touches, and the
//                            texture path handles one line per lane whatever it returns: softmax for 8 frames 8.2 ms
//                            with every slice sampling all views, 3.2 ms with per-voxel statistics and its own view.
//   unproject_confidence_sums  confidence aggregation with the confidences' gradient wanted, before the slice kernel:
//                            sum over voxels of gradient x sample per (view, channel), in runs of 2048 voxels whose
//                            totals the slice kernel adds up in order.
//   unproject_backward       whatever the kernels above do not take (rows of more than 9 700 pixels, more than 64
//                            channels or 16 views with the confidences' gradient): one lane per voxel, float32 atomic
//                            adds at agent scope into a gradient the entry point has zeroed (for 8 frames of the
//                            model's shape 52 ms: every addition is a round trip to the memory side, and their order
//                            is not fixed: two runs differ in the last bits).
//
// 3-D soft-argmax.  With p the forward's output volume, X_i the coordinates of voxel i, gc the gradient of the
// volume's coordinates and gp_i the gradient of p_i, G_i = gp_i + gc . X_i and
//   softmax:                 dL/dx_i = m p_i (G_i - sum_j p_j G_j / sum_j p_j)
//   ReLU (with coordinates): dL/dx_i = m G_i where p_i > 0
//   ReLU (index space, where the reference divides the coordinates by the mass M = sum p; c = the coordinates):
//                            dL/dx_i = m (gp_i + gc . (X_i - c) / M) where p_i > 0
// (m = the multiplier folded into the forward call).  G and the sums are float64.  A volume is cut into segments so
// that a training batch of a few frames still fills the chip: one kernel leaves each segment's partial sums in the
// workspace, the next adds them up (in a fixed order) and writes its segment.
#include "grid_coords.h"
#include "unproject_taps.h"

namespace mvx {

namespace {

constexpr int kBackwardThreads = 256;

__device__ __forceinline__ void load_camera(const float* __restrict__ proj, int64_t bv, double sx, double sy,
                                            double (&Pd)[12]) {
#pragma unroll
    for (int i = 0; i < 12; ++i) Pd[i] = camera_entry(proj + bv * 12, i, sx, sy);
}

// Channels 4q .. 4q+3 of one view at a voxel's taps: the forward kernels' interpolation, operation for operation.
__device__ __forceinline__ void sample_quad(const float* __restrict__ img, const Tap& t, int W, int Cp, int q,
                                            float (&val)[4]) {
    const float* p = img + static_cast<size_t>(t.pixel) * Cp + 4 * q;
    const float4 f00 = *reinterpret_cast<const float4*>(p);
    const float4 f01 = *reinterpret_cast<const float4*>(p + Cp);
    const float4 f10 = *reinterpret_cast<const float4*>(p + static_cast<size_t>(W) * Cp);
    const float4 f11 = *reinterpret_cast<const float4*>(p + static_cast<size_t>(W + 1) * Cp);
    val[0] = fmaf(f11.x, t.w11, fmaf(f10.x, t.w10, fmaf(f01.x, t.w01, f00.x * t.w00)));
    val[1] = fmaf(f11.y, t.w11, fmaf(f10.y, t.w10, fmaf(f01.y, t.w01, f00.y * t.w00)));
    val[2] = fmaf(f11.z, t.w11, fmaf(f10.z, t.w10, fmaf(f01.z, t.w01, f00.z * t.w00)));
    val[3] = fmaf(f11.w, t.w11, fmaf(f10.w, t.w10, fmaf(f01.w, t.w01, f00.w * t.w00)));
}

__device__ __forceinline__ void add_quad(float* p, float w, const float (&d)[4]) {
    if (w == 0.0f) return;  // a tap outside the map, or a voxel behind the camera: the reference passes no gradient
#pragma unroll
    for (int k = 0; k < 4; ++k) unsafeAtomicAdd(p + k, d[k] * w);
}
// The transpose of sample_quad: d[k] times each tap's weight, added to the taps.
__device__ __forceinline__ void scatter_quad(float* __restrict__ gimg, const Tap& t, int W, int Cp, int q,
                                             const float (&d)[4]) {
    float* p = gimg + static_cast<size_t>(t.pixel) * Cp + 4 * q;
    add_quad(p, t.w00, d);
    add_quad(p + Cp, t.w01, d);
    add_quad(p + static_cast<size_t>(W) * Cp, t.w10, d);
    add_quad(p + static_cast<size_t>(W + 1) * Cp, t.w11, d);
}

// feat_cl / g_feat_cl: [frames][V][H][W][Cp]; g_vol: [frames][C][n_vox]; g_conf: [frames][V][C] or null.
template <int AGG>
__global__ __launch_bounds__(kBackwardThreads) void unproject_backward(
    const float* __restrict__ feat_cl, const float* __restrict__ proj, VoxelSource coord,
    const float* __restrict__ vconf, const float* __restrict__ g_vol, int64_t n_total, int V, int C, int Cp, int H, int W,
    int64_t n_vox, double sx, double sy, float* __restrict__ g_feat_cl, float* __restrict__ g_conf) {
    const int64_t idx = static_cast<int64_t>(blockIdx.x) * kBackwardThreads + threadIdx.x;
    const bool active = idx < n_total;
    const int64_t at = active ? idx : n_total - 1;  // idle lanes of the last wave shadow the last voxel and add nothing
    const int64_t b = at / n_vox, n = at - b * n_vox;
    double X, Y, Z;
    voxel_position(coord, b, n_vox, n, X, Y, Z);
    const size_t map = static_cast<size_t>(H) * W * Cp;
    constexpr bool kNeedsValues = AGG == MVX_AGG_MAX || AGG == MVX_AGG_SOFTMAX;
    // the confidence gradient of a wave goes out as one addition per (view, channel) when all its lanes are in one frame
    const int frame = static_cast<int>(b);
    const bool one_frame = __all(frame == __builtin_amdgcn_readfirstlane(frame));

    for (int q = 0; q < Cp / 4; ++q) {
        float g[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int c = 4 * q + k;
            g[k] = (active && c < C) ? g_vol[(b * C + c) * n_vox + n] : 0.0f;
        }
        // what the aggregation needs of all views before any view's gradient is known
        float top[4], den[4], out[4];
        if (kNeedsValues) {
            float num[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                top[k] = -INFINITY;
                den[k] = 0.0f;
                num[k] = 0.0f;
            }
            for (int v = 0; v < V; ++v) {
                double Pd[12];
                load_camera(proj, b * V + v, sx, sy, Pd);
                const Tap t = project_tap(Pd, X, Y, Z, H, W, 1.0f);
                float val[4];
                sample_quad(feat_cl + (b * V + v) * map, t, W, Cp, q, val);
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    if (AGG == MVX_AGG_MAX) {
                        top[k] = fmaxf(top[k], val[k]);
                    } else {  // running maximum; what is summed so far is rescaled when it moves
                        const float d = val[k] - top[k];
                        const float r = exp_neg(-fabsf(d));
                        const bool up = d > 0.0f;
                        const float keep = up ? r : 1.0f, e = up ? 1.0f : r;
                        den[k] = fmaf(den[k], keep, e);
                        num[k] = fmaf(num[k], keep, e * val[k]);
                        top[k] = up ? val[k] : top[k];
                    }
                }
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) out[k] = num[k] / den[k];
        }
        bool taken[4] = {false, false, false, false};  // max: the gradient goes to the first view that holds the maximum
        for (int v = 0; v < V; ++v) {
            double Pd[12];
            load_camera(proj, b * V + v, sx, sy, Pd);
            Tap t = project_tap(Pd, X, Y, Z, H, W, 1.0f);
            float val[4] = {0.0f, 0.0f, 0.0f, 0.0f}, d[4];
            if (kNeedsValues || (AGG == MVX_AGG_CONF && g_conf != nullptr))
                sample_quad(feat_cl + (b * V + v) * map, t, W, Cp, q, val);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int c = 4 * q + k;
                if (AGG == MVX_AGG_SUM) {
                    d[k] = g[k];
                } else if (AGG == MVX_AGG_CONF) {
                    d[k] = (c < C) ? g[k] * vconf[(b * V + v) * C + c] : 0.0f;
                } else if (AGG == MVX_AGG_MAX) {
                    const bool hit = (val[k] == top[k]) && !taken[k];
                    taken[k] = taken[k] || hit;
                    d[k] = hit ? g[k] : 0.0f;
                } else {  // out = sum_v s_v x_v with s = softmax(x):  d out / d x_v = s_v (1 + x_v - out)
                    const float s = exp_neg(val[k] - top[k]) / den[k];
                    d[k] = g[k] * s * (1.0f + val[k] - out[k]);
                }
            }
            if (active) scatter_quad(g_feat_cl + (b * V + v) * map, t, W, Cp, q, d);
            if (AGG == MVX_AGG_CONF && g_conf != nullptr) {
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int c = 4 * q + k;
                    if (c >= C) continue;  // uniform over the wave
                    const float r = g[k] * val[k];  // g is zero on idle lanes
                    if (one_frame) {
                        const double total = wave_sum(static_cast<double>(r));
                        if ((threadIdx.x & 63) == 0) unsafeAtomicAdd(g_conf + (b * V + v) * C + c, static_cast<float>(total));
                    } else if (r != 0.0f) {
                        unsafeAtomicAdd(g_conf + (b * V + v) * C + c, r);
                    }
                }
            }
        }
    }
}

constexpr int kMapsThreads = 1024;
constexpr int kMapsMaxViews = 16;
constexpr size_t kMapsMaxSliceBytes = 152 * 1024;

__host__ __device__ inline int slice_row(int W) { return 2 * W + 1; }  // doubles per padded row of a slice

// f(u) for every view: unrolled for a number of views known at compile time (NV > 0), a plain loop otherwise.
template <int NV, typename F>
__device__ __forceinline__ void for_views(int V, F&& f) {
    if constexpr (NV > 0) {
#pragma unroll
        for (int u = 0; u < NV; ++u) f(u);
    } else {
        for (int u = 0; u < V; ++u) f(u);
    }
}

// For max and softmax, before the slice kernel.  softmax: the derivative of the aggregate with respect to every view's
// sample, factors[frame][view][channel pair][voxel] (float2): s_v (1 + x_v - out) with s the softmax over the views'
// samples x and out = sum_v s_v x_v.  max: factors[frame][channel pair][voxel] (float2), the index of the first view
// that holds the maximum (the only one whose sample the aggregate depends on).
// A workgroup takes 256 / (Cp / 4) consecutive voxels.  First its threads project them, one (view, voxel) each, and
// leave the taps in LDS; then one lane per (voxel, channel quad), consecutive lanes the quads of one voxel, so that
// 8 lanes read one whole 128-byte pixel (for 32 channels) and the texture path serves a tap of 8 voxels per
// instruction rather than of one.  NV = 2, 3 or 4 views (what the reference's collate function draws) keep the samples
// in registers between the two passes; NV = 0 is any number of views, sampled a second time (out of L1).
// Dynamic LDS: V * (256 / (Cp / 4)) taps.
template <int AGG, int NV>
__global__ __launch_bounds__(kBackwardThreads) void unproject_view_factors(
    const float* __restrict__ feat_cl, const float* __restrict__ proj, VoxelSource coord, int64_t n_total, int V, int Cp,
    int H, int W, int64_t n_vox, double sx, double sy, float* __restrict__ factors) {
    extern __shared__ Tap taps[];  // [V][voxels of the workgroup]
    const int Cq = Cp / 4, pairs = Cp / 2;
    const int vpw = kBackwardThreads / Cq;
    const int64_t first = static_cast<int64_t>(blockIdx.x) * vpw;
    for (int i = threadIdx.x; i < vpw * V; i += kBackwardThreads) {
        const int u = i / vpw, vl = i - u * vpw;
        const int64_t at = first + vl;
        if (at >= n_total) continue;
        const int64_t b = at / n_vox, n = at - b * n_vox;
        double X, Y, Z, Pd[12];
        voxel_position(coord, b, n_vox, n, X, Y, Z);
        load_camera(proj, b * V + u, sx, sy, Pd);
        taps[u * vpw + vl] = project_tap(Pd, X, Y, Z, H, W, 1.0f);
    }
    __syncthreads();
    const int vl = threadIdx.x / Cq, q = threadIdx.x - vl * Cq;
    const int64_t at = first + vl;
    if (vl >= vpw || at >= n_total) return;
    const int64_t b = at / n_vox, n = at - b * n_vox;
    const size_t map = static_cast<size_t>(H) * W * Cp;
    float top[4], den[4], num[4];
    int best[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        top[k] = -INFINITY;
        den[k] = 0.0f;
        num[k] = 0.0f;
        best[k] = 0;
    }
    constexpr bool kKeep = NV > 0 && AGG == MVX_AGG_SOFTMAX;
    float kept[kKeep ? NV : 1][4];
    for_views<NV>(V, [&](int u) {
        float val[4];
        sample_quad(feat_cl + (b * V + u) * map, taps[u * vpw + vl], W, Cp, q, val);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (kKeep) kept[kKeep ? u : 0][k] = val[k];
            if (AGG == MVX_AGG_MAX) {
                const bool up = (u == 0) || (val[k] > top[k]);  // strictly greater: the first maximal view stays
                best[k] = up ? u : best[k];
                top[k] = up ? val[k] : top[k];
            } else {  // running maximum; what is summed so far is rescaled when it moves
                const float dd = val[k] - top[k];
                const float r = exp_neg(-fabsf(dd));
                const bool up = dd > 0.0f;
                const float keep = up ? r : 1.0f, e = up ? 1.0f : r;
                den[k] = fmaf(den[k], keep, e);
                num[k] = fmaf(num[k], keep, e * val[k]);
                top[k] = up ? val[k] : top[k];
            }
        }
    });
    float rden[4], out[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        rden[k] = 1.0f / den[k];
        out[k] = num[k] * rden[k];
    }
    if (AGG == MVX_AGG_MAX) {  // one record for all views: which of them holds the maximum
        float2* row = reinterpret_cast<float2*>(factors) + (static_cast<size_t>(b) * pairs + 2 * q) * n_vox + n;
        row[0] = make_float2(static_cast<float>(best[0]), static_cast<float>(best[1]));
        row[n_vox] = make_float2(static_cast<float>(best[2]), static_cast<float>(best[3]));
        return;
    }
    for_views<NV>(V, [&](int u) {
        float f[4];
        {
            float val[4];
            if (kKeep) {
#pragma unroll
                for (int k = 0; k < 4; ++k) val[k] = kept[kKeep ? u : 0][k];
            } else {
                sample_quad(feat_cl + (b * V + u) * map, taps[u * vpw + vl], W, Cp, q, val);
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) f[k] = exp_neg(val[k] - top[k]) * rden[k] * (1.0f + val[k] - out[k]);
        }
        float2* row = reinterpret_cast<float2*>(factors) + ((static_cast<size_t>(b) * V + u) * pairs + 2 * q) * n_vox + n;
        row[0] = make_float2(f[0], f[1]);
        row[n_vox] = make_float2(f[2], f[3]);
    });
}

// Confidence aggregation, before the slice kernel, when the gradient of the confidences is wanted:
// sum over voxels of (gradient x this view's sample) per (view, channel), which is that gradient.  One workgroup per
// (frame, run of kConfChunk voxels); lanes as in unproject_view_factors (8 lanes read one whole pixel).  A lane sums
// its voxels of the run in registers, the workgroup's lanes meet in LDS, and the run's totals go to
// partial[frame][run][view][Cp] (float64) for the slice kernel to add up in run order.
// NV = 2, 3 or 4 views: rounds of 256 / (Cp / 4) voxels are projected once per (view, voxel) into LDS (dynamic LDS:
// NV x that many taps) and all views are summed in one sweep; NV = 0 (any number of views): one sweep per view, every
// lane projecting for itself.
constexpr int kConfChunk = 2048;
constexpr int kConfMaxChannels = 64;  // kMapsMaxViews x kConfMaxChannels doubles of LDS
template <int NV>
__global__ __launch_bounds__(kBackwardThreads) void unproject_confidence_sums(
    const float* __restrict__ feat_cl, const float* __restrict__ proj, VoxelSource coord, const float* __restrict__ g_vol,
    int runs, int V, int C, int Cp, int H, int W, int64_t n_vox, double sx, double sy, double* __restrict__ partial) {
    __shared__ double acc[kMapsMaxViews * kConfMaxChannels];
    __shared__ double cam[kMapsMaxViews * 12];
    extern __shared__ Tap taps[];  // NV > 0: [NV][voxels of a round]
    const int tid = threadIdx.x;
    const int Cq = Cp / 4;
    const int64_t b = blockIdx.x / runs;
    const int run = blockIdx.x % runs;
    for (int i = tid; i < V * Cp; i += kBackwardThreads) acc[i] = 0.0;
    if (tid < V * 12) cam[tid] = camera_entry(proj + (b * V + tid / 12) * 12, tid % 12, sx, sy);
    __syncthreads();
    const int q = tid % Cq, vl = tid / Cq;
    const int vpw = kBackwardThreads / Cq;  // voxels of a round
    const bool working = vl < vpw;
    const int64_t n0 = static_cast<int64_t>(run) * kConfChunk;
    const int64_t n1 = (n0 + kConfChunk < n_vox) ? n0 + kConfChunk : n_vox;
    const size_t map = static_cast<size_t>(H) * W * Cp;
    if constexpr (NV > 0) {
        float sum[NV][4];
#pragma unroll
        for (int u = 0; u < NV; ++u) {
#pragma unroll
            for (int k = 0; k < 4; ++k) sum[u][k] = 0.0f;
        }
        for (int64_t r0 = n0; r0 < n1; r0 += vpw) {
            for (int i = tid; i < vpw * NV; i += kBackwardThreads) {
                const int u = i / vpw, at = i - u * vpw;
                if (r0 + at >= n1) continue;
                double X, Y, Z;
                voxel_position(coord, b, n_vox, r0 + at, X, Y, Z);
                taps[u * vpw + at] = project_tap(cam + u * 12, X, Y, Z, H, W, 1.0f);
            }
            __syncthreads();
            const int64_t n = r0 + vl;
            if (working && n < n1) {
                float g[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) g[k] = (4 * q + k < C) ? g_vol[(b * C + 4 * q + k) * n_vox + n] : 0.0f;
#pragma unroll
                for (int u = 0; u < NV; ++u) {
                    float val[4];
                    sample_quad(feat_cl + (b * V + u) * map, taps[u * vpw + vl], W, Cp, q, val);
#pragma unroll
                    for (int k = 0; k < 4; ++k) sum[u][k] = fmaf(g[k], val[k], sum[u][k]);
                }
            }
            __syncthreads();
        }
        if (working) {
#pragma unroll
            for (int u = 0; u < NV; ++u) {
#pragma unroll
                for (int k = 0; k < 4; ++k) atomicAdd(&acc[u * Cp + 4 * q + k], static_cast<double>(sum[u][k]));
            }
        }
    } else {
        for (int u = 0; u < V; ++u) {
            float sum[4] = {0.0f, 0.0f, 0.0f, 0.0f};
            if (working) {
                for (int64_t n = n0 + vl; n < n1; n += vpw) {
                    double X, Y, Z;
                    voxel_position(coord, b, n_vox, n, X, Y, Z);
                    const Tap t = project_tap(cam + u * 12, X, Y, Z, H, W, 1.0f);
                    float val[4];
                    sample_quad(feat_cl + (b * V + u) * map, t, W, Cp, q, val);
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const float g = (4 * q + k < C) ? g_vol[(b * C + 4 * q + k) * n_vox + n] : 0.0f;
                        sum[k] = fmaf(g, val[k], sum[k]);
                    }
                }
#pragma unroll
                for (int k = 0; k < 4; ++k) atomicAdd(&acc[u * Cp + 4 * q + k], static_cast<double>(sum[k]));
            }
        }
    }
    __syncthreads();
    double* out = partial + (static_cast<size_t>(b) * runs + run) * V * Cp;
    for (int i = tid; i < V * Cp; i += kBackwardThreads) out[i] = acc[i];
}

// One workgroup per (frame, view, channel pair, band of rows): the slice [rows][w][2] of the gradient lives in LDS as
// float64, rows padded to slice_row(W).  Dynamic LDS: band_rows * slice_row(W) doubles.  Channels c0 = 2 * pair and
// c0 + 1.  A map of up to about 9 600 pixels is one band; a larger one is cut into bands of band_rows rows, each walked
// by a workgroup of its own that keeps the taps falling into its rows.
template <int AGG>
__global__ __launch_bounds__(kMapsThreads) void unproject_backward_maps(
    const float* __restrict__ proj, VoxelSource coord, const float* __restrict__ vconf, const float* __restrict__ g_vol, const float* __restrict__ factors,
    const double* __restrict__ conf_partial, int conf_runs, int V, int C, int Cp, int H, int W, int bands, int band_rows,
    int64_t n_vox, double sx, double sy, float* __restrict__ g_feat_cl, float* __restrict__ g_conf) {
    extern __shared__ double slice[];
    __shared__ double cam[12];
    const int tid = threadIdx.x;
    const int pairs = Cp / 2;
    unsigned id = blockIdx.x;
    const int band = static_cast<int>(id % bands);
    id /= bands;
    const int pair = static_cast<int>(id % pairs);
    id /= pairs;
    const int v = static_cast<int>(id % V);
    const int64_t b = id / V;
    const int c0 = 2 * pair;  // c0 + 1 may be padding
    const int RS = slice_row(W);
    const int y_first = band * band_rows;
    const int rows = (H - y_first < band_rows) ? H - y_first : band_rows;
    for (int i = tid; i < rows * RS; i += kMapsThreads) slice[i] = 0.0;
    if (tid < 12) cam[tid] = camera_entry(proj + (b * V + v) * 12, tid, sx, sy);
    __syncthreads();
    const size_t map = static_cast<size_t>(H) * W * Cp;
    const bool wants_conf = AGG == MVX_AGG_CONF && g_conf != nullptr;
    float conf[2] = {0.0f, 0.0f};
    if (AGG == MVX_AGG_CONF) {
#pragma unroll
        for (int k = 0; k < 2; ++k) conf[k] = (c0 + k < C) ? vconf[(b * V + v) * C + c0 + k] : 0.0f;
    }
    for (int64_t n = tid; n < n_vox; n += kMapsThreads) {
        double X, Y, Z;
        voxel_position(coord, b, n_vox, n, X, Y, Z);
        float g[2];
#pragma unroll
        for (int k = 0; k < 2; ++k) g[k] = (c0 + k < C) ? g_vol[(b * C + c0 + k) * n_vox + n] : 0.0f;
        const TapXY t = project_tap_xy(cam, X, Y, Z, H, W, 1.0f);
        float d[2];
        if (AGG == MVX_AGG_MAX) {  // the gradient goes to the first view that holds the maximum (from the pre-pass)
            const float2 f = reinterpret_cast<const float2*>(factors)[(static_cast<size_t>(b) * pairs + pair) * n_vox + n];
            d[0] = static_cast<int>(f.x) == v ? g[0] : 0.0f;
            d[1] = static_cast<int>(f.y) == v ? g[1] : 0.0f;
        } else if (AGG == MVX_AGG_SOFTMAX) {  // the derivative of the aggregate with respect to this view's sample
            const float2 f = reinterpret_cast<const float2*>(factors)[((static_cast<size_t>(b) * V + v) * pairs + pair) * n_vox + n];
            d[0] = g[0] * f.x;
            d[1] = g[1] * f.y;
        } else {
            d[0] = AGG == MVX_AGG_CONF ? g[0] * conf[0] : g[0];
            d[1] = AGG == MVX_AGG_CONF ? g[1] * conf[1] : g[1];
        }
        const int r0 = t.yb - y_first;  // the taps' upper row, relative to this band
        double* p = slice + r0 * RS + 2 * t.xb;
        const float w4[4] = {t.w00, t.w01, t.w10, t.w11};
        const int off[4] = {0, 2, RS, RS + 2};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            // nothing for a tap outside the map or a voxel behind the camera (the reference passes no gradient there),
            // nor for a row that belongs to another band
            if (w4[i] == 0.0f || static_cast<unsigned>(r0 + (i >> 1)) >= static_cast<unsigned>(rows)) continue;
#pragma unroll
            for (int k = 0; k < 2; ++k) atomicAdd(p + off[i] + k, static_cast<double>(d[k]) * static_cast<double>(w4[i]));
        }
    }
    __syncthreads();
    float* out = g_feat_cl + (b * V + v) * map + static_cast<size_t>(y_first) * W * Cp + c0;
    for (int i = tid; i < rows * W; i += kMapsThreads) {
        const int y = i / W, x = i - y * W;
        const double* e = slice + y * RS + 2 * x;
        *reinterpret_cast<float2*>(out + static_cast<size_t>(i) * Cp) = make_float2(static_cast<float>(e[0]), static_cast<float>(e[1]));
    }
    if (wants_conf && band == 0 && tid < 2 && c0 + tid < C) {  // the runs' sums of gradient x sample, added in run order
        double total = 0.0;
        for (int r = 0; r < conf_runs; ++r) total += conf_partial[((static_cast<size_t>(b) * conf_runs + r) * V + v) * Cp + c0 + tid];
        g_conf[(b * V + v) * C + c0 + tid] = static_cast<float>(total);
    }
}

// ---- 3-D soft-argmax ------------------------------------------------------------------------------------------
constexpr int kVolumeThreads = 512;
constexpr int kMaxSegments = 16;

// Four consecutive voxels starting at i (a multiple of 4): probabilities, their gradients and positions.
struct Voxels4 {
    float p[4], gp[4];
    double x[4][3];
};
template <bool VEC>
__device__ __forceinline__ Voxels4 load_voxels(const float* __restrict__ p, const float* __restrict__ gp,
                                               const float* __restrict__ xyz, int64_t i, int count, int vy, int vz) {
    Voxels4 r;
    if (VEC) {  // 16-byte loads: N is a multiple of 4 and the arrays are 16-byte aligned
        const float4 pv = *reinterpret_cast<const float4*>(p + i);
        r.p[0] = pv.x, r.p[1] = pv.y, r.p[2] = pv.z, r.p[3] = pv.w;
        if (gp != nullptr) {
            const float4 gv = *reinterpret_cast<const float4*>(gp + i);
            r.gp[0] = gv.x, r.gp[1] = gv.y, r.gp[2] = gv.z, r.gp[3] = gv.w;
        }
        if (xyz != nullptr) {
            const float4 a = *reinterpret_cast<const float4*>(xyz + 3 * i);
            const float4 b = *reinterpret_cast<const float4*>(xyz + 3 * i + 4);
            const float4 c = *reinterpret_cast<const float4*>(xyz + 3 * i + 8);
            const float f[12] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w, c.x, c.y, c.z, c.w};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
#pragma unroll
                for (int d = 0; d < 3; ++d) r.x[j][d] = f[3 * j + d];
            }
        }
    } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            r.p[j] = j < count ? p[i + j] : 0.0f;
            if (gp != nullptr) r.gp[j] = j < count ? gp[i + j] : 0.0f;
            if (xyz != nullptr) {
#pragma unroll
                for (int d = 0; d < 3; ++d) r.x[j][d] = j < count ? xyz[3 * (i + j) + d] : 0.0f;
            }
        }
    }
    if (gp == nullptr) {
#pragma unroll
        for (int j = 0; j < 4; ++j) r.gp[j] = 0.0f;
    }
    if (xyz == nullptr) {  // the voxel indices themselves
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int64_t row = (i + j) / vz;
            r.x[j][2] = static_cast<double>(i + j - row * vz);
            r.x[j][0] = static_cast<double>(row / vy);
            r.x[j][1] = static_cast<double>(row - (row / vy) * vy);
        }
    }
    return r;
}

// mode 0: softmax; 1: ReLU, coordinates not divided by the mass; 2: ReLU, coordinates divided by the mass.
// Workgroup (volume, segment) leaves in partial[(volume * segments + segment) * 4 ..]: mode 0 sum p G and sum p;
// mode 2 the mass and the first moments of p.  A lane takes four consecutive voxels per round (seg_len is a multiple of 4).
template <bool VEC>
__global__ __launch_bounds__(kVolumeThreads) void soft_argmax_3d_backward_sums(
    const float* __restrict__ probs, const float* __restrict__ coords, int n_volumes, int vy, int vz, int64_t N,
    int segments, int64_t seg_len, int mode, const float* __restrict__ g_coords, const float* __restrict__ g_probs,
    double* __restrict__ partial) {
    __shared__ double slots[4 * (kVolumeThreads / 64)];
    const int64_t vol = blockIdx.x / segments;
    const int seg = blockIdx.x % segments;
    const int64_t frame = vol / n_volumes;
    const float* p = probs + vol * N;
    const float* gp = g_probs != nullptr ? g_probs + vol * N : nullptr;
    const float* xyz = coords != nullptr ? coords + frame * N * 3 : nullptr;
    double gc[3] = {0.0, 0.0, 0.0};
    if (g_coords != nullptr) {
#pragma unroll
        for (int a = 0; a < 3; ++a) gc[a] = g_coords[vol * 3 + a];
    }
    const int64_t i0 = seg * seg_len, i1 = (i0 + seg_len < N) ? i0 + seg_len : N;
    double s[4] = {0.0, 0.0, 0.0, 0.0};
    for (int64_t i = i0 + 4 * threadIdx.x; i < i1; i += 4 * kVolumeThreads) {
        const int count = (i1 - i < 4) ? static_cast<int>(i1 - i) : 4;
        const Voxels4 v = load_voxels<VEC>(p, gp, xyz, i, count, vy, vz);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const double pi = v.p[j];  // zero beyond the segment's end
            if (mode == 0) {
                const double G = static_cast<double>(v.gp[j]) + gc[0] * v.x[j][0] + gc[1] * v.x[j][1] + gc[2] * v.x[j][2];
                s[0] += pi * G;
                s[1] += pi;
            } else {
                s[0] += pi;
                s[1] += pi * v.x[j][0];
                s[2] += pi * v.x[j][1];
                s[3] += pi * v.x[j][2];
            }
        }
    }
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const double w = wave_sum(s[k]);
        if (lane == 0) slots[k * (kVolumeThreads / 64) + wave] = w;
    }
    __syncthreads();
    if (threadIdx.x < 4) {
        double total = 0.0;
        for (int w = 0; w < kVolumeThreads / 64; ++w) total += slots[threadIdx.x * (kVolumeThreads / 64) + w];
        partial[(vol * segments + seg) * 4 + threadIdx.x] = total;
    }
}

template <bool VEC>
__global__ __launch_bounds__(kVolumeThreads) void soft_argmax_3d_backward_write(
    const float* __restrict__ probs, const float* __restrict__ coords, int n_volumes, int vy, int vz, int64_t N,
    int segments, int64_t seg_len, float multiplier, int mode, const float* __restrict__ g_coords,
    const float* __restrict__ g_probs, const double* __restrict__ partial, float* __restrict__ g_volumes) {
    const int64_t vol = blockIdx.x / segments;
    const int seg = blockIdx.x % segments;
    const int64_t frame = vol / n_volumes;
    const float* p = probs + vol * N;
    const float* gp = g_probs != nullptr ? g_probs + vol * N : nullptr;
    const float* xyz = coords != nullptr ? coords + frame * N * 3 : nullptr;
    float* gx = g_volumes + vol * N;
    double gc[3] = {0.0, 0.0, 0.0};
    if (g_coords != nullptr) {
#pragma unroll
        for (int a = 0; a < 3; ++a) gc[a] = g_coords[vol * 3 + a];
    }
    double s[4] = {0.0, 0.0, 0.0, 0.0};
    if (mode != 1) {  // the segments' partial sums, added in segment order by every thread alike
        for (int k = 0; k < segments; ++k) {
#pragma unroll
            for (int a = 0; a < 4; ++a) s[a] += partial[(vol * segments + k) * 4 + a];
        }
    }
    const double m = static_cast<double>(multiplier);
    // softmax: the float32 probabilities add up to 1 only to rounding; G carries a large common offset (gc . X with X
    // in the thousands), so sum p G is divided by sum p, which removes that offset's share of the rounding
    const double dot = mode == 0 ? s[0] / s[1] : 0.0, mass = s[0];
    const double c[3] = {s[1] / mass, s[2] / mass, s[3] / mass};
    const int64_t i0 = seg * seg_len, i1 = (i0 + seg_len < N) ? i0 + seg_len : N;
    for (int64_t i = i0 + 4 * threadIdx.x; i < i1; i += 4 * kVolumeThreads) {
        const int count = (i1 - i < 4) ? static_cast<int>(i1 - i) : 4;
        const Voxels4 v = load_voxels<VEC>(p, gp, xyz, i, count, vy, vz);
        float out[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const double gpi = v.gp[j];
            if (mode == 0) {
                const double G = gpi + gc[0] * v.x[j][0] + gc[1] * v.x[j][1] + gc[2] * v.x[j][2];
                out[j] = static_cast<float>(m * static_cast<double>(v.p[j]) * (G - dot));
            } else if (mode == 1) {
                const double G = gpi + gc[0] * v.x[j][0] + gc[1] * v.x[j][1] + gc[2] * v.x[j][2];
                out[j] = v.p[j] > 0.0f ? static_cast<float>(m * G) : 0.0f;
            } else {
                const double G = gpi + (gc[0] * (v.x[j][0] - c[0]) + gc[1] * (v.x[j][1] - c[1]) + gc[2] * (v.x[j][2] - c[2])) / mass;
                out[j] = v.p[j] > 0.0f ? static_cast<float>(m * G) : 0.0f;
            }
        }
        if (VEC) {
            *reinterpret_cast<float4*>(gx + i) = make_float4(out[0], out[1], out[2], out[3]);
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (j < count) gx[i + j] = out[j];
            }
        }
    }
}

// Segments per volume: enough workgroups for the chip, segments of at least 4096 voxels.
inline int volume_segments(int64_t count, int64_t N) {
    int64_t s = (2048 + count - 1) / count;
    const int64_t by_size = (N + 4095) / 4096;
    if (s > by_size) s = by_size;
    if (s > kMaxSegments) s = kMaxSegments;
    return s < 1 ? 1 : static_cast<int>(s);
}

// Rows of a map that one slice holds (all of them for maps of up to about 9 600 pixels); 0: not even two rows fit.
int slice_band_rows(int h, int w) {
    const size_t row_bytes = static_cast<size_t>(slice_row(w)) * sizeof(double);
    const size_t fit = kMapsMaxSliceBytes / row_bytes;
    if (fit < 2) return 0;
    return fit >= static_cast<size_t>(h) ? h : static_cast<int>(fit);
}

bool slices_fit(int h, int w, int n_views, int n_channels, int aggregation) {
    const int cp = (n_channels + 3) / 4 * 4;
    if (slice_band_rows(h, w) == 0 || cp / 4 > kBackwardThreads) return false;
    // the pre-passes keep per-view data of a workgroup in LDS
    if (aggregation == MVX_AGG_CONF && (cp > kConfMaxChannels || n_views > kMapsMaxViews)) return false;
    if ((aggregation == MVX_AGG_MAX || aggregation == MVX_AGG_SOFTMAX) &&
        static_cast<size_t>(n_views) * (kBackwardThreads / (cp / 4)) * sizeof(Tap) > 48 * 1024)
        return false;
    return !(kernel_selection() & MVX_SELECT_NO_BACKWARD_IN_LDS);
}

}  // namespace

}  // namespace mvx

extern "C" size_t mvx_unproject_backward_workspace_bytes(int64_t n_frames, int n_views, int n_channels, int h, int w,
                                                         int vx, int vy, int vz, int aggregation) {
    if (n_frames <= 0 || n_views <= 0 || n_channels <= 0 || h <= 0 || w <= 0 || vx <= 0 || vy <= 0 || vz <= 0) return 0;
    if (aggregation == MVX_AGG_SUM || !mvx::slices_fit(h, w, n_views, n_channels, aggregation)) return 0;
    const size_t cp = static_cast<size_t>((n_channels + 3) / 4 * 4);
    const size_t n_voxels = static_cast<size_t>(vx) * vy * vz;
    if (aggregation == MVX_AGG_CONF)  // per run of voxels, the sums of gradient x sample (float64)
        return static_cast<size_t>(n_frames) * ((n_voxels + mvx::kConfChunk - 1) / mvx::kConfChunk) * n_views * cp * sizeof(double);
    // softmax: a factor per (view, channel, voxel); max: the index of the first maximal view per (channel, voxel)
    return static_cast<size_t>(n_frames) * (aggregation == MVX_AGG_SOFTMAX ? n_views : 1) * cp * n_voxels * sizeof(float);
}

extern "C" int mvx_unproject_backward(const float* features_cl, const float* proj, const float* coords,
                                      const float* vol_conf, const float* grad_volumes, int64_t n_frames, int n_views,
                                      int n_channels, int h, int w, int vx, int vy, int vz, int aggregation,
                                      float* grad_features_cl, float* grad_vol_conf, void* workspace,
                                      size_t workspace_bytes, void* stream) {
    using namespace mvx;
    const char* name = "mvx_unproject_backward";
    MVX_REQUIRE(n_frames >= 0 && n_views > 0 && n_channels > 0 && h >= 2 && w >= 2 && vx >= 0 && vy >= 0 && vz >= 0,
                MVX_ERR_INVALID_ARGUMENT, "%s: bad shape n_frames=%lld n_views=%d n_channels=%d h=%d w=%d grid=%dx%dx%d",
                name, (long long)n_frames, n_views, n_channels, h, w, vx, vy, vz);
    MVX_REQUIRE(aggregation >= MVX_AGG_SUM && aggregation <= MVX_AGG_SOFTMAX, MVX_ERR_INVALID_ARGUMENT,
                "%s: unknown aggregation %d", name, aggregation);
    const int64_t n_voxels = static_cast<int64_t>(vx) * vy * vz;
    const int C = n_channels, Cp = (C + 3) / 4 * 4, V = n_views;
    const size_t map = static_cast<size_t>(h) * w * Cp;
    if (n_frames == 0) return MVX_OK;
    MVX_REQUIRE(grad_features_cl != nullptr && aligned16(grad_features_cl), MVX_ERR_INVALID_ARGUMENT,
                "%s: grad_features_cl must be a 16-byte aligned device pointer", name);
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int band_rows = slice_band_rows(h, w);
    const int bands = band_rows > 0 ? (h + band_rows - 1) / band_rows : 1;
    const bool in_lds = slices_fit(h, w, V, C, aggregation) && n_voxels > 0 &&
                        n_frames * V * (Cp / 2) * bands <= 0x7fffffffLL;
    if (!in_lds) {  // the any-shape kernel adds into gradients that start at zero (and an empty grid leaves them zero)
        hipError_t e = hipMemsetAsync(grad_features_cl, 0, static_cast<size_t>(n_frames) * V * map * sizeof(float), st);
        if (e == hipSuccess && grad_vol_conf != nullptr)
            e = hipMemsetAsync(grad_vol_conf, 0, static_cast<size_t>(n_frames) * V * C * sizeof(float), st);
        MVX_REQUIRE(e == hipSuccess, MVX_ERR_HIP, "%s: clearing the gradients failed: %s", name, hipGetErrorString(e));
    }
    if (n_voxels == 0) return MVX_OK;
    const bool needs_maps = aggregation == MVX_AGG_MAX || aggregation == MVX_AGG_SOFTMAX ||
                            (aggregation == MVX_AGG_CONF && grad_vol_conf != nullptr);
    MVX_REQUIRE(proj != nullptr && coords != nullptr && grad_volumes != nullptr, MVX_ERR_INVALID_ARGUMENT,
                "%s: null pointer argument", name);
    MVX_REQUIRE(!needs_maps || (features_cl != nullptr && aligned16(features_cl)), MVX_ERR_INVALID_ARGUMENT,
                "%s: this aggregation needs the forward call's maps (channel-last, 16-byte aligned)", name);
    MVX_REQUIRE(aggregation != MVX_AGG_CONF || vol_conf != nullptr, MVX_ERR_INVALID_ARGUMENT,
                "%s: confidence aggregation needs vol_conf", name);
    MVX_REQUIRE(map < (1ULL << 30), MVX_ERR_UNSUPPORTED, "%s: one feature map exceeds 2^30 elements (4 GB)", name);
    MVX_REQUIRE(n_frames < (1LL << 31), MVX_ERR_UNSUPPORTED, "%s: too many frames", name);
    const double sx = static_cast<double>(w - 1) / static_cast<double>(h);
    const double sy = static_cast<double>(h - 1) / static_cast<double>(w);
    // rounds of frames, so that a one-lane-per-voxel launch stays below 2^31 workgroups
    const int64_t per_round = n_voxels >= (1LL << 38) ? 1 : (1LL << 38) / n_voxels;
    if (in_lds) {
        const bool with_factors = aggregation == MVX_AGG_MAX || aggregation == MVX_AGG_SOFTMAX;
        float* factors = nullptr;
        if (with_factors) {
            const size_t need = mvx_unproject_backward_workspace_bytes(n_frames, V, C, h, w, vx, vy, vz, aggregation);
            MVX_REQUIRE(workspace != nullptr && workspace_bytes >= need, MVX_ERR_WORKSPACE,
                        "%s: workspace of %zu bytes is too small (%zu needed)", name, workspace_bytes, need);
            MVX_REQUIRE(aligned16(workspace), MVX_ERR_INVALID_ARGUMENT, "%s: workspace must be 16-byte aligned", name);
            factors = static_cast<float*>(workspace);
            const size_t per_frame = need / static_cast<size_t>(n_frames) / sizeof(float);
            const int vpw = kBackwardThreads / (Cp / 4);
            const size_t taps_lds = static_cast<size_t>(V) * vpw * sizeof(Tap);
            for (int64_t f0 = 0; f0 < n_frames; f0 += per_round) {
                const int64_t nf = (n_frames - f0 < per_round) ? n_frames - f0 : per_round;
                const int64_t n_total = nf * n_voxels;
                const int64_t blocks = (n_total + vpw - 1) / vpw;
                MVX_REQUIRE(blocks <= 0x7fffffffLL, MVX_ERR_UNSUPPORTED, "%s: voxel grid too large", name);
                VoxelSource cd{};
                cd.volume = coords + f0 * n_voxels * 3;
                cd.first_frame = f0;
                cd.vy = vy;
                cd.vz = vz;
#define MVX_LAUNCH_FACTORS(AGG, NV)                                                                                    \
    hipLaunchKernelGGL((unproject_view_factors<AGG, NV>), dim3(static_cast<unsigned>(blocks)), dim3(kBackwardThreads),   \
                       taps_lds, st, features_cl + f0 * V * map, proj + f0 * V * 12, cd, n_total, V, Cp, h, w, n_voxels, \
                       sx, sy, factors + f0 * per_frame)
                if (aggregation == MVX_AGG_MAX) {
                    MVX_LAUNCH_FACTORS(MVX_AGG_MAX, 0);
                } else if (V == 4) {
                    MVX_LAUNCH_FACTORS(MVX_AGG_SOFTMAX, 4);
                } else if (V == 3) {
                    MVX_LAUNCH_FACTORS(MVX_AGG_SOFTMAX, 3);
                } else if (V == 2) {
                    MVX_LAUNCH_FACTORS(MVX_AGG_SOFTMAX, 2);
                } else {
                    MVX_LAUNCH_FACTORS(MVX_AGG_SOFTMAX, 0);
                }
#undef MVX_LAUNCH_FACTORS
                MVX_CHECK_LAUNCH("mvx_unproject_backward (view factors)");
            }
        }
        VoxelSource cd{};
        cd.volume = coords;
        cd.vy = vy;
        cd.vz = vz;
        double* conf_partial = nullptr;
        const int conf_runs = static_cast<int>((n_voxels + kConfChunk - 1) / kConfChunk);
        if (aggregation == MVX_AGG_CONF && grad_vol_conf != nullptr) {
            const size_t need = mvx_unproject_backward_workspace_bytes(n_frames, V, C, h, w, vx, vy, vz, aggregation);
            MVX_REQUIRE(workspace != nullptr && workspace_bytes >= need, MVX_ERR_WORKSPACE,
                        "%s: workspace of %zu bytes is too small (%zu needed)", name, workspace_bytes, need);
            MVX_REQUIRE(aligned16(workspace), MVX_ERR_INVALID_ARGUMENT, "%s: workspace must be 16-byte aligned", name);
            MVX_REQUIRE(n_frames * conf_runs <= 0x7fffffffLL, MVX_ERR_UNSUPPORTED, "%s: voxel grid too large", name);
            conf_partial = static_cast<double*>(workspace);
            const dim3 conf_grid(static_cast<unsigned>(n_frames * conf_runs));
            const size_t conf_lds = static_cast<size_t>(V) * (kBackwardThreads / (Cp / 4)) * sizeof(Tap);
#define MVX_LAUNCH_CONF_SUMS(NV, LDS)                                                                                 \
    hipLaunchKernelGGL(unproject_confidence_sums<NV>, conf_grid, dim3(kBackwardThreads), LDS, st, features_cl, proj, cd, \
                       grad_volumes, conf_runs, V, C, Cp, h, w, n_voxels, sx, sy, conf_partial)
            if (V == 4) {
                MVX_LAUNCH_CONF_SUMS(4, conf_lds);
            } else if (V == 3) {
                MVX_LAUNCH_CONF_SUMS(3, conf_lds);
            } else if (V == 2) {
                MVX_LAUNCH_CONF_SUMS(2, conf_lds);
            } else {
                MVX_LAUNCH_CONF_SUMS(0, 0);
            }
#undef MVX_LAUNCH_CONF_SUMS
            MVX_CHECK_LAUNCH("mvx_unproject_backward (confidence sums)");
        }
        const size_t lds = static_cast<size_t>(band_rows) * slice_row(w) * sizeof(double);
        const dim3 grid(static_cast<unsigned>(n_frames * V * (Cp / 2) * bands));
#define MVX_LAUNCH_BACKWARD_MAPS(AGG)                                                                                  \
    do {                                                                                                               \
        hipError_t ea = hipFuncSetAttribute(reinterpret_cast<const void*>(unproject_backward_maps<AGG>),               \
                                            hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds));        \
        MVX_REQUIRE(ea == hipSuccess, MVX_ERR_HIP, "%s: %zu bytes of LDS refused: %s", name, lds, hipGetErrorString(ea)); \
        hipLaunchKernelGGL(unproject_backward_maps<AGG>, grid, dim3(kMapsThreads), lds, st, proj, cd, vol_conf, grad_volumes, \
                           factors, conf_partial, conf_runs, V, C, Cp, h, w, bands, band_rows, n_voxels, sx, sy,       \
                           grad_features_cl, grad_vol_conf);                                                           \
    } while (0)
        switch (aggregation) {
            case MVX_AGG_SUM: MVX_LAUNCH_BACKWARD_MAPS(MVX_AGG_SUM); break;
            case MVX_AGG_CONF: MVX_LAUNCH_BACKWARD_MAPS(MVX_AGG_CONF); break;
            case MVX_AGG_MAX: MVX_LAUNCH_BACKWARD_MAPS(MVX_AGG_MAX); break;
            default: MVX_LAUNCH_BACKWARD_MAPS(MVX_AGG_SOFTMAX); break;
        }
#undef MVX_LAUNCH_BACKWARD_MAPS
        MVX_CHECK_LAUNCH(name);
        return MVX_OK;
    }
    for (int64_t f0 = 0; f0 < n_frames; f0 += per_round) {
        const int64_t nf = (n_frames - f0 < per_round) ? n_frames - f0 : per_round;
        const int64_t n_total = nf * n_voxels;
        const int64_t blocks = (n_total + kBackwardThreads - 1) / kBackwardThreads;
        MVX_REQUIRE(blocks <= 0x7fffffffLL, MVX_ERR_UNSUPPORTED, "%s: voxel grid too large", name);
        VoxelSource cd{};
        cd.volume = coords + f0 * n_voxels * 3;
        cd.first_frame = f0;
        cd.vy = vy;
        cd.vz = vz;
        const float* fc = features_cl != nullptr ? features_cl + f0 * V * map : nullptr;
        const float* vc = vol_conf != nullptr ? vol_conf + f0 * V * C : nullptr;
        float* gconf = grad_vol_conf != nullptr ? grad_vol_conf + f0 * V * C : nullptr;
#define MVX_LAUNCH_BACKWARD(AGG)                                                                                     \
    hipLaunchKernelGGL(unproject_backward<AGG>, dim3(static_cast<unsigned>(blocks)), dim3(kBackwardThreads), 0, st, fc, \
                       proj + f0 * V * 12, cd, vc, grad_volumes + f0 * C * n_voxels, n_total, V, C, Cp, h, w, n_voxels, \
                       sx, sy, grad_features_cl + f0 * V * map, gconf)
        switch (aggregation) {
            case MVX_AGG_SUM: MVX_LAUNCH_BACKWARD(MVX_AGG_SUM); break;
            case MVX_AGG_CONF: MVX_LAUNCH_BACKWARD(MVX_AGG_CONF); break;
            case MVX_AGG_MAX: MVX_LAUNCH_BACKWARD(MVX_AGG_MAX); break;
            default: MVX_LAUNCH_BACKWARD(MVX_AGG_SOFTMAX); break;
        }
#undef MVX_LAUNCH_BACKWARD
        MVX_CHECK_LAUNCH(name);
    }
    return MVX_OK;
}

extern "C" size_t mvx_soft_argmax_3d_backward_workspace_bytes(int64_t n_frames, int n_volumes, int sx, int sy, int sz) {
    if (n_frames <= 0 || n_volumes <= 0 || sx <= 0 || sy <= 0 || sz <= 0) return 0;
    return static_cast<size_t>(n_frames) * n_volumes * mvx::kMaxSegments * 4 * sizeof(double);
}

extern "C" int mvx_soft_argmax_3d_backward(const float* volumes_out, const float* coords, int64_t n_frames,
                                           int n_volumes, int sx, int sy, int sz, float multiplier, int softmax,
                                           const float* grad_coordinates, const float* grad_volumes_out,
                                           float* grad_volumes, void* workspace, size_t workspace_bytes, void* stream) {
    using namespace mvx;
    const char* name = "mvx_soft_argmax_3d_backward";
    MVX_REQUIRE(n_frames >= 0 && n_volumes >= 0 && sx > 0 && sy > 0 && sz > 0, MVX_ERR_INVALID_ARGUMENT,
                "%s: bad shape n_frames=%lld n_volumes=%d grid=%dx%dx%d", name, (long long)n_frames, n_volumes, sx, sy, sz);
    const int64_t count = n_frames * n_volumes;
    if (count == 0) return MVX_OK;
    MVX_REQUIRE(volumes_out != nullptr && grad_volumes != nullptr, MVX_ERR_INVALID_ARGUMENT, "%s: null pointer argument",
                name);
    const int64_t N = static_cast<int64_t>(sx) * sy * sz;
    const int segments = volume_segments(count, N);
    MVX_REQUIRE(count * segments <= 0x7fffffffLL, MVX_ERR_UNSUPPORTED, "%s: too many volumes", name);
    // the reference divides by the mass only in index space (mvn/utils/op.py:73-76)
    const int mode = softmax ? 0 : (coords != nullptr ? 1 : 2);
    const int64_t seg_len = ((N + segments - 1) / segments + 3) / 4 * 4;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const dim3 grid(static_cast<unsigned>(count * segments));
    double* partial = static_cast<double*>(workspace);
    const bool vec = N % 4 == 0 && aligned16(volumes_out) && aligned16(grad_volumes) &&
                     (coords == nullptr || aligned16(coords)) && (grad_volumes_out == nullptr || aligned16(grad_volumes_out));
    if (mode != 1) {
        MVX_REQUIRE(workspace != nullptr &&
                        workspace_bytes >= mvx_soft_argmax_3d_backward_workspace_bytes(n_frames, n_volumes, sx, sy, sz),
                    MVX_ERR_WORKSPACE, "%s: workspace of %zu bytes is too small", name, workspace_bytes);
        MVX_REQUIRE((reinterpret_cast<uintptr_t>(workspace) & 7u) == 0, MVX_ERR_INVALID_ARGUMENT,
                    "%s: workspace must be 8-byte aligned", name);
        if (vec)
            hipLaunchKernelGGL(soft_argmax_3d_backward_sums<true>, grid, dim3(kVolumeThreads), 0, st, volumes_out, coords,
                               n_volumes, sy, sz, N, segments, seg_len, mode, grad_coordinates, grad_volumes_out, partial);
        else
            hipLaunchKernelGGL(soft_argmax_3d_backward_sums<false>, grid, dim3(kVolumeThreads), 0, st, volumes_out, coords,
                               n_volumes, sy, sz, N, segments, seg_len, mode, grad_coordinates, grad_volumes_out, partial);
        MVX_CHECK_LAUNCH("mvx_soft_argmax_3d_backward (sums)");
    }
    if (vec)
        hipLaunchKernelGGL(soft_argmax_3d_backward_write<true>, grid, dim3(kVolumeThreads), 0, st, volumes_out, coords,
                           n_volumes, sy, sz, N, segments, seg_len, multiplier, mode, grad_coordinates, grad_volumes_out,
                           partial, grad_volumes);
    else
        hipLaunchKernelGGL(soft_argmax_3d_backward_write<false>, grid, dim3(kVolumeThreads), 0, st, volumes_out, coords,
                           n_volumes, sy, sz, N, segments, seg_len, multiplier, mode, grad_coordinates, grad_volumes_out,
                           partial, grad_volumes);
    MVX_CHECK_LAUNCH(name);
    return MVX_OK;
}
