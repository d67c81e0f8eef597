// Backward kernels of the two volumetric operations: what torch autograd computes when the reference trains through
// unproject_heatmaps (mvn/utils/op.py:99-166: grid_sample per view, zeroing of voxels behind the camera, aggregation
// over views) and integrate_tensor_3d / integrate_tensor_3d_with_coordinates (mvn/utils/op.py:50-96: softmax or ReLU,
// then the expectation of the coordinates); the reference's training step is train.py:281-290.
//
// Unprojection: one lane per voxel.  The lane projects its voxel into every view with the forward kernels' own
// routine (unproject_taps.h: the same taps, the same weights), samples the maps again where the aggregation needs
// the sampled values (max, softmax, and the gradient of the confidences), and adds weight x gradient to the four
// taps of every channel with float32 atomic adds at agent scope into a channel-last gradient [frame][view][h][w][Cp]
// that the entry point has zeroed.  The order of those additions is not fixed: two runs differ in the last bits.
//
// 3-D soft-argmax: one workgroup per volume.  With p the forward's output volume, X_i the coordinates of voxel i,
// gc the gradient of the volume's coordinates and gp_i the gradient of p_i, G_i = gp_i + gc . X_i and
//   softmax:                 dL/dx_i = m p_i (G_i - sum_j p_j G_j)
//   ReLU (with coordinates): dL/dx_i = m G_i where p_i > 0
//   ReLU (index space, where the reference divides the coordinates by the mass M = sum p; c = the coordinates):
//                            dL/dx_i = m (gp_i + gc . (X_i - c) / M) where p_i > 0
// (m = the multiplier folded into the forward call).  G and the sums are float64.
#include "grid_coords.h"
#include "unproject_taps.h"

namespace mvx {

namespace {

constexpr int kBackwardThreads = 256;
constexpr int kVolumeThreads = 1024;

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

// Sum over the workgroup, returned to every thread.  `slots`: kVolumeThreads / 64 doubles of LDS per value.
template <int N>
__device__ __forceinline__ void block_sums(double (&v)[N], double* slots) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, waves = blockDim.x >> 6;
#pragma unroll
    for (int i = 0; i < N; ++i) {
        const double s = wave_sum(v[i]);
        if (lane == 0) slots[i * (kVolumeThreads / 64) + wave] = s;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < N; ++i) {
        double s = 0.0;
        for (int w = 0; w < waves; ++w) s += slots[i * (kVolumeThreads / 64) + w];
        v[i] = s;
    }
    __syncthreads();
}

// mode 0: softmax; 1: ReLU, coordinates not divided by the mass; 2: ReLU, coordinates divided by the mass.
// coords: [frames][N][3], or null for the voxel indices themselves.
__global__ __launch_bounds__(kVolumeThreads) void soft_argmax_3d_backward(
    const float* __restrict__ probs, const float* __restrict__ coords, int n_volumes, int vy, int vz, int64_t N,
    float multiplier, int mode, const float* __restrict__ g_coords, const float* __restrict__ g_probs,
    float* __restrict__ g_volumes) {
    __shared__ double slots[4 * (kVolumeThreads / 64)];
    const int64_t vol = blockIdx.x;
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
    auto position = [&](int64_t i, double (&x)[3]) {
        if (xyz != nullptr) {
            x[0] = xyz[3 * i];
            x[1] = xyz[3 * i + 1];
            x[2] = xyz[3 * i + 2];
        } else {
            const int64_t row = i / vz;
            x[2] = static_cast<double>(i - row * vz);
            x[0] = static_cast<double>(row / vy);
            x[1] = static_cast<double>(row - (row / vy) * vy);
        }
    };
    const double m = static_cast<double>(multiplier);
    if (mode == 1) {
        for (int64_t i = threadIdx.x; i < N; i += kVolumeThreads) {
            double x[3];
            position(i, x);
            const double G = (gp != nullptr ? static_cast<double>(gp[i]) : 0.0) + gc[0] * x[0] + gc[1] * x[1] + gc[2] * x[2];
            gx[i] = p[i] > 0.0f ? static_cast<float>(m * G) : 0.0f;
        }
        return;
    }
    // first sweep: softmax: sum p G; normalised ReLU: the mass and the first moments
    double s[4] = {0.0, 0.0, 0.0, 0.0};
    for (int64_t i = threadIdx.x; i < N; i += kVolumeThreads) {
        double x[3];
        position(i, x);
        const double pi = p[i];
        if (mode == 0) {
            const double G = (gp != nullptr ? static_cast<double>(gp[i]) : 0.0) + gc[0] * x[0] + gc[1] * x[1] + gc[2] * x[2];
            s[0] += pi * G;
        } else {
            s[0] += pi;
            s[1] += pi * x[0];
            s[2] += pi * x[1];
            s[3] += pi * x[2];
        }
    }
    block_sums(s, slots);
    if (mode == 0) {
        const double dot = s[0];
        for (int64_t i = threadIdx.x; i < N; i += kVolumeThreads) {
            double x[3];
            position(i, x);
            const double G = (gp != nullptr ? static_cast<double>(gp[i]) : 0.0) + gc[0] * x[0] + gc[1] * x[1] + gc[2] * x[2];
            gx[i] = static_cast<float>(m * static_cast<double>(p[i]) * (G - dot));
        }
    } else {
        const double mass = s[0];
        const double c[3] = {s[1] / mass, s[2] / mass, s[3] / mass};
        for (int64_t i = threadIdx.x; i < N; i += kVolumeThreads) {
            double x[3];
            position(i, x);
            const double G = (gp != nullptr ? static_cast<double>(gp[i]) : 0.0) +
                             (gc[0] * (x[0] - c[0]) + gc[1] * (x[1] - c[1]) + gc[2] * (x[2] - c[2])) / mass;
            gx[i] = p[i] > 0.0f ? static_cast<float>(m * G) : 0.0f;
        }
    }
}

}  // namespace

}  // namespace mvx

extern "C" int mvx_unproject_backward(const float* features_cl, const float* proj, const float* coords,
                                      const float* vol_conf, const float* grad_volumes, int64_t n_frames, int n_views,
                                      int n_channels, int h, int w, int vx, int vy, int vz, int aggregation,
                                      float* grad_features_cl, float* grad_vol_conf, void* stream) {
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
    hipError_t e = hipMemsetAsync(grad_features_cl, 0, static_cast<size_t>(n_frames) * V * map * sizeof(float), st);
    if (e == hipSuccess && grad_vol_conf != nullptr)
        e = hipMemsetAsync(grad_vol_conf, 0, static_cast<size_t>(n_frames) * V * C * sizeof(float), st);
    MVX_REQUIRE(e == hipSuccess, MVX_ERR_HIP, "%s: clearing the gradients failed: %s", name, hipGetErrorString(e));
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
    const double sx = static_cast<double>(w - 1) / static_cast<double>(h);
    const double sy = static_cast<double>(h - 1) / static_cast<double>(w);
    MVX_REQUIRE(n_frames < (1LL << 31), MVX_ERR_UNSUPPORTED, "%s: too many frames", name);
    // rounds of frames, so that a launch stays below 2^31 workgroups
    const int64_t per_round = n_voxels >= (1LL << 38) ? 1 : (1LL << 38) / n_voxels;
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

extern "C" int mvx_soft_argmax_3d_backward(const float* volumes_out, const float* coords, int64_t n_frames,
                                           int n_volumes, int sx, int sy, int sz, float multiplier, int softmax,
                                           const float* grad_coordinates, const float* grad_volumes_out,
                                           float* grad_volumes, void* stream) {
    using namespace mvx;
    const char* name = "mvx_soft_argmax_3d_backward";
    MVX_REQUIRE(n_frames >= 0 && n_volumes >= 0 && sx > 0 && sy > 0 && sz > 0, MVX_ERR_INVALID_ARGUMENT,
                "%s: bad shape n_frames=%lld n_volumes=%d grid=%dx%dx%d", name, (long long)n_frames, n_volumes, sx, sy, sz);
    const int64_t count = n_frames * n_volumes;
    if (count == 0) return MVX_OK;
    MVX_REQUIRE(volumes_out != nullptr && grad_volumes != nullptr, MVX_ERR_INVALID_ARGUMENT, "%s: null pointer argument",
                name);
    MVX_REQUIRE(count <= 0x7fffffffLL, MVX_ERR_UNSUPPORTED, "%s: too many volumes", name);
    const int64_t N = static_cast<int64_t>(sx) * sy * sz;
    // the reference divides by the mass only in index space (mvn/utils/op.py:73-76)
    const int mode = softmax ? 0 : (coords != nullptr ? 1 : 2);
    hipLaunchKernelGGL(soft_argmax_3d_backward, dim3(static_cast<unsigned>(count)), dim3(kVolumeThreads), 0,
                       static_cast<hipStream_t>(stream), volumes_out, coords, n_volumes, sy, sz, N, multiplier, mode,
                       grad_coordinates, grad_volumes_out, grad_volumes);
    MVX_CHECK_LAUNCH(name);
    return MVX_OK;
}
