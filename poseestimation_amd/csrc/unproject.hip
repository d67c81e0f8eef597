// Unprojection of per-view feature maps into a voxel grid with view aggregation.
//
// Step 1 copies the feature maps of a few frames from [view][channel][h][w] to channel-last
// [view][h][w][channel] (channels padded to a multiple of 4), so that the C channels of one bilinear tap are
// one contiguous C*4-byte run (a full 128-byte line for C = 32).  The copy of a chunk of frames is consumed
// straight away by step 2 and is small enough to stay in L2 / Infinity Cache.
//
// Step 2, tile kernel (C = 32): a workgroup owns 128 consecutive voxels of one frame.
//   phase A  one thread per voxel projects it into every view and leaves the four bilinear weights (zero for
//            taps outside the map and for voxels behind the camera) and the tap address in LDS;
//   phase B  8 lanes per voxel, 4 channels per lane: four 16-byte loads per view fetch the 2x2 taps (each
//            wave-instruction consumes whole 128-byte lines), the view aggregation runs in registers
//            (sum / confidence-weighted / max / online softmax over views), results go to an LDS tile
//            [channel][voxel] (XOR-swizzled so both sides are conflict-free);
//   phase C  the tile is stored with 16 bytes per lane: 512 contiguous bytes per channel.
// The per-view volumes of the reference (V x C x N floats per frame) are never materialised.
// Workgroup ids are remapped so that each XCD works through consecutive tiles of the same frame and its L2
// holds just that frame's feature maps.
//
// Step 2, generic kernel (any C, V, grid): one lane per (voxel, 4 channels), no LDS, direct stores.
//
// Measured on MI355X (8 frames per launch, V = 4): the tile kernel is bound by the vector-memory return path
// (64 B/clk per CU; 512 B of taps per voxel and view whether or not the lines hit in L1), at about 21 us per
// frame against a 6.6 us HBM floor.  A variant that fed the taps from an LDS patch staged per view and 2x2x32
// tile was measured slower and is not kept (its stage / sample phases serialise at 2-3 workgroups per CU:
// 3.0 ms against 1.6 ms for 64 frames); taps read from LDS with no staging cost bound what such a design
// could reach at about 0.9-1.1 ms.
//
// HBM traffic per frame: V*C*h*w*4 (features) + N*12 (coordinates) + C*N*4 (volume written).
#include <stdlib.h>

#include "mvx_common.h"

namespace mvx {

namespace {

constexpr int kThreads = 256;
constexpr int kChunkFrames = 8;  // frames per channel-last copy + sampling round (one per XCD)

struct Tap {
    float w00, w01, w10, w11;  // weights of (yb, xb), (yb, xb+1), (yb+1, xb), (yb+1, xb+1)
    unsigned pixel;            // yb * W + xb, always inside the map
};

constexpr float kLog2e = 1.44269504088896340736f;
constexpr float kLn2 = 0.69314718055994530942f;

// Projects one voxel and returns its bilinear taps with zero padding.
// The reference maps the projected pixel (x, y) = (u/z, v/z) through gx = 2(x/H - 0.5), gy = 2(y/W - 0.5)
// (H for x and W for y) and grid_sample(align_corners=True): ix = (gx + 1)/2 * (W - 1) = x/H * (W - 1),
// iy = y/W * (H - 1).  Here the camera Pd is float64 with rows 0 and 1 already multiplied by (W-1)/H and
// (H-1)/W, so ix = u'/z, iy = v'/z; the position is evaluated in float64 and only the interpolation weights
// are rounded to float32 (the reference's float32 position carries about 1e-5 px of rounding noise, which the
// feature gradient would multiply).  1/z is v_rcp_f32 refined by two Newton steps in float64 (relative error
// below 1e-15) instead of a float64 division.  A voxel with z <= 0 gets all-zero weights; `wscale`
// multiplies the weights (log2 e for the softmax aggregation, which works in base 2).
__device__ __forceinline__ Tap project_tap(const double* __restrict__ Pd, double X, double Y, double Z, int H, int W,
                                           float wscale) {
    const double u = fma(X, Pd[0], fma(Y, Pd[1], fma(Z, Pd[2], Pd[3])));
    const double v = fma(X, Pd[4], fma(Y, Pd[5], fma(Z, Pd[6], Pd[7])));
    const double z0 = fma(X, Pd[8], fma(Y, Pd[9], fma(Z, Pd[10], Pd[11])));
    const bool behind = z0 <= 0.0;
    const double z = (z0 == 0.0) ? 1.0 : z0;
    double rz = static_cast<double>(__builtin_amdgcn_rcpf(static_cast<float>(z)));
    rz = fma(rz, fma(-z, rz, 1.0), rz);
    rz = fma(rz, fma(-z, rz, 1.0), rz);
    const double ixd = u * rz, iyd = v * rz;
    const double x0d = floor(ixd), y0d = floor(iyd);
    const float wx1 = static_cast<float>(ixd - x0d), wy1 = static_cast<float>(iyd - y0d);
    const float wx0 = static_cast<float>((x0d + 1.0) - ixd), wy0 = static_cast<float>((y0d + 1.0) - iyd);
    const float x0f = static_cast<float>(x0d), y0f = static_cast<float>(y0d);
    // clamp before the conversion so that far-away or non-finite positions become "outside"
    const int x0 = static_cast<int>(fminf(fmaxf(x0f, -2.0f), static_cast<float>(W)));
    const int y0 = static_cast<int>(fminf(fmaxf(y0f, -2.0f), static_cast<float>(H)));
    const bool finite = (x0f - x0f == 0.0f) && (y0f - y0f == 0.0f) && !behind;
    const bool vx0 = finite && x0 >= 0 && x0 <= W - 1, vx1 = finite && x0 + 1 >= 0 && x0 + 1 <= W - 1;
    const bool vy0 = finite && y0 >= 0 && y0 <= H - 1, vy1 = finite && y0 + 1 >= 0 && y0 + 1 <= H - 1;
    const int xb = min(max(x0, 0), W - 2), yb = min(max(y0, 0), H - 2);
    // column xb holds tap x0 (if x0 == xb) or tap x0+1 (if x0 + 1 == xb, i.e. x0 == -1); likewise xb+1
    const float cxa = (vx0 && x0 == xb) ? wx0 : ((vx1 && x0 + 1 == xb) ? wx1 : 0.0f);
    const float cxb = (vx1 && x0 == xb) ? wx1 : ((vx0 && x0 == xb + 1) ? wx0 : 0.0f);
    const float cya = ((vy0 && y0 == yb) ? wy0 : ((vy1 && y0 + 1 == yb) ? wy1 : 0.0f)) * wscale;
    const float cyb = ((vy1 && y0 == yb) ? wy1 : ((vy0 && y0 == yb + 1) ? wy0 : 0.0f)) * wscale;
    Tap t;
    t.w00 = cxa * cya;
    t.w01 = cxb * cya;
    t.w10 = cxa * cyb;
    t.w11 = cxb * cyb;
    t.pixel = static_cast<unsigned>(yb * W + xb);
    return t;
}

// Camera row r of a float32 3x4 matrix as float64, scaled for project_tap.
__device__ __forceinline__ double camera_entry(const float* __restrict__ P, int i, double sx, double sy) {
    const int r = i >> 2;
    return static_cast<double>(P[i]) * (r == 0 ? sx : (r == 1 ? sy : 1.0));
}

// Aggregation over views of four channels at a time, for any number of views.  For the softmax the sampled
// values arrive multiplied by log2 e (see project_tap) and the result is multiplied by ln 2.
template <int AGG>
struct Aggregator {
    float a[4], num[4], den[4];
    __device__ __forceinline__ void init() {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            a[k] = (AGG == MVX_AGG_MAX || AGG == MVX_AGG_SOFTMAX) ? -INFINITY : 0.0f;
            num[k] = 0.0f;
            den[k] = 0.0f;
        }
    }
    __device__ __forceinline__ void add(const float (&val)[4], const float (&conf)[4]) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (AGG == MVX_AGG_SUM) {
                a[k] += val[k];
            } else if (AGG == MVX_AGG_CONF) {
                a[k] = fmaf(val[k], conf[k], a[k]);
            } else if (AGG == MVX_AGG_MAX) {
                a[k] = fmaxf(a[k], val[k]);
            } else {
                // online softmax over views with one exp per value: a = running max,
                // num = sum val*2^(val - a), den = sum 2^(val - a)
                const float d = val[k] - a[k];
                const float r = __builtin_amdgcn_exp2f(-fabsf(d));
                const bool up = d > 0.0f;
                const float keep = up ? r : 1.0f;  // rescale of what is already summed
                const float e = up ? 1.0f : r;     // weight of the new value
                den[k] = fmaf(den[k], keep, e);
                num[k] = fmaf(num[k], keep, e * val[k]);
                a[k] = up ? val[k] : a[k];
            }
        }
    }
    __device__ __forceinline__ float result(int k) const {
        return AGG == MVX_AGG_SOFTMAX ? num[k] * __builtin_amdgcn_rcpf(den[k]) * kLn2 : a[k];
    }
};

typedef float float2v __attribute__((ext_vector_type(2)));

// The 2 x 2 taps of four channels.  `img` is uniform over the wave, so the four loads share two scalar bases
// (the two rows) and one 32-bit lane offset (host code keeps an image below 4 GB).
struct Taps4 {
    float4 f00, f01, f10, f11;
};
__device__ __forceinline__ Taps4 load_taps(const float* __restrict__ img, unsigned pixel, int W, int Cp, int q) {
    const char* row0 = reinterpret_cast<const char*>(img);
    const char* row1 = row0 + static_cast<size_t>(W) * Cp * sizeof(float);
    const unsigned off = (pixel * static_cast<unsigned>(Cp) + static_cast<unsigned>(q) * 4u) * 4u;
    Taps4 t;
    t.f00 = *reinterpret_cast<const float4*>(row0 + off);
    t.f01 = *reinterpret_cast<const float4*>(row0 + off + Cp * sizeof(float));
    t.f10 = *reinterpret_cast<const float4*>(row1 + off);
    t.f11 = *reinterpret_cast<const float4*>(row1 + off + Cp * sizeof(float));
    return t;
}
// fma(f11, w11, fma(f10, w10, fma(f01, w01, f00 * w00))) per channel, two channels per packed instruction
// (each half rounds exactly as the scalar instruction does).
__device__ __forceinline__ void interpolate(const Taps4& t, const float4& w, float (&val)[4]) {
    const float2v w00 = {w.x, w.x}, w01 = {w.y, w.y}, w10 = {w.z, w.z}, w11 = {w.w, w.w};
    const float2v lo = __builtin_elementwise_fma(
        float2v{t.f11.x, t.f11.y}, w11,
        __builtin_elementwise_fma(float2v{t.f10.x, t.f10.y}, w10,
                                  __builtin_elementwise_fma(float2v{t.f01.x, t.f01.y}, w01, float2v{t.f00.x, t.f00.y} * w00)));
    const float2v hi = __builtin_elementwise_fma(
        float2v{t.f11.z, t.f11.w}, w11,
        __builtin_elementwise_fma(float2v{t.f10.z, t.f10.w}, w10,
                                  __builtin_elementwise_fma(float2v{t.f01.z, t.f01.w}, w01, float2v{t.f00.z, t.f00.w} * w00)));
    val[0] = lo.x;
    val[1] = lo.y;
    val[2] = hi.x;
    val[3] = hi.y;
}
__device__ __forceinline__ void sample4(const float* __restrict__ img, const Tap& t, int W, int Cp, int q,
                                        float (&val)[4]) {
    const Taps4 f = load_taps(img, t.pixel, W, Cp, q);
    interpolate(f, make_float4(t.w00, t.w01, t.w10, t.w11), val);
}

// [n][C][HW] -> [n][HW][Cp], Cp = C rounded up to 4 (pad = 0).  64-pixel x 32-channel tiles through LDS.
__global__ __launch_bounds__(kThreads) void to_channel_last(const float* __restrict__ src, int C, int Cp, int HW,
                                                             int p_tiles, int c_tiles, float* __restrict__ dst) {
    __shared__ float tile[32][65];
    const int tid = threadIdx.x;
    unsigned id = blockIdx.x;
    const int pt = id % p_tiles;
    id /= p_tiles;
    const int ct = id % c_tiles;
    const size_t n = id / c_tiles;
    const int p0 = pt * 64, c0 = ct * 32;
    {
        const int tx = tid & 63, ty = tid >> 6;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int c = c0 + ty + 4 * i, px = p0 + tx;
            tile[ty + 4 * i][tx] = (c < C && px < HW) ? src[(n * C + c) * HW + px] : 0.0f;
        }
    }
    __syncthreads();
    {
        const int cx = tid & 31, py = tid >> 5;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int px = p0 + py + 8 * i, c = c0 + cx;
            if (px < HW && c < Cp) dst[(n * HW + px) * Cp + c] = tile[cx][py + 8 * i];
        }
    }
}

// One lane per (voxel, channel quad).  feat_cl: channel-last maps of the frames this launch covers.
template <int AGG>
__global__ __launch_bounds__(kThreads) void unproject_generic(const float* __restrict__ feat_cl,
                                                              const float* __restrict__ proj,
                                                              const float* __restrict__ coord,
                                                              const float* __restrict__ vconf, int64_t n_items,
                                                              int V, int C, int Cp, int H, int W, int64_t n_vox,
                                                              double sx, double sy, float* __restrict__ vol) {
    const int64_t idx = static_cast<int64_t>(blockIdx.x) * kThreads + threadIdx.x;
    if (idx >= n_items) return;
    const int Cq = Cp >> 2;
    const int q = static_cast<int>(idx % Cq);
    const int64_t bn = idx / Cq;
    const int64_t b = bn / n_vox, n = bn - b * n_vox;
    const float* cp = coord + (b * n_vox + n) * 3;
    const double X = cp[0], Y = cp[1], Z = cp[2];
    Aggregator<AGG> agg;
    agg.init();
    for (int v = 0; v < V; ++v) {
        double Pd[12];
#pragma unroll
        for (int i = 0; i < 12; ++i) Pd[i] = camera_entry(proj + (b * V + v) * 12, i, sx, sy);
        const Tap t = project_tap(Pd, X, Y, Z, H, W, AGG == MVX_AGG_SOFTMAX ? kLog2e : 1.0f);
        float val[4], conf[4] = {0.0f, 0.0f, 0.0f, 0.0f};
        sample4(feat_cl + (b * V + v) * static_cast<size_t>(H) * W * Cp, t, W, Cp, q, val);
        if (AGG == MVX_AGG_CONF) {
#pragma unroll
            for (int k = 0; k < 4; ++k) conf[k] = (q * 4 + k < C) ? vconf[(b * V + v) * C + q * 4 + k] : 0.0f;
        }
        agg.add(val, conf);
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int c = q * 4 + k;
        if (c < C) vol[(b * C + c) * n_vox + n] = agg.result(k);
    }
}

// C = 32, TILE voxels per workgroup (n_vox % TILE == 0), V <= kTileMaxViews.
constexpr int kTile = 128;
constexpr int kTileMaxViews = 16;

// FOUR: exactly four views, unrolled, with the view softmax evaluated exactly (max, then exp2); otherwise any
// V <= kTileMaxViews with the online form.
template <int AGG, bool FOUR>
__global__ __launch_bounds__(kThreads) void unproject_tile32(const float* __restrict__ feat_cl,
                                                             const float* __restrict__ proj,
                                                             const float* __restrict__ coord,
                                                             const float* __restrict__ vconf, int V, int H, int W,
                                                             int64_t n_vox, int tiles_per_frame, double sx, double sy,
                                                             float* __restrict__ vol) {
    constexpr int C = 32;
    extern __shared__ __align__(16) unsigned char smem[];
    float* out_t = reinterpret_cast<float*>(smem);                                       // [32][kTile], swizzled
    float4* wts = reinterpret_cast<float4*>(smem + C * kTile * sizeof(float));           // [V][kTile]
    unsigned* pix = reinterpret_cast<unsigned*>(wts + static_cast<size_t>(V) * kTile);   // [V][kTile]
    double* cam = reinterpret_cast<double*>(pix + static_cast<size_t>(V) * kTile);       // [V][12]

    const int tid = threadIdx.x;
    const unsigned work = xcd_contiguous(blockIdx.x, gridDim.x);
    const int64_t b = work / tiles_per_frame;
    const int64_t n0 = static_cast<int64_t>(work % tiles_per_frame) * kTile;
    const size_t img_stride = static_cast<size_t>(H) * W * C;

    // phase A: cameras to float64, then the taps of every (view, voxel) of the tile
    if (tid < V * 12) cam[tid] = camera_entry(proj + b * V * 12 + (tid / 12) * 12, tid % 12, sx, sy);
    __syncthreads();
    {
        const int t = tid & (kTile - 1);
        const float* cp = coord + (b * n_vox + n0 + t) * 3;
        const double X = cp[0], Y = cp[1], Z = cp[2];
        for (int v = tid / kTile; v < V; v += kThreads / kTile) {
            const Tap tap = project_tap(cam + v * 12, X, Y, Z, H, W, AGG == MVX_AGG_SOFTMAX ? kLog2e : 1.0f);
            wts[v * kTile + t] = make_float4(tap.w00, tap.w01, tap.w10, tap.w11);
            pix[v * kTile + t] = tap.pixel;
        }
    }
    __syncthreads();

    // phase B: 8 lanes per voxel, 4 channels per lane
    const int q = tid & 7;
    const float* fb = feat_cl + b * V * img_stride;
    if (FOUR) {
        float cc[4][4];
        if (AGG == MVX_AGG_CONF) {
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                const float4 c4 = *reinterpret_cast<const float4*>(vconf + (b * 4 + v) * C + q * 4);
                cc[v][0] = c4.x;
                cc[v][1] = c4.y;
                cc[v][2] = c4.z;
                cc[v][3] = c4.w;
            }
        }
        for (int step = 0; step < kTile / 32; ++step) {
            const int t = step * 32 + (tid >> 3);
            // all sixteen loads of the step are issued before the first value is used
            float4 w4[4];
            Taps4 f[4];
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                w4[v] = wts[v * kTile + t];
                f[v] = load_taps(fb + v * img_stride, pix[v * kTile + t], W, C, q);
            }
            __builtin_amdgcn_sched_barrier(0);
            float val[4][4];
#pragma unroll
            for (int v = 0; v < 4; ++v) interpolate(f[v], w4[v], val[v]);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                float r;
                if (AGG == MVX_AGG_SUM) {
                    r = ((val[0][k] + val[1][k]) + val[2][k]) + val[3][k];
                } else if (AGG == MVX_AGG_CONF) {
                    r = fmaf(val[3][k], cc[3][k], fmaf(val[2][k], cc[2][k], fmaf(val[1][k], cc[1][k], val[0][k] * cc[0][k])));
                } else if (AGG == MVX_AGG_MAX) {
                    r = fmaxf(fmaxf(val[0][k], val[1][k]), fmaxf(val[2][k], val[3][k]));
                } else {
                    const float m = fmaxf(fmaxf(val[0][k], val[1][k]), fmaxf(val[2][k], val[3][k]));
                    const float e0 = __builtin_amdgcn_exp2f(val[0][k] - m), e1 = __builtin_amdgcn_exp2f(val[1][k] - m);
                    const float e2 = __builtin_amdgcn_exp2f(val[2][k] - m), e3 = __builtin_amdgcn_exp2f(val[3][k] - m);
                    const float den = (e0 + e1) + (e2 + e3);
                    const float num = fmaf(e3, val[3][k], fmaf(e2, val[2][k], fmaf(e1, val[1][k], e0 * val[0][k])));
                    r = num * __builtin_amdgcn_rcpf(den) * kLn2;
                }
                out_t[(q * 4 + k) * kTile + ((((t >> 2) ^ q) << 2) | (t & 3))] = r;
            }
        }
    } else {
#pragma unroll 2
        for (int step = 0; step < kTile / 32; ++step) {
            const int t = step * 32 + (tid >> 3);
            Aggregator<AGG> agg;
            agg.init();
            for (int v = 0; v < V; ++v) {
                const float4 w4 = wts[v * kTile + t];
                Tap tap;
                tap.w00 = w4.x;
                tap.w01 = w4.y;
                tap.w10 = w4.z;
                tap.w11 = w4.w;
                tap.pixel = pix[v * kTile + t];
                float val[4], conf[4] = {0.0f, 0.0f, 0.0f, 0.0f};
                sample4(fb + v * img_stride, tap, W, C, q, val);
                if (AGG == MVX_AGG_CONF) {
                    const float4 c4 = *reinterpret_cast<const float4*>(vconf + (b * V + v) * C + q * 4);
                    conf[0] = c4.x;
                    conf[1] = c4.y;
                    conf[2] = c4.z;
                    conf[3] = c4.w;
                }
                agg.add(val, conf);
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) out_t[(q * 4 + k) * kTile + ((((t >> 2) ^ q) << 2) | (t & 3))] = agg.result(k);
        }
    }
    __syncthreads();

    // phase C: 16 bytes per lane, 512 contiguous bytes per channel
#pragma unroll
    for (int i = 0; i < (C * kTile / 4) / kThreads; ++i) {
        const int flat = i * kThreads + tid;
        const int c = flat / (kTile / 4), grp = flat % (kTile / 4);
        const float4 r = *reinterpret_cast<const float4*>(out_t + c * kTile + ((grp ^ ((c >> 2) & 7)) << 2));
        *reinterpret_cast<float4*>(vol + (b * C + c) * n_vox + n0 + grp * 4) = r;
    }
}

}  // namespace

}  // namespace mvx

extern "C" size_t mvx_unproject_workspace_bytes(int64_t n_frames, int n_views, int n_channels, int h, int w) {
    if (n_frames <= 0 || n_views <= 0 || n_channels <= 0 || h <= 0 || w <= 0) return 0;
    const int64_t frames = n_frames < mvx::kChunkFrames ? n_frames : mvx::kChunkFrames;
    const int64_t cp = (n_channels + 3) / 4 * 4;
    return static_cast<size_t>(frames) * n_views * h * w * cp * sizeof(float);
}

namespace mvx {
namespace {
// channels_last: features are [n_frames][n_views][h][w][n_channels] already and no workspace is used.
int unproject_any_layout(const char* name, bool channels_last, const float* features, const float* proj,
                         const float* coords, const float* vol_conf, int64_t n_frames, int n_views, int n_channels, int h,
                         int w, int vx, int vy, int vz, int aggregation, float* volumes, void* workspace,
                         size_t workspace_bytes, void* stream);
}  // namespace
}  // namespace mvx

extern "C" int mvx_unproject(const float* features, const float* proj, const float* coords, const float* vol_conf,
                             int64_t n_frames, int n_views, int n_channels, int h, int w, int vx, int vy, int vz,
                             int aggregation, float* volumes, void* workspace, size_t workspace_bytes, void* stream) {
    return mvx::unproject_any_layout("mvx_unproject", false, features, proj, coords, vol_conf, n_frames, n_views,
                                     n_channels, h, w, vx, vy, vz, aggregation, volumes, workspace, workspace_bytes,
                                     stream);
}

extern "C" int mvx_unproject_channels_last(const float* features, const float* proj, const float* coords,
                                           const float* vol_conf, int64_t n_frames, int n_views, int n_channels, int h,
                                           int w, int vx, int vy, int vz, int aggregation, float* volumes,
                                           void* stream) {
    return mvx::unproject_any_layout("mvx_unproject_channels_last", true, features, proj, coords, vol_conf, n_frames,
                                     n_views, n_channels, h, w, vx, vy, vz, aggregation, volumes, nullptr, 0, stream);
}

namespace mvx {
namespace {
int unproject_any_layout(const char* name, bool channels_last, const float* features, const float* proj,
                         const float* coords, const float* vol_conf, int64_t n_frames, int n_views, int n_channels, int h,
                         int w, int vx, int vy, int vz, int aggregation, float* volumes, void* workspace,
                         size_t workspace_bytes, void* stream) {
    MVX_REQUIRE(n_frames >= 0 && n_views > 0 && n_channels > 0 && h >= 2 && w >= 2 && vx >= 0 && vy >= 0 && vz >= 0,
                MVX_ERR_INVALID_ARGUMENT, "%s: bad shape n_frames=%lld n_views=%d n_channels=%d h=%d w=%d grid=%dx%dx%d",
                name, (long long)n_frames, n_views, n_channels, h, w, vx, vy, vz);
    const int64_t n_voxels = static_cast<int64_t>(vx) * vy * vz;
    MVX_REQUIRE(aggregation >= MVX_AGG_SUM && aggregation <= MVX_AGG_SOFTMAX, MVX_ERR_INVALID_ARGUMENT,
                "%s: unknown aggregation %d", name, aggregation);
    if (n_frames == 0 || n_voxels == 0) return MVX_OK;
    MVX_REQUIRE(features != nullptr && proj != nullptr && coords != nullptr && volumes != nullptr &&
                    (channels_last || workspace != nullptr),
                MVX_ERR_INVALID_ARGUMENT, "%s: null pointer argument", name);
    MVX_REQUIRE(aggregation != MVX_AGG_CONF || vol_conf != nullptr, MVX_ERR_INVALID_ARGUMENT,
                "%s: confidence aggregation needs vol_conf", name);
    if (channels_last) {
        MVX_REQUIRE(n_channels % 4 == 0 && aligned16(features), MVX_ERR_INVALID_ARGUMENT,
                    "%s: channel-last features need n_channels %% 4 == 0 (got %d) and a 16-byte aligned pointer", name,
                    n_channels);
    } else {
        MVX_REQUIRE(workspace_bytes >= mvx_unproject_workspace_bytes(n_frames, n_views, n_channels, h, w),
                    MVX_ERR_WORKSPACE, "%s: workspace of %zu bytes is too small", name, workspace_bytes);
        MVX_REQUIRE(aligned16(workspace), MVX_ERR_INVALID_ARGUMENT, "%s: workspace must be 16-byte aligned", name);
    }
    const int C = n_channels, Cp = (C + 3) / 4 * 4, V = n_views, HW = h * w;
    MVX_REQUIRE(static_cast<int64_t>(h) * w * Cp < (1LL << 30), MVX_ERR_UNSUPPORTED,
                "%s: one feature map exceeds 2^30 elements (4 GB)", name);
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int p_tiles = (HW + 63) / 64, c_tiles = (Cp + 31) / 32;
    const bool tiled = (C == 32) && (V <= kTileMaxViews) && (n_voxels % kTile == 0) && aligned16(volumes) &&
                       (aggregation != MVX_AGG_CONF || aligned16(vol_conf));
    const size_t smem = static_cast<size_t>(32) * kTile * 4 + static_cast<size_t>(V) * kTile * 20 + static_cast<size_t>(V) * 12 * 8;
    const double sx = static_cast<double>(w - 1) / static_cast<double>(h);  // ix = x / H * (W - 1)
    const double sy = static_cast<double>(h - 1) / static_cast<double>(w);  // iy = y / W * (H - 1)

    // With the layout copy, rounds of a few frames keep the copied maps in L2 for the sampling that follows;
    // maps sampled in place have no such coupling and go in as few launches as the grid limits allow.
    const int64_t chunk = channels_last ? 1024 : kChunkFrames;
    for (int64_t f0 = 0; f0 < n_frames; f0 += chunk) {
        const int64_t nf = (n_frames - f0 < chunk) ? n_frames - f0 : chunk;
        const float* feat_cl = features + f0 * V * C * HW;
        if (!channels_last) {
            const int64_t t_blocks = nf * V * static_cast<int64_t>(p_tiles) * c_tiles;
            MVX_REQUIRE(t_blocks <= 0x7fffffffLL, MVX_ERR_UNSUPPORTED, "%s: feature maps too large", name);
            hipLaunchKernelGGL(to_channel_last, dim3(static_cast<unsigned>(t_blocks)), dim3(kThreads), 0, st,
                               features + f0 * V * C * HW, C, Cp, HW, p_tiles, c_tiles, static_cast<float*>(workspace));
            MVX_CHECK_LAUNCH("mvx_unproject (channel-last copy)");
            feat_cl = static_cast<const float*>(workspace);
        }
        const float* pj = proj + f0 * V * 12;
        const float* cd = coords + f0 * n_voxels * 3;
        const float* vc = vol_conf != nullptr ? vol_conf + f0 * V * C : nullptr;
        float* vo = volumes + f0 * C * n_voxels;
        if (tiled) {
            const int64_t tiles = n_voxels / kTile;
            MVX_REQUIRE(nf * tiles <= 0x7fffffffLL, MVX_ERR_UNSUPPORTED, "%s: voxel grid too large", name);
            const dim3 grid(static_cast<unsigned>(nf * tiles));
#define MVX_LAUNCH_TILE(AGG)                                                                                        \
    if (V == 4)                                                                                                     \
        hipLaunchKernelGGL((unproject_tile32<AGG, true>), grid, dim3(kThreads), smem, st, feat_cl, pj, cd, vc, V, h, w, \
                           n_voxels, static_cast<int>(tiles), sx, sy, vo);                                          \
    else                                                                                                            \
        hipLaunchKernelGGL((unproject_tile32<AGG, false>), grid, dim3(kThreads), smem, st, feat_cl, pj, cd, vc, V, h, \
                           w, n_voxels, static_cast<int>(tiles), sx, sy, vo)
            switch (aggregation) {
                case MVX_AGG_SUM: MVX_LAUNCH_TILE(MVX_AGG_SUM); break;
                case MVX_AGG_CONF: MVX_LAUNCH_TILE(MVX_AGG_CONF); break;
                case MVX_AGG_MAX: MVX_LAUNCH_TILE(MVX_AGG_MAX); break;
                default: MVX_LAUNCH_TILE(MVX_AGG_SOFTMAX); break;
            }
#undef MVX_LAUNCH_TILE
        } else {
            const int64_t n_items = nf * n_voxels * (Cp / 4);
            const int64_t blocks = (n_items + kThreads - 1) / kThreads;
            MVX_REQUIRE(blocks <= 0x7fffffffLL, MVX_ERR_UNSUPPORTED, "%s: voxel grid too large", name);
            const dim3 grid(static_cast<unsigned>(blocks));
#define MVX_LAUNCH_GENERIC(AGG)                                                                                  \
    hipLaunchKernelGGL(unproject_generic<AGG>, grid, dim3(kThreads), 0, st, feat_cl, pj, cd, vc, n_items, V, C, Cp, h, \
                       w, n_voxels, sx, sy, vo)
            switch (aggregation) {
                case MVX_AGG_SUM: MVX_LAUNCH_GENERIC(MVX_AGG_SUM); break;
                case MVX_AGG_CONF: MVX_LAUNCH_GENERIC(MVX_AGG_CONF); break;
                case MVX_AGG_MAX: MVX_LAUNCH_GENERIC(MVX_AGG_MAX); break;
                default: MVX_LAUNCH_GENERIC(MVX_AGG_SOFTMAX); break;
            }
#undef MVX_LAUNCH_GENERIC
        }
        MVX_CHECK_LAUNCH("mvx_unproject (sampling)");
    }
    return MVX_OK;
}
}  // namespace
}  // namespace mvx
