// Unprojection of per-view feature maps into a voxel grid with view aggregation.
//
// Maps arrive either channel-last already ([view][h][w][channel]) or as [view][channel][h][w]; the latter are first
// copied to channel-last (channels padded to a multiple of 4) in rounds of kChunkFrames = 32 frames, so that the C
// channels of one bilinear tap are one contiguous C*4-byte run (a full 128-byte line for C = 32).  Each round's
// copy is sampled straight away by the kernels below.
//
// Brick kernel (unproject_brick32; the default for 4 views of 32 channels and grid sides that are multiples of 8):
// a workgroup owns 8 x 8 x 8 voxels, stages the <= 16 x 16-pixel patches of the maps that its taps fall into in LDS
// (global -> LDS copies, two views at a time) and reads the taps from there.  See the comment above the kernel.
//
// Tile kernel (unproject_tile32: C = 32, up to 16 views, any grid whose voxel count is a multiple of 128): a
// workgroup owns 128 consecutive voxels of one frame.
//   phase A  one thread per voxel projects it into every view and leaves the four bilinear weights (zero for
//            taps outside the map and for voxels behind the camera) and the tap address in LDS;
//   phase B  8 lanes per voxel, 4 channels per lane: four 16-byte loads per view fetch the 2x2 taps (each
//            wave-instruction consumes whole 128-byte lines), the view aggregation runs in registers
//            (sum / confidence-weighted / max / softmax over views), results go to an LDS tile
//            [channel][voxel] (XOR-swizzled so both sides are conflict-free);
//   phase C  the tile is stored with 16 bytes per lane: 512 contiguous bytes per channel.
// It is bound by the vector-memory return path (512 B of taps per voxel and view through the L1 whether or not the
// lines hit), which is what the brick kernel's LDS patches avoid.
//
// Generic kernel (any C, V, grid): one lane per (voxel, 4 channels), no LDS, direct stores.
//
// The per-view volumes of the reference (V x C x N floats per frame) are never materialised.  Workgroup ids are
// remapped so that each XCD works through consecutive bricks / tiles of the same frame and its L2 holds just that
// frame's feature maps.
//
// HBM traffic per frame: V*C*h*w*4 (features) + N*12 (coordinates, unless the grid parameters are given) + C*N*4
// (volume written).
#include <limits.h>
#include <stdlib.h>

#include "grid_coords.h"
#include "unproject_taps.h"

namespace mvx {

namespace {

constexpr int kThreads = 256;
constexpr int kChunkFrames = 32;  // frames per channel-last copy + sampling round

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
// the same, results left as two packed pairs (channels 0,1 and 2,3)
__device__ __forceinline__ void interpolate_x2(const Taps4& t, const float4& w, float2v& lo, float2v& hi) {
    const float2v w00 = {w.x, w.x}, w01 = {w.y, w.y}, w10 = {w.z, w.z}, w11 = {w.w, w.w};
    lo = __builtin_elementwise_fma(
        float2v{t.f11.x, t.f11.y}, w11,
        __builtin_elementwise_fma(float2v{t.f10.x, t.f10.y}, w10,
                                  __builtin_elementwise_fma(float2v{t.f01.x, t.f01.y}, w01, float2v{t.f00.x, t.f00.y} * w00)));
    hi = __builtin_elementwise_fma(
        float2v{t.f11.z, t.f11.w}, w11,
        __builtin_elementwise_fma(float2v{t.f10.z, t.f10.w}, w10,
                                  __builtin_elementwise_fma(float2v{t.f01.z, t.f01.w}, w01, float2v{t.f00.z, t.f00.w} * w00)));
}
__device__ __forceinline__ void sample4(const float* __restrict__ img, const Tap& t, int W, int Cp, int q,
                                        float (&val)[4]) {
    const Taps4 f = load_taps(img, t.pixel, W, Cp, q);
    interpolate(f, make_float4(t.w00, t.w01, t.w10, t.w11), val);
}

// Aggregation of exactly four views of one channel (the view softmax evaluated exactly: max, then exp2; sampled
// values arrive multiplied by log2 e).  One definition, so that every kernel rounds alike.
template <int AGG>
__device__ __forceinline__ float combine_four(float v0, float v1, float v2, float v3, float c0, float c1, float c2,
                                              float c3) {
    if (AGG == MVX_AGG_SUM) return ((v0 + v1) + v2) + v3;
    if (AGG == MVX_AGG_CONF) return fmaf(v3, c3, fmaf(v2, c2, fmaf(v1, c1, v0 * c0)));
    if (AGG == MVX_AGG_MAX) return fmaxf(fmaxf(v0, v1), fmaxf(v2, v3));
    const float m = fmaxf(fmaxf(v0, v1), fmaxf(v2, v3));
    const float e0 = __builtin_amdgcn_exp2f(v0 - m), e1 = __builtin_amdgcn_exp2f(v1 - m);
    const float e2 = __builtin_amdgcn_exp2f(v2 - m), e3 = __builtin_amdgcn_exp2f(v3 - m);
    const float den = (e0 + e1) + (e2 + e3);
    const float num = fmaf(e3, v3, fmaf(e2, v2, fmaf(e1, v1, e0 * v0)));
    return num * __builtin_amdgcn_rcpf(den) * kLn2;
}

// The same for two channels at a time in packed float32 instructions (v_pk_add / v_pk_mul / v_pk_fma: two
// IEEE operations per instruction, each half rounding exactly as the scalar instruction does), which halves the
// instruction count of everything but the maxima, the exponentials and the reciprocal.
template <int AGG>
__device__ __forceinline__ float2v combine_four_x2(float2v v0, float2v v1, float2v v2, float2v v3, float2v c0,
                                                   float2v c1, float2v c2, float2v c3) {
    if (AGG == MVX_AGG_SUM) return ((v0 + v1) + v2) + v3;
    if (AGG == MVX_AGG_CONF)
        return __builtin_elementwise_fma(v3, c3, __builtin_elementwise_fma(v2, c2, __builtin_elementwise_fma(v1, c1, v0 * c0)));
    float2v m;
    m.x = fmaxf(fmaxf(v0.x, v1.x), fmaxf(v2.x, v3.x));
    m.y = fmaxf(fmaxf(v0.y, v1.y), fmaxf(v2.y, v3.y));
    if (AGG == MVX_AGG_MAX) return m;
    const float2v d0 = v0 - m, d1 = v1 - m, d2 = v2 - m, d3 = v3 - m;
    const float2v e0 = {__builtin_amdgcn_exp2f(d0.x), __builtin_amdgcn_exp2f(d0.y)};
    const float2v e1 = {__builtin_amdgcn_exp2f(d1.x), __builtin_amdgcn_exp2f(d1.y)};
    const float2v e2 = {__builtin_amdgcn_exp2f(d2.x), __builtin_amdgcn_exp2f(d2.y)};
    const float2v e3 = {__builtin_amdgcn_exp2f(d3.x), __builtin_amdgcn_exp2f(d3.y)};
    const float2v den = (e0 + e1) + (e2 + e3);
    const float2v num = __builtin_elementwise_fma(e3, v3, __builtin_elementwise_fma(e2, v2, __builtin_elementwise_fma(e1, v1, e0 * v0)));
    const float2v rden = {__builtin_amdgcn_rcpf(den.x), __builtin_amdgcn_rcpf(den.y)};
    const float2v ln2 = {kLn2, kLn2};
    return num * rden * ln2;
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

// The same for C % 32 == 0 and HW % 64 == 0 (the model's 32 x 96 x 96 maps) with 16-byte accesses on both sides: a
// lane reads four consecutive pixels of one channel and writes four consecutive channels of one pixel, so a wave
// instruction reads 4 runs of 256 bytes and writes 2 KB in one piece.
__global__ __launch_bounds__(kThreads) void to_channel_last_x4(const float* __restrict__ src, int C, int HW, int p_tiles,
                                                                int c_tiles, float* __restrict__ dst) {
    __shared__ float tile[32][65];
    const int tid = threadIdx.x;
    unsigned id = blockIdx.x;
    const int pt = id % p_tiles;
    id /= p_tiles;
    const int ct = id % c_tiles;
    const size_t n = id / c_tiles;
    const int p0 = pt * 64, c0 = ct * 32;
    typedef float v4 __attribute__((ext_vector_type(4)));
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int c = (tid >> 4) + 16 * i, px = (tid & 15) * 4;  // 16 lanes x 16 bytes per channel row
        const v4 r = __builtin_nontemporal_load(reinterpret_cast<const v4*>(src + (n * C + c0 + c) * HW + p0 + px));
        tile[c][px + 0] = r.x;
        tile[c][px + 1] = r.y;
        tile[c][px + 2] = r.z;
        tile[c][px + 3] = r.w;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int px = (tid >> 3) + 32 * i, c = (tid & 7) * 4;  // 8 lanes x 16 bytes per pixel
        const v4 r = {tile[c + 0][px], tile[c + 1][px], tile[c + 2][px], tile[c + 3][px]};
        *reinterpret_cast<v4*>(dst + (n * HW + p0 + px) * C + c0 + c) = r;
    }
}

// One lane per (voxel, channel quad).  feat_cl: channel-last maps of the frames this launch covers.
template <int AGG>
__global__ __launch_bounds__(kThreads) void unproject_generic(const float* __restrict__ feat_cl,
                                                              const float* __restrict__ proj,
                                                              VoxelSource coord,
                                                              const float* __restrict__ vconf, int64_t n_items,
                                                              int V, int C, int Cp, int H, int W, int64_t n_vox,
                                                              double sx, double sy, float* __restrict__ vol) {
    const int64_t idx = static_cast<int64_t>(blockIdx.x) * kThreads + threadIdx.x;
    if (idx >= n_items) return;
    const int Cq = Cp >> 2;
    const int q = static_cast<int>(idx % Cq);
    const int64_t bn = idx / Cq;
    const int64_t b = bn / n_vox, n = bn - b * n_vox;
    double X, Y, Z;
    voxel_position(coord, b, n_vox, n, X, Y, Z);
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
constexpr int kTileViewChunk = 4;  // views whose tap tables are in LDS at a time (more would cost residency: 16 views
                                   // of tables are 40 KB, two workgroups per CU)

// FOUR: exactly four views, unrolled, with the view softmax evaluated exactly (max, then exp2); otherwise any
// V <= kTileMaxViews with the online form.
template <int AGG, bool FOUR>
__global__ __launch_bounds__(kThreads) void unproject_tile32(const float* __restrict__ feat_cl,
                                                             const float* __restrict__ proj,
                                                             VoxelSource coord,
                                                             const float* __restrict__ vconf, int V, int H, int W,
                                                             int64_t n_vox, int tiles_per_frame, double sx, double sy,
                                                             float* __restrict__ vol) {
    constexpr int C = 32;
    extern __shared__ __align__(16) unsigned char smem[];
    float* out_t = reinterpret_cast<float*>(smem);                                       // [32][kTile], swizzled
    const int Vc = V < kTileViewChunk ? V : kTileViewChunk;                               // views per chunk of tables
    float4* wts = reinterpret_cast<float4*>(smem + C * kTile * sizeof(float));           // [Vc][kTile]
    unsigned* pix = reinterpret_cast<unsigned*>(wts + static_cast<size_t>(Vc) * kTile);  // [Vc][kTile]
    double* cam = reinterpret_cast<double*>(pix + static_cast<size_t>(Vc) * kTile);      // [V][12]

    const int tid = threadIdx.x;
    const unsigned work = xcd_contiguous(blockIdx.x, gridDim.x);
    const int64_t b = work / tiles_per_frame;
    const int64_t n0 = static_cast<int64_t>(work % tiles_per_frame) * kTile;
    const size_t img_stride = static_cast<size_t>(H) * W * C;

    // phase A: cameras to float64, then the taps of every (view, voxel) of the tile, up to kTileViewChunk views at
    // a time
    if (tid < V * 12) cam[tid] = camera_entry(proj + b * V * 12 + (tid / 12) * 12, tid % 12, sx, sy);
    __syncthreads();
    double X, Y, Z;
    voxel_position(coord, b, n_vox, n0 + (tid & (kTile - 1)), X, Y, Z);
    auto project_chunk = [&](int v0) {
        const int t = tid & (kTile - 1);
        const int v_end = v0 + Vc < V ? v0 + Vc : V;
        for (int v = v0 + tid / kTile; v < v_end; v += kThreads / kTile) {
            const Tap tap = project_tap(cam + v * 12, X, Y, Z, H, W, AGG == MVX_AGG_SOFTMAX ? kLog2e : 1.0f);
            wts[(v - v0) * kTile + t] = make_float4(tap.w00, tap.w01, tap.w10, tap.w11);
            pix[(v - v0) * kTile + t] = tap.pixel;
        }
    };
    project_chunk(0);
    __syncthreads();

    // phase B: 8 lanes per voxel, 4 channels per lane
    const int q = tid & 7;
    const float* fb = feat_cl + b * V * img_stride;
    if (FOUR) {
        float cc[4][4] = {};
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
                const float r = combine_four<AGG>(val[0][k], val[1][k], val[2][k], val[3][k], cc[0][k], cc[1][k],
                                                  cc[2][k], cc[3][k]);
                out_t[(q * 4 + k) * kTile + ((((t >> 2) ^ q) << 2) | (t & 3))] = r;
            }
        }
    } else if (V <= kTileViewChunk) {
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
    } else {
        // more views than one chunk of tables holds: the aggregates of a lane's four voxels stay in registers from
        // one chunk to the next
        Aggregator<AGG> agg[kTile / 32];
#pragma unroll
        for (int step = 0; step < kTile / 32; ++step) agg[step].init();
        for (int v0 = 0; v0 < V; v0 += kTileViewChunk) {
            if (v0 > 0) {
                __syncthreads();  // every lane has read the tables of the chunk before
                project_chunk(v0);
                __syncthreads();
            }
            const int nv = V - v0 < kTileViewChunk ? V - v0 : kTileViewChunk;
#pragma unroll
            for (int step = 0; step < kTile / 32; ++step) {
                const int t = step * 32 + (tid >> 3);
                for (int vv = 0; vv < nv; ++vv) {
                    const int v = v0 + vv;
                    const float4 w4 = wts[vv * kTile + t];
                    Tap tap;
                    tap.w00 = w4.x;
                    tap.w01 = w4.y;
                    tap.w10 = w4.z;
                    tap.w11 = w4.w;
                    tap.pixel = pix[vv * kTile + t];
                    float val[4], conf[4] = {0.0f, 0.0f, 0.0f, 0.0f};
                    sample4(fb + v * img_stride, tap, W, C, q, val);
                    if (AGG == MVX_AGG_CONF) {
                        const float4 c4 = *reinterpret_cast<const float4*>(vconf + (b * V + v) * C + q * 4);
                        conf[0] = c4.x;
                        conf[1] = c4.y;
                        conf[2] = c4.z;
                        conf[3] = c4.w;
                    }
                    agg[step].add(val, conf);
                }
            }
        }
#pragma unroll
        for (int step = 0; step < kTile / 32; ++step) {
            const int t = step * 32 + (tid >> 3);
#pragma unroll
            for (int k = 0; k < 4; ++k)
                out_t[(q * 4 + k) * kTile + ((((t >> 2) ^ q) << 2) | (t & 3))] = agg[step].result(k);
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


// ---- brick kernel: C = 32, three or four views, grid dimensions multiples of 8 ------------------------------------
// A workgroup of 512 threads owns a brick of 8 x 8 x 8 voxels, one voxel per lane.  Neighbouring voxels sample
// neighbouring pixels, so the taps of a brick lie in a small patch of each view's map (about 11 x 11 pixels for the
// camera set-up of the reference: one voxel is about one heatmap pixel); the patch is copied once into LDS and the
// 2 x 2 x 32-channel taps are read from there: 512 bytes per voxel and view come out of LDS (256 B/clk per CU)
// instead of the vector L1 (46 B/clk for this access shape), and about 13 times fewer bytes cross the L1.
//   1. every lane projects its voxel into the views (weights and tap pixel stay in its registers);
//   2. the bounding box of the taps is taken from the brick's 8 corner voxels (exact for any grid whose voxel
//      positions are affine in the voxel index and in front of the camera: the projection of a convex set is
//      convex); every lane then checks its own taps against the box, so for anything else -- arbitrary coordinate
//      tensors, a corner behind a camera, a patch larger than 16 x 16 pixels -- the whole brick takes the direct
//      path (taps from global memory, the tile kernel's arithmetic);
//   3. one unit per view: the patch of view v (rows of 16 pixels x 128 bytes) lives in LDS buffer v & 1, copied
//      global -> LDS by global_load_lds_dwordx4 (no registers) while the unit before it is sampled; the 16-byte
//      channel quad q of a pixel in patch row py sits at position q ^ (py & 7) of the pixel's 128 bytes, so that the
//      lanes of a wave -- which read the same quad of pixels in different rows -- hit different banks;
//   4. per channel quad a lane reads its taps with ds_read_b128 (the two taps of a row share one address, the
//      second at +128 bytes), interpolates (packed FMA) and adds the view to its running aggregate; in the last unit
//      it finishes the aggregate and stores: a wave's store of one channel covers 8 runs of 8 voxels (32 bytes
//      each); bricks adjacent along z run back to back on one XCD and complete the 128-byte lines in its L2 (with
//      z-neighbours on different XCDs the same stores take seven times as long).
// The lane -> voxel map puts the 16 lanes that one ds_read_b128 cycle serves (lanes whose 4-lane chunks have an even,
// respectively odd, number of set bits in the chunk index) on 2 (y) x 8 (z) voxels, i.e. on a column of pixels.
constexpr int kBrick = 8;
constexpr int kBrickThreads = 512;
constexpr int kPatchW = 16;
constexpr int kPatchRows = 16;
constexpr int kPixelBytes = 128;
constexpr int kPatchBytes = kPatchW * kPatchRows * kPixelBytes;  // 32 KB per view, two views resident

struct BrickShared {
    double cam[4 * 12];
    int corner[4][2][8];  // floor position (x0, y0) of the 8 corner voxels per view; INT_MAX for a corner behind the camera
    int bad;              // some lane's taps lie outside the patches
    int redo;             // the view softmax left the range in which its unshifted form is exact
};

#ifdef MVX_BRICK_STATS  // development builds only (tools/brick_stats.hip)
__device__ unsigned brick_stats[4];  // bricks, bricks on the direct path, bricks whose patches do not fit, max rows
#endif
#ifdef MVX_BRICK_TRACE  // development builds only (tools/brick_lab.hip): s_memtime at the phase boundaries of every wave
constexpr int kTraceStamps = 20;
__device__ unsigned long long* brick_trace;  // [workgroup][wave][kTraceStamps]
#define MVX_TRACE(i) trace_[i] = __builtin_readcyclecounter()
#else
#define MVX_TRACE(i)
#endif


typedef __attribute__((address_space(3))) unsigned char lds_byte;
typedef __attribute__((address_space(1))) const unsigned char global_byte;

// The four taps of channel quad q read from a staged patch.  a0 / a1 = LDS address of the pixel (rx, ry) /
// (rx, ry + 1) of the patch plus the swizzle (row & 7) << 4 of its row.  A pixel starts on a multiple of 128 bytes
// (the dynamic LDS block is declared 128-byte aligned), so the swizzled position of quad q is one XOR away:
// (pixel + (s << 4)) ^ (q << 4) = pixel + ((s ^ q) << 4), and that XOR is all the address arithmetic per row of taps.
__device__ __forceinline__ Taps4 patch_taps(unsigned a0, unsigned a1, int q) {
    typedef float v4 __attribute__((ext_vector_type(4)));
    typedef __attribute__((address_space(3))) const v4 lds_v4;
    const lds_byte* p0 = (const lds_byte*)(static_cast<uintptr_t>(a0 ^ (static_cast<unsigned>(q) << 4)));
    const lds_byte* p1 = (const lds_byte*)(static_cast<uintptr_t>(a1 ^ (static_cast<unsigned>(q) << 4)));
    const v4 r00 = *(lds_v4*)(p0), r01 = *(lds_v4*)(p0 + kPixelBytes);
    const v4 r10 = *(lds_v4*)(p1), r11 = *(lds_v4*)(p1 + kPixelBytes);
    Taps4 t;
    t.f00 = make_float4(r00.x, r00.y, r00.z, r00.w);
    t.f01 = make_float4(r01.x, r01.y, r01.z, r01.w);
    t.f10 = make_float4(r10.x, r10.y, r10.z, r10.w);
    t.f11 = make_float4(r11.x, r11.y, r11.z, r11.w);
    return t;
}

// 4 x 4 transpose across the four lanes of a quad: lane i holds a[0..3] on entry and (lane 0's a[i], lane 1's a[i],
// lane 2's a[i], lane 3's a[i]) on exit.  Two butterfly stages (lane ^ 1, then lane ^ 2), each four selects that take
// one operand through a DPP quad permutation (v_cndmask_b32_dpp: the select and the cross-lane move are one
// instruction, eight in all; written as separate selects and moves the compiler needs about 45, recomputing the
// lane-parity masks for every quad of channels).  Stage 1: an even lane takes its partner's a[0], a[2] as its
// a[1], a[3], an odd lane its partner's a[1], a[3] as its a[0], a[2]; stage 2 likewise on pairs of registers between
// the lower and the upper two lanes.  The lane masks are wave-uniform constants handed in through SGPR pairs.
// (s_nop: a VGPR written by the VALU needs two wait states before a DPP operand reads it, which the assembler does
// not insert inside an asm block.)
__device__ __forceinline__ void quad_transpose(float (&a)[4], int /*lane*/) {
    float n0, n1, n2, n3, m0, m1, m2, m3;
    const unsigned long long odd = 0xaaaaaaaaaaaaaaaaull, even = 0x5555555555555555ull;
    const unsigned long long upper = 0xccccccccccccccccull, lower = 0x3333333333333333ull;
    asm volatile(
        "s_nop 1\n\t"
        "s_mov_b64 vcc, %[odd]\n\t"
        "v_cndmask_b32_dpp %[n1], %[a0], %[a1], vcc quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
        "v_cndmask_b32_dpp %[n3], %[a2], %[a3], vcc quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
        "s_mov_b64 vcc, %[even]\n\t"
        "v_cndmask_b32_dpp %[n0], %[a1], %[a0], vcc quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
        "v_cndmask_b32_dpp %[n2], %[a3], %[a2], vcc quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
        "s_mov_b64 vcc, %[upper]\n\t"
        "s_nop 1\n\t"
        "v_cndmask_b32_dpp %[m2], %[n0], %[n2], vcc quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
        "v_cndmask_b32_dpp %[m3], %[n1], %[n3], vcc quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
        "s_mov_b64 vcc, %[lower]\n\t"
        "v_cndmask_b32_dpp %[m0], %[n2], %[n0], vcc quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
        "v_cndmask_b32_dpp %[m1], %[n3], %[n1], vcc quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1"
        : [n0] "=&v"(n0), [n1] "=&v"(n1), [n2] "=&v"(n2), [n3] "=&v"(n3), [m0] "=&v"(m0), [m1] "=&v"(m1), [m2] "=&v"(m2),
          [m3] "=&v"(m3)
        : [a0] "v"(a[0]), [a1] "v"(a[1]), [a2] "v"(a[2]), [a3] "v"(a[3]), [odd] "s"(odd), [even] "s"(even), [upper] "s"(upper),
          [lower] "s"(lower)
        : "vcc");
    a[0] = m0;
    a[1] = m1;
    a[2] = m2;
    a[3] = m3;
}

template <int AGG, bool GRID, int NV>
__global__ __launch_bounds__(kBrickThreads, 4) void unproject_brick32(const float* __restrict__ feat_cl,
                                                                       const float* __restrict__ proj,
                                                                       VoxelSource coord,
                                                                       const float* __restrict__ vconf, int H, int W,
                                                                       int vx, int vy, int vz, FastDiv bricks_per_frame,
                                                                       FastDiv nbz, FastDiv nby, double sx, double sy,
                                                                       float* __restrict__ vol) {
    static_assert(NV >= 2 && NV <= 4, "two to four views (the library launches three and four)");
    constexpr int C = 32;
    extern __shared__ __align__(128) unsigned char smem[];
    BrickShared* sh = reinterpret_cast<BrickShared*>(smem + 2 * kPatchBytes);

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const unsigned work = xcd_contiguous(blockIdx.x, gridDim.x);
    const unsigned frame = fast_div(work, bricks_per_frame);
    const int64_t b = frame;
    const unsigned brick = work - frame * bricks_per_frame.divisor;
    const unsigned brick_xy = fast_div(brick, nbz);
    const int bz = static_cast<int>(brick - brick_xy * nbz.divisor);
    int bx = static_cast<int>(fast_div(brick_xy, nby));
    int by = static_cast<int>(brick_xy - static_cast<unsigned>(bx) * nby.divisor);
    if ((nby.divisor & 3u) == 0 && ((vx / kBrick) & 1) == 0) {
        // Columns of bricks (all z) in tiles of 2 (x) by 4 (y): the 64 bricks in flight on an XCD are then neighbours
        // in both directions and share more of their patches in its L2 (25 % fewer bytes fetched from memory).
        const unsigned t = brick_xy >> 3, in = brick_xy & 7u, tiles_y = nby.divisor >> 2;
        const unsigned tx = t / tiles_y, ty = t - tx * tiles_y;
        bx = static_cast<int>(tx * 2 + (in >> 2));
        by = static_cast<int>(ty * 4 + (in & 3u));
    }
    // lane -> voxel of the brick
    const int chunk = (lane >> 2) & 7;
    const int grp = ((lane >> 5) << 1) | (__builtin_popcount(chunk) & 1);
    const int idx = ((chunk >> 1) << 2) | (lane & 3);
    const int lx = wave, ly = (grp << 1) | (idx >> 3), lz = idx & 7;
    const int64_t n_vox = static_cast<int64_t>(vx) * vy * vz;
    const int64_t n = (static_cast<int64_t>(bx * kBrick + lx) * vy + by * kBrick + ly) * vz + bz * kBrick + lz;
    const size_t img_stride = static_cast<size_t>(H) * W * C;
    const float* fb = feat_cl + b * NV * img_stride;
#ifdef MVX_BRICK_TRACE
    unsigned long long trace_[kTraceStamps] = {};
#endif
    MVX_TRACE(0);

    if (tid < NV * 12) sh->cam[tid] = camera_entry(proj + b * NV * 12 + (tid / 12) * 12, tid % 12, sx, sy);
    if (tid == 64) {
        sh->bad = 0;
        sh->redo = 0;
    }
    __syncthreads();

    // 1./2. projection, patch rectangles and per-lane check, a pair of views at a time: the patches of views 0 and 1
    // are on their way into LDS while views 2 and 3 are projected.
    double X, Y, Z;
    // world position of this lane's voxel; `lz_` is lz (passed so that a second evaluation is not merged with the first)
    auto load_position = [&](int lz_) __attribute__((always_inline)) {
        if (GRID) {
            FrameGrid fg;
            fg.load(coord.grid, coord.first_frame + b, vy);
            float x, y, z;
            fg.at(bx * kBrick + lx, by * kBrick + ly, bz * kBrick + lz_, x, y, z);
            X = x;
            Y = y;
            Z = z;
        } else {
            const float* cp = coord.volume + (b * n_vox + (n - lz + lz_)) * 3;
            X = cp[0];
            Y = cp[1];
            Z = cp[2];
        }
    };
    MVX_TRACE(1);  // cameras in LDS (first barrier)
    load_position(lz);
    float4 wt[NV];
    int txb[NV], tyb[NV];
    int ox[NV], oy[NV], rows[NV];
    unsigned a0[NV], a1[NV];  // LDS addresses of the taps' two rows (see patch_taps)
    bool fits = true, inside = true;
    const bool corner = (lx == 0 || lx == kBrick - 1) && (ly == 0 || ly == kBrick - 1) && (lz == 0 || lz == kBrick - 1);
    const int corner_id = ((lx & 4) | (ly & 2) | (lz & 1));  // for lx, ly, lz in {0, 7}

    // projects the (up to) two views of a pair; the corner voxels leave their floor positions for everybody
    auto project_pair = [&](int pass) {
#pragma unroll
        for (int vv = 0; vv < 2; ++vv) {
            const int v = pass * 2 + vv;
            if (v < NV) {
                const TapXY t = project_tap_xy(sh->cam + v * 12, X, Y, Z, H, W, AGG == MVX_AGG_SOFTMAX ? kLog2e : 1.0f);
                wt[v] = make_float4(t.w00, t.w01, t.w10, t.w11);
                txb[v] = t.xb;
                tyb[v] = t.yb;
                if (corner) {
                    sh->corner[v][0][corner_id] = t.behind ? INT_MAX : t.x0;
                    sh->corner[v][1][corner_id] = t.y0;
                }
            }
        }
    };
    // patch rectangles of the views of a pair (wave-uniform): minimum and maximum over each group of 8 lanes holding
    // the 8 corners of one (view, axis); then the per-lane check and the tap addresses.  View v lives in buffer v & 1.
    auto rectangles_pair = [&](int pass) {
        int cmin = reinterpret_cast<const int*>(sh->corner)[pass * 32 + (lane & 31)], cmax = cmin;
        cmin = min(cmin, dpp_movi<0xB1>(cmin));
        cmax = max(cmax, dpp_movi<0xB1>(cmax));
        cmin = min(cmin, dpp_movi<0x4E>(cmin));
        cmax = max(cmax, dpp_movi<0x4E>(cmax));
        cmin = min(cmin, dpp_movi<0x141>(cmin));
        cmax = max(cmax, dpp_movi<0x141>(cmax));
#pragma unroll
        for (int vv = 0; vv < 2; ++vv) {
            const int v = pass * 2 + vv;
            if (v < NV) {
                const int minx = __builtin_amdgcn_readlane(cmin, vv * 16), miny = __builtin_amdgcn_readlane(cmin, vv * 16 + 8);
                const int maxx = __builtin_amdgcn_readlane(cmax, vv * 16), maxy = __builtin_amdgcn_readlane(cmax, vv * 16 + 8);
                ox[v] = min(max(minx, 0), W - 2);
                oy[v] = min(max(miny, 0), H - 2);
                const int ex = min(max(maxx, 0), W - 2) + 1, ey = min(max(maxy, 0), H - 2) + 1;  // last column / row read
                rows[v] = ey - oy[v] + 1;
                // a corner behind the camera reads as INT_MAX: no patch
                fits = fits && (maxx != INT_MAX) && (ex - ox[v] + 1 <= kPatchW) && (rows[v] <= kPatchRows);
                const unsigned any = __builtin_bit_cast(unsigned, wt[v].x) | __builtin_bit_cast(unsigned, wt[v].y) |
                                     __builtin_bit_cast(unsigned, wt[v].z) | __builtin_bit_cast(unsigned, wt[v].w);
                // voxels whose weights are all zero read the patch origin (always staged) instead of wherever the clamp went
                const int rx = any ? txb[v] - ox[v] : 0, ry = any ? tyb[v] - oy[v] : 0;
                inside = inside && rx >= 0 && rx + 1 < kPatchW && ry >= 0 && ry + 1 < rows[v];
                const unsigned pixel = static_cast<unsigned>(reinterpret_cast<uintptr_t>((lds_byte*)(smem))) + vv * kPatchBytes +
                                       static_cast<unsigned>(ry * kPatchW + rx) * kPixelBytes;
                a0[v] = pixel + (static_cast<unsigned>(ry & 7) << 4);
                a1[v] = pixel + kPatchW * kPixelBytes + (static_cast<unsigned>((ry + 1) & 7) << 4);
            }
        }
    };
    // starts the copy of the patch of view v into buffer v & 1: rows of 16 pixels, half a row (8 pixels x 128 bytes)
    // per wave-instruction, row py by wave 2 (py & 3) + half.  The 16-byte channel quad q of a pixel in patch row py
    // sits at position q ^ (py & 7) of the pixel's 128 bytes: rows prow and prow + 8 have py & 7 = prow, rows
    // prow + 4 and prow + 12 have prow ^ 4, so two lane offsets computed once serve the four rows.
    auto stage_one = [&](int v) {
        global_byte* img = (global_byte*)(fb + v * img_stride);
        lds_byte* dst = (lds_byte*)(smem) + (v & 1) * kPatchBytes;
        const int half = wave & 1, prow = wave >> 1;
        const unsigned col = static_cast<unsigned>(min(ox[v] + (half << 3) + (lane >> 3), W - 1)) * kPixelBytes;
        const unsigned sw = static_cast<unsigned>((lane & 7) ^ prow) << 4;
        const unsigned voff0 = col + sw, voff1 = col + (sw ^ 64u);
#pragma unroll
        for (int k = 0; k < kPatchRows / 4; ++k) {
            const int py = k * 4 + prow;  // wave-uniform
            if (py < rows[v]) {
                const unsigned row_off = static_cast<unsigned>((oy[v] + py) * W) * kPixelBytes;  // scalar
                __builtin_amdgcn_global_load_lds(img + row_off + ((k & 1) ? voff1 : voff0),
                                                 dst + (py * kPatchW + (half << 3)) * kPixelBytes, 16, 0, 0);
            }
        }
    };

    project_pair(0);
    MVX_TRACE(2);  // position + projection of views 0, 1
    __syncthreads();
    MVX_TRACE(3);  // second barrier (corners)
    rectangles_pair(0);
    if (fits) {  // harmless if a lane turns out not to fit: the vote below decides who reads
        stage_one(0);
        stage_one(1);
    }
    MVX_TRACE(4);  // rectangles, copies issued
    if (NV > 2) project_pair(1);  // (the copies are in flight)
    MVX_TRACE(5);  // projection of views 2, 3
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    MVX_TRACE(6);  // own copies landed
    __syncthreads();
    MVX_TRACE(7);  // third barrier
    if (NV > 2) rectangles_pair(1);
    if (!inside) sh->bad = 1;
    __syncthreads();
    MVX_TRACE(8);  // rectangles of views 2, 3 and fourth barrier
    const bool staged = fits && (sh->bad == 0);

    // voxel of lane (lane & ~3), channel (lane & 3) of a quad: where this lane's 16-byte stores start
    const int64_t quad_off = b * C * n_vox + (n - (lane & 3)) + static_cast<int64_t>(lane & 3) * n_vox;
#ifdef MVX_BRICK_STATS  // development builds only (tools/brick_stats.hip): how many bricks leave the staged path, and why
    if (tid == 0) {
        atomicAdd(&brick_stats[0], 1u);
        if (!staged) atomicAdd(&brick_stats[1], 1u);
        if (!fits) atomicAdd(&brick_stats[2], 1u);
        int most = rows[0];
#pragma unroll
        for (int v = 1; v < NV; ++v) most = max(most, rows[v]);
        atomicMax(&brick_stats[3], static_cast<unsigned>(most));
    }
#endif
    // Direct path: taps from global memory with the tile kernel's access shape -- 8 lanes per voxel, each fetching
    // its 16 bytes of whole 128-byte lines -- and the tile kernel's arithmetic (four views: all loads of a step in
    // flight together and the shifted softmax; two or three: view by view).  The weights and tap pixels change
    // lanes through tables in the (unused) patch area; slots are ordered so that the 8 voxels of a step are
    // consecutive along z.
    auto direct_path = [&]() __attribute__((always_inline)) {
        float4* wts = reinterpret_cast<float4*>(smem);                                            // [NV][512]
        unsigned* pix = reinterpret_cast<unsigned*>(smem + NV * kBrickThreads * sizeof(float4));  // [NV][512]
        unsigned* vox = pix + NV * kBrickThreads;                                                 // [512]
        const int slot = wave * 64 + ly * kBrick + lz;
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            wts[v * kBrickThreads + slot] = wt[v];
            pix[v * kBrickThreads + slot] = static_cast<unsigned>(tyb[v] * W + txb[v]);
        }
        vox[slot] = static_cast<unsigned>(n);
        __syncthreads();
        const int q = lane & 7;
        float cc[NV][4] = {};
        if (AGG == MVX_AGG_CONF) {
#pragma unroll
            for (int v = 0; v < NV; ++v) {
                const float4 c4 = *reinterpret_cast<const float4*>(vconf + (b * NV + v) * C + q * 4);
                cc[v][0] = c4.x;
                cc[v][1] = c4.y;
                cc[v][2] = c4.z;
                cc[v][3] = c4.w;
            }
        }
#pragma unroll 1
        for (int step = 0; step < 8; ++step) {
            const int t = wave * 64 + step * 8 + (lane >> 3);
            float* dst = vol + b * C * n_vox + vox[t];
            if constexpr (NV == 4) {
                float4 w4[NV];
                Taps4 f[NV];
#pragma unroll
                for (int v = 0; v < NV; ++v) {
                    w4[v] = wts[v * kBrickThreads + t];
                    f[v] = load_taps(fb + v * img_stride, pix[v * kBrickThreads + t], W, C, q);
                }
                float val[NV][4];
#pragma unroll
                for (int v = 0; v < NV; ++v) interpolate(f[v], w4[v], val[v]);
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    dst[static_cast<int64_t>(q * 4 + k) * n_vox] = combine_four<AGG>(
                        val[0][k], val[1][k], val[2][k], val[NV - 1][k], cc[0][k], cc[1][k], cc[2][k], cc[NV - 1][k]);
            } else {
                Aggregator<AGG> agg;
                agg.init();
#pragma unroll
                for (int v = 0; v < NV; ++v) {
                    const float4 w4 = wts[v * kBrickThreads + t];
                    Tap tap;
                    tap.w00 = w4.x;
                    tap.w01 = w4.y;
                    tap.w10 = w4.z;
                    tap.w11 = w4.w;
                    tap.pixel = pix[v * kBrickThreads + t];
                    float val[4];
                    sample4(fb + v * img_stride, tap, W, C, q, val);
                    agg.add(val, cc[v]);
                }
#pragma unroll
                for (int k = 0; k < 4; ++k) dst[static_cast<int64_t>(q * 4 + k) * n_vox] = agg.result(k);
            }
        }
    };
    if (!staged) {
        direct_path();
        return;
    }

    // 3./4. one unit per view.  View v lives in buffer v & 1: while unit v - 1 is sampled the patch of view v + 1 is
    // on its way into the buffer that unit v - 2 has left, so no wave waits for a copy that has just been issued.
    // The views are aggregated one at a time in registers.  sum / conf / max: the same operations in the same order
    // as the tile kernel (bit-identical).  The view softmax sum_v x_v e^{x_v} / sum_v e^{x_v} is accumulated without
    // subtracting the maximum over views -- two registers per channel, no second look at the values -- which is the
    // same quotient as long as no term overflows and the largest term is not flushed.  That is checked, not assumed:
    // every denominator must lie in [2^-80, 2^100] (a term below 2^-46 of the largest may then have been lost to
    // underflow at most); a brick that fails the check is computed again by the direct path, which evaluates the
    // shifted form.  Results differ from the shifted form by the rounding of the exponentials only (a few float32
    // ulps).
    float2v acc[C / 2];  // sum / conf / max: the aggregate; softmax: the numerator
    float2v den[AGG == MVX_AGG_SOFTMAX ? C / 2 : 1];
    float2v chk_d = {0.0f, 0.0f}, chk_r = {0.0f, 0.0f};
#pragma unroll
    for (int v = 0; v < NV; ++v) {
        if (v >= 2) {
            // this wave's copies of view v have landed; then: everybody's have, and every lane has read view v - 1
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
            if (v + 1 < NV) stage_one(v + 1);
        } else if (v == 1 && NV > 2) {
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");  // every lane has read view 0
            stage_one(2);
        }
        MVX_TRACE(9 + 2 * v);  // barrier ahead of unit v (and the next copy issued)
        constexpr int kLast = NV - 1;
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            float2v val[2];
            const Taps4 f = patch_taps(a0[v], a1[v], q);
            interpolate_x2(f, wt[v], val[0], val[1]);
            float2v cc[2] = {};
            if (AGG == MVX_AGG_CONF) {
                const float* cv = vconf + (b * NV + v) * C + q * 4;
                cc[0] = float2v{cv[0], cv[1]};
                cc[1] = float2v{cv[2], cv[3]};
            }
            float res[4];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int cp = q * 2 + h;  // channel pair
                if (AGG == MVX_AGG_SUM) {
                    acc[cp] = v == 0 ? val[h] : acc[cp] + val[h];
                } else if (AGG == MVX_AGG_CONF) {
                    acc[cp] = v == 0 ? val[h] * cc[h] : __builtin_elementwise_fma(val[h], cc[h], acc[cp]);
                } else if (AGG == MVX_AGG_MAX) {
                    // (the tile kernel starts two or three views from -infinity, which is not the identity for a NaN)
                    const float2v prev = v == 0 ? float2v{-INFINITY, -INFINITY} : acc[cp];
                    acc[cp] = (v == 0 && NV == 4) ? val[h] : float2v{fmaxf(prev.x, val[h].x), fmaxf(prev.y, val[h].y)};
                } else {
                    const float2v e = {__builtin_amdgcn_exp2f(val[h].x), __builtin_amdgcn_exp2f(val[h].y)};
                    den[cp] = v == 0 ? e : den[cp] + e;
                    acc[cp] = v == 0 ? e * val[h] : __builtin_elementwise_fma(e, val[h], acc[cp]);
                }
                // pin the running values here: left alone, the compiler sinks the interpolation towards the next
                // unit and carries the raw taps (four times as many registers) instead
                asm volatile("" : "+v"(acc[cp]));
                if (AGG == MVX_AGG_SOFTMAX) asm volatile("" : "+v"(den[cp]));
                if (v == kLast) {
                    float2v r = acc[cp];
                    if (AGG == MVX_AGG_SOFTMAX) {
                        const float2v rden = {__builtin_amdgcn_rcpf(den[cp].x), __builtin_amdgcn_rcpf(den[cp].y)};
                        chk_d = chk_d + den[cp];
                        chk_r = chk_r + rden;
                        r = r * rden * float2v{kLn2, kLn2};
                    }
                    res[h * 2] = r.x;
                    res[h * 2 + 1] = r.y;
                }
            }
            if (v == kLast) {
                // the four lanes of a quad hold four consecutive voxels along z: after the transpose each lane
                // stores one channel of all four with a single 16-byte store
                quad_transpose(res, lane);
                *reinterpret_cast<float4*>(vol + quad_off + static_cast<int64_t>(q * 4) * n_vox) =
                    make_float4(res[0], res[1], res[2], res[3]);
            }
            // keep the quads in order: hoisting the reads of later quads would need hundreds of registers
            __builtin_amdgcn_sched_barrier(0);
        }
        MVX_TRACE(10 + 2 * v);  // unit v sampled (the last one: stores issued)
    }
#ifdef MVX_BRICK_TRACE
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    MVX_TRACE(17);  // stores acknowledged
    if (lane == 0) {
        unsigned long long* dst = brick_trace + (static_cast<size_t>(blockIdx.x) * 8 + wave) * kTraceStamps;
        for (int i = 0; i < kTraceStamps; ++i) dst[i] = trace_[i];
    }
#endif
    if (AGG == MVX_AGG_SOFTMAX) {
        // every denominator in [2^-80, 2^100] (a NaN fails both comparisons)
        if (!(chk_d.x + chk_d.y <= 0x1p100f) || !(chk_r.x + chk_r.y <= 0x1p80f)) sh->redo = 1;
        // (a barrier that does not wait for the stores above: a wave may end with its stores still on their way)
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        if (sh->redo != 0) {
            // The stores above have left this wave before anybody stores the same voxels again.
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            // weights and tap pixels once more (rather than kept in registers through the sampling above)
            int lz_again = lz;
            asm volatile("" : "+v"(lz_again));
            load_position(lz_again);
            project_pair(0);
            if (NV > 2) project_pair(1);
            direct_path();
        }
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
                         const float* coords, const GridSpec* grid_spec, const float* vol_conf, int64_t n_frames, int n_views, int n_channels, int h,
                         int w, int vx, int vy, int vz, int aggregation, float* volumes, void* workspace,
                         size_t workspace_bytes, void* stream);
}  // namespace
}  // namespace mvx

extern "C" int mvx_unproject(const float* features, const float* proj, const float* coords, const float* vol_conf,
                             int64_t n_frames, int n_views, int n_channels, int h, int w, int vx, int vy, int vz,
                             int aggregation, float* volumes, void* workspace, size_t workspace_bytes, void* stream) {
    return mvx::unproject_any_layout("mvx_unproject", false, features, proj, coords, nullptr, vol_conf, n_frames, n_views,
                                     n_channels, h, w, vx, vy, vz, aggregation, volumes, workspace, workspace_bytes,
                                     stream);
}

extern "C" int mvx_unproject_channels_last(const float* features, const float* proj, const float* coords,
                                           const float* vol_conf, int64_t n_frames, int n_views, int n_channels, int h,
                                           int w, int vx, int vy, int vz, int aggregation, float* volumes,
                                           void* stream) {
    return mvx::unproject_any_layout("mvx_unproject_channels_last", true, features, proj, coords, nullptr, vol_conf, n_frames,
                                     n_views, n_channels, h, w, vx, vy, vz, aggregation, volumes, nullptr, 0, stream);
}

extern "C" int mvx_unproject_grid(const float* features, int channels_last, const float* proj, const float* position,
                                  const float* center, const float* rotation, float step0, float step1, float step2,
                                  int swap_yz_flip, const float* vol_conf, int64_t n_frames, int n_views,
                                  int n_channels, int h, int w, int vx, int vy, int vz, int aggregation, float* volumes,
                                  void* workspace, size_t workspace_bytes, void* stream) {
    if (n_frames > 0) {
        const int rc = mvx::check_grid_spec("mvx_unproject_grid", position, center, rotation, swap_yz_flip, vy, vz);
        if (rc != MVX_OK) return rc;
        MVX_REQUIRE(static_cast<int64_t>(vx) * vy * vz < (1LL << 31), MVX_ERR_UNSUPPORTED,
                    "mvx_unproject_grid: grid of %dx%dx%d voxels is too large", vx, vy, vz);
    }
    const mvx::GridSpec spec{position, center, rotation, step0, step1, step2, swap_yz_flip};
    return mvx::unproject_any_layout("mvx_unproject_grid", channels_last != 0, features, proj, nullptr, &spec, vol_conf,
                                     n_frames, n_views, n_channels, h, w, vx, vy, vz, aggregation, volumes, workspace,
                                     workspace_bytes, stream);
}

namespace mvx {
namespace {
int unproject_any_layout(const char* name, bool channels_last, const float* features, const float* proj,
                         const float* coords, const GridSpec* grid_spec, const float* vol_conf, int64_t n_frames, int n_views, int n_channels, int h,
                         int w, int vx, int vy, int vz, int aggregation, float* volumes, void* workspace,
                         size_t workspace_bytes, void* stream) {
    MVX_REQUIRE(n_frames >= 0 && n_views > 0 && n_channels > 0 && h >= 2 && w >= 2 && vx >= 0 && vy >= 0 && vz >= 0,
                MVX_ERR_INVALID_ARGUMENT, "%s: bad shape n_frames=%lld n_views=%d n_channels=%d h=%d w=%d grid=%dx%dx%d",
                name, (long long)n_frames, n_views, n_channels, h, w, vx, vy, vz);
    const int64_t n_voxels = static_cast<int64_t>(vx) * vy * vz;
    MVX_REQUIRE(aggregation >= MVX_AGG_SUM && aggregation <= MVX_AGG_SOFTMAX, MVX_ERR_INVALID_ARGUMENT,
                "%s: unknown aggregation %d", name, aggregation);
    if (n_frames == 0 || n_voxels == 0) return MVX_OK;
    MVX_REQUIRE(features != nullptr && proj != nullptr && (coords != nullptr || grid_spec != nullptr) && volumes != nullptr &&
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
    // the brick kernel: three or four views of 32 channels into a grid whose sides are multiples of 8 (the reference's
    // 64^3 set-up; its collate function draws 2 to 4 cameras).  Two views are left to the tile kernel, which is the
    // faster one there (0.87 against 0.94 ms per 64 frames: two units do not amortise a brick's set-up); other shapes
    // take the tile or the any-shape kernel.
    const bool bricked = (C == 32) && (V == 3 || V == 4) && (vx % kBrick == 0) && (vy % kBrick == 0) && (vz % kBrick == 0) &&
                         (static_cast<int64_t>(h) * w * C * 4 < (1LL << 31)) && (n_voxels < (1LL << 32)) &&
                         aligned16(volumes) &&
                         (aggregation != MVX_AGG_CONF || aligned16(vol_conf)) &&
                         !(kernel_selection() & MVX_SELECT_NO_UNPROJECT_BRICKS);
    const size_t smem = static_cast<size_t>(32) * kTile * 4 + static_cast<size_t>(V < kTileViewChunk ? V : kTileViewChunk) * kTile * 20 +
                        static_cast<size_t>(V) * 12 * 8;
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
            if (C % 32 == 0 && HW % 64 == 0 && aligned16(features) && aligned16(workspace))
                hipLaunchKernelGGL(to_channel_last_x4, dim3(static_cast<unsigned>(t_blocks)), dim3(kThreads), 0, st,
                                   features + f0 * V * C * HW, C, static_cast<int>(HW), p_tiles, c_tiles,
                                   static_cast<float*>(workspace));
            else
                hipLaunchKernelGGL(to_channel_last, dim3(static_cast<unsigned>(t_blocks)), dim3(kThreads), 0, st,
                                   features + f0 * V * C * HW, C, Cp, HW, p_tiles, c_tiles, static_cast<float*>(workspace));
            MVX_CHECK_LAUNCH("mvx_unproject (channel-last copy)");
            feat_cl = static_cast<const float*>(workspace);
        }
        const float* pj = proj + f0 * V * 12;
        VoxelSource cd{};
        cd.volume = coords != nullptr ? coords + f0 * n_voxels * 3 : nullptr;
        if (grid_spec != nullptr) {
            cd.grid = *grid_spec;
            cd.has_grid = 1;
        }
        cd.first_frame = f0;
        cd.vy = vy;
        cd.vz = vz;
        const float* vc = vol_conf != nullptr ? vol_conf + f0 * V * C : nullptr;
        float* vo = volumes + f0 * C * n_voxels;
        if (bricked) {
            const int bricks = (vx / kBrick) * (vy / kBrick) * (vz / kBrick);
            MVX_REQUIRE(nf * bricks <= 0x7fffffffLL, MVX_ERR_UNSUPPORTED, "%s: voxel grid too large", name);
            const dim3 grid(static_cast<unsigned>(nf * bricks));
            const size_t brick_smem = 2 * kPatchBytes + sizeof(BrickShared);
            const FastDiv d_bricks = fast_divisor(static_cast<unsigned>(bricks));
            const FastDiv d_nbz = fast_divisor(static_cast<unsigned>(vz / kBrick));
            const FastDiv d_nby = fast_divisor(static_cast<unsigned>(vy / kBrick));
#define MVX_LAUNCH_BRICK_NV(AGG, NV)                                                                                   \
    if (cd.has_grid)                                                                                                    \
        hipLaunchKernelGGL((unproject_brick32<AGG, true, NV>), grid, dim3(kBrickThreads), brick_smem, st, feat_cl, pj, cd, \
                           vc, h, w, vx, vy, vz, d_bricks, d_nbz, d_nby, sx, sy, vo);                                   \
    else                                                                                                                \
        hipLaunchKernelGGL((unproject_brick32<AGG, false, NV>), grid, dim3(kBrickThreads), brick_smem, st, feat_cl, pj, cd, \
                           vc, h, w, vx, vy, vz, d_bricks, d_nbz, d_nby, sx, sy, vo)
#define MVX_LAUNCH_BRICK(AGG)                      \
    if (V == 4) {                                  \
        MVX_LAUNCH_BRICK_NV(AGG, 4);               \
    } else {                                       \
        MVX_LAUNCH_BRICK_NV(AGG, 3);               \
    }
            switch (aggregation) {
                case MVX_AGG_SUM: MVX_LAUNCH_BRICK(MVX_AGG_SUM); break;
                case MVX_AGG_CONF: MVX_LAUNCH_BRICK(MVX_AGG_CONF); break;
                case MVX_AGG_MAX: MVX_LAUNCH_BRICK(MVX_AGG_MAX); break;
                default: MVX_LAUNCH_BRICK(MVX_AGG_SOFTMAX); break;
            }
#undef MVX_LAUNCH_BRICK_NV
#undef MVX_LAUNCH_BRICK
        } else if (tiled) {
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
