// Soft-argmax over voxel grids with world (or index) coordinates.
//
// Pass 1 (partial): every wave owns one contiguous segment of voxels of one frame, loads the segment's
// coordinates once into registers and walks the frame's volumes (joints): wave max, exp, and the float64
// sums (S, Sx, Sy, Sz) of that segment, written as one 40-byte record.  No LDS, no barriers; the coordinate
// volume is read once per frame instead of once per joint.
// Pass 2 (finish): per (frame, volume) the records are combined (rescaled to the global max) into the
// coordinates; when the normalised volumes are wanted the same kernel re-reads the volume and writes
// exp(t - M) / S.  (Feeding the two passes a few frames at a time so that the second read would hit the
// Infinity Cache was measured slower than two full-batch launches: the small launches leave CUs idle.)
//
// Algorithmic bytes per frame: J*N*4 (volumes) + N*12 (coordinates) [+ J*N*4 written] + J*12 out; with the
// write-back the volumes cross HBM a second time on the read side.
#include <stdlib.h>

#include "mvx_common.h"

namespace mvx {

namespace {

constexpr int kThreads = 256;
constexpr int kRecord = 5;  // doubles per partial record: max, S, Sx, Sy, Sz

struct Grid3 {
    int sx, sy, sz;
};

template <bool VEC>
__device__ __forceinline__ float4 load4(const float* __restrict__ p, int64_t i, int64_t n, float fill) {
    if (VEC) {
        return i < n ? *reinterpret_cast<const float4*>(p + i) : make_float4(fill, fill, fill, fill);
    } else {
        float4 r;
        r.x = i + 0 < n ? p[i + 0] : fill;
        r.y = i + 1 < n ? p[i + 1] : fill;
        r.z = i + 2 < n ? p[i + 2] : fill;
        r.w = i + 3 < n ? p[i + 3] : fill;
        return r;
    }
}

// Coordinates of the four voxels i..i+3 (zeros past the end).
template <bool VEC>
__device__ __forceinline__ void load_coords4(const float* __restrict__ coord, int64_t i, int64_t n, Grid3 g,
                                             float (&cx)[4], float (&cy)[4], float (&cz)[4]) {
    if (coord == nullptr) {  // index space: voxel (x, y, z) of the flat index
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int64_t v = i + k;
            const int z = static_cast<int>(v % g.sz);
            const int y = static_cast<int>((v / g.sz) % g.sy);
            const int x = static_cast<int>(v / (static_cast<int64_t>(g.sz) * g.sy));
            const bool ok = v < n;
            cx[k] = ok ? static_cast<float>(x) : 0.0f;
            cy[k] = ok ? static_cast<float>(y) : 0.0f;
            cz[k] = ok ? static_cast<float>(z) : 0.0f;
        }
        return;
    }
    float f[12];
    if (VEC) {
        if (i < n) {
            const float4* p = reinterpret_cast<const float4*>(coord + i * 3);
            const float4 a = p[0], b = p[1], c = p[2];
            f[0] = a.x; f[1] = a.y; f[2] = a.z; f[3] = a.w;
            f[4] = b.x; f[5] = b.y; f[6] = b.z; f[7] = b.w;
            f[8] = c.x; f[9] = c.y; f[10] = c.z; f[11] = c.w;
        } else {
#pragma unroll
            for (int k = 0; k < 12; ++k) f[k] = 0.0f;
        }
    } else {
#pragma unroll
        for (int k = 0; k < 12; ++k) f[k] = (i + k / 3) < n ? coord[i * 3 + k] : 0.0f;
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        cx[k] = f[3 * k + 0];
        cy[k] = f[3 * k + 1];
        cz[k] = f[3 * k + 2];
    }
}

// Pass 1.  grid.x = n_frames * blocks_per_frame; each wave handles segment (block_in_frame*4 + wave).
template <int NV, bool VEC>
__global__ __launch_bounds__(kThreads) void soft_argmax_3d_partial(const float* __restrict__ vol,
                                                                    const float* __restrict__ coord, int J,
                                                                    int64_t n_vox, Grid3 g, int n_seg,
                                                                    int blocks_per_frame, float mult, int softmax,
                                                                    double* __restrict__ partial) {
    constexpr int kSeg = 64 * NV * 4;
    const int lane = threadIdx.x & 63;
    const int64_t b = blockIdx.x / blocks_per_frame;
    const int seg = (blockIdx.x % blocks_per_frame) * (kThreads / 64) + (threadIdx.x >> 6);
    if (seg >= n_seg) return;  // whole wave leaves; nothing below synchronises across waves
    const int64_t base = static_cast<int64_t>(seg) * kSeg;

    float cx[NV][4], cy[NV][4], cz[NV][4];
    const float* cb = coord != nullptr ? coord + b * n_vox * 3 : nullptr;
#pragma unroll
    for (int k = 0; k < NV; ++k) load_coords4<VEC>(cb, base + (k * 64 + lane) * 4, n_vox, g, cx[k], cy[k], cz[k]);

    const float fill = softmax ? -INFINITY : 0.0f;
    const float* vb = vol + b * J * n_vox;
    float4 cur[NV], nxt[NV];
#pragma unroll
    for (int k = 0; k < NV; ++k) cur[k] = load4<VEC>(vb, base + (k * 64 + lane) * 4, n_vox, fill);

    for (int j = 0; j < J; ++j) {
        if (j + 1 < J) {
            const float* vn = vb + static_cast<int64_t>(j + 1) * n_vox;
#pragma unroll
            for (int k = 0; k < NV; ++k) nxt[k] = load4<VEC>(vn, base + (k * 64 + lane) * 4, n_vox, fill);
        }
        float t[NV][4];
#pragma unroll
        for (int k = 0; k < NV; ++k) {
            t[k][0] = cur[k].x * mult;
            t[k][1] = cur[k].y * mult;
            t[k][2] = cur[k].z * mult;
            t[k][3] = cur[k].w * mult;
        }
        float m = 0.0f;
        if (softmax) {
            m = -INFINITY;
#pragma unroll
            for (int k = 0; k < NV; ++k) m = fmaxf(m, fmaxf(fmaxf(t[k][0], t[k][1]), fmaxf(t[k][2], t[k][3])));
            m = wave_max(m);
        }
        // a segment holding only -inf contributes nothing (its record keeps max = -inf, sums = 0)
        const float mref = (m == -INFINITY) ? 0.0f : m;
        double s = 0.0, sx = 0.0, sy = 0.0, sz = 0.0;
#pragma unroll
        for (int k = 0; k < NV; ++k) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float e = softmax ? exp_neg(t[k][i] - mref) : fmaxf(t[k][i], 0.0f);
                const double ed = e;
                s += ed;
                sx = fma(ed, static_cast<double>(cx[k][i]), sx);
                sy = fma(ed, static_cast<double>(cy[k][i]), sy);
                sz = fma(ed, static_cast<double>(cz[k][i]), sz);
            }
        }
        s = wave_sum(s);
        sx = wave_sum(sx);
        sy = wave_sum(sy);
        sz = wave_sum(sz);
        if (lane == 0) {
            double* rec = partial + ((b * J + j) * n_seg + seg) * kRecord;
            rec[0] = static_cast<double>(m);
            rec[1] = s;
            rec[2] = sx;
            rec[3] = sy;
            rec[4] = sz;
        }
#pragma unroll
        for (int k = 0; k < NV; ++k) cur[k] = nxt[k];
    }
}

// Pass 2.  grid.x = n_volumes_total * blocks_per_volume.  Every block combines the volume's records; block 0
// of a volume stores the coordinates; with WRITE each block normalises its 256*16-voxel slice.
//   mode 0: softmax, out = Sx/S      mode 1: relu, out = Sx (no normalisation)      mode 2: relu, out = Sx/S
template <bool WRITE, bool VEC>
__global__ __launch_bounds__(kThreads) void soft_argmax_3d_finish(const float* __restrict__ vol, int64_t n_vox,
                                                                   int n_seg, int blocks_per_volume, float mult,
                                                                   int mode, const double* __restrict__ partial,
                                                                   float* __restrict__ out,
                                                                   float* __restrict__ vol_out) {
    __shared__ double red[kThreads / 64][4];
    __shared__ float redm[kThreads / 64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t volume = blockIdx.x / blocks_per_volume;
    const int slice = blockIdx.x % blocks_per_volume;
    const bool softmax = mode == 0;
    if (!softmax && slice != 0 && !WRITE) return;

    const double* rec = partial + volume * n_seg * kRecord;
    float M = 0.0f;
    if (softmax) {
        float m = -INFINITY;
        for (int i = tid; i < n_seg; i += kThreads) m = fmaxf(m, static_cast<float>(rec[i * kRecord]));
        m = wave_max(m);
        if (lane == 0) redm[wave] = m;
        __syncthreads();
        M = fmaxf(fmaxf(redm[0], redm[1]), fmaxf(redm[2], redm[3]));
    }
    double s = 0.0, sx = 0.0, sy = 0.0, sz = 0.0;
    if (softmax || slice == 0) {
        for (int i = tid; i < n_seg; i += kThreads) {
            const double* r = rec + i * kRecord;
            // a segment whose values are all -inf has weight exp(-inf) = 0
            const double wgt = softmax ? exp(r[0] - static_cast<double>(M)) : 1.0;
            s = fma(r[1], wgt, s);
            sx = fma(r[2], wgt, sx);
            sy = fma(r[3], wgt, sy);
            sz = fma(r[4], wgt, sz);
        }
        s = wave_sum(s);
        sx = wave_sum(sx);
        sy = wave_sum(sy);
        sz = wave_sum(sz);
        if (lane == 0) {
            red[wave][0] = s;
            red[wave][1] = sx;
            red[wave][2] = sy;
            red[wave][3] = sz;
        }
        __syncthreads();
        s = (red[0][0] + red[1][0]) + (red[2][0] + red[3][0]);
        sx = (red[0][1] + red[1][1]) + (red[2][1] + red[3][1]);
        sy = (red[0][2] + red[1][2]) + (red[2][2] + red[3][2]);
        sz = (red[0][3] + red[1][3]) + (red[2][3] + red[3][3]);
    }
    if (slice == 0 && tid == 0) {
        const double d = (mode == 1) ? 1.0 : s;
        out[volume * 3 + 0] = static_cast<float>(sx / d);
        out[volume * 3 + 1] = static_cast<float>(sy / d);
        out[volume * 3 + 2] = static_cast<float>(sz / d);
    }
    if (WRITE) {
        const float inv = softmax ? static_cast<float>(1.0 / s) : 1.0f;
        const float* src = vol + volume * n_vox;
        float* dst = vol_out + volume * n_vox;
        const int64_t first = static_cast<int64_t>(slice) * (kThreads * 16);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int64_t i = first + (k * kThreads + tid) * 4;
            if (i >= n_vox) continue;
            float4 t = load4<VEC>(src, i, n_vox, 0.0f);
            t.x *= mult;
            t.y *= mult;
            t.z *= mult;
            t.w *= mult;
            float4 p;
            if (softmax) {
                p.x = exp_neg(t.x - M) * inv;
                p.y = exp_neg(t.y - M) * inv;
                p.z = exp_neg(t.z - M) * inv;
                p.w = exp_neg(t.w - M) * inv;
            } else {
                p.x = fmaxf(t.x, 0.0f);
                p.y = fmaxf(t.y, 0.0f);
                p.z = fmaxf(t.z, 0.0f);
                p.w = fmaxf(t.w, 0.0f);
            }
            if (VEC) {
                *reinterpret_cast<float4*>(dst + i) = p;
            } else {
                dst[i] = p.x;
                if (i + 1 < n_vox) dst[i + 1] = p.y;
                if (i + 2 < n_vox) dst[i + 2] = p.z;
                if (i + 3 < n_vox) dst[i + 3] = p.w;
            }
        }
    }
}

// Voxels per wave segment for a batch: long segments amortise the per-joint wave reductions, short ones keep
// every CU busy when there are few frames.
int pick_nv(int64_t n_frames, int64_t n_vox) {
    const int64_t blocks_long = n_frames * ((n_vox + 4095) / 4096);
    return blocks_long >= 512 ? 4 : 1;
}

}  // namespace

}  // namespace mvx

extern "C" size_t mvx_soft_argmax_3d_workspace_bytes(int64_t n_frames, int n_volumes, int sx, int sy, int sz) {
    if (n_frames <= 0 || n_volumes <= 0 || sx <= 0 || sy <= 0 || sz <= 0) return 0;
    const int64_t n_vox = static_cast<int64_t>(sx) * sy * sz;
    const int64_t n_seg = (n_vox + 255) / 256;  // shortest segment the kernels use
    return static_cast<size_t>(n_frames) * n_volumes * n_seg * mvx::kRecord * sizeof(double);
}

extern "C" int mvx_soft_argmax_3d(const float* volumes, const float* coords_volume, int64_t n_frames, int n_volumes,
                                  int sx, int sy, int sz, float multiplier, int softmax, float* out,
                                  float* volumes_out, void* workspace, size_t workspace_bytes, void* stream) {
    using namespace mvx;
    MVX_REQUIRE(n_frames >= 0 && n_volumes >= 0 && sx > 0 && sy > 0 && sz > 0, MVX_ERR_INVALID_ARGUMENT,
                "mvx_soft_argmax_3d: bad shape n_frames=%lld n_volumes=%d grid=%dx%dx%d", (long long)n_frames,
                n_volumes, sx, sy, sz);
    if (n_frames == 0 || n_volumes == 0) return MVX_OK;
    MVX_REQUIRE(volumes != nullptr && out != nullptr && workspace != nullptr, MVX_ERR_INVALID_ARGUMENT,
                "mvx_soft_argmax_3d: null volumes, out or workspace pointer");
    MVX_REQUIRE(workspace_bytes >= mvx_soft_argmax_3d_workspace_bytes(n_frames, n_volumes, sx, sy, sz),
                MVX_ERR_WORKSPACE, "mvx_soft_argmax_3d: workspace of %zu bytes is too small", workspace_bytes);
    MVX_REQUIRE((reinterpret_cast<uintptr_t>(workspace) & 7u) == 0, MVX_ERR_INVALID_ARGUMENT,
                "mvx_soft_argmax_3d: workspace must be 8-byte aligned");
    const int64_t n_vox = static_cast<int64_t>(sx) * sy * sz;
    const Grid3 g{sx, sy, sz};
    const bool vec = (n_vox % 4 == 0) && aligned16(volumes) && (coords_volume == nullptr || aligned16(coords_volume)) &&
                     (volumes_out == nullptr || aligned16(volumes_out));
    const int nv = pick_nv(n_frames, n_vox);
    const int64_t seg_len = 64 * nv * 4;
    const int64_t n_seg64 = (n_vox + seg_len - 1) / seg_len;
    const int64_t bpf64 = (n_seg64 + 3) / 4;
    const int64_t bpv64 = volumes_out != nullptr ? (n_vox + kThreads * 16 - 1) / (kThreads * 16) : 1;
    MVX_REQUIRE(n_seg64 <= (1 << 24) && bpv64 <= (1 << 24), MVX_ERR_UNSUPPORTED,
                "mvx_soft_argmax_3d: grid of %lld voxels is too large", (long long)n_vox);
    const int n_seg = static_cast<int>(n_seg64), bpf = static_cast<int>(bpf64), bpv = static_cast<int>(bpv64);
    const int mode = softmax ? 0 : (coords_volume != nullptr ? 1 : 2);
    hipStream_t st = static_cast<hipStream_t>(stream);
    double* partial = static_cast<double*>(workspace);

    MVX_REQUIRE(n_frames * bpf <= 0x7fffffffLL && n_frames * n_volumes * bpv <= 0x7fffffffLL, MVX_ERR_UNSUPPORTED,
                "mvx_soft_argmax_3d: batch exceeds the launch limits");

    const dim3 grid1(static_cast<unsigned>(n_frames * bpf)), block(kThreads);
#define MVX_LAUNCH_PARTIAL(NV, VEC)                                                                              \
    hipLaunchKernelGGL((soft_argmax_3d_partial<NV, VEC>), grid1, block, 0, st, volumes, coords_volume, n_volumes, \
                       n_vox, g, n_seg, bpf, multiplier, softmax, partial)
    if (nv == 4 && vec) MVX_LAUNCH_PARTIAL(4, true);
    else if (nv == 4) MVX_LAUNCH_PARTIAL(4, false);
    else if (vec) MVX_LAUNCH_PARTIAL(1, true);
    else MVX_LAUNCH_PARTIAL(1, false);
#undef MVX_LAUNCH_PARTIAL
    MVX_CHECK_LAUNCH("mvx_soft_argmax_3d (partial)");

    const dim3 grid2(static_cast<unsigned>(n_frames * n_volumes * bpv));
#define MVX_LAUNCH_FINISH(WRITE, VEC)                                                                             \
    hipLaunchKernelGGL((soft_argmax_3d_finish<WRITE, VEC>), grid2, block, 0, st, volumes, n_vox, n_seg, bpv, \
                       multiplier, mode, partial, out, volumes_out)
    if (volumes_out != nullptr && vec) MVX_LAUNCH_FINISH(true, true);
    else if (volumes_out != nullptr) MVX_LAUNCH_FINISH(true, false);
    else MVX_LAUNCH_FINISH(false, false);
#undef MVX_LAUNCH_FINISH
    MVX_CHECK_LAUNCH("mvx_soft_argmax_3d (finish)");
    return MVX_OK;
}
