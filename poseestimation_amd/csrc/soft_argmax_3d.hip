// Soft-argmax over voxel grids with world (or index) coordinates.
//
// Pass 1 (partial): every wave owns one contiguous segment of voxels of one frame, loads the segment's
// coordinates once into registers and walks the frame's volumes (joints): wave max, exp, and the float64
// sums (S, Sx, Sy, Sz) of that segment, written as one 40-byte record.  No LDS, no barriers; the coordinate
// volume is read once per frame instead of once per joint.
// Pass 2 (combine): per (frame, volume) the records are combined (rescaled to the global max) into the
// coordinates.  Pass 3 (normalize, only when the normalised volumes are wanted): an element-wise stream that
// re-reads the volumes and writes exp(t - M) / S.  (Feeding the passes a few frames at a time so that the second
// read would hit the Infinity Cache was measured slower than full-batch launches: small launches leave CUs idle.)
//
// Coordinates come from a coordinate volume, from the voxel index (integrate_tensor_3d), or from the parameters of
// a regular grid (grid_coords.h), in which case no coordinate bytes are read at all.
//
// Algorithmic bytes per frame: J*N*4 (volumes) + N*12 (coordinates) [+ J*N*4 written] + J*12 out; with the
// write-back the volumes cross HBM a second time on the read side.  (A one-read variant -- teams of 4 to 16
// workgroups holding a volume in registers as exp(t - own max), exchanging 40-byte records through memory and
// rescaling on the way out -- was built, verified and measured slower on MI355X, 0.74 to 0.93 ms against 0.61 ms for
// 64 frames: a compute unit alternates between its read phase and its write phase instead of overlapping them.)
#include <stdlib.h>

#include <mutex>

#include "grid_coords.h"

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
        if (i >= n) return make_float4(fill, fill, fill, fill);
        typedef float v4 __attribute__((ext_vector_type(4)));
        const v4 r = __builtin_nontemporal_load(reinterpret_cast<const v4*>(p + i));  // each volume is streamed once per pass
        return make_float4(r.x, r.y, r.z, r.w);
    } else {
        float4 r;
        r.x = i + 0 < n ? p[i + 0] : fill;
        r.y = i + 1 < n ? p[i + 1] : fill;
        r.z = i + 2 < n ? p[i + 2] : fill;
        r.w = i + 3 < n ? p[i + 3] : fill;
        return r;
    }
}

// Where the voxel positions come from.
struct CoordSource {
    const float* volume;  // coordinate volume [n_frames][N][3], or NULL
    GridSpec grid;        // used when has_grid
    int has_grid;
};

// Coordinates of the four voxels i..i+3 of frame b (zeros past the end).
template <bool VEC>
__device__ __forceinline__ void load_coords4(const CoordSource& src, int64_t b, int64_t i, int64_t n, Grid3 g,
                                             float (&cx)[4], float (&cy)[4], float (&cz)[4]) {
    if (src.volume == nullptr) {  // voxel (x, y, z) of the flat index: index space, or the grid evaluated there
        FrameGrid fg;
        if (src.has_grid) fg.load(src.grid, b, g.sy);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int64_t v = i + k;
            const bool ok = v < n;
            // 32-bit arithmetic (the host limits the grid to 2^31 voxels); past the end any in-range index will do
            const unsigned vu = ok ? static_cast<unsigned>(v) : 0u;
            const unsigned row = vu / static_cast<unsigned>(g.sz);
            const int z = static_cast<int>(vu - row * static_cast<unsigned>(g.sz));
            const int x = static_cast<int>(row / static_cast<unsigned>(g.sy)), y = static_cast<int>(row) - x * g.sy;
            float px = static_cast<float>(x), py = static_cast<float>(y), pz = static_cast<float>(z);
            if (src.has_grid) fg.at(x, y, z, px, py, pz);
            cx[k] = ok ? px : 0.0f;
            cy[k] = ok ? py : 0.0f;
            cz[k] = ok ? pz : 0.0f;
        }
        return;
    }
    const float* coord = src.volume + b * n * 3;
    float f[12];
    if (VEC) {
        if (i < n) {
            const float4* p = reinterpret_cast<const float4*>(coord + i * 3);
            const float4 a = p[0], b4 = p[1], c = p[2];
            f[0] = a.x; f[1] = a.y; f[2] = a.z; f[3] = a.w;
            f[4] = b4.x; f[5] = b4.y; f[6] = b4.z; f[7] = b4.w;
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
__global__ __launch_bounds__(kThreads) void soft_argmax_3d_partial(const float* __restrict__ vol, CoordSource coord,
                                                                    int J,
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
#pragma unroll
    for (int k = 0; k < NV; ++k) load_coords4<VEC>(coord, b, base + (k * 64 + lane) * 4, n_vox, g, cx[k], cy[k], cz[k]);

    const float fill = softmax ? -INFINITY : 0.0f;
    const float* vb = vol + b * J * n_vox;
    // two joints are kept in flight behind the one being reduced, so that every wave always has 2 x NV KB of
    // loads outstanding
    float4 cur[NV], nxt[NV], nx2[NV];
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        cur[k] = load4<VEC>(vb, base + (k * 64 + lane) * 4, n_vox, fill);
        nxt[k] = make_float4(fill, fill, fill, fill);
        nx2[k] = nxt[k];
    }
    if (J > 1) {
#pragma unroll
        for (int k = 0; k < NV; ++k) nxt[k] = load4<VEC>(vb + n_vox, base + (k * 64 + lane) * 4, n_vox, fill);
    }

    for (int j = 0; j < J; ++j) {
        if (j + 2 < J) {
            const float* vn = vb + static_cast<int64_t>(j + 2) * n_vox;
#pragma unroll
            for (int k = 0; k < NV; ++k) nx2[k] = load4<VEC>(vn, base + (k * 64 + lane) * 4, n_vox, fill);
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
                const float e = softmax ? exp_neg(t[k][i] - mref) : relu_keep_nan(t[k][i]);
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
        for (int k = 0; k < NV; ++k) {
            cur[k] = nxt[k];
            nxt[k] = nx2[k];
        }
    }
}

// Pass 2.  One workgroup per (frame, volume): combines the volume's records into the coordinates and leaves
// (global max, 1 / sum) for the write-back pass.
//   mode 0: softmax, out = Sx/S      mode 1: relu, out = Sx (no normalisation)      mode 2: relu, out = Sx/S
__global__ __launch_bounds__(kThreads) void soft_argmax_3d_combine(int n_seg, int mode,
                                                                   const double* __restrict__ partial,
                                                                   float* __restrict__ out,
                                                                   float2* __restrict__ stats) {
    __shared__ double red[kThreads / 64][4];
    __shared__ float redm[kThreads / 64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t volume = blockIdx.x;
    const bool softmax = mode == 0;
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
    if (tid == 0) {
        s = (red[0][0] + red[1][0]) + (red[2][0] + red[3][0]);
        sx = (red[0][1] + red[1][1]) + (red[2][1] + red[3][1]);
        sy = (red[0][2] + red[1][2]) + (red[2][2] + red[3][2]);
        sz = (red[0][3] + red[1][3]) + (red[2][3] + red[3][3]);
        const double d = (mode == 1) ? 1.0 : s;
        out[volume * 3 + 0] = static_cast<float>(sx / d);
        out[volume * 3 + 1] = static_cast<float>(sy / d);
        out[volume * 3 + 2] = static_cast<float>(sz / d);
        stats[volume] = make_float2(M, softmax ? static_cast<float>(1.0 / s) : 1.0f);
    }
}

typedef float vec4f __attribute__((ext_vector_type(4)));

// Pass 3 (write-back only): p = exp(t - M) / S (or max(t, 0)) as a plain element-wise stream over all volumes,
// 4 KB per workgroup, streaming (non-temporal) loads and stores -- the read + write pattern HBM sustains best.
template <bool VEC>
__global__ __launch_bounds__(kThreads) void soft_argmax_3d_normalize(const float* __restrict__ vol, int64_t n_vox,
                                                                      int64_t n_total, float mult, int softmax,
                                                                      const float2* __restrict__ stats,
                                                                      float* __restrict__ vol_out) {
    const int64_t i = (static_cast<int64_t>(blockIdx.x) * kThreads + threadIdx.x) * 4;
    if (i >= n_total) return;
    if (VEC) {  // n_vox % 4 == 0: the four elements belong to one volume
        const float2 st = stats[i / n_vox];
        const vec4f v = __builtin_nontemporal_load(reinterpret_cast<const vec4f*>(vol + i));
        vec4f p;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float t = v[k] * mult;
            p[k] = softmax ? exp_neg(t - st.x) * st.y : relu_keep_nan(t);
        }
        __builtin_nontemporal_store(p, reinterpret_cast<vec4f*>(vol_out + i));
    } else {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (i + k < n_total) {
                const float2 st = stats[(i + k) / n_vox];
                const float t = vol[i + k] * mult;
                vol_out[i + k] = softmax ? exp_neg(t - st.x) * st.y : relu_keep_nan(t);
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
    return static_cast<size_t>(n_frames) * n_volumes * (n_seg * mvx::kRecord * sizeof(double) + sizeof(float2));
}

namespace mvx {
namespace {
// `index_space`: the expectation of the voxel index (integrate_tensor_3d) rather than of a position.
int soft_argmax_3d_any(const float* volumes, const CoordSource& coord, bool index_space, int64_t n_frames, int n_volumes,
                       int sx, int sy, int sz, float multiplier, int softmax, float* out, float* volumes_out,
                       void* workspace, size_t workspace_bytes, void* stream) {
    const float* coords_volume = coord.volume;
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
    MVX_REQUIRE(n_seg64 <= (1 << 24) && n_vox < (1LL << 31), MVX_ERR_UNSUPPORTED,
                "mvx_soft_argmax_3d: grid of %lld voxels is too large", (long long)n_vox);
    const int n_seg = static_cast<int>(n_seg64), bpf = static_cast<int>(bpf64);
    const int mode = softmax ? 0 : (index_space ? 2 : 1);
    hipStream_t st = static_cast<hipStream_t>(stream);
    double* partial = static_cast<double*>(workspace);

    const int64_t n_total = n_frames * n_volumes * n_vox;
    const int64_t norm_blocks = (n_total + kThreads * 4 - 1) / (kThreads * 4);
    MVX_REQUIRE(n_frames * bpf <= 0x7fffffffLL && n_frames * n_volumes <= 0x7fffffffLL && norm_blocks <= 0x7fffffffLL,
                MVX_ERR_UNSUPPORTED, "mvx_soft_argmax_3d: batch exceeds the launch limits");
    // the (max, 1/sum) pairs of pass 2 live behind the records (sized for the shortest segments)
    float2* stats = reinterpret_cast<float2*>(partial + n_frames * n_volumes * ((n_vox + 255) / 256) * kRecord);

    const dim3 grid1(static_cast<unsigned>(n_frames * bpf)), block(kThreads);
#define MVX_LAUNCH_PARTIAL(NV, VEC)                                                                              \
    hipLaunchKernelGGL((soft_argmax_3d_partial<NV, VEC>), grid1, block, 0, st, volumes, coord, n_volumes, \
                       n_vox, g, n_seg, bpf, multiplier, softmax, partial)
    if (nv == 4 && vec) MVX_LAUNCH_PARTIAL(4, true);
    else if (nv == 4) MVX_LAUNCH_PARTIAL(4, false);
    else if (vec) MVX_LAUNCH_PARTIAL(1, true);
    else MVX_LAUNCH_PARTIAL(1, false);
#undef MVX_LAUNCH_PARTIAL
    MVX_CHECK_LAUNCH("mvx_soft_argmax_3d (partial)");

    hipLaunchKernelGGL(soft_argmax_3d_combine, dim3(static_cast<unsigned>(n_frames * n_volumes)), block, 0, st, n_seg,
                       mode, partial, out, stats);
    MVX_CHECK_LAUNCH("mvx_soft_argmax_3d (combine)");
    if (volumes_out != nullptr) {
        const dim3 grid3(static_cast<unsigned>(norm_blocks));
        if (vec)
            hipLaunchKernelGGL(soft_argmax_3d_normalize<true>, grid3, block, 0, st, volumes, n_vox, n_total, multiplier,
                               softmax, stats, volumes_out);
        else
            hipLaunchKernelGGL(soft_argmax_3d_normalize<false>, grid3, block, 0, st, volumes, n_vox, n_total,
                               multiplier, softmax, stats, volumes_out);
        MVX_CHECK_LAUNCH("mvx_soft_argmax_3d (normalize)");
    }
    return MVX_OK;
}
}  // namespace
}  // namespace mvx

extern "C" int mvx_soft_argmax_3d(const float* volumes, const float* coords_volume, int64_t n_frames, int n_volumes,
                                  int sx, int sy, int sz, float multiplier, int softmax, float* out,
                                  float* volumes_out, void* workspace, size_t workspace_bytes, void* stream) {
    mvx::CoordSource src{};
    src.volume = coords_volume;
    return mvx::soft_argmax_3d_any(volumes, src, coords_volume == nullptr, n_frames, n_volumes, sx, sy, sz, multiplier,
                                   softmax, out, volumes_out, workspace, workspace_bytes, stream);
}

extern "C" int mvx_soft_argmax_3d_grid(const float* volumes, const float* position, const float* center,
                                       const float* rotation, float step0, float step1, float step2, int swap_yz_flip,
                                       int64_t n_frames, int n_volumes, int sx, int sy, int sz, float multiplier,
                                       int softmax, float* out, float* volumes_out, void* workspace,
                                       size_t workspace_bytes, void* stream) {
    if (n_frames > 0 && n_volumes > 0) {
        const int rc = mvx::check_grid_spec("mvx_soft_argmax_3d_grid", position, center, rotation, swap_yz_flip, sy, sz);
        if (rc != MVX_OK) return rc;
    }
    mvx::CoordSource src{};
    src.grid = mvx::GridSpec{position, center, rotation, step0, step1, step2, swap_yz_flip};
    src.has_grid = 1;
    return mvx::soft_argmax_3d_any(volumes, src, false, n_frames, n_volumes, sx, sy, sz, multiplier, softmax, out,
                                   volumes_out, workspace, workspace_bytes, stream);
}

// ---- the same operation for callers that cannot allocate -------------------------------------------------
// The partial records and the (max, 1 / sum) pairs live in a fixed array of the code object instead of a
// caller-owned workspace.  Every GPU holds its own copy of the array at its own address; the host-side name
// `scratch` is only a handle for hipGetSymbolAddress, never a pointer to pass to a kernel.
static __device__ __align__(256) unsigned char scratch[MVX_SOFT_ARGMAX_3D_STATIC_BYTES];

namespace mvx {
namespace {
constexpr int kMaxDevices = 64;
std::mutex scratch_mutex;
void* scratch_of_device[kMaxDevices];  // device address of `scratch` per HIP device, resolved on first use

int static_scratch(void** ptr) {
    int dev = -1;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess || dev < 0 || dev >= kMaxDevices) {
        set_error("mvx_soft_argmax_3d_static: hipGetDevice failed or device %d out of range: %s", dev,
                  hipGetErrorString(e));
        return MVX_ERR_HIP;
    }
    std::lock_guard<std::mutex> lock(scratch_mutex);
    if (scratch_of_device[dev] == nullptr) {
        void* p = nullptr;
        e = hipGetSymbolAddress(&p, HIP_SYMBOL(scratch));
        if (e != hipSuccess || p == nullptr) {
            set_error("mvx_soft_argmax_3d_static: hipGetSymbolAddress failed on device %d: %s", dev,
                      hipGetErrorString(e));
            return MVX_ERR_HIP;
        }
        scratch_of_device[dev] = p;
    }
    *ptr = scratch_of_device[dev];
    return MVX_OK;
}
}  // namespace
}  // namespace mvx

extern "C" int mvx_soft_argmax_3d_static(const float* volumes, const float* coords_volume, int64_t n_frames,
                                         int n_volumes, int sx, int sy, int sz, float multiplier, int softmax,
                                         float* out, float* volumes_out, void* stream) {
    using namespace mvx;
    const size_t need = mvx_soft_argmax_3d_workspace_bytes(n_frames, n_volumes, sx, sy, sz);
    MVX_REQUIRE(need <= MVX_SOFT_ARGMAX_3D_STATIC_BYTES, MVX_ERR_WORKSPACE,
                "mvx_soft_argmax_3d_static: %lld frames x %d volumes of %dx%dx%d need %zu bytes of partial records, the "
                "built-in buffer holds %zu; use mvx_soft_argmax_3d with a caller-owned workspace",
                (long long)n_frames, n_volumes, sx, sy, sz, need, (size_t)MVX_SOFT_ARGMAX_3D_STATIC_BYTES);
    void* ws = nullptr;
    const int rc = static_scratch(&ws);
    if (rc != MVX_OK) return rc;
    // every record and pair that the later passes read is rewritten by the earlier pass of the same call, so the
    // contents a previous call left behind need no clearing
    CoordSource src{};
    src.volume = coords_volume;
    return soft_argmax_3d_any(volumes, src, coords_volume == nullptr, n_frames, n_volumes, sx, sy, sz, multiplier,
                              softmax, out, volumes_out, ws, MVX_SOFT_ARGMAX_3D_STATIC_BYTES, stream);
}
