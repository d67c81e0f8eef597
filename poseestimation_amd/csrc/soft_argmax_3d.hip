// Soft-argmax over voxel grids with world (or index) coordinates.
//
// Coordinates only, or no softmax: three passes.
// Pass 1 (partial): every wave owns one contiguous segment of voxels of one frame, loads the segment's
// coordinates once into registers and walks the frame's volumes (joints): wave max, exp, and the float64
// sums (S, Sx, Sy, Sz) of that segment, written as one 40-byte record.  No LDS, no barriers; the coordinate
// volume is read once per frame instead of once per joint.
// Pass 2 (combine): per (frame, volume) the records are combined (rescaled to the global max) into the
// coordinates.  Pass 3 (normalize, only when the normalised volumes are wanted): an element-wise stream that
// re-reads the volumes and writes exp(t - M) / S.  (Feeding the passes a few frames at a time so that the second
// read would hit the Infinity Cache was measured slower than full-batch launches: small launches leave CUs idle.)
//
// Softmax with the normalised volumes wanted (what the volumetric model asks for): one kernel that reads every volume
// once, soft_argmax_3d_resident below.  The three passes stay as the path for shapes it does not take and behind
// MVX_SELECT_NO_SOFT_ARGMAX_RESIDENT.  Measured on MI355X, 64 frames x 17 volumes of 64^3: three passes 0.61 ms,
// one read 0.47 ms (regular grid) / 0.54 ms (coordinate volume).  What made the one-read kernel pay, in the order
// found: (1) no release fence on the hand-off -- it writes back the whole L2, which is full of streamed output --
// but write-through stores of self-validating records; (2) the records of a team read by one lane each instead of
// a serial loop of L2 round trips; (3) the share parked in LDS and written a step later, so no workgroup ever
// waits for another on its critical path, with the record loads issued behind the arithmetic; (4) for regular
// grids, moments of the voxel index (4 float64 operations per voxel instead of 17) mapped through the grid once
// per volume.
//
// Coordinates come from a coordinate volume, from the voxel index (integrate_tensor_3d), or from the parameters of
// a regular grid (grid_coords.h), in which case no coordinate bytes are read at all.
//
// Algorithmic bytes per frame: J*N*4 (volumes) + N*12 (coordinates) [+ J*N*4 written] + J*12 out.
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

// ---- one read per volume: resident workgroups that park a share of a volume in LDS ----------------------------
// Softmax with the normalised volumes wanted.  A volume is cut into `team` shares of 512 threads x NV4 float4.  A
// workgroup reduces a share from registers (max, then exp(t - own max) and the float64 sums), parks the exponentials in
// LDS and publishes one 40-byte record; one step later, when the records of the whole team have long been published,
// it combines them, rescales the parked share and writes it out.  Every volume crosses HBM once in each direction.
// With a coordinate volume the sums are over the positions read from it.  With a regular grid (and in index space) the
// sums are moments of the voxel index, four float64 operations per voxel, and the grid, affine in the index, is
// applied to their quotient once per volume, in float64.
//
// The grid is no larger than what is resident at once, and workgroup w walks the shares w, w + G, w + 2G, ...  At any
// time it has: the loads of its next share in flight (second register set), the share it reduces (first register
// set), and the previous share parked in the other LDS buffer.  Nothing on the critical path waits for another
// workgroup: the team's records are polled a full step after they were written.  G is a multiple of the team size, so
// the members of a team are the same step of `team` different workgroups, and in the normal case -- the whole grid
// resident -- every record is there when it is polled.  No workgroup depends on that: the poll is bounded, and a
// record that has not arrived by then (a team-mate whose slot another kernel or process holds) is recomputed from
// memory by the workgroup that needs it.  Progress never hangs on co-residency; only speed does.
constexpr int kShareThreads = 512;
constexpr int kTeamMax = 64;                    // one lane of a wave per member when the records are read
constexpr unsigned kTeamPolls = 64;             // polls (a few microseconds each) before a workgroup helps itself

// Coordinates of the four voxels i..i+3 (i a multiple of 4) from a frame's coordinate volume.
__device__ __forceinline__ void row_coords4(const float* __restrict__ coord_frame, unsigned i, float (&cx)[4],
                                            float (&cy)[4], float (&cz)[4]) {
    const float4* p = reinterpret_cast<const float4*>(coord_frame + static_cast<size_t>(i) * 3);
    const float4 a = p[0], b4 = p[1], c = p[2];
    cx[0] = a.x; cy[0] = a.y; cz[0] = a.z;
    cx[1] = a.w; cy[1] = b4.x; cz[1] = b4.y;
    cx[2] = b4.z; cy[2] = b4.w; cz[2] = c.x;
    cx[3] = c.y; cy[3] = c.z; cz[3] = c.w;
}

// Adds the four voxels i..i+3 of a share (exponentials e) to its sums: over the positions of a coordinate volume
// (SRC 0), or as moments of the voxel index (x, y, z..z+3 of one grid row).
template <int SRC>
__device__ __forceinline__ void add_moments(const float* __restrict__ coord_frame, unsigned i, const vec4f& e,
                                            FastDiv d_sy, FastDiv d_sz, double& s, double& sx, double& sy,
                                            double& sz) {
    if (SRC == 0) {
        float cx[4], cy[4], cz[4];
        row_coords4(coord_frame, i, cx, cy, cz);
#pragma unroll
        for (int l = 0; l < 4; ++l) {
            const double ed = e[l];
            s += ed;
            sx = fma(ed, static_cast<double>(cx[l]), sx);
            sy = fma(ed, static_cast<double>(cy[l]), sy);
            sz = fma(ed, static_cast<double>(cz[l]), sz);
        }
    } else {
        const unsigned row = fast_div(i, d_sz), z = i - row * d_sz.divisor;
        const unsigned x = fast_div(row, d_sy), y = row - x * d_sy.divisor;
        const double e0 = e[0], e1 = e[1], e2 = e[2], e3 = e[3];
        const double q = (e0 + e1) + (e2 + e3);
        s += q;
        sx = fma(q, static_cast<double>(x), sx);
        sy = fma(q, static_cast<double>(y), sy);
        sz = fma(q, static_cast<double>(z), sz) + fma(3.0, e3, fma(2.0, e2, e1));
    }
}

// The record (max, S, Sx, Sy, Sz) of the share of `share_vox` voxels starting at voxel `first` of a volume, computed
// by the calling wave alone from memory (two passes, few registers): what a workgroup falls back on when a
// team-mate's record does not arrive, so that no workgroup ever depends on another one being resident.  Kept out of
// line: it is cold, and inlined it costs the hot loop registers.  The sums are taken in another order than the
// owner's (their last bits differ).
template <int SRC>
__device__ __noinline__ void share_record_by_one_wave(const float* __restrict__ volume, const float* __restrict__ coord_frame,
                                                      unsigned first, unsigned share_vox, float mult, FastDiv d_sy,
                                                      FastDiv d_sz, double* __restrict__ rec) {
    const unsigned lane = threadIdx.x & 63u;
    const float* src = volume + first;
    float m = -INFINITY;
    for (unsigned i = lane * 4; i < share_vox; i += 64 * 4) {
        const vec4f t = *reinterpret_cast<const vec4f*>(src + i) * mult;
        m = fmaxf(m, fmaxf(fmaxf(t.x, t.y), fmaxf(t.z, t.w)));
    }
    m = wave_max(m);
    const float mref = (m == -INFINITY) ? 0.0f : m;
    double s = 0.0, sx = 0.0, sy = 0.0, sz = 0.0;
    for (unsigned i = lane * 4; i < share_vox; i += 64 * 4) {
        const vec4f t = *reinterpret_cast<const vec4f*>(src + i) * mult;
        const vec4f d = (t - mref) * 1.44269504088896340736f;
        vec4f e;
#pragma unroll
        for (int l = 0; l < 4; ++l) e[l] = __builtin_amdgcn_exp2f(d[l]);
        add_moments<SRC>(coord_frame, first + i, e, d_sy, d_sz, s, sx, sy, sz);
    }
    rec[0] = static_cast<double>(m);
    rec[1] = wave_sum(s);
    rec[2] = wave_sum(sx);
    rec[3] = wave_sum(sy);
    rec[4] = wave_sum(sz);
}

// records[volume][member] = (max, S, Sx, Sy, Sz) of the member's share,
// sums relative to its own max; all zero bits at launch (a clearing kernel ahead of it), never zero once published.
// Dynamic LDS: two buffers of 512 x NV4 float4.
template <int NV4, int SRC>
__global__ __launch_bounds__(kShareThreads, 4) void soft_argmax_3d_resident(
    const float* __restrict__ vol, CoordSource coord, int J, unsigned n_vox, Grid3 g, FastDiv d_sy, FastDiv d_sz, int team,
    unsigned n_shares, float mult, int help_always, double* __restrict__ records, float* __restrict__ out,
    float* __restrict__ vol_out) {
    extern __shared__ __align__(16) unsigned char smem[];
    vec4f* parked = reinterpret_cast<vec4f*>(smem);  // [2][NV4][512]
    __shared__ float s_max[kShareThreads / 64];
    __shared__ double s_sum[kShareThreads / 16][4];  // per 16-lane row: S, Sy, Sx, Sz
    __shared__ float s_scale[2];  // per LDS buffer: two write-outs can follow each other without a barrier between
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const unsigned share_vox = n_vox / static_cast<unsigned>(team);

    auto share_base = [&](unsigned item) {  // element offset of a share in the batch of volumes
        const unsigned volume = item / static_cast<unsigned>(team);
        const unsigned member = item - volume * static_cast<unsigned>(team);
        return static_cast<size_t>(volume) * n_vox + static_cast<size_t>(member) * share_vox;
    };
    auto load_share = [&](unsigned item, vec4f (&v)[NV4]) {
        const float* src = vol + share_base(item);
#pragma unroll
        for (int k = 0; k < NV4; ++k)
            v[k] = __builtin_nontemporal_load(reinterpret_cast<const vec4f*>(src + (k * kShareThreads + tid) * 4));
    };
    // Lane i of wave 0 reads the record of member i of a share's team with agent-scope loads (served by L2, never by
    // a stale L1 line).  A record is complete when none of its words is zero.
    long long w[kRecord];
    auto fetch_records = [&](unsigned item) {
        const unsigned volume = item / static_cast<unsigned>(team);
        const long long* theirs = reinterpret_cast<const long long*>(
            records + (static_cast<size_t>(volume) * team + min(lane, team - 1)) * kRecord);
#pragma unroll
        for (int i = 0; i < kRecord; ++i)
            w[i] = __hip_atomic_load(theirs + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    };
    // reduces a share, parks e = exp(t - own max) in LDS buffer `buf` and publishes its record; on the way, wave 0
    // fetches the records of the share written next (`pending`, published a step ago) behind the arithmetic
    auto reduce_share = [&](unsigned item, vec4f (&v)[NV4], int buf, bool has_pending, unsigned pending) {
        const unsigned volume = item / static_cast<unsigned>(team);
        const unsigned member = item - volume * static_cast<unsigned>(team);
        const int64_t b = volume / static_cast<unsigned>(J);
        const unsigned first = member * share_vox;
        float m = -INFINITY;
#pragma unroll
        for (int k = 0; k < NV4; ++k) {
            v[k] *= mult;
            m = fmaxf(m, fmaxf(fmaxf(v[k].x, v[k].y), fmaxf(v[k].z, v[k].w)));
        }
        m = wave_max(m);
        if (lane == 0) s_max[wave] = m;
        __syncthreads();  // also: every wave has finished reading this LDS buffer (the write-out two steps ago)
        m = s_max[0];
#pragma unroll
        for (int i = 1; i < kShareThreads / 64; ++i) m = fmaxf(m, s_max[i]);
        const float mref = (m == -INFINITY) ? 0.0f : m;  // a share holding only -inf contributes nothing
        if (wave == 0 && has_pending) fetch_records(pending);

        const float* coord_frame = SRC == 0 ? coord.volume + static_cast<size_t>(b) * n_vox * 3 : nullptr;
        double s = 0.0, sx = 0.0, sy = 0.0, sz = 0.0;
#pragma unroll
        for (int k = 0; k < NV4; ++k) {
            const unsigned i = first + (k * kShareThreads + tid) * 4;
            const vec4f d = (v[k] - mref) * 1.44269504088896340736f;  // exp_neg, with the packed subtract and multiply
            vec4f e;
#pragma unroll
            for (int l = 0; l < 4; ++l) e[l] = __builtin_amdgcn_exp2f(d[l]);
            add_moments<SRC>(coord_frame, i, e, d_sy, d_sz, s, sx, sy, sz);
            parked[(buf * NV4 + k) * kShareThreads + tid] = e;
        }
        // The four sums are reduced together.  Two quad exchanges hand each lane one quantity of its quad (lane & 3 =
        // 0, 1, 2, 3 holds S, Sy, Sx, Sz), two row shifts leave the totals of a 16-lane row in its lanes 12 to 15,
        // and those go to LDS: 32 row totals per quantity, which four lanes of wave 0 add up after the barrier.
        {
            const bool odd = lane & 1, upper = lane & 2;
            const double a0 = (odd ? sy : s) + dpp_mov<0xB1>(odd ? s : sy);
            const double a1 = (odd ? sz : sx) + dpp_mov<0xB1>(odd ? sx : sz);
            double t = (upper ? a1 : a0) + dpp_mov<0x4E>(upper ? a0 : a1);
            t += dpp_mov<0x114>(t);  // row_shr:4, zero shifted in
            t += dpp_mov<0x118>(t);  // row_shr:8
            if ((lane & 12) == 12) s_sum[wave * 4 + (lane >> 4)][lane & 3] = t;
        }
        __syncthreads();
        if (wave == 0) {
            double total = 0.0;
            for (int i = 0; i < kShareThreads / 16; ++i) total += s_sum[i][lane & 3];
            // Publish: agent-scope (write-through) 8-byte stores, one word per lane; +0.0, the one value that is
            // all zero bits, goes out as the smallest denormal, so no published word is zero.  There is no flag,
            // nothing to wait for on the writing side, and no cache write-back or invalidate: a release fence would
            // flush the whole L2 of the XCD, which is full of the volumes being streamed out.
            if (lane < 4) {
                long long* mine =
                    reinterpret_cast<long long*>(records + (static_cast<size_t>(volume) * team + member) * kRecord);
                const int word = lane == 0 ? 1 : (lane == 1 ? 3 : (lane == 2 ? 2 : 4));  // S, Sy, Sx, Sz
                const long long bits = __double_as_longlong(total);
                __hip_atomic_store(mine + word, bits != 0 ? bits : 1LL, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (lane == 0) {
                    const long long mbits = __double_as_longlong(static_cast<double>(m));
                    __hip_atomic_store(mine, mbits != 0 ? mbits : 1LL, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
        }
    };
    // combines the records of a share's team and writes the share parked in LDS buffer `buf`
    auto write_share = [&](unsigned item, int buf, bool prefetched) {
        const unsigned volume = item / static_cast<unsigned>(team);
        const unsigned member = item - volume * static_cast<unsigned>(team);
        if (wave == 0) {
            // The records were fetched during the reduction of the next share; fetch again, a bounded number of
            // times, until all are complete.  What is still missing then (a team-mate that is not resident: another
            // kernel holds its slot) this wave computes itself.
            unsigned spins = 0;
            if (!prefetched) fetch_records(item);
            unsigned long long missing;
            while (true) {
                bool ready = true;
#pragma unroll
                for (int i = 0; i < kRecord; ++i) ready = ready && w[i] != 0;
                missing = __builtin_amdgcn_ballot_w64(!ready || (help_always && static_cast<unsigned>(lane) != member));
                if (missing == 0 || help_always || spins++ >= kTeamPolls) break;
                __builtin_amdgcn_s_sleep(32);
                fetch_records(item);
            }
            const bool counts = lane < team;  // the lanes beyond the team re-read its last record
            double rm = counts ? __longlong_as_double(w[0]) : -INFINITY;
            double rs = counts ? __longlong_as_double(w[1]) : 0.0;
            double rx = counts ? __longlong_as_double(w[2]) : 0.0;
            double ry = counts ? __longlong_as_double(w[3]) : 0.0;
            double rz = counts ? __longlong_as_double(w[4]) : 0.0;
            if (team < 64) missing &= (1ull << team) - 1ull;
            while (missing != 0) {  // wave-uniform
                const int i = __builtin_ctzll(missing);
                missing &= missing - 1ull;
                double h[kRecord];
                const size_t b = volume / static_cast<unsigned>(J);
                share_record_by_one_wave<SRC>(vol + static_cast<size_t>(volume) * n_vox,
                                              SRC == 0 ? coord.volume + b * n_vox * 3 : nullptr,
                                              static_cast<unsigned>(i) * share_vox, share_vox, mult, d_sy, d_sz, h);
                if (lane == i) {
                    rm = h[0];
                    rs = h[1];
                    rx = h[2];
                    ry = h[3];
                    rz = h[4];
                }
            }
            const double M = static_cast<double>(wave_max(static_cast<float>(rm)));  // the maxima are float32 values
            const double wgt = exp(rm - M);  // a share of -inf, and a lane beyond the team, has weight 0
            const double S = wave_sum(rs * wgt);
            const double Sx = wave_sum(rx * wgt), Sy = wave_sum(ry * wgt), Sz = wave_sum(rz * wgt);
            const double own = __shfl(wgt, static_cast<int>(member));
            if (lane == 0) {
                if (member == 0) {
                    double ex = Sx / S, ey = Sy / S, ez = Sz / S;
                    if (SRC == 2) {
                        // expectation of the voxel index -> world position: the grid is affine in the index
                        const GridSpec& gs = coord.grid;
                        const size_t b = volume / static_cast<unsigned>(J);
                        const double i1 = gs.swap_yz_flip ? ez : ey;
                        const double i2 = gs.swap_yz_flip ? static_cast<double>(g.sy - 1) - ey : ez;
                        const double c0 = gs.center[b * 3 + 0], c1 = gs.center[b * 3 + 1], c2 = gs.center[b * 3 + 2];
                        const double d0 = (gs.position[b * 3 + 0] + static_cast<double>(gs.step0) * ex) - c0;
                        const double d1 = (gs.position[b * 3 + 1] + static_cast<double>(gs.step1) * i1) - c1;
                        const double d2 = (gs.position[b * 3 + 2] + static_cast<double>(gs.step2) * i2) - c2;
                        const float* r = gs.rotation + b * 9;
                        ex = fma(r[2], d2, fma(r[1], d1, r[0] * d0)) + c0;
                        ey = fma(r[5], d2, fma(r[4], d1, r[3] * d0)) + c1;
                        ez = fma(r[8], d2, fma(r[7], d1, r[6] * d0)) + c2;
                    }
                    out[static_cast<size_t>(volume) * 3 + 0] = static_cast<float>(ex);
                    out[static_cast<size_t>(volume) * 3 + 1] = static_cast<float>(ey);
                    out[static_cast<size_t>(volume) * 3 + 2] = static_cast<float>(ez);
                }
                s_scale[buf] = static_cast<float>(own / S);
            }
        }
        __syncthreads();
        const float scale = s_scale[buf];
        float* dst = vol_out + share_base(item);
#pragma unroll
        for (int k = 0; k < NV4; ++k) {
            const vec4f p = parked[(buf * NV4 + k) * kShareThreads + tid] * scale;
            __builtin_nontemporal_store(p, reinterpret_cast<vec4f*>(dst + (k * kShareThreads + tid) * 4));
        }
    };

    const unsigned G = gridDim.x;
    vec4f a[NV4], c[NV4];
    unsigned item = blockIdx.x;
    if (item >= n_shares) return;  // (the host launches no more workgroups than shares)
    load_share(item, a);
    if (item + G < n_shares) load_share(item + G, c);
    bool first_step = true;
    while (true) {
        // even step: share `item` from register set a into LDS buffer 0; the share of the step before leaves buffer 1
        reduce_share(item, a, 0, !first_step, item - G);
        if (item + 2 * G < n_shares) load_share(item + 2 * G, a);
        if (!first_step) write_share(item - G, 1, true);
        first_step = false;
        if (item + G >= n_shares) {
            write_share(item, 0, false);
            break;
        }
        // odd step: share item + G from register set c into LDS buffer 1; the even step's share leaves buffer 0
        reduce_share(item + G, c, 1, true, item);
        if (item + 3 * G < n_shares) load_share(item + 3 * G, c);
        write_share(item, 0, true);
        if (item + 2 * G >= n_shares) {
            write_share(item + G, 1, false);
            break;
        }
        item += 2 * G;
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
constexpr size_t resident_lds_bytes(int nv4) { return static_cast<size_t>(2) * nv4 * kShareThreads * 16; }

// Workgroups of soft_argmax_3d_resident<NV4, SRC> that the current device holds at once (cached per device):
// compute units x workgroups per compute unit as the runtime reports them, the latter capped at two.
__global__ void clear_records(unsigned long long* __restrict__ words, size_t n) {
    const size_t i = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i < n) words[i] = 0ull;
}

int resident_workgroups(int nv4, int source) {
    static std::mutex mutex;
    static int cached[64][3][3];  // [device][log2 nv4][source], 0 = not asked yet
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 0;
    const int slot = nv4 == 4 ? 2 : (nv4 == 2 ? 1 : 0);
    std::lock_guard<std::mutex> lock(mutex);
    if (cached[dev][slot][source] == 0) {
        int per_cu = 0, cus = 0;
        hipError_t e = hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
#define MVX_OCCUPANCY(NV4, SRC)                                                                                    \
    e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, soft_argmax_3d_resident<NV4, SRC>, kShareThreads, \
                                                     resident_lds_bytes(NV4))
#define MVX_OCCUPANCY_SRC(NV4)                  \
    if (source == 0) MVX_OCCUPANCY(NV4, 0);     \
    else if (source == 1) MVX_OCCUPANCY(NV4, 1); \
    else MVX_OCCUPANCY(NV4, 2)
        if (e == hipSuccess) {
            switch (nv4) {
                case 1: MVX_OCCUPANCY_SRC(1); break;
                case 2: MVX_OCCUPANCY_SRC(2); break;
                default: MVX_OCCUPANCY_SRC(4); break;
            }
        }
#undef MVX_OCCUPANCY_SRC
#undef MVX_OCCUPANCY
        if (e != hipSuccess || per_cu < 1 || cus < 1) return 0;
        cached[dev][slot][source] = cus * (per_cu > 2 ? 2 : per_cu);
    }
    return cached[dev][slot][source];
}

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
    // In place (volumes_out == volumes) is served by the three passes, whose last pass rewrites every element from
    // its own value; the one-read kernel may re-read a team-mate's share from `volumes` after that share's owner has
    // written its output.  Any other overlap of the two ranges is refused.
    const bool in_place = volumes_out == volumes;
    if (volumes_out != nullptr && !in_place) {
        const uintptr_t a = reinterpret_cast<uintptr_t>(volumes), b = reinterpret_cast<uintptr_t>(volumes_out);
        const uintptr_t bytes = static_cast<uintptr_t>(n_frames) * n_volumes * n_vox * sizeof(float);
        MVX_REQUIRE(a + bytes <= b || b + bytes <= a, MVX_ERR_INVALID_ARGUMENT,
                    "mvx_soft_argmax_3d: volumes_out overlaps volumes (pass the same pointer for in-place use, or disjoint buffers)");
    }
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

    // softmax with the normalised volumes wanted: one kernel, one read of each volume, resident workgroups
    if (softmax && volumes_out != nullptr && !in_place && vec && sz % 4 == 0 &&
        !(kernel_selection() & MVX_SELECT_NO_SOFT_ARGMAX_RESIDENT)) {
        // a share is NV4 float4 per lane of 512 threads: the largest NV4 of 4, 2, 1 that cuts the volume into 1 to 64
        const int64_t lane_span = static_cast<int64_t>(kShareThreads) * 4;
        int nv4 = 0, team = 0;
        for (int cand = 4; cand >= 1 && nv4 == 0; cand >>= 1) {
            const int64_t share = lane_span * cand;
            if (n_vox % share == 0 && n_vox / share <= kTeamMax) {
                nv4 = cand;
                team = static_cast<int>(n_vox / share);
            }
        }
        const int64_t n_vol_total = n_frames * n_volumes;
        const int source = coord.volume != nullptr ? 0 : (coord.has_grid ? 2 : 1);
        const int resident = nv4 != 0 ? resident_workgroups(nv4, source) : 0;
        const int64_t n_shares = n_vol_total * team;
        int64_t grid_size = resident / (team > 0 ? team : 1) * team;  // whole teams only
        if (grid_size > n_shares) grid_size = n_shares;
        // workspace: one record per share, all zero when the kernel starts
        const size_t exchange_bytes = static_cast<size_t>(n_shares) * kRecord * sizeof(double);
        if (nv4 != 0 && grid_size >= team && n_shares < (1LL << 31) && exchange_bytes <= workspace_bytes) {
            double* records = static_cast<double*>(workspace);
            const int help_always = (kernel_selection() & MVX_SELECT_SOFT_ARGMAX_HELP_ALWAYS) ? 1 : 0;
            // cleared by a kernel of our own rather than hipMemsetAsync: captured into a HIP graph, the memset node
            // took effect on the first replay only (the HIP runtime of PyTorch 2.10 for ROCm 7), a kernel node replays like any other
            const size_t n_words = exchange_bytes / sizeof(unsigned long long);
            hipLaunchKernelGGL(clear_records, dim3(static_cast<unsigned>((n_words + kThreads - 1) / kThreads)), dim3(kThreads),
                               0, st, static_cast<unsigned long long*>(workspace), n_words);
            MVX_CHECK_LAUNCH("mvx_soft_argmax_3d (clearing the team records)");
            const dim3 grid(static_cast<unsigned>(grid_size)), block(kShareThreads);
            const FastDiv d_sy = fast_divisor(static_cast<unsigned>(sy)), d_sz = fast_divisor(static_cast<unsigned>(sz));
#define MVX_LAUNCH_RESIDENT_SRC(NV4, SRC)                                                                            \
    hipLaunchKernelGGL((soft_argmax_3d_resident<NV4, SRC>), grid, block, resident_lds_bytes(NV4), st, volumes, coord, n_volumes, \
                       static_cast<unsigned>(n_vox), g, d_sy, d_sz, team, static_cast<unsigned>(n_shares), multiplier, help_always, \
                       records, out, volumes_out)
#define MVX_LAUNCH_RESIDENT(NV4)                               \
    if (source == 0) MVX_LAUNCH_RESIDENT_SRC(NV4, 0);          \
    else if (source == 1) MVX_LAUNCH_RESIDENT_SRC(NV4, 1);     \
    else MVX_LAUNCH_RESIDENT_SRC(NV4, 2)
            switch (nv4) {
                case 1: MVX_LAUNCH_RESIDENT(1); break;
                case 2: MVX_LAUNCH_RESIDENT(2); break;
                default: MVX_LAUNCH_RESIDENT(4); break;
            }
#undef MVX_LAUNCH_RESIDENT
#undef MVX_LAUNCH_RESIDENT_SRC
            MVX_CHECK_LAUNCH("mvx_soft_argmax_3d (resident)");
            return MVX_OK;
        }
    }

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
    // What a previous call left behind needs no clearing here: the three passes rewrite every record and pair before
    // a later pass reads it, and the one-read kernel's records are cleared by the kernel launched ahead of it.
    CoordSource src{};
    src.volume = coords_volume;
    return soft_argmax_3d_any(volumes, src, coords_volume == nullptr, n_frames, n_volumes, sx, sy, sz, multiplier,
                              softmax, out, volumes_out, ws, MVX_SOFT_ARGMAX_3D_STATIC_BYTES, stream);
}
