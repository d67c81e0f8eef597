// RANSAC triangulation: integer keypoints from the heatmap maxima, then per (frame, joint) a consensus search
// over view pairs, a DLT over the inlier views and, optionally, a Huber-loss refinement of the reprojection error.
//
// heatmap_argmax   one 256-thread workgroup per map, 16-byte non-temporal loads when the map allows them; the
//                  map is read once (h*w*4 bytes per map, HBM-bound) and 16 bytes are written.
// ransac_points    one lane per (frame, joint), everything in float64 registers; the projection matrices and
//                  the keypoints are re-read through the caches whenever a hypothesis needs them.
#include "mvx_common.h"
#include "sym4_eigen.h"

namespace mvx {

namespace {

constexpr int kArgmaxThreads = 256;

// torch.max over a flattened map: NaN beats every number, larger beats smaller, and among equals (or among NaNs)
// the lowest index wins.  The common case carries no NaN, so the scan keeps a strict running maximum (a NaN never
// compares greater) and only notes whether a NaN was seen; a map with a NaN is then searched for its first one.
struct Best {
    float v;
    int i;
};
__device__ __forceinline__ void take(Best& b, bool& nan_seen, float v, int i) {
    nan_seen = nan_seen || (v != v);
    const bool up = v > b.v;
    b.v = up ? v : b.v;
    b.i = up ? i : b.i;
}

// NV > 0: the map is at most NV * 256 float4 (exactly that many with EXACT) and every load is issued before the
// first comparison, slots past the end holding -inf, which never wins; NV == 0: 16-byte loads in a loop;
// NV < 0: scalar loads (map size or alignment rules 16-byte loads out).
template <int NV, bool EXACT>
__global__ __launch_bounds__(kArgmaxThreads) void heatmap_argmax(const float* __restrict__ maps, int hw, int w,
                                                                 float scale_x, float scale_y,
                                                                 long long* __restrict__ out) {
    typedef float float4v __attribute__((ext_vector_type(4)));
    constexpr int kNone = 0x7fffffff;
    const int tid = threadIdx.x;
    const float* map = maps + static_cast<size_t>(blockIdx.x) * hw;
    Best best{-INFINITY, kNone};   // kNone survives only if every element is -inf (index 0 then) or NaN
    bool nan_seen = false;
    if (NV > 0) {
        const float4v* m4 = reinterpret_cast<const float4v*>(map);
        float4v f[NV > 0 ? NV : 1];
        const int n4 = hw >> 2;
#pragma unroll
        for (int k = 0; k < NV; ++k) {
            f[k] = float4v{-INFINITY, -INFINITY, -INFINITY, -INFINITY};
            if (EXACT || k * kArgmaxThreads + tid < n4) f[k] = __builtin_nontemporal_load(m4 + k * kArgmaxThreads + tid);
        }
#pragma unroll
        for (int k = 0; k < NV; ++k) {
            const int i = 4 * (k * kArgmaxThreads + tid);
            take(best, nan_seen, f[k].x, i);
            take(best, nan_seen, f[k].y, i + 1);
            take(best, nan_seen, f[k].z, i + 2);
            take(best, nan_seen, f[k].w, i + 3);
        }
    } else if (NV == 0) {
        const float4v* m4 = reinterpret_cast<const float4v*>(map);
        const int n4 = hw >> 2;
#pragma unroll 4
        for (int i = tid; i < n4; i += kArgmaxThreads) {
            const float4v f = __builtin_nontemporal_load(m4 + i);
            take(best, nan_seen, f.x, 4 * i);
            take(best, nan_seen, f.y, 4 * i + 1);
            take(best, nan_seen, f.z, 4 * i + 2);
            take(best, nan_seen, f.w, 4 * i + 3);
        }
    } else {
#pragma unroll 4
        for (int i = tid; i < hw; i += kArgmaxThreads) take(best, nan_seen, map[i], i);
    }
    __shared__ float wave_v[kArgmaxThreads / 64];
    __shared__ int wave_i[kArgmaxThreads / 64];
    int idx;
    if (__syncthreads_or(nan_seen)) {
        // first NaN of the map (the second read comes from the caches)
        int first = kNone;
        for (int i = tid; i < hw; i += kArgmaxThreads) first = (map[i] != map[i] && i < first) ? i : first;
        first = wave_min_i(first);
        if ((tid & 63) == 0) wave_i[tid >> 6] = first;
        __syncthreads();
        idx = wave_i[0];
#pragma unroll
        for (int k = 1; k < kArgmaxThreads / 64; ++k) idx = min(idx, wave_i[k]);
    } else {
        const float m = wave_max(best.v);
        const int first = wave_min_i(best.v == m ? best.i : kNone);
        if ((tid & 63) == 0) {
            wave_v[tid >> 6] = m;
            wave_i[tid >> 6] = first;
        }
        __syncthreads();
        float bm = wave_v[0];
        idx = wave_i[0];
#pragma unroll
        for (int k = 1; k < kArgmaxThreads / 64; ++k) {
            const bool better = wave_v[k] > bm || (wave_v[k] == bm && wave_i[k] < idx);
            bm = better ? wave_v[k] : bm;
            idx = better ? wave_i[k] : idx;
        }
        idx = idx == kNone ? 0 : idx;
    }
    if (tid == 0) {
        // index -> (x, y), scaled in float32 and truncated, which is what assigning the product of an int64
        // tensor and a Python float into an int64 tensor does
        const float x = static_cast<float>(idx % w) * scale_x;
        const float y = static_cast<float>(idx / w) * scale_y;
        out[static_cast<size_t>(blockIdx.x) * 2 + 0] = static_cast<long long>(x);
        out[static_cast<size_t>(blockIdx.x) * 2 + 1] = static_cast<long long>(y);
    }
}

// ---- RANSAC ----------------------------------------------------------------------------------------------------
struct Views {
    const float* proj;     // [V][12] of this frame
    const double* points;  // (x, y) of view v at points[v * stride]
    int64_t stride;        // doubles between consecutive views
    int n;
};

// DLT over the views of `mask`: rows x*P[2] - P[0], y*P[2] - P[1] in float64 (two roundings each, as numpy
// evaluates them), Gram matrix, eigenvector of the smallest eigenvalue, dehomogenised.
__device__ __forceinline__ void dlt_subset(const Views& vw, unsigned mask, double (&X)[3]) {
    double g[4][4];
#pragma unroll
    for (int p = 0; p < 4; ++p)
#pragma unroll
        for (int q = 0; q < 4; ++q) g[p][q] = 0.0;
    for (int v = 0; v < vw.n; ++v) {
        if (!((mask >> v) & 1u)) continue;
        const float* P = vw.proj + v * 12;
        const double x = vw.points[v * vw.stride], y = vw.points[v * vw.stride + 1];
        double r0[4], r1[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const double p2 = P[8 + k];
            r0[k] = x * p2 - static_cast<double>(P[k]);
            r1[k] = y * p2 - static_cast<double>(P[4 + k]);
        }
#pragma unroll
        for (int p = 0; p < 4; ++p)
#pragma unroll
            for (int q = p; q < 4; ++q) g[p][q] = fma(r1[p], r1[q], fma(r0[p], r0[q], g[p][q]));
    }
#pragma unroll
    for (int p = 1; p < 4; ++p)
#pragma unroll
        for (int q = 0; q < p; ++q) g[p][q] = g[q][p];
    double h[4];
    smallest_eigenvector(g, h);
    X[0] = h[0] / h[3];
    X[1] = h[1] / h[3];
    X[2] = h[2] / h[3];
}

// Image of X in view v and its offset from the keypoint.
__device__ __forceinline__ void reproject(const Views& vw, int v, const double (&X)[3], double& ex, double& ey,
                                          double& px, double& py, double& z) {
    const float* P = vw.proj + v * 12;
    const double u = ((X[0] * P[0] + X[1] * P[1]) + X[2] * P[2]) + static_cast<double>(P[3]);
    const double w = ((X[0] * P[4] + X[1] * P[5]) + X[2] * P[6]) + static_cast<double>(P[7]);
    z = ((X[0] * P[8] + X[1] * P[9]) + X[2] * P[10]) + static_cast<double>(P[11]);
    px = u / z;
    py = w / z;
    ex = vw.points[v * vw.stride] - px;
    ey = vw.points[v * vw.stride + 1] - py;
}

// Half the pixel distance between the keypoint and the reprojection: the reference's "reprojection error".
__device__ __forceinline__ double reprojection_error(const Views& vw, int v, const double (&X)[3]) {
    double ex, ey, px, py, z;
    reproject(vw, v, X, ex, ey, px, py, z);
    return 0.5 * sqrt(ex * ex + ey * ey);
}

// Cost 1/2 sum rho(s_v), s_v = r_v^2, r_v the reprojection error above, rho the Huber loss with unit scale
// (rho(s) = s for s <= 1, 2 sqrt(s) - 1 beyond), with its gradient g = sum w_v J_v^T e_v / 4 and the Gauss-Newton
// Hessian of the robust cost H = sum w_v (J_v^T J_v / 4 - [s_v > 1] q_v q_v^T / (16 s_v)), where e_v = projection
// minus keypoint, r_v = |e_v| / 2, w_v = rho'(s_v), q_v = J_v^T e_v: beyond the quadratic zone the loss is flat
// along the residual's own direction, which the second term removes (H stays positive semi-definite).
struct Normal {
    double cost, h00, h01, h02, h11, h12, h22, g0, g1, g2;
};
__device__ __forceinline__ Normal huber_normal(const Views& vw, unsigned mask, const double (&X)[3]) {
    Normal n{0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
    for (int v = 0; v < vw.n; ++v) {
        if (!((mask >> v) & 1u)) continue;
        const float* P = vw.proj + v * 12;
        double ex, ey, px, py, z;
        reproject(vw, v, X, ex, ey, px, py, z);
        ex = -ex;
        ey = -ey;
        const double s = 0.25 * (ex * ex + ey * ey);
        const double r = sqrt(s);
        const double wgt = (s <= 1.0 ? 1.0 : 1.0 / r) * 0.25;
        n.cost += 0.5 * (s <= 1.0 ? s : 2.0 * r - 1.0);
        const double iz = 1.0 / z;
        double jx[3], jy[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            jx[k] = (static_cast<double>(P[k]) - px * P[8 + k]) * iz;
            jy[k] = (static_cast<double>(P[4 + k]) - py * P[8 + k]) * iz;
        }
        n.h00 += wgt * (jx[0] * jx[0] + jy[0] * jy[0]);
        n.h01 += wgt * (jx[0] * jx[1] + jy[0] * jy[1]);
        n.h02 += wgt * (jx[0] * jx[2] + jy[0] * jy[2]);
        n.h11 += wgt * (jx[1] * jx[1] + jy[1] * jy[1]);
        n.h12 += wgt * (jx[1] * jx[2] + jy[1] * jy[2]);
        n.h22 += wgt * (jx[2] * jx[2] + jy[2] * jy[2]);
        const double q0 = jx[0] * ex + jy[0] * ey, q1 = jx[1] * ex + jy[1] * ey, q2 = jx[2] * ex + jy[2] * ey;
        n.g0 += wgt * q0;
        n.g1 += wgt * q1;
        n.g2 += wgt * q2;
        if (s > 1.0) {
            const double f = wgt / (4.0 * s);
            n.h00 -= f * q0 * q0;
            n.h01 -= f * q0 * q1;
            n.h02 -= f * q0 * q2;
            n.h11 -= f * q1 * q1;
            n.h12 -= f * q1 * q2;
            n.h22 -= f * q2 * q2;
        }
    }
    return n;
}

// Damped Gauss-Newton from the DLT solution to the minimiser of the Huber cost: a step is kept only if it does
// not raise the cost; the loop ends when a step, kept or not, is below 1e-10 relative to 1 + |X| (X in
// millimetres here), when the damping has grown past any useful step, or after kRefineSteps evaluations.
constexpr int kRefineSteps = 100;
__device__ void huber_refine(const Views& vw, unsigned mask, double (&X)[3]) {
    Normal cur = huber_normal(vw, mask, X);
    if (!(cur.cost == cur.cost) || isinf(cur.cost)) return;
    double lam = 1e-6;
    for (int it = 0; it < kRefineSteps; ++it) {
        const double a00 = cur.h00 * (1.0 + lam), a11 = cur.h11 * (1.0 + lam), a22 = cur.h22 * (1.0 + lam);
        const double a01 = cur.h01, a02 = cur.h02, a12 = cur.h12;
        // symmetric 3 x 3 solve by cofactors
        const double c00 = a11 * a22 - a12 * a12, c01 = a02 * a12 - a01 * a22, c02 = a01 * a12 - a02 * a11;
        const double c11 = a00 * a22 - a02 * a02, c12 = a01 * a02 - a00 * a12, c22 = a00 * a11 - a01 * a01;
        const double det = a00 * c00 + a01 * c01 + a02 * c02;
        const double id = -1.0 / det;
        const double d0 = (c00 * cur.g0 + c01 * cur.g1 + c02 * cur.g2) * id;
        const double d1 = (c01 * cur.g0 + c11 * cur.g1 + c12 * cur.g2) * id;
        const double d2 = (c02 * cur.g0 + c12 * cur.g1 + c22 * cur.g2) * id;
        const double cand[3] = {X[0] + d0, X[1] + d1, X[2] + d2};
        const Normal nxt = huber_normal(vw, mask, cand);
        const double step2 = d0 * d0 + d1 * d1 + d2 * d2;
        const double size2 = 1.0 + X[0] * X[0] + X[1] * X[1] + X[2] * X[2];
        if (nxt.cost <= cur.cost) {  // false for NaN
            X[0] = cand[0];
            X[1] = cand[1];
            X[2] = cand[2];
            cur = nxt;
            lam = fmax(lam * 0.2, 1e-15);
            if (step2 <= 1e-20 * size2) break;
        } else {
            if (step2 <= 1e-20 * size2) break;  // rounding noise of the cost decides at this step length
            lam *= 10.0;
            if (lam > 1e12) break;
        }
    }
}

__global__ __launch_bounds__(128) void ransac_points(const float* __restrict__ proj, const double* __restrict__ points,
                                                     const unsigned char* __restrict__ pairs, int64_t n_points, int V,
                                                     int J, int n_iters, double epsilon, int refine,
                                                     double* __restrict__ out, unsigned* __restrict__ inliers) {
    const int64_t idx = static_cast<int64_t>(blockIdx.x) * 128 + threadIdx.x;
    if (idx >= n_points) return;
    const int64_t b = idx / J;
    const int j = static_cast<int>(idx - b * J);
    const Views vw{proj + b * V * 12, points + (b * V * J + j) * 2, static_cast<int64_t>(J) * 2, V};
    const unsigned all = V >= 32 ? 0xffffffffu : ((1u << V) - 1u);

    unsigned best = 0;
    int best_count = 0;
    double X[3];
    for (int it = 0; it < n_iters; ++it) {
        const unsigned v0 = pairs[(idx * n_iters + it) * 2], v1 = pairs[(idx * n_iters + it) * 2 + 1];
        if (v0 >= static_cast<unsigned>(V) || v1 >= static_cast<unsigned>(V)) continue;  // not a pair of views
        unsigned mask = (1u << v0) | (1u << v1);
        dlt_subset(vw, mask, X);
        for (int v = 0; v < V; ++v)
            if (reprojection_error(vw, v, X) < epsilon) mask |= 1u << v;
        const int count = __builtin_popcount(mask);
        if (count > best_count) {
            best_count = count;
            best = mask;
        }
    }
    if (best_count == 0) best = all;
    dlt_subset(vw, best, X);
    if (refine) huber_refine(vw, best, X);
    out[idx * 3 + 0] = X[0];
    out[idx * 3 + 1] = X[1];
    out[idx * 3 + 2] = X[2];
    inliers[idx] = best;
}

}  // namespace

}  // namespace mvx

extern "C" int mvx_heatmap_argmax(const float* heatmaps, int64_t n_maps, int h, int w, float scale_x, float scale_y,
                                  int64_t* keypoints, void* stream) {
    using namespace mvx;
    MVX_REQUIRE(n_maps >= 0 && h > 0 && w > 0 && static_cast<int64_t>(h) * w < 0x7fffffffLL, MVX_ERR_INVALID_ARGUMENT,
                "mvx_heatmap_argmax: bad shape n_maps=%lld h=%d w=%d", (long long)n_maps, h, w);
    if (n_maps == 0) return MVX_OK;
    MVX_REQUIRE(heatmaps != nullptr && keypoints != nullptr, MVX_ERR_INVALID_ARGUMENT,
                "mvx_heatmap_argmax: null pointer");
    MVX_REQUIRE(n_maps <= 0x7fffffffLL, MVX_ERR_UNSUPPORTED, "mvx_heatmap_argmax: %lld maps beyond the launch limit",
                (long long)n_maps);
    const int hw = h * w;
    const bool vec = (hw % 4 == 0) && aligned16(heatmaps);
    hipStream_t st = static_cast<hipStream_t>(stream);
    long long* out = reinterpret_cast<long long*>(keypoints);
    // A read-only stream runs fastest with about 144 KB of maps in flight per CU (4 workgroups of a 96 x 96 map);
    // LDS the kernel does not use is requested only to cap the residency at that.
    const size_t cap_lds = residency_cap_lds(hw, 144 * 1024);
    const dim3 grid(static_cast<unsigned>(n_maps)), block(kArgmaxThreads);
#define MVX_LAUNCH_ARGMAX(NV, EXACT) \
    hipLaunchKernelGGL((heatmap_argmax<NV, EXACT>), grid, block, cap_lds, st, heatmaps, hw, w, scale_x, scale_y, out)
    const int n4 = hw / 4;
    if (!vec) MVX_LAUNCH_ARGMAX(-1, false);
    else if (n4 == 256 * 9) MVX_LAUNCH_ARGMAX(9, true);    // 96 x 96
    else if (n4 == 256 * 4) MVX_LAUNCH_ARGMAX(4, true);    // 64 x 64
    else if (n4 == 256 * 16) MVX_LAUNCH_ARGMAX(16, true);  // 128 x 128
    else if (n4 <= 256 * 2) MVX_LAUNCH_ARGMAX(2, false);
    else if (n4 <= 256 * 4) MVX_LAUNCH_ARGMAX(4, false);
    else if (n4 <= 256 * 6) MVX_LAUNCH_ARGMAX(6, false);
    else if (n4 <= 256 * 9) MVX_LAUNCH_ARGMAX(9, false);
    else if (n4 <= 256 * 12) MVX_LAUNCH_ARGMAX(12, false);
    else if (n4 <= 256 * 16) MVX_LAUNCH_ARGMAX(16, false);
    else MVX_LAUNCH_ARGMAX(0, false);
#undef MVX_LAUNCH_ARGMAX
    MVX_CHECK_LAUNCH("mvx_heatmap_argmax");
    return MVX_OK;
}

extern "C" int mvx_triangulate_ransac(const float* proj, const double* points, const uint8_t* view_pairs,
                                      int64_t n_frames, int n_views, int n_joints, int n_iters, double epsilon,
                                      int direct_optimization, double* out, uint32_t* inlier_masks, void* stream) {
    using namespace mvx;
    MVX_REQUIRE(n_frames >= 0 && n_views >= 2 && n_views <= 32 && n_joints >= 0 && n_iters >= 0,
                MVX_ERR_INVALID_ARGUMENT,
                "mvx_triangulate_ransac: bad shape n_frames=%lld n_views=%d (2..32) n_joints=%d n_iters=%d",
                (long long)n_frames, n_views, n_joints, n_iters);
    const int64_t n_points = n_frames * n_joints;
    if (n_points == 0) return MVX_OK;
    MVX_REQUIRE(proj != nullptr && points != nullptr && out != nullptr && inlier_masks != nullptr &&
                    (n_iters == 0 || view_pairs != nullptr),
                MVX_ERR_INVALID_ARGUMENT, "mvx_triangulate_ransac: null pointer");
    const int64_t blocks = (n_points + 127) / 128;
    MVX_REQUIRE(blocks <= 0x7fffffffLL, MVX_ERR_UNSUPPORTED, "mvx_triangulate_ransac: too many points");
    hipLaunchKernelGGL(ransac_points, dim3(static_cast<unsigned>(blocks)), dim3(128), 0,
                       static_cast<hipStream_t>(stream), proj, points, view_pairs, n_points, n_views, n_joints,
                       n_iters, epsilon, direct_optimization, out, inlier_masks);
    MVX_CHECK_LAUNCH("mvx_triangulate_ransac");
    return MVX_OK;
}
