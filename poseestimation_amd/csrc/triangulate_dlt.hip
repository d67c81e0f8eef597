// Confidence-weighted DLT triangulation: one lane per (frame, joint).
//
// The 2V x 4 design matrix A is formed row by row in float32 with the reference's roundings
// (A = fl(fl(fl(P[2]*x) - P[0:2]) * c)), its Gram matrix A^T A (10 unique entries) is accumulated in float64
// registers (products of two float32 values are exact in float64), and the eigenvector of the smallest
// eigenvalue -- the right singular vector of A's smallest singular value -- comes from cyclic Jacobi
// rotations on the 4x4 symmetric matrix, all in registers with compile-time indices (sweeps repeat until every
// off-diagonal entry is below 1e-18 of the geometric mean of its diagonal neighbours).
// HBM traffic per point: 48*V (P, shared by the joints of a frame) + 8*V (points) + 4*V (conf) + 12 bytes out.
#include "mvx_common.h"
#include "sym4_eigen.h"

namespace mvx {

namespace {

constexpr int kThreads = 128;
constexpr int64_t kRowsMaxPoints = 2048;  // up to here the 16-lanes-per-point kernel is used (V <= 16)

// GLUE: points are heatmap-pixel keypoints and conf the raw confidences; the kernel normalises the
// confidences over the views (+1e-5), scales the keypoints to image pixels, stores both, and triangulates
// from the stored float32 values.
template <bool GLUE>
__global__ __launch_bounds__(kThreads) void triangulate_dlt(const float* __restrict__ proj,
                                                            const float* __restrict__ points,
                                                            const float* __restrict__ conf, int64_t n_points, int V,
                                                            int J, float scale_x, float scale_y,
                                                            float* __restrict__ points_out,
                                                            float* __restrict__ conf_out, float* __restrict__ out) {
    const int64_t idx = static_cast<int64_t>(blockIdx.x) * kThreads + threadIdx.x;
    if (idx >= n_points) return;
    const int64_t b = idx / J;
    const int j = static_cast<int>(idx - b * J);

    float view_sum = 1.0f;
    if (GLUE) {
        double csum = 0.0;
        for (int v = 0; v < V; ++v) csum += static_cast<double>(conf[(b * V + v) * J + j]);
        view_sum = static_cast<float>(csum);  // the exact view sum rounded once; divided by below
    }

    double g[4][4];
#pragma unroll
    for (int p = 0; p < 4; ++p)
#pragma unroll
        for (int q = 0; q < 4; ++q) g[p][q] = 0.0;

    for (int v = 0; v < V; ++v) {
        const float* P = proj + (b * V + v) * 12;
        const int64_t pj = (b * V + v) * J + j;
        float x = points[pj * 2 + 0], y = points[pj * 2 + 1];
        float c = conf != nullptr ? conf[pj] : 1.0f;
        if (GLUE) {
            c = c / view_sum + 1e-5f;
            x = x * scale_x;
            y = y * scale_y;
            conf_out[pj] = c;
            points_out[pj * 2 + 0] = x;
            points_out[pj * 2 + 1] = y;
        }
        double r0[4], r1[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            // three separately rounded float32 operations (the library is built with -ffp-contract=off)
            const float p2 = P[8 + k];
            const float a0 = (p2 * x - P[k]) * c;
            const float a1 = (p2 * y - P[4 + k]) * c;
            r0[k] = a0;
            r1[k] = a1;
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
    out[idx * 3 + 0] = static_cast<float>(h[0] / h[3]);
    out[idx * 3 + 1] = static_cast<float>(h[1] / h[3]);
    out[idx * 3 + 2] = static_cast<float>(h[2] / h[3]);
}

// The same for small batches (a real-time stream triangulates 17 points per frame): 16 lanes per point, lane v
// handling view v (V <= 16).  The confidence sum and the Gram matrix are reduced across the 16 lanes by DPP row
// operations instead of a serial loop over the views, which is where a one-lane-per-point launch of 17 points spends
// most of its time.  The Gram entries are the same exact float64 products summed in another order, so a result can
// differ from the lane-per-point kernel's in the last float32 bit.
template <typename T>
__device__ __forceinline__ T row16_sum(T v) {
    v += dpp_mov<0xB1>(v);
    v += dpp_mov<0x4E>(v);
    v += dpp_mov<0x141>(v);
    v += dpp_mov<0x140>(v);
    return v;  // every lane of the 16-lane row holds the row's sum
}

template <bool GLUE>
__global__ __launch_bounds__(256) void triangulate_dlt_rows(const float* __restrict__ proj, const float* __restrict__ points,
                                                            const float* __restrict__ conf, int64_t n_points, int V, int J,
                                                            float scale_x, float scale_y, float* __restrict__ points_out,
                                                            float* __restrict__ conf_out, float* __restrict__ out) {
    const int v = threadIdx.x & 15;
    const int64_t idx = static_cast<int64_t>(blockIdx.x) * 16 + (threadIdx.x >> 4);
    // rows past the end and lanes past the last view compute on zeros (no early exit: the row reductions need
    // every lane) and store nothing
    const bool live = idx < n_points && v < V;
    const int64_t b = idx < n_points ? idx / J : 0;
    const int j = idx < n_points ? static_cast<int>(idx - b * J) : 0;
    const int64_t pj = (b * V + (v < V ? v : 0)) * J + j;
    const float* P = proj + (b * V + (v < V ? v : 0)) * 12;

    float x = points[pj * 2 + 0], y = points[pj * 2 + 1];
    float c = conf != nullptr ? conf[pj] : 1.0f;
    if (GLUE) {
        const double csum = row16_sum(live ? static_cast<double>(c) : 0.0);  // exact: float32 terms in float64
        const float view_sum = static_cast<float>(csum);
        c = c / view_sum + 1e-5f;
        x = x * scale_x;
        y = y * scale_y;
        if (live) {
            conf_out[pj] = c;
            points_out[pj * 2 + 0] = x;
            points_out[pj * 2 + 1] = y;
        }
    }
    double r0[4], r1[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        // three separately rounded float32 operations (the library is built with -ffp-contract=off)
        const float p2 = P[8 + k];
        const float a0 = (p2 * x - P[k]) * c;
        const float a1 = (p2 * y - P[4 + k]) * c;
        r0[k] = live ? a0 : 0.0f;
        r1[k] = live ? a1 : 0.0f;
    }
    double g[4][4];
#pragma unroll
    for (int p = 0; p < 4; ++p)
#pragma unroll
        for (int q = p; q < 4; ++q) g[p][q] = row16_sum(fma(r1[p], r1[q], r0[p] * r0[q]));
#pragma unroll
    for (int p = 1; p < 4; ++p)
#pragma unroll
        for (int q = 0; q < p; ++q) g[p][q] = g[q][p];

    double h[4];
    smallest_eigenvector(g, h);  // every lane of the row solves the same matrix; lane 0 stores
    if (v == 0 && idx < n_points) {
        out[idx * 3 + 0] = static_cast<float>(h[0] / h[3]);
        out[idx * 3 + 1] = static_cast<float>(h[1] / h[3]);
        out[idx * 3 + 2] = static_cast<float>(h[2] / h[3]);
    }
}

__global__ __launch_bounds__(256) void project_points(const float* __restrict__ proj, const float* __restrict__ pts,
                                                       int64_t n, int to_euclidean, float* __restrict__ out) {
    const int64_t i = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;
    if (i >= n) return;
    const float x = pts[i * 3 + 0], y = pts[i * 3 + 1], z = pts[i * 3 + 2];
    float r[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) r[k] = fmaf(x, proj[k * 4 + 0], fmaf(y, proj[k * 4 + 1], fmaf(z, proj[k * 4 + 2], proj[k * 4 + 3])));
    if (to_euclidean) {
        out[i * 2 + 0] = r[0] / r[2];
        out[i * 2 + 1] = r[1] / r[2];
    } else {
        out[i * 3 + 0] = r[0];
        out[i * 3 + 1] = r[1];
        out[i * 3 + 2] = r[2];
    }
}

int launch_dlt(bool glue, const float* proj, const float* points, const float* conf, int64_t n_frames, int n_views,
               int n_joints, float sx, float sy, float* points_out, float* conf_out, float* out, void* stream,
               const char* name) {
    MVX_REQUIRE(n_frames >= 0 && n_views > 0 && n_joints >= 0, MVX_ERR_INVALID_ARGUMENT,
                "%s: bad shape n_frames=%lld n_views=%d n_joints=%d", name, (long long)n_frames, n_views, n_joints);
    const int64_t n_points = n_frames * n_joints;
    if (n_points == 0) return MVX_OK;
    MVX_REQUIRE(proj != nullptr && points != nullptr && out != nullptr, MVX_ERR_INVALID_ARGUMENT,
                "%s: null proj, points or out pointer", name);
    if (glue)
        MVX_REQUIRE(conf != nullptr && points_out != nullptr && conf_out != nullptr, MVX_ERR_INVALID_ARGUMENT,
                    "%s: null conf_raw, points_img_out or conf_norm_out pointer", name);
    const int64_t blocks = (n_points + kThreads - 1) / kThreads;
    MVX_REQUIRE(blocks <= 0x7fffffffLL, MVX_ERR_UNSUPPORTED, "%s: %lld points beyond the launch limit", name,
                (long long)n_points);
    hipStream_t st = static_cast<hipStream_t>(stream);
    // few points (a real-time stream): 16 lanes per point, views in parallel; one lane per point cannot fill the
    // chip below a few thousand points anyway
    if (n_views <= 16 && n_points <= kRowsMaxPoints && !(kernel_selection() & MVX_SELECT_NO_DLT_ROWS)) {
        const dim3 grid(static_cast<unsigned>((n_points + 15) / 16));
        if (glue)
            hipLaunchKernelGGL(triangulate_dlt_rows<true>, grid, dim3(256), 0, st, proj, points, conf, n_points, n_views,
                               n_joints, sx, sy, points_out, conf_out, out);
        else
            hipLaunchKernelGGL(triangulate_dlt_rows<false>, grid, dim3(256), 0, st, proj, points, conf, n_points, n_views,
                               n_joints, 1.0f, 1.0f, nullptr, nullptr, out);
        MVX_CHECK_LAUNCH(name);
        return MVX_OK;
    }
    if (glue)
        hipLaunchKernelGGL(triangulate_dlt<true>, dim3(static_cast<unsigned>(blocks)), dim3(kThreads), 0, st, proj,
                           points, conf, n_points, n_views, n_joints, sx, sy, points_out, conf_out, out);
    else
        hipLaunchKernelGGL(triangulate_dlt<false>, dim3(static_cast<unsigned>(blocks)), dim3(kThreads), 0, st, proj,
                           points, conf, n_points, n_views, n_joints, 1.0f, 1.0f, nullptr, nullptr, out);
    MVX_CHECK_LAUNCH(name);
    return MVX_OK;
}

}  // namespace

}  // namespace mvx

extern "C" int mvx_triangulate_dlt(const float* proj, const float* points, const float* conf, int64_t n_frames,
                                   int n_views, int n_joints, float* out, void* stream) {
    return mvx::launch_dlt(false, proj, points, conf, n_frames, n_views, n_joints, 1.0f, 1.0f, nullptr, nullptr, out,
                           stream, "mvx_triangulate_dlt");
}

extern "C" int mvx_algebraic_glue_dlt(const float* proj, const float* points_hm, const float* conf_raw,
                                      int64_t n_frames, int n_views, int n_joints, float scale_x, float scale_y,
                                      float* points_img_out, float* conf_norm_out, float* out, void* stream) {
    return mvx::launch_dlt(true, proj, points_hm, conf_raw, n_frames, n_views, n_joints, scale_x, scale_y,
                           points_img_out, conf_norm_out, out, stream, "mvx_algebraic_glue_dlt");
}

extern "C" int mvx_project_points(const float* proj, const float* points, int64_t n_points, int to_euclidean,
                                  float* out, void* stream) {
    using namespace mvx;
    MVX_REQUIRE(n_points >= 0, MVX_ERR_INVALID_ARGUMENT, "mvx_project_points: n_points=%lld", (long long)n_points);
    if (n_points == 0) return MVX_OK;
    MVX_REQUIRE(proj != nullptr && points != nullptr && out != nullptr, MVX_ERR_INVALID_ARGUMENT,
                "mvx_project_points: null pointer");
    const int64_t blocks = (n_points + 255) / 256;
    MVX_REQUIRE(blocks <= 0x7fffffffLL, MVX_ERR_UNSUPPORTED, "mvx_project_points: too many points");
    hipLaunchKernelGGL(project_points, dim3(static_cast<unsigned>(blocks)), dim3(256), 0,
                       static_cast<hipStream_t>(stream), proj, points, n_points, to_euclidean, out);
    MVX_CHECK_LAUNCH("mvx_project_points");
    return MVX_OK;
}

extern "C" int mvx_algebraic_forward(const float* heatmaps, const float* conf_raw, const float* proj, int64_t n_frames,
                                     int n_views, int n_joints, int h, int w, float multiplier, int softmax,
                                     float scale_x, float scale_y, float* heatmaps_out, float* points_img_out,
                                     float* conf_norm_out, float* out, float* points_hm_workspace, void* stream) {
    using namespace mvx;
    MVX_REQUIRE(n_frames >= 0 && n_views > 0 && n_joints >= 0, MVX_ERR_INVALID_ARGUMENT,
                "mvx_algebraic_forward: bad shape n_frames=%lld n_views=%d n_joints=%d", (long long)n_frames, n_views,
                n_joints);
    if (n_frames == 0 || n_joints == 0) return MVX_OK;
    MVX_REQUIRE(points_hm_workspace != nullptr, MVX_ERR_INVALID_ARGUMENT,
                "mvx_algebraic_forward: null points_hm_workspace");
    const int rc = mvx_soft_argmax_2d(heatmaps, n_frames * n_views * n_joints, h, w, multiplier, softmax,
                                      points_hm_workspace, heatmaps_out, stream);
    if (rc != MVX_OK) return rc;
    return mvx_algebraic_glue_dlt(proj, points_hm_workspace, conf_raw, n_frames, n_views, n_joints, scale_x, scale_y,
                                  points_img_out, conf_norm_out, out, stream);
}
