// Coordinate volumes of the volumetric model: for every frame the world position of each voxel of an
// S0 x S1 x S2 grid spanning a cuboid, optionally rotated about the frame's base point.
//
// Float32 operation order of the reference (mvn/models/triangulation.py:354-377), so that the result is
// bit-identical for the unrotated case:  g_k = pos_k + step_k * idx_k;  d = g - center;  r = rot . d;
// out = r + center.  One lane per voxel, 12 contiguous bytes per lane: a pure write stream of 12 B / voxel.
#include "mvx_common.h"

namespace mvx {

namespace {

__global__ __launch_bounds__(256) void build_coord_volumes(const float* __restrict__ position,
                                                           const float* __restrict__ center,
                                                           const float* __restrict__ rot, int s0, int s1, int s2,
                                                           float step0, float step1, float step2, int64_t n_vox,
                                                           int swap_yz_flip, float* __restrict__ out) {
    const int64_t n = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;
    const int64_t b = blockIdx.y;
    if (n >= n_vox) return;
    int i2 = static_cast<int>(n % s2);
    int i1 = static_cast<int>((n / s2) % s1);
    const int i0 = static_cast<int>(n / (static_cast<int64_t>(s2) * s1));
    if (swap_yz_flip) {
        // the reference permutes axes 1 and 2 of the finished volume and reverses the new axis 1
        // (triangulation.py:380-383): output voxel (i0, i1, i2) holds the source voxel (i0, i2, S1' - 1 - i1),
        // where the source grid is S0 x S2 x S1
        const int src1 = i2, src2 = s1 - 1 - i1;
        i1 = src1;
        i2 = src2;
    }
    const float* p = position + b * 3;
    const float* c = center + b * 3;
    const float* r = rot + b * 9;
    const float g0 = p[0] + step0 * static_cast<float>(i0);
    const float g1 = p[1] + step1 * static_cast<float>(i1);
    const float g2 = p[2] + step2 * static_cast<float>(i2);
    const float d0 = g0 - c[0], d1 = g1 - c[1], d2 = g2 - c[2];
    float* o = out + (b * n_vox + n) * 3;
    o[0] = fmaf(r[2], d2, fmaf(r[1], d1, r[0] * d0)) + c[0];
    o[1] = fmaf(r[5], d2, fmaf(r[4], d1, r[3] * d0)) + c[1];
    o[2] = fmaf(r[8], d2, fmaf(r[7], d1, r[6] * d0)) + c[2];
}

}  // namespace

}  // namespace mvx

extern "C" int mvx_build_coord_volumes(const float* position, const float* center, const float* rotation,
                                       int64_t n_frames, int s0, int s1, int s2, float step0, float step1,
                                       float step2, int swap_yz_flip, float* out, void* stream) {
    using namespace mvx;
    MVX_REQUIRE(n_frames >= 0 && s0 > 0 && s1 > 0 && s2 > 0, MVX_ERR_INVALID_ARGUMENT,
                "mvx_build_coord_volumes: bad shape n_frames=%lld grid=%dx%dx%d", (long long)n_frames, s0, s1, s2);
    if (n_frames == 0) return MVX_OK;
    MVX_REQUIRE(position != nullptr && center != nullptr && rotation != nullptr && out != nullptr,
                MVX_ERR_INVALID_ARGUMENT, "mvx_build_coord_volumes: null pointer argument");
    MVX_REQUIRE(!swap_yz_flip || s1 == s2, MVX_ERR_UNSUPPORTED,
                "mvx_build_coord_volumes: the axis swap needs s1 == s2 (got %d, %d)", s1, s2);
    const int64_t n_vox = static_cast<int64_t>(s0) * s1 * s2;
    const int64_t bx = (n_vox + 255) / 256;
    MVX_REQUIRE(bx <= 0x7fffffffLL && n_frames <= 65535, MVX_ERR_UNSUPPORTED,
                "mvx_build_coord_volumes: grid or batch beyond the launch limits");
    hipLaunchKernelGGL(build_coord_volumes, dim3(static_cast<unsigned>(bx), static_cast<unsigned>(n_frames)), dim3(256),
                       0, static_cast<hipStream_t>(stream), position, center, rotation, s0, s1, s2, step0, step1,
                       step2, n_vox, swap_yz_flip, out);
    MVX_CHECK_LAUNCH("mvx_build_coord_volumes");
    return MVX_OK;
}
