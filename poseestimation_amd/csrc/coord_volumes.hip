// Coordinate volumes of the volumetric model: for every frame the world position of each voxel of an
// S0 x S1 x S2 grid spanning a cuboid, optionally rotated about the frame's base point.
//
// Float32 operation order of the reference (mvn/models/triangulation.py:354-377), so that the result is
// bit-identical for the unrotated case:  g_k = pos_k + step_k * idx_k;  d = g - center;  r = rot . d;
// out = r + center.  One lane per voxel, 12 contiguous bytes per lane: a pure write stream of 12 B / voxel.
#include "grid_coords.h"

namespace mvx {

namespace {

__global__ __launch_bounds__(256) void build_coord_volumes(GridSpec spec, int s0, int s1, int s2, int64_t n_vox,
                                                           float* __restrict__ out) {
    const int64_t n = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;
    const int64_t b = blockIdx.y;
    if (n >= n_vox) return;
    const int i2 = static_cast<int>(n % s2);
    const int i1 = static_cast<int>((n / s2) % s1);
    const int i0 = static_cast<int>(n / (static_cast<int64_t>(s2) * s1));
    FrameGrid grid;
    grid.load(spec, b, s1);
    float* o = out + (b * n_vox + n) * 3;
    grid.at(i0, i1, i2, o[0], o[1], o[2]);
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
    const GridSpec spec{position, center, rotation, step0, step1, step2, swap_yz_flip};
    hipLaunchKernelGGL(build_coord_volumes, dim3(static_cast<unsigned>(bx), static_cast<unsigned>(n_frames)), dim3(256),
                       0, static_cast<hipStream_t>(stream), spec, s0, s1, s2, n_vox, out);
    MVX_CHECK_LAUNCH("mvx_build_coord_volumes");
    return MVX_OK;
}
