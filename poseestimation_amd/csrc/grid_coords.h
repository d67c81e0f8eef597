// World position of the voxels of a regular grid, evaluated where it is needed instead of being read from a
// coordinate volume.  The float32 operation order is that of VolumetricTriangulationNet.forward
// (mvn/models/triangulation.py:354-377) and of mvx_build_coord_volumes, so that a kernel fed the grid parameters
// computes on bit-identical positions to one fed the materialised volume:
//     g_k = position_k + step_k * index_k;   d = g - center;   out = rotation . d + center
// (with -ffp-contract=off the product and the sum of g_k stay two operations).
#pragma once

#include "mvx_common.h"

namespace mvx {

// Device pointers to the per-frame parameters, the voxel pitch and the axis transfer flag
// (see mvx_build_coord_volumes in mvx.h).
struct GridSpec {
    const float* position;  // [n_frames][3]
    const float* center;    // [n_frames][3]
    const float* rotation;  // [n_frames][3][3]
    float step0, step1, step2;
    int swap_yz_flip;
};

// The grid of one frame with its parameters in registers (wave-uniform values).
struct FrameGrid {
    float p[3], c[3], r[9];
    float step0, step1, step2;
    int swap_yz_flip, s1;

    __device__ __forceinline__ void load(const GridSpec& g, int64_t b, int s1_) {
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            p[k] = g.position[b * 3 + k];
            c[k] = g.center[b * 3 + k];
        }
#pragma unroll
        for (int k = 0; k < 9; ++k) r[k] = g.rotation[b * 9 + k];
        step0 = g.step0;
        step1 = g.step1;
        step2 = g.step2;
        swap_yz_flip = g.swap_yz_flip;
        s1 = s1_;
    }

    // Position of output voxel (i0, i1, i2).  With the axis transfer the output voxel holds the source voxel
    // (i0, i2, s1 - 1 - i1) (mvn/models/triangulation.py:380-383).
    __device__ __forceinline__ void at(int i0, int i1, int i2, float& x, float& y, float& z) const {
        if (swap_yz_flip) {
            const int src1 = i2, src2 = s1 - 1 - i1;
            i1 = src1;
            i2 = src2;
        }
        const float g0 = p[0] + step0 * static_cast<float>(i0);
        const float g1 = p[1] + step1 * static_cast<float>(i1);
        const float g2 = p[2] + step2 * static_cast<float>(i2);
        const float d0 = g0 - c[0], d1 = g1 - c[1], d2 = g2 - c[2];
        x = fmaf(r[2], d2, fmaf(r[1], d1, r[0] * d0)) + c[0];
        y = fmaf(r[5], d2, fmaf(r[4], d1, r[3] * d0)) + c[1];
        z = fmaf(r[8], d2, fmaf(r[7], d1, r[6] * d0)) + c[2];
    }
};

inline int check_grid_spec(const char* name, const float* position, const float* center, const float* rotation,
                           int swap_yz_flip, int s1, int s2) {
    MVX_REQUIRE(position != nullptr && center != nullptr && rotation != nullptr, MVX_ERR_INVALID_ARGUMENT,
                "%s: null grid parameter pointer", name);
    MVX_REQUIRE(!swap_yz_flip || s1 == s2, MVX_ERR_UNSUPPORTED, "%s: the axis swap needs s1 == s2 (got %d, %d)", name,
                s1, s2);
    return MVX_OK;
}

}  // namespace mvx
