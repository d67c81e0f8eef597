// Device helpers shared by the unprojection kernels and their backward kernels: where a voxel lies, and the
// bilinear taps of its projection into one view.  One definition, so that the forward and backward kernels sample
// exactly the same positions with exactly the same weights.
#pragma once

#include "grid_coords.h"

namespace mvx {

namespace {

struct Tap {
    float w00, w01, w10, w11;  // weights of (yb, xb), (yb, xb+1), (yb+1, xb), (yb+1, xb+1)
    unsigned pixel;            // yb * W + xb, always inside the map
};

constexpr float kLog2e = 1.44269504088896340736f;
constexpr float kLn2 = 0.69314718055994530942f;

// Where the world position of a voxel comes from: a coordinate volume [frames of this launch][N][3], or the
// parameters of a regular grid evaluated on the spot (grid_coords.h), frame index relative to `first_frame`.
struct VoxelSource {
    const float* volume;
    GridSpec grid;
    int has_grid;
    int64_t first_frame;  // frame of the batch that frame 0 of this launch is
    int vy, vz;           // grid dimensions 1 and 2
};
// Position of voxel n (flat index) of frame b of the launch.
__device__ __forceinline__ void voxel_position(const VoxelSource& src, int64_t b, int64_t n_vox, int64_t n, double& X,
                                               double& Y, double& Z) {
    if (src.has_grid) {
        const unsigned nu = static_cast<unsigned>(n);
        const unsigned row = nu / static_cast<unsigned>(src.vz);
        const int i2 = static_cast<int>(nu - row * static_cast<unsigned>(src.vz));
        const int i0 = static_cast<int>(row / static_cast<unsigned>(src.vy)), i1 = static_cast<int>(row) - i0 * src.vy;
        FrameGrid fg;
        fg.load(src.grid, src.first_frame + b, src.vy);
        float x, y, z;
        fg.at(i0, i1, i2, x, y, z);
        X = x;
        Y = y;
        Z = z;
    } else {
        const float* cp = src.volume + (b * n_vox + n) * 3;
        X = cp[0];
        Y = cp[1];
        Z = cp[2];
    }
}

// Projects one voxel and returns its bilinear taps with zero padding.
// The reference maps the projected pixel (x, y) = (u/z, v/z) through gx = 2(x/H - 0.5), gy = 2(y/W - 0.5)
// (H for x and W for y) and grid_sample(align_corners=True): ix = (gx + 1)/2 * (W - 1) = x/H * (W - 1),
// iy = y/W * (H - 1).  Here the camera Pd is float64 with rows 0 and 1 already multiplied by (W-1)/H and
// (H-1)/W, so ix = u'/z, iy = v'/z; the position is evaluated in float64 and only the interpolation weights
// are rounded to float32 (the reference's float32 position carries about 1e-5 px of rounding noise, which the
// feature gradient would multiply).  1/z is v_rcp_f32 (1 ulp of float32) refined by one Newton step in float64
// (relative error below 1e-14: under 1e-12 pixels) instead of a float64 division.  A voxel with z <= 0 gets all-zero weights; `wscale`
// multiplies the weights (log2 e for the softmax aggregation, which works in base 2).
struct TapXY {
    float w00, w01, w10, w11;  // as in Tap
    int xb, yb;                // top-left pixel of the 2 x 2 taps, inside the map: 0 <= xb <= W-2, 0 <= yb <= H-2
    int x0, y0;                // floor of the sampling position, clamped to [-2, W] x [-2, H]
    bool behind;               // z <= 0: all weights are zero
};

__device__ __forceinline__ TapXY project_tap_xy(const double* __restrict__ Pd, double X, double Y, double Z, int H,
                                                int W, float wscale) {
    const double u = fma(X, Pd[0], fma(Y, Pd[1], fma(Z, Pd[2], Pd[3])));
    const double v = fma(X, Pd[4], fma(Y, Pd[5], fma(Z, Pd[6], Pd[7])));
    const double z0 = fma(X, Pd[8], fma(Y, Pd[9], fma(Z, Pd[10], Pd[11])));
    const bool behind = z0 <= 0.0;
    const double z = (z0 == 0.0) ? 1.0 : z0;
    double rz = static_cast<double>(__builtin_amdgcn_rcpf(static_cast<float>(z)));
    rz = fma(rz, fma(-z, rz, 1.0), rz);
    const double ixd = u * rz, iyd = v * rz;
    const double x0d = floor(ixd), y0d = floor(iyd);
    // the fractions are rounded to float32 once; the complementary weights are 1 - fraction in float32
    const float wx1 = static_cast<float>(ixd - x0d), wy1 = static_cast<float>(iyd - y0d);
    const float wx0 = 1.0f - wx1, wy0 = 1.0f - wy1;
    const float x0f = static_cast<float>(x0d), y0f = static_cast<float>(y0d);
    // far-away or non-finite positions, and voxels behind the camera, become "outside" (-2)
    const bool usable = (fabsf(x0f) < 1.0e6f) && (fabsf(y0f) < 1.0e6f) && !behind;
    const int x0 = usable ? static_cast<int>(fminf(fmaxf(x0f, -2.0f), static_cast<float>(W))) : -2;
    const int y0 = usable ? static_cast<int>(fminf(fmaxf(y0f, -2.0f), static_cast<float>(H))) : -2;
    // The taps are columns x0 and x0 + 1 with weights wx0 and wx1, zero outside the map.  They are read as
    // columns xb and xb + 1 with xb = clamp(x0, 0, W - 2): the same columns when x0 lies in [0, W - 2]; for
    // x0 = -1 column xb = 0 is tap x0 + 1, for x0 = W - 1 column xb + 1 is tap x0.  Likewise for rows.
    const bool in_x = static_cast<unsigned>(x0) <= static_cast<unsigned>(W - 2);
    const bool in_y = static_cast<unsigned>(y0) <= static_cast<unsigned>(H - 2);
    const float cxa = in_x ? wx0 : (x0 == -1 ? wx1 : 0.0f);
    const float cxb = in_x ? wx1 : (x0 == W - 1 ? wx0 : 0.0f);
    const float cya = (in_y ? wy0 : (y0 == -1 ? wy1 : 0.0f)) * wscale;
    const float cyb = (in_y ? wy1 : (y0 == H - 1 ? wy0 : 0.0f)) * wscale;
    TapXY t;
    t.w00 = cxa * cya;
    t.w01 = cxb * cya;
    t.w10 = cxa * cyb;
    t.w11 = cxb * cyb;
    t.xb = min(max(x0, 0), W - 2);
    t.yb = min(max(y0, 0), H - 2);
    t.x0 = x0;
    t.y0 = y0;
    t.behind = behind;
    return t;
}

__device__ __forceinline__ Tap project_tap(const double* __restrict__ Pd, double X, double Y, double Z, int H, int W,
                                           float wscale) {
    const TapXY x = project_tap_xy(Pd, X, Y, Z, H, W, wscale);
    Tap t;
    t.w00 = x.w00;
    t.w01 = x.w01;
    t.w10 = x.w10;
    t.w11 = x.w11;
    t.pixel = static_cast<unsigned>(x.yb * W + x.xb);
    return t;
}

// Camera row r of a float32 3x4 matrix as float64, scaled for project_tap.
__device__ __forceinline__ double camera_entry(const float* __restrict__ P, int i, double sx, double sy) {
    const int r = i >> 2;
    return static_cast<double>(P[i]) * (r == 0 ? sx : (r == 1 ? sy : 1.0));
}

}  // namespace

}  // namespace mvx
