// Temporal post-filter of the real-time pipelines and the axis/joint mapping of their wire format.
//
// A Savitzky-Golay filter is linear: with H = A pinv(A) the hat matrix of the polynomial fit on a window
// (A the Vandermonde matrix of the window positions), the filtered value at window position i is row i of H
// applied to the window.  Away from the ends the window is centred on the sample and the row is the middle one;
// the first and last window/2 samples read rows 0.. and ..window-1 of the fit to the first and last window
// (scipy's mode='interp').  The caller supplies H in float64; sums run in float64 and are rounded once.
#include "mvx_common.h"

namespace mvx {

namespace {

// Filtered value of series s at step t.  x is [n_steps][n_series].
__device__ __forceinline__ float savgol_at(const float* __restrict__ x, int64_t n_steps, int64_t n_series,
                                           const double* __restrict__ hat, int window, int64_t t, int64_t s) {
    const int half = window / 2;
    int row;
    int64_t base;
    if (t < half) {
        row = static_cast<int>(t);
        base = 0;
    } else if (t >= n_steps - half) {
        row = static_cast<int>(t - (n_steps - window));
        base = n_steps - window;
    } else {
        row = half;
        base = t - half;
    }
    const double* h = hat + static_cast<size_t>(row) * window;
    double acc = 0.0;
    for (int i = 0; i < window; ++i) acc = fma(h[i], static_cast<double>(x[(base + i) * n_series + s]), acc);
    return static_cast<float>(acc);
}

__global__ __launch_bounds__(256) void savgol_filter(const float* __restrict__ x, int64_t n_steps, int64_t n_series,
                                                     const double* __restrict__ hat, int window,
                                                     float* __restrict__ out) {
    const int64_t idx = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;
    if (idx >= n_steps * n_series) return;
    const int64_t t = idx / n_series, s = idx - t * n_series;
    out[idx] = savgol_at(x, n_steps, n_series, hat, window, t, s);
}

// (x, y, z) of joint order[j] -> scale * (-x, z, -y)
__device__ __forceinline__ float unity_component(float x, float y, float z, int c, float scale) {
    const float v = c == 0 ? -x : (c == 1 ? z : -y);
    return v * scale;
}

__global__ __launch_bounds__(256) void unity_points(const float* __restrict__ points, const int* __restrict__ order,
                                                    int64_t n_frames, int J, float scale, float* __restrict__ out) {
    const int64_t idx = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;
    if (idx >= n_frames * J * 3) return;
    const int c = static_cast<int>(idx % 3);
    const int64_t fj = idx / 3;
    const int j = static_cast<int>(fj % J);
    const int64_t f = fj / J;
    const float* p = points + (f * J + order[j]) * 3;
    out[idx] = unity_component(p[0], p[1], p[2], c, scale);
}

// One frame of the real-time tail.  `ring` holds the last n_steps poses as a ring of slots [n_steps][J][3]; the new
// pose goes into slot ring_pos (the oldest one), so that time runs t = 0 .. n_steps-1 over slots
// (ring_pos + 1 + t) % n_steps.  filtered == 0: out[0] = the new pose in the wire convention.  Otherwise the window
// is smoothed and out[0] = mean of the last two smoothed poses, out[1] = the last smoothed pose, both in the wire
// convention.  Lanes read the newest step from new_pose itself, never from the slot other lanes are writing.
__global__ __launch_bounds__(64) void realtime_tail(float* __restrict__ ring, const float* __restrict__ new_pose,
                                                    int ring_pos, int n_steps, int J, const double* __restrict__ hat,
                                                    int window, const int* __restrict__ order, float scale, int filtered,
                                                    float* __restrict__ out) {
    const int idx = blockIdx.x * 64 + threadIdx.x;
    if (idx >= J * 3) return;
    const int j = idx / 3, c = idx % 3;
    const int src = order[j];
    const int n_series = J * 3;
    const float mine = new_pose[idx];
    if (!filtered) {
        out[idx] = unity_component(new_pose[src * 3 + 0], new_pose[src * 3 + 1], new_pose[src * 3 + 2], c, scale);
        ring[ring_pos * n_series + idx] = mine;
        return;
    }
    // both outputs lie in the last window: rows window-1 and window-2 of the hat matrix applied to it
    const double* h1 = hat + static_cast<size_t>(window - 1) * window;
    const double* h2 = hat + static_cast<size_t>(window - 2) * window;
    float last[3], mid[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const int s = src * 3 + k;
        double a1 = 0.0, a2 = 0.0;
        for (int i = 0; i < window; ++i) {
            const int t = n_steps - window + i;
            const int slot = (ring_pos + 1 + t) % n_steps;
            const double x = static_cast<double>(t == n_steps - 1 ? new_pose[s] : ring[slot * n_series + s]);
            a1 = fma(h1[i], x, a1);
            a2 = fma(h2[i], x, a2);
        }
        const float y1 = static_cast<float>(a1), y2 = static_cast<float>(a2);
        last[k] = y1;
        mid[k] = (y1 + y2) / 2.0f;
    }
    out[idx] = unity_component(mid[0], mid[1], mid[2], c, scale);
    out[n_series + idx] = unity_component(last[0], last[1], last[2], c, scale);
    ring[ring_pos * n_series + idx] = mine;
}

int check_window(const char* name, int64_t n_steps, int window) {
    MVX_REQUIRE(window >= 1 && window <= n_steps && (window % 2 == 1 || window == n_steps), MVX_ERR_INVALID_ARGUMENT,
                "%s: window=%d must be odd (or equal to n_steps) and at most n_steps=%lld", name, window,
                (long long)n_steps);
    return MVX_OK;
}

}  // namespace

}  // namespace mvx

extern "C" int mvx_savgol_filter(const float* x, int64_t n_steps, int64_t n_series, const double* hat, int window,
                                 float* out, void* stream) {
    using namespace mvx;
    MVX_REQUIRE(n_steps >= 0 && n_series >= 0, MVX_ERR_INVALID_ARGUMENT,
                "mvx_savgol_filter: bad shape n_steps=%lld n_series=%lld", (long long)n_steps, (long long)n_series);
    if (n_steps == 0 || n_series == 0) return MVX_OK;
    const int rc = check_window("mvx_savgol_filter", n_steps, window);
    if (rc != MVX_OK) return rc;
    MVX_REQUIRE(x != nullptr && hat != nullptr && out != nullptr, MVX_ERR_INVALID_ARGUMENT,
                "mvx_savgol_filter: null pointer");
    const int64_t blocks = (n_steps * n_series + 255) / 256;
    MVX_REQUIRE(blocks <= 0x7fffffffLL, MVX_ERR_UNSUPPORTED, "mvx_savgol_filter: too many samples");
    hipLaunchKernelGGL(savgol_filter, dim3(static_cast<unsigned>(blocks)), dim3(256), 0,
                       static_cast<hipStream_t>(stream), x, n_steps, n_series, hat, window, out);
    MVX_CHECK_LAUNCH("mvx_savgol_filter");
    return MVX_OK;
}

extern "C" int mvx_unity_points(const float* points, const int32_t* joint_order, int64_t n_frames, int n_joints,
                                float scale, float* out, void* stream) {
    using namespace mvx;
    MVX_REQUIRE(n_frames >= 0 && n_joints >= 0, MVX_ERR_INVALID_ARGUMENT,
                "mvx_unity_points: bad shape n_frames=%lld n_joints=%d", (long long)n_frames, n_joints);
    if (n_frames == 0 || n_joints == 0) return MVX_OK;
    MVX_REQUIRE(points != nullptr && joint_order != nullptr && out != nullptr, MVX_ERR_INVALID_ARGUMENT,
                "mvx_unity_points: null pointer");
    const int64_t blocks = (n_frames * n_joints * 3 + 255) / 256;
    MVX_REQUIRE(blocks <= 0x7fffffffLL, MVX_ERR_UNSUPPORTED, "mvx_unity_points: too many points");
    hipLaunchKernelGGL(unity_points, dim3(static_cast<unsigned>(blocks)), dim3(256), 0,
                       static_cast<hipStream_t>(stream), points, joint_order, n_frames, n_joints, scale, out);
    MVX_CHECK_LAUNCH("mvx_unity_points");
    return MVX_OK;
}

extern "C" int mvx_realtime_tail(float* ring, const float* new_pose, int ring_pos, int n_steps, int n_joints,
                                 const double* hat, int window, const int32_t* joint_order, float scale, int filtered,
                                 float* out, void* stream) {
    using namespace mvx;
    MVX_REQUIRE(n_steps >= 2 && n_joints > 0 && n_joints <= (1 << 20) && ring_pos >= 0 && ring_pos < n_steps,
                MVX_ERR_INVALID_ARGUMENT, "mvx_realtime_tail: bad shape n_steps=%d (>= 2) n_joints=%d ring_pos=%d", n_steps,
                n_joints, ring_pos);
    if (filtered) {
        const int rc = check_window("mvx_realtime_tail", n_steps, window);
        if (rc != MVX_OK) return rc;
        MVX_REQUIRE(window >= 2 && hat != nullptr, MVX_ERR_INVALID_ARGUMENT,
                    "mvx_realtime_tail: the filtered tail needs a hat matrix and window >= 2 (got %d)", window);
    }
    MVX_REQUIRE(ring != nullptr && new_pose != nullptr && joint_order != nullptr && out != nullptr,
                MVX_ERR_INVALID_ARGUMENT, "mvx_realtime_tail: null pointer");
    hipLaunchKernelGGL(realtime_tail, dim3((n_joints * 3 + 63) / 64), dim3(64), 0, static_cast<hipStream_t>(stream),
                       ring, new_pose, ring_pos, n_steps, n_joints, hat, window, joint_order, scale, filtered, out);
    MVX_CHECK_LAUNCH("mvx_realtime_tail");
    return MVX_OK;
}
