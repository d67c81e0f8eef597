// Development tool: HBM write rate of the unprojection output for two tilings of a [frames][32][64][64][64]
// volume: 128 consecutive voxels per workgroup (512-byte runs per channel) against 4x4x8 voxel tiles
// (32-byte runs per channel; whole 128-byte lines only once the four z-neighbour tiles have all been written).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <algorithm>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1); } } while (0)

__device__ __forceinline__ unsigned xcd_contiguous(unsigned id, unsigned n) {
    const unsigned xcd = id & 7u, k = id >> 3, q = n >> 3, r = n & 7u;
    const unsigned base = xcd < r ? xcd * (q + 1u) : r * (q + 1u) + (xcd - r) * q;
    return base + k;
}

// 128 consecutive voxels: thread -> (channel, group of 4 voxels)
__global__ __launch_bounds__(256) void store_linear(float* __restrict__ vol, int tiles_per_frame, int remap) {
    const unsigned work = remap ? xcd_contiguous(blockIdx.x, gridDim.x) : blockIdx.x;
    const size_t b = work / tiles_per_frame, n0 = (size_t)(work % tiles_per_frame) * 128;
    const size_t n_vox = 262144;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int flat = i * 256 + threadIdx.x;
        const int c = flat / 32, grp = flat % 32;
        *reinterpret_cast<float4*>(vol + (b * 32 + c) * n_vox + n0 + grp * 4) = make_float4(1.f, 2.f, 3.f, (float)c);
    }
}

// 4x4x8 tiles, z-tile fastest in the work order
__global__ __launch_bounds__(256) void store_448(float* __restrict__ vol, int remap) {
    unsigned work = remap ? xcd_contiguous(blockIdx.x, gridDim.x) : blockIdx.x;
    const int tz = work % 8; work /= 8;
    const int ty = work % 16; work /= 16;
    const int tx = work % 16;
    const size_t b = work / 16;
    const size_t n_vox = 262144;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int flat = i * 256 + threadIdx.x;
        const int c = flat / 32, grp = flat % 32;
        const int xy = grp >> 1, zh = grp & 1;
        const size_t n = ((size_t)(tx * 4 + (xy >> 2)) * 64 + (ty * 4 + (xy & 3))) * 64 + tz * 8 + zh * 4;
        *reinterpret_cast<float4*>(vol + (b * 32 + c) * n_vox + n) = make_float4(1.f, 2.f, 3.f, (float)c);
    }
}

template <typename F>
double time_ms(F launch, int reps = 9) {
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    launch(); CHECK(hipDeviceSynchronize());
    std::vector<float> ts;
    for (int r = 0; r < reps; ++r) {
        CHECK(hipEventRecord(a)); launch(); CHECK(hipEventRecord(b)); CHECK(hipEventSynchronize(b));
        float ms; CHECK(hipEventElapsedTime(&ms, a, b)); ts.push_back(ms);
    }
    std::sort(ts.begin(), ts.end());
    return ts[ts.size() / 2];
}

int main() {
    const int frames = 8;
    const size_t bytes = (size_t)frames * 32 * 262144 * 4;
    float* vol;
    CHECK(hipMalloc(&vol, bytes));
    CHECK(hipMemset(vol, 0, bytes));
    const double gb = bytes / 1e9;
    for (int remap = 0; remap < 2; ++remap) {
        double ms = time_ms([&] { store_linear<<<frames * 2048, 256>>>(vol, 2048, remap); });
        printf("128 consecutive voxels, remap=%d: %.3f ms %.0f GB/s\n", remap, ms, gb / (ms * 1e-3));
        ms = time_ms([&] { store_448<<<frames * 2048, 256>>>(vol, remap); });
        printf("4x4x8 tiles,            remap=%d: %.3f ms %.0f GB/s\n", remap, ms, gb / (ms * 1e-3));
    }
    return 0;
}
