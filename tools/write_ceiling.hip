// Development tool: write-only streams of the unprojection's output size (64 frames x 32 channels x 64^3 floats =
// 2.15 GB), 16 bytes per lane, with the access shapes the brick kernel could use.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/write_ceiling.hip -o tools/write_ceiling
#include <hip/hip_runtime.h>
#include <stdio.h>

#include <algorithm>
#include <vector>

// every workgroup of 512 threads writes 64 KB
// MODE 0: 64 KB contiguous per workgroup (1 KB per wave instruction)
// MODE 1: the brick kernel's pattern: per wave instruction 4 channel planes x 8 y-rows x 32 bytes
template <int MODE>
__global__ __launch_bounds__(512) void write_only(float4* out, float v) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const size_t n_vox = 64 * 64 * 64;
    const unsigned work = blockIdx.x;
    const unsigned frame = work / 512, brick = work % 512;
    const unsigned bz = brick & 7, by = (brick >> 3) & 7, bx = brick >> 6;
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        size_t idx;
        if (MODE == 0) {
            idx = static_cast<size_t>(work) * 4096 + (q * 8 + wave) * 64 + lane;
        } else {
            // lane -> (channel lane & 3 of quad q, y-row (lane >> 3), z-half (lane >> 2) & 1), x = wave
            const unsigned c = q * 4 + (lane & 3), y = by * 8 + (lane >> 3), z = bz * 8 + ((lane >> 2) & 1) * 4, x = bx * 8 + wave;
            idx = ((static_cast<size_t>(frame) * 32 + c) * n_vox + (static_cast<size_t>(x) * 64 + y) * 64 + z) / 4;
        }
        out[idx] = make_float4(v, v, v, v);
    }
}

template <class F>
static void time_it(const char* what, double bytes, F&& launch) {
    for (int i = 0; i < 30; ++i) launch();
    hipDeviceSynchronize();
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    std::vector<float> t;
    for (int i = 0; i < 30; ++i) {
        hipEventRecord(e0, 0);
        launch();
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        t.push_back(ms);
    }
    std::sort(t.begin(), t.end());
    printf("%-60s median %.3f ms  %.2f TB/s\n", what, t[15], bytes / (t[15] * 1e-3) / 1e12);
}

int main() {
    const size_t n = static_cast<size_t>(64) * 32 * 64 * 64 * 64;  // floats
    float4* out;
    hipMalloc(&out, n * 4);
    const double bytes = static_cast<double>(n) * 4;
    for (int lds : {0, 66000}) {   // 66 KB of LDS per workgroup: two workgroups per CU, as the brick kernel
        printf("dynamic LDS %d bytes\n", lds);
        time_it("contiguous 64 KB per workgroup", bytes, [&] { hipLaunchKernelGGL(write_only<0>, dim3(32768), dim3(512), lds, 0, out, 1.0f); });
        time_it("brick pattern (32-byte runs, 32 channel planes)", bytes, [&] { hipLaunchKernelGGL(write_only<1>, dim3(32768), dim3(512), lds, 0, out, 1.0f); });
    }
    return 0;
}
