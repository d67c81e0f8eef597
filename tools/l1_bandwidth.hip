// Development tool: bytes per clock a CU's vector-memory path returns for L1-resident 16-byte loads, for the
// access shapes of the unprojection kernel (8 lanes x 16 B = one 128-byte line per voxel, 8 voxels per wave).
//   hipcc -O3 --offload-arch=gfx950 tools/l1_bandwidth.hip -o tools/l1_bandwidth && tools/l1_bandwidth
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

// mode 0: every lane a consecutive float4 (1 KB contiguous per wave instruction)
// mode 1: 8 lanes per 128-byte line, the 8 lines of a wave scattered over a 16 KB window (unprojection taps)
// mode 2: like 1 but every wave instruction touches a single line (all 8 groups the same line)
// mode 3: 16 lanes per 256 contiguous bytes (two adjacent lines), four such pairs scattered per wave instruction
__global__ __launch_bounds__(256) void hammer(const float4* __restrict__ buf, int iters, int mode, float* sink) {
    const int lane = threadIdx.x & 63, q = lane & 7, grp = lane >> 3;
    const float4* base = buf + (size_t)(blockIdx.x & 255) * 1024;  // 16 KB window per CU-ish
    float4 acc = make_float4(0, 0, 0, 0);
    unsigned h = threadIdx.x * 2654435761u + blockIdx.x;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            h = h * 1664525u + 1013904223u;
            unsigned line;
            if (mode == 0) line = 0;
            else if (mode == 1) line = __builtin_amdgcn_readlane(h, 0) * 7u + grp * 13u + k * 5u;
            else line = __builtin_amdgcn_readlane(h, 0) + k;
            const float4* p = mode == 0 ? base + ((k * 64 + lane + i) & 1023) : base + ((line & 127u) * 8 + q);
            if (mode == 3) {
                const unsigned pair = (__builtin_amdgcn_readlane(h, 0) * 7u + (lane >> 4) * 13u + k * 5u) & 63u;
                p = base + pair * 16 + (lane & 15);
            }
            const float4 v = *p;
            acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
        }
    }
    if (acc.x == 12345.f) sink[0] = acc.y + acc.z + acc.w;
}

int main() {
    float4* buf;
    float* sink;
    CHECK(hipMalloc(&buf, 256 * 16384));
    CHECK(hipMemset(buf, 0, 256 * 16384));
    CHECK(hipMalloc(&sink, 4));
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a));
    CHECK(hipEventCreate(&b));
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const double ghz = prop.clockRate / 1e6;
    const int cus = prop.multiProcessorCount;
    for (int mode = 0; mode < 4; ++mode) {
        for (int per_cu : {1, 2, 4, 8}) {
            const int blocks = cus * per_cu, iters = 2000;
            hipLaunchKernelGGL(hammer, dim3(blocks), dim3(256), 0, 0, buf, 10, mode, sink);
            CHECK(hipEventRecord(a));
            hipLaunchKernelGGL(hammer, dim3(blocks), dim3(256), 0, 0, buf, iters, mode, sink);
            CHECK(hipEventRecord(b));
            CHECK(hipEventSynchronize(b));
            float ms;
            CHECK(hipEventElapsedTime(&ms, a, b));
            const double bytes = (double)blocks * 256 * iters * 16 * 16;
            printf("mode %d  %d workgroups/CU: %.1f TB/s chip, %.1f B/clk/CU at %.2f GHz\n", mode, per_cu,
                   bytes / (ms * 1e-3) / 1e12, bytes / (ms * 1e-3) / cus / (ghz * 1e9), ghz);
        }
    }
    return 0;
}
