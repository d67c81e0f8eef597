// Development tool: what launching the brick kernel's grid costs by itself.  Kernels that do nothing, with the brick
// kernel's launch shape (32 768 workgroups of 512 threads, 64.6 KB of dynamic LDS, 128 VGPRs) and variations of it.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/dispatch_probe.hip -o tools/dispatch_probe
#include <hip/hip_runtime.h>
#include <stdio.h>

#include <algorithm>
#include <vector>

template <int REGS>
__global__ __launch_bounds__(512, 4) void empty_kernel(float* out, int never) {
    extern __shared__ unsigned char smem[];
    if (REGS > 64) asm volatile("v_mov_b32 v120, 0" ::: "v120");  // raises the kernel's VGPR count to 128
    if (never) {
        smem[threadIdx.x] = 1;
        __syncthreads();
        out[threadIdx.x] = smem[threadIdx.x ^ 1];
    }
}

// touches LDS and one barrier, stores 16 bytes per lane: the minimum a real workgroup does
__global__ __launch_bounds__(512, 4) void tiny_kernel(float4* out) {
    extern __shared__ unsigned char smem[];
    asm volatile("v_mov_b32 v120, 0" ::: "v120");
    reinterpret_cast<float*>(smem)[threadIdx.x] = threadIdx.x;
    __syncthreads();
    const float v = reinterpret_cast<float*>(smem)[threadIdx.x ^ 1];
    out[static_cast<size_t>(blockIdx.x) * 512 + threadIdx.x] = make_float4(v, v, v, v);
}

template <class F>
static void time_it(const char* what, F&& launch) {
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
    printf("%-70s median %.3f ms (min %.3f)\n", what, t[15], t[0]);
}

int main() {
    float* out;
    hipMalloc(&out, static_cast<size_t>(32768) * 512 * 16);
    const int lds = 2 * 32768 + 616;
    time_it("empty, 32768 x 512 threads, 64.6 KB LDS, 128 VGPRs", [&] { hipLaunchKernelGGL(empty_kernel<128>, dim3(32768), dim3(512), lds, 0, out, 0); });
    time_it("empty, 32768 x 512 threads, 64.6 KB LDS, few VGPRs", [&] { hipLaunchKernelGGL(empty_kernel<8>, dim3(32768), dim3(512), lds, 0, out, 0); });
    time_it("empty, 32768 x 512 threads, no LDS, 128 VGPRs", [&] { hipLaunchKernelGGL(empty_kernel<128>, dim3(32768), dim3(512), 0, 0, out, 0); });
    time_it("empty, 32768 x 512 threads, no LDS, few VGPRs", [&] { hipLaunchKernelGGL(empty_kernel<8>, dim3(32768), dim3(512), 0, 0, out, 0); });
    time_it("empty, 65536 x 256 threads, 32 KB LDS, 128 VGPRs", [&] { hipLaunchKernelGGL(empty_kernel<128>, dim3(65536), dim3(256), 32768, 0, out, 0); });
    time_it("LDS touch + barrier + 16-byte store, 32768 x 512, 64.6 KB LDS, 128 VGPRs",
            [&] { hipLaunchKernelGGL(tiny_kernel, dim3(32768), dim3(512), lds, 0, reinterpret_cast<float4*>(out)); });
    return 0;
}
