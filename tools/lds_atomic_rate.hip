// Development tool: throughput of LDS atomic additions on one MI355X, by type and address pattern.
//   hipcc -O3 --offload-arch=gfx950 tools/lds_atomic_rate.hip -o tools/lds_atomic_rate
#include <hip/hip_runtime.h>
#include <stdio.h>

template <typename T, int PATTERN>
__global__ __launch_bounds__(1024) void rate(T* out, int iters) {
    extern __shared__ char raw[];
    T* lds = reinterpret_cast<T*>(raw);
    const int tid = threadIdx.x;
    constexpr int kWords = 131072 / sizeof(T) < 32768 ? 131072 / sizeof(T) : 32768;
    for (int i = tid; i < kWords; i += 1024) lds[i] = T(0);
    __syncthreads();
    // PATTERN 0: lane i -> word i (conflict-free); 1: lane i -> word 4*i (8 banks used); 2: pairs of lanes share a word;
    // 3: all lanes of a wave one word; 4: pseudo-random words
    unsigned a;
    if (PATTERN == 0) a = tid;
    else if (PATTERN == 1) a = (tid * 4) & (kWords - 1);
    else if (PATTERN == 2) a = tid >> 1;
    else if (PATTERN == 3) a = tid >> 6;
    else a = (tid * 2654435761u >> 17) & (kWords - 1);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < 16; ++k) atomicAdd(lds + ((a + k * 1031 + it) & (kWords - 1)), T(1));
    }
    __syncthreads();
    if (tid == 0) out[blockIdx.x] = lds[5];
}

template <typename T, int PATTERN>
void run(const char* what) {
    T* out;
    hipMalloc(&out, 4096 * sizeof(T));
    const int iters = 256, blocks = 256;
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    hipFuncSetAttribute(reinterpret_cast<const void*>(rate<T, PATTERN>), hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
    rate<T, PATTERN><<<blocks, 1024, 131072>>>(out, 4);
    hipEventRecord(a);
    rate<T, PATTERN><<<blocks, 1024, 131072>>>(out, iters);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    const double per_cu = 1024.0 * 16 * iters;  // lane-atomics per workgroup (one per CU)
    printf("%-28s %8.3f ms   %6.2f lane-atomics per ns per CU (%.1f per clock at 2.4 GHz)\n", what, ms, per_cu / (ms * 1e6),
           per_cu / (ms * 1e6) / 2.4);
    hipFree(out);
}

int main() {
    run<float, 0>("float, conflict-free");
    run<float, 1>("float, stride 4 words");
    run<float, 2>("float, pairs share a word");
    run<float, 3>("float, wave shares a word");
    run<float, 4>("float, scattered");
    run<unsigned, 0>("uint, conflict-free");
    run<unsigned, 1>("uint, stride 4 words");
    run<unsigned, 4>("uint, scattered");
    run<unsigned long long, 0>("uint64, conflict-free");
    run<unsigned long long, 1>("uint64, stride 4 words");
    run<unsigned long long, 2>("uint64, pairs share a word");
    run<unsigned long long, 4>("uint64, scattered");
    run<double, 0>("double, conflict-free");
    run<double, 4>("double, scattered");
    run<int, 2>("int, pairs share a word");
    run<int, 3>("int, wave shares a word");
    return 0;
}
