// Development tool: read+write copy of 36 KiB tiles (the 2-D soft-argmax shape: 9 float4 per lane loaded, then
// stored, 2 workgroups per CU) under every combination of the gfx950 cache-policy bits on the loads and stores.
//   hipcc -O3 --offload-arch=gfx950 tools/cache_policy.hip -o tools/cache_policy && tools/cache_policy
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>

typedef float vec4f __attribute__((ext_vector_type(4)));
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1); } } while (0)

#define KERNEL(NAME, LBITS, SBITS)                                                                              \
    __global__ __launch_bounds__(256) void NAME(const vec4f* __restrict__ src, vec4f* __restrict__ dst) {       \
        extern __shared__ float cap[];                                                                          \
        const size_t base = (size_t)blockIdx.x * 256 * 9 + threadIdx.x;                                         \
        vec4f v[9];                                                                                             \
        _Pragma("unroll") for (int k = 0; k < 9; ++k)                                                           \
            asm volatile("global_load_dwordx4 %0, %1, off " LBITS : "=v"(v[k]) : "v"(src + base + k * 256) : "memory"); \
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                                        \
        _Pragma("unroll") for (int k = 0; k < 9; ++k)                                                           \
            asm volatile("global_store_dwordx4 %0, %1, off " SBITS : : "v"(dst + base + k * 256), "v"(v[k]) : "memory"); \
    }

#define ROW(L, LB)                      \
    KERNEL(k_##L##_p, LB, "")           \
    KERNEL(k_##L##_nt, LB, "nt")        \
    KERNEL(k_##L##_sc1, LB, "sc1")      \
    KERNEL(k_##L##_sc01, LB, "sc0 sc1") \
    KERNEL(k_##L##_sc1nt, LB, "sc1 nt") \
    KERNEL(k_##L##_sc01nt, LB, "sc0 sc1 nt")
ROW(p, "")
ROW(nt, "nt")
ROW(sc1, "sc1")
ROW(sc01, "sc0 sc1")
ROW(sc1nt, "sc1 nt")
ROW(sc01nt, "sc0 sc1 nt")

typedef void (*kern_t)(const vec4f*, vec4f*);
struct Entry { const char* name; kern_t k; };
#define E(L, S) {#L "/" #S, k_##L##_##S}
#define ER(L) E(L, p), E(L, nt), E(L, sc1), E(L, sc01), E(L, sc1nt), E(L, sc01nt)
static Entry entries[] = {ER(p), ER(nt), ER(sc1), ER(sc01), ER(sc1nt), ER(sc01nt)};

int main() {
    const size_t n_tiles = 278528;  // 4096 frames x 4 views x 17 joints
    const size_t bytes = n_tiles * 36864;
    vec4f *src, *dst;
    CHECK(hipMalloc(&src, bytes));
    CHECK(hipMalloc(&dst, bytes));
    CHECK(hipMemset(src, 1, bytes));
    CHECK(hipMemset(dst, 0, bytes));
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a));
    CHECK(hipEventCreate(&b));
    const size_t cap_lds = 79 * 1024;  // 2 workgroups per CU
    printf("loads/stores  GB/s (read + written), median of 7\n");
    for (auto& ent : entries) {
        CHECK(hipFuncSetAttribute((const void*)ent.k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)cap_lds));
        std::vector<float> ms;
        for (int r = 0; r < 9; ++r) {
            CHECK(hipEventRecord(a));
            hipLaunchKernelGGL(ent.k, dim3((unsigned)n_tiles), dim3(256), cap_lds, 0, src, dst);
            CHECK(hipEventRecord(b));
            CHECK(hipEventSynchronize(b));
            float t;
            CHECK(hipEventElapsedTime(&t, a, b));
            if (r >= 2) ms.push_back(t);
        }
        std::sort(ms.begin(), ms.end());
        printf("%-14s %7.0f\n", ent.name, 2.0 * bytes / (ms[ms.size() / 2] * 1e-3) / 1e9);
        fflush(stdout);
    }
    // the same copy (nt loads, nt stores) with the destination shifted against the source: do the read and the
    // write stream collide in the DRAM banks when both buffers start on the same alignment?
    printf("dst shift (bytes)  GB/s\n");
    const size_t shifts[] = {0, 256, 1024, 4096, 16384, 65536, 262144, 1 << 20, (1 << 20) + 4096, 3 << 20, 16 << 20};
    vec4f* dst2;
    CHECK(hipMalloc(&dst2, bytes + (32 << 20)));
    CHECK(hipFuncSetAttribute((const void*)k_nt_nt, hipFuncAttributeMaxDynamicSharedMemorySize, (int)cap_lds));
    for (size_t sh : shifts) {
        std::vector<float> ms;
        vec4f* d = reinterpret_cast<vec4f*>(reinterpret_cast<char*>(dst2) + sh);
        for (int r = 0; r < 9; ++r) {
            CHECK(hipEventRecord(a));
            hipLaunchKernelGGL(k_nt_nt, dim3((unsigned)n_tiles), dim3(256), cap_lds, 0, src, d);
            CHECK(hipEventRecord(b));
            CHECK(hipEventSynchronize(b));
            float t;
            CHECK(hipEventElapsedTime(&t, a, b));
            if (r >= 2) ms.push_back(t);
        }
        std::sort(ms.begin(), ms.end());
        printf("%-18zu %7.0f   (src %p dst %p)\n", sh, 2.0 * bytes / (ms[ms.size() / 2] * 1e-3) / 1e9, (void*)src, (void*)d);
    }
    return 0;
}
