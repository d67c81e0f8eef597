// Development tool: measured HBM ceilings on the box for the access shapes the kernels use.
//   hipcc -O3 --offload-arch=gfx950 tools/copy_ceiling.hip -o tools/copy_ceiling && tools/copy_ceiling
// Prints GB/s (read + written bytes) for float4 copy / read-only / write-only streams over a 10 GiB buffer,
// for several grid shapes and cache policies.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <algorithm>

typedef float vec4f __attribute__((ext_vector_type(4)));

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1); } } while (0)

template <bool NTL, bool NTS>
__global__ __launch_bounds__(256) void copy_gridstride(const vec4f* __restrict__ src, vec4f* __restrict__ dst, size_t n) {
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * 256;
    for (; i < n; i += stride) {
        vec4f v = NTL ? __builtin_nontemporal_load(src + i) : src[i];
        if (NTS) __builtin_nontemporal_store(v, dst + i); else dst[i] = v;
    }
}

// one block per 36 KiB tile, 9 loads in flight per lane, then 9 stores (the soft-argmax shape)
template <bool NTL, bool NTS, int NV>
__global__ __launch_bounds__(256) void copy_tile(const vec4f* __restrict__ src, vec4f* __restrict__ dst) {
    const size_t base = (size_t)blockIdx.x * 256 * NV + threadIdx.x;
    vec4f v[NV];
#pragma unroll
    for (int k = 0; k < NV; ++k) v[k] = NTL ? __builtin_nontemporal_load(src + base + k * 256) : src[base + k * 256];
#pragma unroll
    for (int k = 0; k < NV; ++k) { if (NTS) __builtin_nontemporal_store(v[k], dst + base + k * 256); else dst[base + k * 256] = v[k]; }
}

// persistent: each block walks tiles with a stride of gridDim.x
template <bool NTL, bool NTS, int NV>
__global__ __launch_bounds__(256) void copy_tile_persistent(const vec4f* __restrict__ src, vec4f* __restrict__ dst, size_t n_tiles) {
    for (size_t t = blockIdx.x; t < n_tiles; t += gridDim.x) {
        const size_t base = t * 256 * NV + threadIdx.x;
        vec4f v[NV];
#pragma unroll
        for (int k = 0; k < NV; ++k) v[k] = NTL ? __builtin_nontemporal_load(src + base + k * 256) : src[base + k * 256];
#pragma unroll
        for (int k = 0; k < NV; ++k) { if (NTS) __builtin_nontemporal_store(v[k], dst + base + k * 256); else dst[base + k * 256] = v[k]; }
    }
}

// persistent and software-pipelined: the loads of the next tile are issued before the stores of the current one
template <bool NTL, bool NTS, int NV>
__global__ __launch_bounds__(256) void copy_tile_pipelined(const vec4f* __restrict__ src, vec4f* __restrict__ dst, size_t n_tiles) {
    size_t t = blockIdx.x;
    if (t >= n_tiles) return;
    vec4f cur[NV], nxt[NV];
    {
        const size_t base = t * 256 * NV + threadIdx.x;
#pragma unroll
        for (int k = 0; k < NV; ++k) cur[k] = NTL ? __builtin_nontemporal_load(src + base + k * 256) : src[base + k * 256];
    }
    for (; t < n_tiles; t += gridDim.x) {
        const size_t tn = t + gridDim.x;
        if (tn < n_tiles) {
            const size_t base = tn * 256 * NV + threadIdx.x;
#pragma unroll
            for (int k = 0; k < NV; ++k) nxt[k] = NTL ? __builtin_nontemporal_load(src + base + k * 256) : src[base + k * 256];
        }
        const size_t base = t * 256 * NV + threadIdx.x;
#pragma unroll
        for (int k = 0; k < NV; ++k) { if (NTS) __builtin_nontemporal_store(cur[k], dst + base + k * 256); else dst[base + k * 256] = cur[k]; }
#pragma unroll
        for (int k = 0; k < NV; ++k) cur[k] = nxt[k];
    }
}

// occupancy-limited tile copy: dynamic LDS reserves space so that only a few blocks fit on a CU
template <bool NTL, bool NTS, int NV>
__global__ __launch_bounds__(256) void copy_tile_lds(const vec4f* __restrict__ src, vec4f* __restrict__ dst) {
    extern __shared__ float pad[];
    if (threadIdx.x == 9999) pad[0] = 0.f;
    const size_t base = (size_t)blockIdx.x * 256 * NV + threadIdx.x;
    vec4f v[NV];
#pragma unroll
    for (int k = 0; k < NV; ++k) v[k] = NTL ? __builtin_nontemporal_load(src + base + k * 256) : src[base + k * 256];
#pragma unroll
    for (int k = 0; k < NV; ++k) { if (NTS) __builtin_nontemporal_store(v[k], dst + base + k * 256); else dst[base + k * 256] = v[k]; }
}

template <bool NT>
__global__ __launch_bounds__(256) void read_only(const vec4f* __restrict__ src, float* __restrict__ sink, size_t n) {
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * 256;
    vec4f acc = {0, 0, 0, 0};
    for (; i < n; i += stride) acc += NT ? __builtin_nontemporal_load(src + i) : src[i];
    if (acc.x + acc.y + acc.z + acc.w == 12345.678f) sink[0] = acc.x;
}

template <bool NT>
__global__ __launch_bounds__(256) void write_only(vec4f* __restrict__ dst, size_t n) {
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * 256;
    const vec4f v = {1.f, 2.f, 3.f, 4.f};
    for (; i < n; i += stride) { if (NT) __builtin_nontemporal_store(v, dst + i); else dst[i] = v; }
}

template <typename F>
double time_ms(F launch, int reps = 7) {
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
    const size_t n_tiles = 278528;                 // 4096 frames x 68 maps
    const size_t n = n_tiles * 256 * 9;            // float4 elements (10.27 GB)
    const size_t bytes = n * 16;
    vec4f *src, *dst; float* sink;
    CHECK(hipMalloc(&src, bytes)); CHECK(hipMalloc(&dst, bytes)); CHECK(hipMalloc(&sink, 4));
    CHECK(hipMemset(src, 1, bytes)); CHECK(hipMemset(dst, 0, bytes));
    const double gb = bytes / 1e9;
    auto rep = [&](const char* name, double ms, double factor) { printf("%-52s %8.3f ms  %8.1f GB/s\n", name, ms, factor * gb / (ms * 1e-3)); };

    for (int blocks : {2048, 4096, 8192, 65536}) {
        char nm[128];
        snprintf(nm, sizeof nm, "copy grid-stride %d blocks", blocks);
        rep(nm, time_ms([&] { copy_gridstride<false, false><<<blocks, 256>>>(src, dst, n); }), 2);
        snprintf(nm, sizeof nm, "copy grid-stride %d blocks nt-load nt-store", blocks);
        rep(nm, time_ms([&] { copy_gridstride<true, true><<<blocks, 256>>>(src, dst, n); }), 2);
    }
    rep("copy tile 36KiB/block", time_ms([&] { copy_tile<false, false, 9><<<n_tiles, 256>>>(src, dst); }), 2);
    rep("copy tile 36KiB/block nt-store", time_ms([&] { copy_tile<false, true, 9><<<n_tiles, 256>>>(src, dst); }), 2);
    rep("copy tile 36KiB/block nt-load nt-store", time_ms([&] { copy_tile<true, true, 9><<<n_tiles, 256>>>(src, dst); }), 2);
    rep("copy tile 12KiB/block nt nt", time_ms([&] { copy_tile<true, true, 3><<<n_tiles * 3, 256>>>(src, dst); }), 2);
    rep("copy tile 4KiB/block nt nt", time_ms([&] { copy_tile<true, true, 1><<<n_tiles * 9, 256>>>(src, dst); }), 2);
    for (int blocks : {1024, 2048, 4096}) {
        char nm[128];
        snprintf(nm, sizeof nm, "copy tile persistent %d blocks nt nt", blocks);
        rep(nm, time_ms([&] { copy_tile_persistent<true, true, 9><<<blocks, 256>>>(src, dst, n_tiles); }), 2);
    }
    for (int blocks : {512, 1024, 2048}) {
        char nm[128];
        snprintf(nm, sizeof nm, "copy tile pipelined %d blocks nt nt", blocks);
        rep(nm, time_ms([&] { copy_tile_pipelined<true, true, 9><<<blocks, 256>>>(src, dst, n_tiles); }), 2);
    }
    for (int kb : {20, 40, 80}) {
        char nm[128];
        snprintf(nm, sizeof nm, "copy tile 36KiB/block nt nt, %d KiB LDS/block", kb);
        rep(nm, time_ms([&] { copy_tile_lds<true, true, 9><<<n_tiles, 256, kb * 1024>>>(src, dst); }), 2);
    }
    rep("copy tile 8KiB/block nt nt", time_ms([&] { copy_tile<true, true, 2><<<n_tiles * 9 / 2, 256>>>(src, dst); }), 2);
    rep("read-only grid-stride 4096 blocks", time_ms([&] { read_only<false><<<4096, 256>>>(src, sink, n); }), 1);
    rep("read-only grid-stride 4096 blocks nt", time_ms([&] { read_only<true><<<4096, 256>>>(src, sink, n); }), 1);
    rep("write-only grid-stride 4096 blocks", time_ms([&] { write_only<false><<<4096, 256>>>(dst, n); }), 1);
    rep("write-only grid-stride 4096 blocks nt", time_ms([&] { write_only<true><<<4096, 256>>>(dst, n); }), 1);
    rep("hipMemcpyAsync d2d", time_ms([&] { CHECK(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, 0)); }), 2);
    return 0;
}
