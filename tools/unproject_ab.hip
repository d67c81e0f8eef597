// Development tool: times mvx_unproject_grid of several builds of the library on the benchmark geometry (64 frames,
// 4 views, 32 channels, 64^3 voxels, channel-last maps, grid parameters) in one process, interleaved, each after its
// own warm-up launches (the clocks ramp up over the first ~15 launches after an idle spell).
//   hipcc --offload-arch=gfx950 -O2 -std=c++17 tools/unproject_ab.hip -o tools/unproject_ab -ldl
//   tools/unproject_ab <case from tools/brick_lab.py> <aggregation 0..3, or -1 for mvx_soft_argmax_3d_grid> <libmvx.so> [<libmvx.so> ...]
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include <algorithm>
#include <vector>

typedef int (*soft_argmax_3d_grid_fn)(const float*, const float*, const float*, const float*, float, float, float, int, int64_t,
                                      int, int, int, int, float, int, float*, float*, void*, size_t, void*);
typedef size_t (*workspace_fn)(int64_t, int, int, int, int);
typedef int (*unproject_grid_fn)(const float*, int, const float*, const float*, const float*, const float*, float, float, float,
                                 int, const float*, int64_t, int, int, int, int, int, int, int, int, float*, void*, size_t,
                                 void*);

__global__ void fill_normalish(float* p, size_t n, unsigned seed) {
    size_t i = blockIdx.x * static_cast<size_t>(blockDim.x) + threadIdx.x;
    if (i >= n) return;
    unsigned long long x = (i + 1) * 0x9E3779B97F4A7C15ull + seed;
    float s = 0.0f;
    for (int k = 0; k < 4; ++k) {
        x ^= x >> 29;
        x *= 0xBF58476D1CE4E5B9ull;
        x ^= x >> 32;
        s += static_cast<float>(x & 0xffffff) * (1.0f / 16777216.0f);
    }
    p[i] = (s - 2.0f) * 1.7320508f;
}

int main(int argc, char** argv) {
    if (argc < 4) return 1;
    FILE* f = fopen(argv[1], "rb");
    if (!f) return 1;
    int B = 0;
    if (fread(&B, 4, 1, f) != 1) return 1;
    const int S = 64, H = 96, W = 96, agg = atoi(argv[2]);
    std::vector<float> proj(B * 48), base(B * 3), pos(B * 3), rot(B * 9, 0.0f);
    if (fread(proj.data(), 4, proj.size(), f) != proj.size() || fread(base.data(), 4, base.size(), f) != base.size()) return 1;
    fclose(f);
    for (int b = 0; b < B; ++b) {
        for (int k = 0; k < 3; ++k) pos[b * 3 + k] = base[b * 3 + k] - 1250.0f;
        rot[b * 9 + 0] = rot[b * 9 + 4] = rot[b * 9 + 8] = 1.0f;
    }
    const float step = static_cast<float>(2500.0 / 63.0);
    const size_t nfeat = static_cast<size_t>(B) * 4 * 32 * H * W, nvol = static_cast<size_t>(B) * 32 * S * S * S;
    float *d_proj, *d_pos, *d_center, *d_rot, *d_feat, *d_vol, *d_conf;
    hipMalloc(&d_proj, proj.size() * 4);
    hipMalloc(&d_pos, pos.size() * 4);
    hipMalloc(&d_center, base.size() * 4);
    hipMalloc(&d_rot, rot.size() * 4);
    hipMalloc(&d_feat, nfeat * 4);
    hipMalloc(&d_vol, nvol * 4);
    hipMalloc(&d_conf, static_cast<size_t>(B) * 4 * 32 * 4);
    hipMemcpy(d_proj, proj.data(), proj.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(d_pos, pos.data(), pos.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(d_center, base.data(), base.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(d_rot, rot.data(), rot.size() * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(fill_normalish, dim3((nfeat + 255) / 256), dim3(256), 0, 0, d_feat, nfeat, 17u);
    hipLaunchKernelGGL(fill_normalish, dim3((B * 128 + 255) / 256), dim3(256), 0, 0, d_conf, static_cast<size_t>(B) * 128, 5u);
    hipDeviceSynchronize();

    if (agg < 0) {
        // aggregation -1: mvx_soft_argmax_3d_grid (17 volumes of 64^3 per frame, softmax, normalised volumes written)
        const size_t nv = static_cast<size_t>(B) * 17 * S * S * S;
        float *d_in, *d_out, *d_kp;
        void* d_ws;
        hipMalloc(&d_in, nv * 4);
        hipMalloc(&d_out, nv * 4);
        hipMalloc(&d_kp, static_cast<size_t>(B) * 17 * 3 * 4);
        hipLaunchKernelGGL(fill_normalish, dim3((nv + 255) / 256), dim3(256), 0, 0, d_in, nv, 23u);
        hipDeviceSynchronize();
        hipEvent_t e0, e1;
        hipEventCreate(&e0);
        hipEventCreate(&e1);
        std::vector<soft_argmax_3d_grid_fn> fs;
        size_t ws_bytes = 0;
        for (int i = 3; i < argc; ++i) {
            void* h = dlopen(argv[i], RTLD_NOW | RTLD_LOCAL);
            if (!h) { printf("cannot load %s: %s\n", argv[i], dlerror()); return 1; }
            fs.push_back(reinterpret_cast<soft_argmax_3d_grid_fn>(dlsym(h, "mvx_soft_argmax_3d_grid")));
            ws_bytes = reinterpret_cast<workspace_fn>(dlsym(h, "mvx_soft_argmax_3d_workspace_bytes"))(B, 17, S, S, S);
        }
        hipMalloc(&d_ws, ws_bytes);
        for (int round = 0; round < 3; ++round) {
            for (size_t k = 0; k < fs.size(); ++k) {
                auto launch = [&] {
                    return fs[k](d_in, d_pos, d_center, d_rot, step, step, step, 0, B, 17, S, S, S, 1.0f, 1, d_kp, d_out, d_ws, ws_bytes,
                                 nullptr);
                };
                for (int i = 0; i < 40; ++i)
                    if (launch() != 0) { printf("call failed\n"); return 1; }
                if (hipDeviceSynchronize() != hipSuccess) { printf("kernel failed\n"); return 1; }
                std::vector<float> t;
                for (int i = 0; i < 50; ++i) {
                    hipEventRecord(e0, 0);
                    launch();
                    hipEventRecord(e1, 0);
                    hipEventSynchronize(e1);
                    float ms = 0;
                    hipEventElapsedTime(&ms, e0, e1);
                    t.push_back(ms);
                }
                hipEventRecord(e0, 0);
                for (int i = 0; i < 50; ++i) launch();
                hipEventRecord(e1, 0);
                hipEventSynchronize(e1);
                float burst = 0;
                hipEventElapsedTime(&burst, e0, e1);
                std::sort(t.begin(), t.end());
                printf("round %d  soft-argmax 3-D  %-36s  min %.3f  median %.3f  max %.3f ms   back to back %.3f ms\n", round,
                       argv[3 + k], t.front(), t[t.size() / 2], t.back(), burst / 50);
                fflush(stdout);
            }
        }
        return 0;
    }
    std::vector<unproject_grid_fn> fns;
    for (int i = 3; i < argc; ++i) {
        void* h = dlopen(argv[i], RTLD_NOW | RTLD_LOCAL);
        if (!h) { printf("cannot load %s: %s\n", argv[i], dlerror()); return 1; }
        fns.push_back(reinterpret_cast<unproject_grid_fn>(dlsym(h, "mvx_unproject_grid")));
        if (!fns.back()) return 1;
    }
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int round = 0; round < 3; ++round) {
        for (size_t k = 0; k < fns.size(); ++k) {
            auto launch = [&] {
                return fns[k](d_feat, 1, d_proj, d_pos, d_center, d_rot, step, step, step, 0, d_conf, B, 4, 32, H, W, S, S, S, agg,
                              d_vol, nullptr, 0, nullptr);
            };
            for (int i = 0; i < 40; ++i)
                if (launch() != 0) { printf("call failed\n"); return 1; }
            if (hipDeviceSynchronize() != hipSuccess) { printf("kernel failed\n"); return 1; }
            std::vector<float> t;
            for (int i = 0; i < 50; ++i) {
                hipEventRecord(e0, 0);
                launch();
                hipEventRecord(e1, 0);
                hipEventSynchronize(e1);
                float ms = 0;
                hipEventElapsedTime(&ms, e0, e1);
                t.push_back(ms);
            }
            // and 50 launches back to back, one interval
            hipEventRecord(e0, 0);
            for (int i = 0; i < 50; ++i) launch();
            hipEventRecord(e1, 0);
            hipEventSynchronize(e1);
            float burst = 0;
            hipEventElapsedTime(&burst, e0, e1);
            std::sort(t.begin(), t.end());
            printf("round %d  %-40s  min %.3f  median %.3f  max %.3f ms   back to back %.3f ms\n", round, argv[3 + k], t.front(),
                   t[t.size() / 2], t.back(), burst / 50);
            fflush(stdout);
        }
    }
    return 0;
}
