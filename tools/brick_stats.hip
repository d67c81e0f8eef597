// Development tool: runs mvx_unproject on the benchmark's synthetic geometry and reports how many 8 x 8 x 8 bricks
// leave the staged (LDS patch) path.  Inputs come from a file written by tools/brick_stats.py.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -DMVX_BRICK_STATS tools/brick_stats.hip poseestimation_amd/csrc/mvx_core.hip -o tools/brick_stats
#include <vector>

#include "../poseestimation_amd/csrc/unproject.hip"

int main(int argc, char** argv) {
    const char* path = argc > 1 ? argv[1] : "gpurun_out/brick_case.bin";
    FILE* f = fopen(path, "rb");
    if (!f) { printf("cannot open %s\n", path); return 1; }
    int hdr[4];  // frames, S, H, W
    if (fread(hdr, 4, 4, f) != 4) return 1;
    const int B = hdr[0], S = hdr[1], H = hdr[2], W = hdr[3];
    std::vector<float> proj(B * 48), coord(static_cast<size_t>(B) * S * S * S * 3);
    if (fread(proj.data(), 4, proj.size(), f) != proj.size() || fread(coord.data(), 4, coord.size(), f) != coord.size()) return 1;
    fclose(f);
    float *dproj, *dcoord, *dfeat, *dvol;
    hipMalloc(&dproj, proj.size() * 4);
    hipMalloc(&dcoord, coord.size() * 4);
    hipMalloc(&dfeat, static_cast<size_t>(B) * 4 * 32 * H * W * 4);
    hipMalloc(&dvol, static_cast<size_t>(B) * 32 * S * S * S * 4);
    hipMemset(dfeat, 0, static_cast<size_t>(B) * 4 * 32 * H * W * 4);
    hipMemcpy(dproj, proj.data(), proj.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(dcoord, coord.data(), coord.size() * 4, hipMemcpyHostToDevice);
#ifdef MVX_BRICK_STATS
    unsigned zero[4] = {0, 0, 0, 0};
    hipMemcpyToSymbol(HIP_SYMBOL(mvx::brick_stats), zero, sizeof(zero));
#endif
    // features: something other than zeros, so that the arithmetic is not trivial
    hipMemset(dfeat, 0x3c, static_cast<size_t>(B) * 4 * 32 * H * W * 4);
    for (int rep = 0; rep < 2; ++rep)
        mvx_unproject_channels_last(dfeat, dproj, dcoord, nullptr, B, 4, 32, H, W, S, S, S, MVX_AGG_SOFTMAX, dvol, nullptr);
    hipDeviceSynchronize();
    {
        hipEvent_t e0, e1;
        hipEventCreate(&e0);
        hipEventCreate(&e1);
        hipEventRecord(e0, 0);
        for (int rep = 0; rep < 5; ++rep)
            mvx_unproject_channels_last(dfeat, dproj, dcoord, nullptr, B, 4, 32, H, W, S, S, S, MVX_AGG_SOFTMAX, dvol, nullptr);
        hipEventRecord(e1, 0);
        hipDeviceSynchronize();
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        printf("%.3f ms per launch of %d frames (%.3f ms per 64 frames)\n", ms / 5, B, ms / 5 * 64 / B);
    }
#ifndef MVX_BRICK_STATS
    return 0;
#else
    hipMemcpyToSymbol(HIP_SYMBOL(mvx::brick_stats), zero, sizeof(zero));
    int rc = mvx_unproject_channels_last(dfeat, dproj, dcoord, nullptr, B, 4, 32, H, W, S, S, S, MVX_AGG_SOFTMAX, dvol, nullptr);
    hipDeviceSynchronize();
    unsigned st[4];
    hipMemcpyFromSymbol(st, HIP_SYMBOL(mvx::brick_stats), sizeof(st));
    // (per-phase times: tools/brick_lab.hip built with -DMVX_BRICK_TRACE)
    printf("rc=%d bricks=%u direct=%u (%.2f%%) not-fitting=%u max rows=%u\n", rc, st[0], st[1], 100.0 * st[1] / (st[0] ? st[0] : 1), st[2], st[3]);
    return 0;
#endif
}
