// Development tool: resident workgroups per CU the runtime reports for the unprojection kernels.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/occupancy_probe.hip poseestimation_amd/csrc/mvx_core.hip -o tools/occupancy_probe
#include "../poseestimation_amd/csrc/unproject.hip"

int main() {
    using namespace mvx;
    int n = -1;
    const size_t brick_smem = 2 * kPatchBytes + sizeof(BrickShared);
    hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, unproject_brick32<MVX_AGG_SOFTMAX, true, 4>, kBrickThreads, brick_smem);
    printf("unproject_brick32<softmax, grid, 4 views>: %d workgroups of %d threads per CU with %zu bytes of LDS (%s)\n", n, kBrickThreads,
           brick_smem, hipGetErrorString(e));
    e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, unproject_brick32<MVX_AGG_SUM, true, 4>, kBrickThreads, brick_smem);
    printf("unproject_brick32<sum, grid, 4 views>: %d (%s)\n", n, hipGetErrorString(e));
    hipFuncAttributes a;
    e = hipFuncGetAttributes(&a, reinterpret_cast<const void*>(unproject_brick32<MVX_AGG_SOFTMAX, true, 4>));
    printf("softmax: regs %d, static LDS %zu, scratch %zu, max dynamic LDS %d (%s)\n", a.numRegs, a.sharedSizeBytes,
           a.localSizeBytes, a.maxDynamicSharedSizeBytes, hipGetErrorString(e));
    return 0;
}
