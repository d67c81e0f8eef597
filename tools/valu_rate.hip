// Development tool: cycles per wave64 VALU instruction on one SIMD as a function of resident waves.
//   hipcc --offload-arch=gfx950 -O3 tools/valu_rate.hip -o tools/valu_rate
#include <hip/hip_runtime.h>
#include <stdio.h>

template <int KIND>
__global__ void loop(float* out, int iters, long long* cycles) {
    float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7};
    double d0 = a0, d1 = a1, d2 = a2, d3 = a3;
    const float m = 1.0001f, c = 0.5f;
    long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
        if (KIND == 0) {
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                             "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m), "v"(c));
            }
        } else if (KIND == 1) {
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                asm volatile("v_pk_fma_f32 %0, %0, %4, %5\n v_pk_fma_f32 %1, %1, %4, %5\n v_pk_fma_f32 %2, %2, %4, %5\n v_pk_fma_f32 %3, %3, %4, %5\n"
                             "v_pk_fma_f32 %0, %0, %4, %5\n v_pk_fma_f32 %1, %1, %4, %5\n v_pk_fma_f32 %2, %2, %4, %5\n v_pk_fma_f32 %3, %3, %4, %5\n"
                             : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(p0), "v"(p1));
            }
        } else if (KIND == 2) {
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                asm volatile("v_fma_f64 %0, %0, %4, %5\n v_fma_f64 %1, %1, %4, %5\n v_fma_f64 %2, %2, %4, %5\n v_fma_f64 %3, %3, %4, %5\n"
                             "v_fma_f64 %0, %0, %4, %5\n v_fma_f64 %1, %1, %4, %5\n v_fma_f64 %2, %2, %4, %5\n v_fma_f64 %3, %3, %4, %5\n"
                             : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(d0), "v"(d1));
            }
        } else {
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                asm volatile("v_exp_f32 %0, %0\n v_exp_f32 %1, %1\n v_exp_f32 %2, %2\n v_exp_f32 %3, %3\n"
                             "v_exp_f32 %4, %4\n v_exp_f32 %5, %5\n v_exp_f32 %6, %6\n v_exp_f32 %7, %7\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
            }
        }
    }
    long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0.x + p1.x + p2.y + p3.x + (float)(d0 + d1 + d2 + d3);
    if (threadIdx.x == 0 && blockIdx.x == 0) *cycles = t1 - t0;
}

int main() {
    float* out;
    long long* cyc;
    hipMalloc(&out, 256 * 2048 * 4);
    hipMalloc(&cyc, 8);
    const char* names[4] = {"v_fma_f32", "v_pk_fma_f32", "v_fma_f64", "v_exp_f32"};
    for (int kind = 0; kind < 4; ++kind) {
        for (int waves_per_simd = 1; waves_per_simd <= 8; waves_per_simd *= 2) {
            const int threads = 256 * waves_per_simd > 1024 ? 1024 : 256 * waves_per_simd;
            const int blocks_per_cu = (256 * waves_per_simd) / threads;
            const int iters = 2000;
            long long h = 0;
            hipEvent_t e0, e1;
            hipEventCreate(&e0);
            hipEventCreate(&e1);
            for (int rep = 0; rep < 2; ++rep) {
                hipEventRecord(e0, 0);
                if (kind == 0) hipLaunchKernelGGL(loop<0>, dim3(256 * blocks_per_cu), dim3(threads), 0, 0, out, iters, cyc);
                if (kind == 1) hipLaunchKernelGGL(loop<1>, dim3(256 * blocks_per_cu), dim3(threads), 0, 0, out, iters, cyc);
                if (kind == 2) hipLaunchKernelGGL(loop<2>, dim3(256 * blocks_per_cu), dim3(threads), 0, 0, out, iters, cyc);
                if (kind == 3) hipLaunchKernelGGL(loop<3>, dim3(256 * blocks_per_cu), dim3(threads), 0, 0, out, iters, cyc);
                hipEventRecord(e1, 0);
                hipDeviceSynchronize();
            }
            float ms = 0;
            hipEventElapsedTime(&ms, e0, e1);
            hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
            printf("  [wall %.3f ms: %.2f ns of SIMD time per wave-instruction] ", ms, ms * 1e6 / (iters * 64.0 * waves_per_simd));
            const double per_wave_instr = (double)h / (iters * 64.0);
            printf("%-13s %d waves/SIMD: %.2f cycles per instruction per wave -> %.2f cycles of SIMD time per wave-instruction\n",
                   names[kind], waves_per_simd, per_wave_instr, per_wave_instr / waves_per_simd);
        }
    }
    return 0;
}
