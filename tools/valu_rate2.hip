// Development tool: SIMD time per wave64 instruction for the float64 / conversion / integer-multiply instructions the
// soft-argmax kernels lean on (4 waves per SIMD, 8 independent chains per wave).
//   hipcc --offload-arch=gfx950 -O3 tools/valu_rate2.hip -o tools/valu_rate2
#include <hip/hip_runtime.h>
#include <stdio.h>

#define KERNEL(NAME, TEXT, CD, CS)                                                                                \
    __global__ void NAME(float* out, int iters) {                                                                 \
        double d0 = threadIdx.x, d1 = d0 + 1, d2 = d0 + 2, d3 = d0 + 3;                                           \
        float f0 = threadIdx.x, f1 = f0 + 1, f2 = f0 + 2, f3 = f0 + 3;                                            \
        unsigned u0 = threadIdx.x, u1 = u0 + 1, u2 = u0 + 2, u3 = u0 + 3;                                         \
        for (int i = 0; i < iters; ++i) {                                                                         \
            _Pragma("unroll") for (int k = 0; k < 16; ++k) {                                                      \
                asm volatile(TEXT : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3), \
                                    "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3));                                      \
            }                                                                                                     \
        }                                                                                                         \
        out[blockIdx.x * blockDim.x + threadIdx.x] = (float)(d0 + d1 + d2 + d3) + f0 + f1 + f2 + f3 + u0 + u1 + u2 + u3; \
    }

// operands: %0-%3 doubles, %4-%7 floats, %8-%11 unsigned
KERNEL(k_add_f64, "v_add_f64 %0, %0, %1\n v_add_f64 %1, %1, %2\n v_add_f64 %2, %2, %3\n v_add_f64 %3, %3, %0\n", 0, 0)
KERNEL(k_mul_f64, "v_mul_f64 %0, %0, %1\n v_mul_f64 %1, %1, %2\n v_mul_f64 %2, %2, %3\n v_mul_f64 %3, %3, %0\n", 0, 0)
KERNEL(k_fma_f64, "v_fma_f64 %0, %0, %1, %2\n v_fma_f64 %1, %1, %2, %3\n v_fma_f64 %2, %2, %3, %0\n v_fma_f64 %3, %3, %0, %1\n", 0, 0)
KERNEL(k_cvt_f64_f32, "v_cvt_f64_f32 %0, %4\n v_cvt_f64_f32 %1, %5\n v_cvt_f64_f32 %2, %6\n v_cvt_f64_f32 %3, %7\n", 0, 0)
KERNEL(k_cvt_f64_u32, "v_cvt_f64_u32 %0, %8\n v_cvt_f64_u32 %1, %9\n v_cvt_f64_u32 %2, %10\n v_cvt_f64_u32 %3, %11\n", 0, 0)
KERNEL(k_cvt_f32_f64, "v_cvt_f32_f64 %4, %0\n v_cvt_f32_f64 %5, %1\n v_cvt_f32_f64 %6, %2\n v_cvt_f32_f64 %7, %3\n", 0, 0)
KERNEL(k_mul_hi_u32, "v_mul_hi_u32 %8, %8, %9\n v_mul_hi_u32 %9, %9, %10\n v_mul_hi_u32 %10, %10, %11\n v_mul_hi_u32 %11, %11, %8\n", 0, 0)
KERNEL(k_mul_lo_u32, "v_mul_lo_u32 %8, %8, %9\n v_mul_lo_u32 %9, %9, %10\n v_mul_lo_u32 %10, %10, %11\n v_mul_lo_u32 %11, %11, %8\n", 0, 0)
KERNEL(k_add_f32, "v_add_f32 %4, %4, %5\n v_add_f32 %5, %5, %6\n v_add_f32 %6, %6, %7\n v_add_f32 %7, %7, %4\n", 0, 0)
KERNEL(k_dpp_mov, "v_mov_b32_dpp %4, %5 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %5, %6 row_shr:1 row_mask:0xf bank_mask:0xf\n"
                  "v_mov_b32_dpp %6, %7 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %7, %4 row_shr:1 row_mask:0xf bank_mask:0xf\n", 0, 0)
KERNEL(k_cvt_f32_u32, "v_cvt_f32_u32 %4, %8\n v_cvt_f32_u32 %5, %9\n v_cvt_f32_u32 %6, %10\n v_cvt_f32_u32 %7, %11\n", 0, 0)
KERNEL(k_mad_u32_u24, "v_mad_u32_u24 %8, %8, %9, %10\n v_mad_u32_u24 %9, %9, %10, %11\n v_mad_u32_u24 %10, %10, %11, %8\n v_mad_u32_u24 %11, %11, %8, %9\n", 0, 0)

int main() {
    float* out;
    hipMalloc(&out, 256 * 1024 * 4);
    struct { const char* name; void (*fn)(float*, int); } tests[] = {
        {"v_add_f32", k_add_f32}, {"v_add_f64", k_add_f64}, {"v_mul_f64", k_mul_f64}, {"v_fma_f64", k_fma_f64},
        {"v_cvt_f64_f32", k_cvt_f64_f32}, {"v_cvt_f64_u32", k_cvt_f64_u32}, {"v_cvt_f32_f64", k_cvt_f32_f64},
        {"v_cvt_f32_u32", k_cvt_f32_u32}, {"v_mul_hi_u32", k_mul_hi_u32}, {"v_mul_lo_u32", k_mul_lo_u32},
        {"v_mad_u32_u24", k_mad_u32_u24}, {"v_mov_b32_dpp", k_dpp_mov}};
    const int iters = 1000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (auto& t : tests) {
        float ms = 0;
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0, 0);
            hipLaunchKernelGGL(t.fn, dim3(256), dim3(1024), 0, 0, out, iters);  // 16 waves per CU = 4 per SIMD
            hipEventRecord(e1, 0);
            hipDeviceSynchronize();
            hipEventElapsedTime(&ms, e0, e1);
        }
        // per SIMD: 4 waves x iters x 64 instructions
        printf("%-14s %.2f ns of SIMD time per wave-instruction\n", t.name, ms * 1e6 / (4.0 * iters * 64));
    }
    return 0;
}
