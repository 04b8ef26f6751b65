// Microbenchmark: sustained bf16 MFMA rates on gfx950 (registers only): the legacy K=16 form
// (v_mfma_f32_16x16x16_bf16) against the K=32 form (v_mfma_f32_16x16x32_bf16), plus the cost of producing the
// operands from fp32 registers with v_cvt_pk_bf16_f32 (the "fp32 storage, bf16 MFMA" variant of the conv kernels).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ unsigned cvt2(float a, float b) {
    unsigned r;
    asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

template <int MODE, int NACC>
__global__ __launch_bounds__(256) void k(float* out, int iters, float a0, float b0) {
    f32x4 acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = f32x4{0, 0, 0, 0};
    f32x4 fa = f32x4{a0, a0 + 1, a0 + 2, a0 + 3} * (1.f + threadIdx.x * 1e-3f);
    f32x4 fb = f32x4{b0, b0 + 1, b0 + 2, b0 + 3} * (1.f + threadIdx.x * 1e-3f);
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) {          // K=16, operands pre-packed
            u32x2 pa = {cvt2(fa.x, fa.y), cvt2(fa.z, fa.w)}, pb = {cvt2(fb.x, fb.y), cvt2(fb.z, fb.w)};
#pragma unroll
            for (int i = 0; i < NACC; ++i)
                acc[i] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(__builtin_bit_cast(s16x4, pa), __builtin_bit_cast(s16x4, pb), acc[i], 0, 0, 0);
        } else if (MODE == 1) {   // K=32, operands pre-packed
            u32x4 pa = {cvt2(fa.x, fa.y), cvt2(fa.z, fa.w), cvt2(fa.y, fa.x), cvt2(fa.w, fa.z)};
            u32x4 pb = {cvt2(fb.x, fb.y), cvt2(fb.z, fb.w), cvt2(fb.y, fb.x), cvt2(fb.w, fb.z)};
#pragma unroll
            for (int i = 0; i < NACC; ++i)
                acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, pa), __builtin_bit_cast(bf16x8, pb), acc[i], 0, 0, 0);
        } else {                  // K=16 with a fresh B conversion per MFMA (2 cvt per MFMA)
#pragma unroll
            for (int i = 0; i < NACC; ++i) {
                fb += 1.0f;
                u32x2 pa = {cvt2(fa.x, fa.y), cvt2(fa.z, fa.w)}, pb = {cvt2(fb.x, fb.y), cvt2(fb.z, fb.w)};
                acc[i] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(__builtin_bit_cast(s16x4, pa), __builtin_bit_cast(s16x4, pb), acc[i], 0, 0, 0);
            }
        }
        fa += 1.0f;
    }
    f32x4 s = acc[0];
    for (int i = 1; i < NACC; ++i) s += acc[i];
    if (s.x == 123.456f) out[0] = s.x + s.y + s.z + s.w;
}

template <int MODE>
void run(const char* name, double flops_per_mfma, float* d) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int wg_per_cu : {1, 2}) {
        const int iters = 20000, NACC = 8;
        dim3 grid(256 * wg_per_cu);
        hipLaunchKernelGGL((k<MODE, NACC>), grid, dim3(256), 0, 0, d, 100, 1.f, 1.f);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        for (int r = 0; r < 5; ++r) hipLaunchKernelGGL((k<MODE, NACC>), grid, dim3(256), 0, 0, d, iters, 1.f, 1.f);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        double n = 5.0 * grid.x * 4 * (double)iters * NACC;
        printf("%-34s wg/cu %d: %7.1f TFLOP/s, %.2f ns per MFMA per SIMD\n", name, wg_per_cu, n * flops_per_mfma / (ms * 1e-3) / 1e12,
               ms * 1e6 / ((double)iters * NACC * 5 * wg_per_cu));
    }
}
int main() {
    float* d; hipMalloc(&d, 4);
    run<0>("16x16x16 bf16 (K=16)", 8192.0, d);
    run<1>("16x16x32 bf16 (K=32)", 16384.0, d);
    run<2>("16x16x16 bf16 + 4 cvt per MFMA", 8192.0, d);
    return 0;
}
