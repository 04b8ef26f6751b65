// Microbenchmark: sustained v_mfma_f32_16x16x4_f32 rate (registers only) -> calibrates the f32 MFMA roofline.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int NACC>
__global__ __launch_bounds__(256) void k(float* out, int iters, float a0, float b0) {
    f32x4 acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = f32x4{0, 0, 0, 0};
    float a = a0 + threadIdx.x * 1e-3f, b = b0 + threadIdx.x * 1e-3f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
    }
    f32x4 s = acc[0];
    for (int i = 1; i < NACC; ++i) s += acc[i];
    if (s.x == 123.456f) out[0] = s.x + s.y + s.z + s.w;
}
int main() {
    float* d; hipMalloc(&d, 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int wg_per_cu : {1, 2, 4}) {
        const int iters = 20000, NACC = 8;
        dim3 grid(256 * wg_per_cu);
        hipLaunchKernelGGL(k<NACC>, grid, dim3(256), 0, 0, d, 100, 1.f, 1.f);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(k<NACC>, grid, dim3(256), 0, 0, d, iters, 1.f, 1.f);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        double flops = 5.0 * grid.x * 4 /*waves*/ * (double)iters * NACC * 2048.0;
        printf("wg/cu %d: %.1f TFLOP/s (%.2f ms)\n", wg_per_cu, flops / (ms * 1e-3) / 1e12, ms);
    }
    return 0;
}
