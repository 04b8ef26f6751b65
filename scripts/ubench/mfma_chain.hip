// Microbenchmark: v_mfma_f32_16x16x4_f32 issue rate as a function of the number of independent accumulator chains per
// wave (NACC) and of the waves per SIMD.  Explains the MFMA-pipe occupancy ceiling of kernels whose waves alternate
// between only two accumulators (the fused level-0 blocks).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int NACC>
__global__ __launch_bounds__(512) void k(float* out, int iters, float a0, float b0) {
    f32x4 acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = f32x4{0, 0, 0, 0};
    float a = a0 + threadIdx.x * 1e-3f, b = b0 + threadIdx.x * 1e-3f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 8; ++r)
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
    }
    f32x4 s = acc[0];
#pragma unroll
    for (int i = 1; i < NACC; ++i) s += acc[i];
    if (s.x == 123.456f) out[0] = s.x + s.y + s.z + s.w;
}
template <int NACC>
void run(float* d, int threads) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 4000;
    dim3 grid(256);
    hipLaunchKernelGGL(k<NACC>, grid, dim3(threads), 0, 0, d, 10, 1.f, 1.f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(k<NACC>, grid, dim3(threads), 0, 0, d, iters, 1.f, 1.f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double n = 5.0 * grid.x * (threads / 64) * (double)iters * 8 * NACC;
    printf("chains/wave %d, waves/SIMD %d: %6.1f TFLOP/s (%4.1f %% of 157.3)\n", NACC, threads / 256, n * 2048.0 / (ms * 1e-3) / 1e12,
           100.0 * n * 2048.0 / (ms * 1e-3) / 157.3e12);
}
int main() {
    float* d; hipMalloc(&d, 4);
    for (int threads : {256, 512}) {
        run<1>(d, threads); run<2>(d, threads); run<4>(d, threads); run<8>(d, threads);
    }
    return 0;
}
