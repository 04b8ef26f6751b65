// Do the bf16 matrix pipeline and the fp32 vector ALU run side by side on ONE SIMD of gfx950?  (mfma_valu_coissue.hip asked the same for the
// fp32 MFMA: no -- one datapath.)  The f32s path multiplies on v_mfma_f32_16x16x32_bf16 (convs_kernel) and, at level 0, on v_pk_fma_f32
// (res8v_*): if a wave of each kind shares a SIMD at full rate, level-0 blocks co-resident with split-product blocks could hide one behind the
// other; if not, the arrangement cannot pay whatever the LDS budget.
// One block per CU, 2 waves per SIMD: waves 0-3 run loop A, waves 4-7 run loop B.  mode 0: MFMA | MFMA, 1: FMA | FMA, 2: MFMA | FMA,
// 3: one wave per SIMD MFMA alone, 4: FMA alone, 5: ONE stream interleaving 1 MFMA + 4 packed FMAs.
//   hipcc -O3 --offload-arch=gfx950 bf16mfma_pkfma_coissue.hip -o bf16mfma_pkfma_coissue && ./bf16mfma_pkfma_coissue
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
#define N_IT 2048
#define UNR 8

__device__ __forceinline__ void mfma_loop(float* out, int lane) {
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(0.001f * (lane + i)); b[i] = (__bf16)(0.002f * (lane - i)); }
    f32x4 c[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
    for (int it = 0; it < N_IT; ++it) {
#pragma unroll
        for (int u = 0; u < UNR; ++u) c[u & 3] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c[u & 3], 0, 0, 0);
    }
    out[lane] = c[0][0] + c[1][1] + c[2][2] + c[3][3];
}
__device__ __forceinline__ void fma_loop(float* out, int lane) {
    f32x2 x[8], w = {1.0001f, 0.9999f}, z = {1e-6f, -1e-6f};
    for (int i = 0; i < 8; ++i) x[i] = f32x2{0.001f * lane, 0.002f * i};
    for (int it = 0; it < N_IT; ++it) {
#pragma unroll
        for (int u = 0; u < UNR; ++u) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(x[u]) : "v"(w), "v"(z));
    }
    float s = 0;
    for (int i = 0; i < 8; ++i) s += x[i][0] + x[i][1];
    out[lane] = s;
}
__device__ __forceinline__ void mixed_loop(float* out, int lane) {
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(0.001f * (lane + i)); b[i] = (__bf16)(0.002f * (lane - i)); }
    f32x4 c[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
    f32x2 x[8], w = {1.0001f, 0.9999f}, z = {1e-6f, -1e-6f};
    for (int i = 0; i < 8; ++i) x[i] = f32x2{0.001f * lane, 0.002f * i};
    for (int it = 0; it < N_IT; ++it) {
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            c[u & 3] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c[u & 3], 0, 0, 0);
#pragma unroll
            for (int k = 0; k < 4; ++k) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(x[(4 * u + k) & 7]) : "v"(w), "v"(z));
        }
    }
    float s = c[0][0] + c[1][1] + c[2][2] + c[3][3];
    for (int i = 0; i < 8; ++i) s += x[i][0] + x[i][1];
    out[lane] = s;
}

__global__ __launch_bounds__(512) void k(float* out, int mode, unsigned long long* cyc) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const bool second = wave >= 4;
    float* o = out + (blockIdx.x * 8 + wave) * 64;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    if (mode == 0) mfma_loop(o, lane);
    else if (mode == 1) fma_loop(o, lane);
    else if (mode == 2) { if (second) fma_loop(o, lane); else mfma_loop(o, lane); }
    else if (mode == 3) { if (!second) mfma_loop(o, lane); }
    else if (mode == 4) { if (!second) fma_loop(o, lane); }
    else if (mode == 5) { if (!second) mixed_loop(o, lane); }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (lane == 0 && blockIdx.x == 0) cyc[wave] = t1 - t0;
}

int main() {
    float* d; unsigned long long* c;
    hipMalloc(&d, 256 * 8 * 64 * sizeof(float)); hipMalloc(&c, 8 * sizeof(unsigned long long));
    const char* names[6] = {"bf16 MFMA | bf16 MFMA (2 waves / SIMD)", "pk_fma | pk_fma (2 waves / SIMD)", "bf16 MFMA | pk_fma (one wave each per SIMD)",
                            "bf16 MFMA alone (1 wave / SIMD)", "pk_fma alone (1 wave / SIMD)", "one stream: 1 MFMA + 4 pk_fma interleaved"};
    for (int mode = 0; mode < 6; ++mode) {
        unsigned long long h[8] = {0};
        hipMemset(c, 0, sizeof(h));
        for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL(k, dim3(256), dim3(512), 0, 0, d, mode, c);
        hipDeviceSynchronize();
        hipMemcpy(h, c, sizeof(h), hipMemcpyDeviceToHost);
        const double n = (double)N_IT * UNR;
        printf("%-52s  wave 0 (A): %7.2f cyc per instr-slot   wave 4 (B): %7.2f\n", names[mode], h[0] / n, h[4] / n);
    }
    printf("(a slot = one MFMA in an MFMA loop, one v_pk_fma_f32 in an FMA loop, one MFMA + 4 FMAs in the interleaved stream;\n"
           " bf16 16x16x32 MFMA alone: 16 cyc; pk_fma alone: 4 cyc.  Side by side at full rate: A stays 16, B stays ~4-5.)\n");
    return 0;
}
