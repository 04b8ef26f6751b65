// Can the instructions of a bf16 kernel's EPILOGUE (v_cvt_pk_bf16_f32, v_pk_max_i16, v_add_u32, v_permlane16_swap, ds_read_b128 / ds_write_b128)
// run in the shadow of v_mfma_f32_16x16x32_bf16 on gfx950 -- (a) interleaved in ONE wave's instruction stream (software pipelining: the epilogue of
// tile s between the MFMAs of tile s + 1), (b) from a second wave of the same SIMD?  bf16mfma_pkfma_coissue.hip asked this for v_pk_fma_f32 (no:
// 1 MFMA + 4 packed FMAs = 48 cycles against 16 + 4 x 5.7); the fused bf16 blocks (res8f / res16f / res32_tail) issue 2.6 - 3.4 such simple vector
// instructions per MFMA and the phase stamps say their cycles ADD (DESIGN_LESSONS 20).  If they hide here, the blocks want software pipelining.
//   hipcc -O3 --offload-arch=gfx950 bf16mfma_epilogue_coissue.hip -o bf16mfma_epilogue_coissue && ./bf16mfma_epilogue_coissue
// One block per CU; modes with one wave per SIMD (waves 0-3) or two (waves 4-7 run the partner loop).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
#define N_IT 1024
#define UNR 8

enum { OP_NONE = 0, OP_PKMAX, OP_CVT, OP_ADD, OP_SWAP, OP_LDSR, OP_LDSW, OP_MIX };

template <int OP>
__device__ __forceinline__ void one_op(unsigned (&x)[8], int i, unsigned ldsaddr) {
    if constexpr (OP == OP_PKMAX) asm volatile("v_pk_max_i16 %0, %0, %1" : "+v"(x[i & 7]) : "v"(x[(i + 4) & 7]));
    else if constexpr (OP == OP_CVT) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(x[i & 7]) : "v"(x[(i + 3) & 7]), "v"(x[(i + 5) & 7]));
    else if constexpr (OP == OP_ADD) asm volatile("v_add_u32 %0, %0, %1" : "+v"(x[i & 7]) : "v"(x[(i + 4) & 7]));
    else if constexpr (OP == OP_SWAP) asm volatile("v_permlane16_swap_b32 %0, %1" : "+v"(x[i & 7]), "+v"(x[(i + 4) & 7]));
    else if constexpr (OP == OP_LDSR) {
        u32x4 r;
        asm volatile("ds_read_b128 %0, %1" : "=v"(r) : "v"(ldsaddr));
        asm volatile("" ::"v"(r));
    } else if constexpr (OP == OP_LDSW) {
        u32x4 r = {x[0], x[1], x[2], x[3]};
        asm volatile("ds_write_b128 %0, %1" ::"v"(ldsaddr), "v"(r));
    } else if constexpr (OP == OP_MIX) {   // the real epilogue's mix per 4 values: 2 cvt, 2 pk_max, 1 add, (swap every other)
        const int k = i % 6;
        if (k < 2) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(x[i & 7]) : "v"(x[(i + 3) & 7]), "v"(x[(i + 5) & 7]));
        else if (k < 4) asm volatile("v_pk_max_i16 %0, %0, %1" : "+v"(x[i & 7]) : "v"(x[(i + 4) & 7]));
        else if (k == 4) asm volatile("v_add_u32 %0, %0, %1" : "+v"(x[i & 7]) : "v"(x[(i + 4) & 7]));
        else asm volatile("v_permlane16_swap_b32 %0, %1" : "+v"(x[i & 7]), "+v"(x[(i + 4) & 7]));
    }
}

// one stream: per slot 1 MFMA (four independent accumulator chains) + K ops of kind OP
template <int OP, int K, bool MFMA>
__device__ __forceinline__ void stream(float* out, int lane, unsigned ldsaddr) {
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(0.001f * (lane + i)); b[i] = (__bf16)(0.002f * (lane - i)); }
    f32x4 c[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
    unsigned x[8];
    for (int i = 0; i < 8; ++i) x[i] = 0x3f803f80u + lane * 7 + i;
    for (int it = 0; it < N_IT; ++it) {
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            if constexpr (MFMA) c[u & 3] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c[u & 3], 0, 0, 0);
#pragma unroll
            for (int k = 0; k < K; ++k) one_op<OP>(x, u * K + k, ldsaddr);
        }
        if constexpr (OP == OP_LDSR || OP == OP_LDSW) asm volatile("s_waitcnt lgkmcnt(0)");
    }
    float s = c[0][0] + c[1][1] + c[2][2] + c[3][3];
    for (int i = 0; i < 8; ++i) s += (float)x[i];
    out[lane] = s;
}

template <int OP, int K>
__global__ __launch_bounds__(512) void k(float* out, int two_waves, unsigned long long* cyc) {
    __shared__ u32x4 lds[512];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    lds[threadIdx.x] = u32x4{1u, 2u, 3u, 4u};
    __syncthreads();
    const bool second = wave >= 4;
    float* o = out + (blockIdx.x * 8 + wave) * 64;
    const unsigned ldsaddr = (unsigned)(threadIdx.x * 16);
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    if (two_waves == 0) { if (!second) stream<OP, K, true>(o, lane, ldsaddr); }                       // one stream: MFMA + K ops interleaved
    else if (two_waves == 1) { if (second) stream<OP, K, false>(o, lane, ldsaddr); else stream<OP_NONE, 0, true>(o, lane, ldsaddr); }   // MFMA wave | op wave
    else { if (!second) stream<OP, K, false>(o, lane, ldsaddr); }                                   // the ops alone
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (lane == 0 && blockIdx.x == 0) cyc[wave] = t1 - t0;
}

template <int OP, int K>
void run(const char* name, float* d, unsigned long long* c) {
    double r[3][2];
    for (int mode = 0; mode < 3; ++mode) {
        unsigned long long h[8] = {0};
        hipMemset(c, 0, sizeof(h));
        for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL((k<OP, K>), dim3(256), dim3(512), 0, 0, d, mode, c);
        hipDeviceSynchronize();
        hipMemcpy(h, c, sizeof(h), hipMemcpyDeviceToHost);
        const double n = (double)N_IT * UNR;
        r[mode][0] = h[0] / n; r[mode][1] = h[4] / n;
    }
    printf("%-22s K=%d | one stream (1 MFMA + K ops): %6.2f cyc/slot | ops alone: %6.2f cyc/slot (%.2f per op) | MFMA wave %6.2f beside op wave %6.2f cyc/slot\n",
           name, K, r[0][0], r[2][0], K ? r[2][0] / K : 0.0, r[1][0], r[1][1]);
}

int main() {
    float* d; unsigned long long* c;
    hipMalloc(&d, 256 * 8 * 64 * sizeof(float)); hipMalloc(&c, 8 * sizeof(unsigned long long));
    printf("memtime ticks are 100 MHz on gfx950: cycles here = ticks of s_memtime; compare rows, the MFMA-alone row is the unit\n");
    run<OP_NONE, 0>("MFMA alone", d, c);
    run<OP_PKMAX, 1>("v_pk_max_i16", d, c); run<OP_PKMAX, 2>("v_pk_max_i16", d, c); run<OP_PKMAX, 3>("v_pk_max_i16", d, c); run<OP_PKMAX, 4>("v_pk_max_i16", d, c); run<OP_PKMAX, 6>("v_pk_max_i16", d, c);
    run<OP_CVT, 2>("v_cvt_pk_bf16_f32", d, c); run<OP_CVT, 4>("v_cvt_pk_bf16_f32", d, c);
    run<OP_ADD, 2>("v_add_u32", d, c); run<OP_ADD, 3>("v_add_u32", d, c); run<OP_ADD, 4>("v_add_u32", d, c);
    run<OP_SWAP, 1>("v_permlane16_swap", d, c); run<OP_SWAP, 2>("v_permlane16_swap", d, c);
    run<OP_LDSR, 1>("ds_read_b128", d, c); run<OP_LDSR, 2>("ds_read_b128", d, c);
    run<OP_LDSW, 1>("ds_write_b128", d, c);
    run<OP_MIX, 3>("epilogue mix", d, c); run<OP_MIX, 4>("epilogue mix", d, c); run<OP_MIX, 6>("epilogue mix", d, c);
    return 0;
}
