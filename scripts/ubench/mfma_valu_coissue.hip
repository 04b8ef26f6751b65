// Do the matrix pipe and the vector ALU of a gfx950 SIMD run side by side when they are fed by DIFFERENT waves?
// 512 threads per block, two waves per SIMD: waves 0-3 issue v_mfma_f32_16x16x4_f32, waves 4-7 v_pk_fma_f32 (mode 2),
// or all eight the same instruction (modes 0 / 1).  Prints the block span per instruction and the clock the run sustained.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 scripts/ubench/mfma_valu_coissue.hip -o /tmp/coissue && /tmp/coissue
#include <hip/hip_runtime.h>
#include <cstdio>
#include <algorithm>
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// mode 0: all waves MFMA; 1: all waves pk_fma; 2: waves 0-3 MFMA, 4-7 pk_fma.  n_mfma / n_fma = instructions per wave
// mode 3: as 2, the MFMA waves run ONE dependent accumulator chain; mode 4: as 2, pk_fma waves at s_setprio 3;
// mode 5: as 2, MFMA waves at s_setprio 3; mode 6: every wave interleaves 1 MFMA with 7 pk_fma in its own stream
__global__ __launch_bounds__(512) void k(float* out, unsigned long long* cyc, int mode, int iters_mfma, int iters_fma) {
    const int wave = threadIdx.x >> 6;
    const bool do_mfma = mode == 0 || (mode >= 2 && mode <= 5 && wave < 4);
    f32x4 m[4];
    f32x2 acc[8];
#pragma unroll
    for (int q = 0; q < 4; ++q) m[q] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int q = 0; q < 8; ++q) acc[q] = f32x2{0.f, 0.f};
    const float av = (float)threadIdx.x, bv = 1.f;
    f32x2 a = f32x2{av, bv};
    f32x2 ws = f32x2{0.5f, 0.25f};
    asm volatile("" : "+s"(ws));
    __syncthreads();
    if (mode == 4 && !do_mfma) __builtin_amdgcn_s_setprio(3);
    if (mode == 5 && do_mfma) __builtin_amdgcn_s_setprio(3);
    const unsigned long long t0 = clock64(), w0 = wall_clock64();
    if (mode == 6) {
        for (int it = 0; it < iters_mfma; ++it) {
#pragma unroll
            for (int u = 0; u < 8; ++u)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(m[q]) : "v"(av), "v"(bv));
#pragma unroll
                    for (int e = 0; e < 7; ++e) asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1]" : "+v"(acc[e]) : "v"(a), "s"(ws));
                }
        }
    } else if (do_mfma && mode == 3) {
        for (int it = 0; it < iters_mfma; ++it) {
#pragma unroll
            for (int u = 0; u < 32; ++u) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(m[0]) : "v"(av), "v"(bv));
        }
    } else if (do_mfma) {
        for (int it = 0; it < iters_mfma; ++it) {
#pragma unroll
            for (int u = 0; u < 8; ++u)
#pragma unroll
                for (int q = 0; q < 4; ++q) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(m[q]) : "v"(av), "v"(bv));
        }
    } else {
        for (int it = 0; it < iters_fma; ++it) {
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int q = 0; q < 8; ++q) asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1]" : "+v"(acc[q]) : "v"(a), "s"(ws));
        }
    }
    const unsigned long long t1 = clock64(), w1 = wall_clock64();
    float s = 0;
#pragma unroll
    for (int q = 0; q < 4; ++q) s += m[q].x + m[q].y + m[q].z + m[q].w;
#pragma unroll
    for (int q = 0; q < 8; ++q) s += acc[q].x + acc[q].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) {
        unsigned long long* c = cyc + (blockIdx.x * 8 + wave) * 4;
        c[0] = t0; c[1] = t1; c[2] = w0; c[3] = w1;
    }
}

int main(int argc, char** argv) {
    float* out; unsigned long long* cyc;
    hipMalloc(&out, 256 * 512 * 4); hipMalloc(&cyc, 256 * 8 * 4 * 8);
    static unsigned long long hc[256 * 8 * 4];
    // per wave: 32 instructions per iteration.  MFMA 16x16x4 f32 = 32 cycles, pk_fma = 4 cycles: the same pipe time for
    // iters_fma = 8 * iters_mfma
    const int im = 400, iv = 3200;
    if (argc > 1) {
        // sustained clocks: the same kernels back to back for about two seconds each (the short runs below start cold)
        for (int mode = 0; mode < 2; ++mode) {
            for (int l = 0; l < 5000; ++l) hipLaunchKernelGGL(k, dim3(256), dim3(512), 0, 0, out, cyc, mode, im, iv);
            hipDeviceSynchronize();
            hipMemcpy(hc, cyc, sizeof(hc), hipMemcpyDeviceToHost);
            double span = 0, wall = 0;
            for (int b = 0; b < 256; ++b) {
                unsigned long long lo = ~0ull, hi = 0, wlo = ~0ull, whi = 0;
                for (int w = 0; w < 8; ++w) {
                    const unsigned long long* c = hc + (b * 8 + w) * 4;
                    lo = std::min(lo, c[0]); hi = std::max(hi, c[1]); wlo = std::min(wlo, c[2]); whi = std::max(whi, c[3]);
                }
                span += (double)(hi - lo); wall += (double)(whi - wlo);
            }
            const double mhz = span / wall * 100, n = mode == 0 ? 32.0 * im : 32.0 * iv;
            const double flop_per_inst = mode == 0 ? 2.0 * 16 * 16 * 4 : 4.0 * 64;
            const double tf = 1024.0 * 2 * n * flop_per_inst / (wall / 256 / 100e6) / 1e12;
            printf("sustained %s: clock %.0f MHz after 5000 launches, %.1f TFLOP/s\n", mode == 0 ? "v_mfma_f32_16x16x4_f32" : "v_pk_fma_f32", mhz, tf);
        }
        return 0;
    }
    for (int rep = 0; rep < 1; ++rep)
        for (int mode = 0; mode < 7; ++mode) {
            hipLaunchKernelGGL(k, dim3(256), dim3(512), 0, 0, out, cyc, mode, im, iv);
            hipDeviceSynchronize();
            hipMemcpy(hc, cyc, sizeof(hc), hipMemcpyDeviceToHost);
            double span = 0, wall = 0, span_m = 0, span_v = 0;
            for (int b = 0; b < 256; ++b) {
                unsigned long long lo = ~0ull, hi = 0, wlo = ~0ull, whi = 0, hm = 0, hv = 0;
                for (int w = 0; w < 8; ++w) {
                    const unsigned long long* c = hc + (b * 8 + w) * 4;
                    lo = std::min(lo, c[0]); hi = std::max(hi, c[1]); wlo = std::min(wlo, c[2]); whi = std::max(whi, c[3]);
                    if (w < 4) hm = std::max(hm, c[1]); else hv = std::max(hv, c[1]);
                }
                span += (double)(hi - lo); wall += (double)(whi - wlo); span_m += (double)(hm - lo); span_v += (double)(hv - lo);
            }
            span /= 256; wall /= 256; span_m /= 256; span_v /= 256;
            const double n_m = 32.0 * im, n_v = 32.0 * iv;
            if (mode == 0) printf("all MFMA   : %.0f cycles, %.2f cycles per MFMA per SIMD (2 waves), clock %.0f MHz\n", span, span / (2 * n_m), span / wall * 100);
            if (mode == 1) printf("all pk_fma : %.0f cycles, %.2f cycles per pk_fma per SIMD (2 waves), clock %.0f MHz\n", span, span / (2 * n_v), span / wall * 100);
            if (mode == 6) printf("each wave 1 MFMA + 7 pk_fma interleaved: %.0f cycles, %.2f cycles per (MFMA + 7 pk_fma) per SIMD (2 waves), clock %.0f MHz\n", span, span / (2 * n_m), span / wall * 100);
            if (mode >= 2 && mode <= 5) printf("[mode %d] ", mode);
            if (mode >= 2 && mode <= 5) printf("1 MFMA wave + 1 pk_fma wave per SIMD: MFMA waves done after %.0f cycles (%.2f per MFMA), pk_fma waves after %.0f (%.2f per pk_fma), clock %.0f MHz\n",
                                  span_m, span_m / n_m, span_v, span_v / n_v, span / wall * 100);
        }
    return 0;
}
