// Microbenchmark: one 3x3 8->8 convolution stage of the fused level-0 block on the VECTOR ALU instead of the MFMA.
// M = 8 output channels only half-fills a 16-row MFMA tile (the pixel-pair mapping recovers 75 %); a v_fmac_f32 with the
// weight as a scalar operand has no such granularity, and the f32 vector peak of gfx950 equals the f32 matrix peak
// (256 CUs x 4 SIMDs x 32 lanes x 2 FLOP x 2.4 GHz = 157 TFLOP/s).  Thread = two adjacent pixels x 8 output channels;
// 512 threads = 16 rows x 32 pairs = one stage (16 x 64 pixels).  Ideal: 1024 px x 576 FMA / (4 SIMDs x 32 lanes) = 4608 cycles.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 scripts/ubench/valu_conv8.hip -o /tmp/valu_conv8 && /tmp/valu_conv8
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
constexpr int PITCH = 72, ROWS_IN = 18, ROWS_OUT = 16;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__device__ __forceinline__ int px_off(int x, int hf) { return ((x >> 1) << 4) + (((((x & 1) << 1) | hf) ^ ((x >> 2) & 3)) << 2); }

// w: [3][3][8 ci][8 co] floats (uniform -> scalar loads), b: [8]
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2)))
void stage_kernel(const float* __restrict__ w, const float* __restrict__ bias, const float* __restrict__ init, float* __restrict__ out,
                  unsigned long long* __restrict__ cyc, int iters) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float* IN = sm;                                   // [18][72][8]
    float* OUT = sm + ROWS_IN * PITCH * 8;            // [16][72][8]
    const int tid = threadIdx.x;
    for (int i = tid; i < ROWS_IN * PITCH * 8; i += 512) IN[i] = init[i];
    __syncthreads();
    const int r = tid >> 5, p = tid & 31;             // output row r (input rows r..r+2), pixels 2p+1, 2p+2 of the frame (cols 1..64)
    int off[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i) { off[i][0] = px_off(2 * p + i, 0); off[i][1] = px_off(2 * p + i, 1); }
    unsigned long long t0 = 0;
    for (int it = 0; it < iters; ++it) {
        if (it == 1) t0 = clock64();
        // packed f32: one v_pk_fma_f32 = two output channels (the weight pair is a scalar register pair, the input value is
        // broadcast to both halves); a plain v_fma_f32 issues at 16 lanes per cycle (measured: 4.9 cycles per wave64
        // instruction with two waves per SIMD), the packed form doubles the FLOPs per issue slot
        f32x2 acc0[4], acc1[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) { acc0[q] = f32x2{bias[2 * q], bias[2 * q + 1]}; acc1[q] = acc0[q]; }
        typedef const float __attribute__((address_space(4)))* cptr;
        cptr wl = (cptr)w;
        asm volatile("" : "+s"(wl));
        float wc[32], wn[32];
#pragma unroll
        for (int k = 0; k < 32; ++k) wc[k] = wl[k];
#ifdef PIPE16
        // as PIPE, but the weight double buffer is 2 x 16 scalar registers (the fused kernels carry ~40 scalars of their own
        // and the SGPR file has 102): a wait every 16 packed FMAs
        float hc[16], hn[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) hc[k] = wl[k];
        f32x4 dA[4], dB[4];
#pragma unroll
#ifdef ROWSWZ
#define RS(ky) ((((r + (ky)) & 1) ? 8 : 0))
#else
#define RS(ky) 0
#endif
        for (int i = 0; i < 4; ++i) dA[i] = *reinterpret_cast<const f32x4*>(IN + r * PITCH * 8 + (off[i][0] ^ RS(0)));
#pragma unroll
        for (int g2 = 0; g2 < 36; ++g2) {
            const int g = g2 >> 1, ch = g2 & 1, kx = g % 3, rh = g / 3;
            asm volatile("" :: "s"(hc[0]), "v"(dA[0]), "v"(dA[1]), "v"(dA[2]), "v"(dA[3]));
            __builtin_amdgcn_sched_barrier(0);
            if (g2 + 1 < 36) {
#pragma unroll
                for (int k = 0; k < 16; ++k) hn[k] = wl[(g2 + 1) * 16 + k];
            }
            if (kx == 0 && ch == 0 && rh + 1 < 6) {
                const int ky2 = (rh + 1) >> 1, hf2 = (rh + 1) & 1;
#pragma unroll
                for (int i = 0; i < 4; ++i) dB[i] = *reinterpret_cast<const f32x4*>(IN + (r + ky2) * PITCH * 8 + (off[i][hf2] ^ RS(ky2)));
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int c2 = 0; c2 < 2; ++c2) {
                const int c = ch * 2 + c2;
                const f32x2 a0 = f32x2{dA[kx][c], dA[kx][c]}, a1 = f32x2{dA[kx + 1][c], dA[kx + 1][c]};
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const f32x2 wv = f32x2{hc[c2 * 8 + 2 * q], hc[c2 * 8 + 2 * q + 1]};
                    acc0[q] = __builtin_elementwise_fma(a0, wv, acc0[q]);
                    acc1[q] = __builtin_elementwise_fma(a1, wv, acc1[q]);
                }
            }
#pragma unroll
            for (int k = 0; k < 16; ++k) hc[k] = hn[k];
            if (kx == 2 && ch == 1) {
#pragma unroll
                for (int i = 0; i < 4; ++i) dA[i] = dB[i];
            }
        }
#elif defined(PIPE)
        // explicit software pipeline.  Scalar loads return out of order, so every wait on them is lgkmcnt(0) and covers
        // the LDS reads as well: per weight group (32 packed FMAs) there is ONE wait, placed first (the empty asm "uses"
        // the group's registers), and only then are the next group's weights and the next row-half's inputs requested.
        f32x4 dA[4], dB[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) dA[i] = *reinterpret_cast<const f32x4*>(IN + r * PITCH * 8 + off[i][0]);
#pragma unroll
        for (int g = 0; g < 18; ++g) {
            const int kx = g % 3, rh = g / 3;           // rh = ky * 2 + hf
            asm volatile("" :: "s"(wc[0]), "s"(wc[16]), "v"(dA[0]), "v"(dA[1]), "v"(dA[2]), "v"(dA[3]));
            __builtin_amdgcn_sched_barrier(0);
            if (g + 1 < 18) {
#pragma unroll
                for (int k = 0; k < 32; ++k) wn[k] = wl[(g + 1) * 32 + k];
            }
            if (kx == 0 && rh + 1 < 6) {
                const int ky2 = (rh + 1) >> 1, hf2 = (rh + 1) & 1;
#pragma unroll
                for (int i = 0; i < 4; ++i) dB[i] = *reinterpret_cast<const f32x4*>(IN + (r + ky2) * PITCH * 8 + off[i][hf2]);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const f32x2 a0 = f32x2{dA[kx][c], dA[kx][c]}, a1 = f32x2{dA[kx + 1][c], dA[kx + 1][c]};
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const f32x2 wv = f32x2{wc[c * 8 + 2 * q], wc[c * 8 + 2 * q + 1]};
                    acc0[q] = __builtin_elementwise_fma(a0, wv, acc0[q]);
                    acc1[q] = __builtin_elementwise_fma(a1, wv, acc1[q]);
                }
            }
#pragma unroll
            for (int k = 0; k < 32; ++k) wc[k] = wn[k];
            if (kx == 2) {
#pragma unroll
                for (int i = 0; i < 4; ++i) dA[i] = dB[i];
            }
        }
#else
#ifdef PREFETCH_ALL
        // all 24 LDS reads of the thread go out before the first FMA (96 VGPRs; the budget at two waves per SIMD is 256)
        f32x4 dd[3][2][4];
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
            for (int hf = 0; hf < 2; ++hf)
#pragma unroll
                for (int i = 0; i < 4; ++i) dd[ky][hf][i] = *reinterpret_cast<const f32x4*>(IN + (r + ky) * PITCH * 8 + off[i][hf]);
        __builtin_amdgcn_sched_barrier(0);
#endif
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
            const float* row = IN + (r + ky) * PITCH * 8;
#pragma unroll
            for (int hf = 0; hf < 2; ++hf) {
                f32x4 d[4];
#pragma unroll
#ifdef PREFETCH_ALL
                for (int i = 0; i < 4; ++i) d[i] = dd[ky][hf][i];
#else
                for (int i = 0; i < 4; ++i) d[i] = *reinterpret_cast<const f32x4*>(row + off[i][hf]);
#endif
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) {
                    const int g = (ky * 2 + hf) * 3 + kx;
#ifndef LOAD_AFTER_FIRST
                    if (g + 1 < 18) {
#pragma unroll
                        for (int k = 0; k < 32; ++k) wn[k] = wl[(g + 1) * 32 + k];
                    }
#endif
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        const f32x2 a0 = f32x2{d[kx][c], d[kx][c]}, a1 = f32x2{d[kx + 1][c], d[kx + 1][c]};
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            const f32x2 wv = f32x2{wc[c * 8 + 2 * q], wc[c * 8 + 2 * q + 1]};
                            acc0[q] = __builtin_elementwise_fma(a0, wv, acc0[q]);
                            acc1[q] = __builtin_elementwise_fma(a1, wv, acc1[q]);
                        }
#ifdef LOAD_AFTER_FIRST
                        // scalar loads return out of order, so every wait on them is lgkmcnt(0): the next group's request
                        // has to come AFTER the wait for this group's weights (= after its first FMAs), or that wait
                        // would cover the request just made
                        if (c == 0 && g + 1 < 18) {
                            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                            for (int k = 0; k < 32; ++k) wn[k] = wl[(g + 1) * 32 + k];
                            __builtin_amdgcn_sched_barrier(0);
                        }
#endif
                    }
#pragma unroll
                    for (int k = 0; k < 32; ++k) wc[k] = wn[k];
                }
            }
        }
#endif
        // ReLU, write both pixels (whole 32-byte pixels) to the output tile
        float* o = OUT + r * PITCH * 8;
#ifdef ROWSWZ
        const int ws = (r & 1) ? 8 : 0;
#else
        const int ws = 0;
#endif
        f32x4 v;
        v = f32x4{fmaxf(acc0[0].x, 0.f), fmaxf(acc0[0].y, 0.f), fmaxf(acc0[1].x, 0.f), fmaxf(acc0[1].y, 0.f)}; *reinterpret_cast<f32x4*>(o + (off[1][0] ^ ws)) = v;
        v = f32x4{fmaxf(acc0[2].x, 0.f), fmaxf(acc0[2].y, 0.f), fmaxf(acc0[3].x, 0.f), fmaxf(acc0[3].y, 0.f)}; *reinterpret_cast<f32x4*>(o + (off[1][1] ^ ws)) = v;
        v = f32x4{fmaxf(acc1[0].x, 0.f), fmaxf(acc1[0].y, 0.f), fmaxf(acc1[1].x, 0.f), fmaxf(acc1[1].y, 0.f)}; *reinterpret_cast<f32x4*>(o + (off[2][0] ^ ws)) = v;
        v = f32x4{fmaxf(acc1[2].x, 0.f), fmaxf(acc1[2].y, 0.f), fmaxf(acc1[3].x, 0.f), fmaxf(acc1[3].y, 0.f)}; *reinterpret_cast<f32x4*>(o + (off[2][1] ^ ws)) = v;
        __syncthreads();
    }
    if (tid == 0) cyc[blockIdx.x] = clock64() - t0;
    for (int i = tid; i < ROWS_OUT * PITCH * 8; i += 512) out[(size_t)blockIdx.x * ROWS_OUT * PITCH * 8 + i] = OUT[i];
}

int main() {
    const int nblk = 256, iters = 41;
    std::vector<float> hw(576), hb(8), hin(ROWS_IN * PITCH * 8);
    for (auto& v : hw) v = (rand() % 200 - 100) / 1000.f;
    for (auto& v : hb) v = 0.1f;
    for (auto& v : hin) v = (rand() % 1000) / 1000.f;
    float *w, *b, *in, *out; unsigned long long* cyc;
    CK(hipMalloc(&w, 576 * 4)); CK(hipMalloc(&b, 32)); CK(hipMalloc(&in, hin.size() * 4));
    CK(hipMalloc(&out, (size_t)nblk * ROWS_OUT * PITCH * 8 * 4)); CK(hipMalloc(&cyc, nblk * 8));
    CK(hipMemcpy(w, hw.data(), 576 * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(b, hb.data(), 32, hipMemcpyHostToDevice));
    CK(hipMemcpy(in, hin.data(), hin.size() * 4, hipMemcpyHostToDevice));
    const size_t lds = (size_t)(ROWS_IN + ROWS_OUT) * PITCH * 8 * 4;
    CK(hipFuncSetAttribute((const void*)stage_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    for (int rep = 0; rep < 3; ++rep) {
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0);
        hipLaunchKernelGGL(stage_kernel, dim3(nblk), dim3(512), lds, 0, w, b, in, out, cyc, iters);
        hipEventRecord(e1); CK(hipEventSynchronize(e1));
        float ms; hipEventElapsedTime(&ms, e0, e1);
        std::vector<unsigned long long> hc(nblk);
        CK(hipMemcpy(hc.data(), cyc, nblk * 8, hipMemcpyDeviceToHost));
        double avg = 0; for (auto v : hc) avg += (double)v; avg /= nblk * (iters - 1);
        const double flops = 2.0 * nblk * (iters) * 1024.0 * 576.0;
        printf("rep %d: %.3f ms, %.0f cycles per stage (ideal 4608 = %.0f %%), %.1f TFLOP/s useful\n", rep, ms, avg, 100.0 * 4608 / avg, flops / (ms * 1e-3) / 1e12);
    }
    // check one block against the host
    std::vector<float> ho(ROWS_OUT * PITCH * 8);
    CK(hipMemcpy(ho.data(), out, ho.size() * 4, hipMemcpyDeviceToHost));
    double maxerr = 0;
#ifdef ROWSWZ
    const int rsw = 8;
#else
    const int rsw = 0;
#endif
    auto IN = [&](int row, int x, int c) { const int hf = c >> 2; const int off = (((x >> 1) << 4) + (((((x & 1) << 1) | hf) ^ ((x >> 2) & 3)) << 2)) ^ ((row & 1) ? rsw : 0); return hin[row * PITCH * 8 + off + (c & 3)]; };
    for (int r = 0; r < 16; ++r) for (int x = 1; x <= 64; ++x) for (int co = 0; co < 8; ++co) {
        double s = hb[co];
        for (int ky = 0; ky < 3; ++ky) for (int kx = 0; kx < 3; ++kx) for (int ci = 0; ci < 8; ++ci)
            s += (double)IN(r + ky, x - 1 + kx, ci) * hw[((((ky * 2 + (ci >> 2)) * 3 + kx) * 4 + (ci & 3)) * 8) + co];
        s = s > 0 ? s : 0;
        const int hf = co >> 2; const int off = (((x >> 1) << 4) + (((((x & 1) << 1) | hf) ^ ((x >> 2) & 3)) << 2)) ^ ((r & 1) ? rsw : 0);
        maxerr = std::max(maxerr, std::abs(s - (double)ho[r * PITCH * 8 + off + (co & 3)]));
    }
    printf("max |err| vs host double: %.2e\n", maxerr);
    return 0;
}
