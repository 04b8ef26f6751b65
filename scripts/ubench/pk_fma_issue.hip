// Issue rate of v_pk_fma_f32 on gfx950 by operand kind (SIMD cycles per wave64 instruction = block span / instructions per SIMD, at 1, 2 and 4 waves per SIMD):
//   mode 0: all operands VGPR            mode 1: weight = SGPR pair
//   mode 2: SGPR pair + op_sel_hi broadcast of the VGPR input (the form the 8->8 stage uses)
//   mode 3: v_fma_f32 with an SGPR weight (plain, for comparison)
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 scripts/ubench/pk_fma_issue.hip -o /tmp/pk_fma_issue && /tmp/pk_fma_issue
#include <hip/hip_runtime.h>
#include <cstdio>
#include <algorithm>
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ __launch_bounds__(1024) void k(float* out, unsigned long long* cyc, const float* w, int iters) {
    f32x2 acc[8];
    f32x2 a = f32x2{(float)threadIdx.x, 1.f};
    typedef const float __attribute__((address_space(4)))* cptr;
    cptr wl = (cptr)w;
    f32x2 ws = f32x2{wl[0], wl[1]};
    f32x2 wv = f32x2{w[threadIdx.x & 7], w[8 + (threadIdx.x & 7)]};
#pragma unroll
    for (int q = 0; q < 8; ++q) acc[q] = f32x2{0.f, 0.f};
    __syncthreads();
    const unsigned long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 16; ++u)
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                if (MODE == 0) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(acc[q]) : "v"(a), "v"(wv));
                if (MODE == 1) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(acc[q]) : "v"(a), "s"(ws));
                if (MODE == 2) asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1]" : "+v"(acc[q]) : "v"(a), "s"(ws));
                if (MODE == 3) { asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(acc[q].x) : "v"(a.x), "s"(ws.x)); }
            }
    }
    const unsigned long long t1 = clock64();
    float s = 0;
#pragma unroll
    for (int q = 0; q < 8; ++q) s += acc[q].x + acc[q].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) { cyc[(blockIdx.x * 16 + (threadIdx.x >> 6)) * 2] = t0; cyc[(blockIdx.x * 16 + (threadIdx.x >> 6)) * 2 + 1] = t1; }
}

template <int MODE>
void run(int threads, float* out, unsigned long long* cyc, float* w) {
    const int iters = 200, nblk = 256;
    hipLaunchKernelGGL(k<MODE>, dim3(nblk), dim3(threads), 0, 0, out, cyc, w, iters);
    hipDeviceSynchronize();
    static unsigned long long hc[256 * 32];
    hipMemcpy(hc, cyc, sizeof(hc), hipMemcpyDeviceToHost);
    double avg = 0;
    const int waves = threads / 64;
    for (int b = 0; b < nblk; ++b) {   // a block's span: first start to last end over its waves
        unsigned long long lo = ~0ull, hi = 0;
        for (int wv = 0; wv < waves; ++wv) { lo = std::min(lo, hc[(b * 16 + wv) * 2]); hi = std::max(hi, hc[(b * 16 + wv) * 2 + 1]); }
        avg += (double)(hi - lo);
    }
    avg /= nblk;
    const int waves_per_simd = threads / 256;
    printf("mode %d, %d waves/SIMD: %.2f cycles per wave64 instruction\n", MODE, waves_per_simd, avg / (iters * 128.0 * waves_per_simd));
}

int main() {
    float *out, *w; unsigned long long* cyc;
    hipMalloc(&out, 256 * 1024 * 4); hipMalloc(&w, 64 * 4); hipMalloc(&cyc, 256 * 32 * 8);
    hipMemset(w, 0, 64 * 4);
    for (int threads : {256, 512, 1024}) {
        run<0>(threads, out, cyc, w); run<1>(threads, out, cyc, w); run<2>(threads, out, cyc, w); run<3>(threads, out, cyc, w);
    }
    return 0;
}
