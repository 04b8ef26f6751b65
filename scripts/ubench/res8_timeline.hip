// Development aid: per-wave cycle stamps inside res8_up_kernel (the dominant kernel).  Built with -DASEP_R8_TIMELINE, which
// turns the R8_MARK() points of csrc/res8_kernels.h into clock64() stores for the first 16 blocks; prints, for one block,
// how long every phase of a pass takes and how long the waves wait at each barrier.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -DASEP_R8_TIMELINE -I citlab-article-separation-new_amd/csrc \
//         scripts/ubench/res8_timeline.hip -o /tmp/res8_timeline && /tmp/res8_timeline [H W]
// -DR8V: the vector-ALU kernel (res8v_up_kernel) instead of the MFMA one.
#include "res8v_kernels.h"
#include <cstdio>
#include <cstdlib>
#include <vector>
using namespace asep;
#ifdef R8V
#define UP_KERNEL res8v_up_kernel
#else
#define UP_KERNEL res8_up_kernel<false>
#endif
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

int main(int argc, char** argv) {
    const int H = argc > 2 ? atoi(argv[1]) : 4500, W = argc > 2 ? atoi(argv[2]) : 3000;
    const size_t n8 = (size_t)H * W * 8;
    float *skip, *dec, *out, *w1, *b1, *wr, *br;
    CK(hipMalloc(&skip, n8 * 4)); CK(hipMalloc(&dec, n8 * 4)); CK(hipMalloc(&out, n8 * 4));
    std::vector<float> h(1 << 20);
    for (auto& v : h) v = (rand() & 1023) / 1024.f - 0.4f;
    for (size_t o = 0; o < n8; o += h.size()) {
        const size_t n = std::min(h.size(), n8 - o);
        CK(hipMemcpy(skip + o, h.data(), n * 4, hipMemcpyHostToDevice));
        CK(hipMemcpy(dec + o, h.data(), n * 4, hipMemcpyHostToDevice));
    }
    CK(hipMalloc(&w1, 2 * 6 * 64 * 16)); CK(hipMalloc(&wr, 3 * 6 * 64 * 16)); CK(hipMalloc(&b1, 32)); CK(hipMalloc(&br, 96));
    for (auto& v : h) v *= 0.1f;
    CK(hipMemcpy(w1, h.data(), 2 * 6 * 64 * 16, hipMemcpyHostToDevice));
    CK(hipMemcpy(wr, h.data(), 3 * 6 * 64 * 16, hipMemcpyHostToDevice));
    CK(hipMemcpy(b1, h.data(), 32, hipMemcpyHostToDevice)); CK(hipMemcpy(br, h.data(), 96, hipMemcpyHostToDevice));
    Res8Args a{};
    a.nprob = 1;
    a.p[0].img = skip; a.p[0].in1 = dec; a.p[0].out = out; a.p[0].H = H; a.p[0].W = W;
    a.p[0].tiles_x = (W + R8_OW - 1) / R8_OW; a.p[0].tile_begin = 0;
    a.total_tiles = a.p[0].tiles_x * ((H + R8_OH * R8_NP - 1) / (R8_OH * R8_NP));
    a.w1 = w1; a.b1 = b1; a.wr = reinterpret_cast<const f32x4*>(wr); a.br = br; a.sched = nullptr;
    CK(hipFuncSetAttribute((const void*)UP_KERNEL, hipFuncAttributeMaxDynamicSharedMemorySize, (int)R8_UP_LDS));
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int it = 0; it < 3; ++it) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(UP_KERNEL, dim3(256), dim3(R8_THREADS), R8_UP_LDS, 0, a);
        hipEventRecord(e1); CK(hipEventSynchronize(e1));
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("launch %d: %.3f ms, %d units\n", it, ms, a.total_tiles);
    }
    static unsigned long long tl[16][8][64];
    CK(hipMemcpyFromSymbol(tl, HIP_SYMBOL(r8_tl), sizeof(tl)));
    unsigned long long clk[4];
    CK(hipMemcpyFromSymbol(clk, HIP_SYMBOL(r8_clk), sizeof(clk)));
    printf("block 0: %llu shader-clock ticks in %llu wall-clock ticks (100 MHz) -> clock64 runs at %.0f MHz\n", clk[2] - clk[0], clk[3] - clk[1],
           (double)(clk[2] - clk[0]) / (double)(clk[3] - clk[1]) * 100.0);
    const char* names[15] = {"pass start", "barrier A0 passed", "tile0 in LDS (barrier)", "conv1 src0 done", "barrier A1 passed",
                             "tile1 in LDS (barrier)", "conv1 src1 done", "t written", "barrier", "stage0 done", "barrier",
                             "stage1 done", "barrier", "prefetch issued", "stage2 done"};
    for (int blk : {0, 7}) {
        printf("---- block %d: cycles since the first mark, per wave; passes 0..3 ----\n", blk);
        const unsigned long long t0 = tl[blk][0][0];
        for (int m = 0; m < 60; ++m) {
            unsigned long long lo = ~0ull, hi = 0;
            for (int w = 0; w < 8; ++w) { lo = std::min(lo, tl[blk][w][m]); hi = std::max(hi, tl[blk][w][m]); }
            static unsigned long long prev_hi = 0;
            printf("pass %d %-26s first %8llu last %8llu  (+%6lld since previous mark's last wave)\n", m / 15, names[m % 15], lo - t0, hi - t0,
                   (long long)(hi - (m ? prev_hi : hi)));
            prev_hi = hi;
        }
    }
    // per-wave phase durations of pass 1 (a carried pass) of block 0
    printf("---- block 0, pass 1: per-wave duration of each phase (cycles) ----\n");
    for (int m = 16; m < 30; ++m) {
        printf("%-26s", names[m % 15]);
        for (int w = 0; w < 8; ++w) printf(" %6lld", (long long)(tl[0][w][m] - tl[0][w][m - 1]));
        printf("\n");
    }
    return 0;
}
