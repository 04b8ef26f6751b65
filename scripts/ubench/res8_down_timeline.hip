// Development aid: per-wave cycle stamps inside res8_down_kernel (see res8_timeline.hip for the up kernel).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -DASEP_R8_TIMELINE -I citlab-article-separation-new_amd/csrc \
//         scripts/ubench/res8_down_timeline.hip -o build_tmp/res8_down_timeline && build_tmp/res8_down_timeline
#include "res8_kernels.h"
#include <cstdio>
#include <cstdlib>
#include <vector>
using namespace asep;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
int main() {
    const int H = 4500, W = 3000;
    float *img, *out, *pool, *w1, *b1, *wr, *br;
    CK(hipMalloc(&img, (size_t)H * W * 4)); CK(hipMalloc(&out, (size_t)H * W * 32)); CK(hipMalloc(&pool, (size_t)H * W * 8));
    std::vector<float> h(1 << 20);
    for (auto& v : h) v = (rand() & 1023) / 1024.f - 0.4f;
    for (size_t o = 0; o < (size_t)H * W; o += h.size()) CK(hipMemcpy(img + o, h.data(), std::min(h.size(), (size_t)H * W - o) * 4, hipMemcpyHostToDevice));
    CK(hipMalloc(&w1, 72 * 4)); CK(hipMalloc(&wr, 3 * 6 * 64 * 16)); CK(hipMalloc(&b1, 32)); CK(hipMalloc(&br, 96));
    for (auto& v : h) v *= 0.1f;
    CK(hipMemcpy(w1, h.data(), 72 * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(wr, h.data(), 3 * 6 * 64 * 16, hipMemcpyHostToDevice));
    CK(hipMemcpy(b1, h.data(), 32, hipMemcpyHostToDevice)); CK(hipMemcpy(br, h.data(), 96, hipMemcpyHostToDevice));
    Res8Args a{};
    a.nprob = 1;
    a.p[0].img = img; a.p[0].out = out; a.p[0].pool = pool; a.p[0].stats = nullptr; a.p[0].H = H; a.p[0].W = W;
    a.p[0].tiles_x = (W + R8_OW - 1) / R8_OW; a.p[0].tile_begin = 0;
    a.total_tiles = a.p[0].tiles_x * ((H + R8_OH * R8_NP - 1) / (R8_OH * R8_NP));
    a.w1 = w1; a.b1 = b1; a.wr = reinterpret_cast<const f32x4*>(wr); a.br = br; a.sched = nullptr;
    CK(hipFuncSetAttribute((const void*)res8_down_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)R8_DOWN_LDS));
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int it = 0; it < 3; ++it) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(res8_down_kernel<false>, dim3(256), dim3(R8_THREADS), R8_DOWN_LDS, 0, a);
        hipEventRecord(e1); CK(hipEventSynchronize(e1));
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("launch %d: %.3f ms, %d units\n", it, ms, a.total_tiles);
    }
    static unsigned long long tl[16][8][64];
    CK(hipMemcpyFromSymbol(tl, HIP_SYMBOL(r8_tl), sizeof(tl)));
    const char* names[11] = {"pass start", "barrier", "image tile + carried rows in LDS", "conv1 (VALU)", "barrier", "stage0", "barrier", "stage1", "barrier", "image prefetch issued", "stage2 (+ pool)"};
    printf("---- block 0, pass 1 (carried): cycles, last wave's stamp minus the previous phase's last stamp ----\n");
    for (int m = 12; m < 22; ++m) {
        unsigned long long hi = 0, ph = 0;
        for (int w = 0; w < 8; ++w) { hi = std::max(hi, tl[0][w][m]); ph = std::max(ph, tl[0][w][m - 1]); }
        printf("  %-34s %7lld\n", names[m - 11], (long long)(hi - ph));
    }
    return 0;
}
