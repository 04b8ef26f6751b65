// Development aid: per-wave cycle stamps inside conv_wino_kernel<2> (32 -> 32 channels, the level-2 residual convs).
// Built with -DASEP_WINO_TIMELINE (WINO_MARK() points of csrc/aru_kernels.h); blocks 4096..4607 of the launch (steady
// state, not the cold first generation) record clock64() per wave at every phase boundary.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -DASEP_WINO_TIMELINE -I citlab-article-separation-new_amd/csrc \
//         scripts/ubench/wino_timeline.hip -o build_tmp/wino_timeline && build_tmp/wino_timeline
#include "aru_kernels.h"
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
using namespace asep;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

int main() {
    const int H = 750, W = 1125, C = 32, G = 2, MTILES = 2;
    const size_t n = (size_t)H * W * C;
    float *in, *out, *res, *bias; f32x4* wpk;
    CK(hipMalloc(&in, n * 4)); CK(hipMalloc(&out, n * 4)); CK(hipMalloc(&res, n * 4)); CK(hipMalloc(&bias, C * 4));
    std::vector<float> h(n);
    for (auto& v : h) v = (rand() & 1023) / 1024.f - 0.4f;
    CK(hipMemcpy(in, h.data(), n * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(res, h.data(), n * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(bias, h.data(), C * 4, hipMemcpyHostToDevice));
    const size_t nw = (size_t)G * 16 * MTILES * 64;
    CK(hipMalloc(&wpk, nw * 16)); CK(hipMemcpy(wpk, h.data(), nw * 16, hipMemcpyHostToDevice));
    ConvArgs a{};
    a.nprob = 1;
    a.p[0].in0 = in; a.p[0].in1 = nullptr; a.p[0].res = nullptr; a.p[0].out = out;
    a.p[0].H = a.p[0].Ho = H; a.p[0].W = a.p[0].Wo = W;
    a.p[0].tiles_x = (W + WINO_TW - 1) / WINO_TW; a.p[0].tile_begin = 0;
    a.total_tiles = a.p[0].tiles_x * ((H + WINO_TH - 1) / WINO_TH);
    a.wpk = wpk; a.bias = bias; a.c0 = C; a.c1 = 0; a.cout = C; a.mtiles = MTILES; a.groups = G; a.relu_in = 0; a.relu_out = 1;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int it = 0; it < 3; ++it) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((conv_wino_kernel<2, false>), dim3(a.total_tiles, 1), dim3(256), 0, 0, a);
        hipEventRecord(e1); CK(hipEventSynchronize(e1));
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("launch %d: %.3f ms, %d blocks (%.1f TFLOP/s-equivalent)\n", it, ms, a.total_tiles, 2.0 * H * W * 9 * C * C / (ms * 1e-3) / 1e12);
    }
    static unsigned long long tl[512][4][32];
    CK(hipMemcpyFromSymbol(tl, HIP_SYMBOL(wino_tl), sizeof(tl)));
    const char* names[20] = {"start", "patch(0) requested", "transform(0) written", "barrier", "g0 patch(1) requested", "g0 p0", "g0 p1", "g0 transform(1)",
                             "g0 p2", "g0 p3", "g0 barrier", "g1 (no request)", "g1 p0", "g1 p1", "g1 p2", "g1 p3", "g1 barrier", "exchange written",
                             "barrier", "inverse + stores"};
    // median over the recorded blocks of each phase's duration (last wave of the block), and of the block lifetime
    printf("phase durations, cycles (median over 512 steady-state blocks; per block: last wave's stamp - previous phase's last stamp)\n");
    std::vector<double> life;
    for (int m = 1; m < 20; ++m) {
        std::vector<long long> dv;
        for (int b = 0; b < 512; ++b) {
            unsigned long long hi = 0, ph = 0;
            for (int w = 0; w < 4; ++w) { hi = std::max(hi, tl[b][w][m]); ph = std::max(ph, tl[b][w][m - 1]); }
            if (tl[b][0][0]) dv.push_back((long long)(hi - ph));
        }
        std::sort(dv.begin(), dv.end());
        if (!dv.empty()) printf("  %-24s %8lld   (p10 %lld, p90 %lld)\n", names[m], dv[dv.size() / 2], dv[dv.size() / 10], dv[dv.size() * 9 / 10]);
    }
    std::vector<long long> lv;
    for (int b = 0; b < 512; ++b) if (tl[b][0][0]) { unsigned long long hi = 0, lo = ~0ull; for (int w = 0; w < 4; ++w) { hi = std::max(hi, tl[b][w][19]); lo = std::min(lo, tl[b][w][0]); } lv.push_back((long long)(hi - lo)); }
    std::sort(lv.begin(), lv.end());
    if (!lv.empty()) printf("block lifetime: median %lld cycles (MFMA issue per wave: 2 groups x 64 x 32 = 4096)\n", lv[lv.size() / 2]);
    return 0;
}
