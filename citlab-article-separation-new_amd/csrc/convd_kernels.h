// ------------------------------------------------------------------------------------------------
// convd_kernel: the >= 64-channel 3x3 convolutions of the bf16 path (levels 3 and 4 of the ARU-Net: 32->64, 64->64, 64->128, 128->128, 128->64;
// ARU_v1.py:186-294) as ONE sixteen-wave block per CU whose stages are double-buffered by LDS-DMA (round 5).
//
// convb_kernel<3,3,2,2,2,8,2,.,8> serves these layers with two eight-wave blocks per CU on 8 x 32-pixel tiles: a stage = the halo tile of 32 input
// channels (21.8 KB, through registers) + the stage's A fragments (36.9 KB, LDS-DMA) in ONE buffer, so a block's MFMAs (72 per wave and stage, 2.5 k
// ticks) wait for a fill of 6-7 k ticks that only the OTHER block of the CU can cover (lesson 22: 0.28-0.34 of the bf16 peak, the largest gap of the
// page).  Here: 16 x 32-pixel tiles (the fragments serve twice the pixels: 76 KB per stage against 2 x 58 KB), sixteen waves (the same per-wave tile
// and registers: two m-tiles x four n-tiles), and TWO stage buffers -- halo AND fragments of stage g + 1 are copied global -> LDS by
// global_load_lds (no registers, no ds_write) while stage g multiplies; one barrier per stage.  What the copy cannot do is done around it: positions
// outside the image are zeroed behind the copy (the SAME padding), and a layer that reads ReLU(t) (convR_0 behind a pre-activation tensor) takes
// the ReLU at the fragment read.  Accumulation order per output = convb_kernel's (stages, then taps in order): bit-identical results.
// ------------------------------------------------------------------------------------------------
#pragma once
#include "bf16_kernels.h"

namespace asep {

constexpr int CD_TH = 16, CD_TW = 32, CD_NW = 16, CD_NTH = 64 * CD_NW;
struct ConvDLayout {
    static constexpr int LH = CD_TH + 2, LW = CD_TW + 2, PLANE = LH * LW * 32;          // two 16-channel planes of 32 bytes per pixel
    static constexpr int NU = LH * LW * 4;                                               // 16-byte units of a stage's halo tile (2448)
    static constexpr int NUP = (NU + 63) / 64 * 64;                                      // ... in whole wave-instructions of the copy (2496)
    static constexpr int NWU = 9 * 4 * 64;                                               // 16-byte units of a stage's A fragments [tap][m-tile][lane]
    static constexpr int W_OFF = NUP * 16, STAGE = W_OFF + NWU * 16, BYTES = 2 * STAGE;  // 39936 + 36864 = 76800 per buffer
};

// The block is PERSISTENT: it walks the work items (tile, 64-channel output block) i = blockIdx.x, + gridDim.x, ... and requests the NEXT item's
// first stage while the current item's last stage multiplies and its epilogue stores (one-shot blocks of this form -- one per CU -- left set-up,
// the first fill and the epilogue uncovered: 130 against convb_kernel's 110 us on the 64 -> 64 layers).  Items are numbered so that XCD x (blocks
// b with b & 7 == x; gridDim.x is a multiple of 8) walks the x-th eighth of the row-major tile list (XcdMap's bands): item i -> output block
// i / (8 chunk), band slot ti = i % (8 chunk), tile (ti & 7) chunk + (ti >> 3).
template <bool RESP, bool RIN>
__global__ __launch_bounds__(CD_NTH, 4) void convd_kernel(const ConvBArgs a, const int ny) {
    typedef ConvDLayout L;
    constexpr int TH = CD_TH, TW = CD_TW, NTH = CD_NTH, LW = L::LW, PLANE = L::PLANE, NU = L::NU;
    constexpr int MT = 2, NT = 4, MTB = 4;
    constexpr int NLOAD = (L::NUP + NTH - 1) / NTH, NWLOAD = (L::NWU + NTH - 1) / NTH;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int j = lane & 15, kk = lane >> 4;
    const int wm = wave & 1, wn = wave >> 1;                  // two waves along the output channels, eight along the pixels (two rows each)
    const int cout = a.cout, ngroups = a.groups;
    const int band = 8 * a.xm.chunk, nitems = ny * band;      // (run_convb always builds the band map for this kernel)

    // the thread's halo units u = tid + i * 1024 of a stage (LDS order = copy order: plane u / 1224, pixel (u % 1224) >> 1, half u & 1): tile
    // position and channel offset inside the stage -- the same for every item
    int sdesc[NLOAD];                                         // ly | lx << 8 | channel offset << 16 (one register per unit: the kernel runs at its register limit)
#pragma unroll
    for (int i = 0; i < NLOAD; ++i) {
        const int u = min(tid + i * NTH, NU - 1);
        const int pl = u / (PLANE / 16), r = u - pl * (PLANE / 16);
        const int pix = r >> 1, ly = pix / LW;
        sdesc[i] = ly | ((pix - ly * LW) << 8) | ((pl * 16 + (r & 1) * 8) << 16);
    }
    int nbase[NT];
#pragma unroll
    for (int n = 0; n < NT; ++n) {
        const int id = wn * NT + n;
        nbase[n] = ((id >> 1) * LW + (id & 1) * 16 + j) * 32 + (kk & 1) * 16 + (kk >> 1) * PLANE;
    }

    // item -> (problem, tile origin, output block); false for the band's padding slots
    auto locate = [&](int i, int& pi, int& x0, int& y0, int& mtb0) {
        const int yb = i / band, ti = i - yb * band;
        const int t = (ti & 7) * a.xm.chunk + (ti >> 3);
        if (t >= a.xm.total) return false;
        pi = prob_of_tile(a, t);
        const int tile = t - a.p[pi].tile_begin;
        const int ty = tile / a.p[pi].tiles_x, tx = tile - ty * a.p[pi].tiles_x;
        x0 = tx * TW; y0 = ty * TH; mtb0 = yb * MTB;
        return true;
    };
    auto next_item = [&](int i, int& pi, int& x0, int& y0, int& mtb0) {          // first valid item behind i of this block's walk, or -1
        for (i += gridDim.x; i < nitems; i += gridDim.x)
            if (locate(i, pi, x0, y0, mtb0)) return i;
        return -1;
    };
    // clamped image pixel (always a valid address) + inside-the-image mask of the thread's halo units for an item
    auto describe = [&](int pi, int x0, int y0, int (&spix)[NLOAD], unsigned& inmask) {
        const int H = a.p[pi].H, W = a.p[pi].W;
        inmask = 0;
#pragma unroll
        for (int i = 0; i < NLOAD; ++i) {
            const int gy = y0 - 1 + (sdesc[i] & 0xff), gx = x0 - 1 + ((sdesc[i] >> 8) & 0xff);
            spix[i] = min(max(gy, 0), H - 1) * W + min(max(gx, 0), W - 1);
            inmask |= ((gy >= 0 && gy < H && gx >= 0 && gx < W) || tid + i * NTH >= NU ? 1u : 0u) << i;
        }
    };
    // stage g of an item: halo tile + A fragments, global -> LDS buffer `b`, whole wave-instructions (lane i to base + 16 i)
    auto copy_stage = [&](int pi, const int (&spix)[NLOAD], int mtb0, int g, int b) {
#if defined(CVD_ABL) && (CVD_ABL & 1)
        if (g >= 0) return;                                  // ablation: no copies at all (the LDS holds whatever it holds)
#endif
        const ConvBProb& P = a.p[pi];
        unsigned char* const buf = lds + b * L::STAGE;
#pragma unroll
        for (int i = 0; i < NLOAD; ++i) {
            const int u0 = i * NTH + wave * 64;              // wave-uniform
            if (u0 < L::NUP) {
                const int c = g * 32 + (sdesc[i] >> 16);
                const bf16_t* __restrict__ src = concat_src(P.in0, P.in1, c, a.c0);
                const int cs = c < a.c0 ? a.c0 : a.c1;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + (size_t)spix[i] * cs),
                                                 (__attribute__((address_space(3))) void*)(buf + u0 * 16), 16, 0, 0);
            }
        }
        const u32x4* __restrict__ wsrc = a.wpk + (size_t)mtb0 * 64;
        const size_t wstride = (size_t)a.mtiles * 64;
        const int mt_have = min(MTB, a.mtiles - mtb0);
#pragma unroll
        for (int i = 0; i < NWLOAD; ++i) {
            const int u0 = i * NTH + wave * 64;
            if (u0 < L::NWU) {
                const int u = u0 + lane;
                const int t = u / (MTB * 64), r = u - t * (MTB * 64);
                const u32x4* gsrc = wsrc + (size_t)(g * 9 + t) * wstride + min(r, mt_have * 64 - 1);
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                                 (__attribute__((address_space(3))) void*)(buf + L::W_OFF + u0 * 16), 16, 0, 0);
            }
        }
    };

    int item = (int)blockIdx.x - (int)gridDim.x, pi, x0, y0, mtb0;
    item = next_item(item, pi, x0, y0, mtb0);
    if (item < 0) return;
    int spix[NLOAD];
    unsigned inmask;
    describe(pi, x0, y0, spix, inmask);
    int sidx = 0;                                             // running stage count: buffer = sidx & 1
    copy_stage(pi, spix, mtb0, 0, 0);

    while (true) {
        const ConvBProb& P = a.p[pi];
        const int H = P.H, W = P.W, mt0 = mtb0 + wm * MT;
        // the residual operand: requested first, part of the accumulators' initial value (convb_kernel's RESP form)
        u32x2 resv[RESP ? MT : 1][RESP ? NT : 1];
        if constexpr (RESP) {
            const unsigned char* __restrict__ const rbase = reinterpret_cast<const unsigned char*>(P.res);
            const unsigned pxb = (unsigned)cout * 2u;
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                const int c = (mt0 + m) * 16 + kk * 4;
                const unsigned cb = c < cout ? (unsigned)c * 2u : 0u;
#pragma unroll
                for (int n = 0; n < NT; ++n) {
                    const int id = wn * NT + n;
                    const unsigned y = (unsigned)min(y0 + (id >> 1), H - 1), x = (unsigned)min(x0 + (id & 1) * 16 + j, W - 1);
                    resv[m][n] = *reinterpret_cast<const u32x2*>(rbase + ((y * (unsigned)W + x) * pxb + cb));
                }
            }
        }
        // the next item of this block's walk (its first stage is requested under this item's last stage)
        int piN = 0, x0N = 0, y0N = 0, mtb0N = 0;
        const int itemN = next_item(item, piN, x0N, y0N, mtb0N);
        f32x4 acc[MT][NT];
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            const int c = (mt0 + m) * 16 + kk * 4;
            const f32x4 b4 = c < cout ? *reinterpret_cast<const f32x4*>(a.bias + c) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int n = 0; n < NT; ++n) acc[m][n] = b4;
        }
        if constexpr (RESP) {
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
                for (int n = 0; n < NT; ++n) acc[m][n] += unpack_bf16x4(resv[m][n]);
        }

        for (int g = 0; g < ngroups; ++g, ++sidx) {
            unsigned char* const buf = lds + (sidx & 1) * L::STAGE;
            const unsigned char* const wlds = buf + L::W_OFF;
            __builtin_amdgcn_s_waitcnt(0x0f70);              // vmcnt(0): this wave's copies of the stage have landed
            if (~inmask & ((1u << NLOAD) - 1u)) {            // SAME padding: positions outside the image hold zeros
#pragma unroll
                for (int i = 0; i < NLOAD; ++i)
                    if (!((inmask >> i) & 1u)) *reinterpret_cast<u32x4*>(buf + (tid + i * NTH) * 16) = u32x4{0u, 0u, 0u, 0u};
            }
            __syncthreads();                                 // every wave's copies of this stage; every wave past the previous stage's MFMAs
            if (g + 1 < ngroups) copy_stage(pi, spix, mtb0, g + 1, (sidx + 1) & 1);          // (that buffer was the previous stage's)
            else if (itemN >= 0) {                           // (its unit table lives only here: recomputed when the item becomes the current one)
                int spixN[NLOAD];
                unsigned inmaskN;
                describe(piN, x0N, y0N, spixN, inmaskN);
                copy_stage(piN, spixN, mtb0N, 0, (sidx + 1) & 1);
            }
            auto chunk = [&](int t, int toff) {
                u32x4 af[MT], bfr[NT];
#pragma unroll
                for (int m = 0; m < MT; ++m) af[m] = *reinterpret_cast<const u32x4*>(wlds + ((t * MTB + wm * MT + m) * 64 + lane) * 16);
#pragma unroll
                for (int n = 0; n < NT; ++n) {
                    bfr[n] = *reinterpret_cast<const u32x4*>(buf + nbase[n] + toff);
                    if constexpr (RIN) bfr[n] = relu_bf16x8(bfr[n]);
                }
#pragma unroll
                for (int m = 0; m < MT; ++m)
#pragma unroll
#if defined(CVD_ABL) && (CVD_ABL & 2)
                    for (int n = 0; n < NT; ++n) acc[m][n].x += __uint_as_float((af[m].x ^ bfr[n].x) & 0x3f800000u);   // ablation: no MFMAs
#else
                    for (int n = 0; n < NT; ++n) acc[m][n] = mfma_bf16_k32(af[m], bfr[n], acc[m][n]);
#endif
            };
#pragma unroll 1
            for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) chunk(ky * 3 + kx, (ky * LW + kx) * 32);
        }

        // ---- epilogue (convb_kernel's): lane = pixel (column block, j), 4 consecutive output channels 16 (mt0 + m) + 4 kk ----
        const int Wp = (W + 1) >> 1;
        if (y0 + TH <= H && x0 + TW <= W && (mtb0 + MTB) * 16 <= cout && a.relu_out && !a.pool_f32) {
            const size_t lane0 = ((size_t)(y0 + wn * (NT / 2)) * W + x0 + j) * cout + mt0 * 16 + kk * 4;
            bf16_t* __restrict__ ob = P.out + lane0;
            const bf16_t* __restrict__ rb = P.res + lane0;
            bf16_t* __restrict__ pb = (bf16_t*)P.pool + ((size_t)((y0 >> 1) + wn * (NT / 4)) * Wp + ((x0 + j) >> 1)) * cout + mt0 * 16 + kk * 4;
            const size_t rs = (size_t)W * cout, prs = (size_t)Wp * cout;
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                u32x2 pk[NT];
#pragma unroll
                for (int n = 0; n < NT; ++n) {
                    const size_t off = (size_t)(n >> 1) * rs + (n & 1) * 16 * cout + m * 16;
                    f32x4 v = acc[m][n];
                    if constexpr (!RESP) { if (P.res) v += unpack_bf16x4(*reinterpret_cast<const u32x2*>(rb + off)); }
                    const u32x2 q = pack_bf16x4(v);
                    pk[n] = u32x2{relu_bf16x2(q.x), relu_bf16x2(q.y)};
                    if (!a.skip_full) *reinterpret_cast<u32x2*>(ob + off) = pk[n];
                }
                if (P.pool) {
#pragma unroll
                    for (int n = 0; n < NT; ++n) {
                        if (n & 2) continue;
                        u32x2 mm = u32x2{pkmax_u16(pk[n].x, pk[n + 2].x), pkmax_u16(pk[n].y, pk[n + 2].y)};
                        mm = u32x2{pkmax_u16(mm.x, __float_as_uint(lane_xor1(__uint_as_float(mm.x)))), pkmax_u16(mm.y, __float_as_uint(lane_xor1(__uint_as_float(mm.y))))};
                        if ((j & 1) == 0) *reinterpret_cast<u32x2*>(pb + (size_t)(n >> 2) * prs + (n & 1) * 8 * cout + m * 16) = mm;
                    }
                }
            }
        } else {
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                const int c = (mt0 + m) * 16 + kk * 4;
                const bool cok = c < cout;
#pragma unroll
                for (int n = 0; n < NT; ++n) {
                    const int id = wn * NT + n;
                    const int y = y0 + (id >> 1), x = x0 + (id & 1) * 16 + j;
                    const bool ok = cok && y < H && x < W;
                    const size_t p = ((size_t)min(y, H - 1) * W + min(x, W - 1)) * cout + (cok ? c : 0);
                    f32x4 v = acc[m][n];
                    if constexpr (!RESP) { if (P.res) v += unpack_bf16x4(*reinterpret_cast<const u32x2*>(P.res + p)); }
                    if (a.relu_out) v = relu4(v);
                    else if (a.act) v = act4(v, a.act);
                    const u32x2 pk = pack_bf16x4(v);
                    acc[m][n] = unpack_bf16x4(pk);
                    if (ok && !a.skip_full) *reinterpret_cast<u32x2*>(P.out + p) = pk;
                }
                if (P.pool) {
#pragma unroll
                    for (int n = 0; n < NT; ++n) {
                        if (n & 2) continue;
                        const int id = wn * NT + n;
                        const int y = y0 + (id >> 1), x = x0 + (id & 1) * 16 + j;
                        f32x4 mm = (y + 1 < H) ? max4(acc[m][n], acc[m][n + 2]) : acc[m][n];
                        const f32x4 nb = f32x4{lane_xor1(mm.x), lane_xor1(mm.y), lane_xor1(mm.z), lane_xor1(mm.w)};
                        if (x + 1 < W) mm = max4(mm, nb);
                        if ((j & 1) == 0 && cok && y < H && x < W) {
                            const size_t q = ((size_t)(y >> 1) * Wp + (x >> 1)) * cout + c;
                            if (a.pool_f32) *reinterpret_cast<f32x4*>((float*)P.pool + q) = mm;
                            else *reinterpret_cast<u32x2*>((bf16_t*)P.pool + q) = pack_bf16x4(mm);
                        }
                    }
                }
            }
        }
        if (itemN < 0) break;
        item = itemN; pi = piN; x0 = x0N; y0 = y0N; mtb0 = mtb0N;
        describe(pi, x0, y0, spix, inmask);
    }
}

}  // namespace asep
