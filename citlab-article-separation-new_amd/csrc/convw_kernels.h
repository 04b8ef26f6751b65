// ------------------------------------------------------------------------------------------------
// convw_kernel: the 3x3 convolutions with 64 output and 32 / 64 input channels of the bf16 path (level 3 of the ARU-Net: 32->64, 64->64;
// ARU_v1.py:186-294) with the layer's WHOLE filter resident in LDS (round 5).
//
// convb_kernel<3,3,2,2,2,8,2,.,8> refetches a stage's A fragments (36.9 KB) for every 8 x 32-pixel tile; its ablations (lesson 48) say the fills
// are 37 % of these layers and that two blocks per CU cannot cover a 3 us fill with 1.1 us of MFMAs per block-stage.  Here one persistent eight-wave
// block per CU copies the layer's fragments (<= 2 stages x 36.9 KB) ONCE, then walks its tiles: the halo tile of ALL input channels of the NEXT tile
// (<= 2 x 21.8 KB) is copied global -> LDS by global_load_lds into the other of two halo buffers while the current tile multiplies -- a prefetch
// distance of a whole tile (both stages and the epilogue) instead of a stage, no fragment traffic, one barrier per tile.  Positions outside the image
// are zeroed behind the copy (SAME padding); a layer that reads ReLU(t) takes the ReLU at the fragment read.  Accumulation order per output =
// convb_kernel's (stages, then taps in order): bit-identical results.
// ------------------------------------------------------------------------------------------------
#pragma once
#include "bf16_kernels.h"

namespace asep {

constexpr int CW_TH = 8, CW_TW = 32, CW_NW = 8, CW_NTH = 64 * CW_NW;
struct ConvWLayout {
    static constexpr int LH = CW_TH + 2, LW = CW_TW + 2, PLANE = LH * LW * 32;          // two 16-channel planes of 32 bytes per pixel and stage
    static constexpr int NUS = LH * LW * 4;                                              // 16-byte units of one stage's halo tile (1360)
    static constexpr int NUMAX = (2 * NUS + 63) / 64 * 64;                               // both stages, whole wave-instructions of the copy (2752)
    static constexpr int NWS = 9 * 4 * 64;                                               // 16-byte units of a stage's A fragments [tap][m-tile][lane]
    static constexpr int H_OFF = 2 * NWS * 16, HB = NUMAX * 16, BYTES = H_OFF + 2 * HB;  // 73728 + 2 x 44032 = 161792
};

// Items (tiles) are numbered so that XCD x (blocks b with b & 7 == x; gridDim.x is a multiple of 8) walks the x-th eighth of the row-major tile list
// (XcdMap's bands): item i -> tile (i & 7) chunk + (i >> 3).
template <bool RESP, bool RIN>
__global__ __launch_bounds__(CW_NTH, 2) void convw_kernel(const ConvBArgs a) {
    typedef ConvWLayout L;
    constexpr int TH = CW_TH, TW = CW_TW, NTH = CW_NTH, LW = L::LW, PLANE = L::PLANE, NUS = L::NUS;
    constexpr int MT = 2, NT = 4, MTB = 4;
    constexpr int NLOAD = (L::NUMAX + NTH - 1) / NTH;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int j = lane & 15, kk = lane >> 4;
    const int wm = wave & 1, wn = wave >> 1;                  // two waves along the output channels, four along the pixels (two rows each)
    const int cout = a.cout, ngroups = a.groups;
    const int nitems = 8 * a.xm.chunk;                        // (run_convb always builds the band map for this kernel)
    const int nut = ngroups * NUS, nutp = (nut + 63) / 64 * 64;   // the item's halo units (all stages), padded to whole wave-instructions

    // ---- once per block: the layer's A fragments -> LDS ----
    for (int u0 = wave * 64; u0 < ngroups * L::NWS; u0 += NTH) {
        const int u = u0 + lane;
        const int t = u / (MTB * 64), r = u - t * (MTB * 64);    // t = stage * 9 + tap
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(a.wpk + (size_t)t * a.mtiles * 64 + min(r, a.mtiles * 64 - 1)),
                                         (__attribute__((address_space(3))) void*)(lds + u0 * 16), 16, 0, 0);
    }

    // the thread's halo units u = tid + i * 512 of an item (LDS order = copy order: stage u / 1360, plane (u % 1360) / 680, pixel (.. % 680) >> 1,
    // half u & 1): tile position and channel -- the same for every item (one register per unit)
    int sdesc[NLOAD];                                         // ly | lx << 8 | channel << 16
#pragma unroll
    for (int i = 0; i < NLOAD; ++i) {
        const int u = min(tid + i * NTH, nut - 1);
        const int st = u / NUS, r0 = u - st * NUS;
        const int pl = r0 / (PLANE / 16), r = r0 - pl * (PLANE / 16);
        const int pix = r >> 1, ly = pix / LW;
        sdesc[i] = ly | ((pix - ly * LW) << 8) | ((st * 32 + pl * 16 + (r & 1) * 8) << 16);
    }
    int nbase[NT];
#pragma unroll
    for (int n = 0; n < NT; ++n) {
        const int id = wn * NT + n;
        nbase[n] = ((id >> 1) * LW + (id & 1) * 16 + j) * 32 + (kk & 1) * 16 + (kk >> 1) * PLANE;
    }

    auto locate = [&](int i, int& pi, int& x0, int& y0) {
        const int t = (i & 7) * a.xm.chunk + (i >> 3);
        if (t >= a.xm.total) return false;
        pi = prob_of_tile(a, t);
        const int tile = t - a.p[pi].tile_begin;
        const int ty = tile / a.p[pi].tiles_x, tx = tile - ty * a.p[pi].tiles_x;
        x0 = tx * TW; y0 = ty * TH;
        return true;
    };
    auto next_item = [&](int i, int& pi, int& x0, int& y0) {   // first valid item behind i of this block's walk, or -1
        for (i += gridDim.x; i < nitems; i += gridDim.x)
            if (locate(i, pi, x0, y0)) return i;
        return -1;
    };
    auto describe = [&](int pi, int x0, int y0, int (&spix)[NLOAD], unsigned& inmask) {
        const int H = a.p[pi].H, W = a.p[pi].W;
        inmask = 0;
#pragma unroll
        for (int i = 0; i < NLOAD; ++i) {
            const int gy = y0 - 1 + (sdesc[i] & 0xff), gx = x0 - 1 + ((sdesc[i] >> 8) & 0xff);
            spix[i] = min(max(gy, 0), H - 1) * W + min(max(gx, 0), W - 1);
            inmask |= ((gy >= 0 && gy < H && gx >= 0 && gx < W) || tid + i * NTH >= nut ? 1u : 0u) << i;
        }
    };
    // an item's halo tile (all stages), global -> halo buffer b
    auto copy_halo = [&](int pi, const int (&spix)[NLOAD], int b) {
        const ConvBProb& P = a.p[pi];
        unsigned char* const buf = lds + L::H_OFF + b * L::HB;
#pragma unroll
        for (int i = 0; i < NLOAD; ++i) {
            const int u0 = i * NTH + wave * 64;              // wave-uniform
            if (u0 < nutp) {
                const int c = sdesc[i] >> 16;
                const bf16_t* __restrict__ src = concat_src(P.in0, P.in1, c, a.c0);
                const int cs = c < a.c0 ? a.c0 : a.c1;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + (size_t)spix[i] * cs),
                                                 (__attribute__((address_space(3))) void*)(buf + u0 * 16), 16, 0, 0);
            }
        }
    };

    int item = (int)blockIdx.x - (int)gridDim.x, pi, x0, y0;
    item = next_item(item, pi, x0, y0);
    if (item < 0) return;
    int spix[NLOAD];
    unsigned inmask;
    describe(pi, x0, y0, spix, inmask);
    int cur = 0;                                              // halo buffer of the current item
    copy_halo(pi, spix, 0);
    const int mtb0 = 0, mt0 = wm * MT;
    // the residual operand of an item (part of the accumulators' initial value, convb_kernel's RESP form): requested an item ahead
    u32x2 resv[RESP ? MT : 1][RESP ? NT : 1];
    auto request_res = [&](int pi_, int x0_, int y0_, u32x2 (&rv)[RESP ? MT : 1][RESP ? NT : 1]) {
        if constexpr (RESP) {
            const ConvBProb& Q = a.p[pi_];
            const unsigned char* __restrict__ const rbase = reinterpret_cast<const unsigned char*>(Q.res);
            const unsigned pxb = (unsigned)cout * 2u;
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                const int c = (mt0 + m) * 16 + kk * 4;
                const unsigned cb = c < cout ? (unsigned)c * 2u : 0u;
#pragma unroll
                for (int n = 0; n < NT; ++n) {
                    const int id = wn * NT + n;
                    const unsigned y = (unsigned)min(y0_ + (id >> 1), Q.H - 1), x = (unsigned)min(x0_ + (id & 1) * 16 + j, Q.W - 1);
                    rv[m][n] = *reinterpret_cast<const u32x2*>(rbase + ((y * (unsigned)Q.W + x) * pxb + cb));
                }
            }
        }
    };
    request_res(pi, x0, y0, resv);
    // the first item: its halo (and the fragments) landed, zeros written, visible to every wave
    __builtin_amdgcn_s_waitcnt(0x0f70);
    if (~inmask & ((1u << NLOAD) - 1u)) {
#pragma unroll
        for (int i = 0; i < NLOAD; ++i)
            if (!((inmask >> i) & 1u)) *reinterpret_cast<u32x4*>(lds + L::H_OFF + (tid + i * NTH) * 16) = u32x4{0u, 0u, 0u, 0u};
    }
    __syncthreads();

    while (true) {
        const ConvBProb& P = a.p[pi];
        const int H = P.H, W = P.W;
        // Per item: the NEXT item's halo (other buffer: every wave is past the MFMAs that read it, see the barrier below) and residual operand are
        // requested, this item multiplies, THEN the wave waits for those requests (the previous item's output stores are older and long done),
        // writes the next halo's zeros, meets the others, and only then stores this item's outputs -- which drain under the next item's MFMAs
        // (first cut: wait at the top of the item, i.e. right behind the epilogue's stores: an HBM write round trip exposed per item).
        int piN = 0, x0N = 0, y0N = 0;
        const int itemN = next_item(item, piN, x0N, y0N);
        unsigned char* const buf = lds + L::H_OFF + cur * L::HB;
        unsigned inmaskN = ~0u;
        u32x2 resvN[RESP ? MT : 1][RESP ? NT : 1];
        if (itemN >= 0) {
            int spixN[NLOAD];
            describe(piN, x0N, y0N, spixN, inmaskN);
            copy_halo(piN, spixN, cur ^ 1);
            request_res(piN, x0N, y0N, resvN);
        }
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
        for (int g = 0; g < ngroups; ++g) {
            const unsigned char* const hb = buf + g * (NUS * 16);
            const unsigned char* const wlds = lds + g * (L::NWS * 16);
            // software pipeline over the nine taps: tap t + 1's six fragments are requested before tap t's eight MFMAs (two waves per SIMD do
            // not cover an LDS round trip per tap; the registers are there: one block per CU)
            struct Frag { u32x4 af[MT], bfr[NT]; };
            auto request = [&](Frag& f, int t) {
                const int toff = ((t / 3) * LW + (t % 3)) * 32;
#pragma unroll
                for (int m = 0; m < MT; ++m) f.af[m] = *reinterpret_cast<const u32x4*>(wlds + ((t * MTB + wm * MT + m) * 64 + lane) * 16);
#pragma unroll
                for (int n = 0; n < NT; ++n) f.bfr[n] = *reinterpret_cast<const u32x4*>(hb + nbase[n] + toff);
            };
            Frag fr[2];
            request(fr[0], 0);
            static_for<9>([&](auto tc) {
                constexpr int t = decltype(tc)::value;
                if constexpr (t + 1 < 9) request(fr[(t + 1) & 1], t + 1);
                __builtin_amdgcn_sched_barrier(0);
                Frag& f = fr[t & 1];
#pragma unroll
                for (int n = 0; n < NT; ++n)
                    if constexpr (RIN) f.bfr[n] = relu_bf16x8(f.bfr[n]);
#pragma unroll
                for (int m = 0; m < MT; ++m)
#pragma unroll
                    for (int n = 0; n < NT; ++n) acc[m][n] = mfma_bf16_k32(f.af[m], f.bfr[n], acc[m][n]);
                __builtin_amdgcn_sched_barrier(0);
            });
        }

        __builtin_amdgcn_s_waitcnt(0x0f70);                  // vmcnt(0): the next item's halo and residual operand have landed
        if (~inmaskN & ((1u << NLOAD) - 1u)) {               // SAME padding: positions outside the image hold zeros
            unsigned char* const bufN = lds + L::H_OFF + (cur ^ 1) * L::HB;
#pragma unroll
            for (int i = 0; i < NLOAD; ++i)
                if (!((inmaskN >> i) & 1u)) *reinterpret_cast<u32x4*>(bufN + (tid + i * NTH) * 16) = u32x4{0u, 0u, 0u, 0u};
        }
        __syncthreads();                                     // every wave's copies of the next item; every wave past this item's MFMAs

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
        item = itemN; pi = piN; x0 = x0N; y0 = y0N; cur ^= 1;
        if constexpr (RESP) {
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
                for (int n = 0; n < NT; ++n) resv[m][n] = resvN[m][n];
        }
    }
}

}  // namespace asep
