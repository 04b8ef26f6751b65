// fp32 convolutions whose PRODUCTS run on the bf16 matrix pipeline (asep_aru_cfg.compute_dtype = 2, "f32 split").
//
// gfx950 multiplies fp32 at 157 TFLOP/s (v_mfma_f32_16x16x4_f32, v_pk_fma_f32) and bf16 at 2.5 PFLOP/s with fp32 accumulation.  An fp32
// number is EXACTLY the sum of three bfloat16 numbers (8 + 8 + 8 significand bits, round-to-nearest at every cut, same exponent range as
// fp32: no scaling, no overflow case):      x = xh + xm + xl,   w = wh + wm + wl.
// A product x w is then the sum of nine bf16 products; the six largest
//        xh wh  +  (xh wm + xm wh)  +  (xm wm + xh wl + xl wh)
// leave out terms of at most 2 * 2^-25 |x w| (xm wl, xl wm, xl wl) -- below the rounding of ONE fp32 multiply-add -- and each of them is
// computed without a rounding of its own (8 x 8 significand bits) and added in fp32 by v_mfma_f32_16x16x32_bf16.  Six MFMAs of 16 x 16 x 32
// replace eight v_mfma_f32_16x16x4_f32 at 1/16 of their cost each: 2.7 x the fp32 matrix rate, no Winograd transform, and the
// tensors in HBM stay fp32 NHWC -- a layer of this file can stand anywhere between layers of aru_kernels.h.
//
// Reference semantics (file:line in /root/reference): layers.py:191-247 conv2d (SAME, stride 1) + bias + activation, ARU_v1.py:212-227
// (residual add before the activation), the same as conv_mfma_kernel of aru_kernels.h.
//
// Data path of a block (TH x 32 output pixels, 4 waves, MT m-tiles of 16 output channels):
//   stage g (32 input channels, or the layer's 16): the halo tile is read as fp32, cut into its three bf16 parts on the way
//   (13 vector instructions per pair of values) and stored as three sets of 16-channel planes (32 bytes per pixel: the ds_read_b128
//   lane groups of gfx950 hit distinct banks, bf16_kernels.h); per K chunk (a tap x 32 channels, or two taps x 16) a wave reads its
//   B fragments of the three parts (one ds_read_b128 each), holds the A fragments of the three weight parts -- fetched from L2 one
//   chunk ahead, no LDS copy: with 6 MFMAs per fragment pair the vector memory path has the time -- and issues 6 MT NT MFMAs.
// Operand layout of v_mfma_f32_16x16x32_bf16: lane l holds A[row l & 15][k = 8 (l >> 4) + i], B[k = 8 (l >> 4) + i][col l & 15], i = 0..7;
// D: col = l & 15 (pixel), rows 4 (l >> 4) + r (output channel).
#pragma once
#include "bf16_kernels.h"
#include "res8_kernels.h"

namespace asep {

// x (fp32) -> its three bfloat16 parts, for a pair of values: packed {a, b} words of the high, middle and low part
__device__ __forceinline__ void split3_pair(float a, float b, unsigned& h, unsigned& m, unsigned& l) {
    h = pack_bf16x2(a, b);
    const f32x2 r = psub(f32x2{a, b}, f32x2{__uint_as_float(h << 16), __uint_as_float(h & 0xffff0000u)});      // exact; one v_pk_add_f32
    m = pack_bf16x2(r.x, r.y);
    const f32x2 q = psub(r, f32x2{__uint_as_float(m << 16), __uint_as_float(m & 0xffff0000u)});                // exact
    l = pack_bf16x2(q.x, q.y);
}
__device__ __forceinline__ void split3_x8(f32x4 v0, f32x4 v1, u32x4& h, u32x4& m, u32x4& l) {
    unsigned hh[4], mm[4], ll[4];
    split3_pair(v0.x, v0.y, hh[0], mm[0], ll[0]);
    split3_pair(v0.z, v0.w, hh[1], mm[1], ll[1]);
    split3_pair(v1.x, v1.y, hh[2], mm[2], ll[2]);
    split3_pair(v1.z, v1.w, hh[3], mm[3], ll[3]);
    h = u32x4{hh[0], hh[1], hh[2], hh[3]};
    m = u32x4{mm[0], mm[1], mm[2], mm[3]};
    l = u32x4{ll[0], ll[1], ll[2], ll[3]};
}

// ------------------------------------------------------------------------------------------------
// convs_kernel: stride-1 SAME convolution (3x3 or 4x4), fp32 NHWC in / out, optional channel concat [in0, in1], residual, activation,
// 2x2 max pool of the output -- ConvArgs as conv_mfma_kernel takes them; a.wpk = the split filter of pack_conv_split:
//   [stage][chunk][part h, m, l][m-tile][lane] x 16 bytes.
//   C16 = true:  cin <= 16 (12 or 16; one stage, one 32-byte plane per part; K chunk = two consecutive taps x 16 channels)
//   C16 = false: cin % 32 == 0 (stages of 32 channels = two planes per part; K chunk = one tap x 32 channels)
// ------------------------------------------------------------------------------------------------
template <int KH, int KW, bool C16, int MT, int TH, int MINB>
__global__ __launch_bounds__(256, MINB) void convs_kernel(const ConvArgs a) {
    constexpr int TW = 32, NT = TH * 2 / 4, NH = NT > 4 ? NT / 2 : NT;          // n-tiles of a wave; NH at a time in registers
    static_assert(NT % 4 == 0, "a wave owns whole row pairs (fused 2x2 pool)");
    constexpr int LH = TH + KH - 1, LW = TW + KW - 1;
    constexpr int PT = (KH - 1) / 2, PL = (KW - 1) / 2;                         // TF SAME: pad_before = (k-1)/2
    constexpr int TAPS = KH * KW;
    constexpr int SUBS = C16 ? 2 : 4;                                           // 8-channel units per pixel and stage
    constexpr int PLANE = LH * LW * 32, NPL = C16 ? 1 : 2, PART = NPL * PLANE;
    constexpr int CPS = C16 ? (TAPS + 1) / 2 : TAPS;                            // K chunks per stage
    constexpr int NU = LH * LW * SUBS, NLOAD = (NU + 255) / 256;
    static_assert(3 * PART <= 65536, "halo tile of the three parts in 64 KB");
    __shared__ __attribute__((aligned(16))) unsigned char lds[3 * PART];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int j = lane & 15, kk = lane >> 4;
    const int bid = sched_tile(a.xm);
    if (bid < 0) return;
    int pi = 0;
    pi = prob_of_tile(a, bid);
    const ConvProb& P = a.p[pi];
    const int tile = bid - P.tile_begin;
    const int ty = tile / P.tiles_x, tx = tile - ty * P.tiles_x;
    const int x0 = tx * TW, y0 = ty * TH, mt0 = blockIdx.y * MT;
    const int H = P.H, W = P.W, cout = a.cout;

    f32x4 acc[MT][NT];
#pragma unroll
    for (int m = 0; m < MT; ++m) {
        const int c = (mt0 + m) * 16 + kk * 4;
        const f32x4 b4 = c < cout ? *reinterpret_cast<const f32x4*>(a.bias + c) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int n = 0; n < NT; ++n) acc[m][n] = b4;
    }
    int nbase[NT];
#pragma unroll
    for (int n = 0; n < NT; ++n) {
        const int id = wave * NT + n;
        nbase[n] = ((id >> 1) * LW + (id & 1) * 16 + j) * 32 + (kk & 1) * 16 + (C16 ? 0 : (kk >> 1) * PLANE);
    }
    // the residual operand joins the accumulators' initial value: its loads are in flight together with the first halo tile
    // (added in the epilogue their latency was exposed: 280 against 167 us for a 16-channel layer without one)
    if (P.res) {
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            const int c = (mt0 + m) * 16 + kk * 4;
            if (c < cout) {
#pragma unroll
                for (int n = 0; n < NT; ++n) {
                    const int id = wave * NT + n;
                    const int y = min(y0 + (id >> 1), H - 1), x = min(x0 + (id & 1) * 16 + j, W - 1);
                    acc[m][n] += *reinterpret_cast<const f32x4*>(P.res + ((size_t)y * W + x) * cout + c);
                }
            }
        }
    }
    // A fragments of chunk q (all stages in one sequence), part s, m-tile m: one 16-byte load per lane, L2-resident
    const int nchunks = (C16 ? 1 : a.groups) * CPS;
    const u32x4* __restrict__ wsrc = reinterpret_cast<const u32x4*>(a.wpk) + (size_t)mt0 * 64 + lane;
    const size_t wpart = (size_t)a.mtiles * 64;
    u32x4 af[3][MT], afn[3][MT];
    auto fetch = [&](int q, u32x4 (&dst)[3][MT]) {
        const u32x4* __restrict__ s0 = wsrc + (size_t)q * 3 * wpart;
#pragma unroll
        for (int s = 0; s < 3; ++s)
#pragma unroll
            for (int m = 0; m < MT; ++m) dst[s][m] = s0[s * wpart + m * 64];
    };
    fetch(0, af);

    // halo loader slots (pixel, 8-channel unit): image pixel (clamped: always a valid address), LDS offset, inside-the-image mask
    int spix[NLOAD], slds[NLOAD];
    unsigned stmask = 0;
#pragma unroll
    for (int i = 0; i < NLOAD; ++i) {
        const int u = min(tid + i * 256, NU - 1);
        const int pix = u / SUBS, sub = u - pix * SUBS;
        const int ly = pix / LW, lx = pix - ly * LW;
        const int gy = y0 - PT + ly, gx = x0 - PL + lx;
        spix[i] = min(max(gy, 0), H - 1) * W + min(max(gx, 0), W - 1);
        slds[i] = (sub >> 1) * PLANE + pix * 32 + (sub & 1) * 16;
        stmask |= ((gy >= 0 && gy < H && gx >= 0 && gx < W && tid + i * 256 < NU) ? 1u : 0u) << i;
    }
    const int sub0 = (tid % SUBS) * 8;                         // (256 is a multiple of SUBS: the same unit for all slots)
    const int cin = a.c0 + a.c1;

    int q = 0;
    const int ngroups = C16 ? 1 : a.groups;
    for (int g = 0; g < ngroups; ++g) {
        {
            const int c = g * 32 + sub0;
            const bool from0 = c < a.c0;
            const float* __restrict__ src = from0 ? P.in0 + c : P.in1 + (c - a.c0);
            const int cs = from0 ? a.c0 : a.c1;
            const bool second = c + 4 < cin;                  // cin = 12: the unit 8..15 holds four channels
            f32x4 st0[NLOAD], st1[NLOAD];
#pragma unroll
            for (int i = 0; i < NLOAD; ++i) {
                const float* p = src + (size_t)spix[i] * cs;
                st0[i] = *reinterpret_cast<const f32x4*>(p);
                st1[i] = second ? *reinterpret_cast<const f32x4*>(p + 4) : f32x4{0.f, 0.f, 0.f, 0.f};
            }
            if (g > 0) __syncthreads();                       // the previous stage's readers are done
#pragma unroll
            for (int i = 0; i < NLOAD; ++i) {
                if (i * 256 + 255 < NU || tid + i * 256 < NU) {
                    f32x4 v0 = st0[i], v1 = st1[i];
                    if (!((stmask >> i) & 1u)) v0 = v1 = f32x4{0.f, 0.f, 0.f, 0.f};
                    if (a.relu_in) { v0 = relu4(v0); v1 = relu4(v1); }
                    u32x4 h, mm, l;
                    if (a.dbg & 2) { h = mm = l = __builtin_bit_cast(u32x4, v0); }
                    else split3_x8(v0, v1, h, mm, l);
                    *reinterpret_cast<u32x4*>(lds + slds[i]) = h;
                    *reinterpret_cast<u32x4*>(lds + PART + slds[i]) = mm;
                    *reinterpret_cast<u32x4*>(lds + 2 * PART + slds[i]) = l;
                }
            }
            __syncthreads();
        }
        auto chunk = [&](int toff) {
            fetch((a.dbg & 1) ? 0 : min(q + 1, nchunks - 1), afn);
            __builtin_amdgcn_sched_barrier(0);                // the next chunk's A fragments are REQUESTED here, a chunk of MFMAs ahead of their use
#pragma unroll
            for (int nh = 0; nh < NT; nh += NH) {
                u32x4 bh[NH], bm[NH], bl[NH];
#pragma unroll
                for (int n = 0; n < NH; ++n) {
                    const unsigned char* p = lds + nbase[nh + n] + toff;
                    bh[n] = *reinterpret_cast<const u32x4*>(p);
                    bm[n] = *reinterpret_cast<const u32x4*>(p + PART);
                    bl[n] = *reinterpret_cast<const u32x4*>(p + 2 * PART);
                }
                if (a.dbg & 4) continue;
#pragma unroll
                for (int m = 0; m < MT; ++m)
#pragma unroll
                    for (int n = 0; n < NH; ++n) {
                        f32x4 c = acc[m][nh + n];
                        c = mfma_bf16_k32(af[2][m], bh[n], c);        // smallest terms first
                        c = mfma_bf16_k32(af[0][m], bl[n], c);
                        c = mfma_bf16_k32(af[1][m], bm[n], c);
                        c = mfma_bf16_k32(af[1][m], bh[n], c);
                        c = mfma_bf16_k32(af[0][m], bm[n], c);
                        c = mfma_bf16_k32(af[0][m], bh[n], c);
                        acc[m][nh + n] = c;
                    }
            }
#pragma unroll
            for (int s = 0; s < 3; ++s)
#pragma unroll
                for (int m = 0; m < MT; ++m) af[s][m] = afn[s][m];
            ++q;
        };
        if constexpr (C16) {
#pragma unroll
            for (int t = 0; t < CPS; ++t) {
                const int ta = 2 * t, tb = 2 * t + 1 < TAPS ? 2 * t + 1 : TAPS - 1;       // padded slot: zero weights, finite data
                const int oa = ((ta / KW) * LW + ta % KW) * 32, ob = ((tb / KW) * LW + tb % KW) * 32;
                chunk((kk >> 1) ? ob : oa);
            }
        } else {
#pragma unroll 1
            for (int ky = 0; ky < KH; ++ky)
#pragma unroll
                for (int kx = 0; kx < KW; ++kx) chunk((ky * LW + kx) * 32);
        }
    }

    // ---- epilogue: lane = pixel (column block, j), 4 consecutive output channels 16 (mt0 + m) + 4 kk ----
    const int Wp = (W + 1) >> 1;
#pragma unroll
    for (int m = 0; m < MT; ++m) {
        const int c = (mt0 + m) * 16 + kk * 4;
        const bool cok = c < cout;                            // cout is a multiple of 4
#pragma unroll
        for (int n = 0; n < NT; ++n) {
            const int id = wave * NT + n;
            const int y = y0 + (id >> 1), x = x0 + (id & 1) * 16 + j;
            const bool ok = cok && y < H && x < W;
            const size_t p = ((size_t)min(y, H - 1) * W + min(x, W - 1)) * cout + (cok ? c : 0);
            f32x4 v = acc[m][n];
            if (a.relu_out) v = relu4(v);
            else if (a.act) v = act4(v, a.act);
            acc[m][n] = v;
            if (ok && !a.skip_full) *reinterpret_cast<f32x4*>(P.out + p) = v;
        }
        if (P.pool) {
            // n-tiles n, n + 2 of a wave are the same 16 columns of rows y, y + 1 (y even); column partner in lane j ^ 1
#pragma unroll
            for (int n = 0; n < NT; ++n) {
                if (n & 2) continue;
                const int id = wave * NT + n;
                const int y = y0 + (id >> 1), x = x0 + (id & 1) * 16 + j;
                pool2_store(acc[m][n], acc[m][n + 2], y + 1 < H, x + 1 < W, (j & 1) == 0 && cok && y < H && x < W,
                            P.pool + ((size_t)(y >> 1) * Wp + (x >> 1)) * cout + (cok ? c : 0));
            }
        }
    }
}


// ------------------------------------------------------------------------------------------------
// convs16_kernel: the 3x3 layers with >= 32 input channels, A fragments through LDS.
// convs_kernel's waves fetch their A fragments from L2 themselves: with all output channels of a 64-channel layer in every wave that is
// 12 KB per wave and chunk, 8 waves per CU = 64 B per clock -- the whole throughput of the CU's vector memory path, hit or miss (timing
// experiments: without any MFMA the kernel still took 60 % of its time, scripts/r4_convs_dbg.sh).  Here the block copies a chunk's fragments to
// LDS ONCE (global_load_lds, no registers, double buffered, one barrier per chunk) and its four waves read them from there; to make
// room the stages are 16 input channels wide (K chunk = two taps x 16 channels, 5 chunks for the 9 taps: 10 % of the MFMA slots idle)
// and the next stage's halo tile is requested while the current one is multiplied.
//   a.wpk = pack_conv_split mode 3: [stage (16 channels)][chunk 5][part 3][m-tile][lane] x 16 bytes.
// ------------------------------------------------------------------------------------------------
template <int MT, int TH, int MINB, bool ALDS>
__global__ __launch_bounds__(256, MINB) void convs16_kernel(const ConvArgs a) {
    constexpr int KH = 3, KW = 3, TW = 32, NT = TH * 2 / 4;
    static_assert(NT == 4, "8 x 32 tiles: a wave owns two rows");
    constexpr int LH = TH + KH - 1, LW = TW + KW - 1, PT = 1, PL = 1, TAPS = 9, CPS = 5;
    constexpr int PART = LH * LW * 32;                                          // one part: a 32-byte plane (16 channels) per pixel
    constexpr int NU = LH * LW * 2, NLOAD = (NU + 255) / 256;                   // 8-channel units of a stage's halo tile
    constexpr int ABUF = ALDS ? 3 * MT * 1024 : 0;                              // a chunk's A fragments: [part][m-tile][lane] x 16 bytes
    __shared__ __attribute__((aligned(16))) unsigned char lds[3 * PART + 2 * ABUF + 16];
    unsigned char* const albs = lds + 3 * PART;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int j = lane & 15, kk = lane >> 4;
    const int bid = sched_tile(a.xm);
    if (bid < 0) return;
    int pi = 0;
    pi = prob_of_tile(a, bid);
    const ConvProb& P = a.p[pi];
    const int tile = bid - P.tile_begin;
    const int ty = tile / P.tiles_x, tx = tile - ty * P.tiles_x;
    const int x0 = tx * TW, y0 = ty * TH, mt0 = blockIdx.y * MT;
    const int H = P.H, W = P.W, cout = a.cout;

    f32x4 acc[MT][NT];
#pragma unroll
    for (int m = 0; m < MT; ++m) {
        const int c = (mt0 + m) * 16 + kk * 4;
        const f32x4 b4 = c < cout ? *reinterpret_cast<const f32x4*>(a.bias + c) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int n = 0; n < NT; ++n) acc[m][n] = b4;
    }
    int nbase[NT];
#pragma unroll
    for (int n = 0; n < NT; ++n) {
        const int id = wave * NT + n;
        nbase[n] = ((id >> 1) * LW + (id & 1) * 16 + j) * 32 + (kk & 1) * 16;
    }
    if (P.res) {                                              // the residual joins the accumulators' initial value (see convs_kernel)
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            const int c = (mt0 + m) * 16 + kk * 4;
            if (c < cout) {
#pragma unroll
                for (int n = 0; n < NT; ++n) {
                    const int id = wave * NT + n;
                    const int y = min(y0 + (id >> 1), H - 1), x = min(x0 + (id & 1) * 16 + j, W - 1);
                    acc[m][n] += *reinterpret_cast<const f32x4*>(P.res + ((size_t)y * W + x) * cout + c);
                }
            }
        }
    }
    // chunk q's A fragments -> LDS buffer q & 1: unit f = part * MT + m (1 KB = one wave-wide copy), dealt to the waves
    const int nstages = a.groups, nchunks = nstages * CPS;
    const u32x4* __restrict__ wsrc = reinterpret_cast<const u32x4*>(a.wpk) + (size_t)mt0 * 64 + lane;
    const size_t wpart = (size_t)a.mtiles * 64;
    // ALDS = false: every wave fetches its fragments itself, one chunk ahead (convs_kernel's form; no barrier inside a stage)
    u32x4 afr[ALDS ? 1 : 3][ALDS ? 1 : MT], afn[ALDS ? 1 : 3][ALDS ? 1 : MT];
    auto fetch = [&](int q, auto& dst) {
        const u32x4* __restrict__ s0 = wsrc + (size_t)q * 3 * wpart;
#pragma unroll
        for (int s = 0; s < 3; ++s)
#pragma unroll
            for (int m = 0; m < MT; ++m) dst[s][m] = s0[s * wpart + m * 64];
    };
    auto copy_a = [&](int q) {
        if constexpr (!ALDS) return;
        const u32x4* __restrict__ s0 = wsrc + (size_t)q * 3 * wpart;
        unsigned char* const dst = albs + (q & 1) * ABUF;
#pragma unroll
        for (int f0 = 0; f0 < 3 * MT; f0 += 4) {
            const int f = f0 + wave;                          // wave-uniform
            if (f < 3 * MT) {
                const int s = f / MT, m = f - s * MT;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(s0 + s * wpart + m * 64),
                                                 (__attribute__((address_space(3))) void*)(dst + f * 1024), 16, 0, 0);
            }
        }
    };
    // halo loader slots (pixel, 8-channel unit)
    int spix[NLOAD], slds[NLOAD];
    unsigned stmask = 0;
#pragma unroll
    for (int i = 0; i < NLOAD; ++i) {
        const int u = min(tid + i * 256, NU - 1);
        const int pix = u >> 1, sub = u & 1;
        const int ly = pix / LW, lx = pix - ly * LW;
        const int gy = y0 - PT + ly, gx = x0 - PL + lx;
        spix[i] = min(max(gy, 0), H - 1) * W + min(max(gx, 0), W - 1);
        slds[i] = pix * 32 + sub * 16;
        stmask |= ((gy >= 0 && gy < H && gx >= 0 && gx < W && tid + i * 256 < NU) ? 1u : 0u) << i;
    }
    const int sub0 = (tid & 1) * 8;
    f32x4 st0[NLOAD], st1[NLOAD];
    auto request = [&](int g) {                               // stage g's halo tile -> registers
        const int c = g * 16 + sub0;
        const bool from0 = c < a.c0;
        const float* __restrict__ src = from0 ? P.in0 + c : P.in1 + (c - a.c0);
        const int cs = from0 ? a.c0 : a.c1;
#pragma unroll
        for (int i = 0; i < NLOAD; ++i) {
            const float* p = src + (size_t)spix[i] * cs;
            st0[i] = *reinterpret_cast<const f32x4*>(p);
            st1[i] = *reinterpret_cast<const f32x4*>(p + 4);
        }
    };
    auto deposit = [&]() {                                    // registers -> the three part planes
#pragma unroll
        for (int i = 0; i < NLOAD; ++i) {
            if (i * 256 + 255 < NU || tid + i * 256 < NU) {
                f32x4 v0 = st0[i], v1 = st1[i];
                if (!((stmask >> i) & 1u)) v0 = v1 = f32x4{0.f, 0.f, 0.f, 0.f};
                if (a.relu_in) { v0 = relu4(v0); v1 = relu4(v1); }
                u32x4 h, mm, l;
                split3_x8(v0, v1, h, mm, l);
                *reinterpret_cast<u32x4*>(lds + slds[i]) = h;
                *reinterpret_cast<u32x4*>(lds + PART + slds[i]) = mm;
                *reinterpret_cast<u32x4*>(lds + 2 * PART + slds[i]) = l;
            }
        }
    };
    request(0);
    copy_a(0);
    if constexpr (!ALDS) fetch(0, afr);
    deposit();
    if constexpr (ALDS) __builtin_amdgcn_s_waitcnt(0x0f70);  // vmcnt(0): this wave's share of chunk 0's fragments has landed
    __syncthreads();

    int q = 0;
    for (int g = 0; g < nstages; ++g) {
        if (g + 1 < nstages) request(g + 1);                  // in flight during this stage's MFMAs
#pragma unroll
        for (int t = 0; t < CPS; ++t) {
            if constexpr (ALDS) { if (q + 1 < nchunks) copy_a(q + 1); }
            else { fetch(min(q + 1, nchunks - 1), afn); __builtin_amdgcn_sched_barrier(0); }
            const int ta = 2 * t, tb = 2 * t + 1 < TAPS ? 2 * t + 1 : TAPS - 1;           // padded slot: zero weights, finite data
            const int oa = ((ta / KW) * LW + ta % KW) * 32, ob = ((tb / KW) * LW + tb % KW) * 32;
            const int toff = (kk >> 1) ? ob : oa;
            const unsigned char* const ab = albs + (q & 1) * ABUF + lane * 16;
            u32x4 af[3][MT];
#pragma unroll
            for (int s = 0; s < 3; ++s)
#pragma unroll
                for (int m = 0; m < MT; ++m) {
                    if constexpr (ALDS) af[s][m] = *reinterpret_cast<const u32x4*>(ab + (s * MT + m) * 1024);
                    else af[s][m] = afr[s][m];
                }
            u32x4 bh[NT], bm[NT], bl[NT];
#pragma unroll
            for (int n = 0; n < NT; ++n) {
                const unsigned char* p = lds + nbase[n] + toff;
                bh[n] = *reinterpret_cast<const u32x4*>(p);
                bm[n] = *reinterpret_cast<const u32x4*>(p + PART);
                bl[n] = *reinterpret_cast<const u32x4*>(p + 2 * PART);
            }
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
                for (int n = 0; n < NT; ++n) {
                    f32x4 c = acc[m][n];
                    c = mfma_bf16_k32(af[2][m], bh[n], c);            // smallest terms first
                    c = mfma_bf16_k32(af[0][m], bl[n], c);
                    c = mfma_bf16_k32(af[1][m], bm[n], c);
                    c = mfma_bf16_k32(af[1][m], bh[n], c);
                    c = mfma_bf16_k32(af[0][m], bm[n], c);
                    c = mfma_bf16_k32(af[0][m], bh[n], c);
                    acc[m][n] = c;
                }
            ++q;
            if constexpr (ALDS) {
                if (t + 1 < CPS) {
                    __builtin_amdgcn_s_waitcnt(0x0f70);      // the next chunk's fragments (and, once, the next halo tile) have landed
                    __syncthreads();                          // ... for every wave; this chunk's buffer may be overwritten
                }
            } else {
#pragma unroll
                for (int s = 0; s < 3; ++s)
#pragma unroll
                    for (int m = 0; m < MT; ++m) afr[s][m] = afn[s][m];
            }
        }
        if (g + 1 < nstages) {
            __syncthreads();                                  // the stage's B fragments are read
            deposit();
            if constexpr (ALDS) __builtin_amdgcn_s_waitcnt(0x0f70);
            __syncthreads();
        }
    }

    // ---- epilogue (as convs_kernel) ----
    const int Wp = (W + 1) >> 1;
#pragma unroll
    for (int m = 0; m < MT; ++m) {
        const int c = (mt0 + m) * 16 + kk * 4;
        const bool cok = c < cout;
#pragma unroll
        for (int n = 0; n < NT; ++n) {
            const int id = wave * NT + n;
            const int y = y0 + (id >> 1), x = x0 + (id & 1) * 16 + j;
            const bool ok = cok && y < H && x < W;
            const size_t p = ((size_t)min(y, H - 1) * W + min(x, W - 1)) * cout + (cok ? c : 0);
            f32x4 v = acc[m][n];
            if (a.relu_out) v = relu4(v);
            else if (a.act) v = act4(v, a.act);
            acc[m][n] = v;
            if (ok && !a.skip_full) *reinterpret_cast<f32x4*>(P.out + p) = v;
        }
        if (P.pool) {
#pragma unroll
            for (int n = 0; n < NT; ++n) {
                if (n & 2) continue;
                const int id = wave * NT + n;
                const int y = y0 + (id >> 1), x = x0 + (id & 1) * 16 + j;
                pool2_store(acc[m][n], acc[m][n + 2], y + 1 < H, x + 1 < W, (j & 1) == 0 && cok && y < H && x < W,
                            P.pool + ((size_t)(y >> 1) * Wp + (x >> 1)) * cout + (cok ? c : 0));
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// res8s_kernel<UP>: a WHOLE level-0 residual block (8 channels) with split products, fp32 tensors in HBM.
//   UP = false (unet_down_0, ARU_v1.py:208-245): t = conv3x3(image, 1 -> 8) [vector ALU, fp32]; r = relu(t); r = relu(convR_0 r);
//                r = relu(convR_1 r); d0 = relu(convR_2 r + t); also maxpool2(d0).
//   UP = true  (unet_up_0, ARU_v1.py:251-292): t = conv3x3([skip, deconv], 16 -> 8), then the same tail.
// Block = R8S_TH x 26 output pixels, 4 waves.  Every 8 -> 8 product uses the PIXEL-PAIR mapping of res8b_kernel (bf16_kernels.h):
// M = 2 adjacent pixels x 8 output channels, N = 16 pairs = ONE ROW of 32 pixels of a stage region, K = a filter row = 4 window
// pixels x 8 channels -- here with the three bf16 parts of both factors, 6 MFMAs per filter row.  The block is 26 pixels wide so
// that the widest region (t, halo 3) is exactly one such row of 32; the narrower regions (30, 28, 26 valid columns) waste the rest
// of their row (a column of the product depends on its own window only: whatever the unused columns hold stays there).
// A wave owns consecutive rows of a stage and walks the INPUT rows: the B fragments of an input row (three ds_read_b128) serve the
// three output rows it belongs to (filter rows 2, 1, 0), so a stage reads (rows + 2) / rows fragments per output row instead of 3 --
// with 18 MFMAs per fragment row the LDS would otherwise be the bound (9 KB per 288 MFMA cycles and SIMD = 128 B/clk per CU).
// conv1 of the up block is two such convolutions (skip half, deconv half of the filter) into the same accumulators; the halves share
// one LDS tile, staged one after the other.  Regions in LDS: three part planes of 16 bytes per pixel (8 bf16 channels), row pitch 34
// pixels; t (pre-ReLU, centre) in fp32 for the residual add.  Positions outside the image hold zeros (SAME padding of every conv).
// ------------------------------------------------------------------------------------------------
constexpr int R8S_TH = 8, R8S_TW = 26, R8S_P = 34, R8S_RP = 32;   // output tile; row pitch (pixels) of the conv1 input tile / of the stage regions.
// (A stage region is 32 pixels wide at most and pair 15's window reaches two pixels further: with pitch 32 those are the first pixels of the
//  next row -- finite values that only the unused columns 30, 31 of the next stage ever see.)
struct Res8SArgs {
    Res8Prob p[MAXP];      // res8_kernels.h: img / in1 / stats / out / pool, fp32
    int nprob;
    const float* w1;       // DOWN: conv1 [9][8] fp32
    const float* b1;       // conv1 bias [8]
    const u32x4* w1s;      // UP: conv1 pair fragments [half: skip, deconv][ky 3][part 3][64 lanes] x 16 bytes
    const u32x4* wrs;      // tail: [conv 3][ky 3][part 3][64 lanes] x 16 bytes
    const float* br;       // tail biases [3][8]
    XcdMap xm;
    int dbg;               // timing experiments (ASEP_R8S_DBG; results are wrong): 1 no MFMAs, 2 no split while staging, 4 no part stores, 8 no staging at all
};

template <bool UP>
struct Res8SLayout {
    static constexpr int TH = R8S_TH, P = R8S_P, RP = R8S_RP;
    static constexpr int INP = (TH + 8) * P * 16;              // one part plane of the conv1 input tile (UP)
    static constexpr int R0P = (TH + 6) * RP * 16, R1P = (TH + 4) * RP * 16;
    static constexpr int TCB = TH * R8S_TW * 32 + 256;         // t of the centre: fp32, 32 bytes per pixel, pitch 26 pixels (+ the reach of the unused columns)
    static constexpr int IMG = (TH + 8) * P * 4;               // DOWN: the image tile, fp32
    // UP: [in x 3 | r0 x 3 | tc], r1 takes in's place; DOWN: [img | r1 x 3 | r0 x 3 | tc]
    static constexpr int IN_OFF = 0, R1_OFF = UP ? 0 : IMG, R0_OFF = UP ? 3 * INP : IMG + 3 * R1P, TC_OFF = R0_OFF + 3 * R0P;
    static constexpr int BYTES = TC_OFF + TCB;
};

template <bool UP>
__global__ __launch_bounds__(256, 3) void res8s_kernel(const Res8SArgs a) {
    typedef Res8SLayout<UP> L;
    constexpr int TH = R8S_TH, TW = R8S_TW, P = R8S_P, RP = R8S_RP;
    constexpr int H0 = TH + 6, H1 = TH + 4, H2 = TH + 2, IH = TH + 8;
    __shared__ __attribute__((aligned(16))) unsigned char lds[L::BYTES];
    unsigned char* const in = lds + L::IN_OFF;
    unsigned char* const r0 = lds + L::R0_OFF;
    unsigned char* const r1 = lds + L::R1_OFF;
    float* const tc = reinterpret_cast<float*>(lds + L::TC_OFF);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int j = lane & 15, kk = lane >> 4, e = kk >> 1, ch = (kk & 1) * 4;     // D layout: pixel 2 j + e, channels ch .. ch + 3
    const int c = 2 * j + e;
    const int bid = sched_tile(a.xm);
    if (bid < 0) return;
    int pi = 0;
    pi = prob_of_tile(a, bid);
    const Res8Prob& Pr = a.p[pi];
    const int tile = bid - Pr.tile_begin;
    const int ty = tile / Pr.tiles_x, tx = tile - ty * Pr.tiles_x;
    const int x0 = tx * TW, y0 = ty * TH;
    const int H = Pr.H, W = Pr.W;

    // A fragments of one 8 -> 8 filter: [ky][part], 36 registers; `fetch` of the next filter is issued a stage ahead
    auto fetch = [&](const u32x4* __restrict__ w, u32x4 (&A)[3][3]) {
        if (a.dbg & 16) return;
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
            for (int s = 0; s < 3; ++s) A[ky][s] = w[(ky * 3 + s) * 64 + lane];
    };
    // NR output rows row0 .. of a stage from the source region `src` (part planes `sp` bytes apart, SR rows): acc += conv(src)
    auto conv_rows = [&](auto nr_c, auto pitch_c, const unsigned char* src, int sp, int SR, int row0, const u32x4 (&A)[3][3], auto& acc) {
        constexpr int NR = decltype(nr_c)::value, PP = decltype(pitch_c)::value;
        if (a.dbg & 1) return;
        const unsigned char* b = src + (row0 * PP + 2 * j + kk) * 16;
#pragma unroll
        for (int i = 0; i < NR + 2; ++i) {
            if (row0 + i < SR) {                               // (wave-uniform: a wave's last rows may not exist)
                const u32x4 bh = *reinterpret_cast<const u32x4*>(b + i * PP * 16);
                const u32x4 bm = *reinterpret_cast<const u32x4*>(b + i * PP * 16 + sp);
                const u32x4 bl = *reinterpret_cast<const u32x4*>(b + i * PP * 16 + 2 * sp);
#pragma unroll
                for (int ky = 0; ky < 3; ++ky) {
                    const int o = i - ky;
                    if (o >= 0 && o < NR) {
                        f32x4 v = acc[o];
                        v = mfma_bf16_k32(A[ky][2], bh, v);
                        v = mfma_bf16_k32(A[ky][0], bl, v);
                        v = mfma_bf16_k32(A[ky][1], bm, v);
                        v = mfma_bf16_k32(A[ky][1], bh, v);
                        v = mfma_bf16_k32(A[ky][0], bm, v);
                        v = mfma_bf16_k32(A[ky][0], bh, v);
                        acc[o] = v;
                    }
                }
            }
        }
    };
    // relu(v) of 4 channels of pixel (row, c) of a region -> its three part planes
    auto put_parts = [&](unsigned char* dst, int dp, int row, f32x4 v) {
        if (a.dbg & 4) return;
        unsigned h0, m0, l0, h1, m1, l1;
        split3_pair(v.x, v.y, h0, m0, l0);
        split3_pair(v.z, v.w, h1, m1, l1);
        unsigned char* q = dst + (row * RP + c) * 16 + ch * 2;
        *reinterpret_cast<u32x2*>(q) = u32x2{h0, h1};
        *reinterpret_cast<u32x2*>(q + dp) = u32x2{m0, m1};
        *reinterpret_cast<u32x2*>(q + 2 * dp) = u32x2{l0, l1};
    };
    // the whole input window inside the image (~95 % of the tiles of a page): no position of any stage needs the zero test
    const bool interior = y0 - 4 >= 0 && y0 + TH + 4 <= H && x0 - 4 >= 0 && x0 - 4 + P <= W;
    auto masked = [&](int halo, int row, f32x4 v) {
        if (!interior) {
            const int gy = y0 - halo + row, gx = x0 - halo + c;
            if (!(gy >= 0 && gy < H && gx >= 0 && gx < W)) v = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        return v;
    };

    u32x4 A[3][3], An[3][3];
    // ---- conv1 -> relu(t) over region 0 (H0 x 32, halo 3), raw t of the centre (tc) ----
    if constexpr (UP) {
        constexpr int NPX = IH * P, NLOAD = (NPX + 255) / 256;
        f32x4 sk[NLOAD][2];
        unsigned mask = 0;
        unsigned soff[NLOAD];                                   // element offset of the slot's pixel (clamped into the image; < 2^32: checked by the launcher)
#pragma unroll
        for (int i = 0; i < NLOAD; ++i) {
            const int u = min(tid + i * 256, NPX - 1);
            const int ly = u / P, lx = u - ly * P;
            const int gy = y0 - 4 + ly, gx = x0 - 4 + lx;
            soff[i] = (unsigned)((min(max(gy, 0), H - 1) * W + min(max(gx, 0), W - 1)) * 8);
            sk[i][0] = *reinterpret_cast<const f32x4*>(Pr.img + soff[i]);
            sk[i][1] = *reinterpret_cast<const f32x4*>(Pr.img + soff[i] + 4);
            mask |= ((gy >= 0 && gy < H && gx >= 0 && gx < W) ? 1u : 0u) << i;
        }
        fetch(a.w1s, A);
        fetch(a.w1s + 9 * 64, An);
        auto stage_in = [&](f32x4 (&v)[NLOAD][2]) {
            if (a.dbg & 8) return;
#pragma unroll
            for (int i = 0; i < NLOAD; ++i) {
                const int u = tid + i * 256;
                if (u < NPX) {
                    const bool ok = (mask >> i) & 1u;
                    u32x4 h, mm, l;
                    if (a.dbg & 2) { h = mm = l = __builtin_bit_cast(u32x4, v[i][0]); }
                    else split3_x8(ok ? v[i][0] : f32x4{0.f, 0.f, 0.f, 0.f}, ok ? v[i][1] : f32x4{0.f, 0.f, 0.f, 0.f}, h, mm, l);
                    *reinterpret_cast<u32x4*>(in + u * 16) = h;
                    *reinterpret_cast<u32x4*>(in + L::INP + u * 16) = mm;
                    *reinterpret_cast<u32x4*>(in + 2 * L::INP + u * 16) = l;
                }
            }
        };
        stage_in(sk);
        // the deconv half of the window is requested now and staged after the first half's MFMAs
#pragma unroll
        for (int i = 0; i < NLOAD; ++i) {
            sk[i][0] = *reinterpret_cast<const f32x4*>(Pr.in1 + soff[i]);
            sk[i][1] = *reinterpret_cast<const f32x4*>(Pr.in1 + soff[i] + 4);
        }
        __syncthreads();
        constexpr int NR = (H0 + 3) / 4;
        const int row0 = wave * NR;
        const f32x4 b4 = *reinterpret_cast<const f32x4*>(a.b1 + ch);
        f32x4 acc[NR];
#pragma unroll
        for (int o = 0; o < NR; ++o) acc[o] = b4;
        conv_rows(std::integral_constant<int, NR>{}, std::integral_constant<int, P>{}, in, L::INP, IH, row0, A, acc);
        __syncthreads();
        stage_in(sk);
        fetch(a.wrs, A);                                        // convR_0's fragments fly during the second half
        __syncthreads();
        conv_rows(std::integral_constant<int, NR>{}, std::integral_constant<int, P>{}, in, L::INP, IH, row0, An, acc);
#pragma unroll
        for (int o = 0; o < NR; ++o) {
            const int row = row0 + o;
            if (row < H0) {
                const f32x4 v = masked(3, row, acc[o]);
                put_parts(r0, L::R0P, row, relu4(v));
                if (row >= 3 && row < 3 + TH && c >= 3 && c < 3 + TW) *reinterpret_cast<f32x4*>(tc + ((row - 3) * TW + c - 3) * 8 + ch) = v;
            }
        }
    } else {
        constexpr int NPX = IH * P, NLOAD = (NPX + 255) / 256;
        float st[NLOAD];
        float mean = 0.f, inv = 1.f;
        if (Pr.stats) { mean = Pr.stats[0]; inv = Pr.stats[1]; }
#pragma unroll
        for (int i = 0; i < NLOAD; ++i) {
            const int u = min(tid + i * 256, NPX - 1);
            const int ly = u / P, lx = u - ly * P;
            st[i] = Pr.img[(size_t)min(max(y0 - 4 + ly, 0), H - 1) * W + min(max(x0 - 4 + lx, 0), W - 1)];
        }
        fetch(a.wrs, A);
        float* const img = reinterpret_cast<float*>(in);
#pragma unroll
        for (int i = 0; i < NLOAD; ++i) {
            const int u = tid + i * 256;
            if (u < NPX) {
                const int ly = u / P, lx = u - ly * P;
                const int gy = y0 - 4 + ly, gx = x0 - 4 + lx;
                img[u] = (gy >= 0 && gy < H && gx >= 0 && gx < W) ? (st[i] - mean) * inv : 0.f;
            }
        }
        __syncthreads();
        typedef const float __attribute__((address_space(4)))* cptr;
        cptr w1 = (cptr)a.w1;
        cptr b1 = (cptr)a.b1;
        // a thread takes 4 channels of a pixel (the D layout's unit): H0 x 32 pixels x 2 halves
        for (int u = tid; u < H0 * 32 * 2; u += 256) {
            const int hf = u & 1, px = u >> 1, ly = px >> 5, lx = px & 31;
            f32x4 v = f32x4{b1[hf * 4], b1[hf * 4 + 1], b1[hf * 4 + 2], b1[hf * 4 + 3]};
#pragma unroll
            for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) {
                    const float x = img[(ly + ky) * P + lx + kx];
                    const int t = (ky * 3 + kx) * 8 + hf * 4;
                    v.x = fmaf(x, w1[t], v.x); v.y = fmaf(x, w1[t + 1], v.y); v.z = fmaf(x, w1[t + 2], v.z); v.w = fmaf(x, w1[t + 3], v.w);
                }
            const int gy = y0 - 3 + ly, gx = x0 - 3 + lx;
            if (!(gy >= 0 && gy < H && gx >= 0 && gx < W)) v = f32x4{0.f, 0.f, 0.f, 0.f};
            const f32x4 r = relu4(v);
            unsigned h0, m0, l0, h1, m1, l1;
            split3_pair(r.x, r.y, h0, m0, l0);
            split3_pair(r.z, r.w, h1, m1, l1);
            unsigned char* q = r0 + (ly * RP + lx) * 16 + hf * 8;
            *reinterpret_cast<u32x2*>(q) = u32x2{h0, h1};
            *reinterpret_cast<u32x2*>(q + L::R0P) = u32x2{m0, m1};
            *reinterpret_cast<u32x2*>(q + 2 * L::R0P) = u32x2{l0, l1};
            if (ly >= 3 && ly < 3 + TH && lx >= 3 && lx < 3 + TW) *reinterpret_cast<f32x4*>(tc + ((ly - 3) * TW + lx - 3) * 8 + hf * 4) = v;
        }
    }
    fetch(a.wrs + 9 * 64, An);
    __syncthreads();

    // ---- stage 1: r0 (H0 rows) -> r1 (H1 rows, halo 2) ----
    {
        constexpr int NR = (H1 + 3) / 4;
        const int row0 = wave * NR;
        const f32x4 b4 = *reinterpret_cast<const f32x4*>(a.br + ch);
        f32x4 acc[NR];
#pragma unroll
        for (int o = 0; o < NR; ++o) acc[o] = b4;
        conv_rows(std::integral_constant<int, NR>{}, std::integral_constant<int, RP>{}, r0, L::R0P, H0, row0, A, acc);
#pragma unroll
        for (int o = 0; o < NR; ++o)
            if (row0 + o < H1) put_parts(r1, L::R1P, row0 + o, masked(2, row0 + o, relu4(acc[o])));
    }
    fetch(a.wrs + 18 * 64, A);
    __syncthreads();
    // ---- stage 2: r1 -> r0 (H2 rows, halo 1; the t region is dead) ----
    {
        constexpr int NR = (H2 + 3) / 4;
        const int row0 = wave * NR;
        const f32x4 b4 = *reinterpret_cast<const f32x4*>(a.br + 8 + ch);
        f32x4 acc[NR];
#pragma unroll
        for (int o = 0; o < NR; ++o) acc[o] = b4;
        conv_rows(std::integral_constant<int, NR>{}, std::integral_constant<int, RP>{}, r1, L::R1P, H1, row0, An, acc);
#pragma unroll
        for (int o = 0; o < NR; ++o)
            if (row0 + o < H2) put_parts(r0, L::R0P, row0 + o, masked(1, row0 + o, relu4(acc[o])));
    }
    __syncthreads();
    // ---- stage 3: r0 -> out = relu(convR_2 + t), rows in pairs per wave (2x2 pool in registers) ----
    {
        constexpr int NR = TH / 4;
        static_assert(NR == 2, "a wave owns one row pair");
        const int row0 = wave * NR;
        const f32x4 b4 = *reinterpret_cast<const f32x4*>(a.br + 16 + ch);
        f32x4 acc[NR];
#pragma unroll
        for (int o = 0; o < NR; ++o) acc[o] = b4 + *reinterpret_cast<const f32x4*>(tc + ((row0 + o) * TW + c) * 8 + ch);
        conv_rows(std::integral_constant<int, NR>{}, std::integral_constant<int, RP>{}, r0, L::R0P, H2, row0, A, acc);
        const int x = x0 + c;
        const bool xin = c < TW && x < W;
#pragma unroll
        for (int o = 0; o < NR; ++o) {
            acc[o] = relu4(acc[o]);
            const int y = y0 + row0 + o;
            if (xin && y < H && !(a.dbg & 32)) *reinterpret_cast<f32x4*>(Pr.out + ((size_t)y * W + x) * 8 + ch) = acc[o];
        }
        if (Pr.pool) {
            // window = rows y, y + 1 (registers) x pixels 2 j, 2 j + 1 (this lane and lane ^ 32); members outside the image are excluded
            const int y = y0 + row0, Wp = (W + 1) >> 1;
            f32x4 mm = (y + 1 < H) ? max4(acc[0], acc[1]) : acc[0];
            f32x4 lo, hi;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float mine = (x < W) ? mm[q] : -INFINITY;
                lo[q] = from_lower_half(mine);
                hi[q] = from_upper_half(mine);
            }
            mm = max4(lo, hi);
            if (e == 0 && xin && y < H) *reinterpret_cast<f32x4*>(Pr.pool + ((size_t)(y >> 1) * Wp + (x >> 1)) * 8 + ch) = mm;
        }
    }
}


}  // namespace asep
