// fp32 convolutions whose PRODUCTS run on the bf16 matrix pipeline (asep_aru_cfg.compute_dtype = 2, "f32 split").
//
// gfx950 multiplies fp32 at 157 TFLOP/s (v_mfma_f32_16x16x4_f32, v_pk_fma_f32) and bf16 at 2.5 PFLOP/s with fp32 accumulation.  An fp32
// number is EXACTLY the sum of three bfloat16 numbers (8 + 8 + 8 significand bits, round-to-nearest at every cut, same exponent range as
// fp32: no scaling, no overflow case):      x = xh + xm + xl,   w = wh + wm + wl.
// A product x w is then the sum of nine bf16 products; the six largest
//        xh wh  +  (xh wm + xm wh)  +  (xm wm + xh wl + xl wh)
// leave out terms of at most 2^-23 |x w| (xm wl + xl wm <= 2 * 2^-24, xl wl <= 2^-32; measured 4.8e-8 against the 5.9e-8 of ONE fp32 multiply) -- and each of them is
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
            const float* __restrict__ src = concat_src(P.in0, P.in1, c, a.c0);
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
                    split3_x8(v0, v1, h, mm, l);
                    *reinterpret_cast<u32x4*>(lds + slds[i]) = h;
                    *reinterpret_cast<u32x4*>(lds + PART + slds[i]) = mm;
                    *reinterpret_cast<u32x4*>(lds + 2 * PART + slds[i]) = l;
                }
            }
            __syncthreads();
        }
        auto chunk = [&](int toff) {
            fetch(min(q + 1, nchunks - 1), afn);
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
        const float* __restrict__ src = concat_src(P.in0, P.in1, c, a.c0);
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

// (Level 0 on split products -- res8s_kernel, one-shot and persistent -- was built in round 4, is correct and is SLOWER than the
//  vector-ALU blocks of res8v_kernels.h (DESIGN_LESSONS.md 32); it left the tree in round 5, git history keeps it: 545c6ae .. 7d11314.)


// ------------------------------------------------------------------------------------------------
// deconvs_kernel: the 3x3 stride-2 SAME transposed convolutions of the levels with >= 32 input channels (unet_up_1 .. : 32 -> 16, 64 -> 32, 128 -> 64)
// on SPLIT PRODUCTS (round 6; layers.py:342-367 deconv2d + bias + activation, the semantics of deconv_mfma_kernel).  These three layers ran on the fp32 MFMA
// (63-90 TFLOP/s: 643 + 477 + 456 us per 4-page launch of a 7.1 ms page) while their tensors move in 340 + 170 + 85 us at 5 TB/s.
// Structure of deconvb_kernel MODE 2: a block = DS_TH x 16 INPUT positions (their 2 DS_TH x 32 output pixels), wave w owns input rows w, w + 4, ...;
// an n-tile = 16 positions of one input row; output class c = 2 py + px of an input position (Y, X) is output pixel (2 Y - pbh + py, 2 X - pbw + px)
//     = sum over (dy, dx) in {0, 1}^2 of  in(Y - dy, X - dx) . W[ky][kx],   ky = py ? 1 : (dy ? 2 : 0) (py = 1 has no dy = 1 term), kx alike;
// per stage of 32 input channels the (DS_TH + 1) x 17 halo tile is read as fp32, cut into its three bfloat16 parts (split3_x8) and stored as three sets of
// two 16-channel planes; a (dy, dx) step reads ONE B fragment triple and feeds up to four classes x MT m-tiles x six MFMAs (h h, h m, m h, m m, h l, l h).
// a.wpk = pack_deconv_split's [stage][tap][part h, m, l][m-tile][lane] x 16 bytes; a.groups = stages of 32 channels.  Results: fp32, bias, ReLU or the
// variants' activation, 16-byte stores of a lane's four channels.
// ------------------------------------------------------------------------------------------------
constexpr int DS_TW = 16;
template <int MT, int DS_TH>
__global__ __launch_bounds__(256, 2) void deconvs_kernel(const ConvArgs a) {
    constexpr int LH = DS_TH + 1, LW = DS_TW + 1, RW = DS_TH / 4;
    constexpr int PLANE = LH * LW * 32, PART = 2 * PLANE;
    constexpr int NU = LH * LW * 4, NLOAD = (NU + 255) / 256;                       // 8-channel units of a stage's halo tile
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
    const int X0 = tx * DS_TW, Y0 = ty * DS_TH, mt0 = blockIdx.y * MT;
    const int Hi = P.H, Wi = P.W, cin = a.c0;

    f32x4 acc[RW][4][MT];
#pragma unroll
    for (int r = 0; r < RW; ++r)
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int m = 0; m < MT; ++m) acc[r][c][m] = f32x4{0.f, 0.f, 0.f, 0.f};
    f32x4 bias4[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) {
        const int c = (mt0 + m) * 16 + kk * 4;
        bias4[m] = *reinterpret_cast<const f32x4*>(a.bias + (c < a.cout ? c : 0));
        if (c >= a.cout) bias4[m] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    const u32x4* __restrict__ wbase = reinterpret_cast<const u32x4*>(a.wpk) + (size_t)mt0 * 64 + lane;
    const size_t wstride = (size_t)a.mtiles * 64;                                   // one (stage, tap, part) of all m-tiles

    int spix[NLOAD];
    unsigned mask = 0;
#pragma unroll
    for (int i = 0; i < NLOAD; ++i) {
        const int u = min(tid + i * 256, NU - 1);
        const int pix = u >> 2;
        const int ly = pix / LW, lx = pix - ly * LW;
        const int gy = Y0 - 1 + ly, gx = X0 - 1 + lx;
        spix[i] = min(max(gy, 0), Hi - 1) * Wi + min(max(gx, 0), Wi - 1);
        mask |= ((gy >= 0 && gy < Hi && gx >= 0 && gx < Wi) ? 1u : 0u) << i;
    }
    const int sub8 = (tid & 3) * 8;                                                 // (256 is a multiple of 4: the thread's 8-channel unit is the same for all its slots)
    for (int g = 0; g < a.groups; ++g) {
        f32x4 st[NLOAD][2];
#pragma unroll
        for (int i = 0; i < NLOAD; ++i) {
            const float* src = P.in0 + (size_t)spix[i] * cin + g * 32 + sub8;
            st[i][0] = *reinterpret_cast<const f32x4*>(src);
            st[i][1] = *reinterpret_cast<const f32x4*>(src + 4);
        }
        if (g > 0) __syncthreads();                                                 // the previous stage's readers are done
#pragma unroll
        for (int i = 0; i < NLOAD; ++i) {
            const int u = tid + i * 256;
            if (u < NU) {
                const int pix = u >> 2, sub = u & 3;
                u32x4 ph, pm, pl;
                split3_x8(st[i][0], st[i][1], ph, pm, pl);
                if (!((mask >> i) & 1u)) ph = pm = pl = u32x4{0u, 0u, 0u, 0u};
                unsigned char* d = lds + (sub >> 1) * PLANE + pix * 32 + (sub & 1) * 16;
                *reinterpret_cast<u32x4*>(d) = ph;
                *reinterpret_cast<u32x4*>(d + PART) = pm;
                *reinterpret_cast<u32x4*>(d + 2 * PART) = pl;
            }
        }
        __syncthreads();
        const u32x4* __restrict__ wg = wbase + (size_t)g * 9 * 3 * wstride;
#pragma unroll
        for (int dy = 0; dy < 2; ++dy)
#pragma unroll
            for (int dx = 0; dx < 2; ++dx)
#pragma unroll
                for (int py = 0; py < 2; ++py) {
                    if (dy == 1 && py == 1) continue;
#pragma unroll
                    for (int px = 0; px < 2; ++px) {
                        if (dx == 1 && px == 1) continue;
                        const int ky = py ? 1 : (dy ? 2 : 0), kx = px ? 1 : (dx ? 2 : 0), tap = ky * 3 + kx;
                        // the tap's A fragments (three parts x MT m-tiles) serve both rows of the wave
                        u32x4 ah[MT], am[MT], al[MT];
#pragma unroll
                        for (int m = 0; m < MT; ++m) {
                            ah[m] = wg[((size_t)tap * 3 + 0) * wstride + (size_t)m * 64];
                            am[m] = wg[((size_t)tap * 3 + 1) * wstride + (size_t)m * 64];
                            al[m] = wg[((size_t)tap * 3 + 2) * wstride + (size_t)m * 64];
                        }
#pragma unroll
                        for (int r = 0; r < RW; ++r) {
                            const int Yl = wave + 4 * r;
                            const unsigned char* bp = lds + (kk >> 1) * PLANE + ((Yl + 1 - dy) * LW + (j + 1 - dx)) * 32 + (kk & 1) * 16;
                            const u32x4 bh = *reinterpret_cast<const u32x4*>(bp), bm = *reinterpret_cast<const u32x4*>(bp + PART),
                                        bl = *reinterpret_cast<const u32x4*>(bp + 2 * PART);
#pragma unroll
                            for (int m = 0; m < MT; ++m) {
                                f32x4 c = acc[r][2 * py + px][m];
                                // (small terms first)
                                c = mfma_bf16_k32(al[m], bh, c);
                                c = mfma_bf16_k32(ah[m], bl, c);
                                c = mfma_bf16_k32(am[m], bm, c);
                                c = mfma_bf16_k32(am[m], bh, c);
                                c = mfma_bf16_k32(ah[m], bm, c);
                                c = mfma_bf16_k32(ah[m], bh, c);
                                acc[r][2 * py + px][m] = c;
                            }
                        }
                    }
                }
    }
    // ---- epilogue: lane = input position (Yl, j), 4 consecutive output channels 16 (mt0 + m) + 4 kk of each of its four output pixels ----
#pragma unroll
    for (int m = 0; m < MT; ++m) {
        const int c = (mt0 + m) * 16 + kk * 4;
        if (c >= a.cout) continue;
#pragma unroll
        for (int r = 0; r < RW; ++r) {
            const int Yl = wave + 4 * r;
#pragma unroll
            for (int cls = 0; cls < 4; ++cls) {
                const int y = 2 * (Y0 + Yl) - P.pbh + (cls >> 1), x = 2 * (X0 + j) - P.pbw + (cls & 1);
                f32x4 v = acc[r][cls][m] + bias4[m];
                if (a.relu_out) v = relu4(v);
                else if (a.act) v = act4(v, a.act);
                if (Y0 + Yl < Hi && X0 + j < Wi && y >= 0 && y < P.Ho && x >= 0 && x < P.Wo)
                    *reinterpret_cast<f32x4*>(P.out + ((size_t)y * P.Wo + x) * a.cout + c) = v;
            }
        }
    }
}

}  // namespace asep
