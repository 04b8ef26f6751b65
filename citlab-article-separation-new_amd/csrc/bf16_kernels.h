// ARU-Net device kernels of the NATIVE bf16 data path (asep_aru_cfg.compute_dtype = 1; BASELINE configs[4] "bf16 convs").
//
// Activations live in HBM and in LDS as bf16 NHWC (half the bytes of the fp32 path: this variant is HBM-bound), every
// product runs on v_mfma_f32_16x16x32_bf16 (K = 32 per instruction, fp32 accumulate), bias / residual / ReLU / 2x2 max pool
// are applied on the fp32 accumulators before ONE rounding to bf16 (round-to-nearest-even, v_cvt_pk_bf16_f32).
//
// Reference semantics (file:line in /root/reference), the same as the fp32 kernels of aru_kernels.h:
//   layers.py:191-247   conv2d (SAME, stride 1) + bias + activation          -> convb_kernel
//   ARU_v1.py:212-227   residual block tail (3 x conv3x3, +t, ReLU)           -> resb_tail_kernel (one kernel per block)
//   layers.py:342-367   deconv2d (conv2d_transpose 3x3, stride 2, SAME)       -> deconvb_kernel
//   layers.py:716-720   upsample_simple: channel sum                          -> chansumb_kernel
//
// Operand layout of v_mfma_f32_16x16x32_bf16 (cdna_hip_programming.md, "Fragment layout"): lane l holds A[row l&15][k = 8(l>>4)+j]
// and B[k = 8(l>>4)+j][col l&15], j = 0..7 (16 bytes); D: col = l&15, rows 4(l>>4)+r.  M = output channels, N = 16 pixels.
// A lane's B fragment is ONE 16-byte LDS read: 8 consecutive channels of one pixel of one tap.  LDS tiles are "planes" of 16
// channels (32 bytes per pixel): the ds_read_b128 lane groups of gfx950 ({0-3,12-15,20-27}, ...) then hit 64 distinct banks
// (MI355X_MICROARCH.md, LDS table) -- no swizzle needed, unlike the 64-byte fp32 records of the fp32 path.
#pragma once
#include "aru_kernels.h"

namespace asep {

typedef unsigned short bf16_t;
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef short s16x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ unsigned pack_bf16x2(float a, float b) {
    return __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2_t{a, b}, bf16x2_t));
}
__device__ __forceinline__ u32x2 pack_bf16x4(f32x4 v) { return u32x2{pack_bf16x2(v.x, v.y), pack_bf16x2(v.z, v.w)}; }
__device__ __forceinline__ f32x4 unpack_bf16x4(u32x2 p) {
    return f32x4{__uint_as_float(p.x << 16), __uint_as_float(p.x & 0xffff0000u), __uint_as_float(p.y << 16), __uint_as_float(p.y & 0xffff0000u)};
}
// ReLU on two packed bf16: a negative bf16 is a negative int16 (v_pk_max_i16 with 0)
__device__ __forceinline__ unsigned relu_bf16x2(unsigned x) {
    return __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_bit_cast(s16x2, x), s16x2{0, 0}));
}
__device__ __forceinline__ u32x4 relu_bf16x8(u32x4 v) { return u32x4{relu_bf16x2(v.x), relu_bf16x2(v.y), relu_bf16x2(v.z), relu_bf16x2(v.w)}; }
__device__ __forceinline__ f32x4 mfma_bf16_k32(u32x4 a, u32x4 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

// ------------------------------------------------------------------------------------------------
// convb_kernel: stride-1 SAME convolution (3x3 or 4x4), bf16 in / out, optional channel concat [in0, in1], residual,
// ReLU on the input and / or output, 2x2 max pool of the output.
//   MODE 0: Cin == 8,  3x3: one 16-byte plane; K chunk = filter row ky: 4 x-consecutive pixels x 8 ch (4th = zero weights)
//   MODE 1: Cin == 16: one 32-byte plane;       K chunk = two consecutive taps x 16 ch
//   MODE 2: Cin % 32 == 0: stages of 32 channels = two 32-byte planes; K chunk = (tap, 32-channel group)
// Block = 256 threads = TH x 32 output pixels x 16 MT WM output channels.
// ------------------------------------------------------------------------------------------------
struct ConvBProb {
    const bf16_t* in0;
    const bf16_t* in1;     // channel concat behind in0, or nullptr
    const bf16_t* res;     // residual [H,W,cout], added before the output ReLU, or nullptr
    bf16_t* out;           // [H,W,cout] (may be nullptr with skip_full)
    void* pool;            // maxpool2(out): bf16 (or fp32 with pool_f32) [ceil(H/2), ceil(W/2), cout], or nullptr
    int H, W;
    int tiles_x, tile_begin;
};
struct ConvBArgs {
    ConvBProb p[MAXP];
    int nprob;
    const u32x4* wpk;      // [chunk][mtile][lane] x 16 bytes (8 bf16)
    const float* bias;     // [cout] fp32
    int c0, c1;            // channels of in0 / in1 (multiples of 8)
    int cout, mtiles, groups;
    int relu_in, relu_out, skip_full, pool_f32;
};

// Waves: WM along the output channels x 4 / WM along the pixels; wave (wm, wn) owns m-tiles wm MT .. and n-tiles id = wn NT + n
// (row id >> 1, column block id & 1).  The A fragments (weights) of a stage are copied to LDS ONCE per block next to the halo
// tile and read from there by all waves: fetched per wave from L2 (first cut of this kernel) they cost 4 KB per 16 MFMAs and
// wave = the whole vector-memory path of a CU, and every chunk waited for an L2 round trip.
template <int KH, int KW, int MODE, int MT, int WM, int TH, int MINB>
__global__ __launch_bounds__(256, MINB) void convb_kernel(const ConvBArgs a) {
    static_assert(MODE != 0 || (KH == 3 && KW == 3), "MODE 0 is the 3x3 conv with 8 input channels");
    static_assert(WM == 1 || WM == 2, "one or two waves along the output channels");
    constexpr int TW = 32, WN = 4 / WM, NT = TH * 2 / WN, MTB = MT * WM;
    static_assert(NT % 4 == 0, "a wave owns whole row pairs (fused 2x2 pool)");
    constexpr int LH = TH + KH - 1, LW = TW + KW - 1 + (MODE == 0 ? 1 : 0);     // MODE 0 reads a 4th (zero-weight) column
    constexpr int PT = (KH - 1) / 2, PL = (KW - 1) / 2;                         // TF SAME: pad_before = (k-1)/2
    constexpr int TAPS = KH * KW;
    constexpr int SUBS = MODE == 0 ? 1 : (MODE == 1 ? 2 : 4);                   // 16-byte units per pixel and stage
    constexpr int PXB = MODE == 0 ? 16 : 32;                                    // bytes per pixel in a plane
    constexpr int PLANE = LH * LW * PXB;
    constexpr int NPL = MODE == 2 ? 2 : 1;
    constexpr int CPS = MODE == 0 ? KH : (MODE == 1 ? (TAPS + 1) / 2 : TAPS);   // K chunks per stage
    constexpr int NU = LH * LW * SUBS, NLOAD = (NU + 255) / 256;
    constexpr int NWU = CPS * MTB * 64, NWLOAD = (NWU + 255) / 256;             // 16-byte units of a stage's A fragments
    __shared__ __attribute__((aligned(16))) unsigned char lds[NPL * PLANE + NWU * 16];
    unsigned char* const wlds = lds + NPL * PLANE;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int j = lane & 15, kk = lane >> 4;
    const int wm = WM == 2 ? (wave & 1) : 0, wn = WM == 2 ? (wave >> 1) : wave;
    int pi = 0;
    while (pi + 1 < a.nprob && (int)blockIdx.x >= a.p[pi + 1].tile_begin) ++pi;
    const ConvBProb& P = a.p[pi];
    const int tile = blockIdx.x - P.tile_begin;
    const int ty = tile / P.tiles_x, tx = tile - ty * P.tiles_x;
    const int x0 = tx * TW, y0 = ty * TH, mtb0 = blockIdx.y * MTB, mt0 = mtb0 + wm * MT;
    const int H = P.H, W = P.W;

    f32x4 acc[MT][NT];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int n = 0; n < NT; ++n) acc[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};

    // byte offset of (row, col + j) of each n-tile of this wave, plus the lane's channel half / plane
    int nbase[NT];
#pragma unroll
    for (int n = 0; n < NT; ++n) {
        const int id = wn * NT + n;
        nbase[n] = ((id >> 1) * LW + (id & 1) * 16 + j) * PXB + (MODE == 0 ? kk * 16 : (kk & 1) * 16 + (MODE == 2 ? (kk >> 1) * PLANE : 0));
    }
    const int ngroups = MODE == 2 ? a.groups : 1;
    const u32x4* __restrict__ wsrc = a.wpk + (size_t)mtb0 * 64;
    const size_t wstride = (size_t)a.mtiles * 64;
    const int mt_have = min(MTB, a.mtiles - mtb0);            // m-tiles of this block that exist (cout 8 / 16: one of MTB)

    for (int g = 0; g < ngroups; ++g) {
        // ---- stage g: halo tile of 32 (16, 8) input channels + the stage's A fragments -> LDS.  Requests first (clamped,
        //      always valid addresses); zero padding / ReLU when the registers go to LDS ----
        u32x4 st[NLOAD];
        unsigned stmask = 0;
#pragma unroll
        for (int i = 0; i < NLOAD; ++i) {
            const int u = min(tid + i * 256, NU - 1);
            const int pix = u / SUBS, sub = u - pix * SUBS;
            const int ly = pix / LW, lx = pix - ly * LW;
            const int gy = y0 - PT + ly, gx = x0 - PL + lx;
            const int c = g * 32 + sub * 8;
            const bool from0 = c < a.c0;
            const bf16_t* __restrict__ src = from0 ? P.in0 + c : P.in1 + (c - a.c0);
            const int cs = from0 ? a.c0 : a.c1;
            const int cy = min(max(gy, 0), H - 1), cx = min(max(gx, 0), W - 1);
            st[i] = *reinterpret_cast<const u32x4*>(src + ((size_t)cy * W + cx) * cs);
            stmask |= ((gy >= 0 && gy < H && gx >= 0 && gx < W) ? 1u : 0u) << i;
        }
        if (g > 0) __syncthreads();                          // the previous stage's readers are done
        // A fragments: global -> LDS without registers (global_load_lds_dwordx4: one wave-instruction copies 1 KB, lane i to
        // base + 16 i); whole waves, NWU is a multiple of 64
#pragma unroll
        for (int i = 0; i < NWLOAD; ++i) {
            const int u0 = i * 256 + wave * 64;              // wave-uniform
            if (u0 < NWU) {
                const int u = u0 + lane;
                const int t = u / (MTB * 64), r = u - t * (MTB * 64);
                const u32x4* gsrc = wsrc + (size_t)(g * CPS + t) * wstride + min(r, mt_have * 64 - 1);
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                                 (__attribute__((address_space(3))) void*)(wlds + u0 * 16), 16, 0, 0);
            }
        }
#pragma unroll
        for (int i = 0; i < NLOAD; ++i) {
            const int u = tid + i * 256;
            if (u < NU) {
                const int pix = u / SUBS, sub = u - pix * SUBS;
                u32x4 v = ((stmask >> i) & 1u) ? st[i] : u32x4{0u, 0u, 0u, 0u};
                if (a.relu_in) v = relu_bf16x8(v);
                *reinterpret_cast<u32x4*>(lds + (sub >> 1) * PLANE + pix * PXB + (sub & 1) * 16) = v;
            }
        }
        __syncthreads();
        auto chunk = [&](int t, int toff) {
            u32x4 af[MT], bfr[NT];
#pragma unroll
            for (int m = 0; m < MT; ++m) af[m] = *reinterpret_cast<const u32x4*>(wlds + ((t * MTB + wm * MT + m) * 64 + lane) * 16);
#pragma unroll
            for (int n = 0; n < NT; ++n) bfr[n] = *reinterpret_cast<const u32x4*>(lds + nbase[n] + toff);
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
                for (int n = 0; n < NT; ++n) acc[m][n] = mfma_bf16_k32(af[m], bfr[n], acc[m][n]);
        };
        if constexpr (MODE == 0) {
#pragma unroll
            for (int t = 0; t < CPS; ++t) chunk(t, t * LW * PXB);        // filter row t; the lane's kx = kk is in nbase
        } else if constexpr (MODE == 1) {
#pragma unroll
            for (int t = 0; t < CPS; ++t) {
                int tap = 2 * t + (kk >> 1);
                tap = tap < TAPS ? tap : TAPS - 1;                       // padded slot: zero weights, finite data
                const int ky = tap / KW, kx = tap - ky * KW;
                chunk(t, (ky * LW + kx) * PXB);
            }
        } else {
            // one filter row per iteration of a rolled loop: fully unrolled, all 9 (16) chunks' fragment reads are hoisted to the
            // front and the kernel needs ~250 VGPRs (one block per SIMD)
            if constexpr (MT * NT >= 16) {                   // 40 registers of fragments per chunk: one chunk per iteration
#pragma unroll 1
                for (int ky = 0; ky < KH; ++ky)
#pragma unroll 1
                    for (int kx = 0; kx < KW; ++kx) chunk(ky * KW + kx, (ky * LW + kx) * PXB);
            } else {
#pragma unroll 1
                for (int ky = 0; ky < KH; ++ky)
#pragma unroll
                    for (int kx = 0; kx < KW; ++kx) chunk(ky * KW + kx, (ky * LW + kx) * PXB);
            }
        }
    }

    // ---- epilogue: lane = pixel (column block, j), 4 consecutive output channels 16 (mt0 + m) + 4 kk ----
    const int cout = a.cout;
    const int Wp = (W + 1) >> 1;
#pragma unroll
    for (int m = 0; m < MT; ++m) {
        const int c = (mt0 + m) * 16 + kk * 4;
        const bool cok = c < cout;                           // cout is a multiple of 4
        const f32x4 b4 = cok ? *reinterpret_cast<const f32x4*>(a.bias + c) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int n = 0; n < NT; ++n) {
            const int id = wn * NT + n;
            const int y = y0 + (id >> 1), x = x0 + (id & 1) * 16 + j;
            const bool ok = cok && y < H && x < W;
            const size_t p = ((size_t)min(y, H - 1) * W + min(x, W - 1)) * cout + (cok ? c : 0);
            f32x4 v = acc[m][n] + b4;
            if (P.res) v += unpack_bf16x4(*reinterpret_cast<const u32x2*>(P.res + p));
            if (a.relu_out) v = relu4(v);
            // the pool takes its maximum over the ROUNDED values (what a separate pool kernel would read back)
            const u32x2 pk = pack_bf16x4(v);
            acc[m][n] = unpack_bf16x4(pk);
            if (ok && !a.skip_full) *reinterpret_cast<u32x2*>(P.out + p) = pk;
        }
        if (P.pool) {
            // n-tiles n, n + 2 of a wave are the same 16 columns of rows y, y + 1 (y even); column partner in lane j ^ 1
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

// ------------------------------------------------------------------------------------------------
// resb_tail_kernel<C>: the tail of a residual block in ONE kernel (ARU_v1.py:212-227 / :266-281):
//     r = relu(t); r = relu(convR_0(r)); r = relu(convR_1(r)); out = relu(convR_2(r) + t)   [+ maxpool2(out)]
// for the 8- and 16-channel levels, where the layer-by-layer form moves 7 tensors through HBM and this one 2 (+ pool).
// Block = 16 x 32 output pixels.  relu(t) with a 3-pixel halo (22 x 38) goes to LDS, stage 1 writes its 20 x 36 result to a
// second LDS region, stage 2 its 18 x 34 result over the (dead) input region, stage 3 leaves through the registers.
// Positions outside the image are written as zeros (SAME padding applies to every conv of the chain).  The n-tiles of a
// stage are 16 CONSECUTIVE pixels of the flattened stage region (no waste on widths that are not multiples of 16).
// ------------------------------------------------------------------------------------------------
struct ResBProb {
    const bf16_t* t;       // [H,W,C] conv1 output (pre-ReLU)
    bf16_t* out;           // [H,W,C]
    bf16_t* pool;          // maxpool2(out) or nullptr
    int H, W;
    int tiles_x, tile_begin;
};
struct ResBArgs {
    ResBProb p[MAXP];
    int nprob;
    const u32x4* wpk;      // [3 convs][CPC chunks][64 lanes] x 16 bytes
    const float* bias;     // [3][C]
};
constexpr int RB_TH = 16, RB_TW = 32;

template <int C>
__global__ __launch_bounds__(256, C == 8 ? 4 : 3) void resb_tail_kernel(const ResBArgs a) {
    static_assert(C == 8 || C == 16, "8- and 16-channel levels");
    constexpr int PXB = C * 2;
    constexpr int CPC = C == 8 ? 3 : 5;                       // K chunks per conv (C == 8: filter rows; C == 16: tap pairs)
    constexpr int H0 = RB_TH + 6, W0 = RB_TW + 6, H1 = RB_TH + 4, W1 = RB_TW + 4, H2 = RB_TH + 2, W2 = RB_TW + 2;
    constexpr int SLACK = 4;                                  // pixels a padded tap / a clamped tail lane may read past a region
    constexpr int R0B = (H0 * W0 + SLACK) * PXB, R1B = (H1 * W1 + SLACK) * PXB;
    __shared__ __attribute__((aligned(16))) unsigned char lds[R0B + R1B];
    unsigned char* const r0 = lds;
    unsigned char* const r1 = lds + R0B;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int j = lane & 15, kk = lane >> 4;
    int pi = 0;
    while (pi + 1 < a.nprob && (int)blockIdx.x >= a.p[pi + 1].tile_begin) ++pi;
    const ResBProb& P = a.p[pi];
    const int tile = blockIdx.x - P.tile_begin;
    const int ty = tile / P.tiles_x, tx = tile - ty * P.tiles_x;
    const int x0 = tx * RB_TW, y0 = ty * RB_TH;
    const int H = P.H, W = P.W;

    // ---- relu(t) halo tile -> r0 (zero outside the image); slack pixels zeroed ----
    {
        constexpr int SUBS = C / 8, NU = H0 * W0 * SUBS, NLOAD = (NU + 255) / 256;
        u32x4 st[NLOAD];
        unsigned mask = 0;
#pragma unroll
        for (int i = 0; i < NLOAD; ++i) {
            const int u = min(tid + i * 256, NU - 1);
            const int pix = u / SUBS, sub = u - pix * SUBS;
            const int ly = pix / W0, lx = pix - ly * W0;
            const int gy = y0 - 3 + ly, gx = x0 - 3 + lx;
            st[i] = *reinterpret_cast<const u32x4*>(P.t + ((size_t)min(max(gy, 0), H - 1) * W + min(max(gx, 0), W - 1)) * C + sub * 8);
            mask |= ((gy >= 0 && gy < H && gx >= 0 && gx < W) ? 1u : 0u) << i;
        }
#pragma unroll
        for (int i = 0; i < NLOAD; ++i) {
            const int u = tid + i * 256;
            if (u < NU) *reinterpret_cast<u32x4*>(r0 + u * 16) = ((mask >> i) & 1u) ? relu_bf16x8(st[i]) : u32x4{0u, 0u, 0u, 0u};
        }
        if (tid < SLACK * SUBS) {
            *reinterpret_cast<u32x4*>(r0 + (H0 * W0 * SUBS + tid) * 16) = u32x4{0u, 0u, 0u, 0u};
            *reinterpret_cast<u32x4*>(r1 + (H1 * W1 * SUBS + tid) * 16) = u32x4{0u, 0u, 0u, 0u};
        }
    }
    // byte offset of the lane's share of K chunk t of a 3x3 window whose top-left pixel is at pixel index `base` of a region
    // that is WIN pixels wide
    auto tap_off = [&](int t, int WIN) {
        if constexpr (C == 8) return (t * WIN + kk) * PXB;                      // filter row t, kx = kk (kk == 3: zero weights)
        else {
            int tap = 2 * t + (kk >> 1);
            tap = tap < 9 ? tap : 8;
            const int ky = tap / 3, kx = tap - ky * 3;
            return (ky * WIN + kx) * PXB + (kk & 1) * 16;
        }
    };
    const u32x4* __restrict__ wl = a.wpk + lane;
    u32x4 af[CPC];
#pragma unroll
    for (int t = 0; t < CPC; ++t) af[t] = wl[t * 64];
    __syncthreads();

    // ---- stages 1 and 2: LDS -> LDS ----
    auto mid_stage = [&](const unsigned char* src, int WIN, unsigned char* dst, int HO, int WO, int halo, const float* bias) {
        const int npix = HO * WO;
        const f32x4 b4 = (C == 16 || kk < 2) ? *reinterpret_cast<const f32x4*>(bias + kk * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
        for (int tl = wave; tl * 16 < npix; tl += 4) {
            const int q = tl * 16 + j, qc = min(q, npix - 1);
            const int oy = qc / WO, ox = qc - oy * WO;
            const int base = (oy * WIN + ox) * PXB;
            u32x4 bfr[CPC];
#pragma unroll
            for (int t = 0; t < CPC; ++t) bfr[t] = *reinterpret_cast<const u32x4*>(src + base + tap_off(t, WIN));
            f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int t = 0; t < CPC; ++t) acc = mfma_bf16_k32(af[t], bfr[t], acc);
            const int gy = y0 - halo + oy, gx = x0 - halo + ox;
            const bool inside = gy >= 0 && gy < H && gx >= 0 && gx < W;
            const f32x4 v = inside ? relu4(acc + b4) : f32x4{0.f, 0.f, 0.f, 0.f};
            if (q < npix && (C == 16 || kk < 2)) *reinterpret_cast<u32x2*>(dst + q * PXB + kk * 8) = pack_bf16x4(v);
        }
    };
    mid_stage(r0, W0, r1, H1, W1, 2, a.bias);
#pragma unroll
    for (int t = 0; t < CPC; ++t) af[t] = wl[(CPC + t) * 64];
    __syncthreads();
    mid_stage(r1, W1, r0, H2, W2, 1, a.bias + C);             // r0 (the input tile) is dead: its space takes stage 2's result
#pragma unroll
    for (int t = 0; t < CPC; ++t) af[t] = wl[(2 * CPC + t) * 64];
    __syncthreads();

    // ---- stage 3: LDS -> registers -> HBM.  Unit = (row pair, 16-column block): rows in registers for the pool ----
    {
        const float* bias = a.bias + 2 * C;
        const f32x4 b4 = (C == 16 || kk < 2) ? *reinterpret_cast<const f32x4*>(bias + kk * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
        const int Wp = (W + 1) >> 1;
        for (int u = wave; u < (RB_TH / 2) * 2; u += 4) {
            const int rp = u >> 1, cb = u & 1;
            const int oy = 2 * rp, ox = cb * 16 + j;
            f32x4 acc2[2];
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                const int base = ((oy + r) * W2 + ox) * PXB;
                u32x4 bfr[CPC];
#pragma unroll
                for (int t = 0; t < CPC; ++t) bfr[t] = *reinterpret_cast<const u32x4*>(r0 + base + tap_off(t, W2));
                f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int t = 0; t < CPC; ++t) acc = mfma_bf16_k32(af[t], bfr[t], acc);
                acc2[r] = acc;
            }
            const int x = x0 + ox;
            const bool cok = C == 16 || kk < 2;
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                const int y = y0 + oy + r;
                const size_t p = ((size_t)min(y, H - 1) * W + min(x, W - 1)) * C + (cok ? kk * 4 : 0);
                f32x4 v = acc2[r] + b4 + unpack_bf16x4(*reinterpret_cast<const u32x2*>(P.t + p));
                v = relu4(v);
                const u32x2 pk = pack_bf16x4(v);
                acc2[r] = unpack_bf16x4(pk);
                if (cok && y < H && x < W) *reinterpret_cast<u32x2*>(P.out + p) = pk;
            }
            if (P.pool) {
                const int y = y0 + oy;
                f32x4 mm = (y + 1 < H) ? max4(acc2[0], acc2[1]) : acc2[0];
                const f32x4 nb = f32x4{lane_xor1(mm.x), lane_xor1(mm.y), lane_xor1(mm.z), lane_xor1(mm.w)};
                if (x + 1 < W) mm = max4(mm, nb);
                if ((j & 1) == 0 && cok && y < H && x < W)
                    *reinterpret_cast<u32x2*>(P.pool + ((size_t)(y >> 1) * Wp + (x >> 1)) * C + kk * 4) = pack_bf16x4(mm);
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// deconvb_kernel: conv2d_transpose 3x3, stride 2, SAME (layers.py:362), bias + ReLU, bf16 in / out.
//   out[i] = sum over (o, k) with 2 o + k - pb = i of in[o] W[k]   (per dimension; pb = pad_before of the SAME rule).
// With I = i + pb: I even <- k = 0 (o = I/2) and k = 2 (o = I/2 - 1); I odd <- k = 1 (o = (I-1)/2).  So the 2 x 2 outputs
// (parity classes) of input-grid position (Y, X) take 4 / 2 / 2 / 1 taps from the positions (Y - dy, X - dx), dy, dx in {0,1}:
// an n-tile = 16 consecutive X of one Y, 4 accumulators (classes) per m-tile, the 4 shifted B fragments shared by the classes.
//   MODE 1 (Cin == 16): K chunk = (dy, both dx) x 16 channels.   MODE 2 (Cin % 32 == 0): K chunk = (shift, 32-channel group).
// Block = 8 x 16 input positions = 16 x 32 output pixels x 16 MT channels, which leave through an LDS tile as whole rows.
// ------------------------------------------------------------------------------------------------
struct DeconvBProb {
    const bf16_t* in;      // [Hi,Wi,cin]
    bf16_t* out;           // [Ho,Wo,cout]
    int Hi, Wi, Ho, Wo;
    int pbh, pbw;
    int tiles_x, tile_begin;
};
struct DeconvBArgs {
    DeconvBProb p[MAXP];
    int nprob;
    const u32x4* wpk;      // MODE 2: [G][tap 0..8][mtile][lane]; MODE 1: [frag 0..5][mtile][lane]
    const float* bias;
    int cin, cout, mtiles, groups;
    int relu_out;
};
constexpr int DCB_TH = 8, DCB_TW = 16;                        // input positions per block

template <int MODE, int MT>
__global__ __launch_bounds__(256, 2) void deconvb_kernel(const DeconvBArgs a) {
    constexpr int LH = DCB_TH + 1, LW = DCB_TW + 1;
    constexpr int PLANE = LH * LW * 32;
    constexpr int NPL = MODE == 2 ? 2 : 1;
    constexpr int SUBS = MODE == 2 ? 4 : 2;
    constexpr int NU = LH * LW * SUBS, NLOAD = (NU + 255) / 256;
    constexpr int OC = 16 * MT;                               // output channels of this block
    constexpr int OUTB = 2 * DCB_TH * 2 * DCB_TW * OC * 2;    // output tile in bytes
    __shared__ __attribute__((aligned(16))) unsigned char lds[NPL * PLANE + OUTB];
    unsigned char* const otile = lds + NPL * PLANE;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int j = lane & 15, kk = lane >> 4;
    int pi = 0;
    while (pi + 1 < a.nprob && (int)blockIdx.x >= a.p[pi + 1].tile_begin) ++pi;
    const DeconvBProb& P = a.p[pi];
    const int tile = blockIdx.x - P.tile_begin;
    const int ty = tile / P.tiles_x, tx = tile - ty * P.tiles_x;
    const int X0 = tx * DCB_TW, Y0 = ty * DCB_TH, mt0 = blockIdx.y * MT;
    const int Hi = P.Hi, Wi = P.Wi, cin = a.cin;

    // n-tile = one input row of the tile (16 positions); wave w owns rows w and w + 4; classes c = 2 py + px
    f32x4 acc[2][4][MT];
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int m = 0; m < MT; ++m) acc[r][c][m] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int ngroups = MODE == 2 ? a.groups : 1;
    const u32x4* __restrict__ wbase = a.wpk + (size_t)mt0 * 64 + lane;
    const size_t wstride = (size_t)a.mtiles * 64;
    for (int g = 0; g < ngroups; ++g) {
        // input tile with one halo row / column at the top / left: LDS (ly, lx) = input (Y0 - 1 + ly, X0 - 1 + lx)
        u32x4 st[NLOAD];
        unsigned mask = 0;
#pragma unroll
        for (int i = 0; i < NLOAD; ++i) {
            const int u = min(tid + i * 256, NU - 1);
            const int pix = u / SUBS, sub = u - pix * SUBS;
            const int ly = pix / LW, lx = pix - ly * LW;
            const int gy = Y0 - 1 + ly, gx = X0 - 1 + lx;
            st[i] = *reinterpret_cast<const u32x4*>(P.in + ((size_t)min(max(gy, 0), Hi - 1) * Wi + min(max(gx, 0), Wi - 1)) * cin + g * 32 + sub * 8);
            mask |= ((gy >= 0 && gy < Hi && gx >= 0 && gx < Wi) ? 1u : 0u) << i;
        }
        if (g > 0) __syncthreads();
#pragma unroll
        for (int i = 0; i < NLOAD; ++i) {
            const int u = tid + i * 256;
            if (u < NU) {
                const int pix = u / SUBS, sub = u - pix * SUBS;
                *reinterpret_cast<u32x4*>(lds + (sub >> 1) * PLANE + pix * 32 + (sub & 1) * 16) = ((mask >> i) & 1u) ? st[i] : u32x4{0u, 0u, 0u, 0u};
            }
        }
        __syncthreads();
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const int Yl = wave + 4 * r;                       // local input row of this n-tile
            if constexpr (MODE == 2) {
                const u32x4* __restrict__ wg = wbase + (size_t)g * 9 * wstride;
#pragma unroll
                for (int dy = 0; dy < 2; ++dy)
#pragma unroll
                    for (int dx = 0; dx < 2; ++dx) {
                        const u32x4 b = *reinterpret_cast<const u32x4*>(lds + (kk >> 1) * PLANE + ((Yl + 1 - dy) * LW + (j + 1 - dx)) * 32 + (kk & 1) * 16);
#pragma unroll
                        for (int py = 0; py < 2; ++py) {
                            if (dy == 1 && py == 1) continue;
#pragma unroll
                            for (int px = 0; px < 2; ++px) {
                                if (dx == 1 && px == 1) continue;
                                const int ky = py ? 1 : (dy ? 2 : 0), kx = px ? 1 : (dx ? 2 : 0);
#pragma unroll
                                for (int m = 0; m < MT; ++m)
                                    acc[r][2 * py + px][m] = mfma_bf16_k32(wg[(size_t)(ky * 3 + kx) * wstride + (size_t)m * 64], b, acc[r][2 * py + px][m]);
                            }
                        }
                    }
            } else {
                // fragments: f = 0..3: dy = 0, class (py, px) = (f >> 1, f & 1); f = 4, 5: dy = 1, class (0, f & 1)
#pragma unroll
                for (int dy = 0; dy < 2; ++dy) {
                    const u32x4 b = *reinterpret_cast<const u32x4*>(lds + ((Yl + 1 - dy) * LW + (j + 1 - (kk >> 1))) * 32 + (kk & 1) * 16);
#pragma unroll
                    for (int f = (dy ? 4 : 0); f < (dy ? 6 : 4); ++f) {
                        const int cls = dy ? (f & 1) : f;
#pragma unroll
                        for (int m = 0; m < MT; ++m)
                            acc[r][cls][m] = mfma_bf16_k32(wbase[(size_t)f * wstride + (size_t)m * 64], b, acc[r][cls][m]);
                    }
                }
            }
        }
    }
    // ---- accumulators -> bias, ReLU, bf16 -> output tile [16][32][OC] in LDS ----
#pragma unroll
    for (int m = 0; m < MT; ++m) {
        const int c = (mt0 + m) * 16 + kk * 4;
        const f32x4 b4 = c < a.cout ? *reinterpret_cast<const f32x4*>(a.bias + c) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const int Yl = wave + 4 * r;
#pragma unroll
            for (int cls = 0; cls < 4; ++cls) {
                f32x4 v = acc[r][cls][m] + b4;
                if (a.relu_out) v = relu4(v);
                const int orow = 2 * Yl + (cls >> 1), ocol = 2 * j + (cls & 1);
                *reinterpret_cast<u32x2*>(otile + ((orow * 2 * DCB_TW + ocol) * OC + m * 16 + kk * 4) * 2) = pack_bf16x4(v);
            }
        }
    }
    __syncthreads();
    // ---- whole rows out: 16-byte units; tile pixel (orow, ocol) is output pixel (2 Y0 - pbh + orow, 2 X0 - pbw + ocol) ----
    constexpr int UPP = OC / 8;                               // 16-byte units per pixel
    const int nvalid = min(OC, a.cout - mt0 * 16);            // real channels of this block (a multiple of 8)
    for (int u = tid; u < 2 * DCB_TH * 2 * DCB_TW * UPP; u += 256) {
        const int pix = u / UPP, sub = u - pix * UPP;
        const int orow = pix / (2 * DCB_TW), ocol = pix - orow * (2 * DCB_TW);
        const int y = 2 * Y0 - P.pbh + orow, x = 2 * X0 - P.pbw + ocol;
        if (y >= 0 && y < P.Ho && x >= 0 && x < P.Wo && sub * 8 < nvalid)
            *reinterpret_cast<u32x4*>(P.out + ((size_t)y * P.Wo + x) * a.cout + mt0 * 16 + sub * 8) = *reinterpret_cast<const u32x4*>(otile + u * 16);
    }
}

// channel sum [H,W,8] bf16 -> [H,W] fp32 (upsample_simple's channel-summing half; same association as chansum_kernel)
struct PoolBProb {
    const bf16_t* in;
    float* out;
    int H, W;
    int blk_begin, pad_;
};
struct PoolBArgs {
    PoolBProb p[MAXP];
    int nprob;
    int C;
};
__global__ __launch_bounds__(256) void chansumb_kernel(const PoolBArgs a) {
    int pi = 0;
    while (pi + 1 < a.nprob && (int)blockIdx.x >= a.p[pi + 1].blk_begin) ++pi;
    const PoolBProb& P = a.p[pi];
    const size_t total = (size_t)P.H * P.W;
    const size_t base = (size_t)(blockIdx.x - P.blk_begin) * POOL_ITEMS;
    for (int k = 0; k < POOL_ITEMS / 256; ++k) {
        const size_t i = base + k * 256 + threadIdx.x;
        if (i >= total) break;
        float s = 0.f;
        for (int c8 = 0; c8 < a.C; c8 += 8) {
            const u32x4 v = *reinterpret_cast<const u32x4*>(P.in + i * a.C + c8);
            const f32x4 lo = unpack_bf16x4(u32x2{v.x, v.y}), hi = unpack_bf16x4(u32x2{v.z, v.w});
            s = (((((((s + lo.x) + lo.y) + lo.z) + lo.w) + hi.x) + hi.y) + hi.z) + hi.w;
        }
        P.out[i] = s;
    }
}

}  // namespace asep
