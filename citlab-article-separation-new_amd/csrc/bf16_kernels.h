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
#include <type_traits>
#include <utility>

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
// a float rounded to bfloat16 (nearest even) and widened again
__device__ __forceinline__ float round_bf16(float v) { return __uint_as_float(pack_bf16x2(v, 0.f) << 16); }
__device__ __forceinline__ f32x4 unpack_bf16x4(u32x2 p) {
    return f32x4{__uint_as_float(p.x << 16), __uint_as_float(p.x & 0xffff0000u), __uint_as_float(p.y << 16), __uint_as_float(p.y & 0xffff0000u)};
}
// The graph variants' activations for the bf16 kernels: act4 of aru_kernels.h with the HARDWARE exponential (v_exp_f32, 1 ulp) in elu -- its result is rounded
// to bfloat16 (2^-9) right behind it; expf's ~20 instructions per value were 0.13-0.15 ms per level-0 block (round 6: elu page 3.50 -> see profiles/r6_variants)
__device__ __forceinline__ float act1b(float x, int mode) {
    return mode == 1 ? (x > 0.f ? x : __expf(x) - 1.f) : fmaxf(x, 0.f) + 0.1f * fminf(x, 0.f);
}
__device__ __forceinline__ f32x4 act4b(f32x4 v, int mode) { return f32x4{act1b(v.x, mode), act1b(v.y, mode), act1b(v.z, mode), act1b(v.w, mode)}; }
// ReLU on two packed bf16: a negative bf16 is a negative int16 (v_pk_max_i16 with 0)
__device__ __forceinline__ unsigned relu_bf16x2(unsigned x) {
    return __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_bit_cast(s16x2, x), s16x2{0, 0}));
}
// max of two packed pairs of NON-NEGATIVE bf16 (the integer order of their bit patterns is their float order): v_pk_max_i16
__device__ __forceinline__ unsigned pkmax_u16(unsigned a, unsigned b) {
    return __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_bit_cast(s16x2, a), __builtin_bit_cast(s16x2, b)));
}
__device__ __forceinline__ u32x4 relu_bf16x8(u32x4 v) { return u32x4{relu_bf16x2(v.x), relu_bf16x2(v.y), relu_bf16x2(v.z), relu_bf16x2(v.w)}; }
__device__ __forceinline__ f32x4 mfma_bf16_k32(u32x4 a, u32x4 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

// ------------------------------------------------------------------------------------------------
// Software pipeline over the PAIR SLOTS of a fused block's phase (res8f / res16f / res32_tail; round 5).  A slot = two tiles (two independent
// accumulator chains) = CP fragment reads each, 2 CP MFMAs, an epilogue (round, ReLU, LDS or HBM store).  Written slot by slot the compiler has
// to keep slot s + 1's LDS reads behind slot s's LDS stores (it cannot see that a phase's source and destination regions differ), so a wave ran
// read -> wait -> MFMAs -> wait -> epilogue -> store and only the other waves of its SIMD covered the latencies.  Here slot s + 2's reads are
// issued behind slot s's MFMAs (their registers are free then) and slot s - 1's epilogue stands beside slot s's MFMAs: simple vector instructions
// hide under a bf16 MFMA of the same wave, two to three per MFMA (scripts/ubench/bf16mfma_epilogue_coissue.hip).
//   ld(integral_constant<int, s>, FragPair&): issue slot s's reads;   st(integral_constant<int, s>, ra, rb): slot s's epilogue.
// ------------------------------------------------------------------------------------------------
template <int CP> struct FragPair { u32x4 a[CP], b[CP]; };
template <int I> using ic = std::integral_constant<int, I>;
template <class F, int... I> __device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<int, I...>) { (f(ic<I>{}), ...); }
template <int N, class F> __device__ __forceinline__ void static_for(F&& f) { static_for_impl(f, std::make_integer_sequence<int, N>{}); }
template <int CP>
__device__ __forceinline__ void mm_pair(const u32x4 (&w)[CP], const FragPair<CP>& f, f32x4 c0, f32x4& ra, f32x4& rb) {
    ra = c0; rb = c0;
#pragma unroll
    for (int t = 0; t < CP; ++t) { ra = mfma_bf16_k32(w[t], f.a[t], ra); rb = mfma_bf16_k32(w[t], f.b[t], rb); }
}
template <int NS, int CP, class LD, class ST>
__device__ __forceinline__ void pipe_slots(const u32x4 (&w)[CP], f32x4 c0, LD&& ld, ST&& st) {
    FragPair<CP> f[2];
    f32x4 ra[2], rb[2];
    ld(ic<0>{}, f[0]);
    if constexpr (NS > 1) ld(ic<1>{}, f[1]);
    static_for<NS>([&](auto sc) {
        constexpr int s = decltype(sc)::value;
        mm_pair<CP>(w, f[s & 1], c0, ra[s & 1], rb[s & 1]);
        if constexpr (s + 2 < NS) ld(ic<s + 2>{}, f[s & 1]);
        if constexpr (s >= 1) st(ic<s - 1>{}, ra[(s - 1) & 1], rb[(s - 1) & 1]);
    });
    st(ic<NS - 1>{}, ra[(NS - 1) & 1], rb[(NS - 1) & 1]);
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
    int act;               // graph variants (asep_aru_cfg.activation; relu_out is 0 then): 1 = elu, 2 = leaky (0.1), applied to the fp32 sums before
                           // the rounding to bf16 (round 5).  Such a launch takes the general epilogue; the ReLU graphs' fast one is untouched.
    XcdMap xm;             // XCD-aware block -> tile map (sched_tile)
};

// Waves: WM along the output channels x 4 / WM along the pixels; wave (wm, wn) owns m-tiles wm MT .. and n-tiles id = wn NT + n
// (row id >> 1, column block id & 1).  The A fragments (weights) of a stage are copied to LDS ONCE per block next to the halo
// tile and read from there by all waves: fetched per wave from L2 (first cut of this kernel) they cost 4 KB per 16 MFMAs and
// wave = the whole vector-memory path of a CU, and every chunk waited for an L2 round trip.
// RESP: the layer has a residual operand; its values are requested at the very start and become part of the accumulators' INITIAL value
// (bias + residual) behind the first barrier -- no registers held through the MFMA loops, no exposed latency (first cut: fetched in the epilogue,
// a residual layer took 74 us against 42 us for its twin without one; second: requested before the last stage's MFMAs into 2 MT NT registers,
// which kept the eight-wave form at one block per CU and the >= 64-channel convR_2 layers on the four-wave kernel: 160 against 117 us).
// NW = waves per block (4, or 8 for the >= 64-channel layers: twice the waves per CU over the same LDS tile -- those layers are
// short chains of dependent LDS reads and MFMAs, more waves overlap them).
// Debug builds (-DCVB_TRACE): s_memtime stamps of thread 0 of every 4th block of the 8-wave >= 64-channel instantiation, read back
// through asep_debug_cvb_trace by scripts/gpu_cvb_trace.py (DESIGN lesson 22); compiled out otherwise.
#if defined(CVB_TRACE)
__device__ unsigned long long g_cvb_trace[4096 * 8];
#define CVB_MARK(i) do { if (NW == 8 && MODE == 2 && !RESP && blockIdx.y == 0 && (blockIdx.x & 3) == 0 && (blockIdx.x >> 2) < 4096 && tid == 0) g_cvb_trace[(blockIdx.x >> 2) * 8 + (i)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define CVB_MARK(i) do { } while (0)
#endif
template <int KH, int KW, int MODE, int MT, int WM, int TH, int MINB, bool RESP = false, int NW = 4>
__global__ __launch_bounds__(64 * NW, MINB) void convb_kernel(const ConvBArgs a) {
    constexpr int NTH = 64 * NW;
    static_assert(MODE != 0 || (KH == 3 && KW == 3), "MODE 0 is the 3x3 conv with 8 input channels");
    static_assert(WM == 1 || WM == 2, "one or two waves along the output channels");
    constexpr int TW = 32, WN = NW / WM, NT = TH * 2 / WN, MTB = MT * WM;
    static_assert(NT % 4 == 0, "a wave owns whole row pairs (fused 2x2 pool)");
    constexpr int LH = TH + KH - 1, LW = TW + KW - 1 + (MODE == 0 ? 1 : 0);     // MODE 0 reads a 4th (zero-weight) column
    constexpr int PT = (KH - 1) / 2, PL = (KW - 1) / 2;                         // TF SAME: pad_before = (k-1)/2
    constexpr int TAPS = KH * KW;
    constexpr int SUBS = MODE == 0 ? 1 : (MODE == 1 ? 2 : 4);                   // 16-byte units per pixel and stage
    constexpr int PXB = MODE == 0 ? 16 : 32;                                    // bytes per pixel in a plane
    constexpr int PLANE = LH * LW * PXB;
    constexpr int NPL = MODE == 2 ? 2 : 1;
    constexpr int CPS = MODE == 0 ? KH : (MODE == 1 ? (TAPS + 1) / 2 : TAPS);   // K chunks per stage
    constexpr int NU = LH * LW * SUBS, NLOAD = (NU + NTH - 1) / NTH;
    constexpr int NWU = CPS * MTB * 64, NWLOAD = (NWU + NTH - 1) / NTH;             // 16-byte units of a stage's A fragments
    __shared__ __attribute__((aligned(16))) unsigned char lds[NPL * PLANE + NWU * 16];
    unsigned char* const wlds = lds + NPL * PLANE;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int j = lane & 15, kk = lane >> 4;
    CVB_MARK(0);
    const int wm = WM == 2 ? (wave & 1) : 0, wn = WM == 2 ? (wave >> 1) : wave;
    const int bid = sched_tile(a.xm);
    if (bid < 0) return;
    int pi = 0;
    pi = prob_of_tile(a, bid);
    const ConvBProb& P = a.p[pi];
    const int tile = bid - P.tile_begin;
    const int ty = tile / P.tiles_x, tx = tile - ty * P.tiles_x;
    const int x0 = tx * TW, y0 = ty * TH, mtb0 = blockIdx.y * MTB, mt0 = mtb0 + wm * MT;
    const int H = P.H, W = P.W;

    // the bias is the accumulators' initial value (rows of channels beyond cout: zero)
    const int cout = a.cout;
    f32x4 acc[MT][NT];
#pragma unroll
    for (int m = 0; m < MT; ++m) {
        const int c = (mt0 + m) * 16 + kk * 4;
        const f32x4 b4 = c < cout ? *reinterpret_cast<const f32x4*>(a.bias + c) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int n = 0; n < NT; ++n) acc[m][n] = b4;
    }

    // byte offset of (row, col + j) of each n-tile of this wave, plus the lane's channel half / plane
    int nbase[NT];
#pragma unroll
    for (int n = 0; n < NT; ++n) {
        const int id = wn * NT + n;
        nbase[n] = ((id >> 1) * LW + (id & 1) * 16 + j) * PXB + (MODE == 0 ? kk * 16 : (kk & 1) * 16 + (MODE == 2 ? (kk >> 1) * PLANE : 0));
    }
    const int ngroups = MODE == 2 ? a.groups : 1;
    u32x2 resv[RESP ? MT : 1][RESP ? NT : 1];
    if constexpr (RESP) {
        // (a uniform base + 32-bit byte offsets -- run_convb keeps residual tensors of 4 GB and more on the kernels without RESP --: written with
        // the epilogue's 64-bit address arithmetic the compiler shared it with the output addresses and kept 16 registers alive across the MFMA loops)
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
    const u32x4* __restrict__ wsrc = a.wpk + (size_t)mtb0 * 64;
    const size_t wstride = (size_t)a.mtiles * 64;
    const int mt_have = min(MTB, a.mtiles - mtb0);            // m-tiles of this block that exist (cout 8 / 16: one of MTB)

    // halo loader: the thread's NLOAD (pixel, 16-byte sub-block) slots do not depend on the stage: image pixel index (clamped
    // into the image: always a valid address), LDS byte offset and the inside-the-image mask are computed ONCE (the per-stage
    // loop had a division, four clamps and a 64-bit multiply per slot: the blocks are short, DESIGN lesson 9)
    int spix[NLOAD], slds[NLOAD];
    unsigned stmask = 0;
#pragma unroll
    for (int i = 0; i < NLOAD; ++i) {
        const int u = min(tid + i * NTH, NU - 1);
        const int pix = u / SUBS, sub = u - pix * SUBS;
        const int ly = pix / LW, lx = pix - ly * LW;
        const int gy = y0 - PT + ly, gx = x0 - PL + lx;
        spix[i] = min(max(gy, 0), H - 1) * W + min(max(gx, 0), W - 1);
        slds[i] = (sub >> 1) * PLANE + pix * PXB + (sub & 1) * 16;
        stmask |= ((gy >= 0 && gy < H && gx >= 0 && gx < W && tid + i * NTH < NU) ? 1u : 0u) << i;
    }
    const int sub0 = (tid % SUBS) * 8;                        // (the block size is a multiple of SUBS: the sub-block is the same for all slots)

    if constexpr (RESP) {                                    // (requested before the tile loader's index arithmetic above)
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int n = 0; n < NT; ++n) acc[m][n] += unpack_bf16x4(resv[m][n]);
    }
    CVB_MARK(1);
    for (int g = 0; g < ngroups; ++g) {
        // ---- stage g: halo tile of 32 (16, 8) input channels + the stage's A fragments -> LDS.  Requests first; zero padding /
        //      ReLU when the registers go to LDS.  (Requesting stage g + 1's tile behind the barrier that releases stage g's MFMAs -- the same
        //      registers, dead while they run -- was measured in round 5: the 64- / 128-channel layers unchanged, two level-1 / 2 layers 4-7 %
        //      slower; DESIGN_LESSONS 41) ----
        u32x4 st[NLOAD];
        {
            const int c = g * 32 + sub0;
            const bool from0 = c < a.c0;
            const bf16_t* __restrict__ src = concat_src(P.in0, P.in1, c, a.c0);
            const int cs = from0 ? a.c0 : a.c1;
#pragma unroll
            for (int i = 0; i < NLOAD; ++i) st[i] = *reinterpret_cast<const u32x4*>(src + (size_t)spix[i] * cs);
        }
        if (g > 0) __syncthreads();                          // the previous stage's readers are done
        // A fragments: global -> LDS without registers (global_load_lds_dwordx4: one wave-instruction copies 1 KB, lane i to
        // base + 16 i); whole waves, NWU is a multiple of 64
#pragma unroll
        for (int i = 0; i < NWLOAD; ++i) {
            const int u0 = i * NTH + wave * 64;              // wave-uniform
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
            if (i * NTH + NTH - 1 < NU || tid + i * NTH < NU) {
                u32x4 v = ((stmask >> i) & 1u) ? st[i] : u32x4{0u, 0u, 0u, 0u};
                if (a.relu_in) v = relu_bf16x8(v);
                *reinterpret_cast<u32x4*>(lds + slds[i]) = v;
            }
        }
        __syncthreads();
        if (g == 0) CVB_MARK(2);
        if (g + 1 == ngroups) CVB_MARK(4);
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
        if (g == 0) CVB_MARK(3);
        if (g + 1 == ngroups) CVB_MARK(5);
    }

    // ---- epilogue: lane = pixel (column block, j), 4 consecutive output channels 16 (mt0 + m) + 4 kk ----
    const int Wp = (W + 1) >> 1;
    if (y0 + TH <= H && x0 + TW <= W && (mtb0 + MTB) * 16 <= cout && a.relu_out && !a.pool_f32) {
        // the output tile lies inside the image, whole m-tiles, ReLU layer (every layer but the block-opening conv1): per-lane base
        // pointers + row strides instead of a bounds test, two clamps and a 64-bit multiply per accumulator; ReLU AFTER the rounding
        // (one v_pk_max_i16 per two values); the pool compares the bf16 bit patterns (non-negative values: integer order = float order)
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
        CVB_MARK(6);
#if defined(CVB_TRACE)
        __builtin_amdgcn_s_waitcnt(0x0f70);                  // vmcnt(0): the tile's stores have left
#endif
        CVB_MARK(7);
        return;
    }
#pragma unroll
    for (int m = 0; m < MT; ++m) {
        const int c = (mt0 + m) * 16 + kk * 4;
        const bool cok = c < cout;                           // cout is a multiple of 4
#pragma unroll
        for (int n = 0; n < NT; ++n) {
            const int id = wn * NT + n;
            const int y = y0 + (id >> 1), x = x0 + (id & 1) * 16 + j;
            const bool ok = cok && y < H && x < W;
            const size_t p = ((size_t)min(y, H - 1) * W + min(x, W - 1)) * cout + (cok ? c : 0);
            f32x4 v = acc[m][n];
            if constexpr (!RESP) { if (P.res) v += unpack_bf16x4(*reinterpret_cast<const u32x2*>(P.res + p)); }
            if (a.relu_out) v = relu4(v);
            else if (a.act) v = act4b(v, a.act);
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
    int ntiles;            // res32_tail_kernel: all problems' tiles (its resident blocks walk them)
    XcdMap xm;             // XCD-aware block -> tile map (sched_tile)
    const int32_t* sched;  // res32_tail_kernel (persistent): work unit -> tile table or nullptr
};
constexpr int RB_TH = 16, RB_TW = 32;

template <int C>
struct ResBLayout {
    static constexpr int PXB = C * 2, SLACK = 4;
    static constexpr int R0B = ((RB_TH + 6) * (RB_TW + 6) + SLACK) * PXB, R1B = ((RB_TH + 4) * (RB_TW + 4) + SLACK) * PXB, BYTES = R0B + R1B;
};

// the tail of one tile (any position; `lds` holds ResBLayout<C>::BYTES)
// ACT: the block's activation (0 ReLU; 1 elu / 2 leaky of the graph variants: on the fp32 sums before the one rounding, like the layer-by-layer path)
template <int C, int ACT = 0>
__device__ __forceinline__ void resb_tail_tile(const ResBArgs& a, const ResBProb& P, int x0, int y0, unsigned char* lds) {
    static_assert(C == 8 || C == 16, "8- and 16-channel levels");
    constexpr int PXB = C * 2;
    constexpr int CPC = C == 8 ? 3 : 5;                       // K chunks per conv (C == 8: filter rows; C == 16: tap pairs)
    constexpr int H0 = RB_TH + 6, W0 = RB_TW + 6, H1 = RB_TH + 4, W1 = RB_TW + 4, H2 = RB_TH + 2, W2 = RB_TW + 2;
    constexpr int SLACK = ResBLayout<C>::SLACK;               // pixels a padded tap / a clamped tail lane may read past a region
    constexpr int R0B = ResBLayout<C>::R0B;
    unsigned char* const r0 = lds;
    unsigned char* const r1 = lds + R0B;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int j = lane & 15, kk = lane >> 4;
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
            const f32x4 v = inside ? (ACT ? act4b(acc + b4, ACT) : relu4(acc + b4)) : f32x4{0.f, 0.f, 0.f, 0.f};
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
                v = ACT ? act4b(v, ACT) : relu4(v);
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

template <int C, int ACT = 0>
__global__ __launch_bounds__(256, C == 8 ? 4 : 3) void resb_tail_kernel(const ResBArgs a) {
    const int bid = sched_tile(a.xm);
    __shared__ __attribute__((aligned(16))) unsigned char lds[ResBLayout<C>::BYTES];
    if (bid < 0) return;
    int pi = 0;
    pi = prob_of_tile(a, bid);
    const ResBProb& P = a.p[pi];
    const int tile = bid - P.tile_begin;
    const int ty = tile / P.tiles_x, tx = tile - ty * P.tiles_x;
    resb_tail_tile<C, ACT>(a, P, tx * RB_TW, ty * RB_TH, lds);
}

// ------------------------------------------------------------------------------------------------
// res32_tail_kernel: the tail of a residual block at 32 channels (level 2) in one kernel -- resb_tail_tile's general form (flattened
// n-tiles of 16 pixels over the stage regions, zeros outside the image) with two m-tiles, K chunk = one tap x 32 channels, the
// regions stored as two planes of 16 channels (32 bytes per pixel: a lane's B fragment is one conflict-free ds_read_b128, as in
// convb_kernel MODE 2), eight waves per block over 100 KB of LDS (one block per CU).  Layer by layer these three convolutions
// run at the HBM rate (the 32-channel layers move 563 MB per 4-page launch at 4.7 TB/s, lesson 22); fused they read t and write
// the output once.  An n-tile carries 18 MFMAs, so the general form's index arithmetic per n-tile is affordable here (at 8 / 16
// channels it was not: res8f / res16f).
// ------------------------------------------------------------------------------------------------
struct Res32Layout {
    static constexpr int C = 32, SLACK = 4;
    static constexpr int H0 = RB_TH + 6, W0 = RB_TW + 6, H1 = RB_TH + 4, W1 = RB_TW + 4, H2 = RB_TH + 2, W2 = RB_TW + 2;
    static constexpr int P0 = (H0 * W0 + SLACK) * 32, P1 = (H1 * W1 + SLACK) * 32;     // bytes of one 16-channel plane of region 0 / 1
    static constexpr int WB = 3 * 9 * 2 * 64 * 16;            // the three convs' A fragments: [conv][tap][m-tile][lane] x 16 bytes = 54 KB
    static constexpr int W_OFF = 2 * P0 + 2 * P1, BYTES = W_OFF + WB;
};
// One resident block per CU walks the tiles blockIdx.x, + gridDim.x, ...: the A fragments of the three convolutions are copied to LDS
// once (LDS-DMA) and a stage's 18 fragments come from there into registers (as one-shot blocks fetching them from L2 per stage the
// kernel ran at the speed of the three separate layers: with one 100 KB block per CU nothing covered the per-stage round trips and
// the 53 KB fill); the next tile's window is requested into registers right after the current one went to LDS.
// ACT (round 6): 0 = the ReLU graphs (ReLU and pool on the packed values); 1 elu / 2 leaky: the activation on the fp32 sums before the rounding, float maxima in the pool
template <int ACT = 0>
__global__ __launch_bounds__(512, 1) void res32_tail_kernel(const ResBArgs a) {
    typedef Res32Layout L;
    constexpr int C = 32, NW = 8, NTH = 512;
    constexpr int H0 = L::H0, W0 = L::W0, H1 = L::H1, W1 = L::W1, H2 = L::H2, W2 = L::W2, P0 = L::P0, P1 = L::P1, SLACK = L::SLACK;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    unsigned char* const r0 = lds;                            // region 0 (22 x 38), later stage 2's result (18 x 34): planes at +0, +P0
    unsigned char* const r1 = lds + 2 * P0;                   // region 1 (20 x 36): planes at +0, +P1
    const unsigned char* const wl = lds + L::W_OFF;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int j = lane & 15, kk = lane >> 4;
    if ((int)blockIdx.x >= a.ntiles) return;

    // ---- once per block: A fragments -> LDS, biases -> registers, slack pixels zeroed ----
    {
        constexpr int NWU = L::WB / 16;                       // 16-byte units: a multiple of 64
        for (int u0 = wave * 64; u0 < NWU; u0 += NTH)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(a.wpk + u0 + lane),
                                             (__attribute__((address_space(3))) void*)(lds + L::W_OFF + u0 * 16), 16, 0, 0);
    }
    f32x4 biasw[3][2];
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
        for (int m = 0; m < 2; ++m) biasw[t][m] = *reinterpret_cast<const f32x4*>(a.bias + t * C + m * 16 + kk * 4);
    if (tid < SLACK * 4) {                                    // slack pixels behind the regions: read by clamped tail lanes, must be finite
        const int pl = (tid >> 1) & 1, off = (tid >> 2) * 32 + (tid & 1) * 16;
        *reinterpret_cast<u32x4*>(r0 + pl * P0 + H0 * W0 * 32 + off) = u32x4{0u, 0u, 0u, 0u};
        *reinterpret_cast<u32x4*>(r1 + pl * P1 + H1 * W1 * 32 + off) = u32x4{0u, 0u, 0u, 0u};
    }
    // byte offset of the lane's share of tap t (8 of the pixel's 32 channels: plane kk >> 1, half kk & 1) of a 3x3 window in a region
    // WIN pixels wide whose planes are PL bytes apart
    auto tap_off = [&](int t, int WIN, int PL) { return ((t / 3) * WIN + (t % 3)) * 32 + (kk & 1) * 16 + (kk >> 1) * PL; };
    u32x4 af[9][2];
    auto stage_weights = [&](int st) {
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int m = 0; m < 2; ++m) af[t][m] = *reinterpret_cast<const u32x4*>(wl + (((st * 9 + t) * 2 + m) * 64 + lane) * 16);
    };

    // ---- the tile walk; the halo tile of relu(t) of the NEXT tile is requested while the current one is computed ----
    constexpr int NU = H0 * W0 * 4, NLOAD = (NU + NTH - 1) / NTH;
    const int sub = tid & 3;                                  // (512 is a multiple of 4: the channel block is the same for all slots)
    u32x4 st[NLOAD];
    unsigned mask = 0;
    auto locate = [&](int unit, int& pi, int& x0, int& y0) {
        const int t = a.sched ? a.sched[unit] : unit;
        pi = 0;
        pi = prob_of_tile(a, t);
        const int tile = t - a.p[pi].tile_begin;
        const int ty = tile / a.p[pi].tiles_x, tx = tile - ty * a.p[pi].tiles_x;
        x0 = tx * RB_TW; y0 = ty * RB_TH;
    };
    auto request = [&](int pi, int x0, int y0) {
        const ResBProb& P = a.p[pi];
        mask = 0;
        int spix[NLOAD];
#pragma unroll
        for (int i = 0; i < NLOAD; ++i) {
            const int u = min(tid + i * NTH, NU - 1);
            const int pix = u >> 2;
            const int ly = pix / W0, lx = pix - ly * W0;
            const int gy = y0 - 3 + ly, gx = x0 - 3 + lx;
            spix[i] = min(max(gy, 0), P.H - 1) * P.W + min(max(gx, 0), P.W - 1);
            mask |= ((gy >= 0 && gy < P.H && gx >= 0 && gx < P.W) ? 1u : 0u) << i;
        }
#pragma unroll
        for (int i = 0; i < NLOAD; ++i) st[i] = *reinterpret_cast<const u32x4*>(P.t + (size_t)spix[i] * C + sub * 8);
    };
    __syncthreads();                                          // the fragments (copied by all waves) are in LDS
    int t = blockIdx.x, pi, x0, y0;
    locate(t, pi, x0, y0);
    request(pi, x0, y0);
    while (true) {
    const ResBProb& P = a.p[pi];
    const int H = P.H, W = P.W;
#pragma unroll
    for (int i = 0; i < NLOAD; ++i) {
        const int u = tid + i * NTH;
        if (u < NU)
            *reinterpret_cast<u32x4*>(r0 + (sub >> 1) * P0 + (u >> 2) * 32 + (sub & 1) * 16) = ((mask >> i) & 1u) ? relu_bf16x8(st[i]) : u32x4{0u, 0u, 0u, 0u};
    }
    const int xc = x0, yc = y0;                               // this tile; (t, pi, x0, y0) move on to the next one
    t += gridDim.x;
    const bool more = t < a.ntiles;
    if (more) { locate(t, pi, x0, y0); request(pi, x0, y0); }
    // residual operand of stage 3 (pre-ReLU t of the wave's two output units): requested now, used at the very end (fetched inside
    // stage 3 each unit waited for an HBM round trip behind its MFMAs)
    u32x2 resv[2][2][2];                                      // [unit][row][m-tile]
#pragma unroll
    for (int ui = 0; ui < 2; ++ui) {
        const int u = wave + ui * NW, oy = 2 * (u >> 1), ox = (u & 1) * 16 + j;
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const size_t p = ((size_t)min(yc + oy + r, H - 1) * W + min(xc + ox, W - 1)) * C + kk * 4;
#pragma unroll
            for (int m = 0; m < 2; ++m) resv[ui][r][m] = *reinterpret_cast<const u32x2*>(P.t + p + m * 16);
        }
    }
    stage_weights(0);
    __syncthreads();

    // ---- stages 1 and 2: LDS -> LDS.  The wave's tiles tl = wave, wave + 8, ...: ALL nine B fragments of a tile are requested together (one LDS
    //      round trip per tile; the first cut's schedule fetched them two at a time between the MFMA pairs -- five exposed round trips per tile with
    //      two waves per SIMD to cover them), and the NEXT tile's are requested behind the current tile's MFMAs, in front of its epilogue and
    //      stores (written in that order by hand: the compiler cannot move LDS reads over LDS stores) ----
    auto mid_stage = [&](const unsigned char* src, int WIN, int SPL, unsigned char* dst, int DPL, int HO, int WO, int halo, int sg) {
        const int npix = HO * WO;
        u32x4 bfr[9];
        auto request = [&](int tl) {
            const int qc = min(tl * 16 + j, npix - 1);
            const int oy = qc / WO, ox = qc - oy * WO;
            const unsigned char* base = src + (oy * WIN + ox) * 32;
#pragma unroll
            for (int k = 0; k < 9; ++k) bfr[k] = *reinterpret_cast<const u32x4*>(base + tap_off(k, WIN, SPL));
        };
        request(wave);
        for (int tl = wave; tl * 16 < npix; tl += NW) {
            const int q = tl * 16 + j, qc = min(q, npix - 1);
            const int oy = qc / WO, ox = qc - oy * WO;
            f32x4 acc[2] = {biasw[sg][0], biasw[sg][1]};
            __builtin_amdgcn_sched_barrier(0);               // (the requests stay in front of the MFMAs)
#pragma unroll
            for (int k = 0; k < 9; ++k)
#pragma unroll
                for (int m = 0; m < 2; ++m) acc[m] = mfma_bf16_k32(af[k][m], bfr[k], acc[m]);
            if ((tl + NW) * 16 < npix) request(tl + NW);      // (wave-uniform)
            __builtin_amdgcn_sched_barrier(0);
            const int gy = yc - halo + oy, gx = xc - halo + ox;
            const bool inside = gy >= 0 && gy < H && gx >= 0 && gx < W;
            if (q < npix) {
#pragma unroll
                for (int m = 0; m < 2; ++m) {
                    const u32x2 pk = pack_bf16x4(ACT ? act4b(acc[m], ACT) : acc[m]);
                    *reinterpret_cast<u32x2*>(dst + m * DPL + q * 32 + kk * 8) = inside ? (ACT ? pk : u32x2{relu_bf16x2(pk.x), relu_bf16x2(pk.y)}) : u32x2{0u, 0u};
                }
            }
        }
    };
    mid_stage(r0, W0, P0, r1, P1, H1, W1, 2, 0);
    stage_weights(1);
    __syncthreads();
    mid_stage(r1, W1, P1, r0, P0, H2, W2, 1, 1);              // region 0 (the input tile) is dead: its planes take stage 2's result
    stage_weights(2);
    __syncthreads();

    // ---- stage 3: LDS -> registers -> HBM.  Unit = (row pair, 16-column block): both rows in registers for the pool.  Four steps (unit, row) per
    //      wave, pipelined like the stages above: a row's nine fragments in one request, the next row's behind the current row's MFMAs ----
    {
        const int Wp = (W + 1) >> 1;
        static_assert((RB_TH / 2) * 2 == 2 * NW, "two output units per wave");
        u32x4 bfr[9];
        auto request = [&](int step) {
            const int u = wave + (step >> 1) * NW;
            const unsigned char* base = r0 + ((2 * (u >> 1) + (step & 1)) * W2 + (u & 1) * 16 + j) * 32;
#pragma unroll
            for (int k = 0; k < 9; ++k) bfr[k] = *reinterpret_cast<const u32x4*>(base + tap_off(k, W2, P0));
        };
        request(0);
        f32x4 acc2[2][2];
        static_for<4>([&](auto sc) {
            constexpr int step = decltype(sc)::value, ui = step >> 1, r = step & 1;
            // (bias + residual operand as the initial value: the order of convb_kernel's RESP form, which the layer-by-layer twin of this block uses)
            acc2[r][0] = biasw[2][0] + unpack_bf16x4(resv[ui][r][0]); acc2[r][1] = biasw[2][1] + unpack_bf16x4(resv[ui][r][1]);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int k = 0; k < 9; ++k)
#pragma unroll
                for (int m = 0; m < 2; ++m) acc2[r][m] = mfma_bf16_k32(af[k][m], bfr[k], acc2[r][m]);
            if constexpr (step < 3) request(step + 1);
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (r == 1) {
                const int u = wave + ui * NW;
                const int oy = 2 * (u >> 1), ox = (u & 1) * 16 + j;
                const int x = xc + ox;
#pragma unroll
                for (int m = 0; m < 2; ++m) {
                    u32x2 pk[2];
#pragma unroll
                    for (int rr = 0; rr < 2; ++rr) {
                        const int y = yc + oy + rr;
                        const size_t p = ((size_t)min(y, H - 1) * W + min(x, W - 1)) * C + m * 16 + kk * 4;
                        const u32x2 q = pack_bf16x4(ACT ? act4b(acc2[rr][m], ACT) : acc2[rr][m]);
                        pk[rr] = ACT ? q : u32x2{relu_bf16x2(q.x), relu_bf16x2(q.y)};
                        if (y < H && x < W) *reinterpret_cast<u32x2*>(P.out + p) = pk[rr];
                    }
                    if (P.pool) {
                        // 2 x 2 max on the packed values (non-negative bf16 order like their bit patterns); ceil mode at the right / bottom border
                        const int y = yc + oy;
                        u32x2 mm;
                        if constexpr (ACT != 0) {                // (values of either sign: maxima of the floats the packed values are)
                            f32x4 fm = (y + 1 < H) ? max4(unpack_bf16x4(pk[0]), unpack_bf16x4(pk[1])) : unpack_bf16x4(pk[0]);
                            const f32x4 fn = f32x4{lane_xor1(fm.x), lane_xor1(fm.y), lane_xor1(fm.z), lane_xor1(fm.w)};
                            if (x + 1 < W) fm = max4(fm, fn);
                            mm = pack_bf16x4(fm);
                        } else {
                            mm = (y + 1 < H) ? u32x2{pkmax_u16(pk[0].x, pk[1].x), pkmax_u16(pk[0].y, pk[1].y)} : pk[0];
                            const u32x2 nb = u32x2{__float_as_uint(lane_xor1(__uint_as_float(mm.x))), __float_as_uint(lane_xor1(__uint_as_float(mm.y)))};
                            if (x + 1 < W) mm = u32x2{pkmax_u16(mm.x, nb.x), pkmax_u16(mm.y, nb.y)};
                        }
                        if ((j & 1) == 0 && y < H && x < W)
                            *reinterpret_cast<u32x2*>(P.pool + ((size_t)(y >> 1) * Wp + (x >> 1)) * C + m * 16 + kk * 4) = mm;
                    }
                }
            }
        });
    }
    if (!more) break;
    __syncthreads();                                          // stage 3's readers of region 0 are done: the next tile may overwrite it
    }
}

// ------------------------------------------------------------------------------------------------
// res16f_kernel: resb_tail_kernel<16> for INTERIOR tiles (the 22 x 38 window of t inside the image), written for instruction count
// like res8f_kernel: an n-tile is 16 pixels of one ROW of a stage region (two per row + "remainder" tiles that gather columns
// 32 .. of several rows), every LDS address is a per-lane constant plus a compile-time offset, no inside-the-image tests, the bias
// is the accumulators' initial value, ReLU after the rounding (v_pk_max_i16), the residual operand is requested at the start.
// Border tiles take resb_tail_tile<16> (the general form) inside the same launch.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256, 3) void res16f_kernel(const ResBArgs a) {
    constexpr int C = 16, PXB = 32, CPC = 5;
    constexpr int TH = RB_TH, TW = RB_TW;
    constexpr int H0 = TH + 6, W0 = TW + 6, H1 = TH + 4, W1 = TW + 4, H2 = TH + 2, W2 = TW + 2;
    constexpr int R0B = H0 * W0 * PXB;
    __shared__ __attribute__((aligned(16))) unsigned char lds[ResBLayout<16>::BYTES];      // (the general form's layout is the larger one)
    unsigned char* const r0 = lds;
    unsigned char* const r1 = lds + R0B;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int j = lane & 15, kk = lane >> 4;
    const int bid = sched_tile(a.xm);
    if (bid < 0) return;
    int pi = 0;
    pi = prob_of_tile(a, bid);
    const ResBProb& P = a.p[pi];
    const int tile = bid - P.tile_begin;
    const int ty = tile / P.tiles_x, tx = tile - ty * P.tiles_x;
    const int x0 = tx * TW, y0 = ty * TH;
    const int H = P.H, W = P.W;
    if (!(y0 - 3 >= 0 && y0 + TH + 3 <= H && x0 - 3 >= 0 && x0 + TW + 3 <= W)) {             // border tile: the general form
        resb_tail_tile<16>(a, P, x0, y0, lds);
        return;
    }

    // ---- requests, cheapest-to-wait-for first (as in res8f_kernel): biases and convR_0's fragments (L2 hits, needed right behind the first
    //      barrier), the residual operand of stage 3, then the 22 x 38 halo tile.  Threads 0 .. 227 own the 16-byte unit t % 76 of the tile
    //      rows t / 76 + 3 k (a row = 38 pixels x 32 bytes, contiguous in HBM): one division per thread, a uniform base pointer + 32-bit byte
    //      offsets (run_resb_tail keeps tensors of 2^27 pixels and more on the general kernel), LDS addresses = base + immediates ----
    f32x4 biasw[3];
#pragma unroll
    for (int t = 0; t < 3; ++t) biasw[t] = *reinterpret_cast<const f32x4*>(a.bias + t * C + kk * 4);
    const u32x4* __restrict__ wl = a.wpk + lane;
    u32x4 af[CPC], ag[CPC];                                    // the fragments of the current / the next conv (requested a stage ahead)
#pragma unroll
    for (int t = 0; t < CPC; ++t) af[t] = wl[t * 64];
    const unsigned char* __restrict__ const tb8 = reinterpret_cast<const unsigned char*>(P.t);
    const unsigned wu = (unsigned)W;
    // residual operand of stage 3 (pre-ReLU t of the lane's 8 output pixels): requested now, used at the very end
    const unsigned roff = (((unsigned)(y0 + 2 * wave) * wu + (unsigned)(x0 + j)) * C + kk * 4) * 2u, rrow = wu * C * 2u;
    u32x2 resv[2][2][2];                                       // [row pair i][row r][column block]
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
            for (int cb = 0; cb < 2; ++cb) resv[i][r][cb] = *reinterpret_cast<const u32x2*>(tb8 + (roff + (unsigned)(8 * i + r) * rrow + cb * 16 * C * 2));
    {
        constexpr int UPR = W0 * 2, LR = 3, NK = (H0 + LR - 1) / LR;      // 76 units per row, 3 rows per pass, 8 passes (the last one: row 21 only)
        const int lr = tid / UPR, lc = tid - lr * UPR;
        const bool ldr = tid < LR * UPR;
        const unsigned goff = ((unsigned)(y0 - 3 + (ldr ? lr : 0)) * wu + (unsigned)(x0 - 3)) * (C * 2u) + (unsigned)lc * 16u, gstep = LR * wu * (C * 2u);
        // (pass NK - 1 covers rows 21 .. 23 of which only row 21 exists: its rows 22, 23 re-read row 21 -- inside the image -- and store nothing)
        const unsigned glast = ((unsigned)(y0 - 3 + H0 - 1) * wu + (unsigned)(x0 - 3)) * (C * 2u) + (unsigned)lc * 16u;
        u32x4 st[NK];
#pragma unroll
        for (int k = 0; k < NK; ++k) st[k] = *reinterpret_cast<const u32x4*>(tb8 + ((k + 1) * LR <= H0 ? goff + k * gstep : glast));
        unsigned char* const dst = r0 + (lr * UPR + lc) * 16;
        if (ldr) {
#pragma unroll
            for (int k = 0; k < NK; ++k)
                if ((k + 1) * LR <= H0 || lr + k * LR < H0) *reinterpret_cast<u32x4*>(dst + k * LR * UPR * 16) = relu_bf16x8(st[k]);
        }
    }
    // the lane's tap of K chunk t (two taps per chunk: lane groups 0, 1 the first, 2, 3 the second) as a pixel offset in a region
    // that is WIN pixels wide
    auto tap_px = [&](int t, int WIN) {
        int tap = 2 * t + (kk >> 1);
        tap = tap < 9 ? tap : 8;                              // padded slot of the last chunk: zero weights, finite data
        const int ky = tap / 3, kx = tap - ky * 3;
        return ky * WIN + kx;
    };
    auto relu_pk = [](u32x2 p) { return u32x2{relu_bf16x2(p.x), relu_bf16x2(p.y)}; };
    typedef FragPair<CPC> Fr;
    static_assert(H1 == 20 && H2 == 18, "slot lists of the stages below");
    __syncthreads();
    // convR_1's fragments: requested a whole stage before their first use (requested behind a stage's MFMAs, the next stage's first MFMA
    // waited for an L2 round trip)
#pragma unroll
    for (int t = 0; t < CPC; ++t) ag[t] = wl[(CPC + t) * 64];

    // ---- stage 1: r0 (22 x 38) -> r1 (20 x 36).  Six pair slots per wave: rows wave + 4 s (s < 5), both 16-column halves; slot 5 = the
    //      wave's remainder tile (4 rows x 4 columns: rows 4 wave .., columns 32 .. 35) and, for wave 0, remainder tile 4 (rows 16 .. 19) ----
    {
        int bm[CPC], br[CPC];
        const int rr = j >> 2, xc = j & 3;                    // remainder tile: lane j -> row rr of 4, column 32 + xc
#pragma unroll
        for (int t = 0; t < CPC; ++t) {
            bm[t] = (wave * W0 + j + tap_px(t, W0)) * PXB + (kk & 1) * 16;
            br[t] = ((wave * 4 + rr) * W0 + 32 + xc + tap_px(t, W0)) * PXB + (kk & 1) * 16;
        }
        unsigned char* const dm = r1 + (wave * W1 + j) * PXB + kk * 8;
        const int offB5 = (16 - 4 * wave) * W0 * PXB;
        pipe_slots<6, CPC>(af, biasw[0],
            [&](auto sc, Fr& f) {
                constexpr int s = decltype(sc)::value;
#pragma unroll
                for (int t = 0; t < CPC; ++t) {
                    if constexpr (s < 5) {
                        f.a[t] = *reinterpret_cast<const u32x4*>(r0 + bm[t] + s * 4 * W0 * PXB);
                        f.b[t] = *reinterpret_cast<const u32x4*>(r0 + bm[t] + s * 4 * W0 * PXB + 16 * PXB);
                    } else {
                        f.a[t] = *reinterpret_cast<const u32x4*>(r0 + br[t]);
                        f.b[t] = *reinterpret_cast<const u32x4*>(r0 + br[t] + offB5);
                    }
                }
            },
            [&](auto sc, f32x4 va, f32x4 vb) {
                constexpr int s = decltype(sc)::value;
                if constexpr (s < 5) {
                    *reinterpret_cast<u32x2*>(dm + s * 4 * W1 * PXB) = relu_pk(pack_bf16x4(va));
                    *reinterpret_cast<u32x2*>(dm + s * 4 * W1 * PXB + 16 * PXB) = relu_pk(pack_bf16x4(vb));
                } else {
                    *reinterpret_cast<u32x2*>(r1 + ((wave * 4 + rr) * W1 + 32 + xc) * PXB + kk * 8) = relu_pk(pack_bf16x4(va));
                    if (wave == 0) *reinterpret_cast<u32x2*>(r1 + ((16 + rr) * W1 + 32 + xc) * PXB + kk * 8) = relu_pk(pack_bf16x4(vb));
                }
            });
    }
    __syncthreads();
#pragma unroll
    for (int t = 0; t < CPC; ++t) af[t] = wl[(2 * CPC + t) * 64];   // convR_2's, a stage ahead (stage 1 was af's last reader)
    // ---- stage 2: r1 (20 x 36) -> r0 as 18 x 34.  Five pair slots per wave: rows wave + 4 s (s < 4), both halves; slot 4 = row 16 + wave
    //      (waves 0, 1) or two of the three remainder tiles (8 rows x 2 columns; wave 2: tiles 0, 1, wave 3: tile 2 = rows 16, 17) ----
    {
        int bm[CPC], b4a[CPC], b4b[CPC];
        const int rr = j >> 1, xc = j & 1;
        const bool m4 = wave < 2;                             // (wave-uniform)
        const int rtA = 2 * (wave - 2), rowA = min(max(8 * rtA + rr, 0), H2 - 1), rowB = min(max(8 * rtA + 8 + rr, 0), H2 - 1);
#pragma unroll
        for (int t = 0; t < CPC; ++t) {
            bm[t] = (wave * W1 + j + tap_px(t, W1)) * PXB + (kk & 1) * 16;
            b4a[t] = m4 ? bm[t] + 16 * W1 * PXB : (rowA * W1 + 32 + xc + tap_px(t, W1)) * PXB + (kk & 1) * 16;
            b4b[t] = m4 ? bm[t] + 16 * W1 * PXB + 16 * PXB : (rowB * W1 + 32 + xc + tap_px(t, W1)) * PXB + (kk & 1) * 16;
        }
        unsigned char* const dm = r0 + (wave * W2 + j) * PXB + kk * 8;
        unsigned char* const d4a = m4 ? dm + 16 * W2 * PXB : r0 + (rowA * W2 + 32 + xc) * PXB + kk * 8;
        unsigned char* const d4b = m4 ? dm + 16 * W2 * PXB + 16 * PXB : r0 + (rowB * W2 + 32 + xc) * PXB + kk * 8;
        const bool ok4a = m4 || 8 * rtA + rr < H2, ok4b = m4 || wave == 2;
        pipe_slots<5, CPC>(ag, biasw[1],
            [&](auto sc, Fr& f) {
                constexpr int s = decltype(sc)::value;
#pragma unroll
                for (int t = 0; t < CPC; ++t) {
                    if constexpr (s < 4) {
                        f.a[t] = *reinterpret_cast<const u32x4*>(r1 + bm[t] + s * 4 * W1 * PXB);
                        f.b[t] = *reinterpret_cast<const u32x4*>(r1 + bm[t] + s * 4 * W1 * PXB + 16 * PXB);
                    } else {
                        f.a[t] = *reinterpret_cast<const u32x4*>(r1 + b4a[t]);
                        f.b[t] = *reinterpret_cast<const u32x4*>(r1 + b4b[t]);
                    }
                }
            },
            [&](auto sc, f32x4 va, f32x4 vb) {
                constexpr int s = decltype(sc)::value;
                if constexpr (s < 4) {
                    *reinterpret_cast<u32x2*>(dm + s * 4 * W2 * PXB) = relu_pk(pack_bf16x4(va));
                    *reinterpret_cast<u32x2*>(dm + s * 4 * W2 * PXB + 16 * PXB) = relu_pk(pack_bf16x4(vb));
                } else {
                    if (ok4a) *reinterpret_cast<u32x2*>(d4a) = relu_pk(pack_bf16x4(va));
                    if (ok4b) *reinterpret_cast<u32x2*>(d4b) = relu_pk(pack_bf16x4(vb));
                }
            });
    }
    __syncthreads();
    // ---- stage 3: r0 (18 x 34) -> HBM; a wave takes row pairs 2 (wave + 4 i), both column blocks (2x2 pool in registers): four pair slots ----
    {
        int bm[CPC];
#pragma unroll
        for (int t = 0; t < CPC; ++t) bm[t] = (2 * wave * W2 + j + tap_px(t, W2)) * PXB + (kk & 1) * 16;
        // (uniform base pointers + 32-bit byte offsets: the row steps are scalar adds)
        unsigned char* __restrict__ const outb = reinterpret_cast<unsigned char*>(P.out);
        unsigned char* __restrict__ const poolb = reinterpret_cast<unsigned char*>(P.pool);
        const int Wp = (W + 1) >> 1;
        const unsigned poff = (((unsigned)((y0 >> 1) + wave) * (unsigned)Wp + (unsigned)((x0 + j) >> 1)) * C + kk * 4) * 2u, prow = (unsigned)Wp * C * 2u;
        pipe_slots<4, CPC>(af, biasw[2],
            [&](auto sc, Fr& f) {
                constexpr int s = decltype(sc)::value, i = s >> 1, cb = s & 1;
#pragma unroll
                for (int t = 0; t < CPC; ++t) {
                    f.a[t] = *reinterpret_cast<const u32x4*>(r0 + bm[t] + (8 * i) * W2 * PXB + cb * 16 * PXB);
                    f.b[t] = *reinterpret_cast<const u32x4*>(r0 + bm[t] + (8 * i + 1) * W2 * PXB + cb * 16 * PXB);
                }
            },
            [&](auto sc, f32x4 va, f32x4 vb) {
                constexpr int s = decltype(sc)::value, i = s >> 1, cb = s & 1;
                const f32x4 v2[2] = {va, vb};
                u32x2 pk[2];
#pragma unroll
                for (int r = 0; r < 2; ++r) {
                    pk[r] = relu_pk(pack_bf16x4(v2[r] + unpack_bf16x4(resv[i][r][cb])));
                    *reinterpret_cast<u32x2*>(outb + (roff + (unsigned)(8 * i + r) * rrow + cb * 16 * C * 2)) = pk[r];    // (the residual operand's offsets)
                }
                if (P.pool) {
                    // 2 x 2 max on the packed values (non-negative bf16 order like their bit patterns): rows, then the neighbour lane
                    const unsigned m0 = pkmax_u16(pk[0].x, pk[1].x), m1 = pkmax_u16(pk[0].y, pk[1].y);
                    const unsigned n0 = __float_as_uint(lane_xor1(__uint_as_float(m0))), n1 = __float_as_uint(lane_xor1(__uint_as_float(m1)));
                    if ((j & 1) == 0)
                        *reinterpret_cast<u32x2*>(poolb + (poff + (unsigned)(4 * i) * prow + cb * 8 * C * 2)) = u32x2{pkmax_u16(m0, n0), pkmax_u16(m1, n1)};
                }
            });
    }
}

// ------------------------------------------------------------------------------------------------
// deconvb_kernel: conv2d_transpose 3x3, stride 2, SAME (layers.py:362), bias + ReLU, bf16 in / out.
//   out[i] = sum over (o, k) with 2 o + k - pb = i of in[o] W[k]   (per dimension; pb = pad_before of the SAME rule).
// With I = i + pb: I even <- k = 0 (o = I/2) and k = 2 (o = I/2 - 1); I odd <- k = 1 (o = (I-1)/2).  So the 2 x 2 outputs
// (parity classes) of input-grid position (Y, X) take 4 / 2 / 2 / 1 taps from the positions (Y - dy, X - dx), dy, dx in {0,1}:
// an n-tile = 16 consecutive X of one Y, 4 accumulators (classes) per m-tile, the 4 shifted B fragments shared by the classes.
//   MODE 1 (Cin == 16): K chunk = (dy, both dx) x 16 channels.   MODE 2 (Cin % 32 == 0): K chunk = (shift, 32-channel group).
// Block = TH x 16 input positions = 2 TH x 32 output pixels x 16 MT channels, which leave through an LDS tile as whole rows.
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
    int act;               // graph variants: 1 = elu, 2 = leaky on the fp32 sums (relu_out is 0 then)
    XcdMap xm;             // XCD-aware block -> tile map (sched_tile)
};
constexpr int DCB_TW = 16;                                    // input columns per block; rows: template parameter TH (8 or 16)

template <int MODE, int MT, int DCB_TH>
__global__ __launch_bounds__(256, 2) void deconvb_kernel(const DeconvBArgs a) {
    constexpr int LH = DCB_TH + 1, LW = DCB_TW + 1, RW = DCB_TH / 4;              // RW input rows per wave
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
    const int bid = sched_tile(a.xm);
    if (bid < 0) return;
    int pi = 0;
    pi = prob_of_tile(a, bid);
    const DeconvBProb& P = a.p[pi];
    const int tile = bid - P.tile_begin;
    const int ty = tile / P.tiles_x, tx = tile - ty * P.tiles_x;
    const int X0 = tx * DCB_TW, Y0 = ty * DCB_TH, mt0 = blockIdx.y * MT;
    const int Hi = P.Hi, Wi = P.Wi, cin = a.cin;

    // n-tile = one input row of the tile (16 positions); wave w owns rows w, w + 4, ...; classes c = 2 py + px
    f32x4 acc[RW][4][MT];
#pragma unroll
    for (int r = 0; r < RW; ++r)
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int m = 0; m < MT; ++m) acc[r][c][m] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int ngroups = MODE == 2 ? a.groups : 1;
    const u32x4* __restrict__ wbase = a.wpk + (size_t)mt0 * 64 + lane;
    const size_t wstride = (size_t)a.mtiles * 64;
    // Requests first, all of them: bias, (MODE 1) the six A fragments, then the input tile.  As first written the block was a chain of
    // five dependent round trips -- each input load sat in its own "inside the image?" branch with its own wait (the zero-padding
    // select had been merged into the load), the fragments were fetched between the MFMAs and the bias in the epilogue -- for twelve
    // MFMAs per wave.
    f32x4 bias4[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) {
        const int c = (mt0 + m) * 16 + kk * 4;
        bias4[m] = *reinterpret_cast<const f32x4*>(a.bias + (c < a.cout ? c : 0));
        if (c >= a.cout) bias4[m] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    u32x4 wf[MODE == 1 ? 6 : 1][MT];
    if constexpr (MODE == 1) {
#pragma unroll
        for (int f = 0; f < 6; ++f)
#pragma unroll
            for (int m = 0; m < MT; ++m) wf[f][m] = wbase[(size_t)f * wstride + (size_t)m * 64];
    }
    // the thread's input slots: clamped pixel index (always a valid address) + inside-the-image mask, computed once
    int spix[NLOAD];
    unsigned mask = 0;
#pragma unroll
    for (int i = 0; i < NLOAD; ++i) {
        const int u = min(tid + i * 256, NU - 1);
        const int pix = u / SUBS;
        const int ly = pix / LW, lx = pix - ly * LW;
        const int gy = Y0 - 1 + ly, gx = X0 - 1 + lx;
        spix[i] = min(max(gy, 0), Hi - 1) * Wi + min(max(gx, 0), Wi - 1);
        mask |= ((gy >= 0 && gy < Hi && gx >= 0 && gx < Wi) ? 1u : 0u) << i;
    }
    const int sub8 = (tid % SUBS) * 8;                          // (256 is a multiple of SUBS: the sub-block is the same for all slots)
    for (int g = 0; g < ngroups; ++g) {
        // input tile with one halo row / column at the top / left: LDS (ly, lx) = input (Y0 - 1 + ly, X0 - 1 + lx)
        u32x4 st[NLOAD];
#pragma unroll
        for (int i = 0; i < NLOAD; ++i) st[i] = *reinterpret_cast<const u32x4*>(P.in + (size_t)spix[i] * cin + g * 32 + sub8);
        // MODE 2, one m-tile: the group's nine fragments, requested with the tile (32 -> 16 level: 64 -> 51 us; with two m-tiles the 72
        // registers cost half the occupancy and the layers did not gain: those fetch their fragments between the MFMAs)
        constexpr bool PRE = MODE == 2 && MT == 1;
        u32x4 wt[PRE ? 9 : 1][MT];
        const u32x4* __restrict__ wg = wbase + (size_t)g * 9 * wstride;
        if constexpr (PRE) {
#pragma unroll
            for (int t = 0; t < 9; ++t)
#pragma unroll
                for (int m = 0; m < MT; ++m) wt[t][m] = wg[(size_t)t * wstride + (size_t)m * 64];
        }
#pragma unroll
        for (int i = 0; i < NLOAD; ++i) asm volatile("" : "+v"(st[i]));      // every request is out before the first value is touched
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
        for (int r = 0; r < RW; ++r) {
            const int Yl = wave + 4 * r;                       // local input row of this n-tile
            if constexpr (MODE == 2) {
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
                                    acc[r][2 * py + px][m] = mfma_bf16_k32(PRE ? wt[PRE ? ky * 3 + kx : 0][m] : wg[(size_t)(ky * 3 + kx) * wstride + (size_t)m * 64], b,
                                                                          acc[r][2 * py + px][m]);
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
                        for (int m = 0; m < MT; ++m) acc[r][cls][m] = mfma_bf16_k32(wf[f][m], b, acc[r][cls][m]);
                    }
                }
            }
        }
    }
    // ---- accumulators -> bias, ReLU, bf16 -> output tile [16][32][OC] in LDS ----
#pragma unroll
    for (int m = 0; m < MT; ++m) {
        const f32x4 b4 = bias4[m];
#pragma unroll
        for (int r = 0; r < RW; ++r) {
            const int Yl = wave + 4 * r;
#pragma unroll
            for (int cls = 0; cls < 4; ++cls) {
                f32x4 v = acc[r][cls][m] + b4;
                if (a.relu_out) v = relu4(v);
                else if (a.act) v = act4b(v, a.act);
                // 16-byte unit S = pixel * UPP + channel quad pair, stored at S ^ ((pixel >> 1) & 7): the 16 lanes of a store (pixels
                // 2 j + px, fixed channels) would otherwise be 64 / 128 bytes apart = 8- / 16-way bank conflicts (74 % of the LDS cycles)
                const int orow = 2 * Yl + (cls >> 1), ocol = 2 * j + (cls & 1), pix = orow * 2 * DCB_TW + ocol;
                const int S = (pix * (OC / 8) + m * 2 + (kk >> 1)) ^ ((pix >> 1) & 7);
                *reinterpret_cast<u32x2*>(otile + S * 16 + (kk & 1) * 8) = pack_bf16x4(v);
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
            *reinterpret_cast<u32x4*>(P.out + ((size_t)y * P.Wo + x) * a.cout + mt0 * 16 + sub * 8) =
                *reinterpret_cast<const u32x4*>(otile + (u ^ ((pix >> 1) & 7)) * 16);
    }
}

// ------------------------------------------------------------------------------------------------
// deconvb8_kernel: the level-0 deconvolution (16 -> 8 channels, 3x3, stride 2, SAME, bias + ReLU) without LDS and without barriers (round 5).
// deconvb_kernel<1, 1, 8> ran it with 16-row m-tiles of which 8 rows are padding (half of every epilogue instruction and of the output tile's
// LDS traffic wasted), staged the input through LDS for twelve MFMAs per wave and the output through a second tile to write whole rows: 106 us
// per page for 324 MB (65 us of HBM time).  Here an m-tile = the TWO parity classes px = 0 / 1 of a row class py x 8 channels (rows 0 .. 7:
// px = 0, rows 8 .. 15: px = 1), so three MFMAs serve an input row (py = 0 <- rows Y and Y - 1, py = 1 <- row Y) and a lane of the D tile holds
// 4 channels of output pixel 2 j + (kk >> 1); v_permlane16_swap between the py = 0 and py = 1 results makes whole pixels (16 bytes per lane: the
// 32 lanes of a row write 512 contiguous bytes), which go straight to HBM.  The B fragment of input row Y (16 positions, lane group kk: position
// j - (kk >> 1), channel half kk & 1) is ONE 16-byte global load per lane and serves rows Y and Y + 1: a wave walks D8_RW consecutive rows of a
// 16-column block with D8_RW + 1 loads, all requested up front.  ReLU graphs (elu / leaky keep deconvb_kernel).
// ------------------------------------------------------------------------------------------------
constexpr int D8_RW = 8, D8_TW = 64;                          // input rows per wave / input columns per block (four waves side by side)
__global__ __launch_bounds__(256, 4) void deconvb8_kernel(const DeconvBArgs a) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int j = lane & 15, kk = lane >> 4;
    const int bid = sched_tile(a.xm);
    if (bid < 0) return;
    const int pi = prob_of_tile(a, bid);
    const DeconvBProb& P = a.p[pi];
    const int tile = bid - P.tile_begin;
    const int ty = tile / P.tiles_x, tx = tile - ty * P.tiles_x;
    const int X0 = tx * D8_TW + wave * 16, Y0 = ty * D8_RW;
    const int Hi = P.Hi, Wi = P.Wi;
    if (X0 >= Wi) return;
    const u32x4 wq0 = a.wpk[lane], wq1 = a.wpk[64 + lane], wq2 = a.wpk[128 + lane];
    const f32x4 b4 = *reinterpret_cast<const f32x4*>(a.bias + (kk & 1) * 4);
    // the wave's D8_RW + 1 fragments: input rows Y0 - 1 .. Y0 + D8_RW - 1, position X0 + j - (kk >> 1), 8 of its 16 channels
    const int xi = X0 + j - (kk >> 1);
    const bool interior = Y0 >= 1 && Y0 + D8_RW <= Hi && X0 >= 1 && X0 + 16 <= Wi;        // (wave-uniform)
    u32x4 fr[D8_RW + 1];
    if (interior) {
        const unsigned char* __restrict__ src = reinterpret_cast<const unsigned char*>(P.in) + (((size_t)(Y0 - 1) * Wi + xi) * 16 + (kk & 1) * 8) * 2;
        const size_t rs = (size_t)Wi * 32;
#pragma unroll
        for (int r = 0; r <= D8_RW; ++r) fr[r] = *reinterpret_cast<const u32x4*>(src + r * rs);
    } else {
        const bool xok = xi >= 0 && xi < Wi;
        const int xc = min(max(xi, 0), Wi - 1);
#pragma unroll
        for (int r = 0; r <= D8_RW; ++r) {
            const int y = Y0 - 1 + r;
            const u32x4 v = *reinterpret_cast<const u32x4*>(P.in + ((size_t)min(max(y, 0), Hi - 1) * Wi + xc) * 16 + (kk & 1) * 8);
            fr[r] = (xok && y >= 0 && y < Hi) ? v : u32x4{0u, 0u, 0u, 0u};
        }
    }
    // the lane's output pixel of input row Y: (2 Y - pbh + (kk & 1), 2 (X0 + j) - pbw + (kk >> 1)) -- row class py = kk & 1 after the lane trade
    const int ox = 2 * (X0 + j) - P.pbw + (kk >> 1);
    const bool oxok = ox >= 0 && ox < P.Wo && X0 + j < Wi;
    bf16_t* __restrict__ const ob = P.out;
#pragma unroll
    for (int r = 0; r < D8_RW; ++r) {
        f32x4 acc0 = mfma_bf16_k32(wq0, fr[r + 1], b4);       // py = 0: filter row 0 from input row Y ...
        f32x4 acc1 = mfma_bf16_k32(wq1, fr[r + 1], b4);       // py = 1: filter row 1 from input row Y
        acc0 = mfma_bf16_k32(wq2, fr[r], acc0);               // ... and filter row 2 from input row Y - 1
        const u32x2 p0 = pack_bf16x4(acc0), p1 = pack_bf16x4(acc1);
        const auto s0 = __builtin_amdgcn_permlane16_swap(p0.x, p1.x, false, false);
        const auto s1 = __builtin_amdgcn_permlane16_swap(p0.y, p1.y, false, false);
        const u32x4 rec = relu_bf16x8(u32x4{s0[0], s1[0], s0[1], s1[1]});
        const int oy = 2 * (Y0 + r) - P.pbh + (kk & 1);
        if (oxok && oy >= 0 && oy < P.Ho && Y0 + r < Hi) *reinterpret_cast<u32x4*>(ob + ((size_t)oy * P.Wo + ox) * 8) = rec;
    }
}

// ------------------------------------------------------------------------------------------------
// res8b_kernel<UP>: a WHOLE level-0 residual block (8 channels) of the bf16 path in one kernel.
//   UP = false (unet_down_0, ARU_v1.py:208-245): t = conv3x3(image, 1 -> 8); r = relu(t); r = relu(convR_0 r); r = relu(convR_1 r);
//                d0 = relu(convR_2 r + t); also maxpool2(d0).         HBM: fp32 image in, bf16 d0 + pool out.
//   UP = true  (unet_up_0, ARU_v1.py:251-292): t = conv3x3([skip, deconv], 16 -> 8), then the same tail.   HBM: 2 x 8 ch in, 8 out.
// (layer by layer the block moves 5 / 6 tensors of 4500 x 3000 x 8 through HBM; the level holds 60 % of the net's activation bytes)
// Block = 16 x 32 output pixels.  All 8 -> 8 products use the PIXEL-PAIR mapping: M = 2 adjacent pixels x 8 output channels,
// N = 16 pairs (32 consecutive pixels of the flattened stage region), K = one filter row = 4 window pixels x 8 channels, so a conv
// is 3 MFMAs per 32 pixels with 75 % of the multipliers useful, every lane of the D tile holds 4 channels of one pixel, and a
// lane's B fragment is still ONE 16-byte LDS read (window pixel 2n + kk).  conv1 of the up block is the same with K = 4 pixels x
// 16 channels = two MFMAs per filter row; conv1 of the down block (one input channel) runs on the vector ALU.
// Regions (flattened, row-major): t 22 x 38 -> stage 1 20 x 36 -> stage 2 18 x 34 -> out 16 x 32; positions outside the image hold
// zeros (SAME padding of every conv); the pre-ReLU t of the centre goes to its own LDS tile for the residual add.
// ------------------------------------------------------------------------------------------------
struct Res8BProb {
    const float* img;      // DOWN: [H,W] fp32 image (pyramid level)
    const float* stats;    // DOWN: {mean, 1/std} or nullptr
    const bf16_t* skip;    // UP: [H,W,8]
    const bf16_t* dec;     // UP: [H,W,8] deconv output
    bf16_t* out;           // [H,W,8]
    bf16_t* pool;          // DOWN: maxpool2(out) or nullptr
    int H, W;
    int tiles_x, tile_begin;
};
struct Res8BArgs {
    Res8BProb p[MAXP];
    int nprob;
    const float* w1;       // DOWN: conv1 [9][8] fp32 values rounded to bfloat16 (border tiles; the interior tiles' fragment is w1pk)
    const float* b1;       // conv1 bias [8]
    const u32x4* w1pk;     // UP: conv1 pair fragments [ky 3][half 2][64 lanes] x 16 bytes; res8f_kernel DOWN: [64 lanes] (bf16 conv1)
    const u32x4* w1pf;     // res8f_kernel UP (interior tiles, planar input tile): [ky 3][source 2][64 lanes] x 16 bytes, k = 8 (window pixel) + channel of the source
    const u32x4* wpk;      // tail: [3 convs][ky 3][64 lanes] x 16 bytes
    const float* bias;     // tail biases [3][8]
    XcdMap xm;             // XCD-aware block -> tile map (sched_tile)
};

template <bool UP>
struct Res8BLayout {
    static constexpr int TH = 16, TW = 32, SLACK = 4;
    static constexpr int R0B = ((TH + 6) * (TW + 6) + SLACK) * 16, R1B = ((TH + 4) * (TW + 4) + SLACK) * 16, TCB = TH * TW * 16;
    static constexpr int INB = UP ? ((TH + 8) * (TW + 8) + SLACK) * 32 : (TH + 8) * (TW + 8) * 4;
    // stage 1's result may take the place of conv1's input tile (dead by then)
    static constexpr int R0_OFF = (INB > R1B ? INB : R1B), TC_OFF = R0_OFF + R0B, BYTES = TC_OFF + TCB;
};

// the block of one tile (any position; `lds` holds Res8BLayout<UP>::BYTES)
// (ymax / xmax: stores are clipped to rows < ymax, columns < xmax -- the border tiles of res8wb_kernel, whose neighbours belong to the strip walker)
template <bool UP, int ACT = 0>
__device__ __forceinline__ void res8b_tile(const Res8BArgs& a, const Res8BProb& P, int x0, int y0, unsigned char* lds, int ymax = 1 << 30, int xmax = 1 << 30) {
    constexpr int TH = 16, TW = 32;
    constexpr int H0 = TH + 6, W0 = TW + 6, H1 = TH + 4, W1 = TW + 4, H2 = TH + 2, W2 = TW + 2;
    constexpr int IH = TH + 8, IW = TW + 8;                   // conv1's input tile (halo 4)
    constexpr int SLACK = Res8BLayout<UP>::SLACK;             // pixels a zero-weight window column / a clamped tail lane may read past a region
    constexpr int R1_OFF = 0, IN_OFF = 0, R0_OFF = Res8BLayout<UP>::R0_OFF, TC_OFF = Res8BLayout<UP>::TC_OFF;
    unsigned char* const r0 = lds + R0_OFF;
    unsigned char* const r1 = lds + R1_OFF;
    unsigned char* const tc = lds + TC_OFF;
    unsigned char* const in = lds + IN_OFF;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int j = lane & 15, kk = lane >> 4, e = kk >> 1, ch = (kk & 1) * 4;     // D layout: pixel parity e, channels ch .. ch + 3
    const int H = P.H, W = P.W;
    // the whole conv1 input window lies inside the image: no position of any stage needs the zero test
    const bool interior = y0 - 4 >= 0 && y0 + TH + 4 <= H && x0 - 4 >= 0 && x0 + TW + 4 <= W;

    // ---- conv1 input tile -> LDS ----
    if constexpr (UP) {
        constexpr int NU = IH * IW * 2, NLOAD = (NU + 255) / 256;
        u32x4 st[NLOAD];
        unsigned mask = 0;

#pragma unroll
        for (int i = 0; i < NLOAD; ++i) {
            const int u = min(tid + i * 256, NU - 1);
            const int pix = u >> 1, sub = u & 1;
            const int ly = pix / IW, lx = pix - ly * IW;
            const int gy = y0 - 4 + ly, gx = x0 - 4 + lx;
            const bf16_t* __restrict__ src = concat_src(P.skip, P.dec, sub, 1);                 // sub 0: skip, sub 1: dec
            st[i] = *reinterpret_cast<const u32x4*>(src + ((size_t)min(max(gy, 0), H - 1) * W + min(max(gx, 0), W - 1)) * 8);
            mask |= ((gy >= 0 && gy < H && gx >= 0 && gx < W) ? 1u : 0u) << i;
        }
#pragma unroll
        for (int i = 0; i < NLOAD; ++i) {
            const int u = tid + i * 256;
            if (u < NU) *reinterpret_cast<u32x4*>(in + u * 16) = ((mask >> i) & 1u) ? st[i] : u32x4{0u, 0u, 0u, 0u};
        }
        if (tid < 2 * SLACK) *reinterpret_cast<u32x4*>(in + (NU + tid) * 16) = u32x4{0u, 0u, 0u, 0u};
    } else {
        constexpr int NLOAD = (IH * IW + 255) / 256;
        float st[NLOAD];
        float mean = 0.f, inv = 1.f;
        if (P.stats) { mean = P.stats[0]; inv = P.stats[1]; }
#pragma unroll
        for (int i = 0; i < NLOAD; ++i) {
            const int u = min(tid + i * 256, IH * IW - 1);
            const int ly = u / IW, lx = u - ly * IW;
            st[i] = P.img[(size_t)min(max(y0 - 4 + ly, 0), H - 1) * W + min(max(x0 - 4 + lx, 0), W - 1)];
        }
#pragma unroll
        for (int i = 0; i < NLOAD; ++i) {
            const int u = tid + i * 256;
            if (u < IH * IW) {
                const int ly = u / IW, lx = u - ly * IW;
                const int gy = y0 - 4 + ly, gx = x0 - 4 + lx;
                // the (standardised) image as bfloat16, like the interior tiles' MFMA form reads it (res8f_kernel): the first layer is
                // ONE function of the page, not one per tile kind -- no seam between border and interior tiles (a.w1 holds the filter
                // rounded to bfloat16 as well; products of two bfloat16 are exact in fp32, only the order of the sum differs)
                reinterpret_cast<float*>(in)[u] = (gy >= 0 && gy < H && gx >= 0 && gx < W) ? round_bf16((st[i] - mean) * inv) : 0.f;
            }
        }
    }
    if (tid < SLACK) *reinterpret_cast<u32x4*>(r0 + (H0 * W0 + tid) * 16) = u32x4{0u, 0u, 0u, 0u};
    const u32x4* __restrict__ wl = a.wpk + lane;
    u32x4 af[3];
#pragma unroll
    for (int t = 0; t < 3; ++t) af[t] = wl[t * 64];           // convR_0's fragments fly during conv1
    __syncthreads();

    // ---- conv1 -> relu(t) over the 22 x 38 region (r0), raw t of the centre (tc) ----
    if constexpr (UP) {
        u32x4 a1[6];
#pragma unroll
        for (int t = 0; t < 6; ++t) a1[t] = a.w1pk[t * 64 + lane];
        const f32x4 b4 = *reinterpret_cast<const f32x4*>(a.b1 + ch);
        constexpr int NPIX = H0 * W0;
        // the lane's pixel q = 32 tl + 2 j + e walks the flattened region in steps of 128: (py, px) are updated incrementally
        // (no division in the loop); the pair of its B operand starts at q - e in the same row (the region widths are even)
        auto conv1_loop = [&](auto interior_c) {
            constexpr bool INT = decltype(interior_c)::value;
            int q = wave * 32 + 2 * j + e;
            int py = q / W0, px = q - py * W0;
            for (; q - 2 * j - e < NPIX; q += 128) {
                const int cpy = q < NPIX ? py : H0 - 1, cpx = q < NPIX ? px : W0 - 2 + e;   // tail tile: clamp to the last pair
                const int base = (cpy * IW + cpx - e) * 32 + (kk >> 1) * 32 + (kk & 1) * 16;
                f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                    for (int hf = 0; hf < 2; ++hf)
                        acc = mfma_bf16_k32(a1[ky * 2 + hf], *reinterpret_cast<const u32x4*>(in + base + (ky * IW + 2 * hf) * 32), acc);
                f32x4 v = acc + b4;
                if constexpr (!INT) {
                    const int gy = y0 - 3 + py, gx = x0 - 3 + px;
                    if (!(gy >= 0 && gy < H && gx >= 0 && gx < W)) v = f32x4{0.f, 0.f, 0.f, 0.f};
                }
                if (q < NPIX) {
                    *reinterpret_cast<u32x2*>(r0 + q * 16 + ch * 2) = pack_bf16x4(relu4(v));
                    const int cy = py - 3, cx = px - 3;
                    if ((unsigned)cy < (unsigned)TH && (unsigned)cx < (unsigned)TW) *reinterpret_cast<u32x2*>(tc + (cy * TW + cx) * 16 + ch * 2) = pack_bf16x4(v);
                }
                px += 128 % W0; py += 128 / W0;
                if (px >= W0) { px -= W0; ++py; }
            }
        };
        if (interior) conv1_loop(std::true_type{}); else conv1_loop(std::false_type{});
    } else {
        typedef const float __attribute__((address_space(4)))* cptr;
        cptr w1 = (cptr)a.w1;
        cptr b1 = (cptr)a.b1;
        const float* img = reinterpret_cast<const float*>(in);
        for (int p = tid; p < H0 * W0; p += 256) {
            const int ly = p / W0, lx = p - ly * W0;
            float v[8];
#pragma unroll
            for (int c = 0; c < 8; ++c) v[c] = b1[c];
#pragma unroll
            for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) {
                    const float x = img[(ly + ky) * IW + lx + kx];
#pragma unroll
                    for (int c = 0; c < 8; ++c) v[c] = fmaf(x, w1[(ky * 3 + kx) * 8 + c], v[c]);
                }
            const int gy = y0 - 3 + ly, gx = x0 - 3 + lx;
            const bool inside = interior || (gy >= 0 && gy < H && gx >= 0 && gx < W);
            if (!inside) {
#pragma unroll
                for (int c = 0; c < 8; ++c) v[c] = 0.f;
            }
            const u32x4 raw = u32x4{pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]), pack_bf16x2(v[4], v[5]), pack_bf16x2(v[6], v[7])};
            *reinterpret_cast<u32x4*>(r0 + p * 16) = relu_bf16x8(raw);
            const int cy = ly - 3, cx = lx - 3;
            if (cy >= 0 && cy < TH && cx >= 0 && cx < TW) *reinterpret_cast<u32x4*>(tc + (cy * TW + cx) * 16) = raw;
        }
    }
    __syncthreads();
    if (tid < SLACK) *reinterpret_cast<u32x4*>(r1 + (H1 * W1 + tid) * 16) = u32x4{0u, 0u, 0u, 0u};   // (r1 may overlay conv1's input)

    // ---- stages 1 and 2: LDS -> LDS, n-tile = 32 consecutive pixels (16 pairs) of the flattened output region ----
    auto mid_stage = [&](auto interior_c, const unsigned char* src, auto win_c, unsigned char* dst, auto ho_c, int halo, const float* bias) {
        constexpr bool INT = decltype(interior_c)::value;
        constexpr int WIN = decltype(win_c)::value, HO = decltype(ho_c)::value, WO = WIN - 2, npix = HO * WO;
        const f32x4 b4 = *reinterpret_cast<const f32x4*>(bias + ch);
        int q = wave * 32 + 2 * j + e;
        int py = q / WO, px = q - py * WO;
        for (; q - 2 * j - e < npix; q += 128) {
            const int cpy = q < npix ? py : HO - 1, cpx = q < npix ? px : WO - 2 + e;      // tail tile: clamp to the last pair
            const int base = (cpy * WIN + cpx - e + kk) * 16;                  // window pixel kk of the pair's 4-pixel window, filter row 0
            f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ky = 0; ky < 3; ++ky) acc = mfma_bf16_k32(af[ky], *reinterpret_cast<const u32x4*>(src + base + ky * WIN * 16), acc);
            f32x4 v = ACT ? act4b(acc + b4, ACT) : relu4(acc + b4);
            if constexpr (!INT) {
                const int gy = y0 - halo + py, gx = x0 - halo + px;
                if (!(gy >= 0 && gy < H && gx >= 0 && gx < W)) v = f32x4{0.f, 0.f, 0.f, 0.f};
            }
            if (q < npix) *reinterpret_cast<u32x2*>(dst + q * 16 + ch * 2) = pack_bf16x4(v);
            px += 128 % WO; py += 128 / WO;
            if (px >= WO) { px -= WO; ++py; }
        }
    };
    auto mid = [&](const unsigned char* src, auto win_c, unsigned char* dst, auto ho_c, int halo, const float* bias) {
        if (interior) mid_stage(std::true_type{}, src, win_c, dst, ho_c, halo, bias);
        else mid_stage(std::false_type{}, src, win_c, dst, ho_c, halo, bias);
    };
    mid(r0, std::integral_constant<int, W0>{}, r1, std::integral_constant<int, H1>{}, 2, a.bias);
#pragma unroll
    for (int t = 0; t < 3; ++t) af[t] = wl[(3 + t) * 64];
    __syncthreads();
    mid(r1, std::integral_constant<int, W1>{}, r0, std::integral_constant<int, H2>{}, 1, a.bias + 8);   // the t region is dead: its space takes stage 2's result
#pragma unroll
    for (int t = 0; t < 3; ++t) af[t] = wl[(6 + t) * 64];
    __syncthreads();

    // ---- stage 3: one n-tile = one row of the output tile; a wave takes row pairs (2x2 pool in registers) ----
    {
        const f32x4 b4 = *reinterpret_cast<const f32x4*>(a.bias + 16 + ch);
        const int Wp = (W + 1) >> 1;
        const int x = x0 + 2 * j + e;
        const int Hs = min(H, ymax), Ws = min(W, xmax);      // (even clip bounds: a 2 x 2 pool window lies on one side)
        for (int rp = wave; rp < TH / 2; rp += 4) {
            f32x4 v2[2];
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                const int oy = 2 * rp + r;
                const int base = (oy * W2 + 2 * j + kk) * 16;
                f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int ky = 0; ky < 3; ++ky) acc = mfma_bf16_k32(af[ky], *reinterpret_cast<const u32x4*>(r0 + base + ky * W2 * 16), acc);
                const f32x4 tres = unpack_bf16x4(*reinterpret_cast<const u32x2*>(tc + (oy * TW + 2 * j + e) * 16 + ch * 2));
                const f32x4 v = ACT ? act4b(acc + b4 + tres, ACT) : relu4(acc + b4 + tres);
                const u32x2 pk = pack_bf16x4(v);
                v2[r] = unpack_bf16x4(pk);
                const int y = y0 + oy;
                if (y < Hs && x < Ws) *reinterpret_cast<u32x2*>(P.out + ((size_t)y * W + x) * 8 + ch) = pk;
            }
            if (P.pool) {
                // window = rows y, y + 1 (registers) x pixels 2j, 2j + 1 (this lane and lane ^ 32); out-of-image members are excluded
                const int y = y0 + 2 * rp;
                f32x4 mm = (y + 1 < H) ? max4(v2[0], v2[1]) : v2[0];
                const bool xin = x < W;
                f32x4 lo, hi;
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const float mine = xin ? mm[c] : -INFINITY;
                    lo[c] = from_lower_half(mine);
                    hi[c] = from_upper_half(mine);
                }
                mm = max4(lo, hi);
                if (e == 0 && y < Hs && x < Ws) *reinterpret_cast<u32x2*>(P.pool + ((size_t)(y >> 1) * Wp + (x >> 1)) * 8 + ch) = pack_bf16x4(mm);
            }
        }
    }
}

template <bool UP, int ACT = 0>
__global__ __launch_bounds__(256, UP ? 3 : 4) void res8b_kernel(const Res8BArgs a) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[Res8BLayout<UP>::BYTES];
    const int bid = sched_tile(a.xm);
    if (bid < 0) return;
    int pi = 0;
    pi = prob_of_tile(a, bid);
    const Res8BProb& P = a.p[pi];
    const int tile = bid - P.tile_begin;
    const int ty = tile / P.tiles_x, tx = tile - ty * P.tiles_x;
    res8b_tile<UP, ACT>(a, P, tx * 32, ty * 16, lds);
}

// ------------------------------------------------------------------------------------------------
// res8f_kernel<UP>: the same block for INTERIOR tiles (the whole 24 x 40 input window inside the image: ~96 % of the tiles of
// a 3000 x 4500 page), written for instruction count.  res8b_kernel issued ~5300 instructions per tile for 178 MFMAs (SQ
// counters: 20 vector instructions per MFMA, the vector ALU issuing 80 % of the SIMD cycles with four waves per SIMD): its
// flattened n-tiles need a division, clamps and an inside-the-image test per tile, conv1 of the down block ran as 72 FMAs per
// pixel.  Here an n-tile is a ROW of a stage region (32 pixels = 16 pairs; the 2 .. 6 remaining pixels of several rows are
// gathered into "remainder" tiles), so every LDS address is a per-lane constant plus a compile-time offset, no position is
// outside the image, the bias is the accumulators' initial value, ReLU is one v_pk_max_i16 per two values AFTER the rounding
// to bf16, and conv1 of the down block is ONE MFMA per 32 pixels on the image tile held as bf16 (K = 3 x 4 window values).
// Border tiles take res8b_tile (the general form) inside the same launch.
// ------------------------------------------------------------------------------------------------
// Debug builds (-DR8F_TRACE -DR8F_TRACE_TID=<thread>): s_memtime stamps of one thread of every 8th block at the phase boundaries,
// read back through asep_debug_r8f_trace (aru_engine.hip) by scripts/gpu_r8f_trace.py.  This is how the kernel's time was split
// into "window from HBM / conv1 / stages / stores" (DESIGN lesson 20); compiled out otherwise.
#if defined(R8F_TRACE)
__device__ unsigned long long g_r8f_trace[2][4096 * 12];
#define R8F_MARK(i) do { if ((blockIdx.x & 7) == 0 && (blockIdx.x >> 3) < 4096 && tid == R8F_TRACE_TID) g_r8f_trace[UP ? 1 : 0][(blockIdx.x >> 3) * 12 + (i)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define R8F_MARK(i) do { } while (0)
#endif
template <bool UP>
__global__ __launch_bounds__(256, UP ? 3 : 4) void res8f_kernel(const Res8BArgs a) {
    constexpr int TH = 16, TW = 32;
    constexpr int H0 = TH + 6, W0 = TW + 6, H1 = TH + 4, W1 = TW + 4, W2 = TW + 2;
    constexpr int IH = TH + 8, IW = TW + 8;
    constexpr int INB = UP ? IH * IW * 32 : 2048;             // DOWN: bf16 image tile, (IH + 1) x IW x 2 bytes
    constexpr int INPL = IH * IW * 16;                        // UP: bytes of one plane of the input tile (plane 0: skip, plane 1: deconv)
    constexpr int R1B = H1 * W1 * 16, R0B = H0 * W0 * 16, TCB = TH * TW * 16;
    constexpr int R1_OFF = UP ? 0 : INB, R0_OFF = UP ? INB : INB + R1B, TC_OFF = R0_OFF + R0B;
    constexpr int LDSB = TC_OFF + TCB > Res8BLayout<UP>::BYTES ? TC_OFF + TCB : Res8BLayout<UP>::BYTES;
    __shared__ __attribute__((aligned(16))) unsigned char lds[LDSB];
    unsigned char* const in = lds;
    unsigned char* const r1 = lds + R1_OFF;
    unsigned char* const r0 = lds + R0_OFF;
    unsigned char* const tc = lds + TC_OFF;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    R8F_MARK(0);
    const int j = lane & 15, kk = lane >> 4, e = kk >> 1, ch = (kk & 1) * 4;     // D layout: pixel parity e, channels ch .. ch + 3
    const int bid = sched_tile(a.xm);
    if (bid < 0) return;
    int pi = 0;
    pi = prob_of_tile(a, bid);
    const Res8BProb& P = a.p[pi];
    const int tile = bid - P.tile_begin;
    const int ty = tile / P.tiles_x, tx = tile - ty * P.tiles_x;
    const int x0 = tx * TW, y0 = ty * TH;
    const int H = P.H, W = P.W;
    R8F_MARK(1);
    if (!(y0 - 4 >= 0 && y0 + TH + 4 <= H && x0 - 4 >= 0 && x0 + TW + 4 <= W)) {             // border tile: the general form
        res8b_tile<UP>(a, P, x0, y0, lds);
        return;
    }

    // ---- requests, cheapest-to-wait-for first: the biases and the A fragments of conv1 / convR_0 (L2 hits, needed right behind the first
    //      barrier: requested behind the input window they used to arrive after it), then the conv1 input window.  Threads 0 .. 239 own
    //      column t % 40 of the window rows t / 40 + 6 k: ONE division per thread, uniform base pointers + 32-bit byte offsets (the
    //      tensors of a launch lie below 4 GB: run_res8b checks), LDS addresses = per-thread base + immediates.  (First cut: slot u =
    //      tid + 256 i with a division per slot and the source pointer selected per lane -- the compiler LOADED the pointer from the
    //      argument block through a vector address, one dependent L2 round trip in front of every window load.) ----
    const f32x4 bias1 = *reinterpret_cast<const f32x4*>(a.b1 + ch);
    f32x4 biasw[3];
#pragma unroll
    for (int t = 0; t < 3; ++t) biasw[t] = *reinterpret_cast<const f32x4*>(a.bias + 8 * t + ch);
    u32x4 a1[UP ? 6 : 1];
#pragma unroll
    for (int t = 0; t < (UP ? 6 : 1); ++t) a1[t] = (UP ? a.w1pf : a.w1pk)[t * 64 + lane];
    const u32x4* __restrict__ wl = a.wpk + lane;
    u32x4 af[3], ag[3];                                      // the fragments of the current / the next convR (requested a stage ahead)
#pragma unroll
    for (int t = 0; t < 3; ++t) af[t] = wl[t * 64];
    const int lr = tid / IW, lc = tid - lr * IW;              // (threads 240 .. 255: row 6, clamped below, nothing stored)
    const bool ldr = tid < 6 * IW;
    const unsigned wu = (unsigned)W;
    if constexpr (UP) {
        const unsigned goff = ((unsigned)(y0 - 4 + (ldr ? lr : 0)) * wu + (unsigned)(x0 - 4 + lc)) * 16u, gstep = 6u * wu * 16u;
        const unsigned char* __restrict__ sk = reinterpret_cast<const unsigned char*>(P.skip);
        const unsigned char* __restrict__ dc = reinterpret_cast<const unsigned char*>(P.dec);
        u32x4 st[8];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            st[2 * k] = *reinterpret_cast<const u32x4*>(sk + (goff + k * gstep));
            st[2 * k + 1] = *reinterpret_cast<const u32x4*>(dc + (goff + k * gstep));
        }
        // two PLANES of 16 bytes per pixel (skip, deconv), not 32-byte pixel records: a store instruction then writes 64 consecutive 16-byte units
        // (55 instead of 130 ticks per instruction and wave, lesson 44) and conv1's fragment reads take the stages' conflict-free pattern
        unsigned char* const dst = in + (lr * IW + lc) * 16;
        if (ldr) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                *reinterpret_cast<u32x4*>(dst + k * 6 * IW * 16) = st[2 * k];
                *reinterpret_cast<u32x4*>(dst + INPL + k * 6 * IW * 16) = st[2 * k + 1];
            }
        }
    } else {
        const unsigned goff = ((unsigned)(y0 - 4 + (ldr ? lr : 0)) * wu + (unsigned)(x0 - 4 + lc)) * 4u, gstep = 6u * wu * 4u;
        const unsigned char* __restrict__ im = reinterpret_cast<const unsigned char*>(P.img);
        float st[4];
        float mean = 0.f, inv = 1.f;
        if (P.stats) { mean = P.stats[0]; inv = P.stats[1]; }
#pragma unroll
        for (int k = 0; k < 4; ++k) st[k] = *reinterpret_cast<const float*>(im + (goff + k * gstep));
        bf16_t* const dst = reinterpret_cast<bf16_t*>(in) + lr * IW + lc;
        if (ldr) {
#pragma unroll
            for (int k = 0; k < 4; ++k) dst[k * 6 * IW] = (bf16_t)(pack_bf16x2((st[k] - mean) * inv, 0.f) & 0xffffu);
        }
        if (tid < IW / 2) reinterpret_cast<unsigned*>(in)[IH * IW / 2 + tid] = 0u;     // row IH: read with zero weights, must be finite
    }
    R8F_MARK(2);

    // remainder tiles: lane j -> (row rr of the tile's row group, pair pc) for PR pairs per row
    const int j3 = j / 3, rr3 = min(j3, 4), pc3 = j - j3 * 3;  // conv1 / region 0: 6 pixels = 3 pairs, 5 rows per tile (lane 15 idle)
    const int rr2 = j >> 1, pc2 = j & 1;                      // stage 1: 4 pixels = 2 pairs, 8 rows per tile
    const int c = 2 * j + e;                                  // the lane's pixel column in a main tile
    auto relu_pk = [](u32x2 p) { return u32x2{relu_bf16x2(p.x), relu_bf16x2(p.y)}; };
    __syncthreads();
    R8F_MARK(3);
    // convR_1's fragments: requested a whole stage before their first use (requested behind stage 1's MFMAs, as the first cut did, the
    // first MFMA of every stage waited for an L2 round trip)
#pragma unroll
    for (int t = 0; t < 3; ++t) ag[t] = wl[(3 + t) * 64];

    // ---- Every phase below is SOFTWARE-PIPELINED over its pair slots (round 5): a slot = two tiles = its fragment reads, 6 (conv1 UP: 12) MFMAs, the
    //      epilogue (round, ReLU, lane trade, one 16-byte store).  Written slot by slot, the compiler must keep slot s + 1's LDS reads behind slot
    //      s's LDS stores (it cannot see that source and destination regions differ), so a wave ran read -> wait -> MFMAs -> wait -> epilogue ->
    //      store, and only the other waves of the SIMD covered its latencies (ISA of the first cut: two exposed LDS round trips and 10-20 cycles
    //      of s_nop behind the MFMAs per slot).  Here the reads of slot s + 1 are issued BEFORE the MFMAs of slot s and the epilogue of slot s - 1
    //      stands beside the MFMAs of slot s: simple vector instructions hide under a bf16 MFMA of the same wave, two to three per MFMA
    //      (scripts/ubench/bf16mfma_epilogue_coissue.hip).  Source and destination of every phase are different LDS regions. ----
    const bool isB = kk & 1;
    // the lane's whole-pixel record of a tile pair: a lane holds 4 of its pixel's 8 channels, the other 4 sit in lane ^ 16; v_permlane16_swap
    // trades them so that lanes kk = 0 / 2 hold the whole pixel of tile A and lanes kk = 1 / 3 that of tile B -- ONE ds_write_b128 per lane and
    // pair instead of two ds_write_b64 whose 16 lanes sat 32 bytes apart (4-way bank conflicts)
    auto whole = [&](u32x2 pa, u32x2 pb) {
        const auto s0 = __builtin_amdgcn_permlane16_swap(pa.x, pb.x, false, false);
        const auto s1 = __builtin_amdgcn_permlane16_swap(pa.y, pb.y, false, false);
        return u32x4{s0[0], s1[0], s0[1], s1[1]};
    };

    // ---- conv1: relu(t) over the 22 x 38 region (r0), raw t of the centre 16 x 32 (tc) ----
    {
        constexpr int NF = UP ? 6 : 1;
        struct C1F { u32x4 a[NF], b[NF]; };
        // fragments of the pairs whose first pixels are (rowA, colA) / (rowB, colB) of region 0 = input-tile pixels (row .. row + 2, col .. col + 3)
        auto c1_load = [&](C1F& f, int rowA, int colA, int rowB, int colB) {
            if constexpr (UP) {
                // fragment (ky, source): the lane's window pixel kk of filter row ky in the source's plane
                const unsigned char* pa = in + (rowA * IW + colA + kk) * 16;
                const unsigned char* pb = in + (rowB * IW + colB + kk) * 16;
#pragma unroll
                for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                    for (int src = 0; src < 2; ++src) {
                        f.a[ky * 2 + src] = *reinterpret_cast<const u32x4*>(pa + src * INPL + ky * IW * 16);
                        f.b[ky * 2 + src] = *reinterpret_cast<const u32x4*>(pb + src * INPL + ky * IW * 16);
                    }
            } else {
                // k = 8 kk + jj: lane group kk holds window rows 2 kk, 2 kk + 1 (jj >> 2), columns jj & 3 (groups 2, 3: zero weights)
                const unsigned* pa = reinterpret_cast<const unsigned*>(in + ((rowA + 2 * (kk & 1)) * IW + colA) * 2);
                const unsigned* pb = reinterpret_cast<const unsigned*>(in + ((rowB + 2 * (kk & 1)) * IW + colB) * 2);
                f.a[0] = u32x4{pa[0], pa[1], pa[IW / 2], pa[IW / 2 + 1]};
                f.b[0] = u32x4{pb[0], pb[1], pb[IW / 2], pb[IW / 2 + 1]};
            }
        };
        auto c1_mm = [&](const C1F& f, f32x4& ra, f32x4& rb) {
            ra = bias1; rb = bias1;
#pragma unroll
            for (int t = 0; t < NF; ++t) { ra = mfma_bf16_k32(a1[t], f.a[t], ra); rb = mfma_bf16_k32(a1[t], f.b[t], rb); }
        };
        // A main pair = two ADJACENT rows (2 p, 2 p + 1), p = wave + 4 ii, columns 0 .. 31: their windows share two of three input rows (the
        // identical fragment reads are merged: 8 ds_read_b128 per pair instead of 12); pair 11 = rows 22, 23 does not exist (wave 3, ii = 2)
        const int rl0 = 2 * wave + (isB ? 1 : 0);                   // the lane's row of pair `wave`
        unsigned char* const d0 = r0 + (rl0 * W0 + c) * 16;
        unsigned char* const dt = tc + ((rl0 - 3) * TW + c - 3) * 16;
        auto c1_store_main = [&](f32x4 va, f32x4 vb, int ii) {
            const u32x4 raw = whole(pack_bf16x4(va), pack_bf16x4(vb)), rl = relu_bf16x8(raw);   // (ReLU of the traded record: two swaps per pair, not four)
            const int r = rl0 + 8 * ii;
            if (ii < 2 || r < H0) {
                *reinterpret_cast<u32x4*>(d0 + ii * 8 * W0 * 16) = rl;
                if (r >= 3 && r < 3 + TH && c >= 3) *reinterpret_cast<u32x4*>(dt + ii * 8 * TW * 16) = raw;
            }
        };
        // 5 remainder tiles of 5 rows (columns 32 .. 37: 3 pairs per row): tile wave (A), tile wave + 4 (B: wave 0 only)
        const int rowA3 = wave * 5 + rr3, rowB3 = (wave + 4) * 5 + rr3, col3 = 32 + 2 * pc3;
        auto c1_store_rem = [&](f32x4 va, f32x4 vb) {
            const u32x4 raw = whole(pack_bf16x4(va), pack_bf16x4(vb)), rl = relu_bf16x8(raw);
            const int row = isB ? rowB3 : rowA3;
            if (row < H0 && j3 < 5) {
                *reinterpret_cast<u32x4*>(r0 + (row * W0 + col3 + e) * 16) = rl;
                if (row >= 3 && row < 3 + TH && col3 + e < 3 + TW) *reinterpret_cast<u32x4*>(tc + ((row - 3) * TW + col3 + e - 3) * 16) = raw;
            }
        };
        const int rA2 = min(2 * wave + 16, H0 - 2);
        C1F f0, f1;
        f32x4 va0, vb0, va1, vb1;
        c1_load(f0, 2 * wave, 2 * j, 2 * wave + 1, 2 * j);
        c1_load(f1, 2 * wave + 8, 2 * j, 2 * wave + 9, 2 * j);
        c1_mm(f0, va0, vb0);
        c1_load(f0, rA2, 2 * j, rA2 + 1, 2 * j);
        c1_mm(f1, va1, vb1);
        c1_store_main(va0, vb0, 0);
        c1_load(f1, min(rowA3, H0 - 1), col3, min(rowB3, H0 - 1), col3);
        c1_mm(f0, va0, vb0);
        c1_store_main(va1, vb1, 1);
        c1_mm(f1, va1, vb1);
        c1_store_main(va0, vb0, 2);
        c1_store_rem(va1, vb1);
    }
    R8F_MARK(4);
    __syncthreads();
    R8F_MARK(5);

    // ---- stages 1 and 2 (LDS -> LDS): main tile = row r, columns 0 .. 31; remainder tiles = columns 32 .. WO - 1 of 16 / PR rows ----
    struct SF { u32x4 a[3], b[3]; };
    auto st_load = [&](SF& f, const unsigned char* pa, const unsigned char* pb, int pitch) {
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) { f.a[ky] = *reinterpret_cast<const u32x4*>(pa + ky * pitch); f.b[ky] = *reinterpret_cast<const u32x4*>(pb + ky * pitch); }
    };
    auto st_mm = [&](const u32x4 (&w)[3], const SF& f, f32x4 c0, f32x4& ra, f32x4& rb) {
        ra = c0; rb = c0;
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) { ra = mfma_bf16_k32(w[ky], f.a[ky], ra); rb = mfma_bf16_k32(w[ky], f.b[ky], rb); }
    };
    // ReLU'd whole-pixel record of the lane's tile of a pair (ReLU on the packed values: one v_pk_max_i16 per two)
    auto whole_relu = [&](f32x4 va, f32x4 vb) { return whole(relu_pk(pack_bf16x4(va)), relu_pk(pack_bf16x4(vb))); };
    {
        const f32x4 b4 = biasw[0];
        // 20 rows = 10 pairs of adjacent rows (pairs wave, wave + 4 by every wave; 8, 9 by waves 0, 1) + 3 remainder tiles of 8 rows (columns
        // 32 .. 35; tiles 0, 1 by wave 2, tile 2 by wave 3): three pair slots per wave
        const unsigned char* const sb = r0 + (2 * wave * W0 + 2 * j + kk) * 16;
        const int rl0 = 2 * wave + (isB ? 1 : 0);
        unsigned char* const db = r1 + (rl0 * W1 + c) * 16;
        const bool main3 = wave < 2;                                // (wave-uniform)
        const int tl = (wave - 2) * 2, col = 32 + 2 * pc2;          // remainder tiles tl (A), tl + 1 (B)
        const int rowA = min(tl * 8 + rr2, H1 - 1), rowB = min((tl + 1) * 8 + rr2, H1 - 1);
        const unsigned char* const pa3 = main3 ? sb + 16 * W0 * 16 : r0 + (rowA * W0 + col + kk) * 16;
        const unsigned char* const pb3 = main3 ? sb + 17 * W0 * 16 : r0 + (rowB * W0 + col + kk) * 16;
        const int trow = (tl + (isB ? 1 : 0)) * 8 + rr2;            // the lane's row of its remainder tile
        unsigned char* const d3 = main3 ? db + 16 * W1 * 16 : r1 + (trow * W1 + col + e) * 16;
        const bool st3 = main3 || (trow < H1 && tl + (isB ? 1 : 0) < 3);
        SF f0, f1;
        f32x4 va0, vb0, va1, vb1;
        st_load(f0, sb, sb + W0 * 16, W0 * 16);
        st_load(f1, sb + 8 * W0 * 16, sb + 9 * W0 * 16, W0 * 16);
        st_mm(af, f0, b4, va0, vb0);
        st_load(f0, pa3, pb3, W0 * 16);
        st_mm(af, f1, b4, va1, vb1);
        *reinterpret_cast<u32x4*>(db) = whole_relu(va0, vb0);
        st_mm(af, f0, b4, va0, vb0);
        *reinterpret_cast<u32x4*>(db + 8 * W1 * 16) = whole_relu(va1, vb1);
        const u32x4 rec = whole_relu(va0, vb0);
        if (st3) *reinterpret_cast<u32x4*>(d3) = rec;
    }
    R8F_MARK(6);
    __syncthreads();
    R8F_MARK(7);
#pragma unroll
    for (int t = 0; t < 3; ++t) af[t] = wl[(6 + t) * 64];     // convR_2's, a stage ahead (stage 1 was af's last reader)
    {
        const f32x4 b4 = biasw[1];
        constexpr int HO = TH + 2;                            // 18 rows of 34: stage 2's result takes region 0's place (dead)
        // 18 rows = 9 pairs of adjacent rows (pairs wave, wave + 4 by every wave, pair 8 by wave 0) + 2 remainder tiles of 16 rows (columns 32,
        // 33: one pair per row; both by wave 1): waves 0, 1 have a third slot
        const unsigned char* const sb = r1 + (2 * wave * W1 + 2 * j + kk) * 16;
        const int rl0 = 2 * wave + (isB ? 1 : 0);
        unsigned char* const db = r0 + (rl0 * W2 + c) * 16;
        const int trow = (isB ? 16 : 0) + j;                  // remainder tile 0: rows 0 .. 15, tile 1: rows 16, 17
        const unsigned char* const pa3 = wave == 0 ? sb + 16 * W1 * 16 : r1 + (j * W1 + 32 + kk) * 16;
        const unsigned char* const pb3 = wave == 0 ? sb + 17 * W1 * 16 : r1 + (min(16 + j, HO - 1) * W1 + 32 + kk) * 16;
        unsigned char* const d3 = wave == 0 ? db + 16 * W2 * 16 : r0 + (trow * W2 + 32 + e) * 16;
        const bool st3 = wave == 0 || trow < HO;
        SF f0, f1;
        f32x4 va0, vb0, va1, vb1;
        st_load(f0, sb, sb + W1 * 16, W1 * 16);
        st_load(f1, sb + 8 * W1 * 16, sb + 9 * W1 * 16, W1 * 16);
        st_mm(ag, f0, b4, va0, vb0);
        if (wave < 2) st_load(f0, pa3, pb3, W1 * 16);         // (wave-uniform)
        st_mm(ag, f1, b4, va1, vb1);
        *reinterpret_cast<u32x4*>(db) = whole_relu(va0, vb0);
        if (wave < 2) {
            st_mm(ag, f0, b4, va0, vb0);
            *reinterpret_cast<u32x4*>(db + 8 * W2 * 16) = whole_relu(va1, vb1);
            const u32x4 rec = whole_relu(va0, vb0);
            if (st3) *reinterpret_cast<u32x4*>(d3) = rec;
        } else {
            *reinterpret_cast<u32x4*>(db + 8 * W2 * 16) = whole_relu(va1, vb1);
        }
    }
    R8F_MARK(8);
    __syncthreads();
    R8F_MARK(9);

    // ---- stage 3: one tile = one row of the output tile; a wave takes row pairs (2x2 pool in registers) ----
    {
        const f32x4 b4 = biasw[2];
        const int Wp = (W + 1) >> 1;
        // (uniform base pointers + 32-bit byte offsets: the row steps are scalar adds, not 64-bit vector multiply-adds)
        unsigned char* __restrict__ const outb = reinterpret_cast<unsigned char*>(P.out);
        unsigned char* __restrict__ const poolb = reinterpret_cast<unsigned char*>(P.pool);
        const unsigned ooff = (((unsigned)(y0 + 2 * wave) * wu + (unsigned)(x0 + c)) * 8u + (unsigned)ch) * 2u, orow = wu * 16u;
        const unsigned poff = (((unsigned)((y0 >> 1) + wave) * (unsigned)Wp + (unsigned)((x0 >> 1) + j)) * 8u + (unsigned)ch) * 2u, prow = (unsigned)Wp * 16u;
        const unsigned char* const sb = r0 + (2 * wave * W2 + 2 * j + kk) * 16;
        const unsigned char* const tb = tc + (2 * wave * TW + c) * 16 + ch * 2;
        SF f0, f1;
        u32x2 t0[2], t1[2];                                   // the residual operand (raw t) of the two slots' rows
        f32x4 v0[2], v1[2];
        st_load(f0, sb, sb + W2 * 16, W2 * 16);
        st_load(f1, sb + 8 * W2 * 16, sb + 9 * W2 * 16, W2 * 16);
#pragma unroll
        for (int r = 0; r < 2; ++r) { t0[r] = *reinterpret_cast<const u32x2*>(tb + r * TW * 16); t1[r] = *reinterpret_cast<const u32x2*>(tb + (8 + r) * TW * 16); }
        auto finish = [&](const f32x4 (&v2)[2], const u32x2 (&tr)[2], int i) {
            u32x2 pk[2];
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                // ReLU after the rounding: one v_pk_max_i16 per two values
                pk[r] = relu_pk(pack_bf16x4(v2[r] + unpack_bf16x4(tr[r])));
                *reinterpret_cast<u32x2*>(outb + (ooff + (unsigned)(8 * i + r) * orow)) = pk[r];
            }
            if (P.pool) {
                // 2 x 2 max on the PACKED values (non-negative bf16 order like their bit patterns: v_pk_max_i16): the two rows, then the
                // pixel pair (lane ^ 32) through v_permlane32_swap.  (Unpacked to fp32 the pool cost 45 vector instructions per row
                // pair -- fmaxf canonicalises both operands -- of a kernel that is bound by the vector instructions it issues.)
                const unsigned m0 = pkmax_u16(pk[0].x, pk[1].x), m1 = pkmax_u16(pk[0].y, pk[1].y);
                const auto s0 = __builtin_amdgcn_permlane32_swap(m0, m0, false, false);
                const auto s1 = __builtin_amdgcn_permlane32_swap(m1, m1, false, false);
                if (e == 0) *reinterpret_cast<u32x2*>(poolb + (poff + (unsigned)(4 * i) * prow)) = u32x2{pkmax_u16(s0[0], s0[1]), pkmax_u16(s1[0], s1[1])};
            }
        };
        st_mm(af, f0, b4, v0[0], v0[1]);
        st_mm(af, f1, b4, v1[0], v1[1]);
        finish(v0, t0, 0);
        finish(v1, t1, 1);
    }
    R8F_MARK(10);
#if defined(R8F_TRACE)
    __builtin_amdgcn_s_waitcnt(0x0f70);                      // vmcnt(0): the tile's stores have left
#endif
    R8F_MARK(11);
}

// ------------------------------------------------------------------------------------------------
// att_headb_kernel: the attention CNN's head (ARU_v1.py:173-175: 4x4 conv 1 -> 12 + ReLU + 2x2 max pool) of the bf16 path on the bf16 MFMA
// (round 5; until then the fp32 vector-ALU kernel att_headv_kernel<true> served both paths: 384 packed FMAs per pooled pixel, 89 us per page
// against 35 us of HBM time).  Like conv1 of res8f_kernel<false> the layer reads the (standardised) image ROUNDED TO BFLOAT16 and its filter
// rounded to bfloat16; products are exact in fp32, sums fp32.  One MFMA per 16 pixels: M = 12 (of 16) output channels, N = 16 consecutive pixels
// of a row, K = 16 taps (lane group kk < 2: filter rows 2 kk, 2 kk + 1, four columns each; groups 2, 3: zero weights).
// Block = 16 x 64 output pixels (8 x 32 pooled), image tile 19 x 68 as bf16 in LDS (2.6 KB).  A wave takes rows 4 w .. 4 w + 3: a slot = the two
// rows of a pool window x 16 columns; the lane reads THREE image rows (rows r .. r + 2 serve both conv rows) as three dwords each and shifts
// odd columns into place with v_alignbit (a window starts at any column: 2-byte alignment).  Epilogue on the bit patterns: max3(a, b, 0) of the
// two rows, max with lane ^ 1 (DPP), round, one 8-byte store per even lane: the pooled pixel leaves as the 16-channel bf16 plane (12 + 4 zeros;
// the zero-weight rows of the fragment and a zero bias produce the zeros) the next conv's K chunks expect.  ReLU graphs only (elu / leaky keep
// att_headv_kernel).
// ------------------------------------------------------------------------------------------------
constexpr int ATTB_TH = 16, ATTB_TW = 64;
struct AttHeadBArgs {
    C1Prob p[MAXP];        // img, out = pooled [ceil(H/2), ceil(W/2), 16] bf16 (passed as float*), stats, H, W, tiles_x, tile_begin
    int nprob;
    const u32x4* wpk;      // [64 lanes] x 16 bytes: A fragment, row = output channel (12 real), k = 8 kk + 4 r + c <-> tap (2 kk + r, c), kk < 2
    const float* bias;     // [12]
    XcdMap xm;
};
__global__ __launch_bounds__(256, 4) void att_headb_kernel(const AttHeadBArgs a) {
    constexpr int TH = ATTB_TH, TW = ATTB_TW, LH = TH + 3, LW = TW + 4;      // SAME for 4x4: 1 before, 2 after (+ 1 column: dword pairs)
    __shared__ __attribute__((aligned(16))) unsigned short img[LH * LW + 8];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int j = lane & 15, kk = lane >> 4;
    const int bid = sched_tile(a.xm);
    if (bid < 0) return;
    const int pi = prob_of_tile(a, bid);
    const C1Prob& P = a.p[pi];
    const int tile = bid - P.tile_begin;
    const int ty = tile / P.tiles_x, tx = tile - ty * P.tiles_x;
    const int x0 = tx * TW, y0 = ty * TH;
    const int H = P.H, W = P.W;
    const u32x4 af[1] = {a.wpk[lane]};
    f32x4 b4 = f32x4{0.f, 0.f, 0.f, 0.f};
    if (kk < 3) b4 = *reinterpret_cast<const f32x4*>(a.bias + 4 * kk);
    float mean = 0.f, inv = 1.f;
    if (P.stats) { mean = P.stats[0]; inv = P.stats[1]; }
    if (y0 >= 1 && y0 + TH + 2 <= H && x0 >= 1 && x0 + TW + 3 <= W) {
        // the whole image tile inside the image (all but the border tiles): threads 0 .. 203 own column t % 68 of the tile rows t / 68 + 3 k -- one
        // division per thread, no clamps, no inside test (the general form below: a division, four clamps and a test per element)
        constexpr int NK = (LH + 2) / 3;
        const int lr = tid / LW, lc = tid - lr * LW;
        const bool ldr = tid < 3 * LW;
        const float* __restrict__ src = P.img + (size_t)(y0 - 1 + (ldr ? lr : 0)) * W + (x0 - 1 + lc);
        float st[NK];
#pragma unroll
        for (int k = 0; k < NK; ++k) st[k] = src[(size_t)min(3 * k, LH - 1 - (ldr ? lr : 0)) * W];
        if (ldr) {
#pragma unroll
            for (int k = 0; k < NK; ++k)
                if (3 * k + 2 < LH || lr + 3 * k < LH) img[(lr + 3 * k) * LW + lc] = (unsigned short)(pack_bf16x2((st[k] - mean) * inv, 0.f) & 0xffffu);
        }
    } else {   // requests first, LDS writes afterwards
        constexpr int NSL = (LH * LW + 255) / 256;
        float st[NSL];
#pragma unroll
        for (int k = 0; k < NSL; ++k) {
            const int i = min(tid + k * 256, LH * LW - 1);
            const int r = i / LW, c = i - r * LW;
            st[k] = P.img[(size_t)min(max(y0 - 1 + r, 0), H - 1) * W + min(max(x0 - 1 + c, 0), W - 1)];
        }
#pragma unroll
        for (int k = 0; k < NSL; ++k) {
            const int i = tid + k * 256;
            if (i < LH * LW) {
                const int r = i / LW, c = i - r * LW;
                const int gy = y0 - 1 + r, gx = x0 - 1 + c;
                img[i] = (gy >= 0 && gy < H && gx >= 0 && gx < W) ? (unsigned short)(pack_bf16x2((st[k] - mean) * inv, 0.f) & 0xffffu) : (unsigned short)0;
            }
        }
    }
    if (tid < 8) img[LH * LW + tid] = 0;                      // (slack behind the last row: finite data for any over-read)
    __syncthreads();
    const bool interior = y0 + TH <= H && x0 + TW <= W;       // every 2x2 window of the tile is complete
    const unsigned Wp = (unsigned)(W + 1) >> 1;
    const unsigned sh = (unsigned)(j & 1) * 16u;
    // the lane's dword of (tile row 4 wave + 2 (kk & 1), window column j) -- slots add (2 rp) rows and 16 cb columns
    const unsigned* const base = reinterpret_cast<const unsigned*>(img) + ((4 * wave + 2 * (kk & 1)) * LW + j) / 2;
    // pooled pixel of slot (rp, cb): (y0 / 2 + 2 wave + rp, x0 / 2 + 8 cb + j / 2): a uniform base pointer + 32-bit element offsets (the pooled
    // plane of a page is far below 2^31 elements)
    unsigned short* __restrict__ const ob = reinterpret_cast<unsigned short*>(P.out);
    const unsigned obase = ((unsigned)((y0 >> 1) + 2 * wave) * Wp + (unsigned)((x0 >> 1) + (j >> 1))) * 16u + 4u * (unsigned)kk, orow = Wp * 16u;
    const bool even = (j & 1) == 0;
    typedef FragPair<1> Fr;
    typedef int i32x4 __attribute__((ext_vector_type(4)));
    pipe_slots<8, 1>(af, b4,
        [&](auto sc, Fr& f) {
            constexpr int s = decltype(sc)::value, rp = s >> 2, cb = s & 3;
            const unsigned* p = base + (2 * rp * LW + 16 * cb) / 2;
            unsigned v[3][2];
#pragma unroll
            for (int r = 0; r < 3; ++r) {
                const unsigned d0 = p[r * (LW / 2)], d1 = p[r * (LW / 2) + 1], d2 = p[r * (LW / 2) + 2];
                v[r][0] = __builtin_amdgcn_alignbit(d1, d0, sh);
                v[r][1] = __builtin_amdgcn_alignbit(d2, d1, sh);
            }
            f.a[0] = u32x4{v[0][0], v[0][1], v[1][0], v[1][1]};
            f.b[0] = u32x4{v[1][0], v[1][1], v[2][0], v[2][1]};
        },
        [&](auto sc, f32x4 va, f32x4 vb) {
            constexpr int s = decltype(sc)::value, rp = s >> 2, cb = s & 3;
            i32x4 ia = __builtin_bit_cast(i32x4, va), ib = __builtin_bit_cast(i32x4, vb);
            bool ok = even;
            if (!interior) {                                  // a conv output outside the image contributes 0 = what the ReLU leaves of it anyway
                const int gy = y0 + 4 * wave + 2 * rp, gx = x0 + 16 * cb + j;
                if (gx >= W) { ia = i32x4{0, 0, 0, 0}; ib = ia; }
                if (gy + 1 >= H) ib = i32x4{0, 0, 0, 0};
                ok = ok && gy < H && gx < W;
            }
            // relu(max over the window) on the bit patterns (a negative float is a negative int): max3(a, b, 0), then the column partner
            int m[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const int t = max(max(ia[c], ib[c]), 0);
                m[c] = max(t, __builtin_amdgcn_update_dpp(0, t, 0xB1, 0xF, 0xF, true));
            }
            if (ok) {
                const u32x2 pk = u32x2{pack_bf16x2(__int_as_float(m[0]), __int_as_float(m[1])), pack_bf16x2(__int_as_float(m[2]), __int_as_float(m[3]))};
                *reinterpret_cast<u32x2*>(ob + (obase + (unsigned)rp * orow + (unsigned)(cb * 8 * 16))) = pk;
            }
        });
}

// channel sum [H,W,8] bf16 -> [H,W] fp32 (upsample_simple's channel-summing half; same association as chansum_kernel)
// maxpool2 (ceil mode) of a bf16 NHWC tensor of NON-NEGATIVE values (a ReLU layer's output): the integer order of the bit patterns is the float order
// (v_pk_max_i16-free: pkmax_u16).  Behind convr_kernel's RES form, which has no fused pool (unet_down_3/convR_2).  One thread = 8 channels of one output pixel.
struct MaxPoolBProb {
    const bf16_t* in;
    bf16_t* out;
    int H, W;              // input size; output ceil(H / 2) x ceil(W / 2)
    int blk_begin, pad_;
};
struct MaxPoolBArgs {
    MaxPoolBProb p[MAXP];
    int nprob, C;          // C % 8 == 0
};
__global__ __launch_bounds__(256) void maxpool2b_kernel(const MaxPoolBArgs a) {
    int pi = 0;
    while (pi + 1 < a.nprob && (int)blockIdx.x >= a.p[pi + 1].blk_begin) ++pi;
    const MaxPoolBProb& P = a.p[pi];
    const int c8 = a.C >> 3, Ho = (P.H + 1) >> 1, Wo = (P.W + 1) >> 1;
    const unsigned item = (blockIdx.x - P.blk_begin) * 256u + threadIdx.x;
    if (item >= (unsigned)Ho * Wo * c8) return;
    const unsigned pix = item / c8, c = item - pix * c8;
    const unsigned oy = pix / Wo, ox = pix - oy * Wo;
    const unsigned y0 = 2 * oy, x0 = 2 * ox, y1 = min(y0 + 1, (unsigned)P.H - 1), x1 = min(x0 + 1, (unsigned)P.W - 1);
    const u32x4* in = reinterpret_cast<const u32x4*>(P.in);
    const u32x4 v00 = in[((size_t)y0 * P.W + x0) * c8 + c], v01 = in[((size_t)y0 * P.W + x1) * c8 + c];
    const u32x4 v10 = in[((size_t)y1 * P.W + x0) * c8 + c], v11 = in[((size_t)y1 * P.W + x1) * c8 + c];
    auto mx = [](unsigned p, unsigned q) { return pkmax_u16(p, q); };
    reinterpret_cast<u32x4*>(P.out)[(size_t)pix * c8 + c] = u32x4{mx(mx(v00.x, v01.x), mx(v10.x, v11.x)), mx(mx(v00.y, v01.y), mx(v10.y, v11.y)),
                                                                  mx(mx(v00.z, v01.z), mx(v10.z, v11.z)), mx(mx(v00.w, v01.w), mx(v10.w, v11.w))};
}

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
    pi = prob_of_blk(a, (int)blockIdx.x);
    const PoolBProb& P = a.p[pi];
    const size_t total = (size_t)P.H * P.W;
    const size_t base = (size_t)(blockIdx.x - P.blk_begin) * POOL_ITEMS;
    if (a.C == 8) {
        // the usual case: one 16-byte load per pixel, the thread's four pixels requested before the first sum (one round trip, not four)
        u32x4 v[POOL_ITEMS / 256];
#pragma unroll
        for (int k = 0; k < POOL_ITEMS / 256; ++k) {
            const size_t i = min(base + k * 256 + threadIdx.x, total - 1);
            v[k] = *reinterpret_cast<const u32x4*>(P.in + i * 8);
        }
#pragma unroll
        for (int k = 0; k < POOL_ITEMS / 256; ++k) {
            const size_t i = base + k * 256 + threadIdx.x;
            if (i >= total) break;
            const f32x4 lo = unpack_bf16x4(u32x2{v[k].x, v[k].y}), hi = unpack_bf16x4(u32x2{v[k].z, v[k].w});
            P.out[i] = (((((((0.f + lo.x) + lo.y) + lo.z) + lo.w) + hi.x) + hi.y) + hi.z) + hi.w;
        }
        return;
    }
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
