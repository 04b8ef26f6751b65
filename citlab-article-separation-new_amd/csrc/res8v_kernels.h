// Fused level-0 residual blocks of the ARU-Net in fp32 on the VECTOR ALU (ARU_v1.py:208-245 and :266-281).
//
// Same tiling, LDS buffers and pass structure as res8_kernels.h (16 x 58 pixel output tiles, R8_NP carried passes), but
// every 3x3 8->8 convolution runs as packed FMAs instead of v_mfma_f32_16x16x4_f32.  Eight output channels fill half of a
// 16-row MFMA tile (the pixel-pair mapping recovers 75 %: 6.1 k cycles of MFMA issue per 16 x 64 pixel stage, 7.5-8.2 k
// measured), while v_pk_fma_f32 has no such granularity and the f32 vector peak of gfx950 equals the f32 matrix peak
// (157 TFLOP/s).  A thread owns two horizontally adjacent pixels x 8 output channels; a weight pair (two output channels
// of one (tap, input channel)) is a scalar register pair, the input value is broadcast to both halves (op_sel_hi), so
// one v_pk_fma_f32 = 2 output channels x 64 lanes and a stage is 576 of them per thread: 4.6 k cycles per stage at
// 4 cycles per instruction (scripts/ubench/pk_fma_issue.hip measures 4.38 at two waves per SIMD), 6.1 k measured
// (scripts/ubench/valu_conv8.hip).  bf16 keeps the MFMA kernels (res8_kernels.h): there the matrix rate is 8x the vector rate.
#pragma once
#include "res8_kernels.h"

namespace asep {

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef const float __attribute__((address_space(4)))* r8v_cptr;   // constant address space: stays on s_load
typedef char __attribute__((address_space(3)))* r8v_lds;           // LDS byte address (32 bits, one VGPR)

#ifndef R8V_WINO
#define R8V_WINO 0                  // 1: Winograd F(2,3) along x (r8v_conv_wino) instead of the direct form: an experiment, see there
#endif
#ifndef R8V_WINO_GS
#define R8V_WINO_GS 16              // scalars per weight group of r8v_conv_wino (16 or 32)
#endif
// floats per 8->8 filter in scalar layout.  Direct: [g = (ky*2 + hf)*3 + kx][c = ci & 3][co], ci = hf*4 + c.
// Winograd: [(ky*2 + hf)*4 + j][c][co], j = position of F(2,3): U_j = sum_kx G[j][kx] g[ky][kx] (packed in double by the engine)
constexpr int R8V_FILTER = R8V_WINO ? 768 : 576;

// The kernels take Res8Args like the MFMA kernels; the filters behind w1 (UP) and wr are in scalar layout instead of
// pixel-pair fragments: UP w1 = [2 sources][R8V_FILTER], wr = [3][R8V_FILTER] floats.

// Float offset of the 16-byte piece (pixel x of the row, channel half hf) inside an LDS row, for the vector-ALU kernels.  The
// layout idea is r8_px's (res8_kernels.h: the four pieces of a 64-byte pixel-pair record permuted by the pair index P = x >> 1),
// but the permutation is chosen for THIS access pattern: a thread reads the same piece of pair P = p + const, p = lane & 31, and
// a ds_read_b128 is served in the lane groups {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31} (MI355X_MICROARCH.md, LDS table) -- not
// in runs of 16 consecutive lanes.  With r8_px's term (P >> 1) & 3 lanes 12..15 and 20..23 (and 0..3 / 24..27) of a group meet
// in the same bank quad: SQ_LDS_BANK_CONFLICT was 37-41 % of SQ_LDS_IDX_ACTIVE in res8v_down / res8v_up (profiles/r2t).  The
// term below, bit 2 of P | (bit 1 ^ bit 3 of P) << 1, gives 16 distinct bank quads per group for every window offset (exhaustive
// check over the XOR-linear candidates: scripts/ubench/lds_swizzle_search.py; none is conflict-free for the 8-lane groups of the
// ds_write_b128 as well -- the stores, 4 against 24 reads per stage, stay 2-way in half of their groups).
__device__ __forceinline__ int r8v_px(int x, int hf) {
    const int P = x >> 1;
    const int g = ((P >> 2) & 1) | ((((P >> 1) ^ (P >> 3)) & 1) << 1);
    return (P << 4) + (((((x & 1) << 1) | hf) ^ g) << 2);
}

// the block's activation: ReLU on the bit patterns (ACT 0, the shipped graphs' fast path), or elu (1) / leaky (2) of the graph variants
// (ARU_v1.py:70-75; act4 of aru_kernels.h: the arithmetic of the layer-by-layer form)
template <int ACT>
__device__ __forceinline__ f32x4 r8v_act(f32x4 v) {
    if constexpr (ACT == 0) return relu4i(v);
    else return act4(v, ACT);
}
__device__ __forceinline__ f32x4 r8v_ld(r8v_lds p) { return *reinterpret_cast<const f32x4 __attribute__((address_space(3)))*>(p); }
__device__ __forceinline__ void r8v_st(r8v_lds p, f32x4 v) { *reinterpret_cast<f32x4 __attribute__((address_space(3)))*>(p) = v; }

// one packed FMA: acc += {v, v} * w, v = element (odd ? 1 : 0) of the 64-bit register pair `pair` (two neighbouring input
// channels).  Written as asm: left to instruction selection, some of the odd elements were first copied to the low half
// of another pair (a v_mov per 8 FMAs; the vector ALU is the bound here, every extra instruction costs its 4 cycles).
template <int ODD>
__device__ __forceinline__ void r8v_fma(f32x2& acc, f32x2 pair, f32x2 w) {
    if (ODD) asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,0,0]" : "+v"(acc) : "v"(pair), "s"(w));
    else asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1]" : "+v"(acc) : "v"(pair), "s"(w));
}

// acc[px][2q..2q+1] += sum over the 3x3 window and 8 input channels, for the thread's two pixels.  a[i][hf] = LDS address
// of (first input row = output row - 1, pixel x - 1 + i, channel half hf); the rows below are immediate offsets of the
// same eight address registers.
//
// Explicit software pipeline: scalar loads return out of order, so every wait on them is lgkmcnt(0) and covers the LDS
// reads as well.  Per weight group (16 scalars = 2 input channels x 8 output channels, 16 packed FMAs) there is ONE wait,
// placed first (the empty asm "uses" the group's registers), and only then are the next group's weights and the next
// row-half's four input pieces requested; left to the compiler the request goes out BEFORE the wait, which then covers
// it.  The double buffer is 2 x 16 SGPRs: the kernels carry ~40 scalars of their own and the file has 102 (with 2 x 32
// the allocator spilled weights to VGPR lanes, thousands of v_readlane); the stage time is the same (6.0 k cycles).
template <bool RELU_IN, int ACT = 0>
__device__ __forceinline__ void r8v_conv_direct(const r8v_lds (&a)[4][2], r8v_cptr wl,
                                         f32x2 (&acc0)[4], f32x2 (&acc1)[4]) {
    float wc[16], wn[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) wc[k] = wl[k];
    f32x4 dA[4], dB[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) dA[i] = *reinterpret_cast<const f32x4 __attribute__((address_space(3)))*>(a[i][0]);
#pragma unroll
    for (int g2 = 0; g2 < 36; ++g2) {
        const int g = g2 >> 1, ch = g2 & 1, kx = g % 3, rh = g / 3;           // rh = ky * 2 + hf
        asm volatile("" :: "s"(wc[0]), "v"(dA[0]), "v"(dA[1]), "v"(dA[2]), "v"(dA[3]));
        __builtin_amdgcn_sched_barrier(0);
        if (g2 + 1 < 36) {
#pragma unroll
            for (int k = 0; k < 16; ++k) wn[k] = wl[(g2 + 1) * 16 + k];
        }
        if (kx == 0 && ch == 0 && rh + 1 < 6) {
            const int ky2 = (rh + 1) >> 1, hf2 = (rh + 1) & 1;
#pragma unroll
            for (int i = 0; i < 4; ++i) dB[i] = *reinterpret_cast<const f32x4 __attribute__((address_space(3)))*>(a[i][hf2] + ky2 * R8_PITCH * 32);
        }
        __builtin_amdgcn_sched_barrier(0);
        if (RELU_IN && kx == 0 && ch == 0) {               // (the ReLU behind conv1 of a residual block is a ReLU in every graph variant: ARU_v1.py:212-216)
#pragma unroll
            for (int i = 0; i < 4; ++i) dA[i] = relu4i(dA[i]);
        }
        // the group's two input channels are one register pair of each pixel: (x, y) for ch = 0, (z, w) for ch = 1
        const f32x2 in0 = ch ? f32x2{dA[kx].z, dA[kx].w} : f32x2{dA[kx].x, dA[kx].y};
        const f32x2 in1 = ch ? f32x2{dA[kx + 1].z, dA[kx + 1].w} : f32x2{dA[kx + 1].x, dA[kx + 1].y};
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const f32x2 wv = f32x2{wc[2 * q], wc[2 * q + 1]};
            r8v_fma<0>(acc0[q], in0, wv);
            r8v_fma<0>(acc1[q], in1, wv);
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const f32x2 wv = f32x2{wc[8 + 2 * q], wc[8 + 2 * q + 1]};
            r8v_fma<1>(acc0[q], in0, wv);
            r8v_fma<1>(acc1[q], in1, wv);
        }
#pragma unroll
        for (int k = 0; k < 16; ++k) wc[k] = wn[k];
        if (kx == 2 && ch == 1) {
#pragma unroll
            for (int i = 0; i < 4; ++i) dA[i] = dB[i];
        }
    }
    // pin the accumulators here: without a use in this block the compiler SINKS all 576 FMAs past the next branch (to the
    // first use of the sums), away from their weights, which it then parks in VGPR lanes (thousands of v_readlane)
    asm volatile("" : "+v"(acc0[0]), "+v"(acc0[1]), "+v"(acc0[2]), "+v"(acc0[3]), "+v"(acc1[0]), "+v"(acc1[1]), "+v"(acc1[2]), "+v"(acc1[3]));
}

// first product of an accumulator: acc = {v, v} * w (no zero-initialisation of the 16 Winograd sums)
template <int ODD>
__device__ __forceinline__ void r8v_mul(f32x2& acc, f32x2 pair, f32x2 w) {
    if (ODD) asm("v_pk_mul_f32 %0, %1, %2 op_sel:[1,0]" : "=v"(acc) : "v"(pair), "s"(w));
    else asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[0,1]" : "=v"(acc) : "v"(pair), "s"(w));
}

// The same sums by Winograd F(2,3) ALONG X: the thread's two output pixels y0, y1 of a row come from its four window
// pixels d0..d3 as  y0 = m0 + m1 + m2,  y1 = m1 - m2 - m3  with  m_j = V_j * U_j,  V = (d0 - d2, d1 + d2, d2 - d1, d1 - d3),
// U = (g0, (g0 + g1 + g2) / 2, (g0 - g1 + g2) / 2, g2)  per (ky, ci, co): 4 instead of 6 products per row and channel.
// Per thread and stage: 48 packed adds for V (on channel pairs), 384 packed FMAs, 24 for the output: 456 vector
// instructions instead of 576.  One-dimensional on purpose: the thread mapping, the LDS reads and the epilogues stay
// those of the direct form (a 2-D tile would need 2 x 2 pixels per thread: half the threads, or the output channels
// split across waves with the input transform done twice).  fp32 throughout; the transform matrices hold 0, +-1, 1/2.
// Measured (parity-green, profiles/r2p_valu): a stage whose weights hit the scalar cache gains 6-9 % (conv1 5.7 -> 5.2 k cycles,
// stage 1 6.5 -> 6.1 k: the waits now come every 8 FMAs), res8v_down 3.76 -> 3.69 ms; but the UP block's five filters grow
// from 11.5 to 15.4 KB, which no longer fits the 16 KB scalar data cache: its second conv1 half went from 5.8 to 9.1 k
// cycles, the last stage from 6.8 to 9.7 k, the kernel from 5.40 to 6.33 ms.  Not used (-DR8V_WINO=1 builds it).
template <bool RELU_IN, int ACT = 0>
__device__ __forceinline__ void r8v_conv_wino(const r8v_lds (&a)[4][2], r8v_cptr wl,
                                              f32x2 (&acc0)[4], f32x2 (&acc1)[4]) {
    constexpr int GS = R8V_WINO_GS, NG = 768 / GS, GPR = 128 / GS;           // groups per (ky, hf) row-half
    float wc[GS], wn[GS];
#pragma unroll
    for (int k = 0; k < GS; ++k) wc[k] = wl[k];
    f32x4 dA[4], dB[4], V[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) dA[i] = r8v_ld(a[i][0]);
    f32x2 M[4][4];                                      // [position][output channel pair]
#pragma unroll
    for (int g = 0; g < NG; ++g) {
        const int rh = g / GPR, gi = g % GPR;           // rh = ky * 2 + hf
        const int j = GS == 32 ? gi : gi >> 1;
        asm volatile("" :: "s"(wc[0]), "v"(dA[0]), "v"(dA[1]), "v"(dA[2]), "v"(dA[3]));
        __builtin_amdgcn_sched_barrier(0);
        if (g + 1 < NG) {
#pragma unroll
            for (int k = 0; k < GS; ++k) wn[k] = wl[(g + 1) * GS + k];
        }
        if (gi == 0 && rh + 1 < 6) {
            const int ky2 = (rh + 1) >> 1, hf2 = (rh + 1) & 1;
#pragma unroll
            for (int i = 0; i < 4; ++i) dB[i] = r8v_ld(a[i][hf2] + ky2 * R8_PITCH * 32);
        }
        __builtin_amdgcn_sched_barrier(0);
        if (gi == 0) {
            if (RELU_IN) {
#pragma unroll
                for (int i = 0; i < 4; ++i) dA[i] = relu4i(dA[i]);
            }
            V[0] = dA[0] - dA[2]; V[1] = dA[1] + dA[2]; V[2] = dA[2] - dA[1]; V[3] = dA[1] - dA[3];
        }
#pragma unroll
        for (int cc = 0; cc < GS / 16; ++cc) {
            const int ch = GS == 32 ? cc : gi & 1;          // channel pair of the half: (x, y) or (z, w)
            const f32x2 in = ch ? f32x2{V[j].z, V[j].w} : f32x2{V[j].x, V[j].y};
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f32x2 wv = f32x2{wc[cc * 16 + 2 * q], wc[cc * 16 + 2 * q + 1]};
                if (rh == 0 && ch == 0) r8v_mul<0>(M[j][q], in, wv);
                else r8v_fma<0>(M[j][q], in, wv);
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f32x2 wv = f32x2{wc[cc * 16 + 8 + 2 * q], wc[cc * 16 + 8 + 2 * q + 1]};
                r8v_fma<1>(M[j][q], in, wv);
            }
        }
#pragma unroll
        for (int k = 0; k < GS; ++k) wc[k] = wn[k];
        if (gi == GPR - 1) {
#pragma unroll
            for (int i = 0; i < 4; ++i) dA[i] = dB[i];
        }
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        acc0[q] += M[0][q] + M[1][q] + M[2][q];
        acc1[q] += M[1][q] - M[2][q] - M[3][q];
    }
    // pinned like the direct form (see there)
    asm volatile("" : "+v"(acc0[0]), "+v"(acc0[1]), "+v"(acc0[2]), "+v"(acc0[3]), "+v"(acc1[0]), "+v"(acc1[1]), "+v"(acc1[2]), "+v"(acc1[3]));
}

template <bool RELU_IN, int ACT = 0>
__device__ __forceinline__ void r8v_conv(const r8v_lds (&a)[4][2], r8v_cptr wl, f32x2 (&acc0)[4], f32x2 (&acc1)[4]) {
    if constexpr (R8V_WINO) r8v_conv_wino<RELU_IN, ACT>(a, wl, acc0, acc1);
    else r8v_conv_direct<RELU_IN, ACT>(a, wl, acc0, acc1);
}

__device__ __forceinline__ f32x4 r8v_lo(const f32x2 (&a)[4]) { return f32x4{a[0].x, a[0].y, a[1].x, a[1].y}; }
__device__ __forceinline__ f32x4 r8v_hi(const f32x2 (&a)[4]) { return f32x4{a[2].x, a[2].y, a[3].x, a[3].y}; }
__device__ __forceinline__ f32x4 r8v_max4(f32x4 a, f32x4 b) { return f32x4{fmaxf(a.x, b.x), fmaxf(a.y, b.y), fmaxf(a.z, b.z), fmaxf(a.w, b.w)}; }
__device__ __forceinline__ f32x4 r8v_upper4(f32x4 v) {
    return f32x4{from_upper_half(v.x), from_upper_half(v.y), from_upper_half(v.z), from_upper_half(v.w)};
}

// byte offsets inside an LDS row of the seven pixels 2p .. 2p + 6 (p = tid & 31) a thread ever touches, both channel halves:
// a stage with first output column c0 reads pixels c0 - 1 + 2p + i (i = 0..3) = entries c0 - 1 + i.  Computed once per kernel.
__device__ __forceinline__ void r8v_pixel_offsets(int tid, int (&poff)[7][2]) {
#pragma unroll
    for (int j = 0; j < 7; ++j) { poff[j][0] = r8v_px(2 * (tid & 31) + j, 0) * 4; poff[j][1] = r8v_px(2 * (tid & 31) + j, 1) * 4; }
}
// the eight address registers of a window: row_bytes = byte offset of the first input row from the LDS base.  Laundered
// so that the compiler keeps the sums in registers (re-associated, every LDS access would pay a v_add of its own: the
// buffers lie beyond the 64 KiB reach of the instruction's offset field)
__device__ __forceinline__ void r8v_window(r8v_lds (&a)[4][2], float* sm, const int (&poff)[7][2], int j0, int row_bytes) {
    asm volatile("" : "+v"(row_bytes));   // one row term, eight sums (not eight induction variables in the callers' row loops)
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) { a[i][hf] = (r8v_lds)reinterpret_cast<char*>(sm) + (poff[j0 + i][hf] + row_bytes); asm volatile("" : "+v"(a[i][hf])); }
}


// one 3x3 8->8 convolution stage on LDS tiles.  IN holds rows in_r0.. of the frame, OUT rows out_r0.. ; computes rows
// [row_start, row_start + nrows) x columns [out_c0, out_c0 + 64): thread -> (row tid >> 5 (+16), pixels out_c0 + 2 (tid & 31), +1).
// FINAL: add T centre, ReLU, store the OW valid columns to global (+ 2x2 max pool).  Otherwise ReLU into OUT, zero outside
// the image (= the SAME padding of the next convolution).  interior (scalar): the whole frame lies inside the image.
template <bool RELU_IN, bool FINAL, bool POOL, int ACT = 0>
__device__ __forceinline__ void res8v_stage(float* __restrict__ sm, const float* __restrict__ IN, int in_r0, float* __restrict__ OUT, int out_r0,
                                            int row_start, int nrows, int out_c0, const float* __restrict__ w,
                                            const float* __restrict__ bias, int tid, const int (&poff)[7][2], bool interior,
                                            int fy0, int fx0, int H, int W,
                                            const float* __restrict__ T, int t_r0, float* __restrict__ gout, float* __restrict__ gpool) {
    const int x = out_c0 + 2 * (tid & 31);
    const int gx = fx0 + x;
    // centre pixels of OUT / T relative to the window's address registers: compile-time constants
    const int d_out = FINAL ? 0 : (int)(OUT - IN) * 4 + (in_r0 + 1 - out_r0) * R8_PITCH * 32;
    const int d_t = FINAL ? (int)(T - IN) * 4 + (in_r0 + 1 - t_r0) * R8_PITCH * 32 : 0;
#pragma unroll 1
    for (int r = tid >> 5; r < nrows; r += 16) {
        r8v_cptr wl = (r8v_cptr)w, bl = (r8v_cptr)bias;
        asm volatile("" : "+s"(wl));                  // the loads stay inside the loop (hoisted they would be spilled to VGPR lanes)
        f32x2 acc0[4], acc1[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) { acc0[q] = f32x2{bl[2 * q], bl[2 * q + 1]}; acc1[q] = acc0[q]; }
        const int row = row_start + r;               // frame row
        r8v_lds a[4][2];
        r8v_window(a, sm, poff, out_c0 - 1, ((int)(IN - sm) + (row - 1 - in_r0) * R8_PITCH * 8) * 4);
        r8v_conv<RELU_IN, ACT>(a, wl, acc0, acc1);
        const int gy = fy0 + row;
        f32x4 p0l = r8v_lo(acc0), p0h = r8v_hi(acc0), p1l = r8v_lo(acc1), p1h = r8v_hi(acc1);
        if (!FINAL) {
            p0l = r8v_act<ACT>(p0l); p0h = r8v_act<ACT>(p0h); p1l = r8v_act<ACT>(p1l); p1h = r8v_act<ACT>(p1h);
            if (!interior) {
                const bool oky = gy >= 0 && gy < H;
                const bool ok0 = oky && gx >= 0 && gx < W, ok1 = oky && gx + 1 >= 0 && gx + 1 < W;
                const f32x4 z = f32x4{0.f, 0.f, 0.f, 0.f};
                p0l = ok0 ? p0l : z; p0h = ok0 ? p0h : z; p1l = ok1 ? p1l : z; p1h = ok1 ? p1h : z;
            }
            r8v_st(a[1][0] + d_out, p0l);
            r8v_st(a[1][1] + d_out, p0h);
            r8v_st(a[2][0] + d_out, p1l);
            r8v_st(a[2][1] + d_out, p1h);
        } else {
            p0l = r8v_act<ACT>(p0l + r8v_ld(a[1][0] + d_t));
            p0h = r8v_act<ACT>(p0h + r8v_ld(a[1][1] + d_t));
            p1l = r8v_act<ACT>(p1l + r8v_ld(a[2][0] + d_t));
            p1h = r8v_act<ACT>(p1h + r8v_ld(a[2][1] + d_t));
            // only the OW valid columns of the tile (frame columns 4 .. 4+OW-1) are stored; gy >= 0 and gx >= 0 there.
            // Element offsets fit 32 bits (the launcher sends larger tensors to the MFMA kernels).
            const bool oky = interior || gy < H;
            const bool ok0 = x >= 4 && x < 4 + R8_OW && (interior || gx < W), ok1 = x + 1 >= 4 && x + 1 < 4 + R8_OW && (interior || gx + 1 < W);
            float* o = gout + ((unsigned)gy * (unsigned)W + (unsigned)gx) * 8u;
            if (ok0 && oky) { *reinterpret_cast<f32x4*>(o) = p0l; *reinterpret_cast<f32x4*>(o + 4) = p0h; }
            if (ok1 && oky) { *reinterpret_cast<f32x4*>(o + 8) = p1l; *reinterpret_cast<f32x4*>(o + 12) = p1h; }
            if (POOL && gpool) {
                // 2x2 max: the x neighbour is the thread's second pixel (gx is even), the row below sits in lane + 32
                // (gy is even for lanes < 32: a wave holds rows 2 * wave and 2 * wave + 1); windows never straddle tiles
                f32x4 ml = ok1 ? r8v_max4(p0l, p1l) : p0l, mh = ok1 ? r8v_max4(p0h, p1h) : p0h;
                const f32x4 ul = r8v_upper4(ml), uh = r8v_upper4(mh);     // only lanes < 32 use it
                if ((tid & 32) == 0 && ok0 && oky) {
                    if (interior || gy + 1 < H) { ml = r8v_max4(ml, ul); mh = r8v_max4(mh, uh); }
                    const unsigned Wp = (unsigned)(W + 1) >> 1;
                    float* po = gpool + ((unsigned)(gy >> 1) * Wp + (unsigned)(gx >> 1)) * 8u;
                    *reinterpret_cast<f32x4*>(po) = ml;
                    *reinterpret_cast<f32x4*>(po + 4) = mh;
                }
            }
        }
    }
}

// DOWN block of level 0: image (1 channel) -> d0 [H,W,8] (+ maxpool2)
template <int ACT = 0>
__global__ __launch_bounds__(R8_THREADS) __attribute__((amdgpu_waves_per_eu(2, 2))) void res8v_down_kernel(const Res8Args a) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const float* wr = reinterpret_cast<const float*>(a.wr);
    float* IMG = sm;                                         // [24][76]
    float* T = IMG + R8_FH * R8_IMGP;                        // frame rows 1..22  [22][72][8]
    float* R0 = T + 22 * R8_PITCH * 8;                       // frame rows 2..21  [20][72][8]
    float* R1 = R0 + 20 * R8_PITCH * 8;                      // frame rows 3..20  [18][72][8]
    __shared__ float w1s[9 * 8 + 8];
    const int tid = threadIdx.x;
    if (tid < 72) w1s[tid] = a.w1[tid];
    if (tid < 8) w1s[72 + tid] = a.b1[tid];
    int poff[7][2];
    r8v_pixel_offsets(tid, poff);

    constexpr int NPRE = (R8_FH * R8_IMGP + R8_THREADS - 1) / R8_THREADS;
    float pre[NPRE], pre_mean = 0.f, pre_inv = 1.f;
    unsigned pre_mask = 0;
    // frame = one 24-row window of a work unit: (tile, pass) -> image rows [(tyb * NP + pass) * OH - 4, +24)
    auto image_load = [&](int tile_id, int pass) {          // next frame's image values -> registers (in flight under the stages)
        int pi = 0;
        pi = prob_of_tile_search(a, tile_id);
        const Res8Prob& Q = a.p[pi];
        const int t = tile_id - Q.tile_begin;
        const int tyb = t / Q.tiles_x, txb = t - tyb * Q.tiles_x;
        const int qy0 = (tyb * R8_NP + pass) * R8_OH - 4, qx0 = txb * R8_OW - 4;
        // the standardisation is applied when the registers are written to LDS, not here (arithmetic on the loaded value
        // would make this phase wait for the loads it has only just issued)
        pre_mean = 0.f; pre_inv = 1.f; pre_mask = 0;
        if (Q.stats) { pre_mean = Q.stats[0]; pre_inv = Q.stats[1]; }
#pragma unroll
        for (int k = 0; k < NPRE; ++k) {
            const int i = tid + k * R8_THREADS;
            const int r = i / R8_IMGP, c = i - r * R8_IMGP;
            const int gy = qy0 + r, gx = qx0 + c - 2;
            const bool ok = i < R8_FH * R8_IMGP && gy >= 0 && gy < Q.H && gx >= 0 && gx < Q.W;
            pre[k] = Q.img[(size_t)min(max(gy, 0), Q.H - 1) * Q.W + min(max(gx, 0), Q.W - 1)];   // clamped: always a valid address
            pre_mask |= (ok ? 1u : 0u) << k;
        }
    };
    int tile_id = (int)blockIdx.x < a.total_tiles ? res8_tile_of(a, blockIdx.x) : 0;
    if ((int)blockIdx.x < a.total_tiles) image_load(tile_id, 0);

    for (int k = blockIdx.x; k < a.total_tiles; k += gridDim.x) {
        const bool has_next = k + (int)gridDim.x < a.total_tiles;
        const int next_id = has_next ? res8_tile_of(a, k + gridDim.x) : 0;   // requested a whole tile ahead of its use
        int pi = 0;
        pi = prob_of_tile_search(a, tile_id);
        const Res8Prob& P = a.p[pi];
        const int t = tile_id - P.tile_begin;
        const int tyb = t / P.tiles_x, txb = t - tyb * P.tiles_x;
        const int H = P.H, W = P.W;
        const int fx0 = txb * R8_OW - 4;
#pragma unroll 1
        for (int pass = 0; pass < R8_NP; ++pass) {
            const int fy0 = (tyb * R8_NP + pass) * R8_OH - 4;        // image coordinates of frame (0,0)
            if (fy0 + 4 >= H) break;                                 // no output rows left in this unit
            const bool more_passes = pass + 1 < R8_NP && fy0 + 4 + R8_OH < H;
            const bool first = pass == 0;
            // a frame that lies inside the image needs no zero masks (scalar condition)
            const bool interior = fy0 >= 0 && fy0 + R8_FH <= H && fx0 >= 0 && fx0 + R8_PITCH <= W;
            __syncthreads();                                 // previous pass / tile finished with all LDS buffers
            if (!first) {
                // rows carried over from the pass above (frame rows shift by OH = 16):
                //   t  rows 20..22 -> 4..6,   r0 rows 20,21 -> 4,5,   r1 rows 19,20 -> 3,4
                constexpr int ROWV = R8_PITCH * 2;                   // f32x4 per row
                for (int i = tid; i < 7 * ROWV; i += R8_THREADS) {
                    const int r = i / ROWV, c = i - r * ROWV;
                    f32x4* base = reinterpret_cast<f32x4*>(r < 3 ? T : (r < 5 ? R0 : R1));
                    const int src = r < 3 ? 19 + r : (r < 5 ? 18 + (r - 3) : 16 + (r - 5));
                    const int dst = r < 3 ? 3 + r : (r < 5 ? 2 + (r - 3) : (r - 5));
                    base[dst * ROWV + c] = base[src * ROWV + c];
                }
            }
            // ---- image tile: frame rows 0..23, frame columns -2..73 (zero outside the image = SAME padding) ----
#pragma unroll
            for (int q = 0; q < NPRE; ++q) {
                const int i = tid + q * R8_THREADS;
                if (i < R8_FH * R8_IMGP) IMG[i] = ((pre_mask >> q) & 1u) ? (pre[q] - pre_mean) * pre_inv : 0.f;
            }
            __syncthreads();
            // ---- t = conv1(image) (identity activation): frame rows 1..22 (first pass) or the 16 new rows 7..22 ----
            const int t_lo = first ? 0 : 6 * R8_PITCH;
            for (int i = t_lo + tid; i < 22 * R8_PITCH; i += R8_THREADS) {
                const int r = i / R8_PITCH, c = i - r * R8_PITCH;   // frame row r+1, frame column c
                const int gy = fy0 + r + 1, gx = fx0 + c;
                float acc[8];
#pragma unroll
                for (int o = 0; o < 8; ++o) acc[o] = w1s[72 + o];
#pragma unroll
                for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                    for (int kx = 0; kx < 3; ++kx) {
                        const float v = IMG[(r + ky) * R8_IMGP + c + kx + 1];       // frame (r+1+ky-1, c+kx-1) -> IMG col +2
#pragma unroll
                        for (int o = 0; o < 8; ++o) acc[o] = fmaf(v, w1s[(ky * 3 + kx) * 8 + o], acc[o]);
                    }
                const bool ok = gy >= 0 && gy < H && gx >= 0 && gx < W;
                f32x4 lo = ok ? f32x4{acc[0], acc[1], acc[2], acc[3]} : f32x4{0.f, 0.f, 0.f, 0.f};
                f32x4 hi = ok ? f32x4{acc[4], acc[5], acc[6], acc[7]} : f32x4{0.f, 0.f, 0.f, 0.f};
                *reinterpret_cast<f32x4*>(T + r * R8_PITCH * 8 + r8v_px(c, 0)) = lo;
                *reinterpret_cast<f32x4*>(T + r * R8_PITCH * 8 + r8v_px(c, 1)) = hi;
            }
            __syncthreads();
            res8v_stage<true, false, false, ACT>(sm, T, 1, R0, 2, first ? 2 : 6, first ? 20 : 16, 2, wr, a.br, tid, poff, interior, fy0, fx0, H, W, nullptr, 0, nullptr, nullptr);
            __syncthreads();
            res8v_stage<false, false, false, ACT>(sm, R0, 2, R1, 3, first ? 3 : 5, first ? 18 : 16, 3, wr + R8V_FILTER, a.br + 8, tid, poff, interior, fy0, fx0, H, W, nullptr, 0, nullptr, nullptr);
            __syncthreads();
            if (more_passes) image_load(tile_id, pass + 1);
            else if (has_next) image_load(next_id, 0);
            res8v_stage<false, true, true, ACT>(sm, R1, 3, nullptr, 4, 4, 16, 4, wr + 2 * R8V_FILTER, a.br + 16, tid, poff, interior, fy0, fx0, H, W, T, 1, P.out, P.pool);
        }
        tile_id = next_id;
    }
}

// UP block of level 0 (ARU_v1.py:262-281): t = conv1(concat[skip, deconv]) ; 3 x convR ; + t ; ReLU.
// The 16-channel concatenation is consumed as two 8-channel passes through one LDS input tile (skip, then the
// deconvolution output) that accumulate into the same registers; afterwards that tile buffer holds r1.
template <int ACT = 0>
__global__ __launch_bounds__(R8_THREADS) __attribute__((amdgpu_waves_per_eu(2, 2))) void res8v_up_kernel(const Res8Args a) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const float* wr = reinterpret_cast<const float*>(a.wr);
    float* Pb = sm;                                          // frame rows 0..23  [24][72][8]  (later r1: rows 3..20)
    float* T = Pb + R8_FH * R8_PITCH * 8;                    // frame rows 1..22  [22][72][8]
    float* R0 = T + 22 * R8_PITCH * 8;                       // frame rows 2..21  [20][72][8]
    float* R1K = R0 + 20 * R8_PITCH * 8;                     // two r1 rows parked between passes [2][72][8]
    char* smb = reinterpret_cast<char*>(sm);
    const int tid = threadIdx.x;
#ifdef ASEP_R8_TIMELINE
    int tl_n = 0;
    const int lane = tid & 63, wave = tid >> 6;
    if (blockIdx.x == 0 && threadIdx.x == 0) { r8_clk[0] = clock64(); r8_clk[1] = wall_clock64(); }
#endif
    int poff[7][2];
    r8v_pixel_offsets(tid, poff);
    // halo tile loader: thread -> (row group rg = tid / 144, slot cs = tid % 144 = pixel column x 2 halves), rows rg + 3k
    // (432 of the 512 threads load, 8 x 16 B each)
    constexpr int ROWV = R8_PITCH * 2;                       // f32x4 per LDS row
    constexpr int NPF = R8_FH / 3;
    f32x4 pf[NPF];
    const int rg = tid / ROWV, cs = tid - rg * ROWV;
    const int pf_dst = (rg * R8_PITCH * 8 + r8v_px(cs >> 1, cs & 1)) * 4;      // byte offset of the thread's piece in tile row rg
    // carried: only frame rows 6..23 are needed (conv1 then reads rows 6..23 only), i.e. k >= 2.  ONE code path for
    // both kinds of frame (a run-time predicate, not two call sites): with two, the register allocator gave the eight
    // destinations different registers per path, and the wait-count pass then made the following stage wait for the
    // loads just issued (a vmcnt(0) in front of its first instruction: 3 k cycles per pass)
    auto tile_load = [&](const float* __restrict__ g, int H_, int W_, int qy0, int qx0, bool carried) {   // 8-channel halo tile -> registers
        const int gx = qx0 + (cs >> 1);
        const bool okc = rg < 3 && gx >= 0 && gx < W_;
        const float* __restrict__ gp = g + ((ptrdiff_t)gx * 8 + (cs & 1) * 4);
#pragma unroll
        for (int k = 0; k < NPF; ++k) {
            const int gy = qy0 + rg + 3 * k;
            f32x4 v = f32x4{0.f, 0.f, 0.f, 0.f};
            if (okc && gy >= 0 && gy < H_ && !(k < 2 && carried)) v = *reinterpret_cast<const f32x4*>(gp + (ptrdiff_t)gy * W_ * 8);
            pf[k] = v;
        }
    };
    auto tile_store = [&](bool carried) {                    // registers -> tile buffer
        if (rg < 3) {
#pragma unroll
            for (int k = 0; k < NPF; ++k) {
                if (k < 2 && carried) continue;
                *reinterpret_cast<f32x4*>(smb + pf_dst + 3 * k * R8_PITCH * 32) = pf[k];
            }
        }
    };
    int tile_id = (int)blockIdx.x < a.total_tiles ? res8_tile_of(a, blockIdx.x) : 0;
    if ((int)blockIdx.x < a.total_tiles) {
        const int first_id = tile_id;
        int qi = 0;
        while (qi + 1 < a.nprob && first_id >= a.p[qi + 1].tile_begin) ++qi;
        const Res8Prob& Q = a.p[qi];
        const int tq = first_id - Q.tile_begin;
        const int qyb = tq / Q.tiles_x, qxb = tq - qyb * Q.tiles_x;
        tile_load(Q.img, Q.H, Q.W, qyb * R8_NP * R8_OH - 4, qxb * R8_OW - 4, false);
    }
    for (int k = blockIdx.x; k < a.total_tiles; k += gridDim.x) {
        const bool has_next = k + (int)gridDim.x < a.total_tiles;
        const int next_id = has_next ? res8_tile_of(a, k + gridDim.x) : 0;   // requested a whole tile ahead of its use
        int pi = 0;
        pi = prob_of_tile_search(a, tile_id);
        const Res8Prob& P = a.p[pi];
        const int t = tile_id - P.tile_begin;
        const int tyb = t / P.tiles_x, txb = t - tyb * P.tiles_x;
        const int H = P.H, W = P.W;
        const int fx0 = txb * R8_OW - 4;
#pragma unroll 1
        for (int pass = 0; pass < R8_NP; ++pass) {
            const int fy0 = (tyb * R8_NP + pass) * R8_OH - 4;
            if (fy0 + 4 >= H) break;                         // no output rows left in this unit
            const bool more_passes = pass + 1 < R8_NP && fy0 + 4 + R8_OH < H;
            const bool first = pass == 0;
            // a frame that lies inside the image needs no zero masks (scalar condition)
            const bool interior = fy0 >= 0 && fy0 + R8_FH <= H && fx0 >= 0 && fx0 + R8_PITCH <= W;
            // conv1 covers frame rows 1..22 in the first pass and only the 16 new rows 7..22 afterwards
            const int c1_row = first ? 1 : 7, c1_rows = first ? 22 : 16;
            R8_MARK();   // 0 pass start
            const int r0 = tid >> 5;
            const bool has2 = r0 + 16 < c1_rows;              // first pass: threads of rows 0..5 own a second row (16..21)
            // conv1 output pixels: frame columns 1 + 2p, 2 + 2p; window = tile pixels 2p .. 2p + 3 of rows c1_row + r0 - 1 ..
            r8v_lds wa[4][2];
            r8v_window(wa, sm, poff, 0, (c1_row + r0 - 1) * R8_PITCH * 32);
            // t accumulators: [row slot][pixel][channel pair]; conv1's bias is the initial value
            f32x2 ta0[4], ta1[4], tb0[4], tb1[4];
            {
                r8v_cptr bl = (r8v_cptr)a.b1;
#pragma unroll
                for (int q = 0; q < 4; ++q) { ta0[q] = f32x2{bl[2 * q], bl[2 * q + 1]}; ta1[q] = ta0[q]; tb0[q] = ta0[q]; tb1[q] = ta0[q]; }
            }
            // ---- conv1: skip half, then deconv half, out of the same tile buffer ----
#pragma unroll 1
            for (int half = 0; half < 2; ++half) {
                __syncthreads();                             // the tile buffer is free (previous pass / tile / half done)
                R8_MARK();   // 1 / 4 barrier passed
                if (half == 0 && !first) {
                    // rows carried over from the pass above: t rows 20..22 -> 4..6, r0 rows 20,21 -> 4,5
                    // (r1 rows 19,20 were parked in R1K during the previous pass' last stage)
                    for (int i = tid; i < 5 * ROWV; i += R8_THREADS) {
                        const int r = i / ROWV, c = i - r * ROWV;
                        f32x4* base = reinterpret_cast<f32x4*>(r < 3 ? T : R0);
                        const int srow = r < 3 ? 19 + r : 18 + (r - 3), drow = r < 3 ? 3 + r : 2 + (r - 3);
                        base[drow * ROWV + c] = base[srow * ROWV + c];
                    }
                }
                tile_store(!first);
                if (half == 0) tile_load(P.in1, H, W, fy0, fx0, !first);  // the deconv half flies while the skip half is multiplied
                __syncthreads();
                R8_MARK();   // 2 / 5 tile in LDS
                r8v_cptr wl = (r8v_cptr)(a.w1 + half * R8V_FILTER);
                asm volatile("" : "+s"(wl));
                r8v_conv<false>(wa, wl, ta0, ta1);
                if (has2) {
                    r8v_lds wb[4][2];
#pragma unroll
                    for (int i = 0; i < 4; ++i) { wb[i][0] = wa[i][0] + 16 * R8_PITCH * 32; wb[i][1] = wa[i][1] + 16 * R8_PITCH * 32; }
                    asm volatile("" : "+s"(wl));
                    r8v_conv<false>(wb, wl, tb0, tb1);
                }
                R8_MARK();   // 3 / 6 conv1 half done
            }
            // ---- write raw t (identity activation), zero outside the image: T row (row - 1) = the window's first row ----
            {
                constexpr int DT = R8_FH * R8_PITCH * 32;     // T - Pb in bytes
                const int gx = fx0 + 1 + 2 * (tid & 31);
                const f32x4 z = f32x4{0.f, 0.f, 0.f, 0.f};
                {
                    f32x4 v0l = r8v_lo(ta0), v0h = r8v_hi(ta0), v1l = r8v_lo(ta1), v1h = r8v_hi(ta1);
                    if (!interior) {
                        const int gy = fy0 + c1_row + r0;
                        const bool oky = gy >= 0 && gy < H;
                        const bool ok0 = oky && gx >= 0 && gx < W, ok1 = oky && gx + 1 >= 0 && gx + 1 < W;
                        v0l = ok0 ? v0l : z; v0h = ok0 ? v0h : z; v1l = ok1 ? v1l : z; v1h = ok1 ? v1h : z;
                    }
                    r8v_st(wa[1][0] + DT, v0l);
                    r8v_st(wa[1][1] + DT, v0h);
                    r8v_st(wa[2][0] + DT, v1l);
                    r8v_st(wa[2][1] + DT, v1h);
                }
                if (has2) {
                    f32x4 v0l = r8v_lo(tb0), v0h = r8v_hi(tb0), v1l = r8v_lo(tb1), v1h = r8v_hi(tb1);
                    if (!interior) {
                        const int gy = fy0 + c1_row + r0 + 16;
                        const bool oky = gy >= 0 && gy < H;
                        const bool ok0 = oky && gx >= 0 && gx < W, ok1 = oky && gx + 1 >= 0 && gx + 1 < W;
                        v0l = ok0 ? v0l : z; v0h = ok0 ? v0h : z; v1l = ok1 ? v1l : z; v1h = ok1 ? v1h : z;
                    }
                    constexpr int D2 = DT + 16 * R8_PITCH * 32;
                    r8v_st(wa[1][0] + D2, v0l);
                    r8v_st(wa[1][1] + D2, v0h);
                    r8v_st(wa[2][0] + D2, v1l);
                    r8v_st(wa[2][1] + D2, v1h);
                }
            }
            R8_MARK();   // 7 t written
            __syncthreads();
            R8_MARK();   // 8
            res8v_stage<true, false, false, ACT>(sm, T, 1, R0, 2, first ? 2 : 6, first ? 20 : 16, 2, wr, a.br, tid, poff, interior, fy0, fx0, H, W, nullptr, 0, nullptr, nullptr);
            R8_MARK();   // 9 stage0 done
            __syncthreads();
            R8_MARK();   // 10
            res8v_stage<false, false, false, ACT>(sm, R0, 2, Pb, 3, first ? 3 : 5, first ? 18 : 16, 3, wr + R8V_FILTER, a.br + 8, tid, poff, interior, fy0, fx0, H, W, nullptr, 0, nullptr, nullptr);
            if (!first) {
                // r1 rows 3,4 of this frame = rows 19,20 of the previous one (the tile buffer is free of conv1 readers here)
                for (int i = tid; i < 2 * ROWV; i += R8_THREADS)
                    reinterpret_cast<f32x4*>(Pb)[i] = reinterpret_cast<const f32x4*>(R1K)[i];
            }
            R8_MARK();   // 11 stage1 done
            __syncthreads();
            R8_MARK();   // 12
            if (more_passes) {
                // park r1 rows 19,20 (tile-buffer rows 16,17) for the next pass: only read, like the stage below
                for (int i = tid; i < 2 * ROWV; i += R8_THREADS)
                    reinterpret_cast<f32x4*>(R1K)[i] = reinterpret_cast<const f32x4*>(Pb)[16 * ROWV + i];
            }
            if (more_passes || has_next) {                   // the next frame's skip half flies under the last stage
                const float* g = P.img;
                int H_ = H, W_ = W, qy0 = fy0 + R8_OH, qx0 = fx0;
                if (!more_passes) {
                    int qi = 0;
                    while (qi + 1 < a.nprob && next_id >= a.p[qi + 1].tile_begin) ++qi;
                    const Res8Prob& Q = a.p[qi];
                    const int tq = next_id - Q.tile_begin;
                    const int qyb = tq / Q.tiles_x, qxb = tq - qyb * Q.tiles_x;
                    g = Q.img; H_ = Q.H; W_ = Q.W; qy0 = qyb * R8_NP * R8_OH - 4; qx0 = qxb * R8_OW - 4;
                }
                tile_load(g, H_, W_, qy0, qx0, more_passes);
            }
            R8_MARK();   // 13 prefetch issued
            res8v_stage<false, true, false, ACT>(sm, Pb, 3, nullptr, 4, 4, 16, 4, wr + 2 * R8V_FILTER, a.br + 16, tid, poff, interior, fy0, fx0, H, W, T, 1, P.out, nullptr);
            R8_MARK();   // 14 stage2 done
        }
        tile_id = next_id;
    }
#ifdef ASEP_R8_TIMELINE
    if (blockIdx.x == 0 && threadIdx.x == 0) { r8_clk[2] = clock64(); r8_clk[3] = wall_clock64(); }
#endif
}

// ------------------------------------------------------------------------------------------------
// Attention CNN head on the vector ALU (fp32; ARU_v1.py:173-175): 4x4 conv 1->12 + ReLU + 2x2 max pool, fused like
// att_head_kernel (aru_kernels.h).  There the MFMA form (K = 16 taps, M = 12 of 16) issued 13 vector instructions per MFMA
// for the ReLU, the pool across lanes and the addresses: MFMA 46 % + VALU 53 % of the SIMD cycles, and on gfx950 the two
// share one fp32 datapath (profiles/r2p/instruction_mix.json).  Here a thread owns one POOLED pixel: its 2 x 2 conv
// outputs x 12 channels = 24 packed accumulators, 96 v_pk_fma_f32 per tap row with the filter row in scalar registers and
// the two image rows of the step in registers; the pool is 2 x v_max3 per channel inside the thread, no cross-lane traffic.
// ------------------------------------------------------------------------------------------------
// OUTBF (native bf16 path): the pooled pixel is written as 16 bf16 = 12 channels + 4 zeros (32 bytes), the 16-channel plane the
// next conv's MFMA K chunks expect (bf16_kernels.h)
template <bool OUTBF = false>
__global__ __launch_bounds__(256) void att_headv_kernel(const AttHeadArgs a) {
    constexpr int LH = ATT_TH + 3, LW = ATT_TW + 4;          // SAME for 4x4: 1 before, 2 after (+1 col of slack)
    __shared__ __attribute__((aligned(16))) float img[LH * LW];
    const int tid = threadIdx.x;
    int pi = 0;
    pi = prob_of_tile(a, (int)blockIdx.x);
    const C1Prob& P = a.p[pi];
    const int tile = blockIdx.x - P.tile_begin;
    const int ty = tile / P.tiles_x, tx = tile - ty * P.tiles_x;
    const int x0 = tx * ATT_TW, y0 = ty * ATT_TH;
    const int H = P.H, W = P.W;
    float mean = 0.f, inv = 1.f;
    if (P.stats) { mean = P.stats[0]; inv = P.stats[1]; }
    {   // requests first, LDS writes afterwards (see deconv8v_kernel)
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
            if (i >= LH * LW) break;
            const int r = i / LW, c = i - r * LW;
            const int gy = y0 - 1 + r, gx = x0 - 1 + c;
            img[i] = (gy >= 0 && gy < H && gx >= 0 && gx < W) ? (st[k] - mean) * inv : 0.f;
        }
    }
    __syncthreads();
    const bool interior = y0 + ATT_TH <= H && x0 + ATT_TW <= W;   // every 2x2 window of the tile is complete
    const unsigned Wp = (unsigned)(W + 1) >> 1;
    const int lx = 2 * (tid & 31);
#pragma unroll 1
    for (int pr = tid >> 5; pr < ATT_TH / 2; pr += 8) {
        const int ly = 2 * pr;
        r8v_cptr bl = (r8v_cptr)a.bias;
        f32x2 acc[4][6];                                       // [2x2 pixel][channel pair]; the bias is the initial value
#pragma unroll
        for (int q = 0; q < 6; ++q) { acc[0][q] = f32x2{bl[2 * q], bl[2 * q + 1]}; acc[1][q] = acc[0][q]; acc[2][q] = acc[0][q]; acc[3][q] = acc[0][q]; }
        // taps of output (ly + dy, lx + dx) start at img[ly + dy][lx + dx]: columns lx .. lx + 4 of rows ly .. ly + 4
        const f32x2* p = reinterpret_cast<const f32x2*>(img + ly * LW + lx);
        f32x2 rA[3], rB[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) rA[c] = p[c];
        r8v_cptr wl = (r8v_cptr)a.w;
#pragma unroll
        for (int ky = 0; ky < 4; ++ky) {
            asm volatile("" : "+s"(wl));                       // this tap row's 48 scalars are loaded here, not hoisted (and spilled)
            float w[48];
#pragma unroll
            for (int k = 0; k < 48; ++k) w[k] = wl[ky * 48 + k];
#pragma unroll
            for (int c = 0; c < 3; ++c) rB[c] = p[(ky + 1) * (LW / 2) + c];
#pragma unroll
            for (int kx = 0; kx < 4; ++kx)
#pragma unroll
                for (int q = 0; q < 6; ++q) {
                    const f32x2 wv = f32x2{w[kx * 12 + 2 * q], w[kx * 12 + 2 * q + 1]};
                    // pixel (0,0) reads column kx, (0,1) column kx + 1 of row A; (1,0) / (1,1) the same of row B
                    if (kx & 1) { r8v_fma<1>(acc[0][q], rA[kx >> 1], wv); r8v_fma<0>(acc[1][q], rA[(kx + 1) >> 1], wv); r8v_fma<1>(acc[2][q], rB[kx >> 1], wv); r8v_fma<0>(acc[3][q], rB[(kx + 1) >> 1], wv); }
                    else { r8v_fma<0>(acc[0][q], rA[kx >> 1], wv); r8v_fma<1>(acc[1][q], rA[kx >> 1], wv); r8v_fma<0>(acc[2][q], rB[kx >> 1], wv); r8v_fma<1>(acc[3][q], rB[kx >> 1], wv); }
                }
#pragma unroll
            for (int c = 0; c < 3; ++c) rA[c] = rB[c];
        }
        asm volatile("" : "+v"(acc[0][0]), "+v"(acc[1][0]), "+v"(acc[2][0]), "+v"(acc[3][0]));   // the FMAs stay here (see r8v_conv_direct)
        const int gy = y0 + ly, gx = x0 + lx;
        if (!interior) {
            // relu(max over the window's pixels inside the image) = max(..., 0): a pixel outside contributes 0 (graph variants: -inf,
            // their activation of the maximum over the pixels inside follows below)
            const bool okx = gx + 1 < W, oky = gy + 1 < H;
            const float zv = a.act ? -INFINITY : 0.f;
            const f32x2 z = f32x2{zv, zv};
#pragma unroll
            for (int q = 0; q < 6; ++q) {
                acc[1][q] = okx ? acc[1][q] : z;
                acc[2][q] = oky ? acc[2][q] : z;
                acc[3][q] = (okx && oky) ? acc[3][q] : z;
            }
        }
        // max(a, b, c, d, 0) on the bit patterns: two v_max3_i32 per channel (like relu4i; a negative float is a negative int)
        float m[12];
        auto pool = [](float a0, float a1, float a2, float a3) {
            const int i = max(max(max(__float_as_int(a0), __float_as_int(a1)), __float_as_int(a2)), max(__float_as_int(a3), 0));
            return __int_as_float(i);
        };
        if (a.act) {                                           // elu / leaky: act(max(window)) = max(act(window)), fmaxf on the floats
#pragma unroll
            for (int q = 0; q < 6; ++q) {
                m[2 * q] = act1(fmaxf(fmaxf(acc[0][q].x, acc[1][q].x), fmaxf(acc[2][q].x, acc[3][q].x)), a.act);
                m[2 * q + 1] = act1(fmaxf(fmaxf(acc[0][q].y, acc[1][q].y), fmaxf(acc[2][q].y, acc[3][q].y)), a.act);
            }
        } else {
#pragma unroll
            for (int q = 0; q < 6; ++q) {
                m[2 * q] = pool(acc[0][q].x, acc[1][q].x, acc[2][q].x, acc[3][q].x);
                m[2 * q + 1] = pool(acc[0][q].y, acc[1][q].y, acc[2][q].y, acc[3][q].y);
            }
        }
        if (gy < H && gx < W) {
            if constexpr (OUTBF) {
                typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
                unsigned short* o = reinterpret_cast<unsigned short*>(P.out) + ((unsigned)(gy >> 1) * Wp + (unsigned)(gx >> 1)) * 16u;
                *reinterpret_cast<u32x4_t*>(o) = u32x4_t{bf16x2_of(m[0], m[1]), bf16x2_of(m[2], m[3]), bf16x2_of(m[4], m[5]), bf16x2_of(m[6], m[7])};
                *reinterpret_cast<u32x4_t*>(o + 8) = u32x4_t{bf16x2_of(m[8], m[9]), bf16x2_of(m[10], m[11]), 0u, 0u};
            } else {
                float* o = P.out + ((unsigned)(gy >> 1) * Wp + (unsigned)(gx >> 1)) * 12u;
                *reinterpret_cast<f32x4*>(o) = f32x4{m[0], m[1], m[2], m[3]};
                *reinterpret_cast<f32x4*>(o + 4) = f32x4{m[4], m[5], m[6], m[7]};
                *reinterpret_cast<f32x4*>(o + 8) = f32x4{m[8], m[9], m[10], m[11]};
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Level-0 deconvolution on the vector ALU (fp32): conv2d_transpose 3x3, stride 2, SAME, 16 -> 8 channels (layers.py:362;
// index algebra as deconv_mfma_kernel).  With 8 output channels the MFMA form fills half of its 16 rows (MFMA 47 % + VALU
// 15 % of the SIMD cycles for 23 % useful work); here a thread owns one input position q = its 2 x 2 output pixels x 8
// channels = 16 packed accumulators, and takes the 9 taps x 16 input channels from scalar registers: 576 v_pk_fma_f32 per
// thread, the four input pixels (q, q - 1 in x / y / both) from an LDS tile.
// ------------------------------------------------------------------------------------------------
constexpr int DCV_T = 16;                                  // 16 x 16 input positions per block, one per thread

__global__ __launch_bounds__(256) void deconv8v_kernel(const ConvArgs a) {
    constexpr int T = DCV_T, L = T + 1;                      // one halo row / column before the tile (o = q - 1)
    // input tile (17 x 17 pixels x 16 channels), later the output tile (32 x 32 pixels x 8 channels = 32 KB)
    __shared__ __attribute__((aligned(16))) float lds[2 * T * 2 * T * 8];
    static_assert(L * L * 16 <= 2 * T * 2 * T * 8, "input tile must fit");
    const int tid = threadIdx.x;
    const int bid = sched_tile(a.xm);
    if (bid < 0) return;
    int pi = 0;
    pi = prob_of_tile(a, bid);
    const ConvProb& P = a.p[pi];
    const int tile = bid - P.tile_begin;
    const int ty = tile / P.tiles_x, tx = tile - ty * P.tiles_x;
    const int qx0 = tx * T, qy0 = ty * T;
    const int H = P.H, W = P.W;
    const int relu_lim = a.relu_in ? 0 : (int)0x80000000;
    // tile -> LDS: (pixel, channel quad) slots, zero outside the image
    // (the four 16-byte quads of a 64-byte pixel record are permuted by the pixel's column, like r8_px: the lanes of a read
    // or write are consecutive pixels taking the SAME quad, 64 bytes apart = two bank groups without the permutation)
    // all of a thread's slots are requested (from clamped, always valid addresses) before the first is written to LDS: as
    // one load-store loop the five round trips to memory were exposed one after the other
    constexpr int NSL = (L * L * 4 + 255) / 256;
    f32x4 st[NSL];
#pragma unroll
    for (int i = 0; i < NSL; ++i) {
        const int idx = min(tid + i * 256, L * L * 4 - 1);
        const int pix = idx >> 2, sub = idx & 3;
        const int ly = pix / L, lx = pix - ly * L;
        const int gy = min(max(qy0 - 1 + ly, 0), H - 1), gx = min(max(qx0 - 1 + lx, 0), W - 1);
        st[i] = *reinterpret_cast<const f32x4*>(P.in0 + ((size_t)gy * W + gx) * 16 + sub * 4);
    }
#pragma unroll
    for (int i = 0; i < NSL; ++i) {
        const int idx = tid + i * 256;
        if (idx >= L * L * 4) break;
        const int pix = idx >> 2, sub = idx & 3;
        const int ly = pix / L, lx = pix - ly * L;
        const int gy = qy0 - 1 + ly, gx = qx0 - 1 + lx;
        const f32x4 v = (gy >= 0 && gy < H && gx >= 0 && gx < W) ? st[i] : f32x4{0.f, 0.f, 0.f, 0.f};
        *reinterpret_cast<f32x4*>(lds + pix * 16 + ((sub ^ ((lx >> 1) & 3)) << 2)) = imax4(v, relu_lim);
    }
    __syncthreads();
    const int qx = tid & 15, qy = tid >> 4;
    // the four input pixels of this position: shift 0 (q), 1 (x - 1), 2 (y - 1), 3 (both); 16 channels each
    f32x4 d[4][4];
#pragma unroll
    for (int sh = 0; sh < 4; ++sh) {
        const int lx = qx + 1 - (sh & 1);
#pragma unroll
        for (int c4 = 0; c4 < 4; ++c4)
            d[sh][c4] = *reinterpret_cast<const f32x4*>(lds + ((qy + 1 - (sh >> 1)) * L + lx) * 16 + ((c4 ^ ((lx >> 1) & 3)) << 2));
    }
    r8v_cptr bl = (r8v_cptr)a.bias;
    f32x2 acc[4][4];                                         // [parity class (ry, rx)][output channel pair]; bias = initial value
#pragma unroll
    for (int q = 0; q < 4; ++q) { acc[0][q] = f32x2{bl[2 * q], bl[2 * q + 1]}; acc[1][q] = acc[0][q]; acc[2][q] = acc[0][q]; acc[3][q] = acc[0][q]; }
    // weights [tap][ci][co] through a 2 x 32-SGPR double buffer, one wait per group (see r8v_conv_direct)
    r8v_cptr wl = (r8v_cptr) reinterpret_cast<const float*>(a.wpk);
    asm volatile("" : "+s"(wl));
    float wc[32], wn[32];
#pragma unroll
    for (int k = 0; k < 32; ++k) wc[k] = wl[k];
#pragma unroll
    for (int g = 0; g < 36; ++g) {
        const int tap = g >> 2, c4 = g & 3;
        const int ky = tap / 3, kx = tap % 3;
        const int cls = (ky & 1) * 2 + (kx & 1), sh = (ky == 2 ? 2 : 0) + (kx == 2 ? 1 : 0);
        asm volatile("" :: "s"(wc[0]), "s"(wc[16]));
        __builtin_amdgcn_sched_barrier(0);
        if (g + 1 < 36) {
#pragma unroll
            for (int k = 0; k < 32; ++k) wn[k] = wl[(g + 1) * 32 + k];
        }
        __builtin_amdgcn_sched_barrier(0);
        const f32x4 v = d[sh][c4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            r8v_fma<0>(acc[cls][q], f32x2{v.x, v.y}, f32x2{wc[2 * q], wc[2 * q + 1]});
            r8v_fma<1>(acc[cls][q], f32x2{v.x, v.y}, f32x2{wc[8 + 2 * q], wc[8 + 2 * q + 1]});
            r8v_fma<0>(acc[cls][q], f32x2{v.z, v.w}, f32x2{wc[16 + 2 * q], wc[16 + 2 * q + 1]});
            r8v_fma<1>(acc[cls][q], f32x2{v.z, v.w}, f32x2{wc[24 + 2 * q], wc[24 + 2 * q + 1]});
        }
#pragma unroll
        for (int k = 0; k < 32; ++k) wc[k] = wn[k];
    }
    // through an LDS output tile, so that one store instruction of a wave writes 1 KB of ONE output row (a thread's own
    // pixels are 64 contiguous bytes per row: stored directly, every instruction would scatter 16-byte pieces at a 64-byte
    // stride over sixteen cache lines)
    const int relu_o = a.relu_out ? 0 : (int)0x80000000;
    __syncthreads();                                         // every thread has its input pixels in registers (read above)
    // output tile: a thread's two pixels of a row are one 64-byte record (four 16-byte pieces); piece k of record qx sits in
    // slot k ^ ((qx >> 1) & 3), undone by the row-wise reader below
#pragma unroll
    for (int cls = 0; cls < 4; ++cls) {
        float* o = lds + ((2 * qy + (cls >> 1)) * (2 * T) + 2 * qx) * 8;
        const int k = (cls & 1) * 2, sw = (qx >> 1) & 3;
        f32x4 vlo = imax4(r8v_lo(acc[cls]), relu_o), vhi = imax4(r8v_hi(acc[cls]), relu_o);
        if (a.act) { vlo = act4(vlo, a.act); vhi = act4(vhi, a.act); }      // graph variants (elu / leaky): uniform branch, not taken by ReLU nets
        *reinterpret_cast<f32x4*>(o + ((k ^ sw) << 2)) = vlo;
        *reinterpret_cast<f32x4*>(o + (((k + 1) ^ sw) << 2)) = vhi;
    }
    __syncthreads();
    const int oy0 = 2 * qy0 - P.pbh, ox0 = 2 * qx0 - P.pbw;   // image coordinates of the output tile's origin
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int idx = tid + i * 256;                       // row = idx / 64, 16-byte piece of the row = idx % 64
        const int row = idx >> 6, piece = idx & 63;
        const int y = oy0 + row, x = ox0 + (piece >> 1);
        // rows / columns of positions beyond the input (q >= H, W) are not outputs of this layer
        if (y >= 0 && y < P.Ho && x >= 0 && x < P.Wo && qy0 + (row >> 1) < H && qx0 + (piece >> 2) < W)
            *reinterpret_cast<f32x4*>(P.out + ((size_t)y * P.Wo + x) * 8 + (piece & 1) * 4) =
                *reinterpret_cast<const f32x4*>(lds + (idx & ~3) * 4 + (((piece & 3) ^ ((piece >> 3) & 3)) << 2));
    }
}

// ------------------------------------------------------------------------------------------------
// Last layer of the attention CNN on the vector ALU (fp32): 4x4 conv, SAME, 32 -> 1 channel (ARU_v1.py:183).  On the MFMA
// path a single output channel occupies one of 16 rows (5 TFLOP/s, 57 us per page); here a thread owns one output pixel,
// sums channel PAIRS in packed accumulators (v_pk_fma_f32, the filter in scalar registers) and adds the halves at the end.
// ------------------------------------------------------------------------------------------------
constexpr int C1O_T = 16;                                  // 16 x 16 output pixels per block, one per thread

__global__ __launch_bounds__(256) void conv_c1out_kernel(const ConvArgs a) {
    constexpr int T = C1O_T, L = T + 3, CIN = 32, Q = CIN / 4;   // 4x4 SAME: pad 1 before, 2 after
    __shared__ __attribute__((aligned(16))) float lds[L * L * CIN];
    const int tid = threadIdx.x;
    const int bid = sched_tile(a.xm);
    if (bid < 0) return;
    int pi = 0;
    pi = prob_of_tile(a, bid);
    const ConvProb& P = a.p[pi];
    const int tile = bid - P.tile_begin;
    const int ty0 = tile / P.tiles_x, tx0 = tile - ty0 * P.tiles_x;
    const int x0 = tx0 * T, y0 = ty0 * T;
    const int H = P.H, W = P.W;
    const int relu_lim = a.relu_in ? 0 : (int)0x80000000;
    // halo tile: 128-byte pixel records, their eight 16-byte quads permuted by the pixel's column (a read takes the same
    // quad of consecutive pixels: without the permutation all lanes of a pass hit one bank group); requests first
    constexpr int NSL = (L * L * Q + 255) / 256;
    f32x4 st[NSL];
#pragma unroll
    for (int i = 0; i < NSL; ++i) {
        const int idx = min(tid + i * 256, L * L * Q - 1);
        const int pix = idx / Q, sub = idx - pix * Q;
        const int ly = pix / L, lx = pix - ly * L;
        const int gy = min(max(y0 - 1 + ly, 0), H - 1), gx = min(max(x0 - 1 + lx, 0), W - 1);
        st[i] = *reinterpret_cast<const f32x4*>(P.in0 + ((size_t)gy * W + gx) * CIN + sub * 4);
    }
#pragma unroll
    for (int i = 0; i < NSL; ++i) {
        const int idx = tid + i * 256;
        if (idx >= L * L * Q) break;
        const int pix = idx / Q, sub = idx - pix * Q;
        const int ly = pix / L, lx = pix - ly * L;
        const int gy = y0 - 1 + ly, gx = x0 - 1 + lx;
        const f32x4 v = (gy >= 0 && gy < H && gx >= 0 && gx < W) ? st[i] : f32x4{0.f, 0.f, 0.f, 0.f};
        *reinterpret_cast<f32x4*>(lds + pix * CIN + ((sub ^ (lx & 7)) << 2)) = imax4(v, relu_lim);
    }
    __syncthreads();
    const int tx = tid & 15, ty = tid >> 4;
    r8v_cptr wl = (r8v_cptr) reinterpret_cast<const float*>(a.wpk);   // [16 taps][32]
    f32x2 acc0 = f32x2{0.f, 0.f}, acc1 = f32x2{0.f, 0.f};
#pragma unroll
    for (int ky = 0; ky < 4; ++ky)
#pragma unroll
        for (int kx = 0; kx < 4; ++kx) {
            asm volatile("" : "+s"(wl));                      // this tap's 32 scalars are loaded here, not hoisted (and spilled)
            float w[CIN];
#pragma unroll
            for (int k = 0; k < CIN; ++k) w[k] = wl[(ky * 4 + kx) * CIN + k];
            const int lx = tx + kx;
            const float* p = lds + ((ty + ky) * L + lx) * CIN;
#pragma unroll
            for (int q = 0; q < Q; ++q) {
                const f32x4 v = *reinterpret_cast<const f32x4*>(p + ((q ^ (lx & 7)) << 2));
                asm("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(acc0) : "v"(f32x2{v.x, v.y}), "s"(f32x2{w[4 * q], w[4 * q + 1]}));
                asm("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(acc1) : "v"(f32x2{v.z, v.w}), "s"(f32x2{w[4 * q + 2], w[4 * q + 3]}));
            }
        }
    const int x = x0 + tx, y = y0 + ty;
    if (x >= W || y >= H) return;
    float s = ((acc0.x + acc0.y) + (acc1.x + acc1.y)) + a.bias[0];
    if (a.relu_out) s = fmaxf(s, 0.f);
    else if (a.act) s = act1(s, a.act);
    P.out[(size_t)y * W + x] = s;
}

}  // namespace asep
