// res8w_kernel: the level-0 UP block of the bf16 path (conv1 over [skip, deconv] 16 -> 8, three 8 -> 8 convolutions, + t, ReLU:
// ARU_v1.py:251-292 at level 0) as a COLUMN-STRIP WALKER -- round 6, the form DESIGN section 7.2 and the round-5 review named.
//
// res8f_kernel (bf16_kernels.h) computes a 16 x 32-pixel tile per block: window from HBM -> barrier -> conv1 -> barrier -> three
// stages with a barrier each -> stores; four waves meet five times per tile, every phase has two or three pair slots per wave (a
// software pipeline that is mostly fill and drain), the 24 x 40 window is requested when the block starts (2-4 us of HBM latency that
// only the other two blocks of the CU cover), and the stacked 3 x 3 convolutions recompute 1.31 x the block's outputs in their halos.
// Its SIMDs issue 43 % of the time (317 us per 3000 x 4500 page against 130 us of MFMA + vector issue).
//
// Here ONE WAVE walks DOWN a strip of 24 output columns, two rows per iteration, and keeps a rolling window of every stage in its own
// LDS rings -- no other wave ever reads them, so there is NO barrier and no cross-wave dependency anywhere in the kernel:
//   * iteration k runs one pair slot of EVERY stage on data that earlier iterations left in the rings: conv1 on rows a, a + 1, stage 1
//     three rows behind it, stage 2 three rows behind stage 1, stage 3 three rows behind stage 2.  The four slots of an iteration are
//     independent of each other: their fragment reads are issued together, their MFMAs and epilogues interleave freely -- a steady
//     state with no fill and drain except at the ends of a strip segment;
//   * every stage region is ONE MFMA tile wide: 30 / 28 / 26 / 24 pixels = 15 / 14 / 13 / 12 of the 16 pixel pairs of the pixel-pair
//     mapping (M = 2 pixels x 8 channels, K = one filter row = 4 window pixels x 8 channels, N = 16 pairs; bf16_kernels.h): no
//     remainder tiles, every LDS address is a per-lane constant + a wave-uniform ring offset;
//   * no vertical halo is recomputed (only 3 + 2 + 1 rows at the head and tail of a segment of ~200 rows); the horizontal one is
//     (30 + 28 + 26) / (3 x 24) = 1.17 x on the stages: 15 MFMAs per 2 x 24 output pixels = 320 per 512 (res8f: 339);
//   * the input rows arrive by LDS-DMA (global_load_lds_dwordx4: one instruction = two rows of one source plane, no registers) D row
//     pairs ahead of conv1, retired by a counted s_waitcnt vmcnt; the output rows leave as the stage-3 epilogue produces them.
// Per wave: 21 KB of LDS (input ring 8 rows x 2 planes x 32 pixels, rings of 6 rows for the three intermediate stages, 12 rows of the
// raw conv1 result for the residual add) -> seven independent waves per CU.  The accumulation order of every output value is that of
// res8f_kernel (bias as the accumulators' initial value, filter rows 0, 1, 2; conv1: (row, source) pairs in the same order): results
// are BIT-IDENTICAL to res8f_kernel's wherever both run their lean form (tests/test_aru_gpu.py).
//
// The walker covers columns [32, 32 + 24 n) x rows [16, y_end) of a page (its 4-pixel window margins inside the image); the frame
// around it -- one 32-pixel tile column left, the rest right, one 16-row tile row at the top, the rest at the bottom -- is computed by
// res8wb_kernel with the general tile function res8b_tile (zero tests, clipped stores).  Pages too small for a strip stay on res8f_kernel.
#pragma once
#include "bf16_kernels.h"

namespace asep {

#ifndef R8W_DEPTH
#define R8W_DEPTH 4
#endif
constexpr int R8W_TW = 24;             // output columns of a strip
constexpr int R8W_D = R8W_DEPTH;               // pair slots of the input ring = input row pairs in flight
constexpr int R8W_X0 = 32, R8W_Y0 = 16;   // the walker's region starts here (one border tile column / row in front of it)

struct Res8WProb {
    const bf16_t* skip;    // UP: [H,W,8]
    const bf16_t* dec;     // UP: [H,W,8] deconv output
    const float* img;      // DOWN: [H,W] fp32 image (pyramid level)
    const float* stats;    // DOWN: {mean, 1/std} or nullptr
    bf16_t* out;           // [H,W,8]
    bf16_t* pool;          // DOWN: maxpool2(out) [ceil(H/2), ceil(W/2), 8] or nullptr
    int H, W;
    int n_strips;          // strips at x0 = 32 + 24 s
    int band;              // output rows of an item (even)
    int y_end;             // the walker's rows end here (even, <= H - 4)
    int tile_begin;        // first ITEM of this problem in the launch (items: band-major, the strips of a band side by side)
};
struct Res8WArgs {
    Res8WProb p[MAXP];
    int nprob;
    const float* b1;       // conv1 bias [8]
    const u32x4* w1pf;     // UP: conv1 pair fragments for the planar input tile [ky 3][source 2][64 lanes] x 16 bytes; DOWN: ONE fragment [64 lanes]
    const u32x4* wpk;      // tail: [3 convs][ky 3][64 lanes] x 16 bytes
    const float* bias;     // tail biases [3][8]
    XcdMap xm;
};

// Ablation builds (scripts/r5_abl_build.sh R8W_ABL <bits>; WRONG RESULTS ON PURPOSE, never the product): 1 no wait for the input requests,
// 2 no output stores, 4 a vector add in place of every MFMA, 8 no input requests, 16 no LDS stores of the stage results, 32 no fragment reads in the steady form
#ifndef R8W_ABL
#define R8W_ABL 0
#endif
template <int N>
__device__ __forceinline__ void r8w_wait_vm() { if (!(R8W_ABL & 1)) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
__device__ __forceinline__ f32x4 r8w_mm(u32x4 a, u32x4 b, f32x4 c) {
    if (R8W_ABL & 4) return c + f32x4{__uint_as_float(a.x ^ b.x), __uint_as_float(a.y ^ b.y), 0.f, 0.f};
    return mfma_bf16_k32(a, b, c);
}
__device__ __forceinline__ void r8w_wait_lds() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
// LDS bank swizzle of a row's 16-byte pixel units: unit u is kept at u ^ ((u >> 3) & 1).  MI355X_MICROARCH.md, LDS: a ds_read_b128 is served in four
// NON-contiguous groups of sixteen lanes ({0-3, 12-15, 20-27}, ...) on banks (a / 4) mod 64; the fragment read of the pair-window mapping (lane (j, kk)
// takes unit 2 j + kk) is conflict-free on those groups as it stands -- and stays so under this swizzle.  A ds_write_b128 is served in eight groups of
// eight CONTIGUOUS lanes on banks (a / 4) mod 32: the stage results (lane j stores unit 2 j + e) put lanes j and j + 4 on one bank quad unswizzled; with
// bit 3 of the unit index flipping its parity they take eight different quads.  (The first cut flipped on bit 4, which is right for contiguous groups of
// sixteen and wrong for the real ones: SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE 0.50 in profiles/r6final_mix_bf16 against res8f_kernel's 0.29-0.38.)
__device__ __forceinline__ int r8w_swz(int u) { return u ^ ((u >> 3) & 1); }
// (x + d) mod N for a ring counter x in [0, N) and -N <= d < N, on the scalar unit
__device__ __forceinline__ int r8w_wrap(int x, int d, int N) {
    int v = x + d;
    v += v < 0 ? N : 0;
    v -= v >= N ? N : 0;
    return v;
}

template <bool UP>
__global__ __launch_bounds__(64, UP ? 2 : 3) void res8w_kernel(const Res8WArgs a) {
    constexpr int TW = R8W_TW, M = R8W_D;                                      // M pair slots of the input ring = M pairs in flight
    constexpr int IW = TW + 8, W0 = TW + 6, W1 = TW + 4, W2 = TW + 2;          // 32, 30, 28, 26 pixels
    constexpr int NR = 4, NT = 16;                                             // ring rows: the three stage rings, raw t
    constexpr int UN = 8;                                                      // every ring position repeats after UN iterations
    static_assert(UN % M == 0 && (2 * UN) % NR == 0 && (2 * UN) % NT == 0, "the steady form is unrolled over the rings' common period");
    constexpr int INPL = 2 * M * IW * 16;                                      // UP: bytes of one plane of the input ring
    // DOWN: the image as bfloat16, two pairs = four rows of 32 pixels (conv1 of iteration k reads pairs k, k + 1; the requests wait in registers)
    constexpr int INB = UP ? 2 * INPL : 4 * IW * 2;
    // (a fragment read of lanes j >= the tile's pairs runs up to 8 pixels past its row: into the next row or the next region, never past tc)
    constexpr int IN_OFF = 0, R0_OFF = INB, R1_OFF = R0_OFF + NR * W0 * 16, R2_OFF = R1_OFF + NR * W1 * 16,
                  TC_OFF = R2_OFF + NR * W2 * 16, TRASH = TC_OFF + NT * TW * 16, LDSB = TRASH + 16;
    static_assert(LDSB <= (UP ? 20480 : 13312), "eight (UP) / twelve (DOWN) waves per CU");
    __shared__ __attribute__((aligned(16))) unsigned char lds[LDSB];
    unsigned char* const in = lds + IN_OFF;
    unsigned char* const r0 = lds + R0_OFF;
    unsigned char* const r1 = lds + R1_OFF;
    unsigned char* const r2 = lds + R2_OFF;
    unsigned char* const tc = lds + TC_OFF;

    const int lane = threadIdx.x;
    const int j = lane & 15, kk = lane >> 4, e = kk >> 1, ch = (kk & 1) * 4;   // D layout: pixel parity e, channels ch .. ch + 3
    const int isB = kk & 1;
    const int c = 2 * j + e;                                                  // the lane's pixel column in a tile
    const int item = sched_tile(a.xm);
    if (item < 0) return;
    const int pi = prob_of_tile(a, item);
    const Res8WProb& P = a.p[pi];
    const int li = item - P.tile_begin;
    const int bi = li / P.n_strips, si = li - bi * P.n_strips;
    const int x0 = R8W_X0 + TW * si;
    const int Ya = R8W_Y0 + bi * P.band;
    const int nb = min(P.band, P.y_end - Ya);                                 // output rows of this item (even)
    const unsigned wu = (unsigned)P.W;

    // ---- filters and biases (L2 hits; once per segment of ~100 iterations) ----
    const f32x4 bias1 = *reinterpret_cast<const f32x4*>(a.b1 + ch);
    f32x4 biasw[3];
#pragma unroll
    for (int t = 0; t < 3; ++t) biasw[t] = *reinterpret_cast<const f32x4*>(a.bias + 8 * t + ch);
    constexpr int NF = UP ? 6 : 1;
    u32x4 a1[NF], w[3][3];
#pragma unroll
    for (int t = 0; t < NF; ++t) a1[t] = a.w1pf[t * 64 + lane];
#pragma unroll
    for (int s = 0; s < 3; ++s)
#pragma unroll
        for (int t = 0; t < 3; ++t) w[s][t] = a.wpk[(s * 3 + t) * 64 + lane];

    // ---- the input ring: pair p = image rows Ya - 4 + 2 p, + 1 in slot p mod M of both planes; lane -> (row of the pair, pixel).
    //      conv1 of iteration k needs pairs k, k + 1: pair k is in registers (iteration k - 1 read it), pair k + 1 is read from the ring, and
    //      when those reads have RETURNED its slot takes the request for pair k + 1 + M: ALL M slots are in flight most of the time -- the
    //      kernel's rate is bytes in flight / memory latency until the waves' own instruction chains bound it ----
    const unsigned char* __restrict__ const sk = reinterpret_cast<const unsigned char*>(P.skip);
    const unsigned char* __restrict__ const dc = reinterpret_cast<const unsigned char*>(P.dec);
    // (the ring is written lane-linear by the DMA: lane l lands in unit l & 31 of its row, so it FETCHES the pixel whose swizzled place that is)
    const unsigned goff0 = ((unsigned)(Ya - 4 + (lane >> 5)) * wu + (unsigned)(x0 - 4 + r8w_swz(lane & 31))) * 16u, gpair = 2u * wu * 16u;
    const int p_last = nb / 2 + 3;
    auto dma = [&](int p, int slot) {                        // (wave-uniform)
        const unsigned go = goff0 + (unsigned)p * gpair;
        if (R8W_ABL & 8) return;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(sk + go),
                                         (__attribute__((address_space(3))) void*)(in + slot * 2 * IW * 16), 16, 0, 0);
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(dc + go),
                                         (__attribute__((address_space(3))) void*)(in + INPL + slot * 2 * IW * 16), 16, 0, 0);
    };
    // DOWN: a pair is 64 fp32 pixels = one per lane; M pairs wait in registers (ordinary loads: the compiler counts their waits), a pair is
    // standardised, rounded to bfloat16 (the first layer reads the image as bfloat16 in every form of the bf16 engine) and stored when conv1 needs it
    const unsigned char* __restrict__ const im8 = reinterpret_cast<const unsigned char*>(P.img);
    const unsigned ioff0 = ((unsigned)(Ya - 4 + (lane >> 5)) * wu + (unsigned)(x0 - 4 + (lane & 31))) * 4u, ipair = 2u * wu * 4u;
    float ireg[M];
    float mean = 0.f, inv = 1.f;
    if constexpr (!UP) {
        if (P.stats) { mean = P.stats[0]; inv = P.stats[1]; }
    }
    auto iload = [&](int p) { return *reinterpret_cast<const float*>(im8 + (ioff0 + (unsigned)p * ipair)); };
    auto istore = [&](float v, int slot) {                   // pair -> rows 2 slot, 2 slot + 1 of the image ring
        reinterpret_cast<bf16_t*>(in)[slot * 2 * IW + lane] = (bf16_t)(pack_bf16x2((v - mean) * inv, 0.f) & 0xffffu);
    };
    if constexpr (UP) {
#pragma unroll
        for (int p = 0; p < M; ++p) dma(p, p);               // (p_last >= 4 >= M - 1)
    } else {
        const float v0 = iload(0);
#pragma unroll
        for (int p = 0; p < M; ++p) ireg[p] = iload(p + 1);  // ireg[p mod M] holds pair p + 1 ... (pair p + 1 sits in ireg[p % M])
        istore(v0, 0);
    }

    const int lcol = r8w_swz(2 * j + kk) * 16;               // the lane's window pixel kk of pair j in a source row
    auto whole = [&](u32x2 pa, u32x2 pb) {                   // lanes kk = 0 / 2 end with the whole pixel of tile A, kk = 1 / 3 with that of tile B
        const auto s0 = __builtin_amdgcn_permlane16_swap(pa.x, pb.x, false, false);
        const auto s1 = __builtin_amdgcn_permlane16_swap(pa.y, pb.y, false, false);
        return u32x4{s0[0], s1[0], s0[1], s1[1]};
    };
    auto relu_pk = [](u32x2 p) { return u32x2{relu_bf16x2(p.x), relu_bf16x2(p.y)}; };
    unsigned char* __restrict__ const outb = reinterpret_cast<unsigned char*>(P.out);
    const unsigned ooff0 = (((unsigned)(Ya - 12) * wu + (unsigned)(x0 + c)) * 8u + (unsigned)ch) * 2u, orow = wu * 16u;
    unsigned char* __restrict__ const poolb = reinterpret_cast<unsigned char*>(P.pool);
    const unsigned Wp = (unsigned)((P.W + 1) >> 1);
    const unsigned poff0 = (((unsigned)((Ya - 12) >> 1) * Wp + (unsigned)((x0 >> 1) + j)) * 8u + (unsigned)ch) * 2u, prow = Wp * 16u;
    // lanes beyond a region's pairs store into a 16-byte dump (an address select instead of a divergent branch around every store)
    const int cs = r8w_swz(c) * 16;
    const int col0 = j < W0 / 2 ? R0_OFF + cs : TRASH, col1 = j < W1 / 2 ? R1_OFF + cs : TRASH, col2 = j < W2 / 2 ? R2_OFF + cs : TRASH;
    const int colt = (c >= 3 && c < 3 + TW) ? TC_OFF + r8w_swz(c - 3) * 16 : TRASH;
    const int tcol = r8w_swz(c) * 16 + ch * 2;               // stage 3 reads the lane's half of raw t at its output column
    const bool ost = j < TW / 2;

    // ring counters of the iteration: (k + 1) mod M, 2 k mod NR, 2 k mod NT (run time in the general form, compile time in the steady one)
    int rim = 1 % M, ri4 = 0, ri16 = 0, rik = 0;            // (rik = k mod M: the register of pair k + 1, DOWN)
    // The two lower source rows of an iteration's slots are the two upper ones of the next iteration's: they stay in registers (11 instead
    // of 21 fragment reads per iteration, stage rings of four rows instead of six, and the input pair is dead when it has been read once)
    u32x4 cin[2][2] = {}, c0[2] = {}, c1r[2] = {}, c2[2] = {};
    auto iteration = [&](auto phase_c, int k) {
        constexpr int PH = decltype(phase_c)::value;         // >= 0: k mod UN of a STEADY iteration (every stage active, a request issued): every LDS
        constexpr bool ST = PH >= 0;                         // address is a per-lane constant + an immediate, the body one straight block
        const int im = ST ? (PH + 1) % M : rim, i4 = ST ? (2 * PH) % NR : ri4, i16 = ST ? (2 * PH) % NT : ri16;
        const bool do_c1 = ST || k < nb / 2 + 3, do_s1 = ST || (k >= 2 && k < nb / 2 + 4), do_s2 = ST || (k >= 4 && k < nb / 2 + 5), do_s3 = ST || k >= 6;
        const bool issue = ST || k + 1 + M <= p_last;
        // ---- fragment reads of the three stages (everything they read was written by earlier iterations; the general form reads all four rows:
        //      rows 0, 1 are still in the ring, this iteration's stores take their places BEHIND these reads) ----
        u32x4 inr[4][2] = {}, q0[4] = {}, q1[4] = {}, q2[4] = {};   // (an idle stage's rows are carried as zeros, never read)
        u32x2 tr[2] = {};
        // (DOWN: three waves per SIMD = 168 registers: no carried rows; every stage reads its four rows, which the four-row rings still hold)
        constexpr bool CARRY = UP;
        constexpr int Q0 = ST && CARRY ? 2 : 0;
        if (ST && CARRY) { q0[0] = c0[0]; q0[1] = c0[1]; q1[0] = c1r[0]; q1[1] = c1r[1]; q2[0] = c2[0]; q2[1] = c2[1]; }
        // UP: all reads of the iteration up front (two waves per SIMD: a wave must cover its own LDS latency); DOWN: each stage's reads in front of
        // its MFMAs (three waves per SIMD on 168 registers: sixteen fragment registers live at a time instead of forty-eight)
        constexpr bool EARLY = UP;
        auto rd0 = [&]() {
#pragma unroll
            for (int q = Q0; q < 4; ++q) if (!(R8W_ABL & 32) || !ST) q0[q] = *reinterpret_cast<const u32x4*>(r0 + r8w_wrap(i4, q - 3, NR) * W0 * 16 + lcol);
        };
        auto rd1 = [&]() {
#pragma unroll
            for (int q = Q0; q < 4; ++q) if (!(R8W_ABL & 32) || !ST) q1[q] = *reinterpret_cast<const u32x4*>(r1 + r8w_wrap(i4, (q - 6) % NR, NR) * W1 * 16 + lcol);
        };
        auto rd2 = [&]() {
#pragma unroll
            for (int q = Q0; q < 4; ++q) if (!(R8W_ABL & 32) || !ST) q2[q] = *reinterpret_cast<const u32x4*>(r2 + r8w_wrap(i4, (q - 9) % NR, NR) * W2 * 16 + lcol);
#pragma unroll
            for (int r = 0; r < 2; ++r) tr[r] = *reinterpret_cast<const u32x2*>(tc + r8w_wrap(i16, r - 8, NT) * TW * 16 + tcol);
        };
        if (EARLY && do_s1) rd0();
        if (EARLY && do_s2) rd1();
        if (EARLY && do_s3) rd2();
        // ---- the wait that retires pair k + 1 (requested in iteration k - M behind that iteration's stores), then conv1's reads of it ----
        u32x4 fa = {}, fb = {};                                  // DOWN: the two tiles' fragments
        if (do_c1) {
            if constexpr (UP) {
                // operations issued behind pair k + 1's request: 2 stores + 2 requests per iteration k - M + 1 .. k - 1
                if (ST) r8w_wait_vm<4 * M - 4>();
                else if (issue && k >= M) r8w_wait_vm<2 * M - 2>();   // (no stores yet, or fewer)
                else r8w_wait_vm<0>();                               // (head: the pairs of the prologue; tail: no request behind the last pairs)
                inr[0][0] = cin[0][0]; inr[0][1] = cin[0][1]; inr[1][0] = cin[1][0]; inr[1][1] = cin[1][1];
                if (!ST && k == 0) {                                 // (pair 0 has no iteration in front of it)
#pragma unroll
                    for (int q = 0; q < 2; ++q) {
                        inr[q][0] = *reinterpret_cast<const u32x4*>(in + q * IW * 16 + lcol);
                        inr[q][1] = *reinterpret_cast<const u32x4*>(in + INPL + q * IW * 16 + lcol);
                    }
                }
#pragma unroll
                for (int q = 2; q < 4; ++q) {
                    if ((R8W_ABL & 32) && ST) continue;
                    const int off = (im * 2 + (q & 1)) * IW * 16;
                    inr[q][0] = *reinterpret_cast<const u32x4*>(in + off + lcol);
                    inr[q][1] = *reinterpret_cast<const u32x4*>(in + INPL + off + lcol);
                }
            } else {
                // pair k + 1 (waiting in ireg[k mod M]) -> image ring slot (k + 1) & 1, its register takes the request for pair k + 1 + M; then the
                // fragments: k = 8 kk + jj <-> window row 2 (kk & 1) + (jj >> 2), column jj & 3 of the pair (lane groups 2, 3 and window row 3 have
                // zero weights: row 3 is read as row 2 again, any finite value)
                const int ik = ST ? PH % M : rik;                    // (a run-time index in the general form: selects, not a register array in scratch)
                float pv = ireg[0];
#pragma unroll
                for (int p = 1; p < M; ++p) pv = ik == p ? ireg[p] : pv;
                istore(pv, (k + 1) & 1);
                if (issue) {
                    const float nv = iload(k + 1 + M);
#pragma unroll
                    for (int p = 0; p < M; ++p) ireg[p] = ik == p ? nv : ireg[p];
                }
                const int h = kk & 1;                                // window rows 2 h, 2 h + 1
                // image ring rows of window row wr of tile A (r0 row 2 k + 1: image rows 2 k .. 2 k + 2 of the item) / tile B (+ 1): pairs k, k + 1
                auto irow = [&](int wr) { return (((k + (wr >> 1)) & 1) * 2 + (wr & 1)) * IW * 2; };
                const int ra0 = irow(2 * h), ra1 = irow(h ? 2 : 1), rb0 = irow(2 * h + 1), rb1 = irow(h ? 3 : 2);
                const unsigned char* pc = in + 4 * j;                // two pixels = one dword; the pair's window = two dwords (4-byte aligned)
                auto dw = [&](int off, int i) { return *reinterpret_cast<const unsigned*>(pc + off + 4 * i); };
                fa = u32x4{dw(ra0, 0), dw(ra0, 1), dw(ra1, 0), dw(ra1, 1)};
                fb = u32x4{dw(rb0, 0), dw(rb0, 1), dw(rb1, 0), dw(rb1, 1)};
            }
        }
        // ---- stages 1 and 2: ring rows 2 k - 2, 2 k - 1 of r1 / 2 k - 5, 2 k - 4 of r2 ----
        auto stage = [&](const u32x4 (&wf)[3], const u32x4 (&q)[4], f32x4 b4, int col, int rtop, int wreg) {
            f32x4 ra = b4, rb = b4;
#pragma unroll
            for (int ky = 0; ky < 3; ++ky) { ra = r8w_mm(wf[ky], q[ky], ra); rb = r8w_mm(wf[ky], q[ky + 1], rb); }
            const u32x4 rec = whole(relu_pk(pack_bf16x4(ra)), relu_pk(pack_bf16x4(rb)));
            const int o0 = r8w_wrap(i4, rtop % NR, NR) * wreg * 16, o1 = r8w_wrap(i4, (rtop + 1) % NR, NR) * wreg * 16;
            if (!(R8W_ABL & 16) || rec.x == 0x12345678u) *reinterpret_cast<u32x4*>(lds + col + (isB ? o1 : o0)) = rec;
        };
        // (consumers in front of producers: with its reads in front of its MFMAs a stage must have read its four ring rows before the stage
        //  above it stores the next two into the same four-row ring)
        // ---- stage 3: output rows Ya - 12 + 2 k, + 1: + raw t, ReLU after the rounding, 8 bytes per lane and row ----
        if (do_s3) {
            if (!EARLY) rd2();
            f32x4 v[2] = {biasw[2], biasw[2]};
#pragma unroll
            for (int ky = 0; ky < 3; ++ky) { v[0] = r8w_mm(w[2][ky], q2[ky], v[0]); v[1] = r8w_mm(w[2][ky], q2[ky + 1], v[1]); }
            const unsigned oo = ooff0 + (unsigned)(2 * k) * orow;
            u32x2 pk[2];
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                pk[r] = relu_pk(pack_bf16x4(v[r] + unpack_bf16x4(tr[r])));
                if (ost && (!(R8W_ABL & 2) || pk[r].x == 0x12345678u)) *reinterpret_cast<u32x2*>(outb + (oo + (unsigned)r * orow)) = pk[r];
            }
            if constexpr (!UP) {
                if (poolb) {
                    // 2 x 2 max on the packed values (non-negative bf16 order like their bit patterns): the two rows, then the pixel pair (lane ^ 32)
                    const unsigned m0 = pkmax_u16(pk[0].x, pk[1].x), m1 = pkmax_u16(pk[0].y, pk[1].y);
                    const auto s0 = __builtin_amdgcn_permlane32_swap(m0, m0, false, false);
                    const auto s1 = __builtin_amdgcn_permlane32_swap(m1, m1, false, false);
                    if (ost && e == 0) *reinterpret_cast<u32x2*>(poolb + (poff0 + (unsigned)k * prow)) = u32x2{pkmax_u16(s0[0], s0[1]), pkmax_u16(s1[0], s1[1])};
                }
            }
        }
        if (do_s2) { if (!EARLY) rd1(); stage(w[1], q1, biasw[1], col2, -5, W2); }
        if (do_s1) { if (!EARLY) rd0(); stage(w[0], q0, biasw[0], col1, -2, W1); }
        // ---- conv1: rows 2 k + 1, 2 k + 2 of relu(t) -> r0, raw t -> tc; its reads have returned: pair k + 1's slot takes the next request ----
        if (do_c1) {
            f32x4 ra = bias1, rb = bias1;
            if constexpr (UP) {
                r8w_wait_lds();
                if (!ST && k == 0 && M <= p_last) dma(M, 0);
                if (issue) dma(k + 1 + M, im);
#pragma unroll
                for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                    for (int src = 0; src < 2; ++src) {
                        ra = r8w_mm(a1[ky * 2 + src], inr[ky][src], ra);
                        rb = r8w_mm(a1[ky * 2 + src], inr[ky + 1][src], rb);
                    }
            } else {
                ra = r8w_mm(a1[0], fa, ra);
                rb = r8w_mm(a1[0], fb, rb);
            }
            const u32x4 raw = whole(pack_bf16x4(ra), pack_bf16x4(rb)), rl = relu_bf16x8(raw);
            const int o0 = r8w_wrap(i4, 1, NR) * W0 * 16, o1 = r8w_wrap(i4, 2, NR) * W0 * 16;
            if (!(R8W_ABL & 16) || rl.x == 0x12345678u) *reinterpret_cast<u32x4*>(lds + col0 + (isB ? o1 : o0)) = rl;
            const int t0 = r8w_wrap(i16, 1, NT) * TW * 16, t1 = r8w_wrap(i16, 2, NT) * TW * 16;
            if (!(R8W_ABL & 16) || raw.x == 0x12345678u) *reinterpret_cast<u32x4*>(lds + colt + (isB ? t1 : t0)) = raw;   // (rows outside the item's output rows are never read)
        }
        // (rows 2, 3 of this iteration's slots are rows 0, 1 of the next one's)
#pragma unroll
        for (int q = 0; q < 2 && CARRY; ++q) { cin[q][0] = inr[2 + q][0]; cin[q][1] = inr[2 + q][1]; c0[q] = q0[2 + q]; c1r[q] = q1[2 + q]; c2[q] = q2[2 + q]; }
        if (!ST) {
            rim = rim + 1 == M ? 0 : rim + 1;
            rik = rik + 1 == M ? 0 : rik + 1;
            ri4 = ri4 + 2 == NR ? 0 : ri4 + 2;
            ri16 = ri16 + 2 == NT ? 0 : ri16 + 2;
        }
    };

    const int K = nb / 2 + 6;
    // steady iterations: all four stages active, a request issued and M - 1 iterations with stores behind them: 5 + M <= k, k + 1 + M <= p_last
    const int k_steady_end = nb / 2 + 3 - M;                 // (exclusive)
    constexpr int K0 = 5 + M;                                // first steady iteration; groups of UN start at K0 + UN g
    int k = 0;
    for (; k < min(K0, K); ++k) iteration(ic<-1>{}, k);
    for (; k + UN <= k_steady_end; k += UN)
        static_for<UN>([&](auto u) { iteration(ic<(K0 + decltype(u)::value) % UN>{}, k + decltype(u)::value); });
    for (; k < K; ++k) iteration(ic<-1>{}, k);
}

// ---- the frame around the walker's region: general tiles with clipped stores --------------------------------------------------
struct Res8WBArgs {
    Res8BArgs b;           // p[i].tile_begin = first border tile of problem i; tiles_x unused
    int nbx[MAXP];         // tiles of a full-width row: ceil(W / 32)
    int nby[MAXP];         // tiles of a side column: ceil((y_end - 16) / 16)
    int y_end[MAXP];
    int xr[MAXP];          // first column behind the strips: 32 + 24 n_strips
};
template <bool UP>
__global__ __launch_bounds__(256, UP ? 3 : 4) void res8wb_kernel(const Res8WBArgs a) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[Res8BLayout<UP>::BYTES];
    const int bid = (int)blockIdx.x;
    const int pi = prob_of_tile(a.b, bid);
    const Res8BProb& P = a.b.p[pi];
    const int t = bid - P.tile_begin;
    const int nbx = a.nbx[pi], nby = a.nby[pi], ye = a.y_end[pi], xr = a.xr[pi];
    int x0, y0, ymax, xmax;
    if (t < nbx) { x0 = 32 * t; y0 = 0; ymax = R8W_Y0; xmax = P.W; }                                  // top row
    else if (t < 2 * nbx) { x0 = 32 * (t - nbx); y0 = ye; ymax = P.H; xmax = P.W; }                   // bottom rows
    else if (t < 2 * nbx + nby) { x0 = 0; y0 = R8W_Y0 + 16 * (t - 2 * nbx); ymax = ye; xmax = R8W_X0; }   // left column
    else { x0 = xr; y0 = R8W_Y0 + 16 * (t - 2 * nbx - nby); ymax = ye; xmax = P.W; }                  // right columns
    res8b_tile<UP>(a.b, P, x0, y0, lds, ymax, xmax);
}

}  // namespace asep
