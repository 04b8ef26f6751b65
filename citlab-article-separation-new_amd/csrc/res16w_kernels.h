// res16w_kernel: the residual tail of the 16-channel level of the bf16 path (three 16 -> 16 convolutions + t, ReLU, optional 2 x 2 pool:
// ARU_v1.py:208-226,270-292 at level 1) as a COLUMN-STRIP WALKER -- the form of res8w_kernels.h for the mapping of res16f_kernel
// (M = 16 channels, K chunk = two taps x 16 channels, N = 16 pixels of a row; five chunks per convolution).
//
// One wave walks down a strip of 26 output columns, two rows per iteration; every stage region is two tiles wide (30 / 28 / 26 pixels =
// 16 + 14 / 16 + 12 / 16 + 10), the input t 32 pixels = 64 units of 16 bytes = ONE load instruction per row.  An iteration runs six slots
// (a slot = one row of a stage = its two tiles = 10 fragment reads + 10 MFMAs + an epilogue), consumers in front of producers (stage 3, 2, 1)
// so that rings of FOUR rows suffice, software-pipelined over the slots: the reads of slot s + 2 behind the MFMAs of slot s, the epilogue of
// slot s - 1 beside them.  The input pairs wait in registers (M pairs in flight), get their ReLU and go to LDS when stage 1 needs them; the
// residual operand (raw t) of stage 3 comes from HBM (L2: the strip read those rows eight rows earlier), requested an iteration ahead.  The
// steady iterations are unrolled over the rings' period of four: a fragment read is a per-lane constant + an immediate (the one chunk whose
// two taps lie in different filter rows selects between two immediates per lane group).  16-byte units of a row are swizzled like
// res8w_kernel's.  12 KB of LDS per wave.  Accumulation order of res16f_kernel (bias first, chunks 0 .. 4): bit-identical where both are lean.
//
// The walker covers columns [32, 32 + 26 n) x rows [16, y_end); the frame around it is res16wb_kernel (resb_tail_tile<16> with clipped stores).
#pragma once
#include "res8w_kernels.h"

namespace asep {

constexpr int R16W_TW = 26;
constexpr int R16W_X0 = 32, R16W_Y0 = 16;
#ifndef R16W_DEPTH
#define R16W_DEPTH 2
#endif

struct Res16WProb {
    const bf16_t* t;       // [H,W,16] conv1 output (pre-ReLU)
    bf16_t* out;           // [H,W,16]
    bf16_t* pool;          // maxpool2(out) or nullptr
    int H, W;
    int n_strips, band, y_end;
    int tile_begin;        // first ITEM of this problem (band-major, the strips of a band side by side)
};
struct Res16WArgs {
    Res16WProb p[MAXP];
    int nprob;
    const u32x4* wpk;      // [3 convs][5 chunks][64 lanes] x 16 bytes
    const float* bias;     // [3][16]
    XcdMap xm;
};

__global__ __launch_bounds__(64, 2) void res16w_kernel(const Res16WArgs a) {
    constexpr int TW = R16W_TW, M = R16W_DEPTH, C = 16, PXB = 32, CPC = 5;
    constexpr int IW = TW + 6, W1 = TW + 4, W2 = TW + 2;                        // 32, 30, 28 pixels
    constexpr int NR = 4, UN = 2;
    static_assert(UN % M == 0 && (2 * UN) % NR == 0, "the steady form is unrolled over the rings' common period");
    constexpr int R0_OFF = 0, R1_OFF = NR * IW * PXB, R2_OFF = R1_OFF + NR * W1 * PXB, TRASH = R2_OFF + NR * W2 * PXB + 256, LDSB = TRASH + 16;
    __shared__ __attribute__((aligned(16))) unsigned char lds[LDSB];

    const int lane = threadIdx.x;
    const int j = lane & 15, kk = lane >> 4;
    const int item = sched_tile(a.xm);
    if (item < 0) return;
    const int pi = prob_of_tile(a, item);
    const Res16WProb& P = a.p[pi];
    const int li = item - P.tile_begin;
    const int bi = li / P.n_strips, si = li - bi * P.n_strips;
    const int x0 = R16W_X0 + TW * si;
    const int Ya = R16W_Y0 + bi * P.band;
    const int nb = min(P.band, P.y_end - Ya);                                 // output rows of this item (even)
    const unsigned wu = (unsigned)P.W;

    f32x4 biasw[3];
#pragma unroll
    for (int s = 0; s < 3; ++s) biasw[s] = *reinterpret_cast<const f32x4*>(a.bias + s * C + kk * 4);
    u32x4 w[3][CPC];
#pragma unroll
    for (int s = 0; s < 3; ++s)
#pragma unroll
        for (int t = 0; t < CPC; ++t) w[s][t] = a.wpk[(s * CPC + t) * 64 + lane];

    // ---- the lane's share of K chunk t: tap 2 t + (kk >> 1) (the padded ninth slot: tap 8 again, zero weights), channel half kk & 1 ----
    int lc[CPC];                                              // byte offset inside a row of the lane's unit of tile a (swizzled); tile b: + 16 pixels =
                                                              // + 32 units, which the swizzle (bit 4 of the unit index) leaves alone: an immediate
    const int hi = kk >> 1;
#pragma unroll
    for (int t = 0; t < CPC; ++t) {
        const int tap = min(2 * t + hi, 8), kx = tap % 3;
        lc[t] = r8w_swz(2 * (j + kx) + (kk & 1)) * 16;
    }
    // filter row of chunk t's tap: uniform for t = 0 (0), 2 (1), 3 (2), 4 (2); chunk 1 = taps 2, 3: row 0 for lane groups 0, 1, row 1 for 2, 3
    // stores of a tile's result: pixel j (+ 16), channels 4 kk .. 4 kk + 3 = 8 bytes
    const int sa = r8w_swz(2 * j + (kk >> 1)) * 16 + (kk & 1) * 8, sb = r8w_swz(2 * (j + 16) + (kk >> 1)) * 16 + (kk & 1) * 8;
    const int s1a = R1_OFF + sa, s1b = j < W1 - 16 ? R1_OFF + sb : TRASH;      // (a lane beyond a region's columns stores into a dump)
    const int s2a = R2_OFF + sa, s2b = j < W2 - 16 ? R2_OFF + sb : TRASH;
    const bool ob = j < TW - 16;

    // ---- input pairs: pair p = image rows Ya - 3 + 2 p, + 1; a row = 64 units = one load; M pairs wait in registers ----
    const unsigned char* __restrict__ const tb8 = reinterpret_cast<const unsigned char*>(P.t);
    const unsigned ioff0 = ((unsigned)(Ya - 3) * wu + (unsigned)(x0 - 3)) * 32u + (unsigned)lane * 16u, irow = wu * 32u;
    const int p_last = nb / 2 + 2;
    u32x4 ireg[M][2];
    auto iload = [&](int p, u32x4 (&v)[2]) {
        v[0] = *reinterpret_cast<const u32x4*>(tb8 + (ioff0 + (unsigned)(2 * p) * irow));
        v[1] = *reinterpret_cast<const u32x4*>(tb8 + (ioff0 + (unsigned)(2 * p + 1) * irow));
    };
    const int ist = r8w_swz(lane) * 16;
    auto istore = [&](const u32x4 (&v)[2], int slot) {       // relu(t) -> rows 2 slot, 2 slot + 1 of r0
        *reinterpret_cast<u32x4*>(lds + R0_OFF + (2 * slot) * IW * PXB + ist) = relu_bf16x8(v[0]);
        *reinterpret_cast<u32x4*>(lds + R0_OFF + (2 * slot + 1) * IW * PXB + ist) = relu_bf16x8(v[1]);
    };
    {
        u32x4 v0[2];
        iload(0, v0);
#pragma unroll
        for (int p = 0; p < M; ++p) iload(p + 1, ireg[p]);   // (pair p + 1 waits in ireg[p mod M]; p_last >= M)
        istore(v0, 0);
    }
    // residual operand and output of stage 3: pixel (row, x0 + j (+ 16)), channels 4 kk ..
    unsigned char* __restrict__ const outb = reinterpret_cast<unsigned char*>(P.out);
    unsigned char* __restrict__ const poolb = reinterpret_cast<unsigned char*>(P.pool);
    const unsigned ooff0 = (((unsigned)(Ya - 8) * wu + (unsigned)(x0 + j)) * C + (unsigned)kk * 4) * 2u, orow = wu * C * 2u;
    const unsigned Wp = (unsigned)((P.W + 1) >> 1);
    const unsigned poff0 = (((unsigned)((Ya - 8) >> 1) * Wp + (unsigned)((x0 + j) >> 1)) * C + (unsigned)kk * 4) * 2u, prow = Wp * C * 2u;
    auto relu_pk = [](u32x2 p) { return u32x2{relu_bf16x2(p.x), relu_bf16x2(p.y)}; };
    typedef FragPair<CPC> Fr;

    int ri4 = 0, rik = 0;                                    // 2 k mod 4, k mod M of the general form
    u32x2 resn[2][2] = {};                                   // the NEXT iteration's residual operand [row][tile]
    auto iteration = [&](auto phase_c, int k) {
        constexpr int PH = decltype(phase_c)::value;         // >= 0: k mod UN of a steady iteration
        constexpr bool ST = PH >= 0;
        const int i4 = ST ? (2 * PH) % NR : ri4;
        const bool do_s1 = ST || k < nb / 2 + 2, do_s2 = ST || (k >= 2 && k < nb / 2 + 3), do_s3 = ST || k >= 4;
        const bool issue = ST || k + 1 + M <= p_last, stage_in = ST || k + 1 <= p_last;
        // ---- pair k + 1 -> r0 (its ReLU here), its register takes the request for pair k + 1 + M; this iteration's residual operand was requested
        //      by the previous one, the next one's is requested now ----
        if (stage_in) {
            const int ik = ST ? PH % M : rik;
            u32x4 pv[2] = {ireg[0][0], ireg[0][1]};
#pragma unroll
            for (int p = 1; p < M; ++p) { pv[0] = ik == p ? ireg[p][0] : pv[0]; pv[1] = ik == p ? ireg[p][1] : pv[1]; }
            istore(pv, (k + 1) & 1);
            if (issue) {
                u32x4 nv[2];
                iload(k + 1 + M, nv);
#pragma unroll
                for (int p = 0; p < M; ++p) { ireg[p][0] = ik == p ? nv[0] : ireg[p][0]; ireg[p][1] = ik == p ? nv[1] : ireg[p][1]; }
            }
        }
        u32x2 res[2][2] = {{resn[0][0], resn[0][1]}, {resn[1][0], resn[1][1]}};
        if (ST || (k + 1 >= 4 && k + 1 < nb / 2 + 4)) {
            const unsigned oo = ooff0 + (unsigned)(2 * (k + 1)) * orow;
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                resn[r][0] = *reinterpret_cast<const u32x2*>(tb8 + (oo + (unsigned)r * orow));
                resn[r][1] = ob ? *reinterpret_cast<const u32x2*>(tb8 + (oo + (unsigned)r * orow + 16 * C * 2)) : u32x2{0u, 0u};
            }
        }
        // ---- six slots: stage 3 rows 2 k - 5, 2 k - 4 (ring coordinates: image row Ya - 3 + r), stage 2 rows 2 k - 2, 2 k - 1, stage 1 rows 2 k + 1,
        //      2 k + 2; a slot's output row r reads source rows r - 1, r, r + 1 of the ring below it ----
        auto src_row = [&](int rel) { return r8w_wrap(i4, ((rel % NR) + NR) % NR, NR); };       // ring slot of row 2 k + rel
        auto ld = [&](auto slot_c, Fr& f) {
            constexpr int s = decltype(slot_c)::value;
            constexpr int stage = 2 - s / 2, r = s & 1;       // slots 0, 1: stage 3 (index 2); 2, 3: stage 2; 4, 5: stage 1
            constexpr int rel = (stage == 2 ? -5 : stage == 1 ? -2 : 1) + r;
            constexpr int SRC = stage == 2 ? R2_OFF : stage == 1 ? R1_OFF : R0_OFF, WIN = stage == 2 ? W2 : stage == 1 ? W1 : IW;
            if ((stage == 2 && !do_s3) || (stage == 1 && !do_s2) || (stage == 0 && !do_s1)) return;
            const int o0 = SRC + src_row(rel - 1) * WIN * PXB, o1 = SRC + src_row(rel) * WIN * PXB, o2 = SRC + src_row(rel + 1) * WIN * PXB;
            const int oy[CPC] = {o0, hi ? o1 : o0, o1, o2, o2};   // filter row of the lane's tap of chunk t
#pragma unroll
            for (int t = 0; t < CPC; ++t) {
                f.a[t] = *reinterpret_cast<const u32x4*>(lds + oy[t] + lc[t]);
                f.b[t] = *reinterpret_cast<const u32x4*>(lds + oy[t] + lc[t] + 16 * PXB);
            }
        };
        u32x2 pk3[2][2];
        auto st = [&](auto slot_c, f32x4 va, f32x4 vb) {
            constexpr int s = decltype(slot_c)::value;
            constexpr int stage = 2 - s / 2, r = s & 1;
            if constexpr (stage == 2) {
                if (!do_s3) return;
                const unsigned oo = ooff0 + (unsigned)(2 * k + r) * orow;
                pk3[r][0] = relu_pk(pack_bf16x4(va + unpack_bf16x4(res[r][0])));
                pk3[r][1] = relu_pk(pack_bf16x4(vb + unpack_bf16x4(res[r][1])));
                *reinterpret_cast<u32x2*>(outb + oo) = pk3[r][0];
                if (ob) *reinterpret_cast<u32x2*>(outb + (oo + 16 * C * 2)) = pk3[r][1];
                if (r == 1 && poolb) {
                    // 2 x 2 max on the packed values: the two rows, then the neighbour lane (pixel j ^ 1)
#pragma unroll
                    for (int cb = 0; cb < 2; ++cb) {
                        const unsigned m0 = pkmax_u16(pk3[0][cb].x, pk3[1][cb].x), m1 = pkmax_u16(pk3[0][cb].y, pk3[1][cb].y);
                        const unsigned n0 = __float_as_uint(lane_xor1(__uint_as_float(m0))), n1 = __float_as_uint(lane_xor1(__uint_as_float(m1)));
                        if ((j & 1) == 0 && (cb == 0 || ob))
                            *reinterpret_cast<u32x2*>(poolb + (poff0 + (unsigned)k * prow + cb * 8 * C * 2)) = u32x2{pkmax_u16(m0, n0), pkmax_u16(m1, n1)};
                    }
                }
            } else {
                if ((stage == 1 && !do_s2) || (stage == 0 && !do_s1)) return;
                constexpr int rel = (stage == 1 ? -2 : 1) + r, WOUT = stage == 1 ? W2 : W1;
                const int orow_l = src_row(rel) * WOUT * PXB;
                *reinterpret_cast<u32x2*>(lds + (stage == 1 ? s2a : s1a) + orow_l) = relu_pk(pack_bf16x4(va));
                *reinterpret_cast<u32x2*>(lds + (stage == 1 ? s2b : s1b) + orow_l) = relu_pk(pack_bf16x4(vb));
            }
        };
        {
            Fr f[2];
            f32x4 ra[2], rb[2];
            ld(ic<0>{}, f[0]);
            ld(ic<1>{}, f[1]);
            static_for<6>([&](auto sc) {
                constexpr int s = decltype(sc)::value;
                constexpr int stage = 2 - s / 2;
                mm_pair<CPC>(w[stage], f[s & 1], biasw[stage], ra[s & 1], rb[s & 1]);
                if constexpr (s + 2 < 6) ld(ic<s + 2>{}, f[s & 1]);
                if constexpr (s >= 1) st(ic<s - 1>{}, ra[(s - 1) & 1], rb[(s - 1) & 1]);
            });
            st(ic<5>{}, ra[1], rb[1]);
        }
        if (!ST) {
            ri4 = ri4 + 2 == NR ? 0 : ri4 + 2;
            rik = rik + 1 == M ? 0 : rik + 1;
        }
    };

    const int K = nb / 2 + 4;
    // steady: every stage active (k >= 4, k < nb / 2 + 2), a request issued (k + 1 + M <= p_last)
    const int k_steady_end = min(nb / 2 + 2, p_last - M);    // (exclusive)
    constexpr int K0 = 4;
    int k = 0;
    for (; k < min(K0, K); ++k) iteration(ic<-1>{}, k);
    for (; k + UN <= k_steady_end; k += UN)
        static_for<UN>([&](auto u) { iteration(ic<(K0 + decltype(u)::value) % UN>{}, k + decltype(u)::value); });
    for (; k < K; ++k) iteration(ic<-1>{}, k);
}

// ---- the frame around the walker's region: general tiles with clipped stores ----
struct Res16WBArgs {
    ResBArgs b;            // p[i].tile_begin = first border tile of problem i
    int nbx[MAXP], nby[MAXP], y_end[MAXP], xr[MAXP];
};
__global__ __launch_bounds__(256, 3) void res16wb_kernel(const Res16WBArgs a) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[ResBLayout<16>::BYTES];
    const int bid = (int)blockIdx.x;
    const int pi = prob_of_tile(a.b, bid);
    const ResBProb& P = a.b.p[pi];
    const int t = bid - P.tile_begin;
    const int nbx = a.nbx[pi], nby = a.nby[pi], ye = a.y_end[pi], xr = a.xr[pi];
    int x0, y0, ymax, xmax;
    if (t < nbx) { x0 = 32 * t; y0 = 0; ymax = R16W_Y0; xmax = P.W; }
    else if (t < 2 * nbx) { x0 = 32 * (t - nbx); y0 = ye; ymax = P.H; xmax = P.W; }
    else if (t < 2 * nbx + nby) { x0 = 0; y0 = R16W_Y0 + 16 * (t - 2 * nbx); ymax = ye; xmax = R16W_X0; }
    else { x0 = xr; y0 = R16W_Y0 + 16 * (t - 2 * nbx - nby); ymax = ye; xmax = P.W; }
    resb_tail_tile<16>(a.b, P, x0, y0, lds, ymax, xmax);
}

}  // namespace asep
