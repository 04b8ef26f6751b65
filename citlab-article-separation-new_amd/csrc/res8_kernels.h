// Fused level-0 residual blocks of the ARU-Net (8 feature channels, ARU_v1.py:208-245 and :266-281).
//
// The 8-channel layers at full page resolution are HBM-bound when run layer by layer (18 FLOP/B) and fill only
// half of a 16-row MFMA tile.  This kernel keeps the whole block
//     t = conv1(x) ; r0 = relu(conv(relu(t))) ; r1 = relu(conv(r0)) ; d = relu(conv(r1) + t) [; pool = maxpool2(d)]
// in LDS for a 16 x 58 pixel output tile (halo recomputation: 22/20/18/16 rows x 64 columns per stage) and maps
// each 3x3x8->8 convolution onto v_mfma_f32_16x16x4_f32 with M = 8 output channels x 2 horizontally adjacent
// pixels ("pixel pair"), N = 16 pixel pairs, K = 3 rows x 4 columns x 8 channels = 96 (75 % useful MACs instead
// of 45 % for the generic kernel).  A stage's 8->8 filter lives in registers for that stage only (24 VGPRs, requested
// one stage ahead); a wave's two pixel-pair units per stage have their LDS fragments requested up front.
#pragma once
#include "aru_kernels.h"

namespace asep {

#ifdef ASEP_R8_TIMELINE   // development aid (scripts/ubench/res8_timeline.hip): per-wave cycle stamps of the first units
__device__ unsigned long long r8_tl[16][8][64];
__device__ unsigned long long r8_clk[4];   // block 0: {clock64, wall_clock64} at kernel start and end
#define R8_MARK() do { if (blockIdx.x < 16 && lane == 0 && tl_n < 64) r8_tl[blockIdx.x][wave][tl_n++] = clock64(); } while (0)
#else
#define R8_MARK() do { } while (0)
#endif

constexpr int R8_OH = 16, R8_OW = 58;          // output tile
constexpr int R8_FH = R8_OH + 8;               // frame rows (4 halo rows each side)
constexpr int R8_PITCH = 72;                   // pixels per LDS row (frame columns 0..71)
constexpr int R8_IMGP = 76;                    // image tile pitch (frame columns -2..73)
constexpr int R8_WAVES = 8;                    // 512 threads: two waves per SIMD hide the LDS->MFMA latency
constexpr int R8_THREADS = R8_WAVES * 64;
#ifndef R8_NP_VALUE
#define R8_NP_VALUE 4
#endif
constexpr int R8_NP = R8_NP_VALUE;                       // vertical passes per work unit: a unit is R8_NP * R8_OH = 64 rows x 58 columns.
                                               // Pass 0 recomputes the 4-row halo above the unit; every further pass
                                               // carries the 2-3 bottom rows of t / r0 / r1 over from the pass above
                                               // instead of recomputing them: 16 rows per stage (two units per wave,
                                               // perfectly balanced) instead of 22 / 20 / 18 / 16.

struct Res8Prob {
    const float* img;      // DOWN: [H,W] single-channel input.  UP: skip tensor d0 [H,W,8]
    const float* in1;      // UP: deconv output v [H,W,8]
    const float* stats;    // DOWN: per-image standardisation {mean, 1/std} or nullptr
    float* out;            // block output [H,W,8]
    float* pool;           // DOWN: maxpool2(out) [ceil(H/2), ceil(W/2), 8] or nullptr
    int H, W;
    int tiles_x, tile_begin;
};
struct Res8Args {
    Res8Prob p[MAXP];
    int nprob, total_tiles;
    const float* w1;       // DOWN: conv1 [9][8] ; UP: packed pair-fragments of conv1 (12 chunks x 64 lanes x 4)
    const float* b1;       // [8]
    const f32x4* wr;       // convR_0..2 packed pair-fragments: [3][6 chunks][64 lanes] x 4 (res8v_kernels.h: scalar layout, see there)
    const float* br;       // [3][8]
    const int32_t* sched;  // XCD-aware schedule: the k-th unit of work (k = blockIdx.x + i * gridDim.x) is tile
                           // sched[k]; nullptr = identity.  Blocks are dealt round-robin to the 8 XCDs, so the table
                           // gives every XCD one compact region of the page (its halo re-reads hit its own L2).
};

__device__ __forceinline__ int res8_tile_of(const Res8Args& a, int k) { return a.sched ? a.sched[k] : k; }

// max(x, 0) on the bit patterns: one v_max_i32 per element (fmaxf costs a second, canonicalising v_max_f32)
__device__ __forceinline__ f32x4 relu4i(f32x4 v) {
    typedef int i32x4 __attribute__((ext_vector_type(4)));
    i32x4 i = __builtin_bit_cast(i32x4, v);
    i.x = max(i.x, 0); i.y = max(i.y, 0); i.z = max(i.z, 0); i.w = max(i.w, 0);
    return __builtin_bit_cast(f32x4, i);
}

// lane-indexed A fragments of one 3x3 8->8 filter (6 chunks).  The pointer is laundered so that the loads stay where
// they are written: a filter lives in registers for one stage only (24 VGPRs instead of 72 for the whole block; the
// fused kernels are register-bound, and with the filters resident the compiler serialised every LDS read with its use).
__device__ __forceinline__ void r8_load_w(const f32x4* __restrict__ base, int lane, f32x4 (&A)[6]) {
    // explicitly a global-memory pointer: behind the asm the compiler would fall back to flat loads, which count in
    // lgkmcnt as well and force every later LDS wait down to zero
    typedef const f32x4 __attribute__((address_space(1)))* gptr;
    gptr q = (gptr)(base + lane);
    asm volatile("" : "+v"(q));
#pragma unroll
    for (int c = 0; c < 6; ++c) A[c] = q[c * 64];
}

// Float offset of the 16-byte piece (pixel x of the row, channel half hf) inside an LDS row.  A B-fragment read takes,
// for 16 consecutive pixel PAIRS, the same piece of each pair: at the natural layout (64 bytes per pair) those sixteen
// 16-byte reads fall into two bank groups (rocprofv3: SQ_LDS_BANK_CONFLICT = 53 % of SQ_LDS_IDX_ACTIVE in res8_up,
// profiles/r2i_instruction_mix).  The four pieces of a pair are therefore permuted by the pair's index: piece k of pair p
// sits in slot k ^ ((p >> 1) & 3), so that eight consecutive pairs present eight different bank groups.  Whole-row
// copies (carried rows) are layout-preserving and need no change.
__device__ __forceinline__ int r8_px(int x, int hf) {
#ifdef R8_NO_SWIZZLE
    return x * 8 + hf * 4;
#else
    return ((x >> 1) << 4) + (((((x & 1) << 1) | hf) ^ ((x >> 2) & 3)) << 2);
#endif
}

// B fragments of one pixel-pair unit (2 output rows x 32 columns): 4 input rows x 2 column halves; `row` = first input
// row, the lane's pixel x and x + 2, channel half hf
__device__ __forceinline__ void r8_load_frags(const float* __restrict__ row, int x, int hf, f32x4 (&b)[4][2]) {
    const int o0 = r8_px(x, hf), o1 = r8_px(x + 2, hf);
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
        b[rr][0] = *reinterpret_cast<const f32x4*>(row + rr * R8_PITCH * 8 + o0);
        b[rr][1] = *reinterpret_cast<const f32x4*>(row + rr * R8_PITCH * 8 + o1);
    }
}

template <bool RELU_IN, bool BF>
__device__ __forceinline__ void r8_mma(const f32x4 (&A)[6], f32x4 (&b)[4][2], f32x4& acc0, f32x4& acc1) {
    if (RELU_IN) {
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) { b[rr][0] = relu4i(b[rr][0]); b[rr][1] = relu4i(b[rr][1]); }
    }
#pragma unroll
    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int c = ky * 2 + h;
            if constexpr (BF) {
                const s16x4 pa = bf16pack(A[c]);
                acc0 = mfma_bf16(pa, bf16pack(b[ky][h]), acc0);
                acc1 = mfma_bf16(pa, bf16pack(b[ky + 1][h]), acc1);
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(A[c][r], b[ky][h][r], acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(A[c][r], b[ky + 1][h][r], acc1, 0, 0, 0);
                }
            }
        }
}

// one 3x3 8->8 convolution stage on LDS tiles.  IN holds rows in_r0.. of the frame, OUT rows out_r0.. ;
// computes rows [row_start, row_start+NROWS) x columns [out_c0, out_c0+64).  FINAL: add T centre, store to global.
// The fragments of a wave's next unit are requested before the MFMAs of the current one; wnext != nullptr: the next
// stage's filter (wnext; every stage but the FINAL one) is requested into An together with the fragments, so that its
// latency hides under the MFMAs and the barrier.
// INTERIOR: the whole frame lies inside the image (scalar per pass): no zero masks, unconditional stores.
template <int NROWS, bool RELU_IN, bool FINAL, bool POOL, bool BF = false, bool INTERIOR = false>
__device__ __forceinline__ void res8_stage(const float* __restrict__ IN, int in_r0, float* __restrict__ OUT, int out_r0,
                                           int row_start, int out_c0, const f32x4 (&A)[6], const f32x4 bias4, int wave, int lane,
                                           int fy0, int fx0, int H, int W, const float* __restrict__ T, int t_r0,
                                           float* __restrict__ gout, float* __restrict__ gpool,
                                           const f32x4* __restrict__ wnext, f32x4 (&An)[6]) {
    const int j = lane & 15, kk = lane >> 4;
    const int e = kk >> 1, ch = (kk & 1) * 4;
    // pu enumerates (row pair, n-tile): NROWS/2 pairs x 2 n-tiles; a wave owns units wave, wave + 8 (, wave + 16)
    const int hf = kk & 1;
    auto load_unit = [&](int pu, f32x4 (&b)[4][2]) {
        const int rp = pu >> 1, nt = pu & 1;
        r8_load_frags(IN + (row_start + 2 * rp - 1 - in_r0) * R8_PITCH * 8, out_c0 + nt * 32 + 2 * j + e - 1, hf, b);
    };
    auto unit = [&](int pu, f32x4 (&b)[4][2]) {
        const int rp = pu >> 1, nt = pu & 1;
        const int row0 = row_start + 2 * rp;         // rows row0, row0+1
        const int colb = out_c0 + nt * 32 + 2 * j;   // this lane's pixel pair starts at colb
        f32x4 v0 = bias4, v1 = bias4;                // the bias is the accumulators' initial value
        r8_mma<RELU_IN, BF>(A, b, v0, v1);
        // D layout: lane (pair j, kk): pixel colb + e, channels ch..ch+3
        const int col = colb + e;
        const int gx = fx0 + col;
        if (!FINAL) {
            const int gy0 = fy0 + row0;
            if (INTERIOR) {
                v0 = relu4i(v0); v1 = relu4i(v1);
            } else {
                const bool okx = gx >= 0 && gx < W;
                v0 = (okx && gy0 >= 0 && gy0 < H) ? relu4i(v0) : f32x4{0.f, 0.f, 0.f, 0.f};
                v1 = (okx && gy0 + 1 >= 0 && gy0 + 1 < H) ? relu4i(v1) : f32x4{0.f, 0.f, 0.f, 0.f};
            }
            float* o = OUT + (row0 - out_r0) * R8_PITCH * 8 + r8_px(col, hf);
            *reinterpret_cast<f32x4*>(o) = v0;
            *reinterpret_cast<f32x4*>(o + R8_PITCH * 8) = v1;
        } else {
            const float* tp = T + (row0 - t_r0) * R8_PITCH * 8 + r8_px(col, hf);
            v0 = relu4i(v0 + *reinterpret_cast<const f32x4*>(tp));
            v1 = relu4i(v1 + *reinterpret_cast<const f32x4*>(tp + R8_PITCH * 8));
            const int gy0 = fy0 + row0;
            // only the OW valid columns of the tile (frame columns 4 .. 4+OW-1) are stored
            const bool okx = col >= 4 && col < 4 + R8_OW && (INTERIOR || gx < W);
            const bool oky0 = INTERIOR || gy0 < H, oky1 = INTERIOR || gy0 + 1 < H, okx1 = INTERIOR || gx + 1 < W;
            if (INTERIOR) {
                if (okx) {
                    float* o = gout + ((size_t)gy0 * W + gx) * 8 + ch;
                    *reinterpret_cast<f32x4*>(o) = v0;
                    *reinterpret_cast<f32x4*>(o + (size_t)W * 8) = v1;
                }
            } else {
                if (okx && oky0) *reinterpret_cast<f32x4*>(gout + ((size_t)gy0 * W + gx) * 8 + ch) = v0;
                if (okx && oky1) *reinterpret_cast<f32x4*>(gout + ((size_t)(gy0 + 1) * W + gx) * 8 + ch) = v1;
            }
            if (POOL && gpool) {
                // 2x2 max: rows in registers, the x neighbour (e = 1) sits in lane ^ 32; windows never straddle tiles
                f32x4 m = oky1 ? f32x4{fmaxf(v0.x, v1.x), fmaxf(v0.y, v1.y), fmaxf(v0.z, v1.z), fmaxf(v0.w, v1.w)} : v0;
                f32x4 o;
                o.x = from_upper_half(m.x); o.y = from_upper_half(m.y); o.z = from_upper_half(m.z); o.w = from_upper_half(m.w);   // only lanes < 32 (e == 0) use it
                if (e == 0 && okx && oky0) {
                    if (okx1) { m.x = fmaxf(m.x, o.x); m.y = fmaxf(m.y, o.y); m.z = fmaxf(m.z, o.z); m.w = fmaxf(m.w, o.w); }
                    const int Wp = (W + 1) >> 1;
                    *reinterpret_cast<f32x4*>(gpool + ((size_t)(gy0 >> 1) * Wp + (gx >> 1)) * 8 + ch) = m;
                }
            }
        }
    };
    // two fragment sets in flight: a unit's LDS reads are requested one unit ahead of its MFMAs (straight-line code,
    // no loop-carried arrays: a copy between the sets would make the wave wait for the reads at once)
    constexpr bool THIRD = NROWS > 2 * R8_WAVES;
    const bool has3 = THIRD && wave + 2 * R8_WAVES < NROWS;
    f32x4 bA[4][2], bB[4][2];
    load_unit(wave, bA);
    load_unit(wave + R8_WAVES, bB);                     // NROWS >= 16: every wave has two units
    // (unconditional and at one place for every wave: behind a run-time branch the compiler's wait counters become
    // conservative and the first MFMA would wait for the filter that was only just requested)
    if (!FINAL) r8_load_w(wnext, lane, An);
    __builtin_amdgcn_sched_barrier(0);                  // all requests go out first (the scheduler otherwise sinks the reads to their uses in some instantiations)
    unit(wave, bA);
    if (THIRD && has3) load_unit(wave + 2 * R8_WAVES, bA);
    unit(wave + R8_WAVES, bB);
    if (THIRD && has3) unit(wave + 2 * R8_WAVES, bA);
}

// DOWN block of level 0: image (1 channel) -> d0 [H,W,8] (+ maxpool2)
template <bool BF = false>
__global__ __launch_bounds__(R8_THREADS) __attribute__((amdgpu_waves_per_eu(2, 2))) void res8_down_kernel(const Res8Args a) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float* IMG = sm;                                         // [24][76]
    float* T = IMG + R8_FH * R8_IMGP;                        // frame rows 1..22  [22][72][8]
    float* R0 = T + 22 * R8_PITCH * 8;                       // frame rows 2..21  [20][72][8]
    float* R1 = R0 + 20 * R8_PITCH * 8;                      // frame rows 3..20  [18][72][8]
    __shared__ float w1s[9 * 8 + 8];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
#ifdef ASEP_R8_TIMELINE
    int tl_n = 0;
#endif
    const int kk = lane >> 4;
    if (tid < 72) w1s[tid] = a.w1[tid];
    if (tid < 8) w1s[72 + tid] = a.b1[tid];
    f32x4 Wa[6], Wb[6];                                      // the current / next stage's 8->8 filter (pixel-pair A fragments)
    const int ch = (kk & 1) * 4;
    const f32x4 bias0 = *reinterpret_cast<const f32x4*>(a.br + 0 + ch);
    const f32x4 bias1 = *reinterpret_cast<const f32x4*>(a.br + 8 + ch);
    const f32x4 bias2 = *reinterpret_cast<const f32x4*>(a.br + 16 + ch);

    constexpr int NPRE = (R8_FH * R8_IMGP + R8_THREADS - 1) / R8_THREADS;
    float pre[NPRE], pre_mean = 0.f, pre_inv = 1.f;
    unsigned pre_mask = 0;
    // frame = one 24-row window of a work unit: (tile, pass) -> image rows [(tyb * NP + pass) * OH - 4, +24)
    auto image_load = [&](int tile_id, int pass) {          // next frame's image values -> registers (in flight under the MFMA stages)
        int pi = 0;
        pi = prob_of_tile_search(a, tile_id);
        const Res8Prob& Q = a.p[pi];
        const int t = tile_id - Q.tile_begin;
        const int tyb = t / Q.tiles_x, txb = t - tyb * Q.tiles_x;
        const int qy0 = (tyb * R8_NP + pass) * R8_OH - 4, qx0 = txb * R8_OW - 4;
        // the standardisation is applied when the registers are written to LDS, not here: arithmetic on the loaded value
        // would make this phase wait for the loads it has only just issued (5 k cycles per pass in the timeline)
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
            // a frame that lies inside the image needs no zero masks (scalar condition; carried passes only, to bound the code size)
            const bool interior = !first && fy0 >= 0 && fy0 + R8_FH <= H && fx0 - 2 >= 0 && fx0 + R8_PITCH + 2 <= W;
            R8_MARK();   // 0 pass start
            __syncthreads();                                 // previous pass / tile finished with all LDS buffers
            R8_MARK();   // 1 barrier
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
            R8_MARK();   // 2 image tile + carried rows in LDS
            r8_load_w(a.wr, lane, Wa);                       // convR_0's filter flies under the scalar conv1
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
                *reinterpret_cast<f32x4*>(T + r * R8_PITCH * 8 + r8_px(c, 0)) = lo;
                *reinterpret_cast<f32x4*>(T + r * R8_PITCH * 8 + r8_px(c, 1)) = hi;
            }
            R8_MARK();   // 3 conv1 (VALU) done
            __syncthreads();
            R8_MARK();   // 4 barrier
            if (first) res8_stage<20, true, false, false, BF>(T, 1, R0, 2, 2, 2, Wa, bias0, wave, lane, fy0, fx0, H, W, nullptr, 0, nullptr, nullptr, a.wr + 6 * 64, Wb);
            else if (interior) res8_stage<16, true, false, false, BF, true>(T, 1, R0, 2, 6, 2, Wa, bias0, wave, lane, fy0, fx0, H, W, nullptr, 0, nullptr, nullptr, a.wr + 6 * 64, Wb);
            else res8_stage<16, true, false, false, BF>(T, 1, R0, 2, 6, 2, Wa, bias0, wave, lane, fy0, fx0, H, W, nullptr, 0, nullptr, nullptr, a.wr + 6 * 64, Wb);
            R8_MARK();   // 5 stage0
            __syncthreads();
            R8_MARK();   // 6 barrier
            if (first) res8_stage<18, false, false, false, BF>(R0, 2, R1, 3, 3, 3, Wb, bias1, wave, lane, fy0, fx0, H, W, nullptr, 0, nullptr, nullptr, a.wr + 12 * 64, Wa);
            else if (interior) res8_stage<16, false, false, false, BF, true>(R0, 2, R1, 3, 5, 3, Wb, bias1, wave, lane, fy0, fx0, H, W, nullptr, 0, nullptr, nullptr, a.wr + 12 * 64, Wa);
            else res8_stage<16, false, false, false, BF>(R0, 2, R1, 3, 5, 3, Wb, bias1, wave, lane, fy0, fx0, H, W, nullptr, 0, nullptr, nullptr, a.wr + 12 * 64, Wa);
            R8_MARK();   // 7 stage1
            __syncthreads();
            R8_MARK();   // 8 barrier
            if (more_passes) image_load(tile_id, pass + 1);
            else if (has_next) image_load(next_id, 0);
            R8_MARK();   // 8b image prefetch issued
            if (interior) res8_stage<16, false, true, true, BF, true>(R1, 3, nullptr, 4, 4, 4, Wa, bias2, wave, lane, fy0, fx0, H, W, T, 1, P.out, P.pool, nullptr, Wb);
            else res8_stage<16, false, true, true, BF>(R1, 3, nullptr, 4, 4, 4, Wa, bias2, wave, lane, fy0, fx0, H, W, T, 1, P.out, P.pool, nullptr, Wb);
            R8_MARK();   // 9 stage2
        }
        tile_id = next_id;
    }
}

// UP block of level 0 (ARU_v1.py:262-281): t = conv1(concat[skip, deconv]) ; 3 x convR ; + t ; ReLU.
// The 16-channel concatenation is consumed as two 8-channel passes through one LDS input tile (skip, then the
// deconvolution output) that accumulate into the same registers; afterwards that tile buffer holds r1.
template <bool BF = false>
__global__ __launch_bounds__(R8_THREADS) __attribute__((amdgpu_waves_per_eu(2, 2))) void res8_up_kernel(const Res8Args a) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float* Pb = sm;                                          // frame rows 0..23  [24][72][8]  (later r1: rows 3..20)
    float* T = Pb + R8_FH * R8_PITCH * 8;                    // frame rows 1..22  [22][72][8]
    float* R0 = T + 22 * R8_PITCH * 8;                       // frame rows 2..21  [20][72][8]
    float* R1K = R0 + 20 * R8_PITCH * 8;                     // two r1 rows parked between passes [2][72][8]
    int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // scalar: unit numbers and their branches stay on the SALU
#ifdef ASEP_R8_TIMELINE
    int tl_n = 0;
    if (blockIdx.x == 0 && threadIdx.x == 0) { r8_clk[0] = clock64(); r8_clk[1] = wall_clock64(); }
#endif
    int j = lane & 15, kk = lane >> 4;
    int e = kk >> 1, ch = (kk & 1) * 4;
    const f32x4* w1 = reinterpret_cast<const f32x4*>(a.w1);  // [2 sources][6 chunks][64 lanes]
    // halo tile loader: thread -> (row group rg = tid / 144, slot cs = tid % 144 = pixel column x 2 halves), rows rg + 3k:
    // one division per pass, then a constant row stride (432 of the 512 threads load, 8 x 16 B each)
    constexpr int ROWV = R8_PITCH * 2;                       // f32x4 per LDS row
    constexpr int NPF = R8_FH / 3;
    f32x4 pf[NPF];
    int rg = tid / ROWV, cs = tid - rg * ROWV;
    // carried: only frame rows 6..23 are needed (conv1 then reads rows 6..23 only), i.e. k >= 2
    auto tile_load = [&](const float* __restrict__ g, int H_, int W_, int qy0, int qx0, bool carried) {   // 8-channel halo tile -> registers
        const int gx = qx0 + (cs >> 1);
        const bool okc = rg < 3 && gx >= 0 && gx < W_;
        const float* __restrict__ gp = g + ((ptrdiff_t)gx * 8 + (cs & 1) * 4);
#pragma unroll
        for (int k = 0; k < NPF; ++k) {
            if (k < 2 && carried) continue;
            const int gy = qy0 + rg + 3 * k;
            f32x4 v = f32x4{0.f, 0.f, 0.f, 0.f};
            if (okc && gy >= 0 && gy < H_) v = *reinterpret_cast<const f32x4*>(gp + (ptrdiff_t)gy * W_ * 8);
            pf[k] = v;
        }
    };
    auto tile_store = [&](bool carried) {                    // registers -> tile buffer
        if (rg < 3) {
#pragma unroll
            for (int k = 0; k < NPF; ++k) {
                if (k < 2 && carried) continue;
                *reinterpret_cast<f32x4*>(Pb + (rg + 3 * k) * R8_PITCH * 8 + r8_px(cs >> 1, cs & 1)) = pf[k];
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
    // the stage biases are re-read from LDS at each use (VGPRs that would otherwise live across the whole loop; a
    // global load here would also make the stage epilogue wait for the tile prefetch that is in flight)
    __shared__ __attribute__((aligned(16))) float bsh[32];
    if (tid < 24) bsh[tid] = a.br[tid];
    if (tid < 8) bsh[24 + tid] = a.b1[tid];
    __syncthreads();
    auto bias_of = [&](int s) { return *reinterpret_cast<const f32x4*>(bsh + 8 * s + ch); };
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
            // conv1 covers frame rows 1..22 in the first pass and only the 16 new rows 7..22 afterwards
            const int c1_row = first ? 1 : 7, c1_units = first ? 22 : 16;
            R8_MARK();   // 0 pass start
            // everything derived from the thread index is recomputed per pass: hoisted out of the persistent loop it
            // occupied (and spilled) dozens of VGPRs, and every scratch reload waited for the tile prefetch in flight
            asm volatile("" : "+v"(tid));
            lane = tid & 63; j = lane & 15; kk = lane >> 4; e = kk >> 1; ch = (kk & 1) * 4;
            rg = tid / ROWV; cs = tid - rg * ROWV;

            // a frame that lies inside the image needs no zero masks (scalar condition; carried passes only, to bound the code size)
            const bool interior = !first && fy0 >= 0 && fy0 + R8_FH <= H && fx0 >= 0 && fx0 + R8_PITCH <= W;
            const f32x4 biasT = bias_of(3);
            // t accumulators of this wave's pair-units (<= 3 per wave)
            f32x4 tacc[3][2];
#pragma unroll
            for (int q = 0; q < 3; ++q) { tacc[q][0] = biasT; tacc[q][1] = biasT; }   // conv1's bias is the initial value

            f32x4 Wa[6], Wb[6];                               // the current / next filter (pixel-pair A fragments)
            // one 8-channel half of conv1 out of the tile buffer, accumulated into tacc
            auto conv1_half = [&](const f32x4 (&Aw)[6]) {
                auto load_unit = [&](int pu, f32x4 (&b)[4][2]) {
                    const int rp = pu >> 1, nt = pu & 1;
                    r8_load_frags(Pb + (c1_row + 2 * rp - 1) * R8_PITCH * 8, nt * 32 + 2 * j + e, ch >> 2, b);
                };
                const bool has3 = wave + 2 * R8_WAVES < c1_units;
                f32x4 bA[4][2], bB[4][2];
                load_unit(wave, bA);
                load_unit(wave + R8_WAVES, bB);
                __builtin_amdgcn_sched_barrier(0);
                r8_mma<false, BF>(Aw, bA, tacc[0][0], tacc[0][1]);
                if (has3) load_unit(wave + 2 * R8_WAVES, bA);
                r8_mma<false, BF>(Aw, bB, tacc[1][0], tacc[1][1]);
                if (has3) r8_mma<false, BF>(Aw, bA, tacc[2][0], tacc[2][1]);
            };
            // ---- conv1, skip half ----
            __syncthreads();                                 // the tile buffer is free (previous pass / previous tile done)
            R8_MARK();   // 1 barrier passed
            if (!first) {
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
            r8_load_w(w1, lane, Wa);
            tile_load(P.in1, H, W, fy0, fx0, !first);  // the deconv half flies while the skip half is multiplied
            __syncthreads();
            R8_MARK();   // 2 tile in LDS
            conv1_half(Wa);
            r8_load_w(w1 + 6 * 64, lane, Wb);                // the deconv half's filter flies across the refill of the tile buffer
            R8_MARK();   // 3 conv1 half done
            // ---- conv1, deconv half ----
            __syncthreads();
            R8_MARK();   // 4 barrier passed
            tile_store(!first);
            __syncthreads();
            R8_MARK();   // 5 tile in LDS
            conv1_half(Wb);
            r8_load_w(a.wr, lane, Wa);                       // convR_0's filter flies under the t write
            R8_MARK();   // 6 conv1 half done
            // ---- write raw t (identity activation), zero outside the image ----
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                const int pu = wave + q * R8_WAVES;
                if (pu < c1_units) {
                    const int rp = pu >> 1, nt = pu & 1;
                    const int row0 = c1_row + 2 * rp, col = 1 + nt * 32 + 2 * j + e;
                    float* o = T + (row0 - 1) * R8_PITCH * 8 + r8_px(col, ch >> 2);
                    if (interior) {
                        *reinterpret_cast<f32x4*>(o) = tacc[q][0];
                        *reinterpret_cast<f32x4*>(o + R8_PITCH * 8) = tacc[q][1];
                    } else {
                        const int gy0 = fy0 + row0, gx = fx0 + col;
                        const bool okx = gx >= 0 && gx < W;
                        const f32x4 z = f32x4{0.f, 0.f, 0.f, 0.f};
                        *reinterpret_cast<f32x4*>(o) = (okx && gy0 >= 0 && gy0 < H) ? tacc[q][0] : z;
                        *reinterpret_cast<f32x4*>(o + R8_PITCH * 8) = (okx && gy0 + 1 >= 0 && gy0 + 1 < H) ? tacc[q][1] : z;
                    }
                }
            }
            R8_MARK();   // 7 t written
            __syncthreads();
            R8_MARK();   // 8
            if (first) res8_stage<20, true, false, false, BF>(T, 1, R0, 2, 2, 2, Wa, bias_of(0), wave, lane, fy0, fx0, H, W, nullptr, 0, nullptr, nullptr, a.wr + 6 * 64, Wb);
            else if (interior) res8_stage<16, true, false, false, BF, true>(T, 1, R0, 2, 6, 2, Wa, bias_of(0), wave, lane, fy0, fx0, H, W, nullptr, 0, nullptr, nullptr, a.wr + 6 * 64, Wb);
            else res8_stage<16, true, false, false, BF>(T, 1, R0, 2, 6, 2, Wa, bias_of(0), wave, lane, fy0, fx0, H, W, nullptr, 0, nullptr, nullptr, a.wr + 6 * 64, Wb);
            R8_MARK();   // 9 stage0 done
            __syncthreads();
            R8_MARK();   // 10
            if (first) {
                res8_stage<18, false, false, false, BF>(R0, 2, Pb, 3, 3, 3, Wb, bias_of(1), wave, lane, fy0, fx0, H, W, nullptr, 0, nullptr, nullptr, a.wr + 12 * 64, Wa);
            } else {
                if (interior) res8_stage<16, false, false, false, BF, true>(R0, 2, Pb, 3, 5, 3, Wb, bias_of(1), wave, lane, fy0, fx0, H, W, nullptr, 0, nullptr, nullptr, a.wr + 12 * 64, Wa);
                else res8_stage<16, false, false, false, BF>(R0, 2, Pb, 3, 5, 3, Wb, bias_of(1), wave, lane, fy0, fx0, H, W, nullptr, 0, nullptr, nullptr, a.wr + 12 * 64, Wa);
                // r1 rows 3,4 of this frame = rows 19,20 of the previous one (the tile buffer is free of conv1 readers here)
                for (int i = tid; i < 2 * ROWV; i += R8_THREADS)
                    reinterpret_cast<f32x4*>(Pb)[i] = reinterpret_cast<const f32x4*>(R1K)[i];
            }
            R8_MARK();   // 11 stage1 done
            __syncthreads();
            R8_MARK();   // 12
            if (more_passes) {                               // the next frame's skip half flies under the last stage
                tile_load(P.img, H, W, fy0 + R8_OH, fx0, true);
            } else if (has_next) {
                int qi = 0;
                while (qi + 1 < a.nprob && next_id >= a.p[qi + 1].tile_begin) ++qi;
                const Res8Prob& Q = a.p[qi];
                const int tq = next_id - Q.tile_begin;
                const int qyb = tq / Q.tiles_x, qxb = tq - qyb * Q.tiles_x;
                tile_load(Q.img, Q.H, Q.W, qyb * R8_NP * R8_OH - 4, qxb * R8_OW - 4, false);
            }
            if (more_passes) {
                // park r1 rows 19,20 (tile-buffer rows 16,17) for the next pass: only read, like the stage below
                for (int i = tid; i < 2 * ROWV; i += R8_THREADS)
                    reinterpret_cast<f32x4*>(R1K)[i] = reinterpret_cast<const f32x4*>(Pb)[16 * ROWV + i];
            }
            R8_MARK();   // 13 prefetch issued
            if (interior) res8_stage<16, false, true, false, BF, true>(Pb, 3, nullptr, 4, 4, 4, Wa, bias_of(2), wave, lane, fy0, fx0, H, W, T, 1, P.out, nullptr, nullptr, Wb);
            else res8_stage<16, false, true, false, BF>(Pb, 3, nullptr, 4, 4, 4, Wa, bias_of(2), wave, lane, fy0, fx0, H, W, T, 1, P.out, nullptr, nullptr, Wb);
            R8_MARK();   // 14 stage2 done
        }
        tile_id = next_id;
    }
#ifdef ASEP_R8_TIMELINE
    if (blockIdx.x == 0 && threadIdx.x == 0) { r8_clk[2] = clock64(); r8_clk[3] = wall_clock64(); }
#endif
}

constexpr size_t R8_UP_LDS = (size_t)((R8_FH + 22 + 20 + 2) * R8_PITCH * 8) * sizeof(float);
constexpr size_t R8_DOWN_LDS = (size_t)(R8_FH * R8_IMGP + (22 + 20 + 18) * R8_PITCH * 8) * sizeof(float);

}  // namespace asep
