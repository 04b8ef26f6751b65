// ARU-Net device kernels for gfx950 (CDNA4).  fp32 path: implicit-GEMM convolutions on the
// f32 MFMA (v_mfma_f32_16x16x4_f32), NHWC activations, LDS-staged halo tiles.
//
// Reference semantics implemented here (file:line in /root/reference):
//   layers.py:191-247   conv2d (SAME, stride 1) + bias + activation      -> conv_mfma_kernel / conv_c1_kernel
//   layers.py:342-367   deconv2d (conv2d_transpose 3x3, stride 2, SAME)   -> deconv_mfma_kernel
//   layers.py:526-544   avg/max pool 2x2 s2 SAME                          -> avgpool2_kernel / maxpool2_kernel
//   layers.py:716-720   upsample_simple (NN upsample + channel sum)       -> chansum_kernel + combine_kernel
//   ARU_v1.py:141-160   softmax over scales, weighted sum, 4x4 logits conv -> combine_kernel
//   net_post_processing_helper.py:75-78 + separator_net_post_processor.py:147 -> combine_kernel epilogue
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>

namespace asep {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));

// bf16 variant (asep_aru_cfg.compute_dtype = 1): activations and weights stay fp32 in HBM / LDS; a lane's 16-byte
// fragment (4 consecutive k-slots) is rounded to 4 bf16 (round-to-nearest-even, v_cvt_pk_bf16_f32) right before the
// MFMA, and ONE v_mfma_f32_16x16x16_bf16 (fp32 accumulate) replaces the four v_mfma_f32_16x16x4_f32 of a K chunk:
// the K-slot numbering (slot = 4*kk + r) is exactly that instruction's operand layout.
__device__ __forceinline__ s16x4 bf16pack(f32x4 v) {
    // two-element vector conversions select one v_cvt_pk_bf16_f32 each (and, unlike inline asm, let the compiler
    // schedule the VALU-write -> MFMA-read hazard)
    const bf16x2_t lo = __builtin_convertvector(f32x2_t{v.x, v.y}, bf16x2_t);
    const bf16x2_t hi = __builtin_convertvector(f32x2_t{v.z, v.w}, bf16x2_t);
    const u32x2 p = {__builtin_bit_cast(unsigned, lo), __builtin_bit_cast(unsigned, hi)};
    return __builtin_bit_cast(s16x4, p);
}
// two floats -> two bf16 (round-to-nearest-even) in one dword: the storage format of the native bf16 path (bf16_kernels.h)
__device__ __forceinline__ unsigned bf16x2_of(float a, float b) {
    return __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2_t{a, b}, bf16x2_t));
}
__device__ __forceinline__ f32x4 mfma_bf16(s16x4 a, s16x4 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a, b, c, 0, 0, 0);
}

// ------------------------------------------------------------------------------------------------
// Implicit-GEMM convolution, D[cout][pixel] = sum_k A[cout][k] * B[k][pixel]
//   A = packed weights (MFMA A operand, M = 16 output channels per tile),
//   B = activations from the LDS halo tile (N = 16 consecutive pixels of one row per tile),
//   K is consumed in chunks of 16 "slots": slot s = 4*kk + r  (kk = lane>>4, r = MFMA index 0..3),
//     C16 mode: chunk = (channel group g of 16, tap)        slot -> channel 16g + s
//     C8  mode: chunk = tap pair (Cin == 8)                  slot -> tap 2c + (s>>3), channel s&7
//     C12 mode: Cin == 12, KW == 4 (attention conv2): pixels are 12 floats apart in LDS, so the four taps of a filter ROW
//               are 48 contiguous floats = 3 chunks with no padding: chunk = (ky, c), slot s -> float 16c + s of that run
//               (kx = (16c + s) / 12, channel (16c + s) % 12): 12 instead of 16 chunks
// One lane's 16-byte LDS read / 16-byte weight load therefore feeds four MFMAs.
// XCD-aware block -> tile map of the one-shot kernels (aru_engine.hip, oneshot_map): workgroup b of a launch runs on XCD b % 8 and
// every XCD has its own L2, so row-major tile numbers put the tiles that share a halo on eight different L2s and each of them
// fetches the overlap from HBM (res8f_kernel<true>: 3.36 GB fetched for 1.57 GB of input, rocprofv3 FETCH_SIZE, round 3).
//   table: unit -> tile, the tiles of every problem in 4 x 8 super-tile order cut into eight chunks (one scalar load per block:
//          the tile's address arithmetic waits for it -- measured, that costs what the saved traffic of kernels that are not
//          HBM-bound gives back, and more);
//   chunk: the arithmetic form -- XCD x takes the x-th eighth of the ROW-MAJOR tile list (a horizontal band of the pages), block
//          b = tile (b & 7) * chunk + (b >> 3); grids are padded to 8 * chunk blocks, blocks beyond the last tile leave at once.
//          No memory access; left / right halos are shared inside the band, the row above is one band row back in the same L2.
// Neither: identity.  A negative result = padding block.
struct XcdMap {
    int chunk, total;      // chunk = ceil(total / 8), or 0: identity order
};
// XCD x (workgroup b runs on XCD b mod 8) takes the x-th eighth of the row-major tile list: left / right halos are shared inside the band,
// the rows above are one band row back in the same 4 MB L2.  Pure arithmetic: a table lookup at the head of a 5-10 us block cost 1.4 % of
// the step (DESIGN_LESSONS.md 27).  The grid is padded to 8 chunk blocks; surplus blocks leave at once.
__device__ __forceinline__ int sched_tile(const XcdMap& x) {
    if (x.chunk) {
        const int t = (int)(blockIdx.x & 7) * x.chunk + (int)(blockIdx.x >> 3);
        return t < x.total ? t : -1;
    }
    return (int)blockIdx.x;
}

// ------------------------------------------------------------------------------------------------
// One launch covers the same layer of several independent "problems" (pages x scale-space levels share
// the layer's weights): blockIdx.x walks the concatenated tile lists, blockIdx.y the output-channel blocks.
// Channel concat [in0, in1]: the source of channel c as ONE base pointer + an offset selected per lane.  Written as a select between
// the two pointers (`from0 ? P.in0 + c : P.in1 + (c - c0)`) the compiler selects between their ADDRESSES in the argument block and loads
// the chosen pointer through a vector address -- a dependent memory round trip in front of every halo load of every stage (found in the
// ISA of convb_kernel / res8f_kernel, round 5).  The distance of the two tensors is a scalar; the select is on integers, the base keeps
// its address space (global_load, not flat_load).
template <class T>
__device__ __forceinline__ const T* concat_src(const T* in0, const T* in1, int c, int c0) {
    const long d10 = (long)(reinterpret_cast<uintptr_t>(in1) - reinterpret_cast<uintptr_t>(in0));      // bytes (unused when there is no in1)
    const long off = c < c0 ? (long)c * (long)sizeof(T) : d10 + (long)(c - c0) * (long)sizeof(T);
    return reinterpret_cast<const T*>(reinterpret_cast<const char*>(in0) + off);
}
constexpr int MAXP = 12;
// Problem of work unit t in a launch whose problems' unit ranges start at a.p[i].tile_begin (increasing with i): the index is a
// COUNT over all starts, so the scalar loads are requested together and answered in one round trip.  (The search loop this replaces
// -- while (bid >= a.p[pi + 1].tile_begin) ++pi -- compiled to one s_load + s_waitcnt per step: up to eleven dependent scalar-cache
// round trips at the head of every block of a 12-problem launch, in kernels whose blocks live 5-10 us.)
template <class Args>
__device__ __forceinline__ int prob_of_tile(const Args& a, int t) {
    int pi = 0;
#pragma unroll
    for (int i = 1; i < MAXP; ++i) pi += (int)((unsigned)((i - a.nprob) & ~(t - a.p[i].tile_begin)) >> 31);   // i < nprob && t >= start, on the sign bits (scalar ALU)
    return pi;
}
// The search form, for the PERSISTENT level-0 kernels (res8v_* / res8_*): they look a problem up once per work unit inside their tile loop,
// most units lie in the first problems, and their filters live in the 16 KB scalar data cache (DESIGN_LESSONS 16) -- the count form's eleven
// argument-block lines per lookup evicted them (res8v_up_kernel +5.4 %, round 5); here the walk stops at the first start beyond t.
template <class Args>
__device__ __forceinline__ int prob_of_tile_search(const Args& a, int t) {
    int pi = 0;
    while (pi + 1 < a.nprob && t >= a.p[pi + 1].tile_begin) ++pi;
    return pi;
}
template <class Args>
__device__ __forceinline__ int prob_of_blk(const Args& a, int t) {
    int pi = 0;
#pragma unroll
    for (int i = 1; i < MAXP; ++i) pi += (int)((unsigned)((i - a.nprob) & ~(t - a.p[i].blk_begin)) >> 31);   // i < nprob && t >= start, on the sign bits (scalar ALU)
    return pi;
}
struct ConvProb {
    const float* in0;      // source 0, NHWC with c0 channels
    const float* in1;      // source 1 (channel concat behind source 0) or nullptr
    const float* res;      // residual NHWC [Ho,Wo,cout] added before the output activation, or nullptr
    float* out;            // NHWC [Ho,Wo,cout]
    float* pool;           // conv_mfma_kernel / conv_winor_kernel: maxpool2(out) [ceil(Ho/2), ceil(Wo/2), cout] or nullptr
    int H, W;              // input spatial size
    int Ho, Wo;            // output spatial size (== H, W for stride-1 conv)
    int tiles_x;           // tiles per row of this problem
    int tile_begin;        // first blockIdx.x of this problem
    int pbh, pbw;          // deconv: pad_before (rows, cols)
};
struct ConvArgs {
    ConvProb p[MAXP];
    int nprob;
    int total_tiles;       // sum of the problems' tile counts (persistent kernels walk them with a grid stride)
    const f32x4* wpk;      // packed weights [chunk][mtile][lane] x 4 floats
    const float* bias;     // [cout]
    int c0, c1;
    int cout;              // real output channels (store bound)
    int mtiles;            // number of 16-channel output tiles in wpk
    int groups;            // number of 16-channel input groups (C16 mode)
    int relu_in;           // apply ReLU while staging the input (pre-activation tensors)
    int relu_out;
    int act;               // graph variants (asep_aru_cfg.activation; relu_out is 0 then): 1 = elu, 2 = leaky (0.1) applied to the output.
                           // The ReLU graphs' fast epilogues are not touched: a launch with act != 0 takes the general ones.
    int skip_full;         // with p[].pool: do not store the unpooled output (nobody reads it)
    XcdMap xm;             // XCD-aware block -> tile map (sched_tile)
};

constexpr int CONV_TH = 8;
constexpr int CONV_TW = 32;
constexpr int CONV_NT = 4;     // n-tiles (16 pixels each) per wave; 4 waves -> 256 pixels per block

// the non-ReLU activations of ARU_v1.py:70-75, fused behind a convolution: 1 = elu (tf.nn.elu: x > 0 ? x : exp(x) - 1), 2 = leaky
// (layers.py:10-30: max(0, x) + 0.1 min(0, x)) -- the arithmetic of act_kernel, so fused and separate passes give the same bits
__device__ __forceinline__ float act1(float x, int mode) {
    return mode == 1 ? (x > 0.f ? x : expf(x) - 1.f) : fmaxf(x, 0.f) + 0.1f * fminf(x, 0.f);
}
__device__ __forceinline__ f32x4 act4(f32x4 v, int mode) { return f32x4{act1(v.x, mode), act1(v.y, mode), act1(v.z, mode), act1(v.w, mode)}; }

__device__ __forceinline__ f32x4 relu4(f32x4 v) {
    v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
    return v;
}
// value of the same lane in the lower / upper half of the wave (lane & 31 / lane | 32) for every lane: one
// v_permlane32_swap_b32 (gfx950) instead of a ds_bpermute through the LDS pipe and its lgkmcnt wait
__device__ __forceinline__ float from_lower_half(float v) {
    return __uint_as_float(__builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false)[0]);
}
__device__ __forceinline__ float from_upper_half(float v) {
    return __uint_as_float(__builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false)[1]);
}
// a - b as ONE packed instruction per two floats: hipcc selects v_pk_add_f32 for vector sums but a v_sub_f32 per element for
// vector differences, and on gfx950 every vector instruction of an fp32 MFMA kernel is paid out of the MFMA's own datapath
// time (DESIGN lesson 15).  Same result bit for bit (a + (-b)).
__device__ __forceinline__ f32x2 psub(f32x2 a, f32x2 b) {
    f32x2 r;
    asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ f32x4 psub(f32x4 a, f32x4 b) {
    const f32x2 lo = psub(f32x2{a.x, a.y}, f32x2{b.x, b.y}), hi = psub(f32x2{a.z, a.w}, f32x2{b.z, b.w});
    return f32x4{lo.x, lo.y, hi.x, hi.y};
}
// acc += {v, v} * w with v = element ODD of the register pair `pair` and w a scalar-register pair: one v_pk_fma_f32 with an
// op_sel broadcast.  Written as asm because instruction selection copies some odd elements to the low half of another pair
// first (a v_mov per two FMAs in combine_kernel).  res8v_kernels.h has the same helper for its stages.
template <int ODD>
__device__ __forceinline__ void pk_fma_bcast(f32x2& acc, f32x2 pair, f32x2 w) {
    if (ODD) asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,0,0]" : "+v"(acc) : "v"(pair), "s"(w));
    else asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1]" : "+v"(acc) : "v"(pair), "s"(w));
}
// value of lane ^ 1 (DPP quad_perm [1,0,3,2]: one VALU instruction, no LDS)
__device__ __forceinline__ float lane_xor1(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xF, 0xF, true));
}
__device__ __forceinline__ f32x4 max4(f32x4 a, f32x4 b) { return f32x4{fmaxf(a.x, b.x), fmaxf(a.y, b.y), fmaxf(a.z, b.z), fmaxf(a.w, b.w)}; }
// 2x2 max pool fused into a conv epilogue whose lanes hold pixel x (even/odd in lanes j, j ^ 1) of rows y (v0) and y + 1 (v1),
// y and the even lane's x even: the even lane stores max over the window's pixels that lie inside the image (ceil mode,
// like maxpool2_kernel).  Called by ALL lanes (the DPP exchange needs its neighbour active).
__device__ __forceinline__ void pool2_store(f32x4 v0, f32x4 v1, bool row1, bool col1, bool store, float* __restrict__ dst) {
    f32x4 mm = row1 ? max4(v0, v1) : v0;
    const f32x4 nb = f32x4{lane_xor1(mm.x), lane_xor1(mm.y), lane_xor1(mm.z), lane_xor1(mm.w)};
    if (store) *reinterpret_cast<f32x4*>(dst) = col1 ? max4(mm, nb) : mm;
}
// max(x, lim) on the bit patterns (one v_max_i32 per element; fmaxf costs a second, canonicalising v_max_f32):
// lim = 0 is ReLU, lim = INT_MIN the identity, so that a run-time "relu?" flag needs no branch
__device__ __forceinline__ f32x4 imax4(f32x4 v, int lim) {
    typedef int i32x4 __attribute__((ext_vector_type(4)));
    i32x4 i = __builtin_bit_cast(i32x4, v);
    i.x = max(i.x, lim); i.y = max(i.y, lim); i.z = max(i.z, lim); i.w = max(i.w, lim);
    return __builtin_bit_cast(f32x4, i);
}

// Software pipeline: the halo tile of channel group g+1 is fetched into registers while group g is multiplied
// out of LDS (two LDS buffers, one barrier per group); the weight fragments of tap t+1 are requested before the
// MFMAs of tap t are issued.
// MINB = blocks per CU the register allocation is bounded for: 2, or 4 for the 16 x 32-pixel single-group variant when the
// layer has no residual operand (the 32 registers of the residual prefetch go, 106 VGPRs, four 39 KB blocks per CU:
// 216 -> 203 us for a 16 -> 16 layer at 2250 x 1500; with a residual operand its latency would be exposed: 2 blocks)
template <int KH, int KW, int MT, bool C8, int TH = CONV_TH, bool DBUF = true, bool BF = false, bool C12 = false, int MINB = 2>
__global__ __launch_bounds__(256, MINB) void conv_mfma_kernel(const ConvArgs a) {
    static_assert(!C12 || (!C8 && KW * 12 % 16 == 0), "C12: dense rows of KW x 12 floats");
    constexpr int TW = CONV_TW, NT = TH / 2;             // TH rows x 2 column blocks of 16 pixels, 4 waves
    constexpr int LH = TH + KH - 1, LW = TW + KW - 1;
    constexpr int CPP = C8 ? 8 : (C12 ? 12 : 16);       // channels per pixel held in LDS
    constexpr int PT = (KH - 1) / 2, PL = (KW - 1) / 2;  // TF SAME: pad_before = (k-1)/2
    constexpr int TAPS = KH * KW;
    constexpr int SUBS = CPP / 4;
    constexpr int NV = LH * LW * SUBS;                   // float4 slots of one halo tile
    constexpr int STR = C12 ? 255 : 256;                 // loader stride: a multiple of SUBS, so that a thread keeps its sub-block
    constexpr int NLOAD = (NV + STR - 1) / STR;
    constexpr int LBUF = LH * LW * CPP;
    __shared__ __attribute__((aligned(16))) float lds[(DBUF ? 2 : 1) * LBUF];   // DBUF=false: one buffer, refilled between channel groups

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
    const int H = P.H, W = P.W;
    const float* __restrict__ in0 = P.in0;
    const float* __restrict__ in1 = P.in1;

    f32x4 acc[MT][NT];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int n = 0; n < NT; ++n) acc[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};

    // (16-channel mode: a B-fragment read takes the same 16-byte quad of 16 consecutive 64-byte pixel records and is 2-way
    // bank conflicted -- SQ_LDS_BANK_CONFLICT = 44-46 % of SQ_LDS_IDX_ACTIVE, 0 % in the 32-byte C8 mode.  Permuting the
    // quads by the pixel index removes the conflicts but costs ~5 VALU per fragment address inside the tap loop, where
    // the offsets are otherwise immediates: measured +6 % (16 -> 16, 16 x 32 tiles) to +17 % (8 x 32 tiles) slower, not
    // kept.  The fused level-0 kernels and the Winograd V image, where the permutation is a per-unit / per-thread
    // constant, do use it.)
    int nbase[NT];   // LDS float index of (row, col + j) of each n-tile of this wave
#pragma unroll
    for (int n = 0; n < NT; ++n) {
        const int id = wave * NT + n;
        nbase[n] = ((id >> 1) * LW + (id & 1) * 16 + j) * CPP;
    }

    // halo loader: slot idx = tid + 256 i -> (pixel, 4-channel sub-block); the pixel advances by a constant per i, so (ly, lx)
    // are updated incrementally (one division per thread instead of one per slot), and both inputs of a concatenation
    // are served by ONE predicated load through a selected pointer (the one-shot blocks are short: the ~1200
    // instructions of the naive prologue cost as much issue time as half of the MFMA phase)
    // NOTHING is computed from the loaded values here (the zero padding is applied from stmask when the registers are
    // written to LDS): any arithmetic or select on them would make the wave wait for the loads it has just issued,
    // which defeats the prefetch of the next channel group under the MFMAs
    f32x4 st[NLOAD];
    unsigned stmask = 0;
    const bool interior = y0 - PT >= 0 && y0 - PT + LH <= H && x0 - PL >= 0 && x0 - PL + LW <= W;
    constexpr int PSTEP = STR / SUBS, QD = PSTEP / LW, RD = PSTEP % LW;
    const int pix0 = tid / SUBS, sub0 = tid % SUBS;
    const int ly0 = pix0 / LW, lx0 = pix0 - ly0 * LW;
    auto stage_load = [&](int g) {
        const int c = g * 16 + sub0 * 4;
        const bool from0 = c < a.c0;
        const bool cok = from0 || c - a.c0 < a.c1;
        // branch-free: the address is clamped into the image (always valid), out-of-image slots are zeroed by a select
        const float* __restrict__ src = (from0 || !cok) ? in0 + (from0 ? c : 0) : in1 + (c - a.c0);
        const int cs = (from0 || !cok) ? a.c0 : a.c1;
        int ly = ly0, lx = lx0;
        if (interior) {
            // the whole halo window lies inside the image (scalar test, ~90 % of the tiles): no clamps, compares or selects
            const float* __restrict__ q = src + (size_t)((y0 - PT + ly0) * W + x0 - PL + lx0) * cs;
            const ptrdiff_t step = (ptrdiff_t)(QD * W + RD) * cs, wrap = (ptrdiff_t)(W - LW) * cs;
#pragma unroll
            for (int i = 0; i < NLOAD; ++i) {
                // threads past the last slot of the window read the tile's first pixel instead (a valid address; never stored)
                const bool have = tid < STR && (i * STR + STR - 1 < NV || tid + i * STR < NV);
                st[i] = *reinterpret_cast<const f32x4*>(have ? q : src);
                lx += RD;
                q += step;
                if (lx >= LW) { lx -= LW; q += wrap; }
            }
            stmask = cok ? ~0u : 0u;
            return;
        }
        stmask = 0;
#pragma unroll
        for (int i = 0; i < NLOAD; ++i) {
            const int gy = y0 - PT + ly, gx = x0 - PL + lx;
            const bool ok = cok && tid < STR && (i * STR + STR - 1 < NV || tid + i * STR < NV) && gy >= 0 && gy < H && gx >= 0 && gx < W;
            const int cy = min(max(gy, 0), H - 1), cx = min(max(gx, 0), W - 1);
            st[i] = *reinterpret_cast<const f32x4*>(src + (size_t)(cy * W + cx) * cs);
            stmask |= (ok ? 1u : 0u) << i;
            lx += RD; ly += QD;
            if (lx >= LW) { lx -= LW; ++ly; }
        }
    };
    const int relu_lim = a.relu_in ? 0 : (int)0x80000000;
    auto stage_store = [&](int buf) {
#pragma unroll
        for (int i = 0; i < NLOAD; ++i) {
            const int idx = tid + i * STR;
            if (idx < NV && tid < STR) {
                const f32x4 v = ((stmask >> i) & 1u) ? st[i] : f32x4{0.f, 0.f, 0.f, 0.f};
                *reinterpret_cast<f32x4*>(lds + buf * LBUF + idx * 4) = imax4(v, relu_lim);   // idx*4 == pix*CPP + sub*4
            }
        }
    };

    const int ngroups = (C8 || C12) ? 1 : a.groups;
    constexpr int CPG = C8 ? (TAPS + 1) / 2 : (C12 ? KH * (KW * 12 / 16) : TAPS);       // K chunks per channel group
    const int nchunks = ngroups * CPG;
    const f32x4* __restrict__ wbase = a.wpk + (size_t)mt0 * 64 + lane;
    const size_t wstride = (size_t)a.mtiles * 64;          // f32x4 elements per chunk

    stage_load(0);
    f32x4 af[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) af[m] = wbase[(size_t)m * 64];
    // residual operand of the epilogue: requested now so that its HBM latency hides under the MFMA phases
    constexpr bool RES_PREFETCH = (MT * NT <= 8) && MINB <= 2;
    f32x4 resv[RES_PREFETCH ? MT : 1][RES_PREFETCH ? NT : 1];
    if constexpr (RES_PREFETCH) {
        // branch-free: clamped addresses; without a residual operand the (ignored) values are read from the input
        const float* __restrict__ rp = P.res ? P.res : in0;
        const int rcs = P.res ? a.cout : a.c0;
#pragma unroll
        for (int n = 0; n < NT; ++n) {
            const int id = wave * NT + n;
            const int y = min(y0 + (id >> 1), P.Ho - 1), x = min(x0 + (id & 1) * 16 + j, P.Wo - 1);
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                const int c = (mt0 + m) * 16 + kk * 4;
                resv[m][n] = *reinterpret_cast<const f32x4*>(rp + ((size_t)y * P.Wo + x) * rcs + (c + 3 < rcs ? c : 0));
            }
        }
    }
    stage_store(0);
    __syncthreads();

    for (int g = 0; g < ngroups; ++g) {
        const float* __restrict__ lb = lds + (DBUF ? (g & 1) : 0) * LBUF;
        if (DBUF && g + 1 < ngroups) stage_load(g + 1);
#pragma unroll NT >= 8 ? 1 : CPG
        for (int t = 0; t < CPG; ++t) {
            const int chunk = g * CPG + t;
            f32x4 an[MT];
            if (chunk + 1 < nchunks) {
#pragma unroll
                for (int m = 0; m < MT; ++m) an[m] = wbase[(size_t)(chunk + 1) * wstride + (size_t)m * 64];
            } else {
#pragma unroll
                for (int m = 0; m < MT; ++m) an[m] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
            int toff;
            if constexpr (C12) {
                constexpr int CPR = KW * 12 / 16;              // chunks per filter row
                const int ky = t / CPR, c = t % CPR;
                toff = ky * LW * 12 + c * 16 + kk * 4;
            } else if constexpr (!C8) {
                const int ky = t / KW, kx = t % KW;
                toff = (ky * LW + kx) * 16 + kk * 4;
            } else {
                int tap = 2 * t + (kk >> 1);
                tap = tap < TAPS ? tap : TAPS - 1;       // padded slot: weights are zero, data must be finite
                const int ky = tap / KW, kx = tap - ky * KW;
                toff = (ky * LW + kx) * 8 + (kk & 1) * 4;
            }
            f32x4 bf[NT];
#pragma unroll
            for (int n = 0; n < NT; ++n) bf[n] = *reinterpret_cast<const f32x4*>(lb + nbase[n] + toff);
            if constexpr (BF) {
                s16x4 pa[MT], pb[NT];
#pragma unroll
                for (int m = 0; m < MT; ++m) pa[m] = bf16pack(af[m]);
#pragma unroll
                for (int n = 0; n < NT; ++n) pb[n] = bf16pack(bf[n]);
#pragma unroll
                for (int m = 0; m < MT; ++m)
#pragma unroll
                    for (int n = 0; n < NT; ++n) acc[m][n] = mfma_bf16(pa[m], pb[n], acc[m][n]);
            } else {
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int m = 0; m < MT; ++m)
#pragma unroll
                    for (int n = 0; n < NT; ++n)
                        acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[m][r], bf[n][r], acc[m][n], 0, 0, 0);
            }
#pragma unroll
            for (int m = 0; m < MT; ++m) af[m] = an[m];
        }
        if (DBUF && g + 1 < ngroups) stage_store((g + 1) & 1);
        if (DBUF) __syncthreads();
        if (!DBUF && g + 1 < ngroups) {
            // single LDS buffer, several channel groups (the four-blocks-per-CU variant on a 32-channel input): the next
            // group is fetched after this one's readers are done; its latency is covered by the other blocks of the CU
            __syncthreads();
            stage_load(g + 1);
            stage_store(0);
            __syncthreads();
        }
    }

    // ---- epilogue: D layout col = lane&15 -> pixel, row = 4*(lane>>4)+reg -> output channel ----
    float* __restrict__ out = P.out;
    const float* __restrict__ res = P.res;
    if (interior && a.cout % 16 == 0 && !a.act) {
        // interior tile, whole 16-channel output tiles: no bounds tests, one base pointer
        const int relu_o = a.relu_out ? 0 : (int)0x80000000;
        const size_t p00 = (size_t)(y0 + ((wave * NT) >> 1)) * P.Wo + x0 + j;   // NT is even: n-tile 0 of a wave is column block 0
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            const int c = (mt0 + m) * 16 + kk * 4;
            const f32x4 b4 = *reinterpret_cast<const f32x4*>(a.bias + c);
            float* __restrict__ o = out + p00 * a.cout + c;
            const float* __restrict__ rp = res + p00 * a.cout + c;
#pragma unroll
            for (int n = 0; n < NT; ++n) {
                const size_t d = ((size_t)(n >> 1) * P.Wo + (n & 1) * 16) * a.cout;
                f32x4 v = acc[m][n] + b4;
                if constexpr (RES_PREFETCH) {
                    if (res) v += resv[m][n];
                } else {
                    if (res) v += *reinterpret_cast<const f32x4*>(rp + d);
                }
                v = imax4(v, relu_o);
                acc[m][n] = v;
                if (!a.skip_full) *reinterpret_cast<f32x4*>(o + d) = v;
            }
            if (P.pool) {
                // a wave's n-tiles n, n + 2 are the same 16 columns of rows y, y + 1 (y even): the 2x2 max needs the row
                // partner from registers and the column partner from lane j ^ 1
                const int Wp = (P.Wo + 1) >> 1;
#pragma unroll
                for (int n = 0; n < NT; ++n) {
                    if (n & 2) continue;
                    const int id = wave * NT + n;
                    const int y = y0 + (id >> 1), x = x0 + (id & 1) * 16 + j;
                    pool2_store(acc[m][n], acc[m][n + 2], true, true, (j & 1) == 0,
                                P.pool + ((size_t)(y >> 1) * Wp + (x >> 1)) * a.cout + c);
                }
            }
        }
        return;
    }
#pragma unroll
    for (int n = 0; n < NT; ++n) {
        const int id = wave * NT + n;
        const int y = y0 + (id >> 1), x = x0 + (id & 1) * 16 + j;
        if (y >= P.Ho || x >= P.Wo) continue;
        const size_t p = (size_t)y * P.Wo + x;
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            const int c = (mt0 + m) * 16 + kk * 4;
            if (c >= a.cout) continue;
            f32x4 v = acc[m][n];
            if (c + 3 < a.cout) {
                v += *reinterpret_cast<const f32x4*>(a.bias + c);
                if constexpr (RES_PREFETCH) {
                    if (res) v += resv[m][n];
                } else {
                    if (res) v += *reinterpret_cast<const f32x4*>(res + p * a.cout + c);
                }
                if (a.relu_out) v = relu4(v);
                else if (a.act) v = act4(v, a.act);
                acc[m][n] = v;
                if (!a.skip_full) *reinterpret_cast<f32x4*>(out + p * a.cout + c) = v;
            } else {
                for (int r = 0; r < 4 && c + r < a.cout; ++r) {
                    float s = v[r] + a.bias[c + r];
                    if (res) s += res[p * a.cout + c + r];
                    if (a.relu_out) s = fmaxf(s, 0.f);
                    else if (a.act) s = act1(s, a.act);
                    out[p * a.cout + c + r] = s;
                }
            }
        }
    }
    if (P.pool) {
        // border tiles (the host requests the fused pool only for cout % 4 == 0): every lane takes part in the exchange,
        // pixels outside the image are excluded by the row / column flags
        const int Wp = (P.Wo + 1) >> 1;
#pragma unroll
        for (int n = 0; n < NT; ++n) {
            if (n & 2) continue;
            const int id = wave * NT + n;
            const int y = y0 + (id >> 1), x = x0 + (id & 1) * 16 + j;
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                const int c = (mt0 + m) * 16 + kk * 4;
                const bool ok = (j & 1) == 0 && y < P.Ho && x < P.Wo && c < a.cout;
                pool2_store(acc[m][n], acc[m][n + 2], y + 1 < P.Ho, x + 1 < P.Wo, ok,
                            P.pool + ((size_t)(y >> 1) * Wp + (x >> 1)) * a.cout + c);
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Winograd F(2x2,3x3) for the 3x3 stride-1 SAME convolutions with Cin % 16 == 0 and Cout % 16 == 0:
//     Y = A^T [ (G g G^T) (.) (B^T d B) ] A      per 4x4 input patch d -> 2x2 outputs,
// i.e. 16 independent [Cout x Cin] x [Cin x tiles] products instead of 9 taps -> 2.25x fewer MFMA FLOPs, fp32.
// One 256-thread block = 4 x 32 output pixels = 2 x 16 Winograd tiles, MT 16-channel output tiles.
//   transform role : thread (tile, channel pair) loads its 4x4x2 patch straight from global memory one channel
//                    group AHEAD (the loads fly during the MFMA phase), forms V = B^T d B and writes
//                    V[pos][tile][16 ch] into the other LDS buffer (2 x 32 KB, one barrier per group);
//   MFMA role      : wave w owns the 4 positions 4w..4w+3 for all 32 tiles: acc[4][MT][2 n-tiles];
//                    A = packed U = G g G^T fragments from global (L2), B = V from LDS;
//   output (once)  : accumulators are exchanged through the LDS image so that thread (tile, channel pair)
//                    holds all 16 positions, applies A^T . A, bias / residual / ReLU, stores 2x2 pixels.
// ------------------------------------------------------------------------------------------------
typedef float f32x2 __attribute__((ext_vector_type(2)));
constexpr int WINO_TH = 4;
constexpr int WINO_TW = 32;
#ifndef WINO_XFORM_AT
#define WINO_XFORM_AT 2
#endif
#ifndef WINO_STORE_AT
#define WINO_STORE_AT 1
#endif

#ifdef ASEP_WINO_TIMELINE   // development aid (scripts/ubench/wino_timeline.hip): per-wave cycle stamps of the first blocks
__device__ unsigned long long wino_tl[512][4][32];
#define WINO_MARK() do { if (blockIdx.x >= 4096 && blockIdx.x < 4608 && lane == 0 && tl_n < 32) wino_tl[blockIdx.x - 4096][wave][tl_n++] = clock64(); } while (0)
#else
#define WINO_MARK() do { } while (0)
#endif

template <int MT, bool BF = false>
__global__ __launch_bounds__(256, 2) void conv_wino_kernel(const ConvArgs a) {
    constexpr int TH = WINO_TH, TW = WINO_TW;
    constexpr int TILES = (TH / 2) * (TW / 2);            // 32 Winograd tiles
    constexpr int VBUF = 16 * TILES * 16;                 // floats per V image (32 KB)
    // input window of one channel group: 6 x 34 pixels x 16 channels, pixel pitch 20 floats (the 8-byte patch reads of
    // four neighbouring tiles then fall into different bank groups)
    constexpr int HH = TH + 2, HW = TW + 2, HP = 20;
    constexpr int NH = HH * HW * 4;                       // float4 slots of the window
    constexpr int NHL = (NH + 255) / 256;
    __shared__ __attribute__((aligned(16))) float V[2 * VBUF];
    __shared__ __attribute__((aligned(16))) float HALO[HH * HW * HP > TH * TW * 32 ? HH * HW * HP : TH * TW * 32];   // also the output tile

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
#ifdef ASEP_WINO_TIMELINE
    int tl_n = 0;
#endif
    WINO_MARK();   // 0 start
    const int j = lane & 15, kk = lane >> 4;
    const int bid = sched_tile(a.xm);
    if (bid < 0) return;
    int pi = 0;
    pi = prob_of_tile(a, bid);
    const ConvProb& P = a.p[pi];
    const int tile = bid - P.tile_begin;
    const int tyb = tile / P.tiles_x, txb = tile - tyb * P.tiles_x;
    const int x0 = txb * TW, y0 = tyb * TH, mt0 = blockIdx.y * MT;
    const int H = P.H, W = P.W;

    // transform role: tile tt (row-major in the 2 x 16 tile grid), channel pair cp (channels 2cp, 2cp+1 of the group)
    const int tt = tid >> 3, cp = tid & 7;
    // Bank-conflict-free V image: an MFMA B-fragment read takes, for 16 consecutive tiles, the same 16-byte channel quad
    // of each tile's 64-byte record -- at the natural layout those reads fall into two bank groups (rocprofv3:
    // SQ_LDS_BANK_CONFLICT = 45 % of SQ_LDS_IDX_ACTIVE, profiles/r2i_instruction_mix).  Quad q of tile t is therefore
    // stored in slot q ^ ((t >> 1) & 3): eight consecutive tiles then present eight different bank groups.  Both roles
    // apply the permutation as a per-thread constant (no cost inside the loops).
    const int cp_sw = ((((cp >> 1) ^ (((tt & 15) >> 1) & 3)) << 2) | ((cp & 1) << 1));   // float offset of the pair inside the record
    const int kk_sw = (kk ^ ((j >> 1) & 3)) << 2;                                        // float offset of the lane's quad

    f32x4 acc[4][MT][2];
#pragma unroll
    for (int p = 0; p < 4; ++p)
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int n = 0; n < 2; ++n) acc[p][m][n] = f32x4{0.f, 0.f, 0.f, 0.f};

    // The input window travels HBM -> registers (16 B per lane, coalesced, one channel group ahead) -> LDS, and the
    // 4x4 patches are read from LDS.  (Every thread fetching its own patch with sixteen 8-byte loads made the blocks
    // vector-memory-issue-bound: scripts/ubench/wino_timeline.hip showed 6100 + 2500 cycles spent just issuing them.)
    f32x4 st[NHL];
    unsigned stmask = 0;
    const int pix0 = tid >> 2, sub0 = tid & 3;
    const int hy0 = pix0 / HW, hx0 = pix0 - hy0 * HW;
    constexpr int QD = 64 / HW, RD = 64 % HW;              // the pixel index advances by 64 per slot
    const int relu_lim = a.relu_in ? 0 : (int)0x80000000;
    auto halo_load = [&](int g) {
        const int c = g * 16 + sub0 * 4;
        const bool from0 = c < a.c0;
        const float* __restrict__ src = concat_src(P.in0, P.in1, c, a.c0);
        const int cs = from0 ? a.c0 : a.c1;
        int hy = hy0, hx = hx0;
        stmask = 0;
#pragma unroll
        for (int i = 0; i < NHL; ++i) {
            const int gy = y0 - 1 + hy, gx = x0 - 1 + hx;  // SAME: pad 1
            const bool ok = gy >= 0 && gy < H && gx >= 0 && gx < W;
            const int cy = min(max(gy, 0), H - 1), cx = min(max(gx, 0), W - 1);   // clamped: always a valid address
            st[i] = *reinterpret_cast<const f32x4*>(src + (size_t)(cy * W + cx) * cs);
            stmask |= (ok ? 1u : 0u) << i;                 // (no select on the loaded value here: it would wait for the load)
            hx += RD; hy += QD;
            if (hx >= HW) { hx -= HW; ++hy; }
        }
    };
    auto halo_store = [&]() {
#pragma unroll
        for (int i = 0; i < NHL; ++i) {
            const int idx = tid + i * 256;
            if (i * 256 + 255 < NH || idx < NH) {
                const f32x4 v = ((stmask >> i) & 1u) ? st[i] : f32x4{0.f, 0.f, 0.f, 0.f};
                *reinterpret_cast<f32x4*>(HALO + (idx >> 2) * HP + sub0 * 4) = imax4(v, relu_lim);
            }
        }
    };
    auto transform_store = [&](int buf) {
        f32x2 d[4][4];
        const float* hb = HALO + ((2 * (tt >> 4)) * HW + 2 * (tt & 15)) * HP + cp * 2;
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int s2 = 0; s2 < 4; ++s2) d[r][s2] = *reinterpret_cast<const f32x2*>(hb + (r * HW + s2) * HP);
        float* vb = V + buf * VBUF + tt * 16 + cp_sw;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            // row r of B^T d, then times B, written straight out (keeps at most one row of temporaries live)
            f32x2 t0, t1, t2, t3;
            if (r == 0) { t0 = psub(d[0][0], d[2][0]); t1 = psub(d[0][1], d[2][1]); t2 = psub(d[0][2], d[2][2]); t3 = psub(d[0][3], d[2][3]); }
            else if (r == 1) { t0 = d[1][0] + d[2][0]; t1 = d[1][1] + d[2][1]; t2 = d[1][2] + d[2][2]; t3 = d[1][3] + d[2][3]; }
            else if (r == 2) { t0 = psub(d[2][0], d[1][0]); t1 = psub(d[2][1], d[1][1]); t2 = psub(d[2][2], d[1][2]); t3 = psub(d[2][3], d[1][3]); }
            else { t0 = psub(d[1][0], d[3][0]); t1 = psub(d[1][1], d[3][1]); t2 = psub(d[1][2], d[3][2]); t3 = psub(d[1][3], d[3][3]); }
            *reinterpret_cast<f32x2*>(vb + (r * 4 + 0) * TILES * 16) = psub(t0, t2);
            *reinterpret_cast<f32x2*>(vb + (r * 4 + 1) * TILES * 16) = t1 + t2;
            *reinterpret_cast<f32x2*>(vb + (r * 4 + 2) * TILES * 16) = psub(t2, t1);
            *reinterpret_cast<f32x2*>(vb + (r * 4 + 3) * TILES * 16) = psub(t1, t3);
        }
    };

    const f32x4* __restrict__ wbase = a.wpk + (size_t)mt0 * 64 + lane;
    const size_t wstride = (size_t)a.mtiles * 64;          // f32x4 per (group, pos)
    const int G = a.groups;

    halo_load(0);
    WINO_MARK();   // 1 first window requested
    f32x4 af[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) af[m] = wbase[((size_t)wave * 4) * wstride + (size_t)m * 64];
    halo_store();
    __syncthreads();
    transform_store(0);
    WINO_MARK();   // 2 first transform written
    __syncthreads();
    WINO_MARK();   // 3 barrier

    for (int g = 0; g < G; ++g) {
        const bool more = g + 1 < G;
        if (more) halo_load(g + 1);                        // global loads in flight during the MFMA phase
        WINO_MARK();   // g: window requested
        // ---- MFMA phase: positions 4*wave .. 4*wave+3 of group g -------------------------------------------
        const float* __restrict__ vcur = V + (g & 1) * VBUF;
        const f32x4* __restrict__ wg = wbase + ((size_t)g * 16 + wave * 4) * wstride;
        const f32x4* __restrict__ wnext = wbase + ((size_t)(more ? g + 1 : g) * 16 + wave * 4) * wstride;
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            f32x4 an[MT];
#pragma unroll
            for (int m = 0; m < MT; ++m)
                an[m] = p + 1 < 4 ? wg[(size_t)(p + 1) * wstride + (size_t)m * 64] : wnext[(size_t)m * 64];
            // the requests stay here, one position ahead of their use (left alone, the scheduler sinks them to just
            // before the MFMAs that need them and every position waits for an L2 round trip)
            __builtin_amdgcn_sched_barrier(0);
            const float* vb = vcur + ((wave * 4 + p) * TILES + j) * 16 + kk_sw;
            f32x4 bf[2];
#pragma unroll
            for (int n = 0; n < 2; ++n) bf[n] = *reinterpret_cast<const f32x4*>(vb + n * 16 * 16);
            if constexpr (BF) {
                s16x4 pa[MT], pb[2];
#pragma unroll
                for (int m = 0; m < MT; ++m) pa[m] = bf16pack(af[m]);
#pragma unroll
                for (int n = 0; n < 2; ++n) pb[n] = bf16pack(bf[n]);
#pragma unroll
                for (int m = 0; m < MT; ++m)
#pragma unroll
                    for (int n = 0; n < 2; ++n) acc[p][m][n] = mfma_bf16(pa[m], pb[n], acc[p][m][n]);
            } else {
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int m = 0; m < MT; ++m)
#pragma unroll
                    for (int n = 0; n < 2; ++n)
                        acc[p][m][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[m][r], bf[n][r], acc[p][m][n], 0, 0, 0);
            }
#pragma unroll
            for (int m = 0; m < MT; ++m) af[m] = an[m];
            WINO_MARK();   // g: position p multiplied
            // the next group's window goes registers -> LDS after position WINO_STORE_AT (the window buffer was last
            // read by the previous transform, a barrier ago), its transform into the other V buffer follows one
            // position later, in the shadow of the remaining MFMAs
            if (p == WINO_STORE_AT && more) { halo_store(); __syncthreads(); }
            if (p == WINO_XFORM_AT && more) { transform_store((g + 1) & 1); WINO_MARK(); }
        }
        __syncthreads();
        WINO_MARK();   // g: barrier
    }

    // ---- output phase: exchange through LDS (two m-tiles per round), inverse transform, then a second pass through
    //      LDS so that the tile leaves as 16-byte stores that cover whole pixels (the 8-byte stores of the
    //      (tile, channel pair) layout were half cache lines and twice as many instructions) -------------------------
    float* __restrict__ out = P.out;
    const float* __restrict__ res = P.res;
    constexpr int ROUNDS = (MT + 1) / 2;
    constexpr int MPR = MT >= 2 ? 2 : 1;                  // m-tiles per round
    constexpr int CPR = MPR * 16, QPP = CPR / 4;          // channels / channel quads per pixel and round
    constexpr int NSLOT = TH * TW * QPP / 256;            // float4 slots per thread and round (4 or 2)
    float* OT = HALO;                                     // output tile [4 x 32 pixels][CPR channels] (the window buffer is free by now)
#pragma unroll
    for (int rd = 0; rd < ROUNDS; ++rd) {
        // residual operand of this round in the store layout: requested before the LDS exchange so that its latency hides under it
        f32x4 rv[NSLOT];
        const int cbase = (mt0 + rd * 2) * 16;
#pragma unroll
        for (int i = 0; i < NSLOT; ++i) {
            const int k = tid + i * 256;
            const int pix = k / QPP, q = k % QPP;
            const int yy = y0 + (pix >> 5), xx = x0 + (pix & 31);
            rv[i] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (res && yy < P.Ho && xx < P.Wo)
                rv[i] = *reinterpret_cast<const f32x4*>(res + ((size_t)yy * P.Wo + xx) * a.cout + cbase + q * 4);
        }
        if (rd > 0) __syncthreads();                       // readers of the previous round are done
#pragma unroll
        for (int h = 0; h < MPR; ++h) {
            const int m = rd * 2 + h;
            if (m < MT) {
                // a wave holds one ROW of the 4 x 4 position grid: the column half of A^T M A (positions s = 0..3 -> output
                // columns dx = 0, 1) is formed in registers, so only 2 instead of 4 vectors per (m, n) cross the LDS
#pragma unroll
                for (int n = 0; n < 2; ++n) {
                    const f32x4 z0 = acc[0][m][n] + acc[1][m][n] + acc[2][m][n];
                    const f32x4 z1 = psub(psub(acc[1][m][n], acc[2][m][n]), acc[3][m][n]);
                    float* xb = V + h * VBUF + ((wave * 2) * TILES + n * 16 + j) * 16 + kk_sw;
                    *reinterpret_cast<f32x4*>(xb) = z0;
                    *reinterpret_cast<f32x4*>(xb + TILES * 16) = z1;
                }
            }
        }
        WINO_MARK();   // exchange written
        __syncthreads();
        WINO_MARK();   // barrier
#pragma unroll
        for (int h = 0; h < MPR; ++h) {
            const int m = rd * 2 + h;
            if (m >= MT) continue;
            const float* mb = V + h * VBUF + tt * 16 + cp_sw;
            // row half of the inverse transform: rows r = 0..3 (one per wave) -> output rows dy = 0, 1
            // 2 x 2 output pixels of tile tt, channels 2cp, 2cp+1 of m-tile h -> output tile
            // (pixels of odd tiles keep their two 16-channel halves swapped: the four tiles of a half-wave then use both
            // halves of the 32 banks instead of all writing the same 16)
            float* ob = OT + ((2 * (tt >> 4)) * TW + 2 * (tt & 15)) * CPR + ((h * 16 + cp * 2) ^ (CPR == 32 ? (tt & 1) << 4 : 0));
#pragma unroll
            for (int dx = 0; dx < 2; ++dx) {
                const f32x2 z0 = *reinterpret_cast<const f32x2*>(mb + (0 * 2 + dx) * TILES * 16);
                const f32x2 z1 = *reinterpret_cast<const f32x2*>(mb + (1 * 2 + dx) * TILES * 16);
                const f32x2 z2 = *reinterpret_cast<const f32x2*>(mb + (2 * 2 + dx) * TILES * 16);
                const f32x2 z3 = *reinterpret_cast<const f32x2*>(mb + (3 * 2 + dx) * TILES * 16);
                *reinterpret_cast<f32x2*>(ob + dx * CPR) = z0 + z1 + z2;
                *reinterpret_cast<f32x2*>(ob + TW * CPR + dx * CPR) = psub(psub(z1, z2), z3);
            }
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < NSLOT; ++i) {
            const int k = tid + i * 256;
            const int pix = k / QPP, q = k % QPP;
            const int yy = y0 + (pix >> 5), xx = x0 + (pix & 31);
            const int co = cbase + q * 4;
            if (yy < P.Ho && xx < P.Wo && co < a.cout) {
                f32x4 v = *reinterpret_cast<const f32x4*>(OT + pix * CPR + ((q * 4) ^ (CPR == 32 ? ((pix >> 1) & 1) << 4 : 0))) +
                          *reinterpret_cast<const f32x4*>(a.bias + co);
                if (res) v += rv[i];
                if (a.relu_out) v = relu4(v);
                else if (a.act) v = act4(v, a.act);
                *reinterpret_cast<f32x4*>(out + ((size_t)yy * P.Wo + xx) * a.cout + co) = v;
            }
        }
    }
    WINO_MARK();   // done
}

// ------------------------------------------------------------------------------------------------
// Register-resident Winograd F(2x2,3x3) (MT = 2: the 32-channel level).  conv_wino_kernel above is bound by LDS traffic
// (3 KB per output pixel: V written and read, accumulators exchanged, output tile), not by MFMAs or memory.  Here a
// lane keeps a whole tile column in registers:
//   wave w = (tile row n = w & 1, m-tile mh = w >> 1);  lane (j, kk) = (tile j of that row, channel quad kk)
//   * reads its 4x4 patch x 4 channels from the LDS window (16 x ds_read_b128), forms V = B^T d B for ALL 16 positions
//     in registers: these are exactly the B fragments of the MFMAs (k-slot kk = channels 4kk..4kk+3);
//   * multiplies all 16 positions for its m-tile (A = packed U fragments from L2, one position ahead);
//   * holds M[16 positions] for (tile j, output channels 4kk..4kk+3) at the end: A^T M A, bias, residual, ReLU and
//     16-byte stores happen in registers.
// No V image, no accumulator exchange, no output tile: the only LDS traffic is the input window (13 KB written, 64 KB
// read per channel group instead of ~160 KB), at the price of forming V twice (once per m-tile wave).
// ------------------------------------------------------------------------------------------------
// MH = 16-channel output tiles per block: 2 (the 32-channel level, 4 x 32 pixel blocks) or 1 (the 16-channel level: the four
// waves are four tile rows of an 8 x 32 pixel block, V is formed once per tile)
// RESP = false: the layer has no residual operand; without the 16 registers of its prefetch the kernel is bounded for three
// blocks per CU instead of two
template <bool BF = false, int MH = 2, bool RESP = true>
__global__ __launch_bounds__(256, RESP ? 2 : 3) void conv_winor_kernel(const ConvArgs a) {
    constexpr int NTR = 4 / MH;                           // tile rows (waves per m-tile)
    constexpr int TH = 2 * NTR, TW = WINO_TW;
    constexpr int HH = TH + 2, HW = TW + 2, HP = 20;      // window: 6 x 34 pixels x 16 channels, pixel pitch 20 floats
    constexpr int NH = HH * HW * 4;                       // float4 slots of the window
    constexpr int NHL = (NH + 255) / 256;
    __shared__ __attribute__((aligned(16))) float HALO[HH * HW * HP];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j = lane & 15, kk = lane >> 4;
    const int n = wave % NTR, mh = wave / NTR;
    const int bid = sched_tile(a.xm);
    if (bid < 0) return;
    int pi = 0;
    pi = prob_of_tile(a, bid);
    const ConvProb& P = a.p[pi];
    const int tile = bid - P.tile_begin;
    const int tyb = tile / P.tiles_x, txb = tile - tyb * P.tiles_x;
    const int x0 = txb * TW, y0 = tyb * TH, mt = blockIdx.y * MH + mh;     // this wave's 16-channel output tile
    const int H = P.H, W = P.W;

    // window: HBM -> registers (16 B per lane, coalesced, one channel group ahead) -> LDS; raw values + validity mask
    f32x4 st[NHL];
    unsigned stmask = 0;
    const int pix0 = tid >> 2, sub0 = tid & 3;
    const int hy0 = pix0 / HW, hx0 = pix0 - hy0 * HW;
    constexpr int QD = 64 / HW, RD = 64 % HW;
    const int relu_lim = a.relu_in ? 0 : (int)0x80000000;
    auto halo_load = [&](int g) {
        const int c = g * 16 + sub0 * 4;
        const bool from0 = c < a.c0;
        const float* __restrict__ src = concat_src(P.in0, P.in1, c, a.c0);
        const int cs = from0 ? a.c0 : a.c1;
        int hy = hy0, hx = hx0;
        stmask = 0;
#pragma unroll
        for (int i = 0; i < NHL; ++i) {
            const int gy = y0 - 1 + hy, gx = x0 - 1 + hx;  // SAME: pad 1
            const bool ok = gy >= 0 && gy < H && gx >= 0 && gx < W;
            const int cy = min(max(gy, 0), H - 1), cx = min(max(gx, 0), W - 1);   // clamped: always a valid address
            st[i] = *reinterpret_cast<const f32x4*>(src + (size_t)(cy * W + cx) * cs);
            stmask |= (ok ? 1u : 0u) << i;
            hx += RD; hy += QD;
            if (hx >= HW) { hx -= HW; ++hy; }
        }
    };
    auto halo_store = [&]() {
#pragma unroll
        for (int i = 0; i < NHL; ++i) {
            const int idx = tid + i * 256;
            if (i * 256 + 255 < NH || idx < NH) {
                const f32x4 v = ((stmask >> i) & 1u) ? st[i] : f32x4{0.f, 0.f, 0.f, 0.f};
                *reinterpret_cast<f32x4*>(HALO + (idx >> 2) * HP + sub0 * 4) = imax4(v, relu_lim);
            }
        }
    };

    f32x4 acc[16];
#pragma unroll
    for (int p = 0; p < 16; ++p) acc[p] = f32x4{0.f, 0.f, 0.f, 0.f};

    const f32x4* __restrict__ wbase = a.wpk + (size_t)mt * 64 + lane;
    const size_t wstride = (size_t)a.mtiles * 64;          // f32x4 per (group, pos)
    const int G = a.groups;
    const float* hb = HALO + ((2 * n) * HW + 2 * j) * HP + kk * 4;   // this lane's patch origin in the window

    const int co = mt * 16 + kk * 4;                       // this lane's 4 output channels
    const int oy = y0 + 2 * n, ox = x0 + 2 * j;            // and the top-left output pixel of its tile
    halo_load(0);
    f32x4 af = wbase[0];
    // residual operand: requested once, up front (requested inside the group loop it became a loop-carried register set
    // that the compiler copied - and waited for - in every iteration)
    f32x4 rv[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int yy = min(oy + (q >> 1), P.Ho - 1), xx = min(ox + (q & 1), P.Wo - 1);
        rv[q] = (RESP && P.res) ? *reinterpret_cast<const f32x4*>(P.res + ((size_t)yy * P.Wo + xx) * a.cout + co) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
    halo_store();
    __syncthreads();

    for (int g = 0; g < G; ++g) {
        const bool more = g + 1 < G;
        // ---- V = B^T d B of this lane's tile, 4 channels wide, ONE ROW OF POSITIONS AT A TIME: row r of B^T d needs two
        //      rows of the patch (re-read from LDS: 32 instead of 16 reads per group, the LDS pipe is 9 % busy here), its
        //      four V values are the B fragments of the next 16 MFMAs.  Only 4 of the 16 V live at a time: with the
        //      residual prefetch gone too (RESP = false) the kernel fits three blocks per CU.
        const f32x4* __restrict__ wg = wbase + (size_t)g * 16 * wstride;
        const f32x4* __restrict__ wfirst_next = wbase + (size_t)(more ? g + 1 : g) * 16 * wstride;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            constexpr int RA[4] = {0, 1, 2, 1}, RB[4] = {2, 2, 1, 3};     // t[r] = d[RA] (- or +) d[RB]
            f32x4 t[4];
#pragma unroll
            for (int s2 = 0; s2 < 4; ++s2) {
                const f32x4 da = *reinterpret_cast<const f32x4*>(hb + (RA[r] * HW + s2) * HP);
                const f32x4 db = *reinterpret_cast<const f32x4*>(hb + (RB[r] * HW + s2) * HP);
                t[s2] = r == 1 ? da + db : psub(da, db);
            }
            if (r == 0 && more) halo_load(g + 1);          // global loads in flight during the MFMA phase
            f32x4 Vr[4];
            Vr[0] = psub(t[0], t[2]);
            Vr[1] = t[1] + t[2];
            Vr[2] = psub(t[2], t[1]);
            Vr[3] = psub(t[1], t[3]);
            // ---- 4 positions x (K = 16 channels) for this wave's m-tile; the filter fragment is requested one position ahead ----
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const int p = r * 4 + c;
                const f32x4 an = p + 1 < 16 ? wg[(size_t)(p + 1) * wstride] : wfirst_next[0];
                __builtin_amdgcn_sched_barrier(0);
                const f32x4 b = Vr[c];
                if constexpr (BF) {
                    acc[p] = mfma_bf16(bf16pack(af), bf16pack(b), acc[p]);
                } else {
#pragma unroll
                    for (int q = 0; q < 4; ++q) acc[p] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[q], b[q], acc[p], 0, 0, 0);
                }
                af = an;
            }
        }
        if (more) {
            __syncthreads();                               // every wave has read its patches of group g
            halo_store();
            __syncthreads();
        }
    }

    // ---- A^T M A in registers: lane (j, kk) holds tile (n, j), output channels 4kk..4kk+3 of m-tile mt ----------------
    if (co >= a.cout) return;                              // (cout is a multiple of 16 on this path)
    f32x4 s0[4], s1[4];
#pragma unroll
    for (int s2 = 0; s2 < 4; ++s2) {
        s0[s2] = acc[0 * 4 + s2] + acc[1 * 4 + s2] + acc[2 * 4 + s2];
        s1[s2] = psub(psub(acc[1 * 4 + s2], acc[2 * 4 + s2]), acc[3 * 4 + s2]);
    }
    const f32x4 b4 = *reinterpret_cast<const f32x4*>(a.bias + co);
    f32x4 y[2][2];
    y[0][0] = s0[0] + s0[1] + s0[2] + b4;
    y[0][1] = psub(psub(s0[1], s0[2]), s0[3]) + b4;
    y[1][0] = s1[0] + s1[1] + s1[2] + b4;
    y[1][1] = psub(psub(s1[1], s1[2]), s1[3]) + b4;
    const int relu_o = a.relu_out ? 0 : (int)0x80000000;
#pragma unroll
    for (int dy = 0; dy < 2; ++dy)
#pragma unroll
        for (int dx = 0; dx < 2; ++dx) {
            const int yy = oy + dy, xx = ox + dx;
            y[dy][dx] = imax4(y[dy][dx] + rv[dy * 2 + dx], relu_o);
            if (a.act) y[dy][dx] = act4(y[dy][dx], a.act);
            if (yy < P.Ho && xx < P.Wo && !a.skip_full) {
                const size_t pp = ((size_t)yy * P.Wo + xx) * a.cout + co;
                *reinterpret_cast<f32x4*>(P.out + pp) = y[dy][dx];
            }
        }
    if (P.pool && oy < P.Ho && ox < P.Wo) {
        // the lane's 2x2 output tile IS a pool window (oy, ox even); ceil mode at the right / bottom border
        f32x4 mm = y[0][0];
        if (ox + 1 < P.Wo) mm = max4(mm, y[0][1]);
        if (oy + 1 < P.Ho) {
            mm = max4(mm, y[1][0]);
            if (ox + 1 < P.Wo) mm = max4(mm, y[1][1]);
        }
        *reinterpret_cast<f32x4*>(P.pool + ((size_t)(oy >> 1) * ((P.Wo + 1) >> 1) + (ox >> 1)) * a.cout + co) = mm;
    }
}

// ------------------------------------------------------------------------------------------------
// conv2d_transpose 3x3, stride 2, SAME (layers.py:362).  With I = i + pad_before:
//   I = 2*o + k  ->  k odd <=> I odd;  I even: k in {0 (o = I/2), 2 (o = I/2 - 1)};  I odd: k = 1.
// The block works on a tile of q = floor(I/2) positions; every q yields the four outputs
// (2q+ry-pbh, 2q+rx-pbw), ry,rx in {0,1} ("parity classes"), each with its own accumulator.
// ------------------------------------------------------------------------------------------------
constexpr int DC_TH = 8;
constexpr int DC_TW = 16;

template <int MT, bool BF = false>
__global__ __launch_bounds__(256, 2) void deconv_mfma_kernel(const ConvArgs a) {
    constexpr int TH = DC_TH, TW = DC_TW, NT = 2;
    constexpr int LH = TH + 1, LW = TW + 1;   // one halo row/column before the tile (o = q - 1)
    __shared__ __attribute__((aligned(16))) float lds[LH * LW * 16];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int j = lane & 15, kk = lane >> 4;
    const int bid = sched_tile(a.xm);
    if (bid < 0) return;
    int pi = 0;
    pi = prob_of_tile(a, bid);
    const ConvProb& P = a.p[pi];
    const int tile = bid - P.tile_begin;
    const int ty = tile / P.tiles_x, tx = tile - ty * P.tiles_x;
    const int qx0 = tx * TW, qy0 = ty * TH, mt0 = blockIdx.y * MT;

    f32x4 acc[MT][NT][4];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int n = 0; n < NT; ++n)
#pragma unroll
            for (int c = 0; c < 4; ++c) acc[m][n][c] = f32x4{0.f, 0.f, 0.f, 0.f};

    int nbase[NT];   // LDS float index of o = (q_row, q_col) itself (halo offset +1,+1)
#pragma unroll
    for (int n = 0; n < NT; ++n) nbase[n] = ((wave * NT + n + 1) * LW + j + 1) * 16;

    // halo loader (same instruction diet as conv_mfma_kernel: incremental (ly, lx), clamped branch-free loads, all slots
    // requested before the first is written to LDS)
    constexpr int NV = LH * LW * 4, NLOAD = (NV + 255) / 256;
    constexpr int QD = 64 / LW, RD = 64 % LW;
    const int pix0 = tid >> 2, sub0 = tid & 3;
    const int ly0 = pix0 / LW, lx0 = pix0 - ly0 * LW;
    const int relu_lim = a.relu_in ? 0 : (int)0x80000000;
    const int H = P.H, W = P.W;

    for (int g = 0; g < a.groups; ++g) {
        if (g > 0) __syncthreads();
        {
            const int c = g * 16 + sub0 * 4;
            const bool cok = c < a.c0;
            const float* __restrict__ src = P.in0 + (cok ? c : 0);
            f32x4 st[NLOAD];
            unsigned stmask = 0;
            int ly = ly0, lx = lx0;
#pragma unroll
            for (int i = 0; i < NLOAD; ++i) {
                const int gy = qy0 - 1 + ly, gx = qx0 - 1 + lx;
                const bool ok = cok && gy >= 0 && gy < H && gx >= 0 && gx < W;
                const int cy = min(max(gy, 0), H - 1), cx = min(max(gx, 0), W - 1);
                st[i] = *reinterpret_cast<const f32x4*>(src + (size_t)(cy * W + cx) * a.c0);
                stmask |= (ok ? 1u : 0u) << i;
                lx += RD; ly += QD;
                if (lx >= LW) { lx -= LW; ++ly; }
            }
#pragma unroll
            for (int i = 0; i < NLOAD; ++i) {
                const int idx = tid + i * 256;
                if (i * 256 + 255 < NV || idx < NV) {
                    const f32x4 v = ((stmask >> i) & 1u) ? st[i] : f32x4{0.f, 0.f, 0.f, 0.f};
                    *reinterpret_cast<f32x4*>(lds + idx * 4) = imax4(v, relu_lim);
                }
            }
        }
        __syncthreads();

        const f32x4* wg = a.wpk + ((size_t)g * 9 * a.mtiles + mt0) * 64 + lane;
        f32x4 bf[NT][4];   // shifts: 0:(0,0) 1:(0,-1) 2:(-1,0) 3:(-1,-1)
#pragma unroll
        for (int n = 0; n < NT; ++n) {
            bf[n][0] = *reinterpret_cast<const f32x4*>(lds + nbase[n] + kk * 4);
            bf[n][1] = *reinterpret_cast<const f32x4*>(lds + nbase[n] - 16 + kk * 4);
            bf[n][2] = *reinterpret_cast<const f32x4*>(lds + nbase[n] - LW * 16 + kk * 4);
            bf[n][3] = *reinterpret_cast<const f32x4*>(lds + nbase[n] - LW * 16 - 16 + kk * 4);
        }
        f32x4 af[MT];
#pragma unroll
        for (int m = 0; m < MT; ++m) af[m] = wg[(size_t)m * 64];
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int ky = tap / 3, kx = tap % 3;
            const int cls = (ky & 1) * 2 + (kx & 1);
            const int sh = (ky == 2 ? 2 : 0) + (kx == 2 ? 1 : 0);
            f32x4 an[MT];
#pragma unroll
            for (int m = 0; m < MT; ++m)
                an[m] = tap + 1 < 9 ? wg[((size_t)(tap + 1) * a.mtiles + m) * 64] : f32x4{0.f, 0.f, 0.f, 0.f};
            if constexpr (BF) {
#pragma unroll
                for (int m = 0; m < MT; ++m) {
                    const s16x4 pa = bf16pack(af[m]);
#pragma unroll
                    for (int n = 0; n < NT; ++n) acc[m][n][cls] = mfma_bf16(pa, bf16pack(bf[n][sh]), acc[m][n][cls]);
                }
            } else {
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int m = 0; m < MT; ++m)
#pragma unroll
                    for (int n = 0; n < NT; ++n)
                        acc[m][n][cls] =
                            __builtin_amdgcn_mfma_f32_16x16x4f32(af[m][r], bf[n][sh][r], acc[m][n][cls], 0, 0, 0);
            }
#pragma unroll
            for (int m = 0; m < MT; ++m) af[m] = an[m];
        }
    }

    // tiles whose 16 x 32 output pixels all exist and whose channels fill the lanes' quads: no bounds tests
    if (2 * qy0 - P.pbh >= 0 && 2 * (qy0 + TH) - P.pbh <= P.Ho && 2 * qx0 - P.pbw >= 0 && 2 * (qx0 + TW) - P.pbw <= P.Wo && a.cout % 16 == 0 && !a.act) {
        const int relu_o = a.relu_out ? 0 : (int)0x80000000;
        const size_t p00 = (size_t)(2 * (qy0 + wave * NT) - P.pbh) * P.Wo + 2 * (qx0 + j) - P.pbw;
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            const int c = (mt0 + m) * 16 + kk * 4;
            const f32x4 b4 = *reinterpret_cast<const f32x4*>(a.bias + c);
            float* __restrict__ o = P.out + p00 * a.cout + c;
#pragma unroll
            for (int n = 0; n < NT; ++n)
#pragma unroll
                for (int cls = 0; cls < 4; ++cls)
                    *reinterpret_cast<f32x4*>(o + ((size_t)(2 * n + (cls >> 1)) * P.Wo + (cls & 1)) * a.cout) = imax4(acc[m][n][cls] + b4, relu_o);
        }
        return;
    }
    if constexpr (MT == 1) {
        // 8 output channels (level 0): lanes kk = 2, 3 hold no channels.  They take over the odd-column parity class from
        // lanes kk = 0, 1 (cross-lane move), so that one store instruction writes 16 pixel pairs x 8 channels = 1 KB
        // contiguous with all 64 lanes instead of two half-empty ones with 32-byte pieces at a 64-byte stride.
        if (a.cout == 8 && mt0 == 0 && 2 * qy0 - P.pbh >= 0 && 2 * (qy0 + TH) - P.pbh <= P.Ho && 2 * qx0 - P.pbw >= 0 &&
            2 * (qx0 + TW) - P.pbw <= P.Wo && !a.act) {
            const int relu_o = a.relu_out ? 0 : (int)0x80000000;
            const int hi = kk >> 1;
            const f32x4 b4 = *reinterpret_cast<const f32x4*>(a.bias + (kk & 1) * 4);
            float* __restrict__ o = P.out + ((size_t)(2 * (qy0 + wave * NT) - P.pbh) * P.Wo + 2 * (qx0 + j) - P.pbw + hi) * 8 + (kk & 1) * 4;
#pragma unroll
            for (int n = 0; n < NT; ++n)
#pragma unroll
                for (int ry = 0; ry < 2; ++ry) {
                    const f32x4 e = acc[0][n][ry * 2], od = acc[0][n][ry * 2 + 1];
                    f32x4 t;
                    t.x = from_lower_half(od.x); t.y = from_lower_half(od.y); t.z = from_lower_half(od.z); t.w = from_lower_half(od.w);
                    const f32x4 v = hi ? t : e;
                    *reinterpret_cast<f32x4*>(o + (size_t)(2 * n + ry) * P.Wo * 8) = imax4(v + b4, relu_o);
                }
            return;
        }
    }
#pragma unroll
    for (int n = 0; n < NT; ++n) {
        const int qy = qy0 + wave * NT + n, qx = qx0 + j;
#pragma unroll
        for (int cls = 0; cls < 4; ++cls) {
            const int y = 2 * qy + (cls >> 1) - P.pbh, x = 2 * qx + (cls & 1) - P.pbw;
            if (y < 0 || y >= P.Ho || x < 0 || x >= P.Wo) continue;
            const size_t p = (size_t)y * P.Wo + x;
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                const int c = (mt0 + m) * 16 + kk * 4;
                if (c >= a.cout) continue;
                f32x4 v = acc[m][n][cls];
                if (c + 3 < a.cout) {
                    v += *reinterpret_cast<const f32x4*>(a.bias + c);
                    if (a.relu_out) v = relu4(v);
                    else if (a.act) v = act4(v, a.act);
                    *reinterpret_cast<f32x4*>(P.out + p * a.cout + c) = v;
                } else {
                    for (int r = 0; r < 4 && c + r < a.cout; ++r) {
                        float s = v[r] + a.bias[c + r];
                        if (a.relu_out) s = fmaxf(s, 0.f);
                        else if (a.act) s = act1(s, a.act);
                        P.out[p * a.cout + c + r] = s;
                    }
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// First layer (Cin == 1): direct KxK convolution, one thread per pixel, all COUT channels.
// Optional per-image standardisation (layers.py:672-711): stats = {mean, 1/max(std,1e-4)}.
// ------------------------------------------------------------------------------------------------
struct C1Prob {
    const float* img;      // [H,W] single channel
    float* out;            // [H,W,COUT]
    const float* stats;    // {mean, 1/std} or nullptr
    int H, W;
    int tiles_x;           // 64-pixel-wide, 4-row blocks per row
    int tile_begin;
};
struct C1Args {
    C1Prob p[MAXP];
    int nprob;
    const float* w;        // [K*K][COUT]
    const float* bias;     // [COUT]
    int relu;
    int act;               // graph variants: 1 = elu, 2 = leaky behind the conv (relu is 0 then)
};

// OUTBF: the output is written as bf16 (native bf16 path; COUT % 8 == 0)
template <int K, int COUT, bool OUTBF = false>
__global__ __launch_bounds__(256) void conv_c1_kernel(const C1Args a) {
    __shared__ float sw[K * K * COUT + COUT];
    // OUTBF (first layer of a bf16 net outside the fused level-0 block): image and filter as bfloat16, like res8f_kernel / res8b_tile read them --
    // the first layer of the bf16 data path is ONE function whichever kernel evaluates it (products of two bfloat16 are exact in fp32)
    auto rbf = [](float v) { return OUTBF ? __uint_as_float(bf16x2_of(v, 0.f) << 16) : v; };
    for (int i = threadIdx.x; i < K * K * COUT; i += 256) sw[i] = rbf(a.w[i]);
    for (int i = threadIdx.x; i < COUT; i += 256) sw[K * K * COUT + i] = a.bias[i];
    __syncthreads();
    int pi = 0;
    pi = prob_of_tile(a, (int)blockIdx.x);
    const C1Prob& P = a.p[pi];
    const int tile = blockIdx.x - P.tile_begin;
    const int ty = tile / P.tiles_x, tx = tile - ty * P.tiles_x;
    const int x = tx * 64 + (threadIdx.x & 63);
    const int y = ty * 4 + (threadIdx.x >> 6);
    const int H = P.H, W = P.W;
    if (x >= W || y >= H) return;
    constexpr int PB = (K - 1) / 2;
    float mean = 0.f, inv = 1.f;
    if (P.stats) { mean = P.stats[0]; inv = P.stats[1]; }
    const float* __restrict__ img = P.img;
    float acc[COUT];
#pragma unroll
    for (int c = 0; c < COUT; ++c) acc[c] = 0.f;
#pragma unroll
    for (int ky = 0; ky < K; ++ky) {
        const int gy = y + ky - PB;
#pragma unroll
        for (int kx = 0; kx < K; ++kx) {
            const int gx = x + kx - PB;
            float v = 0.f;
            if (gy >= 0 && gy < H && gx >= 0 && gx < W) v = rbf((img[(size_t)gy * W + gx] - mean) * inv);
#pragma unroll
            for (int c = 0; c < COUT; ++c) acc[c] = fmaf(v, sw[(ky * K + kx) * COUT + c], acc[c]);
        }
    }
#pragma unroll
    for (int c = 0; c < COUT; ++c) {
        const float s = acc[c] + sw[K * K * COUT + c];
        acc[c] = a.relu ? fmaxf(s, 0.f) : (a.act ? act1(s, a.act) : s);
    }
    if constexpr (OUTBF) {
        static_assert(!OUTBF || COUT % 8 == 0, "bf16 output: whole 16-byte units");
        unsigned short* o = reinterpret_cast<unsigned short*>(P.out) + ((size_t)y * W + x) * COUT;
#pragma unroll
        for (int c = 0; c < COUT; c += 8) {
            typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
            *reinterpret_cast<u32x4_t*>(o + c) = u32x4_t{bf16x2_of(acc[c], acc[c + 1]), bf16x2_of(acc[c + 2], acc[c + 3]),
                                                         bf16x2_of(acc[c + 4], acc[c + 5]), bf16x2_of(acc[c + 6], acc[c + 7])};
        }
    } else {
        float* o = P.out + ((size_t)y * W + x) * COUT;
#pragma unroll
        for (int c = 0; c < COUT; c += 4) *reinterpret_cast<f32x4*>(o + c) = f32x4{acc[c], acc[c + 1], acc[c + 2], acc[c + 3]};
    }
}

// ------------------------------------------------------------------------------------------------
// Attention CNN head (ARU_v1.py:173-175): 4x4 conv 1->12 + ReLU + 2x2 max pool in one pass.  The 12-channel
// full-resolution tensor (648 MB per page) is never written: the conv runs on the MFMA with K = the 16 taps
// (slot 4*kk+r -> tap row kk, tap column r), M = 12 (of 16) output channels, N = 16 pixels of one row, and the
// pool is taken in registers (two rows per unit) and across lanes j ^ 1.
// ------------------------------------------------------------------------------------------------
constexpr int ATT_TH = 32, ATT_TW = 64;                      // output (pre-pool) tile; tile origins are even
struct AttHeadArgs {
    C1Prob p[MAXP];        // img, out = pooled [ceil(H/2), ceil(W/2), 12], stats, H, W, tiles_x, tile_begin
    int nprob;
    const f32x4* wpk;      // [64 lanes] A fragment: row = cout (12 real), slots = taps
    const float* bias;     // [12]
    const float* w;        // [16 taps][12] (att_headv_kernel: scalar operands)
    int act;               // att_headv_kernel, graph variants: 0 = ReLU, 1 = elu, 2 = leaky (applied to the pooled maximum: both are increasing)
};

__global__ __launch_bounds__(256) void att_head_kernel(const AttHeadArgs a) {
    constexpr int LH = ATT_TH + 3, LW = ATT_TW + 4;          // SAME for 4x4: 1 before, 2 after (+1 col of slack)
    __shared__ float img[LH * LW];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int j = lane & 15, kk = lane >> 4;
    int pi = 0;
    pi = prob_of_tile(a, (int)blockIdx.x);
    const C1Prob& P = a.p[pi];
    const int tile = blockIdx.x - P.tile_begin;
    const int ty = tile / P.tiles_x, tx = tile - ty * P.tiles_x;
    const int x0 = tx * ATT_TW, y0 = ty * ATT_TH;
    const int H = P.H, W = P.W;
    float mean = 0.f, inv = 1.f;
    if (P.stats) { mean = P.stats[0]; inv = P.stats[1]; }
    for (int i = tid; i < LH * LW; i += 256) {
        const int r = i / LW, c = i - r * LW;
        const int gy = y0 - 1 + r, gx = x0 - 1 + c;
        float v = 0.f;
        if (gy >= 0 && gy < H && gx >= 0 && gx < W) v = (P.img[(size_t)gy * W + gx] - mean) * inv;
        img[i] = v;
    }
    const f32x4 A = a.wpk[lane];
    f32x4 bias4 = f32x4{0.f, 0.f, 0.f, 0.f};
    if (kk < 3) bias4 = *reinterpret_cast<const f32x4*>(a.bias + kk * 4);
    __syncthreads();
    const int Hp = (H + 1) >> 1, Wp = (W + 1) >> 1;
    // units: (row pair, 16-pixel column block): 16 x 4 = 64 units, 16 per wave
    for (int u = wave; u < (ATT_TH / 2) * (ATT_TW / 16); u += 4) {
        const int rp = u >> 2, cb = u & 3;
        const int ly = 2 * rp, lx = cb * 16 + j;              // output pixel (ly, lx) of the tile; taps start at img[ly + kk][lx + r]
        const float* p0 = img + (ly + kk) * LW + lx;
        f32x4 acc0 = f32x4{0.f, 0.f, 0.f, 0.f}, acc1 = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(A[r], p0[r], acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(A[r], p0[LW + r], acc1, 0, 0, 0);
        }
        const int gy = y0 + ly, gx = x0 + lx;
        f32x4 v0 = relu4(acc0 + bias4), v1 = relu4(acc1 + bias4);
        // pool: rows in registers (second row only if inside the image), x neighbour in lane j ^ 1
        f32x4 m = (gy + 1 < H) ? f32x4{fmaxf(v0.x, v1.x), fmaxf(v0.y, v1.y), fmaxf(v0.z, v1.z), fmaxf(v0.w, v1.w)} : v0;
        f32x4 o;
        o.x = __shfl_xor(m.x, 1); o.y = __shfl_xor(m.y, 1); o.z = __shfl_xor(m.z, 1); o.w = __shfl_xor(m.w, 1);
        if ((j & 1) == 0 && kk < 3 && gy < H && gx < W) {
            if (gx + 1 < W) { m.x = fmaxf(m.x, o.x); m.y = fmaxf(m.y, o.y); m.z = fmaxf(m.z, o.z); m.w = fmaxf(m.w, o.w); }
            *reinterpret_cast<f32x4*>(P.out + ((size_t)(gy >> 1) * Wp + (gx >> 1)) * 12 + kk * 4) = m;
        }
    }
    (void)Hp;
}

// mean / E[x^2] partial sums for per-image standardisation (double accumulation; the host side divides).  16-byte loads and four independent
// accumulator pairs per thread: the first cut read one float per thread and iteration and ran at 1 TB/s (56 us for a 3000 x 4500 page, once per
// page and step: 2.5 % of the bf16 step).
__global__ __launch_bounds__(256) void moments_kernel(const float* __restrict__ x, size_t n, double* __restrict__ sums) {
    double s = 0.0, s2 = 0.0;
    const size_t t0 = (size_t)blockIdx.x * 256 + threadIdx.x, stride = (size_t)gridDim.x * 256;
    if ((reinterpret_cast<uintptr_t>(x) & 15) == 0) {
        const f32x4* __restrict__ x4 = reinterpret_cast<const f32x4*>(x);
        const size_t n4 = n >> 2;
        double a[4] = {0.0, 0.0, 0.0, 0.0}, q[4] = {0.0, 0.0, 0.0, 0.0};
        for (size_t i = t0; i < n4; i += stride) {
            const f32x4 v = x4[i];
#pragma unroll
            for (int c = 0; c < 4; ++c) { const double d = v[c]; a[c] += d; q[c] += d * d; }
        }
        s = (a[0] + a[1]) + (a[2] + a[3]);
        s2 = (q[0] + q[1]) + (q[2] + q[3]);
        for (size_t i = (n4 << 2) + t0; i < n; i += stride) { const double d = x[i]; s += d; s2 += d * d; }
    } else {
        for (size_t i = t0; i < n; i += stride) {
            const double v = x[i];
            s += v;
            s2 += v * v;
        }
    }
    __shared__ double sh[2][256];
    sh[0][threadIdx.x] = s;
    sh[1][threadIdx.x] = s2;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (threadIdx.x < o) {
            sh[0][threadIdx.x] += sh[0][threadIdx.x + o];
            sh[1][threadIdx.x] += sh[1][threadIdx.x + o];
        }
        __syncthreads();
    }
    // one partial pair per block, summed in a fixed order by moments_finish_kernel (first cut: two fp64 atomicAdds per block on ONE address --
    // 4096 serialised L2 atomics per page -- and a memset launch in front of them)
    if (threadIdx.x == 0) {
        sums[2 * blockIdx.x] = sh[0][0];
        sums[2 * blockIdx.x + 1] = sh[1][0];
    }
}

// one block of 256 threads: partial pairs [nparts][2] -> {mean, 1 / max(sd, 1e-4)}
__global__ __launch_bounds__(256) void moments_finish_kernel(const double* __restrict__ parts, int nparts, size_t n, float* __restrict__ stats) {
    double s = 0.0, s2 = 0.0;
    for (int i = threadIdx.x; i < nparts; i += 256) { s += parts[2 * i]; s2 += parts[2 * i + 1]; }
    __shared__ double sh[2][256];
    sh[0][threadIdx.x] = s;
    sh[1][threadIdx.x] = s2;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (threadIdx.x < o) {
            sh[0][threadIdx.x] += sh[0][threadIdx.x + o];
            sh[1][threadIdx.x] += sh[1][threadIdx.x + o];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const double mean = sh[0][0] / (double)n;
        double var = sh[1][0] / (double)n - mean * mean;
        if (var < 0) var = 0;
        const float sd = fmaxf((float)sqrt(var), 1e-4f);
        stats[0] = (float)mean;
        stats[1] = 1.0f / sd;
    }
}

// ------------------------------------------------------------------------------------------------
// 2x2 / stride 2 / SAME pools (ceil mode, padding at the end only) and the channel sum, grouped over problems
// ------------------------------------------------------------------------------------------------
struct PoolProb {
    const float* in;
    float* out;
    int H, W, Ho, Wo;
    int blk_begin;         // first blockIdx.x of this problem; each block covers 1024 work items
    int pad_;
};
struct PoolArgs {
    PoolProb p[MAXP];
    int nprob;
    int C;
};
constexpr int POOL_ITEMS = 1024;

// In-place activation of the non-ReLU graph variants (ARU_v1.py:70-75; asep_aru_cfg.activation): 1 = elu (tf.nn.elu: x > 0 ? x :
// exp(x) - 1, Eigen functor), 2 = leaky (layers.py:10-30: max(0, x) + 0.1 min(0, x)).  These nets run layer by layer (no fused blocks): the conv
// kernels store the pre-activation value and this kernel follows; both functions are increasing, so a 2x2 max pool taken in the
// conv's epilogue commutes with them (the pooled tensor gets the same pass).  p[].in == p[].out, Ho x Wo x C values each.
__global__ __launch_bounds__(256) void act_kernel(const PoolArgs a, int mode) {
    int pi = 0;
    pi = prob_of_blk(a, (int)blockIdx.x);
    const PoolProb& P = a.p[pi];
    const size_t total = (size_t)P.Ho * P.Wo * a.C;
    const size_t base = (size_t)(blockIdx.x - P.blk_begin) * POOL_ITEMS * 4;
    for (int k = 0; k < POOL_ITEMS / 256; ++k) {
        const size_t i = base + ((size_t)k * 256 + threadIdx.x) * 4;
        if (i >= total) break;
        if (i + 4 <= total) {
            f32x4 v = *reinterpret_cast<const f32x4*>(P.out + i);
#pragma unroll
            for (int q = 0; q < 4; ++q) v[q] = mode == 1 ? (v[q] > 0.f ? v[q] : expf(v[q]) - 1.f) : fmaxf(v[q], 0.f) + 0.1f * fminf(v[q], 0.f);
            *reinterpret_cast<f32x4*>(P.out + i) = v;
        } else {
            for (size_t q = i; q < total; ++q) {
                const float x = P.out[q];
                P.out[q] = mode == 1 ? (x > 0.f ? x : expf(x) - 1.f) : fmaxf(x, 0.f) + 0.1f * fminf(x, 0.f);
            }
        }
    }
}

__global__ __launch_bounds__(256) void maxpool2_kernel(const PoolArgs a) {
    int pi = 0;
    pi = prob_of_blk(a, (int)blockIdx.x);
    const PoolProb& P = a.p[pi];
    const int C = a.C, c4n = C >> 2, H = P.H, W = P.W, Wo = P.Wo;
    const size_t total = (size_t)P.Ho * Wo * c4n;
    const float* __restrict__ in = P.in;
    const size_t base = (size_t)(blockIdx.x - P.blk_begin) * POOL_ITEMS;
    for (int k = 0; k < POOL_ITEMS / 256; ++k) {
        const size_t i = base + k * 256 + threadIdx.x;
        if (i >= total) break;
        const int c4 = (int)(i % c4n);
        const size_t p = i / c4n;
        const int x = (int)(p % Wo), y = (int)(p / Wo);
        const int y1 = 2 * y + 1 < H ? 2 * y + 1 : 2 * y, x1 = 2 * x + 1 < W ? 2 * x + 1 : 2 * x;
        const f32x4 q0 = *reinterpret_cast<const f32x4*>(in + ((size_t)(2 * y) * W + 2 * x) * C + c4 * 4);
        const f32x4 q1 = *reinterpret_cast<const f32x4*>(in + ((size_t)(2 * y) * W + x1) * C + c4 * 4);
        const f32x4 q2 = *reinterpret_cast<const f32x4*>(in + ((size_t)y1 * W + 2 * x) * C + c4 * 4);
        const f32x4 q3 = *reinterpret_cast<const f32x4*>(in + ((size_t)y1 * W + x1) * C + c4 * 4);
        f32x4 m;
#pragma unroll
        for (int r = 0; r < 4; ++r) m[r] = fmaxf(fmaxf(q0[r], q1[r]), fmaxf(q2[r], q3[r]));
        *reinterpret_cast<f32x4*>(P.out + p * C + c4 * 4) = m;
    }
}

// single-channel average pool; divisor = number of valid elements (tf.nn.avg_pool2d SAME)
__global__ __launch_bounds__(256) void avgpool2_c1_kernel(const PoolArgs a) {
    int pi = 0;
    pi = prob_of_blk(a, (int)blockIdx.x);
    const PoolProb& P = a.p[pi];
    const int H = P.H, W = P.W, Wo = P.Wo;
    const size_t total = (size_t)P.Ho * Wo;
    const size_t base = (size_t)(blockIdx.x - P.blk_begin) * POOL_ITEMS;
    // (32-bit index arithmetic where the tensor allows it: the 64-bit % and / of the general form cost ~200 instructions per pixel;
    // the four window values of the thread's pixels are requested from clamped addresses before the first sum)
    if (total < ((size_t)1 << 30) && (size_t)H * W < ((size_t)1 << 31)) {
        float v[POOL_ITEMS / 256][4];
        int cnt[POOL_ITEMS / 256];
#pragma unroll
        for (int k = 0; k < POOL_ITEMS / 256; ++k) {
            const unsigned i = min((unsigned)base + k * 256 + threadIdx.x, (unsigned)total - 1);
            const unsigned y = i / (unsigned)Wo, x = i - y * (unsigned)Wo;
            const int y1 = min((int)(2 * y + 1), H - 1), x1 = min((int)(2 * x + 1), W - 1);
            cnt[k] = ((2 * y + 1 < (unsigned)H) ? 2 : 1) * ((2 * x + 1 < (unsigned)W) ? 2 : 1);
            v[k][0] = P.in[(2 * y) * (unsigned)W + 2 * x];
            v[k][1] = P.in[(2 * y) * (unsigned)W + x1];
            v[k][2] = P.in[(unsigned)y1 * (unsigned)W + 2 * x];
            v[k][3] = P.in[(unsigned)y1 * (unsigned)W + x1];
        }
#pragma unroll
        for (int k = 0; k < POOL_ITEMS / 256; ++k) {
            const size_t i = base + k * 256 + threadIdx.x;
            if (i >= total) break;
            const unsigned y = (unsigned)i / (unsigned)Wo, x = (unsigned)i - y * (unsigned)Wo;
            // same association as the general loop: ((v00 + v01) + v10) + v11 over the window values that exist
            float s = v[k][0];
            const bool xin = 2 * x + 1 < (unsigned)W, yin = 2 * y + 1 < (unsigned)H;
            if (xin) s += v[k][1];
            if (yin) { s += v[k][2]; if (xin) s += v[k][3]; }
            P.out[i] = s / (float)cnt[k];
        }
        return;
    }
    for (int k = 0; k < POOL_ITEMS / 256; ++k) {
        const size_t i = base + k * 256 + threadIdx.x;
        if (i >= total) break;
        const int x = (int)(i % Wo), y = (int)(i / Wo);
        float s = 0.f;
        int n = 0;
        for (int dy = 0; dy < 2; ++dy)
            for (int dx = 0; dx < 2; ++dx) {
                const int yy = 2 * y + dy, xx = 2 * x + dx;
                if (yy < H && xx < W) { s += P.in[(size_t)yy * W + xx]; ++n; }
            }
        P.out[i] = s / (float)n;
    }
}

// channel sum [H,W,C] -> [H,W]  (the channel-summing half of upsample_simple, layers.py:716-720)
__global__ __launch_bounds__(256) void chansum_kernel(const PoolArgs a) {
    int pi = 0;
    pi = prob_of_blk(a, (int)blockIdx.x);
    const PoolProb& P = a.p[pi];
    const size_t total = (size_t)P.H * P.W;
    const size_t base = (size_t)(blockIdx.x - P.blk_begin) * POOL_ITEMS;
    if (a.C == 8) {
        // the usual case (feat_root 8): two 16-byte loads per pixel, all four pixels of a thread requested before the sums
        // (same association as the scalar loop: ((((((c0 + c1) + c2) + c3) + c4) + c5) + c6) + c7)
        f32x4 v[POOL_ITEMS / 256][2];
#pragma unroll
        for (int k = 0; k < POOL_ITEMS / 256; ++k) {
            const size_t i = min(base + k * 256 + threadIdx.x, total - 1);
            v[k][0] = *reinterpret_cast<const f32x4*>(P.in + i * 8);
            v[k][1] = *reinterpret_cast<const f32x4*>(P.in + i * 8 + 4);
        }
#pragma unroll
        for (int k = 0; k < POOL_ITEMS / 256; ++k) {
            const size_t i = base + k * 256 + threadIdx.x;
            if (i >= total) break;
            P.out[i] = ((((((v[k][0].x + v[k][0].y) + v[k][0].z) + v[k][0].w) + v[k][1].x) + v[k][1].y) + v[k][1].z) + v[k][1].w;
        }
        return;
    }
    for (int k = 0; k < POOL_ITEMS / 256; ++k) {
        const size_t i = base + k * 256 + threadIdx.x;
        if (i >= total) break;
        float s = 0.f;
        for (int c = 0; c < a.C; ++c) s += P.in[i * a.C + c];
        P.out[i] = s;
    }
}

// ------------------------------------------------------------------------------------------------
// Attention combine + logits + class softmax + uint8/threshold (ARU_v1.py:141-160, helper:75-78)
// ------------------------------------------------------------------------------------------------
constexpr int MAX_SCALES = 5;
struct CombineArgs {
    const float* f0;                 // scale-0 feature map [H,W,FR]
    const float* fsum[MAX_SCALES];   // [s>=1] channel sums of scale-s feature maps [fh,fw]
    const float* att[MAX_SCALES];    // attention maps [ah,aw] (1 channel)
    int fh[MAX_SCALES], fw[MAX_SCALES], fup[MAX_SCALES], fph[MAX_SCALES], fpw[MAX_SCALES];
    int ah[MAX_SCALES], aw[MAX_SCALES], aup[MAX_SCALES], aph[MAX_SCALES], apw[MAX_SCALES];
    int ash[MAX_SCALES], fsh[MAX_SCALES];   // log2 of aup / fup
    int nsc;                         // number of scales (1 = no attention)
    int H, W;
    const float* wl;                 // [4*4][FR][NC]
    const float* bl;                 // [NC]
    const float* wd;                 // NC == 2 with soft-max: [4*4][FR] class-1 minus class-0 filter, [16 FR] = bias difference; or nullptr
    float* out;                      // [H,W,NC] probabilities (or logits)
    uint8_t* out_u8;                 // optional
    uint8_t* out_mask;               // optional
    double thr255;
    int softmax;
    int tiles_x;                     // 32-pixel tile columns; the grid is one-dimensional
    XcdMap xm;                       // XCD-aware block -> tile map (sched_tile)
};

constexpr int COMBINE_TW = 32;   // tile width of combine_kernel (16 rows)
#ifndef CMB_ABL
#define CMB_ABL 0                // ablation builds (WRONG RESULTS ON PURPOSE): 1 no attention / channel-sum gathers, 2 no uint8 / threshold stores, 4 no probability stores, 8 no feature loads
#endif
// F0BF: the scale-0 feature map is bf16 (native bf16 path): one 16-byte load per pixel, widened to fp32 in registers
// NSCT: number of scales as a compile-time constant (1 = no attention, 3 = the default net; 0 = run-time a.nsc).  The first phase
// of this kernel was 850 vector instructions per thread against 460 in the logits conv: five scale iterations of which two are
// dead at run time but paid for in selects and moves, and 64-bit index arithmetic (mad_i64 / ashr / lshl_add_u64) for fifteen
// gathers.  With NSCT the dead iterations vanish, and the specialised forms index with 32-bit offsets from the (scalar) base
// pointers -- the host selects them only for tensors below 4 GB.
template <int FR, int NC, bool F0BF = false, int NSCT = 0>
__global__ __launch_bounds__(256) void combine_kernel(const CombineArgs a) {
    const int nsc = NSCT ? NSCT : a.nsc;
    typedef typename std::conditional<NSCT != 0, unsigned, size_t>::type idx_t;
    static_assert(!F0BF || FR == 8, "bf16 feature map: 8 channels = one 16-byte record");
    // 32 x 16 pixel tile, two horizontally adjacent pixels per thread (their 4 x 5 pixel window is read from LDS once: the
    // kernel is bound by LDS reads, 32 x 16 B per pixel in the one-pixel form)
    constexpr int TW = COMBINE_TW, T = 16, LW = TW + 3, L = T + 3;     // 4x4 SAME: pad 1 before, 2 after
    constexpr int LWP = LW + 1;                                       // row pitch in pixels (even: 64-byte pixel-pair records)
    __shared__ __attribute__((aligned(16))) float m[L * LWP * FR];
    // float index of (row, pixel, channel quad).  FR == 8: the quads of a 64-byte pixel pair are permuted by the pair's index
    // (the second phase reads, per instruction, the same quad of 8 consecutive pairs: two bank groups without it)
    auto mi = [](int ly, int lx, int c4) {
        if constexpr (FR == 8) return ly * LWP * 8 + ((lx >> 1) << 4) + (((((lx & 1) << 1) | c4) ^ ((lx >> 2) & 3)) << 2);
        else return (ly * LWP + lx) * FR + c4 * 4;
    };
    __shared__ float swl[16 * FR * NC + NC];
    const int tid = threadIdx.x;
    const int bid = sched_tile(a.xm);
    if (bid < 0) return;
    const int tyb = bid / a.tiles_x;
    const int x0 = (bid - tyb * a.tiles_x) * TW, y0 = tyb * T;
    if (tid < NC) swl[16 * FR * NC + tid] = a.bl[tid];

    // phase 1 in two sweeps over the thread's (at most NPX) window pixels: ALL loads first -- feature pixel, attention
    // values, channel sums, from clamped (always valid) addresses --, then the arithmetic.  As one loop the three dependent
    // rounds of gathers of every pixel were exposed one after the other (the kernel was latency-bound in this phase).
    constexpr int NPX = (L * LW + 255) / 256;
    f32x4 fv[NPX][FR / 4];
    float avv[NPX][MAX_SCALES], fsv[NPX][MAX_SCALES];
    // (Round 6, measured and not kept: the tile's coarse maps -- at most 4 x 6 attention values and 11 x 19 channel sums per map under a 19 x 35
    // window -- staged in LDS first and read there by the window pixels instead of five 4-byte global gathers per pixel: 114 -> 133 us per page,
    // although the kernel WITHOUT the gathers takes 97: the second index computation, the extra barrier and 25 LDS reads per thread cost more than
    // gathers that hit in L1 / L2.  profiles/r6_combine/)
#pragma unroll
    for (int i = 0; i < NPX; ++i) {
        const int pix = min(tid + i * 256, L * LW - 1);
        const int ly = pix / LW, lx = pix - ly * LW;
        const int gy = min(max(y0 - 1 + ly, 0), a.H - 1), gx = min(max(x0 - 1 + lx, 0), a.W - 1);
        if constexpr (F0BF) {
            typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
            const u32x4_t q = (CMB_ABL & 8) ? u32x4_t{(unsigned)gx, 0u, 0u, 0u} : *reinterpret_cast<const u32x4_t*>(reinterpret_cast<const unsigned short*>(a.f0) + ((idx_t)gy * a.W + gx) * FR);
            fv[i][0] = f32x4{__uint_as_float(q.x << 16), __uint_as_float(q.x & 0xffff0000u), __uint_as_float(q.y << 16), __uint_as_float(q.y & 0xffff0000u)};
            fv[i][FR / 4 - 1] = f32x4{__uint_as_float(q.z << 16), __uint_as_float(q.z & 0xffff0000u), __uint_as_float(q.w << 16), __uint_as_float(q.w & 0xffff0000u)};
        } else {
            const float* f = a.f0 + ((idx_t)gy * a.W + gx) * FR;
#pragma unroll
            for (int c4 = 0; c4 < FR / 4; ++c4) fv[i][c4] = *reinterpret_cast<const f32x4*>(f + c4 * 4);
        }
#pragma unroll
        for (int s = 0; s < MAX_SCALES; ++s) {
            avv[i][s] = 0.f; fsv[i][s] = 0.f;
            if (nsc > 1 && s < nsc && !(CMB_ABL & 1)) {
                const int ay = (gy + a.aph[s]) >> a.ash[s], ax = (gx + a.apw[s]) >> a.ash[s];
                avv[i][s] = a.att[s][(idx_t)ay * a.aw[s] + ax];
                if (s >= 1) {
                    const int fy = (gy + a.fph[s]) >> a.fsh[s], fx = (gx + a.fpw[s]) >> a.fsh[s];
                    fsv[i][s] = a.fsum[s][(idx_t)fy * a.fw[s] + fx];
                }
            }
        }
    }
#pragma unroll
    for (int i = 0; i < NPX; ++i) {
        const int pix = tid + i * 256;
        if (pix >= L * LW) break;
        const int ly = pix / LW, lx = pix - ly * LW;
        const int gy = y0 - 1 + ly, gx = x0 - 1 + lx;
        float v[FR];
#pragma unroll
        for (int c = 0; c < FR; ++c) v[c] = 0.f;
        if (gy >= 0 && gy < a.H && gx >= 0 && gx < a.W) {
            if (nsc <= 1) {
#pragma unroll
                for (int c = 0; c < FR; ++c) v[c] = fv[i][c >> 2][c & 3];
            } else {
                // (the up-sampling factors are powers of two -- 8 * 2^s for the attention maps, 2^s for the feature maps --:
                // shifts instead of run-time integer divisions above; one reciprocal instead of three divisions here)
                float av[MAX_SCALES], mx = -INFINITY;
#pragma unroll
                for (int s = 0; s < MAX_SCALES; ++s)
                    if (s < nsc) { av[s] = avv[i][s]; mx = fmaxf(mx, av[s]); }
                float den = 0.f;
#pragma unroll
                for (int s = 0; s < MAX_SCALES; ++s)
                    if (s < nsc) { av[s] = __expf(av[s] - mx); den += av[s]; }     // (v_exp_f32: arguments <= 0, weights in [0, 1]: 1 ulp of the result is far inside the gates)
                const float inv = 1.f / den;
                float add = 0.f;
#pragma unroll
                for (int s = 1; s < MAX_SCALES; ++s)
                    if (s < nsc) add += fsv[i][s] * (av[s] * inv);
                const float w0 = av[0] * inv;
#pragma unroll
                for (int c = 0; c < FR; ++c) v[c] = fv[i][c >> 2][c & 3] * w0 + add;
            }
        }
#pragma unroll
        for (int c4 = 0; c4 < FR / 4; ++c4)
            *reinterpret_cast<f32x4*>(m + mi(ly, lx, c4)) = f32x4{v[c4 * 4], v[c4 * 4 + 1], v[c4 * 4 + 2], v[c4 * 4 + 3]};
    }
    __syncthreads();

    const int tx = tid & 15, ty = tid >> 4;
    const int xb = x0 + 2 * tx, y = y0 + ty;
    if (xb >= a.W || y >= a.H) return;
    float lg[2][NC];
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
        for (int k = 0; k < NC; ++k) lg[q][k] = swl[16 * FR * NC + k];
    if constexpr (NC == 2 && FR % 4 == 0) {
      typedef const float __attribute__((address_space(4)))* cptr;
      if (a.wd) {
        // two classes behind a soft-max: only the DIFFERENCE of the logits decides (p1 = 1 / (1 + exp(l0 - l1))), so the conv runs with the
        // difference filter -- half the multiplications (64 packed FMAs per pixel instead of 128: the kernel is bound by its vector
        // instructions).  A packed accumulator = two channels of a quad; the larger class gets 1 / (1 + e), the smaller e / (1 + e) with
        // e = exp(-|d|) -- the soft-max's own form (exp(l - max) / sum), its argument rounded once more (d is a sum over the difference
        // filter, not the difference of two sums).
        f32x2 l2[2][2];
#pragma unroll
        for (int q = 0; q < 2; ++q) l2[q][0] = l2[q][1] = f32x2{0.f, 0.f};
#pragma unroll
        for (int ky = 0; ky < 4; ++ky) {
            f32x4 v[5][FR / 4];                                   // the row's five window pixels
#pragma unroll
            for (int i = 0; i < 5; ++i)
#pragma unroll
                for (int c4 = 0; c4 < FR / 4; ++c4) v[i][c4] = *reinterpret_cast<const f32x4*>(m + mi(ty + ky, 2 * tx + i, c4));
#pragma unroll
            for (int kx = 0; kx < 4; ++kx) {
                cptr wp = (cptr)(a.wd + (ky * 4 + kx) * FR);
#pragma unroll
                for (int c4 = 0; c4 < FR / 4; ++c4) {
                    const f32x2 w01 = f32x2{wp[c4 * 4 + 0], wp[c4 * 4 + 1]}, w23 = f32x2{wp[c4 * 4 + 2], wp[c4 * 4 + 3]};
#pragma unroll
                    for (int q = 0; q < 2; ++q) {
                        const f32x4 d = v[kx + q][c4];
                        asm("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(l2[q][0]) : "v"(f32x2{d.x, d.y}), "s"(w01));
                        asm("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(l2[q][1]) : "v"(f32x2{d.z, d.w}), "s"(w23));
                    }
                }
            }
        }
        const float bd = ((cptr)a.wd)[16 * FR];
        // the thread's two pixels x two classes are 16 consecutive bytes of the probability map (4 of the uint8 / threshold images): ONE store each
        // (round 6: four 4-byte and eight 1-byte stores per thread before; 118 -> 114 us per page -- the kernel without ANY store takes 66:
        // profiles/r6_combine/ablation_combine_kernel.txt).  Addresses are 8-byte (2-byte) aligned: enough for the hardware's multi-dword stores.
        float pr[2][2];
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const f32x2 t = l2[q][0] + l2[q][1];
            const float d = (t.x + t.y) + bd;                     // l1 - l0
            const float e = expf(-fabsf(d)), big = 1.f / (1.f + e), small = e / (1.f + e);
            pr[q][0] = d > 0.f ? small : big;
            pr[q][1] = d > 0.f ? big : small;
        }
        const idx_t p = ((idx_t)y * a.W + xb) * NC;
        const bool both = xb + 1 < a.W;
        uint8_t u[2][2];
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
            for (int k = 0; k < 2; ++k) u[q][k] = (uint8_t)(pr[q][k] * 255.0f);                // np.array(p*255, dtype=uint8)
        if (!(CMB_ABL & 4)) {
            if (both) *reinterpret_cast<f32x4*>(a.out + p) = f32x4{pr[0][0], pr[0][1], pr[1][0], pr[1][1]};
            else *reinterpret_cast<f32x2*>(a.out + p) = f32x2{pr[0][0], pr[0][1]};
        }
        if (!(CMB_ABL & 2)) {
            if (a.out_u8) {
                *reinterpret_cast<unsigned short*>(a.out_u8 + p) = (unsigned short)(u[0][0] | (u[0][1] << 8));
                if (both) *reinterpret_cast<unsigned short*>(a.out_u8 + p + 2) = (unsigned short)(u[1][0] | (u[1][1] << 8));
            }
            if (a.out_mask) {
                const double thr = a.thr255;
                auto mk = [&](uint8_t v) { return ((double)v > thr) ? 255u : 0u; };
                *reinterpret_cast<unsigned short*>(a.out_mask + p) = (unsigned short)(mk(u[0][0]) | (mk(u[0][1]) << 8));
                if (both) *reinterpret_cast<unsigned short*>(a.out_mask + p + 2) = (unsigned short)(mk(u[1][0]) | (mk(u[1][1]) << 8));
            }
        }
        return;
      }
        // the two classes are one packed accumulator; four independent chains per pixel (one per channel of a quad) are
        // summed at the end.  Weights: uniform addresses -> scalar loads, SGPR operands of the FMAs.
        f32x2 l2[2][4];
#pragma unroll
        for (int q = 0; q < 2; ++q) { l2[q][0] = f32x2{lg[q][0], lg[q][1]}; l2[q][1] = l2[q][2] = l2[q][3] = f32x2{0.f, 0.f}; }
#pragma unroll
        for (int ky = 0; ky < 4; ++ky) {
            f32x4 v[5][FR / 4];                                   // the row's five window pixels
#pragma unroll
            for (int i = 0; i < 5; ++i)
#pragma unroll
                for (int c4 = 0; c4 < FR / 4; ++c4) v[i][c4] = *reinterpret_cast<const f32x4*>(m + mi(ty + ky, 2 * tx + i, c4));
#pragma unroll
            for (int kx = 0; kx < 4; ++kx) {
                cptr wp = (cptr)(a.wl + (ky * 4 + kx) * FR * 2);
#pragma unroll
                for (int c4 = 0; c4 < FR / 4; ++c4) {
                    const f32x2 w0 = f32x2{wp[(c4 * 4 + 0) * 2], wp[(c4 * 4 + 0) * 2 + 1]}, w1 = f32x2{wp[(c4 * 4 + 1) * 2], wp[(c4 * 4 + 1) * 2 + 1]};
                    const f32x2 w2 = f32x2{wp[(c4 * 4 + 2) * 2], wp[(c4 * 4 + 2) * 2 + 1]}, w3 = f32x2{wp[(c4 * 4 + 3) * 2], wp[(c4 * 4 + 3) * 2 + 1]};
#pragma unroll
                    for (int q = 0; q < 2; ++q) {
                        const f32x4 d = v[kx + q][c4];
                        pk_fma_bcast<0>(l2[q][0], f32x2{d.x, d.y}, w0);
                        pk_fma_bcast<1>(l2[q][1], f32x2{d.x, d.y}, w1);
                        pk_fma_bcast<0>(l2[q][2], f32x2{d.z, d.w}, w2);
                        pk_fma_bcast<1>(l2[q][3], f32x2{d.z, d.w}, w3);
                    }
                }
            }
        }
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const f32x2 t = (l2[q][0] + l2[q][1]) + (l2[q][2] + l2[q][3]);
            lg[q][0] = t.x; lg[q][1] = t.y;
        }
    } else {
#pragma unroll
        for (int ky = 0; ky < 4; ++ky)
#pragma unroll
            for (int kx = 0; kx < 4; ++kx) {
                const float* __restrict__ wp = a.wl + (ky * 4 + kx) * FR * NC;
#pragma unroll
                for (int q = 0; q < 2; ++q) {
#pragma unroll
                    for (int c = 0; c < FR; ++c) {
                        const float mv = m[mi(ty + ky, 2 * tx + q + kx, c >> 2) + (c & 3)];
#pragma unroll
                        for (int k = 0; k < NC; ++k) lg[q][k] = fmaf(mv, wp[c * NC + k], lg[q][k]);
                    }
                }
            }
    }
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int x = xb + q;
        if (x >= a.W) break;
        if (a.softmax) {
            float mx = lg[q][0];
#pragma unroll
            for (int k = 1; k < NC; ++k) mx = fmaxf(mx, lg[q][k]);
            float den = 0.f;
#pragma unroll
            for (int k = 0; k < NC; ++k) { lg[q][k] = expf(lg[q][k] - mx); den += lg[q][k]; }
#pragma unroll
            for (int k = 0; k < NC; ++k) lg[q][k] = lg[q][k] / den;
        }
        const idx_t p = ((idx_t)y * a.W + x) * NC;
#pragma unroll
        for (int k = 0; k < NC; ++k) {
            a.out[p + k] = lg[q][k];
            if (a.out_u8 || a.out_mask) {
                const uint8_t u = (uint8_t)(lg[q][k] * 255.0f);          // np.array(p*255, dtype=uint8)
                if (a.out_u8) a.out_u8[p + k] = u;
                if (a.out_mask) a.out_mask[p + k] = ((double)u > a.thr255) ? 255 : 0;
            }
        }
    }
}

}  // namespace asep
