// convr_kernel: the 64 -> 64 (and the 32 -> 64) 3x3 layers of the bf16 path with the WHOLE FILTER IN REGISTERS (round 6; the form the round-5 review left open).
//
// convb_kernel re-reads a layer's A fragments (72 KB for 64 -> 64) from L2 into LDS for every 8 x 32-pixel tile and reads them from LDS again for
// every 8 MFMAs; lesson 48: 37 % of these layers is fill time, three single-block pipelines lost to the two-block one-shot kernel.  Here a wave
// holds the layer's 2 x 9 x 4 A fragments (stage, tap, m-tile) = 288 registers of the 512 a lane has at ONE wave per SIMD (the unified VGPR / AGPR
// file of gfx950; MFMA A / B operands may be AGPRs), and IS the whole pipeline of a 32-column strip:
//   * no weights in LDS, no weight traffic after the first 72 loads of the wave, no barrier, no other wave to meet;
//   * the input rows of the strip (34 pixels x 64 channels) arrive by LDS-DMA (global_load_lds_dwordx4) in a ring of NR = 8 rows per wave, five
//     rows ahead of the MFMAs, retired by ONE counted s_waitcnt vmcnt per row; zero padding is a SOURCE ADDRESS (lanes outside the image fetch a
//     16-byte zero block), so border strips and rows take the same path;
//   * a row's LDS image = 4 planes of (34 pixels x 32 bytes): plane q holds channels 16 q .. 16 q + 15, lane (j, kk) of a B fragment reads pixel
//     j + kx, 16 bytes at (kk & 1) 16 of plane 2 g + (kk >> 1) -- convb_kernel's MODE-2 layout (conflict-free on the real ds_read_b128 lane groups);
//   * per output row: 2 n-tiles x 18 K chunks: 36 fragment reads, 144 MFMAs (0.25 KB of LDS reads per MFMA against convb_kernel's 0.75), the
//     accumulation order per output value is convb_kernel's (bias, stage 0 taps 0..8, stage 1 taps 0..8): BIT-IDENTICAL results;
//   * work = strip rows, linearised over (problem, strip, row) and cut into one contiguous range per wave (+-1 row: no tail), a range that crosses a
//     strip's end restarts its ring.
#pragma once
#include "bf16_kernels.h"

namespace asep {

struct ConvRProb {
    const bf16_t* in;      // [H,W,64]
    const bf16_t* res;     // residual [H,W,64] or nullptr
    bf16_t* out;           // [H,W,64]
    int H, W, strips, begin;   // strips = ceil(W / 32); begin = first linear strip row of the problem
};
struct ConvRArgs {
    ConvRProb p[MAXP];
    int nprob, total;      // total strip rows of all problems
    const u32x4* wpk;      // convb_kernel's packing: [chunk = stage 9 + tap][m-tile 4][lane] x 16 bytes
    const float* bias;     // [64]
    const void* zero;      // 16 zero bytes in device memory (source of the padding)
};

#ifndef CVR_NR
#define CVR_NR 8           // rows of a wave's input ring (>= 5): NR - 4 .. NR - 3 rows are in flight
#endif
#ifndef CVR_ABL
#define CVR_ABL 0          // ablation builds (WRONG RESULTS ON PURPOSE): 1 no wait for the rows, 2 no output stores, 4 no MFMAs, 8 no row requests, 16 no fragment reads
#endif

// Debug builds (-DCVR_TRACE): s_memtime stamps of lane 0 of every wave, read back through asep_debug_cvr_trace by scripts/gpu_cvr_trace.py
#if defined(CVR_TRACE)
__device__ unsigned long long g_cvr_trace[2048 * 8];
#define CVR_MARK(i, v) do { if (lane == 0 && wid < 2048) g_cvr_trace[wid * 8 + (i)] = (v); } while (0)
#define CVR_NOW() __builtin_amdgcn_s_memtime()
#else
#define CVR_MARK(i, v) do { } while (0)
#define CVR_NOW() 0ull
#endif

template <int N>
__device__ __forceinline__ void cvr_wait_vm() { if (!(CVR_ABL & 1)) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// One MFMA as an asm statement: the A operand is taken from the accumulator half of the register file ("a") where the kernel's filter lives, or
// from a VGPR tuple; hipcc neither schedules nor pads these (cdna_hip_programming.md 5.7): the statements keep their order, the first MFMA of an
// accumulator takes the bias REGISTERS as C, and the row's epilogue stands behind cvr_mfma_done's wait states.  Every statement opens with the two
// wait states a vector write of one of its operands needs: hipcc is free to put a v_mov into an operand's register right in front of the
// statement (it did: the RES form's accumulator [3][1] shared its registers with a fragment buffer and was copied in between two MFMAs; the
// second half of the copy stood directly in front of the MFMA that read it as C -- every other row wrong in two channels of four).  Between
// MFMAs of one wave the s_nop stands in the 8 idle issue cycles of the 16 an MFMA takes: no time.
template <bool AG>
__device__ __forceinline__ void cvr_mfma(f32x4& acc, const u32x4& A, const u32x4& B) {
    if constexpr (AG) asm volatile("s_nop 1\n\tv_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc) : "a"(A), "v"(B));
    else asm volatile("s_nop 1\n\tv_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(A), "v"(B));
}
template <bool AG>
__device__ __forceinline__ void cvr_mfma_first(f32x4& acc, const u32x4& A, const u32x4& B, const f32x4& c) {
    if constexpr (AG) asm volatile("s_nop 1\n\tv_mfma_f32_16x16x32_bf16 %0, %1, %2, %3" : "=&v"(acc) : "a"(A), "v"(B), "v"(c));
    else asm volatile("s_nop 1\n\tv_mfma_f32_16x16x32_bf16 %0, %1, %2, %3" : "=&v"(acc) : "v"(A), "v"(B), "v"(c));
}
// LDS reads as asm statements with their waits counted by hand (hipcc drains lgkmcnt to 0 in front of the first asm statement that uses a read's
// result).  LDS operations of a wave return in order.  cvr_frag: the two fragments of a K chunk (n-tiles 0 / 1) from row base (scalar) + the lane's
// place (vector); before chunk c the reads of chunks c + 1, c + 2 (four) may stay in flight.
template <int NT1>                                          // byte offset of the second n-tile: 16 pixels
__device__ __forceinline__ void cvr_frag(u32x4& b0, u32x4& b1, unsigned rowbase, unsigned place) {
    unsigned tmp;
    asm volatile("v_add_u32 %2, %3, %4\n\tds_read_b128 %0, %2\n\tds_read_b128 %1, %2 offset:%5" : "=v"(b0), "=v"(b1), "=&v"(tmp) : "s"(rowbase), "v"(place), "n"(NT1));
}
template <int OFF>
__device__ __forceinline__ void cvr_lds_read(u32x4& b, unsigned addr) { asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(b) : "v"(addr), "n"(OFF)); }
template <int N>
__device__ __forceinline__ void cvr_lds_wait(u32x4& b0, u32x4& b1) { asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(b0), "+v"(b1) : "n"(N)); }
template <int N>
__device__ __forceinline__ void cvr_lds_wait4(u32x4 (&b)[4]) { asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(b[0]), "+v"(b[1]), "+v"(b[2]), "+v"(b[3]) : "n"(N)); }
// an 8-pass MFMA's result may be read by a vector instruction 12 wait states behind it
__device__ __forceinline__ void cvr_mfma_done(f32x4 (&acc)[4][2]) {
    asm volatile("s_nop 11" : "+v"(acc[0][0]), "+v"(acc[0][1]), "+v"(acc[1][0]), "+v"(acc[1][1]), "+v"(acc[2][0]), "+v"(acc[2][1]), "+v"(acc[3][0]), "+v"(acc[3][1]));
}

// CIN: input channels, 64 (two stages of 32: 72 A fragments) or 32 (unet_down_3/conv1: 36 fragments, 64-byte pixels in the ring)
template <bool RELU_IN, bool RELU_OUT, bool RES = false, int CIN = 64>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void convr_kernel(const ConvRArgs a) {
    // RES (the block-closing convR_2: out = relu(conv(r) + t)): the residual row t arrives by four more requests per row in a staging row of its
    // own and becomes the accumulators' INITIAL value (bias + t, convb_kernel's RESP order) while the previous row's MFMAs run; loads return in
    // order, so its wait is the input rows' wait: a ring of five rows (nothing in flight behind the wait; the ring's depth does not show in the
    // layer's time, profiles/r6_convr)
    static_assert(CIN == 64 || (CIN == 32 && !RELU_IN && !RES), "64 input channels, or the block-opening 32 -> 64 conv1");
    constexpr int G = CIN / 32, NCH = 9 * G, PXB = CIN * 2, UPP = PXB / 16;           // stages, K chunks per row, bytes / 16-byte blocks per input pixel
    constexpr int NR = RES ? 5 : CVR_NR, ROWB = 34 * PXB, RINGB = NR * ROWB, STGB = 32 * 128, WAVEB = RINGB + STGB + (RES ? STGB : 0);
    constexpr int NU = 34 * UPP;                                                    // 16-byte units of a row image: 272 / 136
    constexpr int NDMA = (NU + 63) / 64;                                            // 5 (4 full wave-instructions + 16 lanes) / 3 (2 + 8 lanes)
    __shared__ __attribute__((aligned(16))) unsigned char lds[4 * WAVEB];          // 155648 bytes (RES: 119808)
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int j = lane & 15, kk = lane >> 4;
    unsigned char* const ring = lds + wave * WAVEB;
    unsigned char* const stg = ring + RINGB;                                        // the output row on its way from accumulator layout to whole pixels
    const unsigned ringa = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)ring;   // (wave-uniform LDS byte address)

    const int nwv = gridDim.x * 4, wid = blockIdx.x * 4 + wave;
    CVR_MARK(0, CVR_NOW());
    int lo = (int)((long long)a.total * wid / nwv);
    const int hi = (int)((long long)a.total * (wid + 1) / nwv);
    if (lo >= hi) return;
    CVR_MARK(6, (unsigned long long)(hi - lo));

    // A row's LDS image: pixel p of the strip's 34 at 128 p, its eight 16-byte channel blocks s at (s ^ (p & 6)) 16 -- whole pixels, so that eight
    // consecutive lanes of a request fetch one 128-byte line (the first cut kept convb_kernel's planes of 32 bytes per pixel: four requests per
    // line, 3.0 TB/s), permuted so that a fragment read (lane (j, kk): pixel j + kx, block 4 g + kk) is conflict-free on the real ds_read_b128
    // lane groups (exhaustive search over the XOR-linear maps, like r8v_px's).  DMA unit u = k 64 + lane -> pixel u >> 3, block (u & 7) ^ (p & 6).
    int dpx[NDMA], dch[NDMA];
#pragma unroll
    for (int k = 0; k < NDMA; ++k) {
        const int u = k * 64 + lane;
        dpx[k] = u / UPP;
        // (CIN 32: four blocks per pixel, block b at b ^ 2 (bit 2 of p): the same search, for 64-byte pixels)
        dch[k] = CIN == 64 ? ((u & 7) ^ (dpx[k] & 6)) * 16 : ((u & 3) ^ (((dpx[k] >> 2) & 1) << 1)) * 16;
    }
    // the lane's place inside a row image for a fragment read: [kx][g]; the second n-tile is 2048 bytes further
    unsigned fb[3][2];
#pragma unroll
    for (int kx = 0; kx < 3; ++kx)
#pragma unroll
        for (int g = 0; g < 2; ++g)
            fb[kx][g] = CIN == 64 ? (unsigned)((kx + j) * 128 + (((4 * g + kk) ^ ((kx + j) & 6)) * 16))
                                  : (unsigned)((kx + j) * 64 + ((kk ^ ((((kx + j) >> 2) & 1) << 1)) * 16));
    // staging: lane (j, kk) writes its 8 bytes (m-tile m, n-tile nt) of pixel 16 nt + j at 8-byte block ((4 m + kk) ^ ((j & 7) << 1)) of the pixel's
    // 128 (two lanes per bank pair instead of sixteen); read back linearly: lane l of instruction t gets the 16-byte block (l & 7) ^ ((l >> 3) & 7)
    // of pixel 8 t + (l >> 3), so eight lanes store one 128-byte line
    unsigned swb[4];
#pragma unroll
    for (int m = 0; m < 4; ++m)
        swb[m] = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)(stg + j * 128 + (((4 * m + kk) ^ ((j & 7) << 1)) * 8));
    const unsigned srd = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)(stg + lane * 16);
    const int gpx = lane >> 3, gl = gpx * 128 + (((lane & 7) ^ (gpx & 7)) * 16);
    const unsigned char* const zero = reinterpret_cast<const unsigned char*>(a.zero);

    // ---- a range of the wave: output rows r0 .. r0 + n - 1 of strip x0 of one problem ----
    // Memory instructions take a SCALAR base + a 32-bit lane offset, lanes outside the image are masked off (the first cut selected a 64-bit
    // address per lane between the row and a zero block / a dump: six vector instructions per request, between MFMAs that wait for them).
    int H = 0, W = 0, r0 = 0, n = 0;
    const unsigned char* inb = nullptr;
    unsigned voff[NDMA];             // byte offset of the lane's unit in an input row
    bool colok[NDMA];                // ... if its column is inside the image (and the unit inside the row image)
    bool okt[4];
    __attribute__((address_space(1))) unsigned char* orowp = nullptr;  // output row whose epilogue runs next (scalar); lane -> pixel x0 + 8 t + (lane >> 3), its 16-byte block: + gl + 1024 t
    size_t orow = 0;
    // request piece k of input row t (row r0 - 1 + t of the image) into ring slot t mod NR; rows outside the image and beyond the range: zeros
    // (every lane reads the zero block; the units of columns outside the image are zeroed once per range and never requested)
    const unsigned char* rq_base = nullptr;      // of the row being requested (scalars, set by request_row)
    unsigned rq_mask = 0, rq_dst = 0;
    const unsigned char* resb = nullptr;         // RES: the residual operand's row of the range's first output row, strip column 0 (scalar)
    auto request_res = [&](int r, int t) {       // piece t of residual row r0 + r into the residual staging row (whole pixels, the output staging's order)
        if (CVR_ABL & 8) return;
        const bool rowok = r < n;                                                   // (wave-uniform; a row beyond the range: zeros, never used)
        const unsigned char* const base = rowok ? resb + (size_t)r * orow : zero;
        const unsigned off = rowok ? (unsigned)(gl + t * 1024) : 0u;
        if (okt[t])
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(base + off),
                                             (__attribute__((address_space(3))) void*)(stg + STGB + t * 1024), 16, 0, 0);
    };
    auto request_row = [&](int t) {
        const int y = r0 - 1 + t;
        const bool rowok = y >= 0 && y < H && t <= n + 1;                           // (wave-uniform)
        rq_base = rowok ? inb + (size_t)y * W * PXB : zero;
        rq_mask = rowok ? ~0u : 0u;
        rq_dst = ((unsigned)t % NR) * ROWB;
    };
    auto request_piece = [&](int k) {
        if (CVR_ABL & 8) return;
        if (colok[k])
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(rq_base + (voff[k] & rq_mask)),
                                             (__attribute__((address_space(3))) void*)(ring + rq_dst + k * 1024), 16, 0, 0);
    };
    auto setup = [&](bool again) {
        int pi = 0;
        while (pi + 1 < a.nprob && lo >= a.p[pi + 1].begin) ++pi;
        const ConvRProb& P = a.p[pi];
        H = P.H; W = P.W;
        const int rel = lo - P.begin, s = rel / H;
        r0 = rel - s * H;
        n = min(H - r0, hi - lo);
        lo += n;
        const int x0 = s * 32;
        inb = reinterpret_cast<const unsigned char*>(P.in);
        if constexpr (RES) { resb = reinterpret_cast<const unsigned char*>(P.res) + ((size_t)r0 * W + x0) * 128; asm volatile("" : "+s"(resb)); }
        orowp = (__attribute__((address_space(1))) unsigned char*)reinterpret_cast<unsigned char*>(P.out) + ((size_t)r0 * W + x0) * 128;
        orow = (size_t)W * 128;
        // (the problem's scalars are HERE before the row loop: a scalar load hipcc cannot prove finished makes it drain lgkmcnt -- the fragment reads
        // in flight -- in front of the first use inside the loop)
        asm volatile("" : "+s"(H), "+s"(W), "+s"(r0), "+s"(n), "+s"(inb), "+s"(orowp), "+s"(orow));
#pragma unroll
        for (int k = 0; k < NDMA; ++k) {
            const int gx = x0 - 1 + dpx[k];
            colok[k] = gx >= 0 && gx < W && k * 64 + lane < NU;
            voff[k] = colok[k] ? (unsigned)(gx * PXB + dch[k]) : 0u;
        }
        if (x0 == 0 || x0 + 33 > W) {                                               // a strip at the image's left / right edge: zero columns
            if (again) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");             // (the previous range's last requests have landed)
#pragma unroll 1
            for (int slot = 0; slot < NR; ++slot)
#pragma unroll
                for (int k = 0; k < NDMA; ++k)
                    if (!colok[k] && k * 64 + lane < NU) *reinterpret_cast<u32x4*>(ring + slot * ROWB + k * 1024 + lane * 16) = u32x4{0u, 0u, 0u, 0u};
        }
#pragma unroll
        for (int t = 0; t < 4; ++t) okt[t] = x0 + 8 * t + gpx < W;
#pragma unroll 1
        for (int t = 0; t < NR - 1; ++t) {
            request_row(t);
#pragma unroll
            for (int k = 0; k < NDMA; ++k) request_piece(k);
        }
        if constexpr (RES) {
#pragma unroll
            for (int t = 0; t < 4; ++t) request_res(0, t);
        }
    };
    auto relu_row = [&](int t) {                                                    // in-place ReLU of a landed row (RELU_IN layers; range prologue)
        unsigned char* const rp = ring + ((unsigned)t % NR) * ROWB + lane * 16;
        u32x4 v[NDMA];
#pragma unroll
        for (int k = 0; k < NDMA; ++k)
            if (k * 64 + lane < NU) v[k] = *reinterpret_cast<const u32x4*>(rp + k * 1024);
#pragma unroll
        for (int k = 0; k < NDMA; ++k)
            if (k * 64 + lane < NU) *reinterpret_cast<u32x4*>(rp + k * 1024) = relu_bf16x8(v[k]);
    };

    // the first range's rows are requested BEFORE the filter is loaded (the loads behind them return in order: one wait covers both)
    setup(false);
    CVR_MARK(1, CVR_NOW());
    // ---- the layer's A fragments and biases: registers for the life of the wave ----
    // 64 of the 72 fragments are DEFINED in the accumulator half of the register file (loads into AGPR tuples, inline asm: a value the compiler
    // defines lives in a VGPR first, 288 of them do not fit 256, and what it then "spills" to AGPRs it copies back in front of every use:
    // 120 v_accvgpr_mov / _read per row in the first cut); the MFMAs take them from there.  The last 8 are ordinary values.
    constexpr int NA = NCH * 4, NAA = NA < 64 ? NA : 64;
    u32x4 Aa[NAA], Av[NA > NAA ? NA - NAA : 1];
    {
        const u32x4* wl = a.wpk + lane;
#pragma unroll
        for (int i = 0; i < NAA; ++i) asm volatile("global_load_dwordx4 %0, %1, off" : "=a"(Aa[i]) : "v"(wl + i * 64) : "memory");
#pragma unroll
        for (int i = NAA; i < NA; ++i) Av[i - NAA] = wl[i * 64];
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
        for (int i = 0; i < NAA; ++i) asm volatile("" : "+a"(Aa[i]));                 // (the uses below stay behind the wait)
    }
    f32x4 bias[4];
#pragma unroll
    for (int m = 0; m < 4; ++m) bias[m] = *reinterpret_cast<const f32x4*>(a.bias + m * 16 + kk * 4);
    CVR_MARK(2, CVR_NOW());

    // ---- the epilogue of a row, in pieces that stand between the MFMAs of the NEXT row (two accumulator sets take turns): round + ReLU on the
    //      packed values + 8-byte stores into the staging rows | the staged row read back as whole pixels | 16-byte stores ----
    u32x4 st[4];
    auto ep_pack = [&](f32x4 (&acc)[4][2], int m) {
        u32x2 q0 = pack_bf16x4(acc[m][0]), q1 = pack_bf16x4(acc[m][1]);
        if constexpr (RELU_OUT) {
            q0 = u32x2{relu_bf16x2(q0.x), relu_bf16x2(q0.y)};
            q1 = u32x2{relu_bf16x2(q1.x), relu_bf16x2(q1.y)};
        }
        *reinterpret_cast<__attribute__((address_space(3))) u32x2*>(swb[m]) = q0;
        *reinterpret_cast<__attribute__((address_space(3))) u32x2*>(swb[m] + 2048) = q1;
    };
    auto ep_read = [&]() {
        cvr_lds_read<0>(st[0], srd); cvr_lds_read<1024>(st[1], srd); cvr_lds_read<2048>(st[2], srd); cvr_lds_read<3072>(st[3], srd);
    };
    auto ep_store = [&](bool valid) {                                               // valid: wave-uniform (the first row of a range has no predecessor)
        if (CVR_ABL & 2) return;
        if (valid) {
#pragma unroll
            for (int t = 0; t < 4; ++t)
                if (okt[t]) *reinterpret_cast<__attribute__((address_space(1))) u32x4*>(orowp + (unsigned)(gl + t * 1024)) = st[t];
            orowp += orow;
        }
    };

    // ---- a row: 18 K chunks of 8 MFMAs; the fragment reads run two chunks ahead of their MFMAs (three buffers) ACROSS the row's end (chunks 16, 17
    //      read the next row's first fragments); everything else of the pipeline stands in the gaps between chunks: the previous row's epilogue
    //      (gaps 0, 1, 2, 4), the wait for input row i + 3 (4) and its ReLU (5, 6), the five requests of input row i + 7 (7 .. 11) ----
    u32x4 bq[3][2];
    auto frag = [&](auto cc, const unsigned (&sb)[4], int dk, u32x4 (&b)[2]) {      // chunk c of the row whose first ring row is sb[dk]
        constexpr int c = decltype(cc)::value, g = c / 9, t = c - g * 9, ky = t / 3, kx = t - ky * 3;
        if (CVR_ABL & 16) { b[0] = u32x4{(unsigned)c, 0u, 0u, 0u}; b[1] = b[0]; return; }
        cvr_frag<16 * PXB>(b[0], b[1], sb[dk + ky], fb[kx][g]);
    };
    u32x4 rr4;
    u32x2 rq[4][2];
    auto res_read = [&]() {                      // the lane's 8 bytes (m-tile, n-tile) of the staged residual row: the output staging's places, one row further
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            rq[m][0] = *reinterpret_cast<const __attribute__((address_space(3))) u32x2*>(swb[m] + STGB);
            rq[m][1] = *reinterpret_cast<const __attribute__((address_space(3))) u32x2*>(swb[m] + STGB + 2048);
        }
    };
    auto res_init = [&](auto, f32x4 (&acc)[4][2]) {                                 // acc = bias + residual (convb_kernel's order: b4, then += residual)
        // (ordinary loads: hipcc waits for them -- and with them for the fragment reads in flight, once per row; as asm reads with a counted wait
        // the last of the eight came back wrong in every other row)
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            acc[m][0] = bias[m]; acc[m][0] += unpack_bf16x4(rq[m][0]);
            acc[m][1] = bias[m]; acc[m][1] += unpack_bf16x4(rq[m][1]);
        }
    };
    auto row = [&](int i, f32x4 (&acc)[4][2], f32x4 (&accp)[4][2]) {
        unsigned sb[4];                                                             // LDS addresses of ring rows i .. i + 3 (scalars)
#pragma unroll
        for (int k = 0; k < 4; ++k) sb[k] = ringa + ((unsigned)(i + k) % NR) * ROWB;
        const unsigned rra = sb[3] + (unsigned)lane * 16u;
        static_for<NCH>([&](auto cc) {
            constexpr int c = decltype(cc)::value;
            if constexpr (c + 2 < NCH) frag(ic<c + 2>{}, sb, 0, bq[(c + 2) % 3]);
            else frag(ic<c + 2 - NCH>{}, sb, 1, bq[(c + 2) % 3]);                     // (ky = 0 of the next row: ring row i + 1)
            if (!(CVR_ABL & 16)) cvr_lds_wait<4>(bq[c % 3][0], bq[c % 3][1]);
            static_for<4>([&](auto mc) {
                constexpr int m = decltype(mc)::value, ia = c * 4 + m;
                constexpr bool AG = ia < NAA;
                const u32x4& Af = AG ? Aa[AG ? ia : 0] : Av[AG ? 0 : ia - NAA];
                if (CVR_ABL & 4) {
                    if (c == 0) { acc[m][0] = bias[m]; acc[m][1] = bias[m]; }
                    acc[m][0] += f32x4{__uint_as_float(bq[c % 3][0].x), 0.f, 0.f, 0.f};
                    acc[m][1] += f32x4{__uint_as_float(bq[c % 3][1].x), 0.f, 0.f, 0.f};
                } else if constexpr (c == 0) {
                    if constexpr (RES) {                                            // (the accumulators hold bias + residual)
                        cvr_mfma<AG>(acc[m][0], Af, bq[0][0]);
                        cvr_mfma<AG>(acc[m][1], Af, bq[0][1]);
                    } else {
                        cvr_mfma_first<AG>(acc[m][0], Af, bq[0][0], bias[m]);
                        cvr_mfma_first<AG>(acc[m][1], Af, bq[0][1], bias[m]);
                    }
                } else {
                    cvr_mfma<AG>(acc[m][0], Af, bq[c % 3][0]);
                    cvr_mfma<AG>(acc[m][1], Af, bq[c % 3][1]);
                }
            });
            __builtin_amdgcn_sched_barrier(0);
            // (the previous row's MFMAs ended at least a chunk ago: its accumulators need no wait states)
            if constexpr (c == 0) { ep_pack(accp, 0); ep_pack(accp, 1); }
            if constexpr (c == 1) { ep_pack(accp, 2); ep_pack(accp, 3); }
            if constexpr (c == 2) ep_read();
            if constexpr (c == 4) {
                // The wait for input row i + 3 (rows i + 4 .. i + 6 may be in flight) stands RIGHT IN FRONT of the row's stores: vmcnt counts stores
                // too, and a store that is still on its way when the next wait comes is waited for (v3 had the stores a chunk in front of the
                // wait: 3.8 -> 4.2 k ticks per row); here they have a whole row's time.
                cvr_wait_vm<(NR - 5) * NDMA>();                                     // (RES: NR = 5, and the residual row of output row i + 1 with it)
                cvr_lds_wait4<4>(st);                                               // (behind the staged reads stand the fragment reads of chunks 5, 6)
                ep_store(i > 0);
            }
            if constexpr (c == 5 && RES) res_read();                                // residual row of output row i + 1 -> the next row's accumulators
            if constexpr (c == 6 && RES) res_init(ic<2>{}, accp);                         // (behind the eight reads stand the fragment reads of chunk 8)
            if constexpr (c >= 12 && c < 16 && RES) request_res(i + 2, c - 12);      // (the staging row has been read)
            if constexpr (c == 5 && RELU_IN) {                                      // ReLU of input row i + 3 in place (the staging registers are free again)
                cvr_lds_read<0>(st[0], rra); cvr_lds_read<1024>(st[1], rra); cvr_lds_read<2048>(st[2], rra); cvr_lds_read<3072>(st[3], rra);
                cvr_lds_read<4096>(rr4, rra);                                       // (lanes >= 16 read into the next ring row or the staging rows: discarded)
            }
            if constexpr (c == 6 && RELU_IN) {
                // (behind the five reads stand the fragment reads of chunk 8)
                asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(st[0]), "+v"(st[1]), "+v"(st[2]), "+v"(st[3]), "+v"(rr4));
#pragma unroll
                for (int k = 0; k < 4; ++k) *reinterpret_cast<__attribute__((address_space(3))) u32x4*>(rra + k * 1024) = relu_bf16x8(st[k]);
                if (lane < 16) *reinterpret_cast<__attribute__((address_space(3))) u32x4*>(rra + 4096) = relu_bf16x8(rr4);
            }
            constexpr int RQ0 = NCH == 18 ? 7 : 5;                                  // (nine chunks: the requests right behind the stores)
            if constexpr (c == RQ0 - 1) request_row(i + NR - 1);                    // into the slot row i - 1 has left
            if constexpr (c >= RQ0 && c < RQ0 + NDMA) request_piece(c - RQ0);
            if constexpr (c <= 2 || (c >= 4 && c < RQ0 + NDMA) || (RES && c < 16)) __builtin_amdgcn_sched_barrier(0);
        });
        // No asm read may be in flight where hipcc is free to move registers (the loop's edges: the two row bodies use the accumulator sets and,
        // if its allocation says so, the fragment buffers in different registers, and a v_mov of a read's destination before the data has landed
        // copies the old value -- every other row of the RES form came out wrong in one m-tile).  The next row's first fragments were requested
        // 8 and 16 MFMAs ago: this wait is (almost) free.
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(bq[0][0]), "+v"(bq[0][1]), "+v"(bq[1][0]), "+v"(bq[1][1]));
        if (!(CVR_ABL & 4)) cvr_mfma_done(acc);                                     // (and no MFMA result younger than its wait states)
    };

    f32x4 acc0[4][2], acc1[4][2];
    int nseg = 0;
    for (;;) {
        if constexpr (RES) {
            cvr_wait_vm<0>();                                                       // input rows 0 .. 3 and residual row 0
            res_read();
            res_init(ic<0>{}, acc0);
#pragma unroll
            for (int t = 0; t < 4; ++t) request_res(1, t);
        } else if (nseg > 0) cvr_wait_vm<(NR - 4) * NDMA>();                        // rows 0 .. 2 have landed (rows 3 .. 6 may be in flight)
        if constexpr (RELU_IN) { relu_row(0); relu_row(1); relu_row(2); }
        {
            unsigned sb[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) sb[k] = ringa + (unsigned)(k * ROWB);
            frag(ic<0>{}, sb, 0, bq[0]);
            frag(ic<1>{}, sb, 0, bq[1]);
        }
        if (nseg == 0) CVR_MARK(3, CVR_NOW());
#pragma unroll 1
        for (int i = 0; i < n; i += 2) {
            row(i, acc0, acc1);
            if (i + 1 >= n) break;
            row(i + 1, acc1, acc0);
        }
        if (nseg == 0) CVR_MARK(4, CVR_NOW());
        ++nseg;
        // the reads of the row that does not follow, then the last row's epilogue
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(bq[0][0]), "+v"(bq[0][1]), "+v"(bq[1][0]), "+v"(bq[1][1]));
        auto flush = [&](f32x4 (&acc)[4][2]) {
            if (!(CVR_ABL & 4)) cvr_mfma_done(acc);
#pragma unroll
            for (int m = 0; m < 4; ++m) ep_pack(acc, m);
            ep_read();
            cvr_lds_wait4<0>(st);
            ep_store(true);
        };
        if (n & 1) flush(acc0); else flush(acc1);
        if (lo >= hi) break;
        setup(true);
    }
    CVR_MARK(5, CVR_NOW());
    CVR_MARK(7, (unsigned long long)nseg);
}

}  // namespace asep
