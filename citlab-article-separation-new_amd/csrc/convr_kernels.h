// convr_kernel: the 64 -> 64 3x3 layers of the bf16 path with the WHOLE FILTER IN REGISTERS (round 6; the form the round-5 review left open).
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
    void* trash;           // >= 4 KB of device memory nobody reads (stores of lanes beyond the image's last column)
};

#ifndef CVR_ABL
#define CVR_ABL 0          // ablation builds (WRONG RESULTS ON PURPOSE): 1 no wait for the rows, 2 no output stores, 4 no MFMAs, 8 no row requests, 16 no fragment reads
#endif

template <int N>
__device__ __forceinline__ void cvr_wait_vm() { if (!(CVR_ABL & 1)) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// One MFMA as an asm statement: the A operand is taken from the accumulator half of the register file ("a") where the kernel's filter lives, or
// from a VGPR tuple; hipcc neither schedules nor pads these (cdna_hip_programming.md 5.7): the statements keep their order, the first MFMA of an
// accumulator takes the bias REGISTERS as C (no vector write in front of it), and the row's epilogue stands behind cvr_mfma_done's wait states.
template <bool AG>
__device__ __forceinline__ void cvr_mfma(f32x4& acc, const u32x4& A, const u32x4& B) {
    if constexpr (AG) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc) : "a"(A), "v"(B));
    else asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(A), "v"(B));
}
template <bool AG>
__device__ __forceinline__ void cvr_mfma_first(f32x4& acc, const u32x4& A, const u32x4& B, const f32x4& c) {
    if constexpr (AG) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %3" : "=&v"(acc) : "a"(A), "v"(B), "v"(c));
    else asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %3" : "=&v"(acc) : "v"(A), "v"(B), "v"(c));
}
// A fragment read as an asm statement with its wait counted by hand: hipcc drains lgkmcnt to 0 in front of the first asm statement that uses a
// read's result (every third chunk waited for the reads issued right in front of it).  LDS operations of a wave return in order: before chunk c
// the reads of chunks c + 1, c + 2 (four) may stay in flight.
template <int OFF>
__device__ __forceinline__ void cvr_lds_read(u32x4& b, unsigned addr) { asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(b) : "v"(addr), "n"(OFF)); }
template <int N>
__device__ __forceinline__ void cvr_lds_wait(u32x4& b0, u32x4& b1) { asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(b0), "+v"(b1) : "n"(N)); }
// an 8-pass MFMA's result may be read by a vector instruction 12 wait states behind it
__device__ __forceinline__ void cvr_mfma_done(f32x4 (&acc)[4][2]) {
    asm volatile("s_nop 11" : "+v"(acc[0][0]), "+v"(acc[0][1]), "+v"(acc[1][0]), "+v"(acc[1][1]), "+v"(acc[2][0]), "+v"(acc[2][1]), "+v"(acc[3][0]), "+v"(acc[3][1]));
}

template <bool RELU_IN, bool RELU_OUT>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void convr_kernel(const ConvRArgs a) {
    constexpr int NR = 8, ROWB = 34 * 128, RINGB = NR * ROWB, STGB = 32 * 128, WAVEB = RINGB + STGB;   // 4352, 34816, 4096, 38912 bytes
    constexpr int NDMA = 5;                                                         // 272 16-byte units per row: 4 full wave-instructions + 16 lanes
    __shared__ __attribute__((aligned(16))) unsigned char lds[4 * WAVEB];          // 155648 bytes
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int j = lane & 15, kk = lane >> 4;
    unsigned char* const ring = lds + wave * WAVEB;
    unsigned char* const stg = ring + RINGB;                                        // the output row on its way from accumulator layout to whole pixels

    // ---- the layer's A fragments and biases: registers for the life of the wave ----
    // 64 of the 72 fragments are DEFINED in the accumulator half of the register file (loads into AGPR tuples, inline asm: a value the compiler
    // defines lives in a VGPR first, 288 of them do not fit 256, and what it then "spills" to AGPRs it copies back in front of every use:
    // 120 v_accvgpr_mov / _read per row in the first cut); the MFMAs take them from there.  The last 8 are ordinary values.
    constexpr int NAA = 64;
    u32x4 Aa[NAA], Av[72 - NAA];
    {
        const u32x4* wl = a.wpk + lane;
#pragma unroll
        for (int i = 0; i < NAA; ++i) asm volatile("global_load_dwordx4 %0, %1, off" : "=a"(Aa[i]) : "v"(wl + i * 64) : "memory");
#pragma unroll
        for (int i = NAA; i < 72; ++i) Av[i - NAA] = wl[i * 64];
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
        for (int i = 0; i < NAA; ++i) asm volatile("" : "+a"(Aa[i]));                 // (the uses below stay behind the wait)
    }
    f32x4 bias[4];
#pragma unroll
    for (int m = 0; m < 4; ++m) bias[m] = *reinterpret_cast<const f32x4*>(a.bias + m * 16 + kk * 4);

    const int nwv = gridDim.x * 4, wid = blockIdx.x * 4 + wave;
    int lo = (int)((long long)a.total * wid / nwv);
    const int hi = (int)((long long)a.total * (wid + 1) / nwv);

    // A row's LDS image: pixel p of the strip's 34 at 128 p, its eight 16-byte channel blocks s at (s ^ (p & 6)) 16 -- whole pixels, so that eight
    // consecutive lanes of a request fetch one 128-byte line (the first cut kept convb_kernel's planes of 32 bytes per pixel: four requests per
    // line, 3.0 TB/s), permuted so that a fragment read (lane (j, kk): pixel j + kx, block 4 g + kk) is conflict-free on the real ds_read_b128
    // lane groups (exhaustive search over the XOR-linear maps, like r8v_px's).  DMA unit u = k 64 + lane -> pixel u >> 3, block (u & 7) ^ (p & 6).
    int dpx[NDMA], dch[NDMA];
#pragma unroll
    for (int k = 0; k < NDMA; ++k) {
        const int u = k * 64 + lane;
        dpx[k] = u >> 3;
        dch[k] = ((u & 7) ^ (dpx[k] & 6)) * 16;
    }
    // fragment addresses inside a row image: [kx][g]; the second n-tile is 2048 bytes further
    int fb[3][2];
#pragma unroll
    for (int kx = 0; kx < 3; ++kx)
#pragma unroll
        for (int g = 0; g < 2; ++g) fb[kx][g] = (kx + j) * 128 + (((4 * g + kk) ^ ((kx + j) & 6)) * 16);
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
    unsigned char* const trash = reinterpret_cast<unsigned char*>(a.trash) + lane * 16;

    while (lo < hi) {
        int pi = 0;
        while (pi + 1 < a.nprob && lo >= a.p[pi + 1].begin) ++pi;
        const ConvRProb& P = a.p[pi];
        const int H = P.H, W = P.W;
        const int rel = lo - P.begin, s = rel / H, r0 = rel - s * H;
        const int n = min(H - r0, hi - lo);                                         // output rows r0 .. r0 + n - 1 of strip s
        lo += n;
        const int x0 = s * 32;
        const unsigned char* const inb = reinterpret_cast<const unsigned char*>(P.in);
        // byte offset of the lane's unit in an input row (-1: outside the image's columns, or the 5th instruction's idle lanes)
        int doff[NDMA];
#pragma unroll
        for (int k = 0; k < NDMA; ++k) {
            const int gx = x0 - 1 + dpx[k];
            doff[k] = (gx >= 0 && gx < W && k * 64 + lane < 272) ? gx * 128 + dch[k] : -1;
        }
        // request input row t (row r0 - 1 + t of the image) into ring slot t & 7; rows outside the image and beyond the range: zeros
        auto request = [&](int t) {
            if (CVR_ABL & 8) return;
            const int y = r0 - 1 + t;
            const bool rowok = y >= 0 && y < H && t <= n + 1;                       // (wave-uniform)
            const unsigned char* const rowp = inb + (size_t)(rowok ? y : 0) * W * 128;
            unsigned char* const dst = ring + (t & (NR - 1)) * ROWB;
#pragma unroll
            for (int k = 0; k < NDMA; ++k) {
                const unsigned char* src = (rowok && doff[k] >= 0) ? rowp + doff[k] : zero;
                if (k < NDMA - 1 || lane < 16)
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                                     (__attribute__((address_space(3))) void*)(dst + k * 1024), 16, 0, 0);
            }
        };
        auto relu_row = [&](int t) {                                                // in-place ReLU of a landed row (RELU_IN layers)
            unsigned char* const rp = ring + (t & (NR - 1)) * ROWB + lane * 16;
            u32x4 v[NDMA];
#pragma unroll
            for (int k = 0; k < NDMA; ++k)
                if (k < NDMA - 1 || lane < 16) v[k] = *reinterpret_cast<const u32x4*>(rp + k * 1024);
#pragma unroll
            for (int k = 0; k < NDMA; ++k)
                if (k < NDMA - 1 || lane < 16) *reinterpret_cast<u32x4*>(rp + k * 1024) = relu_bf16x8(v[k]);
        };

#pragma unroll 1
        for (int t = 0; t < NR - 1; ++t) request(t);
        if constexpr (RELU_IN) {
            cvr_wait_vm<(NR - 3) * NDMA>();                                         // rows 0, 1 have landed (rows 2 .. 6 may be in flight)
            relu_row(0);
            relu_row(1);
        }
        // output addressing (whole pixels): lane -> pixel x0 + 8 t + (lane >> 3) of the row, its 16-byte block
        unsigned char* const outb = reinterpret_cast<unsigned char*>(P.out);
        bool okt[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) okt[t] = x0 + 8 * t + gpx < W;
        // (a pointer in registers before the loop: an address built from the problem's scalars inside it made hipcc drain lgkmcnt -- the fragment
        // reads in flight -- in front of the stores, for a scalar load it could not prove finished)
        unsigned char* optr = outb + ((size_t)r0 * W + x0) * 128 + gl;              // of the row whose epilogue runs next
        asm volatile("" : "+v"(optr));
        const size_t orow = (size_t)W * 128;

        // ---- the epilogue of a row, in four pieces that stand between the MFMAs of the NEXT row (two accumulator sets take turns): round + ReLU on
        //      the packed values + 8-byte stores into the staging rows | the staged row read back as whole pixels | 16-byte stores ----
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
        auto ep_store = [&](bool valid) {                                           // valid: wave-uniform (the first row of a range has no predecessor)
            if (CVR_ABL & 2) return;
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                unsigned char* const o = (valid && okt[t]) ? optr + t * 1024 : trash;
                *reinterpret_cast<u32x4*>(o) = st[t];
            }
            optr += valid ? orow : 0;
        };

        f32x4 acc0[4][2], acc1[4][2];
        auto row = [&](int i, f32x4 (&acc)[4][2], f32x4 (&accp)[4][2]) {
            request(i + NR - 1);                                                    // into the slot row i - 1 has left
            cvr_wait_vm<(NR - 3) * NDMA>();                                         // rows .. i + 2 have landed; i + 3 .. i + 7 may be in flight
            if constexpr (RELU_IN) relu_row(i + 2);
            unsigned rb[3][3][2];                                                   // LDS byte addresses: [row i + ky][kx][g]
#pragma unroll
            for (int ky = 0; ky < 3; ++ky) {
                const unsigned rowa = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)(ring + ((i + ky) & (NR - 1)) * ROWB);
#pragma unroll
                for (int kx = 0; kx < 3; ++kx)
#pragma unroll
                    for (int g = 0; g < 2; ++g) rb[ky][kx][g] = rowa + fb[kx][g];
            }
            u32x4 bq[3][2];
            auto ldb = [&](auto cc, u32x4 (&b)[2]) {
                constexpr int c = decltype(cc)::value, g = c / 9, t = c - g * 9, ky = t / 3, kx = t - ky * 3;
                if (CVR_ABL & 16) { b[0] = u32x4{(unsigned)c, 0u, 0u, 0u}; b[1] = b[0]; return; }
                cvr_lds_read<0>(b[0], rb[ky][kx][g]);
                cvr_lds_read<2048>(b[1], rb[ky][kx][g]);
            };
            // the fragment reads run two chunks ahead of their MFMAs (three buffers)
            ldb(ic<0>{}, bq[0]);
            ldb(ic<1>{}, bq[1]);
            static_for<18>([&](auto cc) {
                constexpr int c = decltype(cc)::value;
                if constexpr (c + 2 < 18) ldb(ic<c + 2>{}, bq[(c + 2) % 3]);
                if (!(CVR_ABL & 16)) cvr_lds_wait<(c + 2 < 18 ? 4 : (c + 1 < 18 ? 2 : 0))>(bq[c % 3][0], bq[c % 3][1]);
                static_for<4>([&](auto mc) {
                    constexpr int m = decltype(mc)::value, ia = c * 4 + m;
                    constexpr bool AG = ia < NAA;
                    const u32x4& Af = AG ? Aa[AG ? ia : 0] : Av[AG ? 0 : ia - NAA];
                    if (CVR_ABL & 4) {
                        if (c == 0) { acc[m][0] = bias[m]; acc[m][1] = bias[m]; }
                        acc[m][0] += f32x4{__uint_as_float(bq[c % 3][0].x), 0.f, 0.f, 0.f};
                        acc[m][1] += f32x4{__uint_as_float(bq[c % 3][1].x), 0.f, 0.f, 0.f};
                    } else if constexpr (c == 0) {
                        cvr_mfma_first<AG>(acc[m][0], Af, bq[0][0], bias[m]);
                        cvr_mfma_first<AG>(acc[m][1], Af, bq[0][1], bias[m]);
                    } else {
                        cvr_mfma<AG>(acc[m][0], Af, bq[c % 3][0]);
                        cvr_mfma<AG>(acc[m][1], Af, bq[c % 3][1]);
                    }
                });
                // the previous row's epilogue (its MFMAs ended at least a chunk ago: no wait states needed)
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (c == 1) { ep_pack(accp, 0); ep_pack(accp, 1); }
                if constexpr (c == 3) { ep_pack(accp, 2); ep_pack(accp, 3); }
                if constexpr (c == 5) ep_read();
                if constexpr (c == 9) {
                    // (LDS operations return in order: behind the staged reads stand the fragment reads of chunks 8 .. 11)
                    asm volatile("s_waitcnt lgkmcnt(8)" : "+v"(st[0]), "+v"(st[1]), "+v"(st[2]), "+v"(st[3]));
                    ep_store(i > 0);
                }
                if constexpr (c == 1 || c == 3 || c == 5 || c == 9) __builtin_amdgcn_sched_barrier(0);
            });
        };
#pragma unroll 1
        for (int i = 0; i < n; i += 2) {
            row(i, acc0, acc1);
            if (i + 1 >= n) break;
            row(i + 1, acc1, acc0);
        }
        // the last row's epilogue
        auto flush = [&](f32x4 (&acc)[4][2]) {
            if (!(CVR_ABL & 4)) cvr_mfma_done(acc);
#pragma unroll
            for (int m = 0; m < 4; ++m) ep_pack(acc, m);
            ep_read();
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(st[0]), "+v"(st[1]), "+v"(st[2]), "+v"(st[3]));
            ep_store(true);
        };
        if (n & 1) flush(acc0); else flush(acc1);
    }
}

}  // namespace asep
