// LDS instruction throughput of one CU on gfx950, by width and direction: N waves (one block) issue the same ds instruction back to back on
// conflict-free addresses (lane i -> byte i * width), cycles per instruction and bytes per cycle of the CU.
//   hipcc -O3 --offload-arch=gfx950 lds_rw_rate.hip -o lds_rw_rate && ./lds_rw_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
#define N_IT 512
#define UNR 16
template <int OP>
__global__ __launch_bounds__(1024) void k(unsigned long long* cyc, float* out) {
    __shared__ u32x4 lds[4096];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < 4096; i += blockDim.x) lds[i] = u32x4{1u, 2u, 3u, 4u};
    __syncthreads();
    u32x4 v = {1u, 2u, 3u, 4u}, r4 = {0, 0, 0, 0};
    u32x2 r2 = {0, 0}; unsigned r1 = 0;
    const unsigned a16 = (unsigned)((wave & 3) * 16384 + lane * 16), a8 = (unsigned)((wave & 3) * 16384 + lane * 8), a4 = (unsigned)((wave & 3) * 16384 + lane * 4);
    // stride-32 pattern of a 32-byte-per-pixel tile (a lane's 16-byte half of its pixel): the conv1 (UP) / res16f access shape
    const unsigned a32 = (unsigned)((wave & 3) * 16384 + (lane & 15) * 64 + (lane >> 4) * 16);
    // res16f / convb MODE 1 patterns (32 bytes per pixel): read = pixel j + (kk >> 1), half kk & 1; D-fragment store = pixel j, 8 bytes at kk * 8
    const unsigned a16r = (unsigned)((wave & 3) * 16384 + ((lane & 15) + (lane >> 5)) * 32 + ((lane >> 4) & 1) * 16);
    const unsigned a16w = (unsigned)((wave & 3) * 16384 + (lane & 15) * 32 + (lane >> 4) * 8);
    // the planar alternative: plane kk & 1 (8 KB apart), pixel j + (kk >> 1), 16 bytes; store: plane kk >> 1, pixel j, 8 bytes at (kk & 1) * 8
    const unsigned a16rp = (unsigned)((wave & 3) * 16384 + ((lane >> 4) & 1) * 8192 + ((lane & 15) + (lane >> 5)) * 16);
    const unsigned a16wp = (unsigned)((wave & 3) * 16384 + (lane >> 5) * 8192 + (lane & 15) * 16 + ((lane >> 4) & 1) * 8);
    // res8f whole-pixel store of a tile pair: lane (j, kk) -> row kk & 1, pixel 2 j + (kk >> 1), 16 bytes; and the parity-planar alternative
    // (pixel p in plane p & 1 at index p >> 1): row kk & 1, plane kk >> 1, index j
    const unsigned a8w = (unsigned)((wave & 3) * 16384 + ((lane >> 4) & 1) * 608 + (lane & 15) * 32 + (lane >> 5) * 16);
    const unsigned a8wp = (unsigned)((wave & 3) * 16384 + ((lane >> 4) & 1) * 608 + (lane >> 5) * 304 + (lane & 15) * 16);
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < N_IT; ++it) {
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            if constexpr (OP == 0) asm volatile("ds_write_b128 %0, %1 offset:%2" ::"v"(a16), "v"(v), "n"(u * 1024));
            else if constexpr (OP == 1) asm volatile("ds_write_b64 %0, %1 offset:%2" ::"v"(a8), "v"(u32x2{v.x, v.y}), "n"(u * 512));
            else if constexpr (OP == 2) asm volatile("ds_write_b32 %0, %1 offset:%2" ::"v"(a4), "v"(v.x), "n"(u * 256));
            else if constexpr (OP == 3) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(r4) : "v"(a16), "n"(u * 1024));
            else if constexpr (OP == 4) asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(r2) : "v"(a8), "n"(u * 512));
            else if constexpr (OP == 5) asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(r1) : "v"(a4), "n"(u * 256));
            else if constexpr (OP == 6) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(r4) : "v"(a32), "n"(u * 1024));
            else if constexpr (OP == 7) asm volatile("ds_write_b128 %0, %1 offset:%2" ::"v"(a32), "v"(v), "n"(u * 1024));
            else if constexpr (OP == 8) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(r4) : "v"(a16r), "n"(u * 256));
            else if constexpr (OP == 9) asm volatile("ds_write_b64 %0, %1 offset:%2" ::"v"(a16w), "v"(u32x2{v.x, v.y}), "n"(u * 256));
            else if constexpr (OP == 10) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(r4) : "v"(a16rp), "n"(u * 128));
            else if constexpr (OP == 12) asm volatile("ds_write_b128 %0, %1 offset:%2" ::"v"(a8w), "v"(v), "n"(u * 1216));
            else if constexpr (OP == 13) asm volatile("ds_write_b128 %0, %1 offset:%2" ::"v"(a8wp), "v"(v), "n"(u * 1216));
            else if constexpr (OP == 11) asm volatile("ds_write_b64 %0, %1 offset:%2" ::"v"(a16wp), "v"(u32x2{v.x, v.y}), "n"(u * 128));
        }
        asm volatile("s_waitcnt lgkmcnt(0)");
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (lane == 0 && blockIdx.x == 0) cyc[wave] = t1 - t0;
    out[tid] = (float)(r4.x + r2.x + r1);
}
template <int OP> void run(const char* name, int bytes, unsigned long long* c, float* d) {
    for (int nw : {4, 8, 16}) {
        unsigned long long h[16] = {0};
        for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL((k<OP>), dim3(256), dim3(64 * nw), 0, 0, c, d);
        (void)hipDeviceSynchronize();
        (void)hipMemcpy(h, c, sizeof(h), hipMemcpyDeviceToHost);
        const double cyc = (double)h[0] / (N_IT * UNR);
        printf("%-34s %2d waves: %6.2f ticks per instr and wave, %7.1f bytes per tick and CU\n", name, nw, cyc, nw * 64.0 * bytes / cyc);
    }
}
int main() {
    unsigned long long* c; float* d;
    (void)hipMalloc(&c, 16 * sizeof(unsigned long long)); (void)hipMalloc(&d, 1024 * sizeof(float));
    printf("ticks = s_memtime (100 MHz): RATIOS between rows are what counts\n");
    run<0>("ds_write_b128 (contiguous)", 16, c, d); run<1>("ds_write_b64", 8, c, d); run<2>("ds_write_b32", 4, c, d);
    run<3>("ds_read_b128 (contiguous)", 16, c, d); run<4>("ds_read_b64", 8, c, d); run<5>("ds_read_b32", 4, c, d);
    run<6>("ds_read_b128 (16 B of 64 B pairs)", 16, c, d); run<7>("ds_write_b128 (16 B of 64 B pairs)", 16, c, d);
    run<8>("ds_read_b128 res16f B fragment", 16, c, d); run<10>("ds_read_b128 the same, planar", 16, c, d);
    run<12>("ds_write_b128 res8f pixel records", 16, c, d); run<13>("ds_write_b128 the same, parity planes", 16, c, d);
    run<9>("ds_write_b64 res16f D fragment", 8, c, d); run<11>("ds_write_b64 the same, planar", 8, c, d);
    return 0;
}
