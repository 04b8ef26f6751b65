// Semantics check of v_permlane16_swap_b32 (gfx950) as used by the bf16 level-0 kernels: prints, for lanes 0, 16, 32, 48, what
// the two results hold when a = 1000 + lane, b = 2000 + lane.     hipcc --offload-arch=gfx950 permlane16_swap.hip -o permlane16_swap
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(unsigned* out) {
    const unsigned lane = threadIdx.x;
    const unsigned a = 1000 + lane, b = 2000 + lane;
    auto r = __builtin_amdgcn_permlane16_swap(a, b, false, false);
    out[lane] = r[0];
    out[64 + lane] = r[1];
}
int main() {
    unsigned* d;
    hipMalloc(&d, 128 * 4);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    unsigned h[128];
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    for (int l : {0, 5, 16, 21, 32, 48}) printf("lane %2d: r0 = %u  r1 = %u\n", l, h[l], h[64 + l]);
    return 0;
}
