// Which SIMD does wave w of a 512-thread workgroup run on?  The wave-specialised kernels (producers = waves 0-3,
// consumers = waves 4-7) assume wave w -> SIMD w % 4, i.e. one producer and one consumer per SIMD.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/simd_id.hip -o /tmp/simd_id && /tmp/simd_id
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ __launch_bounds__(1024) void who(unsigned* out) {
    unsigned hwid;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * 16 + (threadIdx.x >> 6)] = hwid;
}
int main() {
    unsigned* d; unsigned h[64 * 16];
    hipMalloc(&d, sizeof(h));
    const int blocks[3] = {512, 768, 1024};
    for (int b = 0; b < 3; ++b) {
        hipMemset(d, 0xff, sizeof(h));
        hipLaunchKernelGGL(who, dim3(8), dim3(blocks[b]), 0, 0, d);
        hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
        for (int g = 0; g < 3; ++g) {
            printf("%4d threads, workgroup %d: wave -> (SIMD, wave slot, CU):", blocks[b], g);
            for (int w = 0; w < blocks[b] / 64; ++w) {
                const unsigned v = h[g * 16 + w];
                printf("  %d->(%u,%u,%u)", w, (v >> 4) & 3, v & 15, (v >> 8) & 15);
            }
            printf("\n");
        }
    }
    return 0;
}
