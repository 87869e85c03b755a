// Which XCD does workgroup i of a 1-D grid land on?  Reads the XCC_ID hardware register per workgroup.
// build + run on the GPU box:  hipcc --offload-arch=gfx950 -O2 tools/micro/xcc_id.hip -o /tmp/xcc_id && /tmp/xcc_id
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>

__global__ void probe(int* out, int spin) {
    unsigned v;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
    // keep the workgroup resident for a while so that later ones cannot reuse its slot
    long long t0 = clock64();
    while (clock64() - t0 < spin) {}
    if (threadIdx.x == 0) out[blockIdx.x] = (int)(v & 0xF);
}

static void run(int nblocks, int threads, int spin, size_t lds) {
    int* d;
    hipMalloc(&d, nblocks * sizeof(int));
    hipLaunchKernelGGL(probe, dim3(nblocks), dim3(threads), lds, 0, d, spin);
    hipDeviceSynchronize();
    std::vector<int> h(nblocks);
    hipMemcpy(h.data(), d, nblocks * sizeof(int), hipMemcpyDeviceToHost);
    printf("grid %d x %d threads, lds %zu:", nblocks, threads, lds);
    int ok = 1;
    for (int i = 0; i < nblocks; ++i) { if (i < 48) printf(" %d", h[i]); if (h[i] != i % 8) ok = 0; }
    printf(" ...  id%%8 rule holds: %s\n", ok ? "yes" : "NO");
    hipFree(d);
}

int main() {
    run(64, 64, 20000, 0);
    run(128, 1024, 200000, 0);
    run(256, 512, 200000, 120 * 1024);
    run(2304, 512, 20000, 120 * 1024);
    return 0;
}
