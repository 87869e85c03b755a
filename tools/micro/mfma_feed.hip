// What halves the MFMA rate in the large-batch consumers?  The consumer loop of apply_coeffs_x3_m256, stripped to
// variants: constant operands / rotating register operands / + LDS fragment reads / + global fragment loads /
// + a workgroup barrier per 8 k-steps with four idle partner waves.  Chip-wide rate of each, to be read against the
// 2.48 PFLOP/s of tools/micro/mfma_peak.hip.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/mfma_feed.hip -o /tmp/mfma_feed && /tmp/mfma_feed
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <stdlib.h>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int COLP = 264, PLANE = 64 * COLP, BUF = 3 * PLANE;

// MODE bits: 1 = rotate A operands through a 4-deep register ring (no loads), 2 = B fragments from LDS every step,
// 4 = A fragments from global memory every step (fragment-major, 1 KiB contiguous), 8 = barrier per 8 steps (8 waves:
// waves 0-3 only take part in the barriers), 16 = no sched_barrier around the MFMA groups
template <int MODE, int RANDOM>
__global__ __launch_bounds__(512) void feed(const unsigned short* __restrict__ W, float* out, int chunks) {
    __shared__ __attribute__((aligned(16))) unsigned char zs[2 * BUF];
    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    // RANDOM != 0: operand bits as in real data (random bf16 in [1,2) x random sign): switching activity of the matrix
    // pipe, hence power and clock, depends on the data
    for (int i = t; i < 2 * BUF / 4; i += blockDim.x) {
        unsigned h = (i * 2654435761u) ^ (blockIdx.x * 40503u);
        h ^= h >> 13; h *= 0x5bd1e995u; h ^= h >> 15;
        reinterpret_cast<unsigned*>(zs)[i] = RANDOM ? ((h & 0x807f807fu) | 0x3f803f80u) : 0x3f803f80u;
    }
    __syncthreads();
    if ((MODE & 8) && wave < 4) {
        for (int c = 0; c <= chunks; ++c) __syncthreads();
        return;
    }
    if (!(MODE & 8) && wave < 4) return;
    const int w = wave - 4;
    bf16x8 A[4][2][3];
    const unsigned short* wb = W + ((int64_t)blockIdx.x * 4 + w) * 64 * 512 + lane * 8;
    auto ldA = [&](int g, int slot) {
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) {
            A[slot][0][pl] = *reinterpret_cast<const bf16x8*>(wb + (int64_t)(pl * 2 + 0) * 16 * 512 + (g & 15) * 512);
            A[slot][1][pl] = *reinterpret_cast<const bf16x8*>(wb + (int64_t)(pl * 2 + 1) * 16 * 512 + (g & 15) * 512);
        }
    };
    for (int sl = 0; sl < 4; ++sl) ldA(sl, sl);
    const int boff0 = (lane & 31) * COLP + 16 * (lane >> 5), boff1 = boff0 + 32 * COLP;
    bf16x8 Bf[2][2][3];
    auto ldB = [&](const unsigned char* zb, int st, int par) {
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) {
            uint2* p0 = reinterpret_cast<uint2*>(&Bf[par][0][pl]);
            uint2* p1 = reinterpret_cast<uint2*>(&Bf[par][1][pl]);
            p0[0] = *reinterpret_cast<const uint2*>(zb + pl * PLANE + boff0 + 32 * st);
            p0[1] = *reinterpret_cast<const uint2*>(zb + pl * PLANE + boff0 + 32 * st + 8);
            p1[0] = *reinterpret_cast<const uint2*>(zb + pl * PLANE + boff1 + 32 * st);
            p1[1] = *reinterpret_cast<const uint2*>(zb + pl * PLANE + boff1 + 32 * st + 8);
        }
    };
    ldB(zs, 0, 0); ldB(zs, 1, 1);
    f32x16 acc00, acc01, acc10, acc11;
    for (int r = 0; r < 16; ++r) { acc00[r] = 0.f; acc01[r] = 0.f; acc10[r] = 0.f; acc11[r] = 0.f; }
    if (MODE & 8) __syncthreads();
    int buf = 0;
    for (int c = 0; c < chunks; ++c, buf ^= 1) {
        const unsigned char* zb = zs + buf * BUF;
        if (MODE & 2) ldB(zb, 0, 0);
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            if (MODE & 4) ldA(c * 8 + s + 3, (s + 3) & 3);
            if ((MODE & 2) && s < 7) ldB(zb, s + 1, (s + 1) & 1);
            if (!(MODE & 16)) __builtin_amdgcn_sched_barrier(0);
            constexpr int dummy = 0; (void)dummy;
            const int sa = (MODE & 5) ? (s & 3) : 0, sb = (MODE & 2) ? (s & 1) : 0;
#define A4(PA, PB)                                                                                  \
            acc00 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[sa][0][PA], Bf[sb][0][PB], acc00, 0, 0, 0);    \
            acc01 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[sa][0][PA], Bf[sb][1][PB], acc01, 0, 0, 0);    \
            acc10 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[sa][1][PA], Bf[sb][0][PB], acc10, 0, 0, 0);    \
            acc11 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[sa][1][PA], Bf[sb][1][PB], acc11, 0, 0, 0);
            A4(1, 1) A4(0, 2) A4(2, 0) A4(0, 1) A4(1, 0) A4(0, 0)
#undef A4
            if (!(MODE & 16)) __builtin_amdgcn_sched_barrier(0);
        }
        if (MODE & 8) __syncthreads();
    }
    float sres = 0.f;
    for (int r = 0; r < 16; ++r) sres += acc00[r] + acc01[r] + acc10[r] + acc11[r];
    out[blockIdx.x * blockDim.x + t] = sres;
}

template <int MODE, int RANDOM = 0>
static void run(const char* what, const unsigned short* W, float* out) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int chunks = 2000;
    float ms = 0.f;
    for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((feed<MODE, RANDOM>), dim3(256), dim3(512), 0, 0, W, out, chunks);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
    }
    const double nm = 256.0 * 4 * chunks * 8 * 24;
    printf("%-78s %8.3f ms  %.3f PFLOP/s\n", what, ms, nm * 32768.0 / ms / 1e12);
}

int main() {
    unsigned short* W; float* out;
    hipMalloc(&W, (size_t)256 * 4 * 64 * 512 * 2 + (1 << 20));
    hipMemset(W, 0x3f, (size_t)256 * 4 * 64 * 512 * 2 + (1 << 20));
    hipMalloc(&out, sizeof(float) * 256 * 512);
    run<0>("4 consumer waves, constant operands, sched barriers", W, out);
    run<16>("constant operands, no sched barriers", W, out);
    run<1>("A operands rotate through a 4-deep register ring", W, out);
    run<2>("+ B fragments from LDS every k-step (one step ahead)", W, out);
    run<3>("A ring + B from LDS", W, out);
    run<4>("+ A fragments from global memory every k-step (three steps ahead), L2-resident", W, out);
    run<6>("A from global + B from LDS", W, out);
    run<8>("constant operands + workgroup barrier per 8 k-steps (4 idle partner waves)", W, out);
    run<14>("A from global + B from LDS + barrier per 8 k-steps", W, out);
    // the same with data-like operand bits
    {
        const size_t n = (size_t)256 * 4 * 64 * 512 + (1 << 19);
        unsigned short* h = (unsigned short*)malloc(n * 2);
        unsigned x = 12345u;
        for (size_t i = 0; i < n; ++i) { x = x * 1664525u + 1013904223u; h[i] = (unsigned short)(((x >> 16) & 0x807f) | 0x3f80); }
        hipMemcpy(W, h, n * 2, hipMemcpyHostToDevice);
        free(h);
    }
    run<3, 1>("RANDOM operand bits: A ring (registers, loaded once from random W) + B from LDS", W, out);
    run<11, 1>("RANDOM operand bits: A ring + B from LDS + barrier per 8 k-steps", W, out);
    run<6, 1>("RANDOM operand bits: A from global + B from LDS", W, out);
    return 0;
}
