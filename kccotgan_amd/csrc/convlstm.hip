// ConvLSTM cell update behind the two convolutions (kccot_convlstm_cell_{fwd,bwd}_f32, include/kccot.h): the gate
// arithmetic of one time step -- a dozen elementwise launches per step and layer when written with stock tensor ops,
// twice that in the backward -- as one streaming kernel each way.  Memory-bound: forward reads 9 F and writes 2 F
// floats per (b, pixel), backward reads 11 F and writes 5 F.  float4 over the F*HW run of one sample when it allows.
#include "common.h"
#pragma clang fp contract(off)      // hard_sigmoid = clip(0.2 x + 0.5): product and sum rounded separately, as the tensor ops do

namespace kccot {

__device__ __forceinline__ float hsig(float x) { return fminf(fmaxf(0.2f * x + 0.5f, 0.f), 1.f); }
__device__ __forceinline__ float dhsig(float x) {          // derivative of clip(0.2 x + 0.5, 0, 1): torch.clamp passes the
    const float y = 0.2f * x + 0.5f;                        // gradient where min <= y <= max (bounds included)
    return (y >= 0.f && y <= 1.f) ? 0.2f : 0.f;
}

template <int VW>
__global__ __launch_bounds__(256) void convlstm_cell_fwd(const float* __restrict__ gx, const float* __restrict__ gh,
                                                         const float* __restrict__ c_prev, int64_t n, int64_t FHW,
                                                         float* __restrict__ c_out, float* __restrict__ h_out) {
    const int64_t e = ((int64_t)blockIdx.x * 256 + threadIdx.x) * VW;
    if (e >= n) return;
    const int64_t b = e / FHW, r = e - b * FHW, gb = b * 4 * FHW + r;
    float gi[VW], gf[VW], gc[VW], go[VW], cp[VW], co[VW], ho[VW];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        float* dst = q == 0 ? gi : (q == 1 ? gf : (q == 2 ? gc : go));
        if constexpr (VW == 4) {
            const float4 a = *reinterpret_cast<const float4*>(gx + gb + q * FHW), c = *reinterpret_cast<const float4*>(gh + gb + q * FHW);
            dst[0] = a.x + c.x; dst[1] = a.y + c.y; dst[2] = a.z + c.z; dst[VW - 1] = a.w + c.w;
        } else {
            dst[0] = gx[gb + q * FHW] + gh[gb + q * FHW];
        }
    }
    if constexpr (VW == 4) {
        const float4 c = *reinterpret_cast<const float4*>(c_prev + e);
        cp[0] = c.x; cp[1] = c.y; cp[2] = c.z; cp[VW - 1] = c.w;
    } else cp[0] = c_prev[e];
#pragma unroll
    for (int j = 0; j < VW; ++j) {
        const float c = hsig(gf[j]) * cp[j] + hsig(gi[j]) * tanhf(gc[j]);
        co[j] = c;
        ho[j] = hsig(go[j]) * tanhf(c);
    }
    if constexpr (VW == 4) {
        *reinterpret_cast<float4*>(c_out + e) = make_float4(co[0], co[1], co[2], co[VW - 1]);
        *reinterpret_cast<float4*>(h_out + e) = make_float4(ho[0], ho[1], ho[2], ho[VW - 1]);
    } else { c_out[e] = co[0]; h_out[e] = ho[0]; }
}

template <int VW>
__global__ __launch_bounds__(256) void convlstm_cell_bwd(const float* __restrict__ gx, const float* __restrict__ gh,
                                                         const float* __restrict__ c_prev, const float* __restrict__ c_out,
                                                         const float* __restrict__ dh, const float* __restrict__ dc_out,
                                                         int64_t n, int64_t FHW, float* __restrict__ dg,
                                                         float* __restrict__ dc_prev) {
    const int64_t e = ((int64_t)blockIdx.x * 256 + threadIdx.x) * VW;
    if (e >= n) return;
    const int64_t b = e / FHW, r = e - b * FHW, gb = b * 4 * FHW + r;
    float g[4][VW], cp[VW], c[VW], dH[VW], dC[VW];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        if constexpr (VW == 4) {
            const float4 a = *reinterpret_cast<const float4*>(gx + gb + q * FHW), x = *reinterpret_cast<const float4*>(gh + gb + q * FHW);
            g[q][0] = a.x + x.x; g[q][1] = a.y + x.y; g[q][2] = a.z + x.z; g[q][VW - 1] = a.w + x.w;
        } else {
            g[q][0] = gx[gb + q * FHW] + gh[gb + q * FHW];
        }
    }
    auto ld = [&](const float* p, float (&dst)[VW]) {
        if (!p) {
#pragma unroll
            for (int j = 0; j < VW; ++j) dst[j] = 0.f;
        } else if constexpr (VW == 4) {
            const float4 v = *reinterpret_cast<const float4*>(p + e);
            dst[0] = v.x; dst[1] = v.y; dst[2] = v.z; dst[VW - 1] = v.w;
        } else dst[0] = p[e];
    };
    ld(c_prev, cp); ld(c_out, c); ld(dh, dH); ld(dc_out, dC);
    float di[VW], df[VW], dcc[VW], dO[VW], dcp[VW];
#pragma unroll
    for (int j = 0; j < VW; ++j) {
        const float i = hsig(g[0][j]), f = hsig(g[1][j]), cc = tanhf(g[2][j]), o = hsig(g[3][j]);
        const float tc = tanhf(c[j]);
        const float dcj = dC[j] + dH[j] * o * (1.f - tc * tc);
        dO[j] = dH[j] * tc * dhsig(g[3][j]);
        df[j] = dcj * cp[j] * dhsig(g[1][j]);
        di[j] = dcj * cc * dhsig(g[0][j]);
        dcc[j] = dcj * i * (1.f - cc * cc);
        dcp[j] = dcj * f;
    }
    auto st = [&](float* p, int64_t at, const float (&src)[VW]) {
        if constexpr (VW == 4) *reinterpret_cast<float4*>(p + at) = make_float4(src[0], src[1], src[2], src[VW - 1]);
        else p[at] = src[0];
    };
    st(dg, gb, di); st(dg, gb + FHW, df); st(dg, gb + 2 * FHW, dcc); st(dg, gb + 3 * FHW, dO);
    st(dc_prev, e, dcp);
}

static bool cell_v4(int64_t FHW, const void* a, const void* b, const void* c, const void* d, const void* e2, const void* f,
                    const void* g, const void* h) {
    const uintptr_t m = (uintptr_t)a | (uintptr_t)b | (uintptr_t)c | (uintptr_t)d | (uintptr_t)e2 | (uintptr_t)f | (uintptr_t)g | (uintptr_t)h;
    return FHW % 4 == 0 && (m & 15) == 0;
}

}  // namespace kccot
using namespace kccot;

extern "C" int kccot_convlstm_cell_fwd_f32(const float* gx, const float* gh, const float* c_prev, int B, int F, int HW,
                                           float* c_out, float* h_out, kccot_stream_t stream) {
    if (!gx || !gh || !c_prev || !c_out || !h_out) return fail(KCCOT_EINVAL, "convlstm_cell_fwd: null pointer");
    if (B <= 0 || F <= 0 || HW <= 0) return fail(KCCOT_EINVAL, "convlstm_cell_fwd: bad shape B=%d F=%d HW=%d", B, F, HW);
    const int64_t FHW = (int64_t)F * HW, n = (int64_t)B * FHW;
    hipStream_t st = (hipStream_t)stream;
    if (cell_v4(FHW, gx, gh, c_prev, c_out, h_out, nullptr, nullptr, nullptr))
        hipLaunchKernelGGL(convlstm_cell_fwd<4>, dim3((unsigned)((n / 4 + 255) / 256)), dim3(256), 0, st, gx, gh, c_prev, n, FHW, c_out, h_out);
    else
        hipLaunchKernelGGL(convlstm_cell_fwd<1>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, gx, gh, c_prev, n, FHW, c_out, h_out);
    return launch_status("convlstm_cell_fwd");
}

extern "C" int kccot_convlstm_cell_bwd_f32(const float* gx, const float* gh, const float* c_prev, const float* c_out,
                                           const float* dh, const float* dc_out, int B, int F, int HW, float* dg,
                                           float* dc_prev, kccot_stream_t stream) {
    if (!gx || !gh || !c_prev || !c_out || !dg || !dc_prev) return fail(KCCOT_EINVAL, "convlstm_cell_bwd: null pointer");
    if (B <= 0 || F <= 0 || HW <= 0) return fail(KCCOT_EINVAL, "convlstm_cell_bwd: bad shape B=%d F=%d HW=%d", B, F, HW);
    const int64_t FHW = (int64_t)F * HW, n = (int64_t)B * FHW;
    hipStream_t st = (hipStream_t)stream;
    if (cell_v4(FHW, gx, gh, c_prev, c_out, dh, dc_out, dg, dc_prev))
        hipLaunchKernelGGL(convlstm_cell_bwd<4>, dim3((unsigned)((n / 4 + 255) / 256)), dim3(256), 0, st, gx, gh, c_prev, c_out, dh, dc_out, n,
                           FHW, dg, dc_prev);
    else
        hipLaunchKernelGGL(convlstm_cell_bwd<1>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, gx, gh, c_prev, c_out, dh, dc_out, n,
                           FHW, dg, dc_prev);
    return launch_status("convlstm_cell_bwd");
}
