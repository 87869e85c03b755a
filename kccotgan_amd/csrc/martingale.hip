// Martingale penalty p_M and its backward (replaces gan_utils.py:179-201 of the reference):
//   N = M[:,1:,:] - M[:,:-1,:];  N_std = N / (std_{b,t}(M)[q] + 1e-6)   (population std)
//   s[t,q] = (1/B) sum_b N_std[b,t,q];  pM = lam * (sum_{t,q} |s[t,q]| * sc)
// A [B,T,J] feature tensor is a few thousand floats: one workgroup, everything staged in LDS.
#include "common.h"
#include <math.h>

namespace kccot {

// LDS layout (dynamic): mean[J] | stdv[J] | sabs[J] | s[(T-1)*J] | red[16]
__device__ __forceinline__ void martingale_stats(const float* __restrict__ M, int B, int T, int J,
                                                 float* mean, float* stdv, float* sabs, float* s) {
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, nw = blockDim.x >> 6;
    const int BT = B * T;
    for (int q = wid; q < J; q += nw) {
        float a = 0.f;
        for (int e = lane; e < BT; e += 64) a += M[(int64_t)e * J + q];
        const float mu = wave_sum(a) / (float)BT;
        float v = 0.f;
        for (int e = lane; e < BT; e += 64) {
            const float d = M[(int64_t)e * J + q] - mu;
            v = fmaf(d, d, v);
        }
        const float var = wave_sum(v) / (float)BT;
        if (lane == 0) { mean[q] = mu; stdv[q] = sqrtf(var); sabs[q] = 0.f; }
    }
    __syncthreads();
    for (int e = threadIdx.x; e < (T - 1) * J; e += blockDim.x) {
        const int q = e % J, t = e / J;
        const float den = stdv[q] + 1e-06f;
        float acc = 0.f;
        for (int b = 0; b < B; ++b) {
            const int64_t o = ((int64_t)b * T + t) * J + q;
            acc += (M[o + J] - M[o]) / den;
        }
        s[e] = acc / (float)B;
    }
    __syncthreads();
}

__global__ __launch_bounds__(1024) void martingale_fwd(const float* __restrict__ M, int B, int T, int J, float lam,
                                                       float sc, float* __restrict__ pm_out) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* mean = smem; float* stdv = mean + J; float* sabs = stdv + J; float* s = sabs + J;
    float* red = s + (T - 1) * J;
    martingale_stats(M, B, T, J, mean, stdv, sabs, s);
    float part = 0.f;
    for (int e = threadIdx.x; e < (T - 1) * J; e += blockDim.x) part += fabsf(s[e]);
    const float tot = block_sum(part, red);
    if (threadIdx.x == 0) pm_out[0] = lam * (tot * sc);
}

// dpM/dM[b,tau,q] = lam*sc * { [sgn(s[tau-1,q]) [tau>=1] - sgn(s[tau,q]) [tau<=T-2]] / (B (std_q+1e-6))
//                             - (sum_t |s[t,q]|) / (std_q+1e-6) * (M[b,tau,q]-mean_q) / (B T std_q) }
// (second line: the path through std; d std/dM = (M-mean)/(N std), as the gradient of
//  tf.math.reduce_std = sqrt(reduce_variance); taken as 0 where std = 0.)
__global__ __launch_bounds__(1024) void martingale_bwd(const float* __restrict__ M, int B, int T, int J, float lam,
                                                       float sc, const float* __restrict__ gpm, float* __restrict__ dM) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* mean = smem; float* stdv = mean + J; float* sabs = stdv + J; float* s = sabs + J;
    martingale_stats(M, B, T, J, mean, stdv, sabs, s);
    for (int q = threadIdx.x; q < J; q += blockDim.x) {
        float a = 0.f;
        for (int t = 0; t < T - 1; ++t) a += fabsf(s[t * J + q]);
        sabs[q] = a;
    }
    __syncthreads();
    const float k = gpm[0] * lam * sc;
    const int n = B * T * J;
    for (int e = threadIdx.x; e < n; e += blockDim.x) {
        const int q = e % J, tau = (e / J) % T;
        const float den = stdv[q] + 1e-06f;
        float sg = 0.f;
        if (tau >= 1) { const float v = s[(tau - 1) * J + q]; sg += (v > 0.f) - (v < 0.f); }
        if (tau <= T - 2) { const float v = s[tau * J + q]; sg -= (v > 0.f) - (v < 0.f); }
        float g = sg / ((float)B * den);
        if (stdv[q] > 0.f)
            g -= sabs[q] / den * (M[e] - mean[q]) / ((float)B * (float)T * stdv[q]);
        dM[e] = k * g;
    }
}

// Extension named by the north star, NOT reference behaviour (the reference only imports
// sklearn's rbf_kernel and never calls it, data_utils.py:16): K = exp(-gamma * D) on the three
// plain squared-distance matrices D3 = [xy, xx, yy] and the biased MMD^2 estimate
// mean(Kxx) + mean(Kyy) - 2 mean(Kxy).  One workgroup; sums in fp64, fixed order.
__global__ __launch_bounds__(1024) void rbf_mmd(const float* __restrict__ D3, int B, float gamma,
                                                float* __restrict__ K3, float* __restrict__ mmd_out) {
    __shared__ double part[3][16];
    const int n = B * B, lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    double tot[3];
    for (int p = 0; p < 3; ++p) {
        double s = 0.0;
        for (int e = threadIdx.x; e < n; e += blockDim.x) {
            const float k = expf(-gamma * D3[(int64_t)p * n + e]);
            if (K3) K3[(int64_t)p * n + e] = k;
            s += (double)k;
        }
        s = wave_sum_d(s);
        if (lane == 0) part[p][wid] = s;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int p = 0; p < 3; ++p) { tot[p] = 0.0; for (int w = 0; w < 16; ++w) tot[p] += part[p][w]; }
        mmd_out[0] = (float)((tot[1] + tot[2] - 2.0 * tot[0]) / (double)n);
    }
}

// d mmd / d D3: mmd = (sum Kxx + sum Kyy - 2 sum Kxy) / B^2 with K = exp(-gamma D), so
// gD3[p] = gmmd * coef_p * (-gamma) * K3[p] / B^2, coef = (-2, +1, +1) for (xy, xx, yy).
__global__ __launch_bounds__(256) void rbf_mmd_bwd(const float* __restrict__ K3, int B, float gamma,
                                                   const float* __restrict__ gmmd, float* __restrict__ gD3) {
    const int64_t n = (int64_t)B * B, e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= 3 * n) return;
    const float coef = e < n ? -2.f : 1.f;
    gD3[e] = gmmd[0] * coef * (-gamma) * K3[e] / (float)n;
}

}  // namespace kccot

using namespace kccot;

extern "C" int kccot_rbf_mmd_bwd_f32(const float* K3, int B, float gamma, const float* gmmd, float* gD3,
                                     kccot_stream_t stream) {
    if (!K3 || !gmmd || !gD3) return fail(KCCOT_EINVAL, "rbf_mmd_bwd: null pointer");
    if (B <= 0 || !(gamma > 0.f)) return fail(KCCOT_EINVAL, "rbf_mmd_bwd: bad arguments B=%d gamma=%g", B, (double)gamma);
    const int64_t n3 = (int64_t)3 * B * B;
    hipLaunchKernelGGL(rbf_mmd_bwd, dim3((unsigned)((n3 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, K3, B, gamma, gmmd, gD3);
    return launch_status("rbf_mmd_bwd");
}

extern "C" int kccot_rbf_mmd_f32(const float* D3, int B, float gamma, float* K3_out, float* mmd_out,
                                 kccot_stream_t stream) {
    if (!D3 || !mmd_out) return fail(KCCOT_EINVAL, "rbf_mmd: null pointer");
    if (B <= 0 || !(gamma > 0.f)) return fail(KCCOT_EINVAL, "rbf_mmd: bad arguments B=%d gamma=%g", B, (double)gamma);
    hipLaunchKernelGGL(rbf_mmd, dim3(1), dim3(1024), 0, (hipStream_t)stream, D3, B, gamma, K3_out, mmd_out);
    return launch_status("rbf_mmd");
}

static int martingale_check(const float* M, int B, int T, int J, size_t* lds) {
    if (!M) return fail(KCCOT_EINVAL, "martingale: null pointer");
    if (B <= 0 || T < 1 || J < 1) return fail(KCCOT_EINVAL, "martingale: bad shape B=%d T=%d J=%d", B, T, J);
    *lds = ((size_t)3 * J + (size_t)(T - 1) * J + 16) * sizeof(float);
    if (*lds > 64 * 1024) return fail(KCCOT_EUNSUPPORTED, "martingale: T*J = %d too large for one workgroup", T * J);
    return 0;
}

extern "C" int kccot_martingale_fwd_f32(const float* M, int B, int T, int J, float lam, float sc, float* pm_out,
                                        kccot_stream_t stream) {
    size_t lds;
    int rc = martingale_check(M, B, T, J, &lds);
    if (rc) return rc;
    if (!pm_out) return fail(KCCOT_EINVAL, "martingale_fwd: null output");
    hipLaunchKernelGGL(martingale_fwd, dim3(1), dim3(1024), lds, (hipStream_t)stream, M, B, T, J, lam, sc, pm_out);
    return launch_status("martingale_fwd");
}

extern "C" int kccot_martingale_bwd_f32(const float* M, int B, int T, int J, float lam, float sc, const float* gpm,
                                        float* dM, kccot_stream_t stream) {
    size_t lds;
    int rc = martingale_check(M, B, T, J, &lds);
    if (rc) return rc;
    if (!gpm || !dM) return fail(KCCOT_EINVAL, "martingale_bwd: null pointer");
    hipLaunchKernelGGL(martingale_bwd, dim3(1), dim3(1024), lds, (hipStream_t)stream, M, B, T, J, lam, sc, gpm, dM);
    return launch_status("martingale_bwd");
}
