// Backward of the pairwise cost matrices: dLoss/dC -> dLoss/d(video), dLoss/dh, dLoss/dM.
//
// With C[i,j] = sc*||x_i - y_j||^2 + sc*sum_{t<T-1,q} h[i,t,q]*(M[j,t+1,q]-M[j,t,q]) and g = dLoss/dC:
//     dx_i = 2 sc ( (sum_j g_ij) x_i - sum_j g_ij y_j )
//     dy_j = 2 sc ( (sum_i g_ij) y_j - sum_i g_ij x_i )
//     dh[i,t,q]  = sc sum_j g_ij (M[j,t+1,q]-M[j,t,q])                      (t < T-1, else 0)
//     dM[j,t,q]  = sc ( sum_i g_ij h[i,t-1,q] [t>=1]  -  sum_i g_ij h[i,t,q] [t<=T-2] )
// i.e. every video gradient is a small coefficient matrix W (B x 2B, built from g) applied to the
// stacked rows [X;Y]:  d[m,k] = sum_r W[m,r] Z[r,k]  -- one streaming pass over the videos
// (reads 2*B*K*4 bytes, writes B*K*4), the coefficients served from the scalar cache.
// For the loss (gan_utils.py:221-223) fake is y in the xy term and both x and y in the yy term:
//     dfake_m = 2sc( cs_xy[m] y_m - sum_i gxy[i,m] x_i ) + 2sc( (rs_yy[m]+cs_yy[m]) y_m - sum_r (gyy[m,r]+gyy[r,m]) y_r )
#include "common.h"

namespace kccot {

enum CoeffMode { CO_LOSS3_DFAKE = 0, CO_DX = 1, CO_DY = 2, CO_SAME = 3 };

// Wt is [R][Bout] (stack-row major) so that one output row block reads contiguous scalars.
// R = n1 + n2 stack rows: first the n1 rows of src1, then the n2 rows of src2.
__global__ __launch_bounds__(256) void build_coeffs(int mode, const float* __restrict__ g, const float* __restrict__ g2,
                                                    int Bx, int By, float sc, float* __restrict__ Wt) {
    // g: [Bx,By] (LOSS3: gxy [B,B]); g2: LOSS3 only: gyy [B,B]
    const int Bout = (mode == CO_DY) ? By : (mode == CO_LOSS3_DFAKE ? By : Bx);
    const int R = (mode == CO_SAME) ? Bx : Bx + By;
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= R * Bout) return;
    const int r = e / Bout, m = e % Bout;
    const float two_sc = 2.f * sc;
    float w = 0.f;
    if (mode == CO_LOSS3_DFAKE) {
        const int B = Bx;
        if (r < B) {
            w = -two_sc * g[(int64_t)r * B + m];                       // -2sc gxy[r][m] * x_r
        } else {
            const int rr = r - B;
            w = -two_sc * (g2[(int64_t)m * B + rr] + g2[(int64_t)rr * B + m]);
            if (rr == m) {
                float d = 0.f;
                for (int i = 0; i < B; ++i)
                    d += g[(int64_t)i * B + m] + g2[(int64_t)m * B + i] + g2[(int64_t)i * B + m];
                w += two_sc * d;
            }
        }
    } else if (mode == CO_DX) {        // out rows = x rows
        if (r < Bx) {
            if (r == m) { float d = 0.f; for (int j = 0; j < By; ++j) d += g[(int64_t)m * By + j]; w = two_sc * d; }
        } else {
            w = -two_sc * g[(int64_t)m * By + (r - Bx)];
        }
    } else if (mode == CO_DY) {        // out rows = y rows
        if (r < Bx) {
            w = -two_sc * g[(int64_t)r * By + m];
        } else if (r - Bx == m) {
            float d = 0.f; for (int i = 0; i < Bx; ++i) d += g[(int64_t)i * By + m]; w = two_sc * d;
        }
    } else {                           // CO_SAME: x is y
        w = -two_sc * (g[(int64_t)m * Bx + r] + g[(int64_t)r * Bx + m]);
        if (r == m) {
            float d = 0.f;
            for (int i = 0; i < Bx; ++i) d += g[(int64_t)m * Bx + i] + g[(int64_t)i * Bx + m];
            w += two_sc * d;
        }
    }
    Wt[e] = w;
}

// out[m0+mm][k] = sum_r Wt[r][m0+mm] * Z_r[k];  one column k per thread, MB output rows per block row.
template <int MB>
__global__ __launch_bounds__(256) void apply_coeffs(const float* __restrict__ Wt, const float* __restrict__ src1, int n1,
                                                    const float* __restrict__ src2, int n2, int Bout, int64_t K,
                                                    float* __restrict__ out) {
    const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int m0 = blockIdx.y * MB;
    const bool kok = k < K;
    float acc[MB];
#pragma unroll
    for (int mm = 0; mm < MB; ++mm) acc[mm] = 0.f;
    const float* wrow = Wt + m0;
    for (int r = 0; r < n1; ++r) {
        const float z = kok ? src1[(int64_t)r * K + k] : 0.f;
#pragma unroll
        for (int mm = 0; mm < MB; ++mm) {
            const float w = (m0 + mm < Bout) ? wrow[(int64_t)r * Bout + mm] : 0.f;
            acc[mm] = fmaf(w, z, acc[mm]);
        }
    }
    for (int r = 0; r < n2; ++r) {
        const float z = kok ? src2[(int64_t)r * K + k] : 0.f;
#pragma unroll
        for (int mm = 0; mm < MB; ++mm) {
            const float w = (m0 + mm < Bout) ? wrow[(int64_t)(n1 + r) * Bout + mm] : 0.f;
            acc[mm] = fmaf(w, z, acc[mm]);
        }
    }
    if (kok) {
#pragma unroll
        for (int mm = 0; mm < MB; ++mm)
            if (m0 + mm < Bout) out[(int64_t)(m0 + mm) * K + k] = acc[mm];
    }
}

// dh[i,t,q] = sc * ( sum_j gA[i,j] dMA[j,t,q] + sum_j gB[i,j] dMB[j,t,q] ), dM = M[t+1]-M[t]; 0 at t = T-1
__global__ __launch_bounds__(256) void causal_grad_h(float* __restrict__ out, int Bi, int Bj, int T, int J, float sc,
                                                     const float* __restrict__ gA, const float* __restrict__ MA,
                                                     const float* __restrict__ gB, const float* __restrict__ MB) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= Bi * T * J) return;
    const int q = e % J, t = (e / J) % T, i = e / (J * T);
    float s = 0.f;
    if (t < T - 1) {
        for (int j = 0; j < Bj; ++j) {
            const int64_t o = ((int64_t)j * T + t) * J + q;
            if (gA) s = fmaf(gA[(int64_t)i * Bj + j], MA[o + J] - MA[o], s);
            if (gB) s = fmaf(gB[(int64_t)i * Bj + j], MB[o + J] - MB[o], s);
        }
    }
    out[e] = s * sc;
}

// dM[j,t,q] = sc * sum_i ( gA[i,j] (hA[i,t-1,q][t>=1] - hA[i,t,q][t<=T-2]) + same for B )
__global__ __launch_bounds__(256) void causal_grad_M(float* __restrict__ out, int Bi, int Bj, int T, int J, float sc,
                                                     const float* __restrict__ gA, const float* __restrict__ hA,
                                                     const float* __restrict__ gB, const float* __restrict__ hB) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= Bj * T * J) return;
    const int q = e % J, t = (e / J) % T, j = e / (J * T);
    float s = 0.f;
    for (int i = 0; i < Bi; ++i) {
        const int64_t o = ((int64_t)i * T + t) * J + q;
        if (gA) {
            const float hd = (t >= 1 ? hA[o - J] : 0.f) - (t <= T - 2 ? hA[o] : 0.f);
            s = fmaf(gA[(int64_t)i * Bj + j], hd, s);
        }
        if (gB) {
            const float hd = (t >= 1 ? hB[o - J] : 0.f) - (t <= T - 2 ? hB[o] : 0.f);
            s = fmaf(gB[(int64_t)i * Bj + j], hd, s);
        }
    }
    out[e] = s * sc;
}

static int launch_apply(const float* Wt, const float* s1, int n1, const float* s2, int n2, int Bout, int64_t K,
                        float* out, hipStream_t st) {
    const unsigned gx = (unsigned)((K + 255) / 256);
    if (Bout <= 8) hipLaunchKernelGGL(apply_coeffs<8>, dim3(gx, 1), dim3(256), 0, st, Wt, s1, n1, s2, n2, Bout, K, out);
    else if (Bout <= 16) hipLaunchKernelGGL(apply_coeffs<16>, dim3(gx, 1), dim3(256), 0, st, Wt, s1, n1, s2, n2, Bout, K, out);
    else if (Bout <= 32) hipLaunchKernelGGL(apply_coeffs<32>, dim3(gx, 1), dim3(256), 0, st, Wt, s1, n1, s2, n2, Bout, K, out);
    else hipLaunchKernelGGL(apply_coeffs<64>, dim3(gx, (Bout + 63) / 64), dim3(256), 0, st, Wt, s1, n1, s2, n2, Bout, K, out);
    return launch_status("apply_coeffs");
}

}  // namespace kccot

using namespace kccot;

extern "C" size_t kccot_pairwise_cost3_bwd_workspace_bytes(int B, int64_t K) {
    (void)K;
    if (B <= 0) return 0;
    return align_up((size_t)2 * B * B * sizeof(float), 256);
}

extern "C" int kccot_pairwise_cost3_bwd_f32(const float* g3, const float* real, const float* fake, int B,
                                            int64_t K, float sc, const float* h_fake, const float* h_real,
                                            const float* m_real, const float* m_fake, int T, int J,
                                            float* dfake, float* dh_fake, float* dh_real, float* dm_real,
                                            float* dm_fake, void* ws, size_t ws_bytes, kccot_stream_t stream) {
    if (!g3 || !real || !fake) return fail(KCCOT_EINVAL, "pairwise_cost3_bwd: null pointer");
    if (B <= 0 || K <= 0 || T < 1 || J < 1)
        return fail(KCCOT_EINVAL, "pairwise_cost3_bwd: bad shape B=%d K=%lld T=%d J=%d", B, (long long)K, T, J);
    if ((dh_fake || dh_real || dm_real || dm_fake) && (!h_fake || !h_real || !m_real || !m_fake))
        return fail(KCCOT_EINVAL, "pairwise_cost3_bwd: feature gradients requested without features");
    hipStream_t st = (hipStream_t)stream;
    const int64_t bb = (int64_t)B * B;
    const float *gxy = g3, *gxx = g3 + bb, *gyy = g3 + 2 * bb;
    int rc;
    if (dfake) {
        const size_t need = kccot_pairwise_cost3_bwd_workspace_bytes(B, K);
        if (!ws || ws_bytes < need)
            return fail(KCCOT_EWORKSPACE, "pairwise_cost3_bwd: workspace %zu < required %zu", ws_bytes, need);
        float* Wt = static_cast<float*>(ws);
        hipLaunchKernelGGL(build_coeffs, dim3((2 * B * B + 255) / 256), dim3(256), 0, st, (int)CO_LOSS3_DFAKE,
                           gxy, gyy, B, B, sc, Wt);
        if ((rc = launch_status("build_coeffs"))) return rc;
        if ((rc = launch_apply(Wt, real, B, fake, B, B, K, dfake, st))) return rc;
    }
    const int nf = B * T * J;
    const dim3 fg((nf + 255) / 256), fb(256);
    // gan_utils.py:221-223: h_fake rows of xy (cols m_real) and of yy (cols m_fake); h_real rows of xx (cols m_real)
    if (dh_fake) hipLaunchKernelGGL(causal_grad_h, fg, fb, 0, st, dh_fake, B, B, T, J, sc, gxy, m_real, gyy, m_fake);
    if (dh_real) hipLaunchKernelGGL(causal_grad_h, fg, fb, 0, st, dh_real, B, B, T, J, sc, gxx, m_real,
                                    (const float*)nullptr, (const float*)nullptr);
    // m_real cols of xy (rows h_fake) and of xx (rows h_real); m_fake cols of yy (rows h_fake)
    if (dm_real) hipLaunchKernelGGL(causal_grad_M, fg, fb, 0, st, dm_real, B, B, T, J, sc, gxy, h_fake, gxx, h_real);
    if (dm_fake) hipLaunchKernelGGL(causal_grad_M, fg, fb, 0, st, dm_fake, B, B, T, J, sc, gyy, h_fake,
                                    (const float*)nullptr, (const float*)nullptr);
    return launch_status("causal_grad");
}

extern "C" size_t kccot_pairwise_cost_bwd_workspace_bytes(int Bx, int By) {
    if (Bx <= 0 || By <= 0) return 0;
    const size_t mx = Bx > By ? Bx : By;
    return 2 * align_up((size_t)(Bx + By) * mx * sizeof(float), 256);
}

extern "C" int kccot_pairwise_cost_bwd_f32(const float* g, const float* x, const float* y, int Bx, int By,
                                           int64_t K, float sc, const float* h, const float* M, int T, int J,
                                           unsigned flags, float* dx, float* dy, float* dh, float* dM,
                                           void* ws, size_t ws_bytes, kccot_stream_t stream) {
    if (!g || !x || !y) return fail(KCCOT_EINVAL, "pairwise_cost_bwd: null pointer");
    if (Bx <= 0 || By <= 0 || K <= 0)
        return fail(KCCOT_EINVAL, "pairwise_cost_bwd: bad shape Bx=%d By=%d K=%lld", Bx, By, (long long)K);
    const bool same = (flags & KCCOT_COST_SAME) != 0;
    if (same && (x != y || Bx != By || dy))
        return fail(KCCOT_EINVAL, "pairwise_cost_bwd: KCCOT_COST_SAME needs x == y, Bx == By and dy == NULL");
    if ((dh || dM) && (!h || !M || T < 1 || J < 1))
        return fail(KCCOT_EINVAL, "pairwise_cost_bwd: feature gradients requested without features");
    hipStream_t st = (hipStream_t)stream;
    int rc;
    if (dx || dy) {
        const size_t need = kccot_pairwise_cost_bwd_workspace_bytes(Bx, By);
        if (!ws || ws_bytes < need)
            return fail(KCCOT_EWORKSPACE, "pairwise_cost_bwd: workspace %zu < required %zu", ws_bytes, need);
    }
    float* W1 = static_cast<float*>(ws);
    float* W2 = reinterpret_cast<float*>(static_cast<char*>(ws) + kccot_pairwise_cost_bwd_workspace_bytes(Bx, By) / 2);
    if (same) {
        if (dx) {
            hipLaunchKernelGGL(build_coeffs, dim3((Bx * Bx + 255) / 256), dim3(256), 0, st, (int)CO_SAME, g,
                               (const float*)nullptr, Bx, Bx, sc, W1);
            if ((rc = launch_status("build_coeffs"))) return rc;
            if ((rc = launch_apply(W1, x, Bx, x, 0, Bx, K, dx, st))) return rc;
        }
    } else {
        const int R = Bx + By;
        if (dx) {
            hipLaunchKernelGGL(build_coeffs, dim3((R * Bx + 255) / 256), dim3(256), 0, st, (int)CO_DX, g,
                               (const float*)nullptr, Bx, By, sc, W1);
            if ((rc = launch_status("build_coeffs"))) return rc;
            if ((rc = launch_apply(W1, x, Bx, y, By, Bx, K, dx, st))) return rc;
        }
        if (dy) {
            hipLaunchKernelGGL(build_coeffs, dim3((R * By + 255) / 256), dim3(256), 0, st, (int)CO_DY, g,
                               (const float*)nullptr, Bx, By, sc, W2);
            if ((rc = launch_status("build_coeffs"))) return rc;
            if ((rc = launch_apply(W2, x, Bx, y, By, By, K, dy, st))) return rc;
        }
    }
    if (dh) hipLaunchKernelGGL(causal_grad_h, dim3((Bx * T * J + 255) / 256), dim3(256), 0, st, dh, Bx, By, T, J, sc,
                               g, M, (const float*)nullptr, (const float*)nullptr);
    if (dM) hipLaunchKernelGGL(causal_grad_M, dim3((By * T * J + 255) / 256), dim3(256), 0, st, dM, Bx, By, T, J, sc,
                               g, h, (const float*)nullptr, (const float*)nullptr);
    return launch_status("pairwise_cost_bwd");
}
