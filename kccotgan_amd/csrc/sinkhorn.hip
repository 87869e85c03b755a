// Log-domain Sinkhorn solve and its exact reverse sweep (replaces gan_utils.py:138-165 and
// :87-121 of the reference, and what tf.GradientTape records through that unrolled loop).
//
// One workgroup per n x n problem; the problem never leaves the CU.  The loop is a chain of
// 2*nits dependent half-steps, each one pass over the n^2 cost entries plus a reduction per
// line, so it is latency-bound: no per-iteration launches, no host sync for the stop rule, and
// for n <= 128 the cost matrix lives in registers in BOTH orientations:
//
//   row layout:    thread (i = t / LPR, q = t % LPR) owns C[i][q + LPR*m], m < EPT
//   column layout: thread (j = t / LPR, q = t % LPR) owns C[q + LPR*m][j], m < EPT
//
// LPR (lanes per line) is a power of two <= 64, so the lanes of a line sit in one wavefront and
// the log-sum-exp of a line is EPT serial terms + log2(LPR) xor-shuffle steps; u and v are
// exchanged through 2*n floats of LDS with one barrier per half-step.
//
// Arithmetic follows the reference op for op: M = ((-C + u) + v) / eps;
// lse = log(sum(exp(M - max))) + max; u = eps*(log(1/n) - lse) + u, then v with the new u.
#include "common.h"
#include <math.h>

namespace kccot {

constexpr int SK_MAXN = 128;      // register-resident kernels
constexpr int SK_MAXT = 1024;

struct SinkArgs {
    const float* C;       // [nprob,n,n]
    int n, L, Lmin, stop_mode, lpr;
    float eps, thresh;
    float* u_hist;        // [nprob,L,n] or null
    float* v_hist;
    float* cost_out;      // [nprob]
    int32_t* nits_out;    // [nprob]
    float* pi_out;        // [nprob,n,n] or null
};

__device__ __forceinline__ float seg_max(float v, int lpr) {
    for (int o = lpr >> 1; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
__device__ __forceinline__ float seg_sum(float v, int lpr) {
    for (int o = lpr >> 1; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// One half-step for the calling thread's line: returns the updated dual value of the line.
//   c[m]   : the thread's cost entries of this line
//   self   : current dual of this line (u_i for a row line, v_j for a column line)
//   other  : LDS array of the other dual, indexed by the entry's position q + lpr*m
template <int EPT>
__device__ __forceinline__ float half_step(const float (&c)[EPT], float self, const float* other,
                                           int q, int lpr, int n, float eps, float log_w, bool row_line) {
    float x[EPT];
    float mx = -INFINITY;
#pragma unroll
    for (int m = 0; m < EPT; ++m) {
        const int idx = q + lpr * m;
        const float o = idx < n ? other[idx] : 0.f;
        // gan_utils.py:153,155: (-C + u + v^T)/eps evaluates as ((-C + u) + v)/eps
        const float t = row_line ? ((-c[m] + self) + o) : ((-c[m] + o) + self);
        x[m] = t / eps;
        mx = fmaxf(mx, x[m]);
    }
    mx = seg_max(mx, lpr);
    // tf.reduce_logsumexp: a non-finite max is replaced by 0
    const float shift = (mx > -INFINITY && mx < INFINITY) ? mx : 0.f;
    float s = 0.f;
#pragma unroll
    for (int m = 0; m < EPT; ++m) s += expf(x[m] - shift);
    s = seg_sum(s, lpr);
    const float lse = logf(s) + shift;
    return eps * (log_w - lse) + self;   // gan_utils.py:154,156
}

template <int EPT>
__global__ __launch_bounds__(SK_MAXT) void sinkhorn_fwd_reg(SinkArgs a) {
    __shared__ float u_s[SK_MAXN], v_s[SK_MAXN], red[16];
    const int p = blockIdx.x, n = a.n, lpr = a.lpr;
    const int t = threadIdx.x, line = t / lpr, q = t % lpr;
    const bool active = line < n;
    const float* C = a.C + (int64_t)p * n * n;
    const float eps = a.eps;

    // +inf marks entries beyond the matrix edge: they turn into exp(-inf) = 0 everywhere
    float crow[EPT], ccol[EPT];
#pragma unroll
    for (int m = 0; m < EPT; ++m) {
        const int idx = q + lpr * m;
        const bool ok = active && idx < n;
        crow[m] = ok ? C[(int64_t)line * n + idx] : INFINITY;
        ccol[m] = ok ? C[(int64_t)idx * n + line] : INFINITY;
    }
    if (t < n) { u_s[t] = 0.f; v_s[t] = 0.f; }   // gan_utils.py:147
    __syncthreads();

    const float log_w = logf(1.0f / (float)n);     // log(mu) = log(nu), gan_utils.py:138-139
    int nits = 0;
    for (int it = 0; it < a.L; ++it) {
        float du = 0.f;
        if (active) {
            const float ui = u_s[line];
            const float un = half_step<EPT>(crow, ui, v_s, q, lpr, n, eps, log_w, true);
            if (q == 0) {
                u_s[line] = un;
                du = fabsf(un - ui);
                if (a.u_hist) a.u_hist[((int64_t)p * a.L + it) * n + line] = un;
            }
        }
        __syncthreads();
        if (active) {
            const float vj = v_s[line];
            const float vn = half_step<EPT>(ccol, vj, u_s, q, lpr, n, eps, log_w, false);
            if (q == 0) {
                v_s[line] = vn;
                if (a.v_hist) a.v_hist[((int64_t)p * a.L + it) * n + line] = vn;
            }
        }
        __syncthreads();
        nits = it + 1;
        // gan_utils.py:157-160 (count-based) / :115-117 (index-based).  err is only needed once
        // the stop rule can fire, and never on the last iteration.
        const bool reached = (a.stop_mode == KCCOT_STOP_INDEX) ? (it >= a.Lmin) : (nits >= a.Lmin);
        if (reached && it + 1 < a.L) {
            const float err = block_sum(du, red);
            if (a.thresh > err) break;
        }
    }

    // gan_utils.py:162-164: pi = exp((-C + u + v^T)/eps); cost = sum(pi * C)
    float part = 0.f;
    if (active) {
        const float ui = u_s[line];
#pragma unroll
        for (int m = 0; m < EPT; ++m) {
            const int idx = q + lpr * m;
            if (idx < n) {
                const float pi = expf(((-crow[m] + ui) + v_s[idx]) / eps);
                part += pi * crow[m];
                if (a.pi_out) a.pi_out[(int64_t)p * n * n + (int64_t)line * n + idx] = pi;
            }
        }
    }
    const float cost = block_sum(part, red);
    if (t == 0) {
        a.cost_out[p] = cost;
        a.nits_out[p] = nits;
    }
}

// ------------------------------------------------------------------------------------------
// reverse sweep
//   u_t,i = a - eps*LSE_j((-C_ij + u_{t-1,i} + v_{t-1,j})/eps) + u_{t-1,i},  a = eps*log(1/n)
//   v_t,j = a - eps*LSE_i((-C_ij + u_{t,i}   + v_{t-1,j})/eps) + v_{t-1,j}
// With P_t = softmax_j of the first argument and Q_t = softmax_i of the second:
//   du_t/dv_{t-1} = -P_t,  du_t/dC = +P_t,  du_t/du_{t-1} = 1 - sum_j P_t = 0
//   dv_t/du_t     = -Q_t,  dv_t/dC = +Q_t,  dv_t/dv_{t-1} = 1 - sum_i Q_t = 0
// and, from the update rules themselves, no reduction has to be redone:
//   Q_t[i,j] = exp((-C_ij + u_t,i + v_t,j     - a)/eps)
//   P_t[i,j] = exp((-C_ij + u_t,i + v_{t-1,j} - a)/eps)
// (the two "= 0" terms are 1e-7-sized rounding residues in the reference's tape; dropped).
// ------------------------------------------------------------------------------------------
struct SinkBwdArgs {
    const float* C;
    const float* u_hist;
    const float* v_hist;
    const int32_t* nits;
    const float* gcost;
    float* dC;
    int n, L, lpr;
    float eps;
};

template <int EPT>
__global__ __launch_bounds__(SK_MAXT) void sinkhorn_bwd_reg(SinkBwdArgs a) {
    __shared__ float ut[SK_MAXN], vt[SK_MAXN], vp[SK_MAXN], gu[SK_MAXN], gv[SK_MAXN];
    const int p = blockIdx.x, n = a.n, lpr = a.lpr;
    const int t = threadIdx.x, line = t / lpr, q = t % lpr;
    const bool active = line < n;
    const float* C = a.C + (int64_t)p * n * n;
    const float eps = a.eps, g = a.gcost[p];
    const int nits = a.nits[p];
    const float* uh = a.u_hist + (int64_t)p * a.L * n;
    const float* vh = a.v_hist + (int64_t)p * a.L * n;

    float crow[EPT], ccol[EPT], drow[EPT], dcol[EPT];
#pragma unroll
    for (int m = 0; m < EPT; ++m) {
        const int idx = q + lpr * m;
        const bool ok = active && idx < n;
        crow[m] = ok ? C[(int64_t)line * n + idx] : INFINITY;
        ccol[m] = ok ? C[(int64_t)idx * n + line] : INFINITY;
        drow[m] = 0.f;
        dcol[m] = 0.f;
    }
    if (t < n) {
        ut[t] = nits > 0 ? uh[(int64_t)(nits - 1) * n + t] : 0.f;
        vt[t] = nits > 0 ? vh[(int64_t)(nits - 1) * n + t] : 0.f;
    }
    __syncthreads();

    // cost = sum_ij pi_ij C_ij, pi = exp((-C+u+v)/eps):
    //   dcost/dC_ij (direct) = pi_ij (1 - C_ij/eps); dcost/du_i = sum_j pi_ij C_ij/eps; same for v
    if (active) {
        float su = 0.f, sv = 0.f;
#pragma unroll
        for (int m = 0; m < EPT; ++m) {
            const int idx = q + lpr * m;
            if (idx < n) {
                const float pr = expf(((-crow[m] + ut[line]) + vt[idx]) / eps);
                drow[m] = g * pr * (1.f - crow[m] / eps);
                su += pr * crow[m];
                const float pc = expf(((-ccol[m] + ut[idx]) + vt[line]) / eps);
                sv += pc * ccol[m];
            }
        }
        su = seg_sum(su, lpr);
        sv = seg_sum(sv, lpr);
        if (q == 0) { gu[line] = g * su / eps; gv[line] = g * sv / eps; }
    }
    const float aconst = eps * logf(1.0f / (float)n);

    for (int it = nits; it >= 1; --it) {
        __syncthreads();   // gu/gv of the previous step complete; ut/vt/vp free to be replaced
        if (t < n) {
            ut[t] = uh[(int64_t)(it - 1) * n + t];
            vt[t] = vh[(int64_t)(it - 1) * n + t];
            vp[t] = it >= 2 ? vh[(int64_t)(it - 2) * n + t] : 0.f;
        }
        __syncthreads();
        // (A) through v_t: row pass with Q_t; gu_i -= sum_j Q_ij gv_j ; dC_ij += Q_ij gv_j
        if (active) {
            const float ui = ut[line];
            float s = 0.f;
#pragma unroll
            for (int m = 0; m < EPT; ++m) {
                const int idx = q + lpr * m;
                if (idx < n) {
                    const float qq = expf((((-crow[m] + ui) + vt[idx]) - aconst) / eps);
                    const float w = qq * gv[idx];
                    drow[m] += w;
                    s += w;
                }
            }
            s = seg_sum(s, lpr);
            // grad wrt u_t: the final-cost term on the last iteration, nothing on older ones
            if (q == 0) gu[line] = (it == nits ? gu[line] : 0.f) - s;
        }
        __syncthreads();
        // (B) through u_t: column pass with P_t; gv_{t-1,j} = -sum_i P_ij gu_i ; dC_ij += P_ij gu_i
        if (active) {
            const float vj = vp[line];
            float r = 0.f;
#pragma unroll
            for (int m = 0; m < EPT; ++m) {
                const int idx = q + lpr * m;
                if (idx < n) {
                    const float pp = expf((((-ccol[m] + ut[idx]) + vj) - aconst) / eps);
                    const float w = pp * gu[idx];
                    dcol[m] += w;
                    r += w;
                }
            }
            r = seg_sum(r, lpr);
            if (q == 0) gv[line] = -r;
        }
    }

    // dC = row-layout part + (column-layout part)^T
    float* dC = a.dC + (int64_t)p * n * n;
    if (active) {
#pragma unroll
        for (int m = 0; m < EPT; ++m) {
            const int idx = q + lpr * m;
            if (idx < n) dC[(int64_t)line * n + idx] = drow[m];
        }
    }
    __threadfence_block();
    __syncthreads();
    if (active) {
#pragma unroll
        for (int m = 0; m < EPT; ++m) {
            const int idx = q + lpr * m;
            if (idx < n) dC[(int64_t)idx * n + line] += dcol[m];
        }
    }
}

// ------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------
struct SinkGeom { int lpr, ept, threads; };

static SinkGeom sink_geom(int n) {
    int lpr = 64;
    while (lpr > 1 && (int64_t)n * lpr > SK_MAXT) lpr >>= 1;
    SinkGeom g;
    g.lpr = lpr;
    const int need = (n + lpr - 1) / lpr;
    g.ept = need <= 1 ? 1 : need <= 2 ? 2 : need <= 4 ? 4 : need <= 8 ? 8 : 16;
    g.threads = (n * lpr + 63) / 64 * 64;
    return g;
}

}  // namespace kccot

using namespace kccot;

extern "C" size_t kccot_sinkhorn_workspace_bytes(int nprob, int n) {
    (void)nprob; (void)n;
    return 0;   // the register-resident kernels need none; kept for the large-n path
}

#define KCCOT_SK_DISPATCH(KERNEL, ARGS, GEOM, NPROB, ST)                                                  \
    switch ((GEOM).ept) {                                                                                 \
        case 1: hipLaunchKernelGGL(KERNEL<1>, dim3(NPROB), dim3((GEOM).threads), 0, ST, ARGS); break;     \
        case 2: hipLaunchKernelGGL(KERNEL<2>, dim3(NPROB), dim3((GEOM).threads), 0, ST, ARGS); break;     \
        case 4: hipLaunchKernelGGL(KERNEL<4>, dim3(NPROB), dim3((GEOM).threads), 0, ST, ARGS); break;     \
        case 8: hipLaunchKernelGGL(KERNEL<8>, dim3(NPROB), dim3((GEOM).threads), 0, ST, ARGS); break;     \
        default: hipLaunchKernelGGL(KERNEL<16>, dim3(NPROB), dim3((GEOM).threads), 0, ST, ARGS); break;   \
    }

extern "C" int kccot_sinkhorn_fwd_f32(const float* C, int nprob, int n, float eps, int L, int Lmin,
                                      float thresh, int stop_mode, float* u_hist, float* v_hist,
                                      float* cost_out, int32_t* nits_out, float* pi_out, void* ws,
                                      size_t ws_bytes, kccot_stream_t stream) {
    (void)ws; (void)ws_bytes;
    if (!C || !cost_out || !nits_out) return fail(KCCOT_EINVAL, "sinkhorn_fwd: null pointer");
    if (nprob <= 0 || n <= 0 || L < 0 || !(eps > 0.f))
        return fail(KCCOT_EINVAL, "sinkhorn_fwd: bad arguments nprob=%d n=%d L=%d eps=%g", nprob, n, L, (double)eps);
    if ((u_hist == nullptr) != (v_hist == nullptr))
        return fail(KCCOT_EINVAL, "sinkhorn_fwd: u_hist and v_hist must be given together");
    if (stop_mode != KCCOT_STOP_COUNT && stop_mode != KCCOT_STOP_INDEX)
        return fail(KCCOT_EINVAL, "sinkhorn_fwd: bad stop_mode %d", stop_mode);
    if (n > SK_MAXN)
        return fail(KCCOT_EUNSUPPORTED, "sinkhorn_fwd: n=%d > %d (the multi-CU solver for larger "
                    "batches is not built yet)", n, SK_MAXN);
    SinkGeom g = sink_geom(n);
    SinkArgs a{C, n, L, Lmin, stop_mode, g.lpr, eps, thresh, u_hist, v_hist, cost_out, nits_out, pi_out};
    hipStream_t st = (hipStream_t)stream;
    KCCOT_SK_DISPATCH(sinkhorn_fwd_reg, a, g, nprob, st)
    return launch_status("sinkhorn_fwd_reg");
}

extern "C" int kccot_sinkhorn_bwd_f32(const float* C, const float* u_hist, const float* v_hist,
                                      const int32_t* nits, int nprob, int n, float eps, int L,
                                      const float* gcost, float* dC_out, void* ws, size_t ws_bytes,
                                      kccot_stream_t stream) {
    (void)ws; (void)ws_bytes;
    if (!C || !u_hist || !v_hist || !nits || !gcost || !dC_out)
        return fail(KCCOT_EINVAL, "sinkhorn_bwd: null pointer");
    if (nprob <= 0 || n <= 0 || L < 0 || !(eps > 0.f))
        return fail(KCCOT_EINVAL, "sinkhorn_bwd: bad arguments nprob=%d n=%d L=%d eps=%g", nprob, n, L, (double)eps);
    if (n > SK_MAXN)
        return fail(KCCOT_EUNSUPPORTED, "sinkhorn_bwd: n=%d > %d", n, SK_MAXN);
    SinkGeom g = sink_geom(n);
    SinkBwdArgs a{C, u_hist, v_hist, nits, gcost, dC_out, n, L, g.lpr, eps};
    hipStream_t st = (hipStream_t)stream;
    KCCOT_SK_DISPATCH(sinkhorn_bwd_reg, a, g, nprob, st)
    return launch_status("sinkhorn_bwd_reg");
}
