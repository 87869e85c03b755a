// Sinkhorn forward / reverse sweep for problems too large for the register-resident kernels
// (128 < n <= 1024: BASELINE configs 3-5 at B = 256, 512).  Same arithmetic and the same
// one-workgroup-per-problem structure as sinkhorn.hip, but the cost matrix stays in L2/HBM and is
// streamed once per half-step: a wave owns whole lines (rows for the u-update; for the v-update
// rows of the TRANSPOSED copy C^T built once in the workspace, so both passes read coalesced
// float4 streams), reduces them with DPP + two cross-row shuffles, and the duals live in LDS.
// One workgroup pulls ~64 B/clk from L2, so a half-step costs n^2*4/64 cycles (7 us at n = 512):
// the multi-CU cooperative solver that would split a problem over an XCD is not built yet.
#include "common.h"
#include <math.h>
#include <stdint.h>
#include <stdlib.h>

namespace kccot {

constexpr int SG_MAXN = 1024;
constexpr int SG_THREADS = 1024;
constexpr float SG_LOG2E = 1.4426950408889634f;
constexpr float SG_LN2 = 0.6931471805599453f;

__global__ __launch_bounds__(256) void transpose_batched(const float* __restrict__ in, float* __restrict__ out, int n) {
    __shared__ float tile[32][33];
    const int p = blockIdx.z;
    const float* src = in + (int64_t)p * n * n;
    float* dst = out + (int64_t)p * n * n;
    const int x0 = blockIdx.x * 32, y0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 32 x 8
    for (int r = ty; r < 32; r += 8)
        if (y0 + r < n && x0 + tx < n) tile[r][tx] = src[(int64_t)(y0 + r) * n + x0 + tx];
    __syncthreads();
    for (int r = ty; r < 32; r += 8)
        if (x0 + r < n && y0 + tx < n) dst[(int64_t)(x0 + r) * n + y0 + tx] = tile[tx][r];
}

// out += in^T   (combining the two halves of dC after the reverse sweep)
__global__ __launch_bounds__(256) void add_transposed_batched(const float* __restrict__ in, float* __restrict__ out, int n) {
    __shared__ float tile[32][33];
    const int p = blockIdx.z;
    const float* src = in + (int64_t)p * n * n;
    float* dst = out + (int64_t)p * n * n;
    const int x0 = blockIdx.x * 32, y0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int r = ty; r < 32; r += 8)
        if (y0 + r < n && x0 + tx < n) tile[r][tx] = src[(int64_t)(y0 + r) * n + x0 + tx];
    __syncthreads();
    for (int r = ty; r < 32; r += 8)
        if (x0 + r < n && y0 + tx < n) dst[(int64_t)(x0 + r) * n + y0 + tx] += tile[tx][r];
}

// LSE update of one line held by a wave: line[j], j < n; other[j] from LDS.
//   ROW:  x = ((-c + self) + other[j]) * inv_eps      (self = u_i, other = v)
//   !ROW: x = ((-c + other[j]) + self) * inv_eps      (self = v_j, other = u)  -- gan_utils.py:153-156
template <bool ROW>
__device__ __forceinline__ float line_update(const float* __restrict__ line, const float* other, int n, float self,
                                             float eps, float inv_eps, float log_w) {
    const int lane = threadIdx.x & 63;
    float mx = -INFINITY;
    for (int j = lane; j < n; j += 64) {
        const float c = line[j], o = other[j];
        const float t = ROW ? ((-c + self) + o) : ((-c + o) + self);
        mx = fmaxf(mx, t * inv_eps);
    }
    mx = wave_max(mx);
    const float shift = (mx > -INFINITY && mx < INFINITY) ? mx : 0.f;
    float s = 0.f;
    for (int j = lane; j < n; j += 64) {
        const float c = line[j], o = other[j];
        const float t = ROW ? ((-c + self) + o) : ((-c + o) + self);
        s += __builtin_amdgcn_exp2f((t * inv_eps - shift) * SG_LOG2E);
    }
    s = wave_sum(s);
    const float lse = __builtin_amdgcn_logf(s) * SG_LN2 + shift;
    return eps * (log_w - lse) + self;
}

struct SinkGenArgs {
    const float* C;      // [nprob,n,n]
    const float* CT;     // [nprob,n,n] transposed (workspace)
    int n, L, Lmin, stop_mode;
    float eps, inv_eps, thresh;
    float* u_hist;
    float* v_hist;
    float* cost_out;
    int32_t* nits_out;
    float* pi_out;
};

__global__ __launch_bounds__(SG_THREADS) void sinkhorn_fwd_gen(SinkGenArgs a) {
    __shared__ float u_s[SG_MAXN], v_s[SG_MAXN], red[16];
    const int p = blockIdx.x, n = a.n;
    const int t = threadIdx.x, lane = t & 63, wid = t >> 6, nw = SG_THREADS / 64;
    const float* C = a.C + (int64_t)p * n * n;
    const float* CT = a.CT + (int64_t)p * n * n;
    const float eps = a.eps, inv_eps = a.inv_eps;
    for (int i = t; i < n; i += SG_THREADS) { u_s[i] = 0.f; v_s[i] = 0.f; }
    __syncthreads();
    const float log_w = logf(1.0f / (float)n);
    int nits = 0;
    for (int it = 0; it < a.L; ++it) {
        float du = 0.f;
        for (int i = wid; i < n; i += nw) {       // u-update: rows of C
            const float ui = u_s[i];
            const float un = line_update<true>(C + (int64_t)i * n, v_s, n, ui, eps, inv_eps, log_w);
            if (lane == 0) {
                du += fabsf(un - ui);
                // written after the barrier below would need a second array; u_s[i] is read by this
                // wave only during the row pass, so the in-place store is safe
                u_s[i] = un;
                if (a.u_hist) a.u_hist[((int64_t)p * a.L + it) * n + i] = un;
            }
        }
        __syncthreads();
        for (int j = wid; j < n; j += nw) {       // v-update: rows of C^T, with the new u
            const float vj = v_s[j];
            const float vn = line_update<false>(CT + (int64_t)j * n, u_s, n, vj, eps, inv_eps, log_w);
            if (lane == 0) {
                v_s[j] = vn;
                if (a.v_hist) a.v_hist[((int64_t)p * a.L + it) * n + j] = vn;
            }
        }
        __syncthreads();
        nits = it + 1;
        const bool reached = (a.stop_mode == KCCOT_STOP_INDEX) ? (it >= a.Lmin) : (nits >= a.Lmin);
        if (reached && it + 1 < a.L) {
            const float err = block_sum(du, red);
            if (a.thresh > err) break;
        }
    }
    float part = 0.f;
    for (int i = wid; i < n; i += nw) {
        const float ui = u_s[i];
        const float* row = C + (int64_t)i * n;
        for (int j = lane; j < n; j += 64) {
            const float c = row[j];
            const float pi = __builtin_amdgcn_exp2f(((-c + ui) + v_s[j]) * inv_eps * SG_LOG2E);
            part += pi * c;
            if (a.pi_out) a.pi_out[(int64_t)p * n * n + (int64_t)i * n + j] = pi;
        }
    }
    const float cost = block_sum(part, red);
    if (t == 0) { a.cost_out[p] = cost; a.nits_out[p] = nits; a.nits_out[gridDim.x + p] = nits; }
}

// ---- forward, wide form (n % 4 == 0) --------------------------------------------------------------
// Same arithmetic as sinkhorn_fwd_gen, restructured so that the stream from L2 is not a latency chain:
// a wave takes RB lines at a time, every lane loads its 16-byte pieces of all of them (NV pieces per
// line, n <= 256 NV) before the first use, the next group's loads are issued before the current group
// is reduced, and each element is read once per half-step (the two-pass log-sum-exp runs on registers).
// The other side's duals are read from LDS once per half-step per wave.
template <int NV>
__device__ __forceinline__ void gen_load_lines(float4 (&c)[8 / NV][NV], const float* __restrict__ M, int n, int first, int stride) {
    constexpr int RB = 8 / NV;
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int r = 0; r < RB; ++r) {
        const int line = first + r * stride;
        const int lc = line < n ? line : n - 1;                 // clamped: no predicated loads
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            const int idx = 4 * lane + 256 * v;
            const int ic = idx < n ? idx : n - 4;
            c[r][v] = *reinterpret_cast<const float4*>(M + (int64_t)lc * n + ic);
        }
    }
}

template <int NV, bool ROW>
__device__ __forceinline__ float gen_update_line(const float4 (&c)[NV], const float4 (&o)[NV], int n, float self, float eps,
                                                 float inv_eps, float log_w) {
    const int lane = threadIdx.x & 63;
    float tt[NV][4];
    float mx = -INFINITY;
#pragma unroll
    for (int v = 0; v < NV; ++v) {
        const float cc[4] = {c[v].x, c[v].y, c[v].z, c[v].w}, oo[4] = {o[v].x, o[v].y, o[v].z, o[v].w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const bool ok = 4 * lane + 256 * v + e < n;
            const float t = ROW ? ((-cc[e] + self) + oo[e]) : ((-cc[e] + oo[e]) + self);
            tt[v][e] = ok ? t * inv_eps : -INFINITY;
            mx = fmaxf(mx, tt[v][e]);
        }
    }
    mx = wave_max_fast(mx);
    const float shift = (mx > -INFINITY && mx < INFINITY) ? mx : 0.f;
    float s = 0.f;
#pragma unroll
    for (int v = 0; v < NV; ++v)
#pragma unroll
        for (int e = 0; e < 4; ++e) s += __builtin_amdgcn_exp2f((tt[v][e] - shift) * SG_LOG2E);
    s = wave_sum_fast(s);
    const float lse = __builtin_amdgcn_logf(s) * SG_LN2 + shift;
    return eps * (log_w - lse) + self;
}

template <int NV>
__global__ __launch_bounds__(SG_THREADS) void sinkhorn_fwd_gen4(SinkGenArgs a) {
    constexpr int RB = 8 / NV;
    __shared__ __attribute__((aligned(16))) float u_s[SG_MAXN], v_s[SG_MAXN];
    __shared__ float red[16];
    const int p = blockIdx.x, n = a.n;
    const int t = threadIdx.x, lane = t & 63, wid = t >> 6, nw = SG_THREADS / 64;
    const float* C = a.C + (int64_t)p * n * n;
    const float* CT = a.CT + (int64_t)p * n * n;
    const float eps = a.eps, inv_eps = a.inv_eps;
    for (int i = t; i < SG_MAXN; i += SG_THREADS) { u_s[i] = 0.f; v_s[i] = 0.f; }
    __syncthreads();
    const float log_w = logf(1.0f / (float)n);
    // one half-step: lines of M (rows of C, or rows of C^T), `self_s` updated in place, `other_s` read-only
    auto half = [&](const float* M, float* self_s, const float* other_s, float* hist, bool row, float& du) {
        float4 o[NV];
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            const int idx = 4 * lane + 256 * v;
            o[v] = idx < n ? *reinterpret_cast<const float4*>(other_s + idx) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
        float4 cur[RB][NV], nxt[RB][NV];
        gen_load_lines<NV>(cur, M, n, wid, nw);
        for (int base = wid; base < n; base += nw * RB) {
            if (base + nw * RB < n) gen_load_lines<NV>(nxt, M, n, base + nw * RB, nw);
#pragma unroll
            for (int r = 0; r < RB; ++r) {
                const int line = base + r * nw;
                const float sv = self_s[line < n ? line : 0];
                const float nv = row ? gen_update_line<NV, true>(cur[r], o, n, sv, eps, inv_eps, log_w)
                                     : gen_update_line<NV, false>(cur[r], o, n, sv, eps, inv_eps, log_w);
                if (line < n && lane == 0) {
                    du += fabsf(nv - sv);
                    self_s[line] = nv;          // read by this wave only during the pass
                    if (hist) hist[line] = nv;
                }
            }
#pragma unroll
            for (int r = 0; r < RB; ++r)
#pragma unroll
                for (int v = 0; v < NV; ++v) cur[r][v] = nxt[r][v];
        }
    };
    int nits = 0;
    for (int it = 0; it < a.L; ++it) {
        float du = 0.f, dv = 0.f;
        half(C, u_s, v_s, a.u_hist ? a.u_hist + ((int64_t)p * a.L + it) * n : nullptr, true, du);
        __syncthreads();
        half(CT, v_s, u_s, a.v_hist ? a.v_hist + ((int64_t)p * a.L + it) * n : nullptr, false, dv);
        __syncthreads();
        nits = it + 1;
        const bool reached = (a.stop_mode == KCCOT_STOP_INDEX) ? (it >= a.Lmin) : (nits >= a.Lmin);
        if (reached && it + 1 < a.L) {
            const float err = block_sum(du, red);
            if (a.thresh > err) break;
        }
    }
    float part = 0.f;
    for (int i = wid; i < n; i += nw) {
        const float ui = u_s[i];
        const float* row = C + (int64_t)i * n;
        for (int j = lane; j < n; j += 64) {
            const float c = row[j];
            const float pi = __builtin_amdgcn_exp2f(((-c + ui) + v_s[j]) * inv_eps * SG_LOG2E);
            part += pi * c;
            if (a.pi_out) a.pi_out[(int64_t)p * n * n + (int64_t)i * n + j] = pi;
        }
    }
    const float cost = block_sum(part, red);
    if (t == 0) { a.cost_out[p] = cost; a.nits_out[p] = nits; a.nits_out[gridDim.x + p] = nits; }
}

// ---- forward, 16 lanes per line (n % 4 == 0, n <= 512) -----------------------------------------------
// At n = 256 a line is 4 elements per lane of a 64-lane wave: the 64-wide reductions and the per-line
// bookkeeping, not the elements, dominate the instruction stream, and the loop is issue-bound on its one
// CU.  Here a wave instruction serves FOUR lines (16 lanes each, NV 16-byte pieces per lane per line,
// n <= 64 NV), the reductions are the in-row DPP trees of the register kernels, and the lines of the
// next group are loaded before the current group is reduced.
template <int NV, bool ROW>
__device__ __forceinline__ float gen16_update(const float4 (&c)[NV], const float4 (&o)[NV], int n, int q, float self,
                                              float eps, float inv_eps, float log_w) {
    float tt[NV][4];
    float mx = -INFINITY;
#pragma unroll
    for (int v = 0; v < NV; ++v) {
        const float cc[4] = {c[v].x, c[v].y, c[v].z, c[v].w}, oo[4] = {o[v].x, o[v].y, o[v].z, o[v].w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const bool ok = 4 * q + 64 * v + e < n;
            const float t = ROW ? ((-cc[e] + self) + oo[e]) : ((-cc[e] + oo[e]) + self);
            tt[v][e] = ok ? t * inv_eps : -INFINITY;
            mx = fmaxf(mx, tt[v][e]);
        }
    }
    mx = seg_max<16>(mx);
    const float shift = (mx > -INFINITY && mx < INFINITY) ? mx : 0.f;
    float s = 0.f;
#pragma unroll
    for (int v = 0; v < NV; ++v)
#pragma unroll
        for (int e = 0; e < 4; ++e) s += __builtin_amdgcn_exp2f((tt[v][e] - shift) * SG_LOG2E);
    s = seg_sum<16>(s);
    const float lse = __builtin_amdgcn_logf(s) * SG_LN2 + shift;
    return eps * (log_w - lse) + self;
}

template <int NV>
__global__ __launch_bounds__(SG_THREADS) void sinkhorn_fwd_gen16(SinkGenArgs a) {
    __shared__ __attribute__((aligned(16))) float u_s[SG_MAXN], v_s[SG_MAXN];
    __shared__ float red[16];
    const int p = blockIdx.x, n = a.n;
    const int t = threadIdx.x, lane = t & 63, wid = t >> 6, nw = SG_THREADS / 64;
    const int q = lane & 15, sub = lane >> 4;
    const float* C = a.C + (int64_t)p * n * n;
    const float* CT = a.CT + (int64_t)p * n * n;
    const float eps = a.eps, inv_eps = a.inv_eps;
    for (int i = t; i < SG_MAXN; i += SG_THREADS) { u_s[i] = 0.f; v_s[i] = 0.f; }
    __syncthreads();
    const float log_w = logf(1.0f / (float)n);
    const int ngroups = (n + 3) >> 2;
    auto load_group = [&](float4 (&c)[NV], const float* M, int g) {
        const int line = 4 * g + sub;
        const int lc = line < n ? line : n - 1;                 // clamped: no predicated loads
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            const int idx = 4 * q + 64 * v;
            c[v] = *reinterpret_cast<const float4*>(M + (int64_t)lc * n + (idx < n ? idx : n - 4));
        }
    };
    auto half = [&](const float* M, float* self_s, const float* other_s, float* hist, bool row, float& du) {
        float4 o[NV];
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            const int idx = 4 * q + 64 * v;
            o[v] = idx < n ? *reinterpret_cast<const float4*>(other_s + idx) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
        float4 cur[NV], nxt[NV];
        if (wid < ngroups) load_group(cur, M, wid);
        for (int g = wid; g < ngroups; g += nw) {
            if (g + nw < ngroups) load_group(nxt, M, g + nw);
            const int line = 4 * g + sub;
            const float sv = self_s[line < n ? line : 0];
            const float nv = row ? gen16_update<NV, true>(cur, o, n, q, sv, eps, inv_eps, log_w)
                                 : gen16_update<NV, false>(cur, o, n, q, sv, eps, inv_eps, log_w);
            if (line < n && q == 0) {
                du += fabsf(nv - sv);
                self_s[line] = nv;              // read by this wave only during the pass
                if (hist) hist[line] = nv;
            }
#pragma unroll
            for (int v = 0; v < NV; ++v) cur[v] = nxt[v];
        }
    };
    int nits = 0;
    for (int it = 0; it < a.L; ++it) {
        float du = 0.f, dv = 0.f;
        half(C, u_s, v_s, a.u_hist ? a.u_hist + ((int64_t)p * a.L + it) * n : nullptr, true, du);
        __syncthreads();
        half(CT, v_s, u_s, a.v_hist ? a.v_hist + ((int64_t)p * a.L + it) * n : nullptr, false, dv);
        __syncthreads();
        nits = it + 1;
        const bool reached = (a.stop_mode == KCCOT_STOP_INDEX) ? (it >= a.Lmin) : (nits >= a.Lmin);
        if (reached && it + 1 < a.L) {
            const float err = block_sum(du, red);
            if (a.thresh > err) break;
        }
    }
    float part = 0.f;
    for (int i = wid; i < n; i += nw) {
        const float ui = u_s[i];
        const float* row = C + (int64_t)i * n;
        for (int j = lane; j < n; j += 64) {
            const float c = row[j];
            const float pi = __builtin_amdgcn_exp2f(((-c + ui) + v_s[j]) * inv_eps * SG_LOG2E);
            part += pi * c;
            if (a.pi_out) a.pi_out[(int64_t)p * n * n + (int64_t)i * n + j] = pi;
        }
    }
    const float cost = block_sum(part, red);
    if (t == 0) { a.cost_out[p] = cost; a.nits_out[p] = nits; a.nits_out[gridDim.x + p] = nits; }
}

struct SinkGenBwdArgs {
    const float* C;
    const float* CT;
    const float* u_hist;
    const float* v_hist;
    const int32_t* nits;
    const float* gcost;
    float* dC;           // [nprob,n,n]: row-layout accumulator, final result
    float* dCT;          // [nprob,n,n]: accumulator of the column passes, in transposed layout
    int n, L;
    float eps, inv_eps;
};

// See sinkhorn.hip for the derivation.  Accumulators live in global memory (L2): the row pass
// owns dC[i][:] of its rows, the column pass owns dCT[j][:] of its columns, so every
// read-modify-write is private to one wave and coalesced.
__global__ __launch_bounds__(SG_THREADS) void sinkhorn_bwd_gen(SinkGenBwdArgs a) {
    __shared__ float ut[SG_MAXN], vt[SG_MAXN], vp[SG_MAXN], gu[SG_MAXN], gv[SG_MAXN];
    const int p = blockIdx.x, n = a.n;
    const int t = threadIdx.x, lane = t & 63, wid = t >> 6, nw = SG_THREADS / 64;
    const float* C = a.C + (int64_t)p * n * n;
    const float* CT = a.CT + (int64_t)p * n * n;
    float* dC = a.dC + (int64_t)p * n * n;
    float* dCT = a.dCT + (int64_t)p * n * n;
    const float eps = a.eps, inv_eps = a.inv_eps, g = a.gcost[p];
    const int nits = a.nits[p];
    const float* uh = a.u_hist + (int64_t)p * a.L * n;
    const float* vh = a.v_hist + (int64_t)p * a.L * n;
    for (int i = t; i < n; i += SG_THREADS) {
        ut[i] = nits > 0 ? uh[(int64_t)(nits - 1) * n + i] : 0.f;
        vt[i] = nits > 0 ? vh[(int64_t)(nits - 1) * n + i] : 0.f;
    }
    __syncthreads();
    // final cost term: dC = g*pi*(1 - C/eps); gu = g*sum_j pi C/eps; gv likewise; dCT = 0
    for (int i = wid; i < n; i += nw) {
        const float ui = ut[i];
        float su = 0.f;
        for (int j = lane; j < n; j += 64) {
            const float c = C[(int64_t)i * n + j];
            const float pr = __builtin_amdgcn_exp2f(((-c + ui) + vt[j]) * inv_eps * SG_LOG2E);
            dC[(int64_t)i * n + j] = g * pr * (1.f - c * inv_eps);
            su += pr * c;
        }
        su = wave_sum(su);
        if (lane == 0) gu[i] = g * su * inv_eps;
    }
    for (int j = wid; j < n; j += nw) {
        const float vj = vt[j];
        float sv = 0.f;
        for (int i = lane; i < n; i += 64) {
            const float c = CT[(int64_t)j * n + i];
            const float pc = __builtin_amdgcn_exp2f(((-c + ut[i]) + vj) * inv_eps * SG_LOG2E);
            dCT[(int64_t)j * n + i] = 0.f;
            sv += pc * c;
        }
        sv = wave_sum(sv);
        if (lane == 0) gv[j] = g * sv * inv_eps;
    }
    const float aconst = eps * logf(1.0f / (float)n);
    for (int it = nits; it >= 1; --it) {
        __syncthreads();
        for (int i = t; i < n; i += SG_THREADS) {
            ut[i] = uh[(int64_t)(it - 1) * n + i];
            vt[i] = vh[(int64_t)(it - 1) * n + i];
            vp[i] = it >= 2 ? vh[(int64_t)(it - 2) * n + i] : 0.f;
        }
        __syncthreads();
        // (A) row pass with Q_t: gu_i = [it==nits] gu_i - sum_j Q_ij gv_j ; dC_ij += Q_ij gv_j
        for (int i = wid; i < n; i += nw) {
            const float ui = ut[i];
            float s = 0.f;
            for (int j = lane; j < n; j += 64) {
                const float c = C[(int64_t)i * n + j];
                const float qq = __builtin_amdgcn_exp2f((((-c + ui) + vt[j]) - aconst) * inv_eps * SG_LOG2E);
                const float w = qq * gv[j];
                dC[(int64_t)i * n + j] += w;
                s += w;
            }
            s = wave_sum(s);
            if (lane == 0) gu[i] = (it == nits ? gu[i] : 0.f) - s;
        }
        __syncthreads();
        // (B) column pass with P_t: gv_j = -sum_i P_ij gu_i ; dC_ij += P_ij gu_i (kept transposed)
        for (int j = wid; j < n; j += nw) {
            const float vj = vp[j];
            float r = 0.f;
            for (int i = lane; i < n; i += 64) {
                const float c = CT[(int64_t)j * n + i];
                const float pp = __builtin_amdgcn_exp2f((((-c + ut[i]) + vj) - aconst) * inv_eps * SG_LOG2E);
                const float w = pp * gu[i];
                dCT[(int64_t)j * n + i] += w;
                r += w;
            }
            r = wave_sum(r);
            if (lane == 0) gv[j] = -r;
        }
    }
}

size_t sinkhorn_gen_workspace_bytes(int nprob, int n) {
    return 2 * align_up((size_t)nprob * n * n * sizeof(float), 256);
}

// sinkhorn_coop.hip
constexpr int SK_COOP_MAX_L = 500000;
bool sinkhorn_coop_eligible(int nprob, int n);
int launch_sinkhorn_fwd_coop(const float* C, int nprob, int n, float eps, int L, int Lmin, float thresh, int stop_mode,
                             float* u_hist, float* v_hist, float* cost_out, int32_t* nits_out, float* pi_out, void* ws,
                             hipStream_t st);
int launch_sinkhorn_bwd_coop(const float* C, const float* u_hist, const float* v_hist, const int32_t* nits, int nprob, int n,
                             float eps, int L, const float* gcost, float* dC, void* ws, hipStream_t st);

int launch_sinkhorn_fwd_gen(const float* C, int nprob, int n, float eps, int L, int Lmin, float thresh, int stop_mode,
                            float* u_hist, float* v_hist, float* cost_out, int32_t* nits_out, float* pi_out, void* ws,
                            size_t ws_bytes, hipStream_t st) {
    if (n > SG_MAXN) return fail(KCCOT_EUNSUPPORTED, "sinkhorn_fwd: n=%d > %d", n, SG_MAXN);
    const size_t need = sinkhorn_gen_workspace_bytes(nprob, n);
    if (!ws || ws_bytes < need) return fail(KCCOT_EWORKSPACE, "sinkhorn_fwd: workspace %zu < required %zu", ws_bytes, need);
    if (sinkhorn_coop_eligible(nprob, n) && L < SK_COOP_MAX_L)     // the exchange tags hold 2 L + 2 half-steps in 20 bits
        return launch_sinkhorn_fwd_coop(C, nprob, n, eps, L, Lmin, thresh, stop_mode, u_hist, v_hist, cost_out, nits_out, pi_out, ws, st);
    float* CT = static_cast<float*>(ws);
    dim3 tg((n + 31) / 32, (n + 31) / 32, nprob);
    hipLaunchKernelGGL(transpose_batched, tg, dim3(256), 0, st, C, CT, n);
    int rc = launch_status("transpose_batched");
    if (rc) return rc;
    SinkGenArgs a{C, CT, n, L, Lmin, stop_mode, eps, (float)(1.0 / (double)eps), thresh, u_hist, v_hist, cost_out, nits_out, pi_out};
    const bool wide = (n % 4 == 0) && ((uintptr_t)C % 16 == 0);
    if (!wide) hipLaunchKernelGGL(sinkhorn_fwd_gen, dim3(nprob), dim3(SG_THREADS), 0, st, a);
    else if (n <= 256) hipLaunchKernelGGL(sinkhorn_fwd_gen16<4>, dim3(nprob), dim3(SG_THREADS), 0, st, a);
    else if (n <= 512) hipLaunchKernelGGL(sinkhorn_fwd_gen16<8>, dim3(nprob), dim3(SG_THREADS), 0, st, a);
    else hipLaunchKernelGGL(sinkhorn_fwd_gen4<4>, dim3(nprob), dim3(SG_THREADS), 0, st, a);
    return launch_status("sinkhorn_fwd_gen");
}

int launch_sinkhorn_bwd_gen(const float* C, const float* u_hist, const float* v_hist, const int32_t* nits, int nprob, int n,
                            float eps, int L, const float* gcost, float* dC, void* ws, size_t ws_bytes, hipStream_t st) {
    if (n > SG_MAXN) return fail(KCCOT_EUNSUPPORTED, "sinkhorn_bwd: n=%d > %d", n, SG_MAXN);
    const size_t need = sinkhorn_gen_workspace_bytes(nprob, n);
    if (!ws || ws_bytes < need) return fail(KCCOT_EWORKSPACE, "sinkhorn_bwd: workspace %zu < required %zu", ws_bytes, need);
    if (sinkhorn_coop_eligible(nprob, n) && L < SK_COOP_MAX_L)     // the exchange tags hold 2 L + 2 half-steps in 20 bits
        return launch_sinkhorn_bwd_coop(C, u_hist, v_hist, nits, nprob, n, eps, L, gcost, dC, ws, st);
    float* CT = static_cast<float*>(ws);
    float* dCT = reinterpret_cast<float*>(static_cast<char*>(ws) + need / 2);
    dim3 tg((n + 31) / 32, (n + 31) / 32, nprob);
    hipLaunchKernelGGL(transpose_batched, tg, dim3(256), 0, st, C, CT, n);
    int rc = launch_status("transpose_batched");
    if (rc) return rc;
    SinkGenBwdArgs a{C, CT, u_hist, v_hist, nits, gcost, dC, dCT, n, L, eps, (float)(1.0 / (double)eps)};
    hipLaunchKernelGGL(sinkhorn_bwd_gen, dim3(nprob), dim3(SG_THREADS), 0, st, a);
    if ((rc = launch_status("sinkhorn_bwd_gen"))) return rc;
    hipLaunchKernelGGL(add_transposed_batched, tg, dim3(256), 0, st, (const float*)dCT, dC, n);
    return launch_status("add_transposed_batched");
}

}  // namespace kccot
