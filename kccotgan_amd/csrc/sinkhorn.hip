// Log-domain Sinkhorn solve and its exact reverse sweep (replaces gan_utils.py:138-165 and
// :87-121 of the reference, and what tf.GradientTape records through that unrolled loop).
//
// One workgroup per n x n problem; the problem never leaves the CU.  The loop is a chain of
// 2*nits dependent half-steps, each one pass over the n^2 cost entries plus one reduction per
// line, so it is latency-bound: no per-iteration launches, no host sync for the stop rule, and
// for n <= 128 the cost matrix lives in registers in BOTH orientations:
//
//   row layout:    thread (i = t / LPR, q = t % LPR) owns C[i][q*EPT + m], m < EPT
//   column layout: thread (j = t / LPR, q = t % LPR) owns C[q*EPT + m][j], m < EPT
//
// LPR (lanes per line) is 8 or 16, i.e. at most one 16-lane DPP row: the log-sum-exp of a line
// is EPT serial terms + log2(LPR) DPP steps (quad_perm, row_half_mirror, row_mirror -- register
// to register, no LDS crossbar); the duals u and v are exchanged through 2*n floats of LDS
// (ds_read_b128: a thread's EPT entries are contiguous) with ONE barrier per half-step.
//
// The iteration is the reference's (max-shifted LSE over ((-C + u) + v)/eps, u then v with the new
// u), carried in log2 units so that exp/log are the hardware v_exp_f32 / v_log_f32 (see half_step).
#include "common.h"
#include <math.h>
#include "options.h"

// No floating-point contraction in this file: the solver kernels exist in several forms (forward / sweep / fused,
// different lanes per line) that are tested for BIT-identical results, and a mul + add that the compiler fuses into
// an fma in one form but not in another (it did, in the far regime, once the sweep's exp2 were hoisted) breaks that.
// The reference's TensorFlow ops do not fuse either.
#pragma clang fp contract(off)

namespace kccot {

constexpr int SK_MAXN = 128;      // register-resident kernels
constexpr int SK_MAXT = 1024;
constexpr float LOG2E = 1.4426950408889634f;
constexpr float LN2 = 0.6931471805599453f;

__device__ __forceinline__ float fast_exp(float x) { return __builtin_amdgcn_exp2f(x * LOG2E); }
__device__ __forceinline__ float fast_log(float x) { return __builtin_amdgcn_logf(x) * LN2; }

// Diagnostic build only (-DKCCOT_DIAG, libkccot_diag.so, tools/diag_sinkhorn.py): in-kernel
// s_memtime stamps of one half-step.  The product library contains no stamp.
#ifdef KCCOT_DIAG
#ifndef KCCOT_DIAG_IT
#define KCCOT_DIAG_IT 50
#endif
#define KCCOT_STAMP(SLOT)                                                                         \
    do {                                                                                          \
        if (a.diag && it == KCCOT_DIAG_IT) {                                                                 \
            unsigned long long tt_;                                                               \
            __builtin_amdgcn_sched_barrier(0);                                                    \
            asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tt_)::"memory");          \
            __builtin_amdgcn_sched_barrier(0);                                                    \
            if ((threadIdx.x & 63) == 0) a.diag[(blockIdx.x * 16 + (threadIdx.x >> 6)) * 16 + (SLOT)] = tt_; \
        }                                                                                         \
    } while (0)
#else
#define KCCOT_STAMP(SLOT) do {} while (0)
#endif

struct SinkArgs {
    const float* C;       // [nprob,n,n]
    int n, L, Lmin, stop_mode;
    float eps, inv_eps, thresh;
    float* u_hist;        // [nprob,L,n] or null
    float* v_hist;
    float* cost_out;      // [nprob]
    int32_t* nits_out;    // [nprob]
    float* pi_out;        // [nprob,n,n] or null
    unsigned long long* diag;   // diagnostic build only; null otherwise
    float* loss_out;      // mixed divergence 2*cost[0]-cost[1]-cost[2] (nprob == 3), or null
    int* ticket;          // arrival counter for loss_out: zero on entry, reset to zero by the last workgroup
    int shortcut;         // 1: exact periodic-state shortcut enabled (see sinkhorn_fwd_reg)
};

// Load the EPT contiguous duals a thread needs (entries q*EPT .. q*EPT+EPT-1) from LDS.
template <int EPT>
__device__ __forceinline__ void load_other(float (&o)[EPT], const float* arr, int q) {
    if constexpr (EPT >= 4) {
#pragma unroll
        for (int m = 0; m < EPT; m += 4) {
            const float4 v = *reinterpret_cast<const float4*>(arr + q * EPT + m);
            o[m] = v.x; o[m + 1] = v.y; o[m + 2] = v.z; o[m + 3] = v.w;
        }
    } else {
#pragma unroll
        for (int m = 0; m < EPT; ++m) o[m] = arr[q * EPT + m];
    }
}

// ---- log2-domain half-step ---------------------------------------------------------------------
// With k2 = log2(e)/eps the kernels carry  c2 = C*k2,  U = u*k2,  V = v*k2  and evaluate
//     y = (U_i - c2_ij) + V_j            [= ((-C+u)+v)/eps * log2(e), same association as gan_utils.py:153]
//     lse2 = log2(sum_j exp2(y - max)) + max,      U_i <- (log2(1/n) - lse2) + U_i
// which is the reference's update  u <- eps*(log(1/n) - LSE) + u  multiplied through by k2: two
// adds per entry instead of add/add/mul/mul, exp2/log2 are single hardware instructions, and the
// two-wide vector type lets hipcc issue packed v_pk_add_f32.  The iterates are the same numbers in
// other units; only fp32 rounding differs (the duals are O(1e3) with an ulp of 1e-4 either way).
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int EPT, int LPR, bool ROW>
__device__ __forceinline__ float half_step(const float (&c2)[EPT], float self, const float* other, int q, float lw2) {
    float o[EPT];
    load_other<EPT>(o, other, q);
    float y[EPT];
    if constexpr (EPT >= 2) {
#pragma unroll
        for (int m = 0; m < EPT; m += 2) {
            f32x2 cc = {c2[m], c2[m + 1]}, oo = {o[m], o[m + 1]}, ss = {self, self};
            const f32x2 t = ROW ? ((ss - cc) + oo) : ((oo - cc) + ss);
            y[m] = t.x; y[m + 1] = t.y;
        }
    } else {
        y[0] = ROW ? ((self - c2[0]) + o[0]) : ((o[0] - c2[0]) + self);
    }
    float mx = y[0];
#pragma unroll
    for (int m = 1; m < EPT; ++m) mx = fmaxf(mx, y[m]);
    mx = seg_max<LPR>(mx);
    // tf.reduce_logsumexp: a non-finite max is replaced by 0
    const float shift = (mx > -INFINITY && mx < INFINITY) ? mx : 0.f;
    float e[EPT];
    if constexpr (EPT >= 2) {
        const f32x2 sh = {shift, shift};
#pragma unroll
        for (int m = 0; m < EPT; m += 2) {
            const f32x2 yy = {y[m], y[m + 1]};
            const f32x2 d = yy - sh;
            e[m] = __builtin_amdgcn_exp2f(d.x);
            e[m + 1] = __builtin_amdgcn_exp2f(d.y);
        }
    } else {
        e[0] = __builtin_amdgcn_exp2f(y[0] - shift);
    }
    // pairwise tree: log2(EPT) dependent adds instead of EPT
#pragma unroll
    for (int w = 1; w < EPT; w *= 2)
#pragma unroll
        for (int m = 0; m + w < EPT; m += 2 * w) e[m] += e[m + w];
    const float s = seg_sum<LPR>(e[0]);
    const float lse2 = __builtin_amdgcn_logf(s) + shift;
    return (lw2 - lse2) + self;   // gan_utils.py:154,156 in log2 units
}

// c2 = C * k2 in both orientations; entries past the matrix edge are +inf (-> exp2(-inf) = 0)
template <int EPT, int LPR>
__device__ __forceinline__ void load_costs(const float* __restrict__ C, int n, int line, int q, float k2,
                                           float (&crow)[EPT], float (&ccol)[EPT]) {
#pragma unroll
    for (int m = 0; m < EPT; ++m) {
        const int idx = q * EPT + m;
        const bool ok = line < n && idx < n;
        crow[m] = ok ? C[(int64_t)line * n + idx] * k2 : INFINITY;
        ccol[m] = ok ? C[(int64_t)idx * n + line] * k2 : INFINITY;
    }
}

template <int EPT, int LPR, bool SHORTCUT>
__device__ __forceinline__ void sinkhorn_fwd_body(const SinkArgs& a) {
    // padded so that the float4 reads of lines past n stay inside the arrays
    __shared__ __attribute__((aligned(16))) float u_s[SK_MAXN + 16 * 16];
    __shared__ __attribute__((aligned(16))) float v_s[SK_MAXN + 16 * 16];
    __shared__ float red[16];
    const int p = blockIdx.x, n = a.n;
    const int t = threadIdx.x, line = t / LPR, q = t % LPR;
    const bool active = line < n;
    const float* C = a.C + (int64_t)p * n * n;
    const float k2 = a.inv_eps * LOG2E;

    float crow[EPT], ccol[EPT];
    load_costs<EPT, LPR>(C, n, line, q, k2, crow, ccol);
    for (int i = t; i < SK_MAXN + 16 * 16; i += blockDim.x) { u_s[i] = 0.f; v_s[i] = 0.f; }   // gan_utils.py:147
    __syncthreads();

    const float lw2 = __builtin_amdgcn_logf(1.0f / (float)n);   // log2(mu) = log2(nu), gan_utils.py:138-139
    // stop rule in log2 units: sum|u-u_prev| = sum|U-U_prev| * eps*ln2
    const float err_scale = a.eps * LN2;

    // EXACT shortcut.  One iteration is a deterministic function of the state (u,v).  As soon as the
    // state after iteration k equals, bit for bit, the state after iteration k-p (p <= 4) the
    // sequence is periodic from there on, and the state after any later iteration K is a stored
    // state congruent to K modulo p: nothing is approximated, the loop is merely not re-executed.
    // (Sharp problems -- costs of O(1e3) against eps = 1, the GAN regime -- reach an fp32 fixed point
    // within a handful of iterations.)  The kernel jumps to one iteration before the first point at
    // which the reference's stop rule could fire (or before L), fills the history the backward
    // needs, and resumes the ordinary loop, so the stop logic and the final iterate are computed
    // by the same code as without the shortcut.  nits_out[p] is the reference-equivalent iteration
    // count; nits_out[nprob + p] the number of iterations actually executed.
    //
    // Detection costs no LDS round trip on the critical path: every lane keeps its line's last four
    // duals in registers and compares in place; one ballot per candidate period folds the wave's
    // lines, lane 0 ORs the wave's mismatch mask into an LDS word, and the word is read back after
    // the closing barrier but only CONSUMED one iteration later (its latency hides under the next
    // half-step).  The ring of the last four states in LDS is written every iteration and read
    // only at the jump.
    __shared__ float ring_u[4][SK_MAXN], ring_v[4][SK_MAXN];
    __shared__ int mis[4];
    if (SHORTCUT) {
        if (t < 4) mis[t] = 0;
        __syncthreads();
    }
    bool detect = SHORTCUT;
    int computed = 0;
    int mprev = ~0;                       // mismatch mask of the previous iteration (bit p: state != state p iterations earlier)
    float pu2 = 0.f, pu3 = 0.f, pu4 = 0.f, pv2 = 0.f, pv3 = 0.f, pv4 = 0.f;

    int nits = 0;
    float ui = 0.f, vj = 0.f;   // this line's duals (every lane of the line holds them)
    for (int it = 0; it < a.L; ++it) {
        // every lane executes the half-steps (DPP reads neighbours); only real lines store
        KCCOT_STAMP(0);
        const float un = half_step<EPT, LPR, true>(crow, ui, v_s, q, lw2);
        KCCOT_STAMP(1);
        const float du = (active && q == 0) ? fabsf(un - ui) : 0.f;
        int bits = 0;
        if (SHORTCUT && detect) {
            const unsigned b = __float_as_uint(un);
            bits = (b != __float_as_uint(ui) ? 2 : 0) | (b != __float_as_uint(pu2) ? 4 : 0) |
                   (b != __float_as_uint(pu3) ? 8 : 0) | (b != __float_as_uint(pu4) ? 16 : 0);
            pu4 = pu3; pu3 = pu2; pu2 = ui;
        }
        ui = un;
        if (active && q == 0) {
            u_s[line] = un;
            if (a.u_hist) a.u_hist[((int64_t)p * a.L + it) * n + line] = un;
            if (SHORTCUT && detect) ring_u[it & 3][line] = un;
        }
        KCCOT_STAMP(2);
        lds_barrier();      // the history stores stay in flight (nothing in this kernel reads them back)
        KCCOT_STAMP(3);
        const float vn = half_step<EPT, LPR, false>(ccol, vj, u_s, q, lw2);
        KCCOT_STAMP(4);
        if (SHORTCUT && detect) {
            const unsigned b = __float_as_uint(vn);
            bits |= (b != __float_as_uint(vj) ? 2 : 0) | (b != __float_as_uint(pv2) ? 4 : 0) |
                    (b != __float_as_uint(pv3) ? 8 : 0) | (b != __float_as_uint(pv4) ? 16 : 0);
            pv4 = pv3; pv3 = pv2; pv2 = vj;
            if (!active) bits = 0;
            int wb = 0;
#pragma unroll
            for (int pp = 1; pp <= 4; ++pp)
                if (__builtin_amdgcn_ballot_w64((bits >> pp) & 1)) wb |= 1 << pp;
            if ((t & 63) == 0 && wb) atomicOr(&mis[it & 3], wb);
        }
        vj = vn;
        if (active && q == 0) {
            v_s[line] = vn;
            if (a.v_hist) a.v_hist[((int64_t)p * a.L + it) * n + line] = vn;
            if (SHORTCUT && detect) ring_v[it & 3][line] = vn;
        }
        KCCOT_STAMP(5);
        lds_barrier();
        KCCOT_STAMP(6);
        nits = it + 1;
        ++computed;
        // gan_utils.py:157-160 (count-based) / :115-117 (index-based).  err is only needed once
        // the stop rule can fire, and never on the last iteration.
        const bool reached = (a.stop_mode == KCCOT_STOP_INDEX) ? (it >= a.Lmin) : (nits >= a.Lmin);
        if (reached && it + 1 < a.L) {
            const float err = block_sum(du, red) * err_scale;
            if (a.thresh > err) break;
        }
        if (SHORTCUT && detect) {
            const int mcur = mis[it & 3];                 // consumed at the end of the NEXT iteration
            if (t == 0) mis[(it + 2) & 3] = 0;
            // mprev describes iteration it-1: bit pp clear <=> S_it == S_{it-pp} (valid for it-1 >= pp)
            int per = 0;
#pragma unroll
            for (int pp = 4; pp >= 1; --pp)
                if (it - 1 >= pp && !((mprev >> pp) & 1)) per = pp;          // smallest matching period
            mprev = mcur;
            KCCOT_STAMP(7);
            if (per) {
                detect = false;
                // first iteration count at which the reference could leave its loop early
                int first_stop = (a.stop_mode == KCCOT_STOP_INDEX) ? a.Lmin + 1 : a.Lmin;
                if (first_stop < 1) first_stop = 1;
                const int K1 = (a.L < first_stop ? a.L : first_stop) - 1;   // resume so that iteration K1+1 is real
                if (K1 > nits) {
                    // the states from S_{it-per} on have period per, and the ring holds S_{nits-3..nits}
                    // (S_k in slot (k-1) & 3): S_k = S_src(k), src(k) = lo + (k - lo) mod per, lo = nits-per+1
                    const int lo = nits - per + 1;
                    if (a.u_hist) {
                        for (int e = t; e < (K1 - nits) * n; e += blockDim.x) {
                            const int k = nits + 1 + e / n, i = e % n;
                            const int slot = (lo + (k - lo) % per - 1) & 3;
                            a.u_hist[((int64_t)p * a.L + (k - 1)) * n + i] = ring_u[slot][i];
                            a.v_hist[((int64_t)p * a.L + (k - 1)) * n + i] = ring_v[slot][i];
                        }
                    }
                    const int slot = (lo + (K1 - lo) % per - 1) & 3;
                    if (t < n) { u_s[t] = ring_u[slot][t]; v_s[t] = ring_v[slot][t]; }
                    __syncthreads();
                    ui = u_s[active ? line : 0];
                    vj = v_s[active ? line : 0];
                    nits = K1;
                    it = K1 - 1;
                }
            }
        }
    }

    // gan_utils.py:162-164: pi = exp((-C + u + v^T)/eps); cost = sum(pi * C)   (C re-read: L2-resident)
    float part = 0.f;
    if (active) {
#pragma unroll
        for (int m = 0; m < EPT; ++m) {
            const int idx = q * EPT + m;
            if (idx < n) {
                const float pi = __builtin_amdgcn_exp2f((ui - crow[m]) + v_s[idx]);
                part += pi * C[(int64_t)line * n + idx];
                if (a.pi_out) a.pi_out[(int64_t)p * n * n + (int64_t)line * n + idx] = pi;
            }
        }
    }
    const float cost = block_sum(part, red);
    if (t == 0) {
        // the last of the three workgroups to arrive combines the costs (gan_utils.py:225).  Placement-independent
        // hand-off without fences -- the FIRST ROW of MI355X_MICROARCH.md's table of measured hand-offs ("one lane of each
        // storing workgroup ... an agent-scope atomic add; the workgroup whose add came last, told by the value its add
        // returned, loads only after its add has returned; stores all sc1, loads all sc1, 4-byte"): the handed-off word is
        // stored with agent scope (written through), the store drained (vmcnt(0)), then the relaxed ticket; the last arriver
        // reads the costs with agent-scope loads.  This is gfx950 behaviour, not a guarantee of the HIP memory model (under
        // which it is a race): the emitted cache-policy bits and the drain are pinned by
        // tests/test_abi.py::test_isa_of_the_three_cost_hand_off_to_the_combining_workgroup, and this library builds for
        // gfx950 only.  (Until round 3: plain store + agent release fence + acquire fence in the last arriver -- an L2
        // write-back and an L1 invalidate, ~1.7 us each, on the one thread the workgroup then waits for.)
        __hip_atomic_store(a.cost_out + p, cost, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        a.nits_out[p] = nits;
        a.nits_out[gridDim.x + p] = computed;
        if (a.loss_out) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            const int tk = __hip_atomic_fetch_add(a.ticket, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (tk == (int)gridDim.x - 1) {
                const float c0 = __hip_atomic_load(a.cost_out + 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const float c1 = __hip_atomic_load(a.cost_out + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const float c2 = __hip_atomic_load(a.cost_out + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                a.loss_out[0] = (2.0f * c0 - c1) - c2;
                __hip_atomic_store(a.ticket, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    }
}

// the default (exact periodic-state shortcut compiled in) and the every-iteration variant
template <int EPT, int LPR>
__global__ __launch_bounds__(SK_MAXT) void sinkhorn_fwd_reg(SinkArgs a) { sinkhorn_fwd_body<EPT, LPR, true>(a); }
template <int EPT, int LPR>
__global__ __launch_bounds__(SK_MAXT) void sinkhorn_fwd_reg_full(SinkArgs a) { sinkhorn_fwd_body<EPT, LPR, false>(a); }

// ------------------------------------------------------------------------------------------
// reverse sweep
//   u_t,i = a - eps*LSE_j((-C_ij + u_{t-1,i} + v_{t-1,j})/eps) + u_{t-1,i},  a = eps*log(1/n)
//   v_t,j = a - eps*LSE_i((-C_ij + u_{t,i}   + v_{t-1,j})/eps) + v_{t-1,j}
// With P_t = softmax_j of the first argument and Q_t = softmax_i of the second:
//   du_t/dv_{t-1} = -P_t,  du_t/dC = +P_t,  du_t/du_{t-1} = 1 - sum_j P_t = 0
//   dv_t/du_t     = -Q_t,  dv_t/dC = +Q_t,  dv_t/dv_{t-1} = 1 - sum_i Q_t = 0
// and, from the update rules themselves, no reduction has to be redone:
//   Q_t[i,j] = exp((-C_ij + u_t,i + v_t,j     - a)/eps) = exp2((U_t,i - c2_ij) + V_t,j     - log2(1/n))
//   P_t[i,j] = exp((-C_ij + u_t,i + v_{t-1,j} - a)/eps) = exp2((U_t,i - c2_ij) + V_{t-1,j} - log2(1/n))
// (the two "= 0" terms are 1e-7-sized rounding residues in the reference's tape; dropped).
// The history holds the duals in the forward's log2 units (U, V); gradients are in natural units.
//
// Per iteration, newest first:   (A) row pass with Q_t:    gu_i -= sum_j Q_ij gv_j ; dC += Q_ij gv_j
//                                (B) column pass with P_t: gv_j = -sum_i P_ij gu_i ; dC += P_ij gu_i
// u_t / v_t come from the forward's history; the vectors of the NEXT (older) iteration are
// prefetched from global memory into registers one iteration ahead and parked in a
// double-buffered LDS slot, so the dependent chain sees LDS latency only: two barriers per
// iteration.
// ------------------------------------------------------------------------------------------
struct SinkBwdArgs {
    const float* C;
    const float* u_hist;
    const float* v_hist;
    const int32_t* nits;
    const float* gcost;   // [nprob], or with div_weights: ONE upstream scalar dLoss/dloss
    float* dC;
    int n, L;
    float eps, inv_eps;
    int div_weights;      // 1: gcost[p] = {2,-1,-1}[p] * gcost[0]   (d(2 xy - xx - yy), gan_utils.py:225)
};

template <int EPT, int LPR>
__global__ __launch_bounds__(SK_MAXT) void sinkhorn_bwd_reg(SinkBwdArgs a) {
    constexpr int PADN = SK_MAXN + 16 * 16;
    __shared__ __attribute__((aligned(16))) float U[2][PADN];
    __shared__ __attribute__((aligned(16))) float V[2][PADN];
    __shared__ __attribute__((aligned(16))) float gu[PADN];
    __shared__ __attribute__((aligned(16))) float gv[PADN];
    const int p = blockIdx.x, n = a.n;
    const int t = threadIdx.x, line = t / LPR, q = t % LPR;
    const bool active = line < n;
    const float* C = a.C + (int64_t)p * n * n;
    const float k2 = a.inv_eps * LOG2E;
    const float g = a.div_weights ? (p == 0 ? 2.0f : -1.0f) * a.gcost[0] : a.gcost[p];
    const int nits = a.nits[p];
    const float* uh = a.u_hist + (int64_t)p * a.L * n;
    const float* vh = a.v_hist + (int64_t)p * a.L * n;

    float crow[EPT], ccol[EPT], drow[EPT], dcol[EPT];
    load_costs<EPT, LPR>(C, n, line, q, k2, crow, ccol);
#pragma unroll
    for (int m = 0; m < EPT; ++m) { drow[m] = 0.f; dcol[m] = 0.f; }
    for (int i = t; i < PADN; i += blockDim.x) {
        U[0][i] = U[1][i] = V[0][i] = V[1][i] = 0.f;
        gu[i] = gv[i] = 0.f;
    }
    __syncthreads();
    // history index k holds (U_{k+1}, V_{k+1}); iteration `it` (1-based) lives in slot it & 1;
    // V_0 = 0
    auto hist_u = [&](int it, int i) { return it >= 1 ? uh[(int64_t)(it - 1) * n + i] : 0.f; };
    auto hist_v = [&](int it, int i) { return it >= 1 ? vh[(int64_t)(it - 1) * n + i] : 0.f; };
    if (t < n) {
        U[nits & 1][t] = hist_u(nits, t);
        V[nits & 1][t] = hist_v(nits, t);
        V[(nits - 1) & 1][t] = hist_v(nits - 1, t);
    }
    // prefetch for the first in-loop refill: U_{nits-1}, V_{nits-2}
    float nu = (t < n) ? hist_u(nits - 1, t) : 0.f;
    float nv = (t < n) ? hist_v(nits - 2, t) : 0.f;
    __syncthreads();

    const int lsafe = active ? line : 0;
    // cost = sum_ij pi_ij C_ij, pi = exp2((U - c2) + V);  C/eps = c2*ln2:
    //   dcost/dC_ij (direct) = pi_ij (1 - C_ij/eps); dcost/du_i = sum_j pi_ij C_ij/eps; same for v
    {
        const float* Uc = U[nits & 1];
        const float* Vc = V[nits & 1];
        const float ui = Uc[lsafe], vj = Vc[lsafe];
        float ov[EPT], ou[EPT];
        load_other<EPT>(ov, Vc, q);
        load_other<EPT>(ou, Uc, q);
        float su = 0.f, sv = 0.f;
#pragma unroll
        for (int m = 0; m < EPT; ++m) {
            const bool ok = active && (q * EPT + m) < n;
            const float cr = ok ? crow[m] : 0.f, cc = ok ? ccol[m] : 0.f;
            const float pr = ok ? __builtin_amdgcn_exp2f((ui - cr) + ov[m]) : 0.f;
            drow[m] = g * pr * (1.f - cr * LN2);
            su += pr * cr;
            const float pc = ok ? __builtin_amdgcn_exp2f((ou[m] - cc) + vj) : 0.f;
            sv += pc * cc;
        }
        su = seg_sum<LPR>(su);
        sv = seg_sum<LPR>(sv);
        if (active && q == 0) { gu[line] = g * su * LN2; gv[line] = g * sv * LN2; }
    }
    const float lw2 = __builtin_amdgcn_logf(1.0f / (float)n);
    __syncthreads();

    for (int it = nits; it >= 1; --it) {
        const float* Uc = U[it & 1];
        const float* Vc = V[it & 1];
        const float* Vp = V[(it - 1) & 1];
        // (A) through v_t: row pass with Q_t
        {
            const float ui = Uc[lsafe] - lw2;
            float ov[EPT], og[EPT];
            load_other<EPT>(ov, Vc, q);
            load_other<EPT>(og, gv, q);
            float s = 0.f;
#pragma unroll
            for (int m = 0; m < EPT; ++m) {
                // entries past the edge: c2 = +inf -> exp2(-inf) = 0
                const float w = __builtin_amdgcn_exp2f((ui - crow[m]) + ov[m]) * og[m];
                drow[m] += w;
                s += w;
            }
            s = seg_sum<LPR>(s);
            // grad wrt u_t: the final-cost term on the last iteration, nothing on older ones
            if (active && q == 0) gu[line] = (it == nits ? gu[line] : 0.f) - s;
        }
        lds_barrier();      // LDS-only: the history prefetch (nu, nv) stays in flight across it
        // refill the slots the older iteration needs: U[(it-1)&1] <- U_{it-1}, V[it&1] <- V_{it-2}
        // (V_t is dead after pass A; U_{it-1}'s slot was last read in iteration it+1)
        if (t < n) {
            U[(it - 1) & 1][t] = nu;
            V[it & 1][t] = nv;
            nu = hist_u(it - 2, t);
            nv = hist_v(it - 3, t);
        }
        // (B) through u_t: column pass with P_t
        {
            const float vj = Vp[lsafe] - lw2;
            float ou[EPT], og[EPT];
            load_other<EPT>(ou, Uc, q);
            load_other<EPT>(og, gu, q);
            float r = 0.f;
#pragma unroll
            for (int m = 0; m < EPT; ++m) {
                const float w = __builtin_amdgcn_exp2f((ou[m] - ccol[m]) + vj) * og[m];
                dcol[m] += w;
                r += w;
            }
            r = seg_sum<LPR>(r);
            if (active && q == 0) gv[line] = -r;
        }
        lds_barrier();
    }

    // dC = row-layout part + (column-layout part)^T
    float* dC = a.dC + (int64_t)p * n * n;
    if (active) {
#pragma unroll
        for (int m = 0; m < EPT; ++m) {
            const int idx = q * EPT + m;
            if (idx < n) dC[(int64_t)line * n + idx] = drow[m];
        }
    }
    __threadfence_block();
    __syncthreads();
    if (active) {
#pragma unroll
        for (int m = 0; m < EPT; ++m) {
            const int idx = q * EPT + m;
            if (idx < n) dC[(int64_t)idx * n + line] += dcol[m];
        }
    }
}

// ------------------------------------------------------------------------------------------
// Fused solve + reverse sweep of the mixed divergence (compute_sinkhorn_loss when a gradient is wanted):
// ONE persistent launch per loss, one workgroup per problem, the whole dual history in LDS.
//
//   * the history row written by a half-step IS the exchange array the next half-step reads (row k of `hu` holds
//     U_k, row 0 = zeros): one ds_write per line and half-step instead of an exchange write, a global history store
//     and (shortcut build) a ring store; rows k-3 .. k are also the ring the exact periodic-state shortcut needs;
//   * the reverse sweep starts the moment the final cost is known and reads U_t / V_t / V_{t-1} straight from LDS:
//     no second launch, no reload of C (it is still in registers in both orientations), no 2*L*n-float history
//     round trip through global memory, no prefetch / refill step inside the sweep;
//   * d loss / d C3 is produced for dLoss = 1 with the weights {2,-1,-1} of gan_utils.py:225; the cost backward
//     multiplies by the upstream scalar when it builds its coefficients (the gradient is linear in it).
// Arithmetic is that of sinkhorn_fwd_body / sinkhorn_bwd_reg, instruction for instruction: costs, iteration counts and
// duals are bit-identical to the two-kernel path, dC to the two-kernel path at the same lanes-per-line.
// Eligibility (host): n <= 128 and 2 (L+1) NS floats of history within the CU's LDS; otherwise the two kernels run.
// ------------------------------------------------------------------------------------------
struct SinkFusedArgs {
    const float* C;       // [3,n,n]
    int n, L, Lmin;
    float eps, inv_eps, thresh;
    float* cost_out;      // [3]
    int32_t* nits_out;    // [6]
    float* loss_out;      // [1]
    int* ticket;          // zero on entry, left zero
    float* dC;            // [3,n,n]
};

template <int EPT, int LPR, bool SHORTCUT>
__global__ __launch_bounds__(LPR * SK_MAXN < SK_MAXT ? LPR * SK_MAXN : SK_MAXT) void sinkhorn_fused_reg(SinkFusedArgs a) {
    constexpr int NS = EPT * LPR;                              // history row stride (>= n, a multiple of 4)
    extern __shared__ __attribute__((aligned(16))) float hist[];
    __shared__ __attribute__((aligned(16))) float gu[SK_MAXN + 16 * 16];
    __shared__ __attribute__((aligned(16))) float gv[SK_MAXN + 16 * 16];
    __shared__ float red[16];
    __shared__ int mis[4];
    const int p = blockIdx.x, n = a.n, L = a.L;
    const int t = threadIdx.x, line = t / LPR, q = t % LPR;
    const bool active = line < n;
    const float* C = a.C + (int64_t)p * n * n;
    const float k2 = a.inv_eps * LOG2E;
    float* hu = hist;                                          // row k: U_k (k = 0 .. L), row 0 = 0
    float* hv = hist + (size_t)(L + 1) * NS;

    float crow[EPT], ccol[EPT];
    load_costs<EPT, LPR>(C, n, line, q, k2, crow, ccol);
    // row 0 and the pad columns [n, NS) of every row are zero (a pad dual pairs with c2 = +inf -> exp2(-inf) = 0)
    for (int i = t; i < NS; i += blockDim.x) { hu[i] = 0.f; hv[i] = 0.f; }
    if (NS > n) {
        const int padw = NS - n;
        for (int e = t; e < L * padw; e += blockDim.x) {
            const int k = 1 + e / padw, i = n + e % padw;
            hu[k * NS + i] = 0.f; hv[k * NS + i] = 0.f;
        }
    }
    for (int i = t; i < SK_MAXN + 16 * 16; i += blockDim.x) { gu[i] = 0.f; gv[i] = 0.f; }
    if (t < 4) mis[t] = 0;
    __syncthreads();

    const float lw2 = __builtin_amdgcn_logf(1.0f / (float)n);
    const float err_scale = a.eps * LN2;
    bool detect = SHORTCUT;
    int computed = 0;
    int mprev = ~0;
    float pu2 = 0.f, pu3 = 0.f, pu4 = 0.f, pv2 = 0.f, pv3 = 0.f, pv4 = 0.f;
    int nits = 0;
    float ui = 0.f, vj = 0.f;
    // ---------------------------------------------------------------- forward (see sinkhorn_fwd_body)
    for (int it = 0; it < L; ++it) {
        const float un = half_step<EPT, LPR, true>(crow, ui, hv + it * NS, q, lw2);
        const float du = (active && q == 0) ? fabsf(un - ui) : 0.f;
        int bits = 0;
        if (SHORTCUT && detect) {
            const unsigned b = __float_as_uint(un);
            bits = (b != __float_as_uint(ui) ? 2 : 0) | (b != __float_as_uint(pu2) ? 4 : 0) |
                   (b != __float_as_uint(pu3) ? 8 : 0) | (b != __float_as_uint(pu4) ? 16 : 0);
            pu4 = pu3; pu3 = pu2; pu2 = ui;
        }
        ui = un;
        if (active && q == 0) hu[(it + 1) * NS + line] = un;
        lds_barrier();
        const float vn = half_step<EPT, LPR, false>(ccol, vj, hu + (it + 1) * NS, q, lw2);
        if (SHORTCUT && detect) {
            const unsigned b = __float_as_uint(vn);
            bits |= (b != __float_as_uint(vj) ? 2 : 0) | (b != __float_as_uint(pv2) ? 4 : 0) |
                    (b != __float_as_uint(pv3) ? 8 : 0) | (b != __float_as_uint(pv4) ? 16 : 0);
            pv4 = pv3; pv3 = pv2; pv2 = vj;
            if (!active) bits = 0;
            int wb = 0;
#pragma unroll
            for (int pp = 1; pp <= 4; ++pp)
                if (__builtin_amdgcn_ballot_w64((bits >> pp) & 1)) wb |= 1 << pp;
            if ((t & 63) == 0 && wb) atomicOr(&mis[it & 3], wb);
        }
        vj = vn;
        if (active && q == 0) hv[(it + 1) * NS + line] = vn;
        lds_barrier();
        nits = it + 1;
        ++computed;
        if (nits >= a.Lmin && it + 1 < L) {                   // gan_utils.py:157-160
            const float err = block_sum(du, red) * err_scale;
            if (a.thresh > err) break;
        }
        if (SHORTCUT && detect) {
            const int mcur = mis[it & 3];
            if (t == 0) mis[(it + 2) & 3] = 0;
            int per = 0;
#pragma unroll
            for (int pp = 4; pp >= 1; --pp)
                if (it - 1 >= pp && !((mprev >> pp) & 1)) per = pp;
            mprev = mcur;
            if (per) {
                detect = false;
                int first_stop = a.Lmin < 1 ? 1 : a.Lmin;
                const int K1 = (L < first_stop ? L : first_stop) - 1;
                if (K1 > nits) {
                    // states from S_{nits-per+1} on repeat with period per: row k <- row lo + (k - lo) mod per
                    const int lo = nits - per + 1;
                    for (int e = t; e < (K1 - nits) * n; e += blockDim.x) {
                        const int k = nits + 1 + e / n, i = e % n;
                        const int src = lo + (k - lo) % per;
                        hu[k * NS + i] = hu[src * NS + i];
                        hv[k * NS + i] = hv[src * NS + i];
                    }
                    __syncthreads();
                    ui = hu[K1 * NS + (active ? line : 0)];
                    vj = hv[K1 * NS + (active ? line : 0)];
                    nits = K1;
                    it = K1 - 1;
                }
            }
        }
    }

    // gan_utils.py:162-164: pi = exp((-C + u + v^T)/eps); cost = sum(pi * C)
    const float* Vn = hv + nits * NS;
    const float* Un = hu + nits * NS;
    float part = 0.f;
    if (active) {
#pragma unroll
        for (int m = 0; m < EPT; ++m) {
            const int idx = q * EPT + m;
            if (idx < n) {
                const float pi = __builtin_amdgcn_exp2f((ui - crow[m]) + Vn[idx]);
                part += pi * C[(int64_t)line * n + idx];
            }
        }
    }
    const float cost = block_sum(part, red);
    if (t == 0) {
        // (fence-free hand-off as in sinkhorn_fwd_body: agent-scope store, drained, then the ticket)
        __hip_atomic_store(a.cost_out + p, cost, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        a.nits_out[p] = nits;
        a.nits_out[gridDim.x + p] = computed;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const int tk = __hip_atomic_fetch_add(a.ticket, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (tk == (int)gridDim.x - 1) {
            const float c0 = __hip_atomic_load(a.cost_out + 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const float c1 = __hip_atomic_load(a.cost_out + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const float c2 = __hip_atomic_load(a.cost_out + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            a.loss_out[0] = (2.0f * c0 - c1) - c2;
            __hip_atomic_store(a.ticket, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }

    // ---------------------------------------------------------------- reverse sweep (see sinkhorn_bwd_reg)
    const float g = (p == 0) ? 2.0f : -1.0f;                  // d(2 xy - xx - yy) at dLoss = 1
    const int lsafe = active ? line : 0;
    float drow[EPT], dcol[EPT];
    {
        const float uf = Un[lsafe], vf = Vn[lsafe];
        float ov[EPT], ou[EPT];
        load_other<EPT>(ov, Vn, q);
        load_other<EPT>(ou, Un, q);
        float su = 0.f, sv = 0.f;
#pragma unroll
        for (int m = 0; m < EPT; ++m) {
            const bool ok = active && (q * EPT + m) < n;
            const float cr = ok ? crow[m] : 0.f, cc = ok ? ccol[m] : 0.f;
            const float pr = ok ? __builtin_amdgcn_exp2f((uf - cr) + ov[m]) : 0.f;
            drow[m] = g * pr * (1.f - cr * LN2);
            dcol[m] = 0.f;
            su += pr * cr;
            const float pc = ok ? __builtin_amdgcn_exp2f((ou[m] - cc) + vf) : 0.f;
            sv += pc * cc;
        }
        su = seg_sum<LPR>(su);
        sv = seg_sum<LPR>(sv);
        if (active && q == 0) { gu[line] = g * su * LN2; gv[line] = g * sv * LN2; }
    }
    __syncthreads();
    // The plans Q_t (pass A) and P_t (pass B) depend on the HISTORY only, not on the running gradients: the dependent
    // chain of a half-step is gv -> 8 multiply-adds -> DPP sum -> gu.  Their exp2 are therefore evaluated one pass
    // AHEAD, in the shadow of the LDS round trip that follows every barrier (ds_read of the gradients): the same
    // instructions, issued while the wave would otherwise wait.
    float qa[EPT], pb[EPT];
    auto plan_a = [&](int it) {      // Q_it[line][q*EPT+m] = exp2((U_it,line - lw2 - c) + V_it,col)
        const float uu = hu[it * NS + lsafe] - lw2;
        float ov[EPT];
        load_other<EPT>(ov, hv + it * NS, q);
#pragma unroll
        for (int m = 0; m < EPT; ++m) qa[m] = __builtin_amdgcn_exp2f((uu - crow[m]) + ov[m]);
    };
    auto plan_b = [&](int it) {      // P_it[q*EPT+m][line] = exp2((U_it,row - c) + V_{it-1,line} - lw2)
        const float vv = hv[(it - 1) * NS + lsafe] - lw2;
        float ou[EPT];
        load_other<EPT>(ou, hu + it * NS, q);
#pragma unroll
        for (int m = 0; m < EPT; ++m) pb[m] = __builtin_amdgcn_exp2f((ou[m] - ccol[m]) + vv);
    };
    if (nits >= 1) plan_a(nits);
    for (int it = nits; it >= 1; --it) {
        {   // (A) through v_t: row pass with Q_t
            float og[EPT];
            load_other<EPT>(og, gv, q);
            plan_b(it);
            __builtin_amdgcn_sched_barrier(0);          // keep the exp2 of the next pass in the shadow of the gv read
            float sa = 0.f;
#pragma unroll
            for (int m = 0; m < EPT; ++m) {
                const float w = qa[m] * og[m];
                drow[m] += w;
                sa += w;
            }
            sa = seg_sum<LPR>(sa);
            if (active && q == 0) gu[line] = (it == nits ? gu[line] : 0.f) - sa;
        }
        lds_barrier();
        {   // (B) through u_t: column pass with P_t
            float og[EPT];
            load_other<EPT>(og, gu, q);
            if (it > 1) plan_a(it - 1);
            __builtin_amdgcn_sched_barrier(0);
            float r = 0.f;
#pragma unroll
            for (int m = 0; m < EPT; ++m) {
                const float w = pb[m] * og[m];
                dcol[m] += w;
                r += w;
            }
            r = seg_sum<LPR>(r);
            if (active && q == 0) gv[line] = -r;
        }
        lds_barrier();
    }
    // dC = row-layout part + (column-layout part)^T
    float* dC = a.dC + (int64_t)p * n * n;
    if (active) {
#pragma unroll
        for (int m = 0; m < EPT; ++m) {
            const int idx = q * EPT + m;
            if (idx < n) dC[(int64_t)line * n + idx] = drow[m];
        }
    }
    __threadfence_block();
    __syncthreads();
    if (active) {
#pragma unroll
        for (int m = 0; m < EPT; ++m) {
            const int idx = q * EPT + m;
            if (idx < n) dC[(int64_t)idx * n + line] += dcol[m];
        }
    }
}

// gan_utils.py:225: loss = 2.0 * loss_xy - loss_xx - loss_yy, evaluated left to right in fp32
__global__ void mixed_divergence_fwd(const float* __restrict__ cost3, float* __restrict__ loss) {
    if (threadIdx.x == 0) loss[0] = (2.0f * cost3[0] - cost3[1]) - cost3[2];
}
__global__ void mixed_divergence_bwd(const float* __restrict__ gloss, float* __restrict__ gcost3) {
    if (threadIdx.x == 0) {
        const float g = gloss[0];
        gcost3[0] = 2.0f * g; gcost3[1] = -g; gcost3[2] = -g;
    }
}

// ------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------
struct SinkGeom { int lpr, ept, threads; };

static SinkGeom sink_geom(int n, bool forward) {
    SinkGeom g;
    // n*lpr <= 1024.  Every wave of the workgroup runs the same instruction stream between two
    // barriers, four waves take turns on each SIMD at 1024 threads, and the per-wave overhead (DPP
    // reduction steps, log, stores) does not shrink with the number of entries per lane.  Measured
    // at n = 64 over 100 iterations (tools/bench_sinkhorn.py, tools/ab_sk.sh): forward 8 lanes per
    // line 72 us, 16: 82 us, 4: slower still; the reverse sweep is within 2 us between 8 and 16
    // (72 us) and keeps 16.
    g.lpr = (n <= 32) ? 16 : (n <= 64 ? (forward ? 8 : 16) : 8);
    if (n > 32 && n <= 64) {
        // option "sinkhorn_lanes_per_line" = 4, 8 or 16 (0 = the rule above): the fused kernel runs 8, so equality tests of
        // its gradients against the two-kernel path put both on 8
        const int v = opt(OPT_SK_LPR);
        if (v == 4 || v == 8 || v == 16) g.lpr = v;
    }
    const int need = (n + g.lpr - 1) / g.lpr;
    g.ept = need <= 1 ? 1 : need <= 2 ? 2 : need <= 4 ? 4 : need <= 8 ? 8 : 16;
    g.threads = (n * g.lpr + 63) / 64 * 64;
    return g;
}

}  // namespace kccot

namespace kccot {
// sinkhorn_gen.hip: streaming kernels for 128 < n <= 1024
size_t sinkhorn_gen_workspace_bytes(int nprob, int n);
int launch_sinkhorn_fwd_gen(const float* C, int nprob, int n, float eps, int L, int Lmin, float thresh, int stop_mode,
                            float* u_hist, float* v_hist, float* cost_out, int32_t* nits_out, float* pi_out, void* ws,
                            size_t ws_bytes, hipStream_t st);
int launch_sinkhorn_bwd_gen(const float* C, const float* u_hist, const float* v_hist, const int32_t* nits, int nprob, int n,
                            float eps, int L, const float* gcost, float* dC, void* ws, size_t ws_bytes, hipStream_t st);
}  // namespace kccot

using namespace kccot;

// option "sinkhorn_shortcut" = 0: always execute every iteration (bench headline, bitwise-equality tests)
static int sink_shortcut_enabled() { return opt(OPT_SK_SHORTCUT); }

// set by the *_divergence_* entry points around their call into the base functions
static thread_local float* g_div_loss = nullptr;
static thread_local int* g_div_ticket = nullptr;
static thread_local int g_div_weights = 0;

extern "C" size_t kccot_sinkhorn_workspace_bytes(int nprob, int n) {
    if (nprob <= 0 || n <= SK_MAXN) return 0;   // the register-resident kernels need none
    return sinkhorn_gen_workspace_bytes(nprob, n);
}

#define KCCOT_SK_LAUNCH(KERNEL, E, P, ARGS, GEOM, NPROB, ST) \
    hipLaunchKernelGGL((KERNEL<E, P>), dim3(NPROB), dim3((GEOM).threads), 0, ST, ARGS)

#define KCCOT_SK_DISPATCH(KERNEL, ARGS, GEOM, NPROB, ST)                                       \
    if ((GEOM).lpr == 16) {                                                                    \
        switch ((GEOM).ept) {                                                                  \
            case 1: KCCOT_SK_LAUNCH(KERNEL, 1, 16, ARGS, GEOM, NPROB, ST); break;              \
            case 2: KCCOT_SK_LAUNCH(KERNEL, 2, 16, ARGS, GEOM, NPROB, ST); break;              \
            default: KCCOT_SK_LAUNCH(KERNEL, 4, 16, ARGS, GEOM, NPROB, ST); break;             \
        }                                                                                      \
    } else if ((GEOM).lpr == 4) {                                                              \
        KCCOT_SK_LAUNCH(KERNEL, 16, 4, ARGS, GEOM, NPROB, ST);                                 \
    } else {                                                                                   \
        switch ((GEOM).ept) {                                                                  \
            case 8: KCCOT_SK_LAUNCH(KERNEL, 8, 8, ARGS, GEOM, NPROB, ST); break;               \
            default: KCCOT_SK_LAUNCH(KERNEL, 16, 8, ARGS, GEOM, NPROB, ST); break;             \
        }                                                                                      \
    }

extern "C" int kccot_sinkhorn_fwd_f32(const float* C, int nprob, int n, float eps, int L, int Lmin,
                                      float thresh, int stop_mode, float* u_hist, float* v_hist,
                                      float* cost_out, int32_t* nits_out, float* pi_out, void* ws,
                                      size_t ws_bytes, kccot_stream_t stream) {
    if (!C || !cost_out || !nits_out) return fail(KCCOT_EINVAL, "sinkhorn_fwd: null pointer");
    if (nprob <= 0 || n <= 0 || L < 0 || !(eps > 0.f))
        return fail(KCCOT_EINVAL, "sinkhorn_fwd: bad arguments nprob=%d n=%d L=%d eps=%g", nprob, n, L, (double)eps);
    if ((u_hist == nullptr) != (v_hist == nullptr))
        return fail(KCCOT_EINVAL, "sinkhorn_fwd: u_hist and v_hist must be given together");
    if (stop_mode != KCCOT_STOP_COUNT && stop_mode != KCCOT_STOP_INDEX)
        return fail(KCCOT_EINVAL, "sinkhorn_fwd: bad stop_mode %d", stop_mode);
    if (n > SK_MAXN)
        return launch_sinkhorn_fwd_gen(C, nprob, n, eps, L, Lmin, thresh, stop_mode, u_hist, v_hist, cost_out, nits_out,
                                       pi_out, ws, ws_bytes, (hipStream_t)stream);
    SinkGeom g = sink_geom(n, true);
    SinkArgs a{C, n, L, Lmin, stop_mode, eps, (float)(1.0 / (double)eps), thresh, u_hist, v_hist, cost_out, nits_out, pi_out,
               nullptr, g_div_loss, g_div_ticket, sink_shortcut_enabled()};
#ifdef KCCOT_DIAG
    a.diag = static_cast<unsigned long long*>(ws);   // diagnostic build: ws carries the stamp buffer
#endif
    hipStream_t st = (hipStream_t)stream;
    if (a.shortcut) {
        KCCOT_SK_DISPATCH(sinkhorn_fwd_reg, a, g, nprob, st)
    } else {
        KCCOT_SK_DISPATCH(sinkhorn_fwd_reg_full, a, g, nprob, st)
    }
    return launch_status("sinkhorn_fwd_reg");
}

extern "C" int kccot_sinkhorn_status(const int32_t* nits, int nprob, kccot_stream_t stream) {
    if (!nits || nprob <= 0 || nprob > 4096) return fail(KCCOT_EINVAL, "sinkhorn_status: bad arguments");
    int32_t host[4096];
    hipStream_t st = (hipStream_t)stream;
    if (hipMemcpyAsync(host, nits, sizeof(int32_t) * nprob, hipMemcpyDeviceToHost, st) != hipSuccess ||
        hipStreamSynchronize(st) != hipSuccess)
        return fail(KCCOT_EINVAL, "sinkhorn_status: could not read the iteration counts");
    for (int p = 0; p < nprob; ++p)
        if (host[p] < 0)
            return fail(KCCOT_EABORTED, "sinkhorn: problem %d of %d was aborted -- a workgroup of the multi-CU solver never "
                        "arrived (device shared or partitioned?); set KCCOT_SK_NO_COOP=1 to use the one-workgroup solver", p, nprob);
    return 0;
}

extern "C" int kccot_sinkhorn_bwd_f32(const float* C, const float* u_hist, const float* v_hist,
                                      const int32_t* nits, int nprob, int n, float eps, int L,
                                      const float* gcost, float* dC_out, void* ws, size_t ws_bytes,
                                      kccot_stream_t stream) {
    if (!C || !u_hist || !v_hist || !nits || !gcost || !dC_out)
        return fail(KCCOT_EINVAL, "sinkhorn_bwd: null pointer");
    if (nprob <= 0 || n <= 0 || L < 0 || !(eps > 0.f))
        return fail(KCCOT_EINVAL, "sinkhorn_bwd: bad arguments nprob=%d n=%d L=%d eps=%g", nprob, n, L, (double)eps);
    if (n > SK_MAXN)
        return launch_sinkhorn_bwd_gen(C, u_hist, v_hist, nits, nprob, n, eps, L, gcost, dC_out, ws, ws_bytes,
                                       (hipStream_t)stream);
    SinkGeom g = sink_geom(n, false);
    SinkBwdArgs a{C, u_hist, v_hist, nits, gcost, dC_out, n, L, eps, (float)(1.0 / (double)eps), g_div_weights};
    hipStream_t st = (hipStream_t)stream;
    KCCOT_SK_DISPATCH(sinkhorn_bwd_reg, a, g, nprob, st)
    return launch_status("sinkhorn_bwd_reg");
}

extern "C" int kccot_mixed_divergence_fwd_f32(const float* cost3, float* loss_out, kccot_stream_t stream) {
    if (!cost3 || !loss_out) return fail(KCCOT_EINVAL, "mixed_divergence_fwd: null pointer");
    hipLaunchKernelGGL(mixed_divergence_fwd, dim3(1), dim3(64), 0, (hipStream_t)stream, cost3, loss_out);
    return launch_status("mixed_divergence_fwd");
}

extern "C" int kccot_mixed_divergence_bwd_f32(const float* gloss, float* gcost3_out, kccot_stream_t stream) {
    if (!gloss || !gcost3_out) return fail(KCCOT_EINVAL, "mixed_divergence_bwd: null pointer");
    hipLaunchKernelGGL(mixed_divergence_bwd, dim3(1), dim3(64), 0, (hipStream_t)stream, gloss, gcost3_out);
    return launch_status("mixed_divergence_bwd");
}

// ---- fused solve + reverse sweep (sinkhorn_fused_reg) -------------------------------------------------
static size_t fused_hist_bytes(int n, int L) {
    const SinkGeom g = sink_geom(n, true);
    return (size_t)2 * ((size_t)L + 1) * g.lpr * g.ept * sizeof(float);
}

extern "C" int kccot_sinkhorn_fused_eligible(int n, int L) {
    if (!opt(OPT_SK_FUSED)) return 0;                     // option "sinkhorn_fused" = 0: always the two-kernel path
    // n <= 64 by default: at 64 < n <= 128 (16 entries per lane and orientation) the fused kernel spills under the
    // 128-VGPR cap of a 1024-thread workgroup and measured SLOWER than the two kernels at configs[2]
    // (1.24 vs 1.08 ms per loss); option "sinkhorn_fused_max_n" = 128 re-enables it.
    const int maxn = opt(OPT_SK_FUSED_MAX_N);
    if (n <= 0 || n > SK_MAXN || n > maxn || L < 0) return 0;
    return fused_hist_bytes(n, L) <= (size_t)144 * 1024;  // + ~4 KB of static LDS, inside the CU's 160 KB
}

template <int EPT, int LPR, bool SC>
static int launch_fused(const SinkFusedArgs& a, size_t lds, hipStream_t st) {
    // on every launch: the attribute belongs to the current device's copy of the function (a per-process "done" flag
    // would be wrong on a second device and racy between threads); it is a host-side table write
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&sinkhorn_fused_reg<EPT, LPR, SC>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, 144 * 1024) != hipSuccess)
        return fail(KCCOT_EUNSUPPORTED, "sinkhorn_fused: cannot raise the dynamic LDS limit");
    hipLaunchKernelGGL((sinkhorn_fused_reg<EPT, LPR, SC>), dim3(3), dim3((a.n * LPR + 63) / 64 * 64), lds, st, a);
    return launch_status("sinkhorn_fused_reg");
}

template <bool SC>
static int dispatch_fused(const SinkGeom& g, const SinkFusedArgs& a, size_t lds, hipStream_t st) {
    if (g.lpr == 16) {
        switch (g.ept) {
            case 1: return launch_fused<1, 16, SC>(a, lds, st);
            case 2: return launch_fused<2, 16, SC>(a, lds, st);
            default: return launch_fused<4, 16, SC>(a, lds, st);
        }
    }
    if (g.lpr == 4) return launch_fused<16, 4, SC>(a, lds, st);
    return g.ept == 8 ? launch_fused<8, 8, SC>(a, lds, st) : launch_fused<16, 8, SC>(a, lds, st);
}

// The three solves of compute_sinkhorn_loss, their combination AND the reverse sweep in ONE launch:
// dC3_unit [3,n,n] = d loss / d C3 at dLoss = 1.  No dual history leaves the CU.  `ticket` as above.
extern "C" int kccot_sinkhorn_divergence_fused_f32(const float* C3, int n, float eps, int L, int Lmin, float thresh,
                                                   float* cost3_out, int32_t* nits_out, float* loss_out, int32_t* ticket,
                                                   float* dC3_unit, kccot_stream_t stream) {
    if (!C3 || !cost3_out || !nits_out || !loss_out || !ticket || !dC3_unit)
        return fail(KCCOT_EINVAL, "sinkhorn_divergence_fused: null pointer");
    if (!(eps > 0.f)) return fail(KCCOT_EINVAL, "sinkhorn_divergence_fused: eps=%g", (double)eps);
    if (!kccot_sinkhorn_fused_eligible(n, L))
        return fail(KCCOT_EUNSUPPORTED, "sinkhorn_divergence_fused: n=%d L=%d does not fit (see kccot_sinkhorn_fused_eligible)", n, L);
    const SinkGeom g = sink_geom(n, true);
    SinkFusedArgs a{C3, n, L, Lmin, eps, (float)(1.0 / (double)eps), thresh, cost3_out, nits_out, loss_out,
                    reinterpret_cast<int*>(ticket), dC3_unit};
    const size_t lds = fused_hist_bytes(n, L);
    hipStream_t st = (hipStream_t)stream;
    return sink_shortcut_enabled() ? dispatch_fused<true>(g, a, lds, st) : dispatch_fused<false>(g, a, lds, st);
}

// Mixed Sinkhorn divergence in one launch each way (compute_sinkhorn_loss, gan_utils.py:221-225):
// the three solves of C3 = [xy, xx, yy] plus loss = 2 xy - xx - yy, combined by the last workgroup
// to finish.  `ticket` is ONE device int that must be zero on entry (the kernel leaves it zero).
extern "C" int kccot_sinkhorn_divergence_fwd_f32(const float* C3, int n, float eps, int L, int Lmin, float thresh,
                                                 float* u_hist, float* v_hist, float* cost3_out, int32_t* nits_out,
                                                 float* loss_out, int32_t* ticket, void* ws, size_t ws_bytes,
                                                 kccot_stream_t stream) {
    if (!loss_out || !ticket) return fail(KCCOT_EINVAL, "sinkhorn_divergence_fwd: null pointer");
    if (n > SK_MAXN) {   // streaming solver: no in-kernel combine; fall back to the separate launch
        int rc = kccot_sinkhorn_fwd_f32(C3, 3, n, eps, L, Lmin, thresh, KCCOT_STOP_COUNT, u_hist, v_hist, cost3_out, nits_out,
                                        nullptr, ws, ws_bytes, stream);
        if (rc) return rc;
        return kccot_mixed_divergence_fwd_f32(cost3_out, loss_out, stream);
    }
    g_div_loss = loss_out; g_div_ticket = reinterpret_cast<int*>(ticket);
    int rc = kccot_sinkhorn_fwd_f32(C3, 3, n, eps, L, Lmin, thresh, KCCOT_STOP_COUNT, u_hist, v_hist, cost3_out, nits_out,
                                    nullptr, ws, ws_bytes, stream);
    g_div_loss = nullptr; g_div_ticket = nullptr;
    return rc;
}

// gloss: ONE device float dLoss/dloss; dC3_out = d loss / d C3 scaled by it.
extern "C" int kccot_sinkhorn_divergence_bwd_f32(const float* C3, const float* u_hist, const float* v_hist,
                                                 const int32_t* nits, int n, float eps, int L, const float* gloss,
                                                 float* dC3_out, void* ws, size_t ws_bytes, kccot_stream_t stream) {
    if (!gloss) return fail(KCCOT_EINVAL, "sinkhorn_divergence_bwd: null pointer");
    if (n > SK_MAXN) return fail(KCCOT_EUNSUPPORTED, "sinkhorn_divergence_bwd: use mixed_divergence_bwd + sinkhorn_bwd for n > %d", SK_MAXN);
    g_div_weights = 1;
    int rc = kccot_sinkhorn_bwd_f32(C3, u_hist, v_hist, nits, 3, n, eps, L, gloss, dC3_out, ws, ws_bytes, stream);
    g_div_weights = 0;
    return rc;
}
