// Multi-CU Sinkhorn for 128 < n <= 1024 (BASELINE configs 3-5 batch sizes).
//
// One CU streaming an n x n cost matrix per half-step is issue-bound (n = 256: 5-10 us per half-step, n = 512:
// four times that).  Here a problem is spread over ceil(n/16) workgroups: workgroup g owns lines 16g .. 16g+15,
// one wave per line, and keeps BOTH orientations of its lines (row i of C and column i of C, n/64 entries per
// lane each) in registers for the whole solve -- the matrix is read once.  A half-step is then a few dozen
// instructions per wave; what it costs is the exchange: every workgroup publishes its 16 new duals, all
// workgroups of the problem meet at a device-wide barrier, and every lane re-reads the n/64 duals of its
// columns.  Two barriers per iteration (~2 us each) replace two passes over the matrix.
//
// The exchange is flag-in-data (below): every dual travels as one 64-bit {value, half-step tag} word, no counter
// barrier, no fences.  (Round 1's counter-barrier kernels -- release fence, atomic arrival counter, polling, acquire
// fence: 0.79 / 0.86 ms per 100 iterations at n = 256 against 0.45 / 0.46 -- and the one-XCD block map were removed
// in round 3; their A/B records are profiles/r01zo_* and DESIGN.md section 4.)
#include "common.h"
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <stddef.h>
#include "options.h"

namespace kccot {

constexpr int SC_LINES = 16;                 // lines (= waves) per workgroup
constexpr int SC_THREADS = SC_LINES * 64;
constexpr int SC_MAXWG = 64;                 // n <= 1024
constexpr unsigned SC_SPIN_LIMIT = 1u << 20; // polls (an L2 round trip each) before a barrier gives up: ~1 s
constexpr float SC_LOG2E = 1.4426950408889634f;
constexpr float SC_LN2 = 0.6931471805599453f;

struct CoopCtrl {            // per launch, zeroed in front of it -- except `epoch`, which ll_zero INCREMENTS
    unsigned bar[32];        // arrival counters, one per problem (ll_same_xcd)
    unsigned long long epoch;// launch counter of this workspace: the upper bits of every exchange tag (ll_tag_base)
    int abort_flag;
    unsigned xcc_ref[8];     // XCC_ID + 1 of workgroup 0 of the problem, 0 = not published yet
    unsigned mismatch[8];    // != 0: some workgroup of the problem sits on another XCD
    int pad[13];
};
static_assert(sizeof(CoopCtrl) == 256, "the exchange words behind it are 8-byte aligned and the area is zeroed in 8-byte words");
static_assert(offsetof(CoopCtrl, epoch) == 128, "ll_zero skips (and bumps) 64-bit word 16 of the area");
constexpr int LL_EPOCH_WORD = 16;

// one line's dual update: lane holds entries idx = lane + 64 e of its line (c) and of the other side's duals (o)
template <int EPT, bool ROW>
__device__ __forceinline__ float coop_update(const float (&c)[EPT], const float (&o)[EPT], int n, int lane, float self, float eps,
                                             float inv_eps, float log_w) {
    float tt[EPT];
    float mx = -INFINITY;
#pragma unroll
    for (int e = 0; e < EPT; ++e) {
        const float t = ROW ? ((-c[e] + self) + o[e]) : ((-c[e] + o[e]) + self);   // gan_utils.py:153-156
        tt[e] = (lane + 64 * e < n) ? t * inv_eps : -INFINITY;
        mx = fmaxf(mx, tt[e]);
    }
    mx = wave_max_fast(mx);
    const float shift = (mx > -INFINITY && mx < INFINITY) ? mx : 0.f;       // tf.reduce_logsumexp
    float s = 0.f;
#pragma unroll
    for (int e = 0; e < EPT; ++e) s += __builtin_amdgcn_exp2f((tt[e] - shift) * SC_LOG2E);
    s = wave_sum_fast(s);
    const float lse = __builtin_amdgcn_logf(s) * SC_LN2 + shift;
    return eps * (log_w - lse) + self;
}

// ------------------------------------------------------------------------------------------------------
// Flag-in-data exchange ("LL" kernels).
//
// A counter barrier costs ~3.9 us per half-step at n = 256: a workgroup barrier that drains the store
// queue, an L2 write-back (release), an atomic round trip, polling, an invalidate (acquire) and only then the loads
// of the new duals.  Here every dual travels as ONE 64-bit word {value, tag}: the tag is the half-step number, the
// store and the load are relaxed agent-scope atomics (a naturally aligned 64-bit access is single-copy atomic), so
// a reader that sees the tag it waits for has the value -- no fences, no counter, no separate flag.  Wave 0 of every
// workgroup polls the n words of the side it needs, puts the values in LDS, and one LDS-only workgroup barrier
// releases the other fifteen waves.  A word is overwritten (tag + 2) only by a workgroup that has gathered the whole
// other side with tag + 1, and those words are published only by workgroups that have gathered this side with
// `tag`: nobody can lose a value.  The exchange words are zeroed in front of every launch (tags start at 1).
// Polling is bounded (SC_SPIN_LIMIT) and honours the abort flag of CoopCtrl: a missing workgroup drains the launch.
// The stop test needs sum_i |u_i - u_i_prev| (gan_utils.py:157): every wave holds all of u (new and previous) across
// its lanes after the gather, so each wave sums it for itself in the same fixed order -- identical decisions, no
// exchange.
//
// Tags carry the LAUNCH in their upper bits (round 4): tag = (epoch & 0x7FF) << 20 | half-step, epoch = a counter that lives
// in the exchange area and is incremented by the zeroing kernel in front of every launch -- also in front of every REPLAY of a
// captured launch, which a host-side counter in the kernel arguments could not be (they are frozen at capture).  Half-step
// numbers restart at 1 in every launch; without the epoch a word left over from the previous launch that escaped the zeroing
// (round 3 saw exactly that with memset nodes in a graph) would carry a valid-looking tag with an OLD value and be consumed
// silently.  With it such a word can only make a poll run to its bound: NaN + KCCOT_EABORTED, never a plausible number.
// (2047 launches later the epoch bits repeat; the zeroing kernel remains the primary guarantee, the epoch the second.)
typedef unsigned long long ll_word;
constexpr unsigned LL_STEP_BITS = 20;                  // half-steps per launch < 2^20 - 1 (host check: 2 L + 2 < 0xFFFFF)
constexpr unsigned LL_FINAL_STEP = 0xFFFFFu;

__device__ __forceinline__ unsigned ll_tag_base(const CoopCtrl* c) {
    const unsigned long long e = __hip_atomic_load(&c->epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return ((unsigned)e & 0x7FFu) << LL_STEP_BITS;
}

// local != 0 (one problem per XCD, ll_block): the word is stored with workgroup scope (sc0: it stays in the L2 of the
// writer's XCD, which every reader of the problem shares; their sc1 loads bypass only their own L1).  Otherwise agent scope
// (sc1: written through to memory, where a reader on any XCD finds it).
__device__ __forceinline__ void ll_store(ll_word* p, float v, unsigned tag, int local) {
    const ll_word w = ((ll_word)tag << 32) | (ll_word)__float_as_uint(v);
    if (local) __hip_atomic_store(p, w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    else __hip_atomic_store(p, w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// One problem per XCD is how the launch is LAID OUT; whether the hardware dealt the workgroups that way is checked here, once
// per launch, with agent-scope traffic only (three round trips, ~5 us): workgroup 0 of the problem publishes its XCC_ID, every
// workgroup compares its own, flags a mismatch, arrives at the problem's counter and waits for all nwg arrivals.  true = every
// workgroup of problem p runs on one XCD and the L2-served exchange (ll_store with local != 0) may be used; false = agent scope
// as before (a partitioned device, a CU mask, another dispatch order: slower, never wrong).  A workgroup that gives up
// raises the abort flag like any other poll.  One wave per workgroup runs it; the result goes through LDS.
__device__ __forceinline__ bool ll_same_xcd(CoopCtrl* c, int p, int wg, int nwg) {
    unsigned xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    xcc = (xcc & 0xFu) + 1u;
    if (wg == 0) __hip_atomic_store(&c->xcc_ref[p], xcc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    unsigned spins = 0, ref;
    while ((ref = __hip_atomic_load(&c->xcc_ref[p], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) == 0u) {
        if ((++spins & 127u) == 0 && (spins > SC_SPIN_LIMIT || __hip_atomic_load(&c->abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))) {
            __hip_atomic_store(&c->abort_flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            return false;
        }
        __builtin_amdgcn_s_sleep(1);
    }
    if (ref != xcc) __hip_atomic_store(&c->mismatch[p], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    // the mismatch store of this workgroup is ordered in front of its arrival (release), the arrivals in front of the read
    __hip_atomic_fetch_add(&c->bar[p], 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    spins = 0;
    while (__hip_atomic_load(&c->bar[p], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)nwg) {
        if ((++spins & 127u) == 0 && (spins > SC_SPIN_LIMIT || __hip_atomic_load(&c->abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))) {
            __hip_atomic_store(&c->abort_flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            return false;
        }
        __builtin_amdgcn_s_sleep(1);
    }
    return __hip_atomic_load(&c->mismatch[p], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u;
}

// Block -> (problem, workgroup of the problem).  xcd_map: a 1-D grid of 8 * nwg blocks, block id -> problem id % 8,
// workgroup id / 8 (workgroups are dealt to the 8 XCDs round-robin, so problem p lives on XCD p; ids whose residue is not a
// problem leave at once); otherwise the 2-D grid (nwg, nprob).
__device__ __forceinline__ bool ll_block(int xcd_map, int nprob, int& p, int& wg) {
    if (xcd_map == 1) { p = blockIdx.x & 7; wg = blockIdx.x >> 3; return p < nprob; }
    p = blockIdx.y; wg = blockIdx.x;
    return true;
}

// Gather of x[0..n) with tag `tag` into sh[0..n): wave c < ceil(n/256) polls words 256c .. 256c+255 (four per lane).
// false = gave up (abort raised).
__device__ __forceinline__ bool ll_gather(const ll_word* x, unsigned tag, float* sh, int lane, int chunk, int n, int* abort_flag) {
    unsigned spins = 0;
    const int base = chunk * 256 + lane;
    for (;;) {
        ll_word w[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int idx = base + 64 * e;
            w[e] = __hip_atomic_load(x + (idx < n ? idx : n - 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        bool ok = true;
#pragma unroll
        for (int e = 0; e < 4; ++e) ok = ok && ((unsigned)(w[e] >> 32) == tag);
        if (__builtin_amdgcn_ballot_w64(!ok) == 0) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int idx = base + 64 * e;
                if (idx < n) sh[idx] = __uint_as_float((unsigned)w[e]);
            }
            return true;
        }
        if ((++spins & 127u) == 0) {
            if (spins > SC_SPIN_LIMIT || __hip_atomic_load(abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
                __hip_atomic_store(abort_flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                return false;
            }
        }
        __builtin_amdgcn_s_sleep(1);
    }
}

struct SinkLLArgs {
    const float* C;
    int n, L, Lmin, stop_mode;
    float eps, inv_eps, thresh;
    float* u_hist;
    float* v_hist;
    float* cost_out;
    int32_t* nits_out;
    float* pi_out;
    CoopCtrl* ctrl;
    ll_word* xu;      // [nprob][n]
    ll_word* xv;      // [nprob][n]
    ll_word* xcost;   // [nprob][SC_MAXWG]
    int nwg, nprob;
    int fault;        // libkccot_diag.so only (KCCOT_SK_FAULT_INJECT=1): the last workgroup of problem 0 never takes part
    int xcd_map;      // ll_block
};

template <int EPT>
__global__ __launch_bounds__(SC_THREADS) void sinkhorn_fwd_ll(SinkLLArgs a) {
    __shared__ float shu[SC_MAXWG * SC_LINES];
    __shared__ float shv[SC_MAXWG * SC_LINES];
    __shared__ float red[SC_MAXWG];
    __shared__ int bflag;
    int p, wg;
    if (!ll_block(a.xcd_map, a.nprob, p, wg)) return;
    if (a.fault && p == 0 && wg == a.nwg - 1) return;      // stands in for a workgroup that is not resident
    const int n = a.n, nwg = a.nwg;
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    const int line = wg * SC_LINES + w;
    const bool live = line < n;
    const int lsafe = live ? line : n - 1;
    const float* C = a.C + (int64_t)p * n * n;
    ll_word* xu = a.xu + (int64_t)p * n;
    ll_word* xv = a.xv + (int64_t)p * n;
    const float eps = a.eps, inv_eps = a.inv_eps;
    const float log_w = logf(1.0f / (float)n);
    const int nchunk = (n + 255) >> 8;                             // gathering waves
    auto clampi = [&](int e) { const int idx = lane + 64 * e; return idx < n ? idx : n - 1; };   // masked in coop_update
    const unsigned tbase = ll_tag_base(a.ctrl);                    // this launch's epoch (issued with the matrix loads)

    float crow[EPT], ccol[EPT], ov[EPT], ou[EPT];
#pragma unroll
    for (int e = 0; e < EPT; ++e) {
        crow[e] = C[(int64_t)lsafe * n + clampi(e)];
        ccol[e] = C[(int64_t)clampi(e) * n + lsafe];
        ov[e] = 0.f; ou[e] = 0.f;                                  // gan_utils.py:147: u = v = 0
    }
    // (one problem per XCD by layout: verified here, once; `local` selects the L2-served exchange)
    __shared__ int s_local;
    if (t == 0) { bflag = 0; s_local = a.xcd_map ? (ll_same_xcd(a.ctrl, p, wg, a.nwg) ? 1 : 0) : 0; }
    __syncthreads();
    const int local = s_local;
    float ui = 0.f, vj = 0.f;
    int nits = 0;
    bool ok = true;
    for (int it = 0; it < a.L; ++it) {
        const unsigned tagu = tbase + 2u * it + 1u, tagv = tbase + 2u * it + 2u;
        const float un = coop_update<EPT, true>(crow, ov, n, lane, ui, eps, inv_eps, log_w);
        ui = un;
        if (lane == 0 && live) {
            ll_store(xu + line, un, tagu, local);
            if (a.u_hist) a.u_hist[((int64_t)p * a.L + it) * n + line] = un;
        }
        if (w < nchunk && !ll_gather(xu, tagu, shu, lane, w, n, &a.ctrl->abort_flag) && lane == 0) bflag = 1;
        lds_barrier();
        if (*(volatile int*)&bflag) { ok = false; break; }
        float errl = 0.f;
#pragma unroll
        for (int e = 0; e < EPT; ++e) {
            const float nu = shu[clampi(e)];
            errl += (lane + 64 * e < n) ? fabsf(nu - ou[e]) : 0.f;
            ou[e] = nu;
        }
        const float vn = coop_update<EPT, false>(ccol, ou, n, lane, vj, eps, inv_eps, log_w);
        vj = vn;
        if (lane == 0 && live) {
            ll_store(xv + line, vn, tagv, local);
            if (a.v_hist) a.v_hist[((int64_t)p * a.L + it) * n + line] = vn;
        }
        if (w < nchunk && !ll_gather(xv, tagv, shv, lane, w, n, &a.ctrl->abort_flag) && lane == 0) bflag = 1;
        lds_barrier();
        if (*(volatile int*)&bflag) { ok = false; break; }
#pragma unroll
        for (int e = 0; e < EPT; ++e) ov[e] = shv[clampi(e)];
        nits = it + 1;
        // gan_utils.py:157-160 (count-based) / :115-117 (index-based); every wave sums the same values in the same order
        const bool reached = (a.stop_mode == KCCOT_STOP_INDEX) ? (it >= a.Lmin) : (nits >= a.Lmin);
        if (reached && it + 1 < a.L) {
            const float err = wave_sum_fast(errl);
            if (a.thresh > err) break;
        }
    }
    // gan_utils.py:162-164: pi = exp((-C + u + v^T)/eps); cost = sum(pi * C)
    float part = 0.f;
#pragma unroll
    for (int e = 0; e < EPT; ++e) {
        const int idx = lane + 64 * e;
        if (ok && live && idx < n) {
            const float pi = __builtin_amdgcn_exp2f(((-crow[e] + ui) + ov[e]) * inv_eps * SC_LOG2E);
            part += pi * crow[e];
            if (a.pi_out) a.pi_out[(int64_t)p * n * n + (int64_t)line * n + idx] = pi;
        }
    }
    part = wave_sum_fast(part);
    __syncthreads();
    if (lane == 0) red[w] = part;
    __syncthreads();
    if (t == 0) {
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < SC_LINES; ++k) s += red[k];
        if (ok) ll_store(a.xcost + p * SC_MAXWG + wg, s, tbase + LL_FINAL_STEP, local);   // a workgroup that gave up never publishes
    }
    if (wg != 0 || w != 0) return;
    // workgroup 0, wave 0: the per-workgroup parts in workgroup order
    {
        const int k = lane < nwg ? lane : nwg - 1;
        unsigned spins = 0;
        bool got = false, dead = !ok;
        float val = 0.f;
        while (!dead) {
            const ll_word wv = __hip_atomic_load(a.xcost + p * SC_MAXWG + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            got = (unsigned)(wv >> 32) == tbase + LL_FINAL_STEP;
            val = __uint_as_float((unsigned)wv);
            if (__builtin_amdgcn_ballot_w64(!got) == 0) break;
            if ((++spins & 127u) == 0 &&
                (spins > SC_SPIN_LIMIT || __hip_atomic_load(&a.ctrl->abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))) {
                __hip_atomic_store(&a.ctrl->abort_flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                dead = true;
            }
            __builtin_amdgcn_s_sleep(1);
        }
        if (__hip_atomic_load(&a.ctrl->abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) dead = true;
        float s = 0.f;
        for (int q = 0; q < nwg; ++q) s += __shfl(val, q, 64);
        if (lane == 0) {
            a.cost_out[p] = dead ? NAN : s;      // an aborted solve must not look like a result ...
            a.nits_out[p] = dead ? -1 : nits;    // ... and says so: a negative count is the status the host reads (kccot_sinkhorn_status)
            a.nits_out[a.nprob + p] = nits;
        }
    }
}

struct SinkLLBwdArgs {
    const float* C;
    const float* u_hist;
    const float* v_hist;
    const int32_t* nits;
    const float* gcost;
    float* dC;
    float* dCT;
    int n, L;
    float eps, inv_eps;
    CoopCtrl* ctrl;
    ll_word* xgu;
    ll_word* xgv;
    int nwg, nprob;
    int xcd_map;      // ll_block
};

template <int EPT>
__global__ __launch_bounds__(SC_THREADS) void sinkhorn_bwd_ll(SinkLLBwdArgs a) {
    __shared__ float shu[SC_MAXWG * SC_LINES];
    __shared__ float shv[SC_MAXWG * SC_LINES];
    __shared__ int bflag;
    int p, wg;
    if (!ll_block(a.xcd_map, a.nprob, p, wg)) return;
    const int n = a.n;
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    const int line = wg * SC_LINES + w;
    const bool live = line < n;
    const int lsafe = live ? line : n - 1;
    const float* C = a.C + (int64_t)p * n * n;
    ll_word* xgu = a.xgu + (int64_t)p * n;
    ll_word* xgv = a.xgv + (int64_t)p * n;
    const float eps = a.eps, inv_eps = a.inv_eps, g = a.gcost[p];
    const int nits = a.nits[p];
    if (nits < 0) {          // the forward solve of this problem was aborted: every workgroup of it writes NaN and leaves
        for (int e = 0; e < EPT; ++e) {
            const int idx = lane + 64 * e;
            if (live && idx < n) { a.dC[((int64_t)p * n + line) * n + idx] = NAN; a.dCT[((int64_t)p * n + line) * n + idx] = NAN; }
        }
        return;
    }
    const float* uh = a.u_hist + (int64_t)p * a.L * n;     // history index k holds (u_{k+1}, v_{k+1}); u_0 = v_0 = 0
    const float* vh = a.v_hist + (int64_t)p * a.L * n;
    const float aconst = eps * logf(1.0f / (float)n);

    const int nchunk = (n + 255) >> 8;                             // gathering waves
    const unsigned tbase = ll_tag_base(a.ctrl);                    // this launch's epoch
    constexpr bool PREFETCH = EPT <= 8;                            // history loads issued ahead of the gathers (registers permitting)
    auto clampi = [&](int e) { const int idx = lane + 64 * e; return idx < n ? idx : n - 1; };
    auto okc = [&](int e) { return live && lane + 64 * e < n; };
    float crow[EPT], ccol[EPT], drow[EPT], dcol[EPT];
#pragma unroll
    for (int e = 0; e < EPT; ++e) {
        crow[e] = C[(int64_t)lsafe * n + clampi(e)];
        ccol[e] = C[(int64_t)clampi(e) * n + lsafe];
        drow[e] = 0.f; dcol[e] = 0.f;
    }
    // (one problem per XCD by layout: verified here, once; `local` selects the L2-served exchange)
    __shared__ int s_local;
    if (t == 0) { bflag = 0; s_local = a.xcd_map ? (ll_same_xcd(a.ctrl, p, wg, a.nwg) ? 1 : 0) : 0; }
    __syncthreads();
    const int local = s_local;
    auto hist = [&](const float* h, int it, int i) { return it >= 1 ? h[(int64_t)(it - 1) * n + i] : 0.f; };
    // final-cost term: dC = g pi (1 - C/eps); gu = g sum_j pi C / eps; gv likewise
    float gu_line, gv_line;
    {
        const float ui = hist(uh, nits, lsafe), vj = hist(vh, nits, lsafe);
        float su = 0.f, sv = 0.f;
#pragma unroll
        for (int e = 0; e < EPT; ++e) {
            const float vo = hist(vh, nits, clampi(e)), uo = hist(uh, nits, clampi(e));
            const float pr = okc(e) ? __builtin_amdgcn_exp2f(((-crow[e] + ui) + vo) * inv_eps * SC_LOG2E) : 0.f;
            drow[e] = g * pr * (1.f - crow[e] * inv_eps);
            su += pr * crow[e];
            const float pc = okc(e) ? __builtin_amdgcn_exp2f(((-ccol[e] + uo) + vj) * inv_eps * SC_LOG2E) : 0.f;
            sv += pc * ccol[e];
        }
        gu_line = g * wave_sum_fast(su) * inv_eps;
        gv_line = g * wave_sum_fast(sv) * inv_eps;
        if (lane == 0 && live) ll_store(xgv + line, gv_line, tbase + 1u, local);
    }
    unsigned tag = tbase + 1u;       // the tag of the gv values the next (A) pass reads
    for (int it = nits; it >= 1; --it) {
        // prefetch this iteration's history (plain loads: written by the forward kernel)
        float vo[PREFETCH ? EPT : 1], uo[PREFETCH ? EPT : 1];
        const float ui = hist(uh, it, lsafe), vj = hist(vh, it - 1, lsafe);
        if (PREFETCH) {
#pragma unroll
            for (int e = 0; e < EPT; ++e) { vo[e] = hist(vh, it, clampi(e)); uo[e] = hist(uh, it, clampi(e)); }
        }
        // (A) row pass with Q_t: gu_i = [it == nits] gu_i - sum_j Q_ij gv_j ; dC_ij += Q_ij gv_j
        if (w < nchunk && !ll_gather(xgv, tag, shv, lane, w, n, &a.ctrl->abort_flag) && lane == 0) bflag = 1;
        lds_barrier();
        if (*(volatile int*)&bflag) break;
        {
            float s = 0.f;
#pragma unroll
            for (int e = 0; e < EPT; ++e) {
                const float gvj = shv[clampi(e)];
                const float qq = okc(e) ? __builtin_amdgcn_exp2f((((-crow[e] + ui) + (PREFETCH ? vo[PREFETCH ? e : 0] : hist(vh, it, clampi(e)))) - aconst) * inv_eps * SC_LOG2E) : 0.f;
                const float wv = qq * gvj;
                drow[e] += wv;
                s += wv;
            }
            s = wave_sum_fast(s);
            gu_line = (it == nits ? gu_line : 0.f) - s;
            if (lane == 0 && live) ll_store(xgu + line, gu_line, tag + 1u, local);
        }
        // (B) column pass with P_t: gv_j = -sum_i P_ij gu_i ; dC_ij += P_ij gu_i (kept in column layout)
        if (w < nchunk && !ll_gather(xgu, tag + 1u, shu, lane, w, n, &a.ctrl->abort_flag) && lane == 0) bflag = 1;
        lds_barrier();
        if (*(volatile int*)&bflag) break;
        {
            float r = 0.f;
#pragma unroll
            for (int e = 0; e < EPT; ++e) {
                const float gui = shu[clampi(e)];
                const float pp = okc(e) ? __builtin_amdgcn_exp2f((((-ccol[e] + (PREFETCH ? uo[PREFETCH ? e : 0] : hist(uh, it, clampi(e)))) + vj) - aconst) * inv_eps * SC_LOG2E) : 0.f;
                const float wv = pp * gui;
                dcol[e] += wv;
                r += wv;
            }
            r = wave_sum_fast(r);
            gv_line = -r;
            if (lane == 0 && live) ll_store(xgv + line, gv_line, tag + 2u, local);
        }
        tag += 2u;
    }
    const bool bad = *(volatile int*)&bflag != 0;
    float* dC = a.dC + (int64_t)p * n * n;
    float* dCT = a.dCT + (int64_t)p * n * n;
#pragma unroll
    for (int e = 0; e < EPT; ++e) {
        if (okc(e)) {
            dC[(int64_t)line * n + clampi(e)] = bad ? NAN : drow[e];
            dCT[(int64_t)line * n + clampi(e)] = bad ? NAN : dcol[e];      // row `line` of dC^T = column `line` of dC
        }
    }
}

// ---- host side -----------------------------------------------------------------------------------------
size_t sinkhorn_gen_workspace_bytes(int nprob, int n);
__global__ void add_transposed_batched(const float* __restrict__ in, float* __restrict__ out, int n);

static bool coop_enabled() { return opt(OPT_SK_COOP) != 0; }   // option "sinkhorn_coop" = 0: the single-workgroup streaming kernels

// How many 1024-thread workgroups of the cooperative kernels the current device can hold AT ONCE: the spin-wait
// exchanges are only safe when every workgroup of a launch is resident (a workgroup that has not started cannot
// publish what its siblings poll for).  multiProcessorCount x the occupancy of the fattest instantiation (EPT = 16,
// reverse sweep), queried once per device; three quarters of it are offered, so that a co-running kernel (an RCCL
// collective of the data-parallel trainer, another stream) does not turn a legal launch into a bounded-poll abort.
// A partitioned / CU-masked / smaller device simply reports fewer CUs and larger batches take the streaming solver.
// Option "sinkhorn_coop_max_wg" = n > 0 overrides the result (a caller that knows its partition; tests force the fallback).
// `ept` = 0: the bound for any n (the fattest instantiation); 4 / 8 / 16: for the instantiation that serves that n.
static int coop_capacity(int ept = 0) {
    static int cached[64][4];
    static bool have[64];
    if (const int forced = opt(OPT_SK_COOP_MAX_WG)) return forced;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 0;
    if (!have[dev]) {
        int cus = 0, per_cu = 0;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) cus = 0;
        int occ[3] = {1 << 30, 1 << 30, 1 << 30};
#define KCCOT_OCC(K, I)                                                                                           \
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, K, SC_THREADS, 0) != hipSuccess) per_cu = 0;        \
        occ[I] = per_cu < occ[I] ? per_cu : occ[I];
        KCCOT_OCC(sinkhorn_fwd_ll<4>, 0) KCCOT_OCC(sinkhorn_fwd_ll<8>, 1) KCCOT_OCC(sinkhorn_fwd_ll<16>, 2)
        KCCOT_OCC(sinkhorn_bwd_ll<4>, 0) KCCOT_OCC(sinkhorn_bwd_ll<8>, 1) KCCOT_OCC(sinkhorn_bwd_ll<16>, 2)
#undef KCCOT_OCC
        int m = occ[0] < occ[1] ? occ[0] : occ[1];
        m = occ[2] < m ? occ[2] : m;
        cached[dev][0] = (int)((long long)cus * m * 3 / 4);
        for (int i = 0; i < 3; ++i) cached[dev][1 + i] = (int)((long long)cus * occ[i] * 3 / 4);
        have[dev] = true;
    }
    return cached[dev][ept == 4 ? 1 : (ept == 8 ? 2 : (ept == 16 ? 3 : 0))];
}

bool sinkhorn_coop_eligible(int nprob, int n) {
    const int nwg = (n + SC_LINES - 1) / SC_LINES;
    return coop_enabled() && n > 128 && n <= 1024 && nprob <= 32 && nwg * nprob <= coop_capacity();   // all workgroups resident
}

// Fault injection exists in the diagnostic twin only (make libkccot_diag.so; tests/test_gpu_parity.py loads it in a child
// process): one workgroup of problem 0 stays away, which exercises the bounded-poll abort path.
static int fault_injected() {
#ifdef KCCOT_DIAG
    const char* e = getenv("KCCOT_SK_FAULT_INJECT");
    return (e && atoi(e) == 1) ? 1 : 0;
#else
    return 0;
#endif
}

// The exchange area is zeroed by a KERNEL whose stores have the scope of the exchange itself (relaxed, agent), not by
// hipMemsetAsync: as a node of a captured graph the memset's zeros were not what the next node's agent-scope loads saw --
// the first replay of a graph ran on fresh (zero) memory, every later one found the previous replay's tags, polled to its
// bound and aborted to NaN (round 3: found when bench.py replayed the configs[3] step as a hipGraph; eager launches were
// never affected; the same symptom in smooth.hip's three scalars).  Whether the memset node lacked its edge or its writes
// bypassed what the kernels' loads read was not isolated -- a kernel node covers both, and since round 4 a stale word could
// not be CONSUMED even if one survived: the tags carry the launch epoch (ll_tag_base).
// tests/test_gpu_parity.py::test_multi_cu_sinkhorn_replays_as_a_graph (replays with different iteration counts).
// Word LL_EPOCH_WORD (CoopCtrl::epoch) is not zeroed but incremented: whatever the caller's workspace held there before the
// first launch is as good a start as any -- the tags only have to differ from one launch (or graph replay) to the next.
__global__ __launch_bounds__(256) void ll_zero(unsigned long long* __restrict__ w, int n64) {
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n64; i += gridDim.x * 256) {
        unsigned long long v = 0ull;
        if (i == LL_EPOCH_WORD) v = __hip_atomic_load(w + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1ull;
        __hip_atomic_store(w + i, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}
static int ll_zero_area(void* p, size_t bytes, hipStream_t st) {      // bytes % 8 == 0 (CoopCtrl is 256 bytes, ll_word 8)
    const int n64 = (int)(bytes / 8);
    hipLaunchKernelGGL(ll_zero, dim3((n64 + 255) / 256 < 64 ? (n64 + 255) / 256 : 64), dim3(256), 0, st,
                       static_cast<unsigned long long*>(p), n64);
    return launch_status("ll_zero");
}

// option "sinkhorn_coop_xcd" = 1 (default): one problem per XCD (ll_block) -- when there are at most 8 problems and HALF an
// XCD's share of the co-residency capacity holds a whole problem
// (value 2, tests: the 2-D grid launched as if it were laid out per XCD -- the in-kernel check must find the mismatch and
// fall back to the agent-scope exchange)
static int ll_xcd_map(int nprob, int nwg, int ept) {
    const int o = opt(OPT_SK_COOP_XCD);
    // half of an XCD's share of the co-residency capacity (16 of 32 CUs on an MI355X: n <= 256): two processes that share the
    // card (or two streams of one) can then both be resident on the XCD -- with partial residency of both, each would poll
    // for workgroups that cannot start until the other finishes, and both would give up
    return (o && nprob <= 8 && nwg <= coop_capacity(ept) / 12) ? o : 0;
}

// flag-in-data kernels: ctrl | xu [nprob][n] words | xv | xcost [nprob][SC_MAXWG] words, zeroed as one block
struct LLCarve { CoopCtrl* ctrl; ll_word* x0; ll_word* x1; ll_word* xc; size_t zero_bytes; float* second; };
static LLCarve ll_carve(void* ws, int nprob, int n) {
    char* b = static_cast<char*>(ws);
    LLCarve c;
    c.ctrl = reinterpret_cast<CoopCtrl*>(b);
    c.x0 = reinterpret_cast<ll_word*>(b + sizeof(CoopCtrl));
    c.x1 = c.x0 + (size_t)nprob * n;
    c.xc = c.x1 + (size_t)nprob * n;
    c.zero_bytes = sizeof(CoopCtrl) + ((size_t)2 * nprob * n + (size_t)nprob * SC_MAXWG) * sizeof(ll_word);
    c.second = reinterpret_cast<float*>(b + sinkhorn_gen_workspace_bytes(nprob, n) / 2);
    return c;
}
int launch_sinkhorn_fwd_coop(const float* C, int nprob, int n, float eps, int L, int Lmin, float thresh, int stop_mode,
                             float* u_hist, float* v_hist, float* cost_out, int32_t* nits_out, float* pi_out, void* ws,
                             hipStream_t st) {
    const LLCarve lv = ll_carve(ws, nprob, n);
    if (lv.zero_bytes > sinkhorn_gen_workspace_bytes(nprob, n) / 2) return fail(KCCOT_EWORKSPACE, "sinkhorn_fwd(coop): exchange area");
    if (int zrc = ll_zero_area(lv.ctrl, lv.zero_bytes, st)) return zrc;
    const int nwg = (n + SC_LINES - 1) / SC_LINES;
    const int ept = (n + 63) / 64;
    SinkLLArgs a{C, n, L, Lmin, stop_mode, eps, (float)(1.0 / (double)eps), thresh, u_hist, v_hist, cost_out, nits_out, pi_out,
                 lv.ctrl, lv.x0, lv.x1, lv.xc, nwg, nprob, fault_injected(), ll_xcd_map(nprob, nwg, ept <= 4 ? 4 : (ept <= 8 ? 8 : 16))};
    const dim3 grid = a.xcd_map == 1 ? dim3(8 * nwg) : dim3(nwg, nprob);
#define KCCOT_LL(E) hipLaunchKernelGGL(sinkhorn_fwd_ll<E>, grid, dim3(SC_THREADS), 0, st, a)
    if (ept <= 4) KCCOT_LL(4); else if (ept <= 8) KCCOT_LL(8); else KCCOT_LL(16);
#undef KCCOT_LL
    return launch_status("sinkhorn_fwd_ll");
}

int launch_sinkhorn_bwd_coop(const float* C, const float* u_hist, const float* v_hist, const int32_t* nits, int nprob, int n,
                             float eps, int L, const float* gcost, float* dC, void* ws, hipStream_t st) {
    const LLCarve lv = ll_carve(ws, nprob, n);
    if (lv.zero_bytes > sinkhorn_gen_workspace_bytes(nprob, n) / 2) return fail(KCCOT_EWORKSPACE, "sinkhorn_bwd(coop): exchange area");
    if (int zrc = ll_zero_area(lv.ctrl, lv.zero_bytes, st)) return zrc;
    const int nwg = (n + SC_LINES - 1) / SC_LINES;
    const int ept = (n + 63) / 64;
    SinkLLBwdArgs a{C, u_hist, v_hist, nits, gcost, dC, lv.second, n, L, eps, (float)(1.0 / (double)eps), lv.ctrl, lv.x0, lv.x1,
                    nwg, nprob, ll_xcd_map(nprob, nwg, ept <= 4 ? 4 : (ept <= 8 ? 8 : 16))};
    const dim3 grid = a.xcd_map == 1 ? dim3(8 * nwg) : dim3(nwg, nprob);
#define KCCOT_LL(E) hipLaunchKernelGGL(sinkhorn_bwd_ll<E>, grid, dim3(SC_THREADS), 0, st, a)
    if (ept <= 4) KCCOT_LL(4); else if (ept <= 8) KCCOT_LL(8); else KCCOT_LL(16);
#undef KCCOT_LL
    int rc = launch_status("sinkhorn_bwd_ll");
    if (rc) return rc;
    dim3 tg((n + 31) / 32, (n + 31) / 32, nprob);
    hipLaunchKernelGGL(add_transposed_batched, tg, dim3(256), 0, st, (const float*)lv.second, dC, n);
    return launch_status("add_transposed_batched");
}

}  // namespace kccot
