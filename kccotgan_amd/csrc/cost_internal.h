// Internal descriptors shared by cost.hip (direct-difference path + dispatch) and
// cost_mfma.hip (stacked-Gram f32-MFMA path).
#pragma once
#include "common.h"

namespace kccot {

struct CostProb {
    const float* x;   // rows  [Bx,K]
    const float* y;   // cols  [By,K]
    int Bx, By;
    int same;         // x is y: only tiles on/above the diagonal are computed
    const float* h1;  // causal term 1: h1 [Bx,T,J] rows, M1 [By,T,J] cols (or null)
    const float* M1;
    const float* h2;  // bi-causal second term (or null)
    const float* M2;
    float* out;       // [Bx,By]
    float* partial;   // direct path: [nchunk,Bx,By] partial sums of (x-y)^2
    int tile, tiles_i, tiles_j;
    // blocked MFMA path (batches above 64): `out` is a block of a larger matrix with this row pitch (0 = By), and
    // `out_t` (optional) receives the TRANSPOSED distances of the block plus the causal term of (h2 rows, M2 cols)
    // -- the mirror block of an x == y problem, whose features differ from the block's own
    int out_pitch;
    float* out_t;
};

struct CostBatch {
    CostProb p[3];
    int nprob;
};

struct CostPlan {
    bool use_mfma;
    int tile;
    int64_t chunk;
    int nchunk;
    size_t partial_off[3];
    size_t ws_bytes;
};

// Causal term of a 16x16 output tile (block = 256 threads, thread (ti = t>>4, tj = t&15)):
//   sum_{t<T-1} sum_q h[i0+ti,t,q] * (M[j0+tj,t+1,q] - M[j0+tj,t,q])            (gan_utils.py:34-38)
// With k = t*J + q running over (T-1)*J values this is a [16 x KK] x [KK x 16] product of the
// flattened h rows and first differences of the flattened M rows (M[k+J] - M[k]); both are
// staged through LDS in CAUSAL_KC-wide k chunks ((T-1)*J = 232 fits in one at the default T = 30,
// J = 8) so that global reads are coalesced and all issued before the single wait.
// sh, sm: CAUSAL_TILE*CAUSAL_PITCH floats of LDS each (per team).  Every thread of the block must call it.
constexpr int CAUSAL_TILE = 16;
constexpr int CAUSAL_KC = 256;
constexpr int CAUSAL_PITCH = CAUSAL_KC + 4;   // 260: rows 16-byte aligned and 4 banks apart -> conflict-free ds_read_b128

// MG: rows fetched per pass.  16 = all 48 loads of a k chunk in flight (one memory round trip: the callers on a critical
// path); 4 = four passes of 12 loads (the spare workgroups of the Gram launch, which finish long before the Gram chunks do:
// sixteen uniform row bases per operand held at once cost that kernel 72 spilled SGPRs).
template <int MG = CAUSAL_TILE>
__device__ __forceinline__ float causal_tile16(const float* __restrict__ h, const float* __restrict__ M, int i0,
                                               int j0, int Bx, int By, int T, int J, float* sh, float* sm,
                                               int t = threadIdx.x) {
    // t: the caller's index 0..255 inside its 256-thread tile team (a 1024-thread block runs four teams)
    const int ti = t >> 4, tj = t & 15;
    const int KK = (T - 1) * J, TJ = T * J;
    float t0 = 0.f, t1 = 0.f, t2 = 0.f, t3 = 0.f;   // four independent chains of 64 FMAs per chunk
    for (int k0 = 0; k0 < KK; k0 += CAUSAL_KC) {
        // every address is clamped into range so that the 48 loads carry no control dependence and are
        // all in flight before the first wait; out-of-range lanes are zeroed by the selects below
        const int k = k0 + t;
        const bool kok = k < KK;
        const int kc = kok ? k : 0;
#pragma unroll 1
        for (int mb = 0; mb < CAUSAL_TILE; mb += MG) {
            float hv[MG], m0[MG], m1[MG];
#pragma unroll
            for (int q = 0; q < MG; ++q) {
                const int m = mb + q;
                const int ri = (i0 + m < Bx) ? i0 + m : Bx - 1, rj = (j0 + m < By) ? j0 + m : By - 1;
                hv[q] = h[(int64_t)ri * TJ + kc];
                const float* mr = M + (int64_t)rj * TJ + kc;
                m0[q] = mr[0];
                m1[q] = mr[J];
            }
#pragma unroll
            for (int q = 0; q < MG; ++q) {
                const int m = mb + q;
                sh[m * CAUSAL_PITCH + t] = (kok && i0 + m < Bx) ? hv[q] : 0.f;
                sm[m * CAUSAL_PITCH + t] = (kok && j0 + m < By) ? m1[q] - m0[q] : 0.f;
            }
        }
        __syncthreads();
#pragma unroll 8
        for (int kk = 0; kk < CAUSAL_KC; kk += 4) {
            const float4 a = *reinterpret_cast<const float4*>(&sh[ti * CAUSAL_PITCH + kk]);
            const float4 b = *reinterpret_cast<const float4*>(&sm[tj * CAUSAL_PITCH + kk]);
            t0 = fmaf(a.x, b.x, t0); t1 = fmaf(a.y, b.y, t1); t2 = fmaf(a.z, b.z, t2); t3 = fmaf(a.w, b.w, t3);
        }
        __syncthreads();
    }
    return (t0 + t1) + (t2 + t3);
}

// ---- stacked-Gram MFMA path (cost_mfma.hip) -------------------------------------------------
// One "stack" is up to 128 rows: rows 0..63 from src1 (n1 valid), rows 64..127 from src2 (n2
// valid).  Its Gram matrix is produced as 32x32 sub-tiles (a,b), a <= b < 4, index
// sub_index(a,b); `mask` selects the sub-tiles that are needed.
constexpr int GRAM_ROWS = 128;
constexpr int GRAM_KT = 32;
constexpr int GRAM_NSUB = 10;
constexpr int GRAM_REDUCE_SPLIT = 8;

__host__ __device__ inline int sub_index(int a, int b) { return a * 4 - a * (a - 1) / 2 + (b - a); }

struct GramPlan {
    int64_t chunk;
    int nchunk;
    size_t gpart_bytes;   // [nchunk][GRAM_NSUB][1024] float
    size_t gsum_bytes;    // [GRAM_REDUCE_SPLIT][GRAM_NSUB*1024] double
    size_t caus_bytes;    // [3][GRAM_ROWS][GRAM_ROWS] float: causal sums
    size_t ws_bytes;
};

enum GramMode {
    GRAM_LOSS3 = 0,   // src1 = real, src2 = fake, pair-difference stack [X; Y-X]; outputs xy, xx, yy
    GRAM_XY = 1,      // src1 = x, src2 = y, plain stack; output xy
    GRAM_SAME = 2,    // src1 = x rows 0..63, src2 = x rows 64..127; output xx (symmetric)
};

bool gram_eligible(const CostBatch& cb, int64_t K, bool loss3);
bool gram_preferred(const CostBatch& cb, int64_t K, bool loss3);
GramPlan plan_gram(int64_t K);
int run_gram(const CostBatch& cb, bool loss3, int64_t K, float sc, int T, int J, void* ws,
             size_t ws_bytes, bool partial_only, hipStream_t st, int stage = 0);
// cost_tiled.hip: batches that are multiples of 128 -- one Gram of the [real ; fake - real] stack in 128 x 128 tiles
bool gram_tiled_eligible(const CostBatch& cb, int64_t K, bool loss3);
size_t gram_tiled_workspace_bytes(int B, int64_t K);
int run_gram_tiled(const CostBatch& cb, int64_t K, float sc, int T, int J, void* ws, size_t ws_bytes, hipStream_t st,
                   int stage = 0);
void gram_tiled_sums_span(int B, int64_t K, size_t* off, size_t* n);
void gram_sums_span(int B, int64_t K, size_t* off, size_t* n);
// cost_tile256.hip: batches that are multiples of 256 -- the same Gram in 256 x 256 tiles, E written as a by-product
bool gram_q256_applies(int B, int64_t K);
bool gram_q256_eligible(const CostBatch& cb, int64_t K, bool loss3);
size_t gram_q256_workspace_bytes(int B, int64_t K);
int run_gram_q256(const CostBatch& cb, int64_t K, float sc, int T, int J, void* ws, size_t ws_bytes, hipStream_t st,
                  int stage = 0);
void gram_q256_sums_span(int B, int64_t K, size_t* off, size_t* n);
bool gram_blocked_eligible(const CostBatch& cb, int64_t K, bool loss3);
// cost_bwd_q256.hip: the video gradient on 256 x 256 output tiles (W panel through LDS)
bool apply_q256_applies(int Bout, int n1, int n2, int64_t K);
int launch_apply_q256(const unsigned short* Wt3use, int Bt, int Rt, const float* s1, int n1, const float* s2, int n2, int Bout,
                      int64_t K, float* out, hipStream_t st);
int run_gram_blocked(const CostBatch& cb, int64_t K, float sc, int T, int J, void* ws, size_t ws_bytes, hipStream_t st);

}  // namespace kccot
