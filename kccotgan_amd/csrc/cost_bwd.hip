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
#include "cost_internal.h"
#include <stdlib.h>
#include "options.h"

namespace kccot {

enum CoeffMode { CO_LOSS3_DFAKE = 0, CO_DX = 1, CO_DY = 2, CO_SAME = 3 };

// exact three-way split of an fp32 value into bf16 pieces (the upper 16 bits of h, m, l): x = h + m + l
__device__ __forceinline__ void split3u(float x, unsigned& h, unsigned& m, unsigned& l) {
    const unsigned xb = __float_as_uint(x);
    const float hf = __uint_as_float(xb & 0xFFFF0000u);
    const float r1 = x - hf;                                               // exact
    const float mf = __uint_as_float(__float_as_uint(r1) & 0xFFFF0000u);
    h = xb; m = __float_as_uint(r1); l = __float_as_uint(r1 - mf);         // pieces = the upper 16 bits of each word
}

// Wt is [R][Bout] (stack-row major) so that one output row block reads contiguous scalars.
// R = n1 + n2 stack rows: first the n1 rows of src1, then the n2 rows of src2.
// One block per output row m: thread r writes Wt[r][m]; the row / column sums that sit on the
// diagonal are a block reduction (no serial loop).
// W3 (optional): the same coefficients as three bf16 planes [3][Bout][R] for apply_coeffs_x3
__device__ __forceinline__ void build_coeffs_body(int m, int mode, const float* __restrict__ g,
                                                  const float* __restrict__ g2, int Bx, int By, float sc,
                                                  float* __restrict__ Wt, unsigned short* __restrict__ W3 = nullptr) {
    // g: [Bx,By] (LOSS3: gxy [B,B]); g2: LOSS3 only: gyy [B,B]
    __shared__ float red[16];
    const int Bout = (mode == CO_DY) ? By : (mode == CO_LOSS3_DFAKE ? By : Bx);
    const int R = (mode == CO_SAME) ? Bx : Bx + By;
    const float two_sc = 2.f * sc;
    // diagonal sum for this m
    float part = 0.f;
    for (int i = threadIdx.x; i < (Bx > By ? Bx : By); i += blockDim.x) {
        if (mode == CO_LOSS3_DFAKE) {
            if (i < Bx) part += g[(int64_t)i * Bx + m] + g2[(int64_t)m * Bx + i] + g2[(int64_t)i * Bx + m];
        } else if (mode == CO_DX) {
            if (i < By) part += g[(int64_t)m * By + i];
        } else if (mode == CO_DY) {
            if (i < Bx) part += g[(int64_t)i * By + m];
        } else {
            if (i < Bx) part += g[(int64_t)m * Bx + i] + g[(int64_t)i * Bx + m];
        }
    }
    const float d = block_sum(part, red);
    for (int r = threadIdx.x; r < R; r += blockDim.x) {
        float w = 0.f;
        if (mode == CO_LOSS3_DFAKE) {
            const int B = Bx;
            if (r < B) {
                w = -two_sc * g[(int64_t)r * B + m];                       // -2sc gxy[r][m] * x_r
            } else {
                const int rr = r - B;
                w = -two_sc * (g2[(int64_t)m * B + rr] + g2[(int64_t)rr * B + m]);
                if (rr == m) w += two_sc * d;
            }
        } else if (mode == CO_DX) {        // out rows = x rows
            if (r < Bx) w = (r == m) ? two_sc * d : 0.f;
            else w = -two_sc * g[(int64_t)m * By + (r - Bx)];
        } else if (mode == CO_DY) {        // out rows = y rows
            if (r < Bx) w = -two_sc * g[(int64_t)r * By + m];
            else w = (r - Bx == m) ? two_sc * d : 0.f;
        } else {                           // CO_SAME: x is y
            w = -two_sc * (g[(int64_t)m * Bx + r] + g[(int64_t)r * Bx + m]);
            if (r == m) w += two_sc * d;
        }
        Wt[(int64_t)r * Bout + m] = w;
        if (W3) {
            unsigned h, mm, l;
            split3u(w, h, mm, l);
            W3[((int64_t)0 * Bout + m) * R + r] = (unsigned short)(h >> 16);
            W3[((int64_t)1 * Bout + m) * R + r] = (unsigned short)(mm >> 16);
            W3[((int64_t)2 * Bout + m) * R + r] = (unsigned short)(l >> 16);
        }
    }
}

// gscale (optional, one device float): the upstream scalar dLoss/dloss when g holds d loss / d C at dLoss = 1
// (fused solve + sweep): every output of this stage is linear in it, so it rides on scaling_coef.
__global__ __launch_bounds__(256) void build_coeffs(int mode, const float* __restrict__ g, const float* __restrict__ g2,
                                                    int Bx, int By, float sc, float* __restrict__ Wt,
                                                    const float* __restrict__ gscale = nullptr) {
    if (gscale) sc *= gscale[0];
    build_coeffs_body(blockIdx.x, mode, g, g2, Bx, By, sc, Wt);
}

// out[m0+mm][k] = sum_r Wt[r][m0+mm] * Z_r[k];  one column k per thread, MB output rows per block row.
template <int MB>
__global__ __launch_bounds__(256) void apply_coeffs(const float* __restrict__ Wt, int wpitch,
                                                    const float* __restrict__ src1, int n1,
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
            const float w = (m0 + mm < Bout) ? wrow[(int64_t)r * wpitch + mm] : 0.f;
            acc[mm] = fmaf(w, z, acc[mm]);
        }
    }
    for (int r = 0; r < n2; ++r) {
        const float z = kok ? src2[(int64_t)r * K + k] : 0.f;
#pragma unroll
        for (int mm = 0; mm < MB; ++mm) {
            const float w = (m0 + mm < Bout) ? wrow[(int64_t)(n1 + r) * wpitch + mm] : 0.f;
            acc[mm] = fmaf(w, z, acc[mm]);
        }
    }
    if (kok) {
#pragma unroll
        for (int mm = 0; mm < MB; ++mm)
            if (m0 + mm < Bout) out[(int64_t)(m0 + mm) * K + k] = acc[mm];
    }
}

// Feature gradients as small LDS-tiled products  out[a,k] = sc * sum_b G(a,b) X(b,k), k < T*J:
//   CG_H (dh):  a = i, b = j, G = g[i,j],  X(j,k) = M[j,k+J] - M[j,k]          (0 for k >= (T-1)J)
//   CG_M (dM):  a = j, b = i, G = g[i,j],  X(i,k) = h[i,k-J][k>=J] - h[i,k][k<(T-1)J]
// with up to two (g, source) terms summed (h_fake / m_real appear in two cost matrices of the
// loss).  One launch handles up to four outputs (blockIdx.z); block = 16 (a) x 16 (k) outputs.
enum { CG_H = 0, CG_M = 1 };
struct CausalGradJob {
    float* out;          // [Ba, T*J]: rows a_begin .. a_begin+Ba-1 of the full problem
    int mode, Ba, Bb, a_begin;
    int pitch;           // row pitch (= full column count Bj) of the g matrices
    const float* g[2];   // [Bi, Bj] row-major (Bi = rows of C); null = term absent
    const float* src[2]; // CG_H: M [Bb,T,J];  CG_M: h [Bb,T,J]
};
struct CausalGradBatch { CausalGradJob job[4]; int njobs; };

__device__ __forceinline__ void causal_grads_body(const CausalGradBatch& cb, int T, int J, float sc, int bx, int by, int bz) {
    __shared__ float sg[16 * 65], sx[64 * 17];
    const CausalGradJob& jb = cb.job[bz];
    const int TJ = T * J, KK = (T - 1) * J;
    const int a0 = by * 16, k0 = bx * 16;
    if (a0 >= jb.Ba) return;   // block-uniform
    const int t = threadIdx.x, ta = t >> 4, tk = t & 15;
    // g is [Bi,Bj] with row pitch jb.pitch (= Bj): for CG_H `a` indexes its rows and b its columns,
    // for CG_M the reverse
    const int Bj = jb.pitch;
    const int ab = jb.a_begin;
    float tot = 0.f;
    for (int term = 0; term < 2; ++term) {
        const float* g = jb.g[term];
        const float* src = jb.src[term];
        if (!g) continue;
        for (int b0 = 0; b0 < jb.Bb; b0 += 64) {
            // addresses clamped into range, zeros selected afterwards: the eight to twelve loads of a
            // thread carry no control dependence and are all in flight before the first wait
            float gv[4], xa[4], xb[4];
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                const int e = t + 256 * m;
                const int ar = (jb.mode == CG_H) ? (e >> 6) : (e & 15), bg = (jb.mode == CG_H) ? (e & 63) : (e >> 4);
                const int ac = (a0 + ar < jb.Ba) ? a0 + ar : jb.Ba - 1, bc = (b0 + bg < jb.Bb) ? b0 + bg : jb.Bb - 1;
                gv[m] = (jb.mode == CG_H) ? g[(int64_t)(ab + ac) * Bj + bc] : g[(int64_t)bc * Bj + ab + ac];
                const int bb = e >> 4, k = k0 + (e & 15);
                const int bx = (b0 + bb < jb.Bb) ? b0 + bb : jb.Bb - 1;
                const float* row = src + (int64_t)bx * TJ;
                const int kc = k < TJ ? k : TJ - 1;
                // CG_H: row[k+J] - row[k] (k < KK);  CG_M: row[k-J] (k >= J) - row[k] (k < KK)
                const int ka = (jb.mode == CG_H) ? (kc < KK ? kc + J : kc) : (kc >= J ? kc - J : kc);
                xa[m] = row[ka];
                xb[m] = row[kc];
            }
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                const int e = t + 256 * m;
                const int ar = (jb.mode == CG_H) ? (e >> 6) : (e & 15), bg = (jb.mode == CG_H) ? (e & 63) : (e >> 4);
                sg[ar * 65 + bg] = (a0 + ar < jb.Ba && b0 + bg < jb.Bb) ? gv[m] : 0.f;
                const int bb = e >> 4, kk = e & 15, k = k0 + kk;
                float x = 0.f;
                if (b0 + bb < jb.Bb && k < TJ) {
                    if (jb.mode == CG_H) x = (k < KK) ? xa[m] - xb[m] : 0.f;
                    else x = (k >= J ? xa[m] : 0.f) - (k < KK ? xb[m] : 0.f);
                }
                sx[bb * 17 + kk] = x;
            }
            __syncthreads();
#pragma unroll 16
            for (int bb = 0; bb < 64; ++bb) tot = fmaf(sg[ta * 65 + bb], sx[bb * 17 + tk], tot);
            __syncthreads();
        }
    }
    const int aa = a0 + ta, k = k0 + tk;
    if (aa < jb.Ba && k < TJ) jb.out[(int64_t)aa * TJ + k] = tot * sc;
}

__global__ __launch_bounds__(256) void causal_grads(CausalGradBatch cb, int T, int J, float sc, const float* __restrict__ gscale) {
    if (gscale) sc *= gscale[0];
    causal_grads_body(cb, T, J, sc, blockIdx.x, blockIdx.y, blockIdx.z);
}

// The coefficient matrix of the video gradient and the feature gradients both depend on dC alone:
// one launch, the first `nbuild` workgroups build W, the others are the causal_grads grid linearised.
__global__ __launch_bounds__(256) void coeffs_and_causal_grads(int mode, const float* __restrict__ g,
                                                               const float* __restrict__ g2, int Bx, int By, float sc,
                                                               float* __restrict__ Wt, int nbuild, CausalGradBatch cb,
                                                               int T, int J, int gx, int gy, unsigned short* __restrict__ W3,
                                                               const float* __restrict__ gscale) {
    if (gscale) sc *= gscale[0];
    if ((int)blockIdx.x < nbuild) {
        build_coeffs_body(blockIdx.x, mode, g, g2, Bx, By, sc, Wt, W3);
    } else {
        const int lin = blockIdx.x - nbuild;
        causal_grads_body(cb, T, J, sc, lin % gx, (lin / gx) % gy, lin / (gx * gy));
    }
}

static int launch_causal_grads(CausalGradBatch& cb, int T, int J, float sc, hipStream_t st, const float* gscale = nullptr) {
    if (cb.njobs == 0) return 0;
    int maxa = 0;
    for (int i = 0; i < cb.njobs; ++i) if (cb.job[i].Ba > maxa) maxa = cb.job[i].Ba;
    dim3 grid((T * J + 15) / 16, (maxa + 15) / 16, cb.njobs);
    hipLaunchKernelGGL(causal_grads, grid, dim3(256), 0, st, cb, T, J, sc, gscale);
    return launch_status("causal_grads");
}

// ---- MFMA form of apply_coeffs for R = n1 + n2 <= 128 stack rows and Bout <= 64 output rows ----
// out[64 x 64-column tile] = W[64 x 128] * Z[128 x 64]: each of the four waves owns one 32x32
// output sub-tile (row block w&1, column block w>>1) and runs 64 v_mfma_f32_32x32x2_f32 over the
// 128 stack rows.  A operand = W fragments, loaded once per workgroup straight into registers
// (lane l, step s: Wt[2s + (l>>5)][32*mblk + (l&31)] -- two coalesced 128-byte rows);
// B operand = the Z tile staged through LDS (lane l, step s: Z[2s + (l>>5)][32*cblk + (l&31)],
// a conflict-free ds_read_b32).  Workgroups walk the column tiles persistently and prefetch the
// next tile's global loads under the current tile's MFMAs.  HBM traffic = the algorithmic
// minimum: every element of real/fake read once, every element of the gradient written once.
typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int AM_COLS = 64;
constexpr int AM_ROWS = 128;

template <int NSTEPS>
__global__ __launch_bounds__(256) void apply_coeffs_mfma(const float* __restrict__ Wt, int wpitch,
                                                         const float* __restrict__ src1, int n1,
                                                         const float* __restrict__ src2, int n2, int Bout,
                                                         int64_t K, int64_t ntiles, float* __restrict__ out, int accumulate) {
    __shared__ __attribute__((aligned(16))) float zs[AM_ROWS * AM_COLS];
    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int mblk = wave & 1, cblk = wave >> 1;
    const int R = n1 + n2;   // <= 2 * NSTEPS

    // W fragments: afrag[s] = W[m = 32*mblk + (lane&31)][r = 2s + (lane>>5)]
    float afrag[NSTEPS];
    {
        const int m = 32 * mblk + (lane & 31);
#pragma unroll
        for (int s = 0; s < NSTEPS; ++s) {
            const int r = 2 * s + (lane >> 5);
            afrag[s] = (r < R && m < Bout) ? Wt[(int64_t)r * wpitch + m] : 0.f;
        }
    }
    // staging: thread holds float4 at columns c4..c4+3 of stack rows (t>>4) + 16*j, j < 8
    const int c4 = (t & 15) * 4, r0 = t >> 4;
    const float* rowp[8];
    bool rowok[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int r = r0 + 16 * j;
        rowok[j] = r < R;
        rowp[j] = r < n1 ? src1 + (int64_t)r * K : src2 + (int64_t)(r - n1) * K;
    }
    float4 v[8];
    auto load_tile = [&](int64_t tile) {
        const int64_t col = tile * AM_COLS + c4;
#pragma unroll
        for (int j = 0; j < 8; ++j)
            v[j] = (rowok[j] && col + 4 <= K) ? *reinterpret_cast<const float4*>(rowp[j] + col)
                                              : make_float4(0.f, 0.f, 0.f, 0.f);
    };
    int64_t tile = blockIdx.x;
    if (tile < ntiles) load_tile(tile);
    for (; tile < ntiles; tile += gridDim.x) {
#pragma unroll
        for (int j = 0; j < 8; ++j) *reinterpret_cast<float4*>(&zs[(r0 + 16 * j) * AM_COLS + c4]) = v[j];
        __syncthreads();
        if (tile + gridDim.x < ntiles) load_tile(tile + gridDim.x);
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        const float* zb = zs + (lane >> 5) * AM_COLS + 32 * cblk + (lane & 31);
#pragma unroll
        for (int s = 0; s < NSTEPS; ++s)   // branch-free: the LDS reads pipeline ahead of the MFMA chain
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(afrag[s], zb[2 * s * AM_COLS], acc, 0, 0, 0);
        __syncthreads();
        const int64_t col = tile * AM_COLS + 32 * cblk + (lane & 31);
        if (col < K) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = 32 * mblk + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                if (m < Bout) {
                    // blocked use (more than 128 stack rows): later stack chunks add to the earlier ones' result
                    float* o = out + (int64_t)m * K + col;
                    *o = accumulate ? *o + acc[r] : acc[r];
                }
            }
        }
    }
}


// ---- the same product on the bf16 matrix pipe, exactly -------------------------------------------------
// The f32-input MFMA form above is bound by the matrix pipe (64 v_mfma_f32_32x32x2_f32 = 4096 pipe cycles per
// wave and tile).  As in the Gram kernel, every fp32 value of W and of the video tile is cut EXACTLY into three
// bf16 pieces (x = h + m + l) and the product is accumulated as hh + (hm + mh) + (hl + lh + mm) in fp32:
// 48 v_mfma_f32_32x32x16_bf16 = 1536 pipe cycles per wave and tile, the dropped terms below 2^-24 |w z|.
// Eight waves: 0-3 stage (load two adjacent stack rows x four columns, split, pack the row pair of each piece
// into one dword and write it k-contiguous: LDS plane [column][stack row]), 4-7 own one 32x32 output sub-tile
// each, keep their W fragments (8 k-steps x 3 pieces) in registers and read the tile's fragments as two
// ds_read_b64 per piece and step.  Two LDS buffers, one barrier per tile.
typedef __bf16 abf16x8 __attribute__((ext_vector_type(8)));
constexpr int AX_COLP = AM_ROWS * 2 + 8;         // 264 bytes per column of a plane: conflict-free b64 reads, 8-byte aligned
constexpr int AX_PLANE = AM_COLS * AX_COLP;      // 16896
constexpr int AX_BUF = 3 * AX_PLANE;             // 50688

// W3[(piece * Bout + m) * R + r] = piece of Wt[r * wpitch + m]   (bf16 bits; stack rows contiguous per output row)
__global__ __launch_bounds__(256) void split_coeffs(const float* __restrict__ Wt, int wpitch, int R, int Bout,
                                                    unsigned short* __restrict__ W3) {
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= R * Bout) return;
    const int m = e / R, r = e % R;
    unsigned h, mm, l;
    split3u(Wt[(int64_t)r * wpitch + m], h, mm, l);
    W3[((int64_t)0 * Bout + m) * R + r] = (unsigned short)(h >> 16);
    W3[((int64_t)1 * Bout + m) * R + r] = (unsigned short)(mm >> 16);
    W3[((int64_t)2 * Bout + m) * R + r] = (unsigned short)(l >> 16);
}

// out[bo x 64-column tile] (+)= W[bo x rr] * Z[rr x 64]: W3 planes hold ALL output rows / stack rows of the
// problem (Bt x Rt); this launch uses output rows [0, bo) of the block W3 points at and stack rows [r0, r0 + rr).
template <bool ACCUM>
__global__ __launch_bounds__(512) void apply_coeffs_x3(const unsigned short* __restrict__ W3, int Bt, int Rt, int r0,
                                                       const float* __restrict__ src1, int n1,
                                                       const float* __restrict__ src2, int n2, int bo, int64_t K,
                                                       int64_t ntiles, float* __restrict__ out) {
    __shared__ __attribute__((aligned(16))) unsigned char zs[2 * AX_BUF];
    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int rr = n1 + n2;                       // stack rows of this chunk (multiple of 16, <= 128)
    if (wave < 4) {
        // ---------------------------------------------------------------- producers
        const int c4 = (t & 15) * 4, rp0 = t >> 4;            // column group, first row pair
        const float* rowp[8];
        bool rowok[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int r = 2 * (rp0 + 16 * (j >> 1)) + (j & 1);
            rowok[j] = r < rr;
            rowp[j] = r < n1 ? src1 + (int64_t)r * K : src2 + (int64_t)(r - n1) * K;
        }
        // (one tile of loads in flight.  The Gram kernels' remedy -- two sets of unconditional clamped loads, peeled loop --
        // was measured here too: bit-identical, 23.07 us against 22.50 us at configs[1]; this kernel already moves its
        // 94 MB of reads + writes at 4.2 TB/s.  profiles/r02z_prof_ab.txt)
        float4 v[8];
        auto load_tile = [&](int64_t tile) {
            const int64_t col = tile * AM_COLS + c4;
#pragma unroll
            for (int j = 0; j < 8; ++j)
                v[j] = (rowok[j] && col + 4 <= K) ? *reinterpret_cast<const float4*>(rowp[j] + col)
                                                  : make_float4(0.f, 0.f, 0.f, 0.f);
        };
        int64_t tile = blockIdx.x;
        if (tile < ntiles) load_tile(tile);
        int buf = 0;
        for (; tile < ntiles; tile += gridDim.x, buf ^= 1) {
            unsigned char* zb = zs + buf * AX_BUF;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int r = 2 * (rp0 + 16 * i);             // even stack row of the pair
                const float a[4] = {v[2 * i].x, v[2 * i].y, v[2 * i].z, v[2 * i].w};
                const float b[4] = {v[2 * i + 1].x, v[2 * i + 1].y, v[2 * i + 1].z, v[2 * i + 1].w};
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    unsigned ha, ma, la, hb, mb, lb;
                    split3u(a[c], ha, ma, la);
                    split3u(b[c], hb, mb, lb);
                    const int off = (c4 + c) * AX_COLP + r * 2;
                    // dword = bf16(row r) | bf16(row r+1) << 16
                    *reinterpret_cast<unsigned*>(zb + off) = __builtin_amdgcn_perm(hb, ha, 0x07060302u);
                    *reinterpret_cast<unsigned*>(zb + AX_PLANE + off) = __builtin_amdgcn_perm(mb, ma, 0x07060302u);
                    *reinterpret_cast<unsigned*>(zb + 2 * AX_PLANE + off) = __builtin_amdgcn_perm(lb, la, 0x07060302u);
                }
            }
            if (tile + gridDim.x < ntiles) load_tile(tile + gridDim.x);
            __syncthreads();
        }
        __syncthreads();          // the consumers' closing barrier
        return;
    }
    // -------------------------------------------------------------------- consumers
    const int w = wave - 4, mblk = w & 1, cblk = w >> 1;
    const int nsteps = rr >> 4;
    abf16x8 Ah[8], Am[8], Al[8];
    {
        int m = 32 * mblk + (lane & 31);
        if (m >= bo) m = bo - 1;                              // rows past the block: any valid row (their outputs are not stored)
        const unsigned short* wr = W3 + (int64_t)m * Rt + r0 + 8 * (lane >> 5);
        const int64_t plane = (int64_t)Bt * Rt;
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            if (s < nsteps) {
                Ah[s] = *reinterpret_cast<const abf16x8*>(wr + 16 * s);
                Am[s] = *reinterpret_cast<const abf16x8*>(wr + plane + 16 * s);
                Al[s] = *reinterpret_cast<const abf16x8*>(wr + 2 * plane + 16 * s);
            }
        }
    }
    const int boff = (32 * cblk + (lane & 31)) * AX_COLP + 16 * (lane >> 5);
    int buf = 0;
    __syncthreads();                                          // the first tile is staged
    for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x, buf ^= 1) {
        const unsigned char* zb = zs + buf * AX_BUF;
        const int64_t col = tile * AM_COLS + 32 * cblk + (lane & 31);
        // later stack chunks add to the earlier ones' result: the old values are fetched under the MFMAs
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = 32 * mblk + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            acc[r] = (ACCUM && col < K && m < bo) ? out[(int64_t)m * K + col] : 0.f;
        }
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            if (s < nsteps) {
                abf16x8 Bh, Bm, Bl;
                uint2* ph = reinterpret_cast<uint2*>(&Bh);
                uint2* pm = reinterpret_cast<uint2*>(&Bm);
                uint2* pl = reinterpret_cast<uint2*>(&Bl);
                ph[0] = *reinterpret_cast<const uint2*>(zb + boff + 32 * s);
                ph[1] = *reinterpret_cast<const uint2*>(zb + boff + 32 * s + 8);
                pm[0] = *reinterpret_cast<const uint2*>(zb + AX_PLANE + boff + 32 * s);
                pm[1] = *reinterpret_cast<const uint2*>(zb + AX_PLANE + boff + 32 * s + 8);
                pl[0] = *reinterpret_cast<const uint2*>(zb + 2 * AX_PLANE + boff + 32 * s);
                pl[1] = *reinterpret_cast<const uint2*>(zb + 2 * AX_PLANE + boff + 32 * s + 8);
                // smallest terms first
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Am[s], Bm, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Ah[s], Bl, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Al[s], Bh, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Ah[s], Bm, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Am[s], Bh, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Ah[s], Bh, acc, 0, 0, 0);
            }
        }
        if (col < K) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = 32 * mblk + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                if (m < bo) out[(int64_t)m * K + col] = acc[r];
            }
        }
        __syncthreads();                                      // this tile is consumed; the next one is staged
    }
}

// ---- the loss's video gradient at B <= 64 in ONE launch (round 4) ---------------------------------------------------------
// Until round 4 the backward of compute_sinkhorn_loss at configs[1] was two launches: coeffs_and_causal_grads (builds the
// coefficient matrix W [B x 2B] from dC and, in further workgroups, the four feature gradients: 6.9 us, nearly all of it
// launch + memory latency) and apply_coeffs_x3 (dfake = W [X ; Y]: 22.8 us), with a kernel boundary between them.  W is a
// function of 32 KB of dC alone, and every consumer wave of the apply kernel needs exactly ONE row block of it as MFMA
// A-fragments -- which it can form itself while its producers are still waiting for the first stack tile from HBM:
//   lane (m = output row, kh = k-half) loads column m of gxy (its 32 stack rows r < B) and row m / column m of gyy (its 32
//   rows r >= B) straight from L2, the two lanes of a pair (kh = 0 / 1) together hold every term of the diagonal sum
//   d[m] = sum_i gxy[i][m] + gyy[m][i] + gyy[i][m]  (one cross-lane add), and the exact three-way split is done in registers.
// The feature gradients (four small products of dC with the features: 960 wave-sized tasks at configs[1]) are done by the
// PRODUCER waves right behind their first tile's loads, while the consumers form their fragments.  (Second attempt: after the
// last tile, in the shadow of the consumers' last MFMAs, coefficient loads straight from L2: 34.2 us against 22.9 + 6.9; third:
// inside the second tile period: 30.8.)  (First attempt: 16 spare
// workgroups taking (job, row tile) units, 15 k-tiles each in sequence -- each tile is one memory round trip, ~2.5 us: the
// spare workgroups ran 37 us past the apply's 23 and the step went from 194 to 231 us.  profiles/r4_ab_apply_one_launch.jsonl)  dfake bits: only d[m]'s summation order differs from build_coeffs (pair of 32-term
// sums instead of a 256-thread tree); both the fused and the staged loss path take this kernel.
// causal_grads_body's arithmetic for FOUR rows x 16 columns by ONE wave, no workgroup barrier: quarter `qt` (rows a0 + 4 qt ..
// + 3) of the 16 x 16 tile (bx, by) of job bz.  LDS private to the wave: per term a 4 x 65 g tile and a 64 x 17 x tile.  Same
// terms in the same order (term, b, for Bb <= 64) as the 256-thread form: identical bits.
// In two halves so that the caller can put other loads between them: `load` issues every global load of the task (both
// terms, unconditionally, at clamped addresses: straight-line code, so the compiler's wait in front of `finish` counts only
// these loads and leaves younger ones in flight), `finish` does the rest.  Bb <= 64 (one b chunk).
struct WaveTask {
    float gv[2][4], xa[2][16], xb[2][16];
    int k, a0, ta, tk, mode, Ba, Bb;
    bool use_a, use_b, term[2];
    float* out;
};

__device__ __forceinline__ void causal_wave_load(WaveTask& w, const CausalGradBatch& cb, int T, int J, int bx, int by, int bz,
                                                 int qt, int lane) {
    const CausalGradJob& jb = cb.job[bz];
    const int TJ = T * J, KK = (T - 1) * J;
    w.mode = jb.mode; w.Ba = jb.Ba; w.Bb = jb.Bb; w.out = jb.out;
    w.a0 = by * 16 + 4 * qt;
    w.ta = lane >> 4; w.tk = lane & 15;
    const int Bj = jb.pitch, ab = jb.a_begin;
    const bool H = jb.mode == CG_H;
    // this lane's column of the x tile and what it reads of a feature row:  CG_H: row[k+J] - row[k] (k < KK);
    // CG_M: row[k-J] (k >= J) - row[k] (k < KK)
    const int k = bx * 16 + w.tk, kc = k < TJ ? k : TJ - 1;
    const int ka = H ? (kc < KK ? kc + J : kc) : (kc >= J ? kc - J : kc);
    w.k = k;
    w.use_a = H ? (k < KK) : (k >= J && k < TJ);
    w.use_b = k < KK;
#pragma unroll
    for (int term = 0; term < 2; ++term) {
        w.term[term] = jb.g[term] != nullptr;
        const float* g = jb.g[term] ? jb.g[term] : jb.g[0];           // an absent term: valid addresses, values unused
        const float* src = jb.src[term] ? jb.src[term] : jb.src[0];
#pragma unroll
        for (int m = 0; m < 4; ++m) {              // g tile: 4 rows x 64 columns
            const int e = lane + 64 * m;
            const int ar = H ? m : (e & 3), bg = H ? lane : (e >> 2);
            const int ac = (w.a0 + ar < jb.Ba) ? w.a0 + ar : jb.Ba - 1, bc = (bg < jb.Bb) ? bg : jb.Bb - 1;
            w.gv[term][m] = H ? g[(int64_t)(ab + ac) * Bj + bc] : g[(int64_t)bc * Bj + ab + ac];
        }
#pragma unroll
        for (int m = 0; m < 16; ++m) {             // x tile: rows ta + 4 m, this lane's column
            const int bb = w.ta + 4 * m;
            const int br = (bb < jb.Bb) ? bb : jb.Bb - 1;
            const float* row = src + (int64_t)br * TJ;
            w.xa[term][m] = row[ka];
            w.xb[term][m] = row[kc];
        }
    }
}

__device__ __forceinline__ void causal_wave_finish(const WaveTask& w, int TJ, float sc, float* lds, int lane, bool store) {
    const bool H = w.mode == CG_H;
    float* sg0 = lds;                               // per term: 4 x 65 + 4 pad, then 64 x 17
#pragma unroll
    for (int term = 0; term < 2; ++term) {
        float* sg = sg0 + term * (4 * 65 + 4 + 64 * 17);
        float* sx = sg + 4 * 65 + 4;
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            const int e = lane + 64 * m;
            const int ar = H ? m : (e & 3), bg = H ? lane : (e >> 2);
            sg[ar * 65 + bg] = (w.a0 + ar < w.Ba && bg < w.Bb) ? w.gv[term][m] : 0.f;
        }
#pragma unroll
        for (int m = 0; m < 16; ++m) {
            const int bb = w.ta + 4 * m;
            const float xh = w.use_a ? w.xa[term][m] - w.xb[term][m] : 0.f;
            const float xm = (w.use_a ? w.xa[term][m] : 0.f) - (w.use_b ? w.xb[term][m] : 0.f);
            sx[bb * 17 + w.tk] = (bb < w.Bb) ? (H ? xh : xm) : 0.f;
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    float tot = 0.f;
#pragma unroll
    for (int term = 0; term < 2; ++term) {
        const float* sg = sg0 + term * (4 * 65 + 4 + 64 * 17);
        const float* sx = sg + 4 * 65 + 4;
        if (w.term[term]) {                         // wave-uniform
#pragma unroll 16
            for (int bb = 0; bb < 64; ++bb) tot = fmaf(sg[w.ta * 65 + bb], sx[bb * 17 + w.tk], tot);
        }
    }
    const int aa = w.a0 + w.ta;
    if (store && aa < w.Ba && w.k < TJ) w.out[(int64_t)aa * TJ + w.k] = tot * sc;
}

struct Loss3Apply {
    const float* gxy;        // [B,B] d loss / d C_xy
    const float* gyy;        // [B,B] d loss / d C_yy
    const float* gscale;     // optional upstream scalar (one device float)
    const float* real;       // [B,K]
    const float* fake;       // [B,K]
    float* out;              // dfake [B,K]
    int B;
    float sc;
    int64_t K, ntiles;
    int nmain;               // = gridDim.x
    CausalGradBatch cg;
    int T, J, gx, gy;        // causal grid: gx k-tiles x gy row tiles x cg.njobs
};

constexpr int L3_TASK_FLOATS = 2 * (4 * 65 + 4 + 64 * 17);   // LDS of one wave task: per term a g tile, pad, an x tile
constexpr int L3_DCP = 65;                                // row pitch of the dC copies (conflict-free row AND column reads)

// B64: the batch is exactly 64 (configs[1]): which stack rows are real / fake is known at compile time; TASKS: feature gradients wanted
template <bool B64, bool TASKS>
__global__ __launch_bounds__(512) void apply_coeffs_x3_loss3(Loss3Apply a) {
    __shared__ __attribute__((aligned(16))) unsigned char zs[2 * AX_BUF];
    __shared__ float task_lds[4 * L3_TASK_FLOATS];        // 43 KB: one task per producer wave
    float* dcs = reinterpret_cast<float*>(zs + AX_BUF);   // gxy, gyy for the consumers' coefficient build: 33 KB of stage
                                                          // buffer 1, which the producers first write in their SECOND iteration,
                                                          // behind the barrier that ends the consumers' build
    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    float sc = a.sc;
    if (a.gscale) sc *= a.gscale[0];
    const int B = B64 ? 64 : a.B, rr = 2 * B;                 // stack rows: B of real, B of fake (a multiple of 16, <= 128)
    const int64_t K = a.K, ntiles = a.ntiles;
    if (wave < 4) {
        // ---------------------------------------------------------------- producers (as apply_coeffs_x3)
        const int c4 = (t & 15) * 4, rp0 = t >> 4;            // column group, first row pair
        const float* rowp[8];
        bool rowok[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int r = 2 * (rp0 + 16 * (j >> 1)) + (j & 1);
            rowok[j] = r < rr;
            rowp[j] = r < B ? a.real + (int64_t)r * K : a.fake + (int64_t)((r < rr ? r : rr - 1) - B) * K;
        }
        float4 v[8];
        auto load_tile = [&](int64_t tile) {
            const int64_t col = tile * AM_COLS + c4;
#pragma unroll
            for (int j = 0; j < 8; ++j)
                v[j] = (rowok[j] && col + 4 <= K) ? *reinterpret_cast<const float4*>(rowp[j] + col)
                                                  : make_float4(0.f, 0.f, 0.f, 0.f);
        };
        // The feature gradients: one wave task = four rows x 16 columns of one job (causal_grads_wave); 4 njobs gy gx tasks over
        // the 4 nmain producer waves of the launch (960 over 960 at configs[1]).
        // One task per producer wave at configs[1] (960 tasks, 960 waves).  Its loads go out FIRST, the first tile's behind
        // them: the task's operands are back from L2 after ~1 us and its arithmetic runs inside the ~2.5 us the first tile
        // needs to arrive from cold HBM (and the consumer waves need to form their fragments) -- nobody waits for it.
        // (Issued behind the tile's loads instead, the in-order wait made the task start only when the tile had landed:
        // +3.6 us on the launch.  Inside the second tile period: +3.5.  After the last tile: +11.  16 spare workgroups: +37.)
        const int total = 4 * a.gx * a.gy * a.cg.njobs;       // TASKS: >= 4
        const int TJ = a.T * a.J;
        int task = blockIdx.x * 4 + wave;
        WaveTask wt;
        if (TASKS) {
            const int tc = task < total ? task : total - 1;   // clamped: the loads are unconditional, the store is not
            const int rest = tc >> 2;
            causal_wave_load(wt, a.cg, a.T, a.J, rest % a.gx, (rest / a.gx) % a.gy, rest / (a.gx * a.gy), tc & 3, lane);
        }
        int64_t tile = blockIdx.x;
        if (tile < ntiles) load_tile(tile);
        __syncthreads();                                      // (the consumers' barrier behind their copy of dC to LDS)
        if (TASKS) {
            causal_wave_finish(wt, TJ, sc, task_lds + wave * L3_TASK_FLOATS, lane, task < total);
            for (task += a.nmain * 4; task < total; task += a.nmain * 4) {  // more tasks than waves (small K): the rest, plainly
                const int rest = task >> 2;
                causal_wave_load(wt, a.cg, a.T, a.J, rest % a.gx, (rest / a.gx) % a.gy, rest / (a.gx * a.gy), task & 3, lane);
                causal_wave_finish(wt, TJ, sc, task_lds + wave * L3_TASK_FLOATS, lane, true);
            }
        }
        int buf = 0;
        for (; tile < ntiles; tile += a.nmain, buf ^= 1) {
            unsigned char* zb = zs + buf * AX_BUF;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int r = 2 * (rp0 + 16 * i);             // even stack row of the pair
                const float x[4] = {v[2 * i].x, v[2 * i].y, v[2 * i].z, v[2 * i].w};
                const float y[4] = {v[2 * i + 1].x, v[2 * i + 1].y, v[2 * i + 1].z, v[2 * i + 1].w};
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    unsigned ha, ma, la, hb, mb, lb;
                    split3u(x[c], ha, ma, la);
                    split3u(y[c], hb, mb, lb);
                    const int off = (c4 + c) * AX_COLP + r * 2;
                    *reinterpret_cast<unsigned*>(zb + off) = __builtin_amdgcn_perm(hb, ha, 0x07060302u);
                    *reinterpret_cast<unsigned*>(zb + AX_PLANE + off) = __builtin_amdgcn_perm(mb, ma, 0x07060302u);
                    *reinterpret_cast<unsigned*>(zb + 2 * AX_PLANE + off) = __builtin_amdgcn_perm(lb, la, 0x07060302u);
                }
            }
            if (tile + a.nmain < ntiles) load_tile(tile + a.nmain);
            __syncthreads();
        }
        __syncthreads();          // the consumers' closing barrier
        return;
    }
    // -------------------------------------------------------------------- consumers: their W fragments first
    const int w = wave - 4, mblk = w & 1, cblk = w >> 1;
    const int nsteps = rr >> 4;
    abf16x8 Ah[8], Am[8], Al[8];
    {
        int m = 32 * mblk + (lane & 31);
        if (m >= B) m = B - 1;                                // rows past the batch: any valid row (their outputs are not stored)
        const int kh = lane >> 5;
        const float two_sc = 2.f * sc;
        float wv[8][8];
        float part = 0.f;
        // gxy and gyy (2 x 16 KB) into LDS with coalesced 16-byte loads by the 256 consumer threads, rows at pitch 65: the
        // fragments need row m AND column m of gyy -- straight from L2 the row reads are 64 cache lines per instruction
        // (2048 line requests per wave on the CU's one vector-memory path: ~4 us)
        {
            const int ct = t - 256;
            const int nvec = B * B / 4;                       // B % 8 == 0 (host)
            for (int e = ct; e < 2 * nvec; e += 256) {
                const int which = e >= nvec, f = (e - which * nvec) * 4;
                const float4 q4 = *reinterpret_cast<const float4*>((which ? a.gyy : a.gxy) + f);
                float* d0 = dcs + which * 64 * L3_DCP + (f / B) * L3_DCP + (f % B);
                d0[0] = q4.x; d0[1] = q4.y; d0[2] = q4.z; d0[3] = q4.w;
            }
        }
        __syncthreads();                                      // (the producers pass this one right behind their first loads)
        const float* lxy = dcs;
        const float* lyy = dcs + 64 * L3_DCP;
#pragma unroll
        for (int s = 0; s < 8; ++s)
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int r = 16 * s + 8 * kh + j;
                if (B64) {                                    // s < 4: a row of real, else a row of fake -- no selects
                    if (s < 4) wv[s][j] = lxy[r * L3_DCP + m];
                    else wv[s][j] = lyy[m * L3_DCP + (r - 64)] + lyy[(r - 64) * L3_DCP + m];
                } else {
                    const int rc = r < rr ? r : rr - 1;
                    const int q = rc >= B ? rc - B : 0, rx = rc < B ? rc : 0;
                    const float vx = lxy[rx * L3_DCP + m], vy = lyy[m * L3_DCP + q] + lyy[q * L3_DCP + m];
                    wv[s][j] = r >= rr ? 0.f : (rc < B ? vx : vy);
                }
            }
#pragma unroll
        for (int s = 0; s < 8; ++s)
#pragma unroll
            for (int j = 0; j < 8; ++j) part += wv[s][j];
        const float d = part + __shfl_xor(part, 32, 64);      // the pair (kh = 0, 1) holds every term of the diagonal sum
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            unsigned hh[8], mm[8], ll[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int r = 16 * s + 8 * kh + j;
                float c = -two_sc * wv[s][j];                 // -2 sc gxy[r][m]  resp.  -2 sc (gyy[m][q] + gyy[q][m])
                if (r == B + m) c += two_sc * d;
                split3u(c, hh[j], mm[j], ll[j]);
            }
            unsigned ph[4], pm[4], pl[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                ph[j] = __builtin_amdgcn_perm(hh[2 * j + 1], hh[2 * j], 0x07060302u);
                pm[j] = __builtin_amdgcn_perm(mm[2 * j + 1], mm[2 * j], 0x07060302u);
                pl[j] = __builtin_amdgcn_perm(ll[2 * j + 1], ll[2 * j], 0x07060302u);
            }
            const uint4 uh = {ph[0], ph[1], ph[2], ph[3]}, um = {pm[0], pm[1], pm[2], pm[3]}, ul = {pl[0], pl[1], pl[2], pl[3]};
            Ah[s] = *reinterpret_cast<const abf16x8*>(&uh);
            Am[s] = *reinterpret_cast<const abf16x8*>(&um);
            Al[s] = *reinterpret_cast<const abf16x8*>(&ul);
        }
    }
    const int boff = (32 * cblk + (lane & 31)) * AX_COLP + 16 * (lane >> 5);
    int buf = 0;
    __syncthreads();                                          // the first tile is staged
    for (int64_t tile = blockIdx.x; tile < ntiles; tile += a.nmain, buf ^= 1) {
        const unsigned char* zb = zs + buf * AX_BUF;
        const int64_t col = tile * AM_COLS + 32 * cblk + (lane & 31);
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            if (s < nsteps) {
                abf16x8 Bh, Bm, Bl;
                uint2* ph = reinterpret_cast<uint2*>(&Bh);
                uint2* pm = reinterpret_cast<uint2*>(&Bm);
                uint2* pl = reinterpret_cast<uint2*>(&Bl);
                ph[0] = *reinterpret_cast<const uint2*>(zb + boff + 32 * s);
                ph[1] = *reinterpret_cast<const uint2*>(zb + boff + 32 * s + 8);
                pm[0] = *reinterpret_cast<const uint2*>(zb + AX_PLANE + boff + 32 * s);
                pm[1] = *reinterpret_cast<const uint2*>(zb + AX_PLANE + boff + 32 * s + 8);
                pl[0] = *reinterpret_cast<const uint2*>(zb + 2 * AX_PLANE + boff + 32 * s);
                pl[1] = *reinterpret_cast<const uint2*>(zb + 2 * AX_PLANE + boff + 32 * s + 8);
                // smallest terms first
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Am[s], Bm, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Ah[s], Bl, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Al[s], Bh, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Ah[s], Bm, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Am[s], Bh, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Ah[s], Bh, acc, 0, 0, 0);
            }
        }
        if (col < K) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = 32 * mblk + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                if (m < B) a.out[(int64_t)m * K + col] = acc[r];
            }
        }
        __syncthreads();                                      // this tile is consumed; the next one is staged
    }
}

// ---- large batches: 256 output rows per workgroup, the whole stack in one pass -------------------------------
// apply_coeffs_x3 produces a 64-row output block from a 128-row stack chunk per launch; at B = 512 that is 8 x 8
// launches, every stack chunk is fetched, split and staged again for each of the 8 row blocks, and the output is
// read-modified-written 7 times (measured: 37 of the 64 ms of a configs[4] loss).  Here a workgroup owns 256 output
// rows x 64 columns and walks ALL R = 2B stack rows in 128-row chunks: a staged chunk (exact 3-way split, same LDS
// planes) now feeds 4 x as many MFMAs, the accumulators stay in registers (2 x 2 tiles per consumer wave), the
// output is written once.  W does not fit registers any more (256 x R x 3 planes): the consumers stream their
// fragments from L2 (the bf16 planes of W are <= 3 MB), three k-steps ahead of use.
constexpr int AB_MT = 256;                        // the tallest output tile

// W3 [3][Bt][Rt] (stack rows contiguous per output row) -> fragment-major tiles for apply_coeffs_x3_m256:
//   Wt3[((pl * Bt/32 + m/32) * Rt/16 + r/16) * 512 + ((m % 32) + 32 * ((r % 16) / 8)) * 8 + r % 8]
// i.e. the 64 x 16-byte MFMA A-fragment of (row tile, k-step) is ONE contiguous KiB.  In the row-major layout a
// fragment load touches 64 different 128-byte lines and uses 32 bytes of each (measured: the W stream from L2 bound
// the kernel -- 0.92 ms at B = 256 against 0.57 ms with the loads removed); tiled, every line is used whole.
__global__ __launch_bounds__(256) void retile_coeffs(const unsigned short* __restrict__ W3, int Bt, int Rt,
                                                     unsigned short* __restrict__ Wt3) {
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;      // one 16-byte group (8 consecutive r) per thread
    const int64_t ngroups = (int64_t)3 * Bt * (Rt / 8);
    if (e >= ngroups) return;
    const int rg = (int)(e % (Rt / 8));
    const int m = (int)((e / (Rt / 8)) % Bt);
    const int pl = (int)(e / ((int64_t)(Rt / 8) * Bt));
    const int r = rg * 8;
    const uint4 v = *reinterpret_cast<const uint4*>(W3 + ((int64_t)pl * Bt + m) * Rt + r);
    const int64_t dst = ((((int64_t)pl * (Bt / 32) + m / 32) * (Rt / 16) + r / 16) * 64 + (m % 32) + 32 * ((r % 16) / 8)) * 8;
    *reinterpret_cast<uint4*>(Wt3 + dst) = v;
}

__global__ __launch_bounds__(512) void apply_coeffs_x3_m256(const unsigned short* __restrict__ W3, int Bt, int Rt,
                                                            const float* __restrict__ src1, int n1,
                                                            const float* __restrict__ src2, int n2, int64_t K,
                                                            int64_t ntiles, float* __restrict__ out) {
    __shared__ __attribute__((aligned(16))) unsigned char zs[2 * AX_BUF];
    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int nchunk = (n1 + n2) / AM_ROWS;                   // 128-row chunks; n1 % 128 == 0 (host)
    const int m0 = blockIdx.y * AB_MT;
    if (wave < 4) {
        // ---------------------------------------------------------------- producers (as apply_coeffs_x3)
        const int c4 = (t & 15) * 4, rp0 = t >> 4;
        // TWO chunks of loads in flight per producer thread (v: even stages, w: odd stages, each set re-issued right after
        // it has been split): a chunk is 192 MFMAs per consumer wave = 2.6 us at the matrix pipe's full rate, less than the
        // HBM round trip under load -- with one chunk in flight the consumers waited for the producers at every barrier
        // (the diagnostic build with NOTHING but MFMAs, barriers and these loads ran at 49 % of the 2.48 PFLOP/s that
        // tools/micro/mfma_peak.hip sustains).
        float4 v[8], w[8];
        auto load_stage = [&](float4 (&dst)[8], int64_t tile, int c) {
            const int g0 = c * AM_ROWS;
            const float* base = g0 < n1 ? src1 + (int64_t)g0 * K : src2 + (int64_t)(g0 - n1) * K;
            const int64_t col = tile * AM_COLS + c4;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int r = 2 * (rp0 + 16 * (j >> 1)) + (j & 1);
                dst[j] = (col + 4 <= K) ? *reinterpret_cast<const float4*>(base + (int64_t)r * K + col)
                                        : make_float4(0.f, 0.f, 0.f, 0.f);
            }
        };
        auto split_stage = [&](const float4 (&src)[8], unsigned char* zb) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int r = 2 * (rp0 + 16 * i);
                const float a[4] = {src[2 * i].x, src[2 * i].y, src[2 * i].z, src[2 * i].w};
                const float b[4] = {src[2 * i + 1].x, src[2 * i + 1].y, src[2 * i + 1].z, src[2 * i + 1].w};
#pragma unroll
                for (int cc = 0; cc < 4; ++cc) {
                    unsigned ha, ma, la, hb, mb, lb;
                    split3u(a[cc], ha, ma, la);
                    split3u(b[cc], hb, mb, lb);
                    const int off = (c4 + cc) * AX_COLP + r * 2;
                    *reinterpret_cast<unsigned*>(zb + off) = __builtin_amdgcn_perm(hb, ha, 0x07060302u);
                    *reinterpret_cast<unsigned*>(zb + AX_PLANE + off) = __builtin_amdgcn_perm(mb, ma, 0x07060302u);
                    *reinterpret_cast<unsigned*>(zb + 2 * AX_PLANE + off) = __builtin_amdgcn_perm(lb, la, 0x07060302u);
                }
            }
        };
        // stage sequence: (tile, chunk) pairs in order; `nt`/`nc` walk two stages ahead of the one being split
        int64_t tile = blockIdx.x, nt = tile;
        int c = 0, nc = 0, buf = 0;
        auto advance = [&](int64_t& tt, int& cc) { if (++cc == nchunk) { cc = 0; tt += gridDim.x; } };
        if (nt < ntiles) load_stage(v, nt, nc);
        advance(nt, nc);
        if (nt < ntiles) load_stage(w, nt, nc);
        advance(nt, nc);
        while (tile < ntiles) {
            split_stage(v, zs + buf * AX_BUF);
            if (nt < ntiles) load_stage(v, nt, nc);
            advance(nt, nc);
            advance(tile, c);
            buf ^= 1;
            __syncthreads();
            if (tile >= ntiles) break;
            split_stage(w, zs + buf * AX_BUF);
            if (nt < ntiles) load_stage(w, nt, nc);
            advance(nt, nc);
            advance(tile, c);
            buf ^= 1;
            __syncthreads();
        }
        __syncthreads();          // the consumers' closing barrier
        return;
    }
    // -------------------------------------------------------------------- consumers
    const int w = wave - 4;                                   // rows m0 + 64 w .. + 63, all 64 columns
    const int nsteps = Rt >> 4;                               // k-steps per tile (a multiple of 8)
    // W3 is the fragment-major copy (retile_coeffs): row tile mt, k-step g -> 1 KiB at ((pl*Bt/32 + mt)*nsteps + g)*512
    const int64_t plane = (int64_t)Bt * Rt;
    const unsigned short* wb0 = W3 + (int64_t)((m0 + 64 * w) / 32) * nsteps * 512 + lane * 8;
    const unsigned short* wb1 = wb0 + (int64_t)nsteps * 512;
    abf16x8 A[4][2][3];                                       // ring of fragment sets: step g lives in A[g & 3]
    auto ldA = [&](int g, int slot) {
        const int64_t o = (int64_t)512 * g;
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) {
            A[slot][0][pl] = *reinterpret_cast<const abf16x8*>(wb0 + pl * plane + o);
            A[slot][1][pl] = *reinterpret_cast<const abf16x8*>(wb1 + pl * plane + o);
        }
    };
    ldA(0, 0); ldA(1, 1); ldA(2, 2);
    const int boff0 = (lane & 31) * AX_COLP + 16 * (lane >> 5), boff1 = boff0 + 32 * AX_COLP;
    int buf = 0;
    __syncthreads();                                          // the first stage is in buffer 0
    for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        f32x16 acc00, acc01, acc10, acc11;
#pragma unroll
        for (int r = 0; r < 16; ++r) { acc00[r] = 0.f; acc01[r] = 0.f; acc10[r] = 0.f; acc11[r] = 0.f; }
        for (int c = 0; c < nchunk; ++c, buf ^= 1) {
            const unsigned char* zb = zs + buf * AX_BUF;
            abf16x8 Bf[2][2][3];                              // [step parity][column tile][piece]: one k-step ahead
            auto ldB = [&](int st, int par) {
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) {
                    uint2* p0 = reinterpret_cast<uint2*>(&Bf[par][0][pl]);
                    uint2* p1 = reinterpret_cast<uint2*>(&Bf[par][1][pl]);
                    p0[0] = *reinterpret_cast<const uint2*>(zb + pl * AX_PLANE + boff0 + 32 * st);
                    p0[1] = *reinterpret_cast<const uint2*>(zb + pl * AX_PLANE + boff0 + 32 * st + 8);
                    p1[0] = *reinterpret_cast<const uint2*>(zb + pl * AX_PLANE + boff1 + 32 * st);
                    p1[1] = *reinterpret_cast<const uint2*>(zb + pl * AX_PLANE + boff1 + 32 * st + 8);
                }
            };
            ldB(0, 0);
#pragma unroll
            for (int s = 0; s < 8; ++s) {
                // W fragments three k-steps ahead (past the end of this tile's walk: the first steps of the next tile),
                // the staged tile's fragments one k-step ahead; the scheduling barrier keeps hipcc from sinking the loads
                // back down to their first use (it did: L2 latency landed on every MFMA group)
                int gn = c * 8 + s + 3;
                if (gn >= nsteps) gn -= nsteps;
                ldA(gn, (s + 3) & 3);
                if (s < 7) ldB(s + 1, (s + 1) & 1);
                __builtin_amdgcn_sched_barrier(0);
                // product-major order over the four accumulators (no MFMA waits on the one before it), smallest terms first
#define KCCOT_A4(PA, PB)                                                                                              \
                acc00 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[s & 3][0][PA], Bf[s & 1][0][PB], acc00, 0, 0, 0); \
                acc01 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[s & 3][0][PA], Bf[s & 1][1][PB], acc01, 0, 0, 0); \
                acc10 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[s & 3][1][PA], Bf[s & 1][0][PB], acc10, 0, 0, 0); \
                acc11 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[s & 3][1][PA], Bf[s & 1][1][PB], acc11, 0, 0, 0);
                KCCOT_A4(1, 1) KCCOT_A4(0, 2) KCCOT_A4(2, 0) KCCOT_A4(0, 1) KCCOT_A4(1, 0) KCCOT_A4(0, 0)
#undef KCCOT_A4
                __builtin_amdgcn_sched_barrier(0);
            }
            __syncthreads();                                  // this stage is consumed; the next one is staged
        }
        const int64_t col = tile * AM_COLS + (lane & 31);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int64_t m = m0 + 64 * w + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            if (col < K) { out[m * K + col] = acc00[r]; out[(m + 32) * K + col] = acc10[r]; }
            if (col + 32 < K) { out[m * K + col + 32] = acc01[r]; out[(m + 32) * K + col + 32] = acc11[r]; }
        }
    }
}

// ---- the same one-launch form for output blocks that are not multiples of 256 rows (round 3) -----------------------------
// The output tile of a workgroup is (32 RT WR) rows x 64 columns: consumer wave (wr, wc) of a WR x WC grid owns RT x CT
// MFMA tiles, CT WC = 2.  <1, 2, 4, 1> = 128 rows (configs[2]: B = 128 ran 2 x 2 launches of the 64-row block form, every
// stack chunk staged twice and the output read-modified-written: 0.69 of the 1.02 ms of its loss step), <1, 1, 2, 2> = 64
// rows and <1, 1, 1, 2> = 32 rows (two consumer waves; a rank's rows of the batch-sharded loss: one launch instead of
// 2B/128 accumulating ones).  Producers and stage pipeline are those of apply_coeffs_x3_m256, which stays as measured.
template <int RT, int CT, int WR, int WC>
__global__ __launch_bounds__(512) void apply_coeffs_x3_rows(const unsigned short* __restrict__ W3, int Bt, int Rt,
                                                            const float* __restrict__ src1, int n1,
                                                            const float* __restrict__ src2, int n2, int64_t K,
                                                            int64_t ntiles, float* __restrict__ out) {
    static_assert(CT * WC == 2 && WR * WC <= 4, "a workgroup tile is 64 columns wide and has four consumer waves");
    constexpr int MT = 32 * RT * WR;
    __shared__ __attribute__((aligned(16))) unsigned char zs[2 * AX_BUF];
    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int nchunk = (n1 + n2) / AM_ROWS;                   // 128-row chunks; n1 % 128 == 0 (host)
    const int m0 = blockIdx.y * MT;
    if (wave < 4) {
        // ---------------------------------------------------------------- producers (as apply_coeffs_x3)
        const int c4 = (t & 15) * 4, rp0 = t >> 4;
        // TWO chunks of loads in flight per producer thread (v: even stages, w: odd stages, each set re-issued right after
        // it has been split): a chunk is 192 MFMAs per consumer wave = 2.6 us at the matrix pipe's full rate, less than the
        // HBM round trip under load -- with one chunk in flight the consumers waited for the producers at every barrier
        // (the diagnostic build with NOTHING but MFMAs, barriers and these loads ran at 49 % of the 2.48 PFLOP/s that
        // tools/micro/mfma_peak.hip sustains).
        float4 v[8], w[8];
        auto load_stage = [&](float4 (&dst)[8], int64_t tile, int c) {
            const int g0 = c * AM_ROWS;
            const float* base = g0 < n1 ? src1 + (int64_t)g0 * K : src2 + (int64_t)(g0 - n1) * K;
            const int64_t col = tile * AM_COLS + c4;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int r = 2 * (rp0 + 16 * (j >> 1)) + (j & 1);
                dst[j] = (col + 4 <= K) ? *reinterpret_cast<const float4*>(base + (int64_t)r * K + col)
                                        : make_float4(0.f, 0.f, 0.f, 0.f);
            }
        };
        auto split_stage = [&](const float4 (&src)[8], unsigned char* zb) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int r = 2 * (rp0 + 16 * i);
                const float a[4] = {src[2 * i].x, src[2 * i].y, src[2 * i].z, src[2 * i].w};
                const float b[4] = {src[2 * i + 1].x, src[2 * i + 1].y, src[2 * i + 1].z, src[2 * i + 1].w};
#pragma unroll
                for (int cc = 0; cc < 4; ++cc) {
                    unsigned ha, ma, la, hb, mb, lb;
                    split3u(a[cc], ha, ma, la);
                    split3u(b[cc], hb, mb, lb);
                    const int off = (c4 + cc) * AX_COLP + r * 2;
                    *reinterpret_cast<unsigned*>(zb + off) = __builtin_amdgcn_perm(hb, ha, 0x07060302u);
                    *reinterpret_cast<unsigned*>(zb + AX_PLANE + off) = __builtin_amdgcn_perm(mb, ma, 0x07060302u);
                    *reinterpret_cast<unsigned*>(zb + 2 * AX_PLANE + off) = __builtin_amdgcn_perm(lb, la, 0x07060302u);
                }
            }
        };
        // stage sequence: (tile, chunk) pairs in order; `nt`/`nc` walk two stages ahead of the one being split
        int64_t tile = blockIdx.x, nt = tile;
        int c = 0, nc = 0, buf = 0;
        auto advance = [&](int64_t& tt, int& cc) { if (++cc == nchunk) { cc = 0; tt += gridDim.x; } };
        if (nt < ntiles) load_stage(v, nt, nc);
        advance(nt, nc);
        if (nt < ntiles) load_stage(w, nt, nc);
        advance(nt, nc);
        while (tile < ntiles) {
            split_stage(v, zs + buf * AX_BUF);
            if (nt < ntiles) load_stage(v, nt, nc);
            advance(nt, nc);
            advance(tile, c);
            buf ^= 1;
            __syncthreads();
            if (tile >= ntiles) break;
            split_stage(w, zs + buf * AX_BUF);
            if (nt < ntiles) load_stage(w, nt, nc);
            advance(nt, nc);
            advance(tile, c);
            buf ^= 1;
            __syncthreads();
        }
        __syncthreads();          // the consumers' closing barrier
        return;
    }
    // -------------------------------------------------------------------- consumers
    const int w = wave - 4;
    const bool active = WR * WC == 4 || w < WR * WC;           // (32- and 64-row tiles leave consumer waves without a tile:
    const int wr = active ? w / WC : 0, wc = active ? w % WC : 0;   //  they only keep the barriers)
    const int mw = m0 + 32 * RT * wr;                         // first output row of this wave
    const int nsteps = Rt >> 4;                               // k-steps per tile (a multiple of 8)
    // W3 is the fragment-major copy (retile_coeffs): row tile mt, k-step g -> 1 KiB at ((pl*Bt/32 + mt)*nsteps + g)*512
    const int64_t plane = (int64_t)Bt * Rt;
    const unsigned short* wb0 = W3 + (int64_t)(mw / 32) * nsteps * 512 + lane * 8;
    abf16x8 A[4][RT][3];                                      // ring of fragment sets: step g lives in A[g & 3]
    auto ldA = [&](int g, int slot) {
        const int64_t o = (int64_t)512 * g;
#pragma unroll
        for (int pl = 0; pl < 3; ++pl)
#pragma unroll
            for (int i = 0; i < RT; ++i)
                A[slot][i][pl] = *reinterpret_cast<const abf16x8*>(wb0 + (int64_t)i * nsteps * 512 + pl * plane + o);
    };
    if (active) { ldA(0, 0); ldA(1, 1); ldA(2, 2); }
    const int boff0 = ((lane & 31) + 32 * CT * wc) * AX_COLP + 16 * (lane >> 5);
    int buf = 0;
    __syncthreads();                                          // the first stage is in buffer 0
    for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        f32x16 acc[RT][CT];
#pragma unroll
        for (int i = 0; i < RT; ++i)
#pragma unroll
            for (int j = 0; j < CT; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
        for (int c = 0; c < nchunk; ++c, buf ^= 1) {
            if (active) {
                const unsigned char* zb = zs + buf * AX_BUF;
                abf16x8 Bf[2][CT][3];                         // [step parity][column tile][piece]: one k-step ahead
                auto ldB = [&](int st, int par) {
#pragma unroll
                    for (int pl = 0; pl < 3; ++pl)
#pragma unroll
                        for (int j = 0; j < CT; ++j) {
                            uint2* p0 = reinterpret_cast<uint2*>(&Bf[par][j][pl]);
                            p0[0] = *reinterpret_cast<const uint2*>(zb + pl * AX_PLANE + boff0 + j * 32 * AX_COLP + 32 * st);
                            p0[1] = *reinterpret_cast<const uint2*>(zb + pl * AX_PLANE + boff0 + j * 32 * AX_COLP + 32 * st + 8);
                        }
                };
                ldB(0, 0);
#pragma unroll
                for (int s = 0; s < 8; ++s) {
                    // W fragments three k-steps ahead (past the end of this tile's walk: the first steps of the next tile),
                    // the staged tile's fragments one k-step ahead; the scheduling barrier keeps hipcc from sinking the loads
                    // back down to their first use (it did: L2 latency landed on every MFMA group)
                    int gn = c * 8 + s + 3;
                    if (gn >= nsteps) gn -= nsteps;
                    ldA(gn, (s + 3) & 3);
                    if (s < 7) ldB(s + 1, (s + 1) & 1);
                    __builtin_amdgcn_sched_barrier(0);
                    // product-major order over the accumulators (no MFMA waits on the one before it), smallest terms first
#define KCCOT_A4(PA, PB)                                                                                              \
                    _Pragma("unroll") for (int i = 0; i < RT; ++i)                                                    \
                        _Pragma("unroll") for (int j = 0; j < CT; ++j)                                                \
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[s & 3][i][PA], Bf[s & 1][j][PB], acc[i][j], 0, 0, 0);
                    KCCOT_A4(1, 1) KCCOT_A4(0, 2) KCCOT_A4(2, 0) KCCOT_A4(0, 1) KCCOT_A4(1, 0) KCCOT_A4(0, 0)
#undef KCCOT_A4
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            __syncthreads();                                  // this stage is consumed; the next one is staged
        }
        if (active) {
            const int64_t col = tile * AM_COLS + 32 * CT * wc + (lane & 31);
#pragma unroll
            for (int i = 0; i < RT; ++i)
#pragma unroll
                for (int j = 0; j < CT; ++j)
                    if (col + 32 * j < K) {
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            const int64_t m = mw + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                            out[m * K + col + 32 * j] = acc[i][j][r];
                        }
                    }
        }
    }
}

// ---- 256 rows x 128 columns per workgroup: half the W bytes per MFMA ------------------------------------------------
// tools/micro/mfma_feed.hip (profiles/r02r_micro_mfma_feed.txt): the consumer loop of apply_coeffs_x3_m256 sustains
// 2.3-2.5 PFLOP/s chip-wide with constant operands, with its LDS fragment reads and with its barriers -- and 1.19 PFLOP/s
// as soon as every k-step loads its six W fragments from (L2-resident) global memory: 24 KB per 96 MFMAs per CU is
// ~19 TB/s at the matrix pipe's full rate, about twice what the L2s deliver.  Here a consumer wave owns 64 rows x 128
// columns (2 x 4 MFMA tiles, 128 accumulator registers): the same six W fragments feed 48 MFMAs.  Stack chunks of 64 rows
// (two 51-KB LDS buffers), staged fragments read per pair of column tiles, one pair ahead.
constexpr int AY_ROWS = 64;                       // stack rows per staged chunk
constexpr int AY_COLS = 128;                      // columns per workgroup tile
constexpr int AY_COLP = AY_ROWS * 2 + 8;          // 136 bytes per column of a plane (conflict-free b64 reads)
constexpr int AY_PLANE = AY_COLS * AY_COLP;       // 17408
constexpr int AY_BUF = 3 * AY_PLANE;              // 52224

__global__ __launch_bounds__(512) void apply_coeffs_x3_m256n128(const unsigned short* __restrict__ W3, int Bt, int Rt,
                                                                const float* __restrict__ src1, int n1,
                                                                const float* __restrict__ src2, int n2, int64_t K,
                                                                int64_t ntiles, float* __restrict__ out) {
    __shared__ __attribute__((aligned(16))) unsigned char zs[2 * AY_BUF];
    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int nchunk = (n1 + n2) / AY_ROWS;                   // n1 % 64 == 0 (host)
    const int m0 = blockIdx.y * AB_MT;
    if (wave < 4) {
        // ---------------------------------------------------------------- producers: 64 stack rows x 128 columns per stage
        const int c4 = (t & 31) * 4, rp0 = t >> 5;            // column group, first row pair (0..7)
        float4 v[8];
        auto load_stage = [&](int64_t tile, int c) {
            const int g0 = c * AY_ROWS;
            const float* base = g0 < n1 ? src1 + (int64_t)g0 * K : src2 + (int64_t)(g0 - n1) * K;
            const int64_t col = tile * AY_COLS + c4;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int r = 2 * (rp0 + 8 * (j >> 1)) + (j & 1);
                v[j] = (col + 4 <= K) ? *reinterpret_cast<const float4*>(base + (int64_t)r * K + col)
                                      : make_float4(0.f, 0.f, 0.f, 0.f);
            }
        };
        int64_t tile = blockIdx.x;
        int c = 0, buf = 0;
        if (tile < ntiles) load_stage(tile, 0);
        while (tile < ntiles) {
            unsigned char* zb = zs + buf * AY_BUF;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int r = 2 * (rp0 + 8 * i);
                const float a[4] = {v[2 * i].x, v[2 * i].y, v[2 * i].z, v[2 * i].w};
                const float b[4] = {v[2 * i + 1].x, v[2 * i + 1].y, v[2 * i + 1].z, v[2 * i + 1].w};
#pragma unroll
                for (int cc = 0; cc < 4; ++cc) {
                    unsigned ha, ma, la, hb, mb, lb;
                    split3u(a[cc], ha, ma, la);
                    split3u(b[cc], hb, mb, lb);
                    const int off = (c4 + cc) * AY_COLP + r * 2;
                    *reinterpret_cast<unsigned*>(zb + off) = __builtin_amdgcn_perm(hb, ha, 0x07060302u);
                    *reinterpret_cast<unsigned*>(zb + AY_PLANE + off) = __builtin_amdgcn_perm(mb, ma, 0x07060302u);
                    *reinterpret_cast<unsigned*>(zb + 2 * AY_PLANE + off) = __builtin_amdgcn_perm(lb, la, 0x07060302u);
                }
            }
            if (++c == nchunk) { c = 0; tile += gridDim.x; }
            if (tile < ntiles) load_stage(tile, c);
            buf ^= 1;
            __syncthreads();
        }
        __syncthreads();          // the consumers' closing barrier
        return;
    }
    // -------------------------------------------------------------------- consumers
    const int w = wave - 4;                                   // rows m0 + 64 w .. + 63, all 128 columns
    const int nsteps = Rt >> 4;                               // k-steps per tile (a multiple of 4)
    const int64_t plane = (int64_t)Bt * Rt;
    const unsigned short* wb0 = W3 + (int64_t)((m0 + 64 * w) / 32) * nsteps * 512 + lane * 8;
    const unsigned short* wb1 = wb0 + (int64_t)nsteps * 512;
    abf16x8 A[2][2][3];                                       // k-step g lives in A[g & 1]: one step (48 MFMAs) ahead
    auto ldA = [&](int g, int slot) {
        const int64_t o = (int64_t)512 * g;
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) {
            A[slot][0][pl] = *reinterpret_cast<const abf16x8*>(wb0 + pl * plane + o);
            A[slot][1][pl] = *reinterpret_cast<const abf16x8*>(wb1 + pl * plane + o);
        }
    };
    ldA(0, 0);
    const int boff = (lane & 31) * AY_COLP + 16 * (lane >> 5);
    int buf = 0;
    __syncthreads();                                          // the first stage is in buffer 0
    for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        f32x16 acc[2][4];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
        for (int c = 0; c < nchunk; ++c, buf ^= 1) {
            const unsigned char* zb = zs + buf * AY_BUF;
            abf16x8 Bf[2][2][3];                              // [pair parity][column tile of the pair][piece]
            auto ldB = [&](int st, int pair, int par) {
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int pl = 0; pl < 3; ++pl) {
                        uint2* q = reinterpret_cast<uint2*>(&Bf[par][j][pl]);
                        const unsigned char* src = zb + pl * AY_PLANE + boff + (64 * pair + 32 * j) * AY_COLP + 32 * st;
                        q[0] = *reinterpret_cast<const uint2*>(src);
                        q[1] = *reinterpret_cast<const uint2*>(src + 8);
                    }
            };
            ldB(0, 0, 0);
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                int gn = c * 4 + s + 1;
                if (gn >= nsteps) gn -= nsteps;
                ldA(gn, (s + 1) & 1);
#pragma unroll
                for (int pair = 0; pair < 2; ++pair) {
                    // the NEXT pair's staged fragments (or the next step's first pair) while this pair's MFMAs run
                    if (pair == 0) ldB(s, 1, 1);
                    else if (s < 3) ldB(s + 1, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
#define KCCOT_A4(PA, PB)                                                                                                         \
                    acc[0][2 * pair] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[s & 1][0][PA], Bf[pair][0][PB], acc[0][2 * pair], 0, 0, 0);         \
                    acc[0][2 * pair + 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[s & 1][0][PA], Bf[pair][1][PB], acc[0][2 * pair + 1], 0, 0, 0); \
                    acc[1][2 * pair] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[s & 1][1][PA], Bf[pair][0][PB], acc[1][2 * pair], 0, 0, 0);         \
                    acc[1][2 * pair + 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[s & 1][1][PA], Bf[pair][1][PB], acc[1][2 * pair + 1], 0, 0, 0);
                    KCCOT_A4(1, 1) KCCOT_A4(0, 2) KCCOT_A4(2, 0) KCCOT_A4(0, 1) KCCOT_A4(1, 0) KCCOT_A4(0, 0)
#undef KCCOT_A4
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            __syncthreads();                                  // this stage is consumed; the next one is staged
        }
        const int64_t col0 = tile * AY_COLS + (lane & 31);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int64_t col = col0 + 32 * j;
            if (col < K) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int64_t m = m0 + 64 * w + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                    out[m * K + col] = acc[0][j][r];
                    out[(m + 32) * K + col] = acc[1][j][r];
                }
            }
        }
    }
}

// Wt points at the first wanted output row's column; wpitch = full number of output rows of W
// W3 (optional): the three bf16 planes of the SAME coefficients, [3][Bt][Rt] (split_coeffs), positioned at the first
// wanted output row; selects the exact bf16 kernel when the stack is a multiple of 16 rows.
static int launch_apply(const float* Wt, int wpitch, const float* s1, int n1, const float* s2, int n2, int Bout,
                        int64_t K, float* out, hipStream_t st, const unsigned short* W3 = nullptr, int Bt = 0, int Rt = 0,
                        const unsigned short* W3base = nullptr, unsigned short* Wt3 = nullptr) {
    const bool al = (K % 4 == 0) && ((uintptr_t)s1 % 16 == 0) && (n2 == 0 || (uintptr_t)s2 % 16 == 0);
    // option "apply_f32" = 1: the f32-input MFMA kernel (parity runs)
    const bool x3 = al && W3 && (n1 + n2) % 16 == 0 && Rt % 8 == 0 && ((uintptr_t)W3 % 16 == 0) && !opt(OPT_APPLY_F32);
    if (al) {
        // MFMA kernel on blocks: 64 output rows x (up to) 128 stack rows at a time.  One block covers the
        // BASELINE configs[1] batch in a single launch; larger batches run (Bout/64) x (R/128) launches, the
        // later stack chunks accumulating into the output (stream order keeps the sum deterministic).
        const int64_t ntiles = (K + AM_COLS - 1) / AM_COLS;
        const unsigned grid = (unsigned)(ntiles < 512 ? ntiles : 512);   // 2 workgroups per CU (VGPR-limited)
        const int R = n1 + n2;
        // large batches: 256-row output tiles over the whole stack, one launch (option "apply_m256" = 0: the block form)
        if (x3 && Wt3 && Bout % 32 == 0 && (Bout % 64 == 0 || Bout == 32) && Bt % 32 == 0 && n1 % AM_ROWS == 0 &&
            n2 % AM_ROWS == 0 && R == Rt && (R > AM_ROWS || Bout > 64) && opt(OPT_APPLY_M256)) {
            const unsigned gxb = (unsigned)(ntiles < 256 ? ntiles : 256);
            // W3 is positioned at the first wanted output row (a multiple of 256 here): retile from the plane base
            const int64_t row0 = (W3 - W3base) / Rt;
            const int64_t ngroups = (int64_t)3 * Bt * (Rt / 8);
            hipLaunchKernelGGL(retile_coeffs, dim3((unsigned)((ngroups + 255) / 256)), dim3(256), 0, st, W3base, Bt, Rt, Wt3);
            int rct = launch_status("retile_coeffs");
            if (rct) return rct;
            const unsigned short* Wuse = Wt3 + (row0 / 32) * (int64_t)(Rt / 16) * 512;
            // 128-column tiles (half the W bytes per MFMA) once there are >= 20 of them per workgroup -- below that the
            // tail of the persistent tile loop costs more than the W stream saves (B = 256, K = 368 640: 0.73 vs 0.68 ms;
            // B = 512, K = 2.36 M: 13.2 vs 13.8 ms).
            const int64_t nt128 = (K + AY_COLS - 1) / AY_COLS;
            // 256 x 256 tiles, W panel through LDS (cost_bwd_q256.hip; option "apply_q256")
            if (opt(OPT_APPLY_Q256) && apply_q256_applies(Bout, n1, n2, K) && (K + 255) / 256 * (Bout / 256) >= 512)
                return launch_apply_q256(Wuse, Bt, Rt, s1, n1, s2, n2, Bout, K, out, st);
            if (Bout % AB_MT == 0 && nt128 >= 20 * 256 && n1 % AY_ROWS == 0 && n2 % AY_ROWS == 0) {
                const unsigned gy = (unsigned)(nt128 < 256 ? nt128 : 256);
                hipLaunchKernelGGL(apply_coeffs_x3_m256n128, dim3(gy, Bout / AB_MT), dim3(512), 0, st, Wuse, Bt, Rt, s1, n1, s2, n2,
                                   K, nt128, out);
                return launch_status("apply_coeffs_x3_m256n128");
            }
            // the tallest tile that divides the wanted rows
#define KCCOT_APPLY_ROWS(RT, CT, WR, WC)                                                                                         \
            hipLaunchKernelGGL((apply_coeffs_x3_rows<RT, CT, WR, WC>), dim3(gxb, Bout / (32 * RT * WR)), dim3(512), 0, st, Wuse, Bt, Rt, \
                               s1, n1, s2, n2, K, ntiles, out)
            if (Bout % AB_MT == 0)
                hipLaunchKernelGGL(apply_coeffs_x3_m256, dim3(gxb, Bout / AB_MT), dim3(512), 0, st, Wuse, Bt, Rt, s1, n1, s2, n2, K, ntiles, out);
            else if (Bout % 128 == 0) KCCOT_APPLY_ROWS(1, 2, 4, 1);
            else if (Bout % 64 == 0) KCCOT_APPLY_ROWS(1, 1, 2, 2);
            else KCCOT_APPLY_ROWS(1, 1, 1, 2);
#undef KCCOT_APPLY_ROWS
            return launch_status("apply_coeffs_x3_m256");
        }
        for (int ob = 0; ob < Bout; ob += 64) {
            const int bo = Bout - ob < 64 ? Bout - ob : 64;
            for (int r0 = 0; r0 < R; r0 += AM_ROWS) {
                const int r1 = r0 + AM_ROWS < R ? r0 + AM_ROWS : R;
                // rows [r0, r1) of the stack: the part inside src1, then the part inside src2
                const int a0 = r0 < n1 ? r0 : n1, a1 = r1 < n1 ? r1 : n1;
                const int b0 = r0 > n1 ? r0 - n1 : 0, b1 = r1 > n1 ? r1 - n1 : 0;
                const int c1 = a1 - a0, c2 = b1 - b0;
                const float* p1 = c1 > 0 ? s1 + (int64_t)a0 * K : s2 + (int64_t)b0 * K;
                const float* p2 = c1 > 0 ? (c2 > 0 ? s2 + (int64_t)b0 * K : p1) : p1;
                const int m1 = c1 > 0 ? c1 : c2, m2 = c1 > 0 ? c2 : 0;
                const float* W = Wt + (int64_t)r0 * wpitch + ob;
                float* o = out + (int64_t)ob * K;
                const int acc = r0 > 0;
                const int rr = r1 - r0;
                if (x3) {
                    const unsigned gx3 = (unsigned)(ntiles < 256 ? ntiles : 256);     // 101 KB of LDS: one workgroup per CU
                    if (acc) hipLaunchKernelGGL(apply_coeffs_x3<true>, dim3(gx3), dim3(512), 0, st, W3 + (int64_t)ob * Rt, Bt, Rt, r0, p1, m1,
                                                p2, m2, bo, K, ntiles, o);
                    else hipLaunchKernelGGL(apply_coeffs_x3<false>, dim3(gx3), dim3(512), 0, st, W3 + (int64_t)ob * Rt, Bt, Rt, r0, p1, m1,
                                            p2, m2, bo, K, ntiles, o);
                    const int rc3 = launch_status("apply_coeffs_x3");
                    if (rc3) return rc3;
                    continue;
                }
                if (rr > 64) hipLaunchKernelGGL(apply_coeffs_mfma<64>, dim3(grid), dim3(256), 0, st, W, wpitch, p1, m1, p2, m2, bo, K, ntiles, o, acc);
                else if (rr > 32) hipLaunchKernelGGL(apply_coeffs_mfma<32>, dim3(grid), dim3(256), 0, st, W, wpitch, p1, m1, p2, m2, bo, K, ntiles, o, acc);
                else hipLaunchKernelGGL(apply_coeffs_mfma<16>, dim3(grid), dim3(256), 0, st, W, wpitch, p1, m1, p2, m2, bo, K, ntiles, o, acc);
                const int rc = launch_status("apply_coeffs_mfma");
                if (rc) return rc;
            }
        }
        return 0;
    }
    const unsigned gx = (unsigned)((K + 255) / 256);
    if (Bout <= 8) hipLaunchKernelGGL(apply_coeffs<8>, dim3(gx, 1), dim3(256), 0, st, Wt, wpitch, s1, n1, s2, n2, Bout, K, out);
    else if (Bout <= 16) hipLaunchKernelGGL(apply_coeffs<16>, dim3(gx, 1), dim3(256), 0, st, Wt, wpitch, s1, n1, s2, n2, Bout, K, out);
    else if (Bout <= 32) hipLaunchKernelGGL(apply_coeffs<32>, dim3(gx, 1), dim3(256), 0, st, Wt, wpitch, s1, n1, s2, n2, Bout, K, out);
    else hipLaunchKernelGGL(apply_coeffs<64>, dim3(gx, (Bout + 63) / 64), dim3(256), 0, st, Wt, wpitch, s1, n1, s2, n2, Bout, K, out);
    return launch_status("apply_coeffs");
}

}  // namespace kccot

using namespace kccot;

extern "C" size_t kccot_pairwise_cost3_bwd_workspace_bytes(int B, int64_t K) {
    (void)K;
    if (B <= 0) return 0;
    // W [2B][B] f32, then its three bf16 planes [3][B][2B], then (B % 256 == 0) their fragment-major copy
    return align_up((size_t)2 * B * B * sizeof(float), 256) + 2 * align_up((size_t)3 * B * 2 * B * sizeof(unsigned short), 256);
}

static int cost3_bwd_rows_impl(const float* g3, const float* gscale, const float* real, const float* fake, int B,
                               int64_t K, float sc, const float* h_fake, const float* h_real,
                               const float* m_real, const float* m_fake, int T, int J,
                               int row_begin, int row_count,
                               float* dfake, float* dh_fake, float* dh_real, float* dm_real,
                               float* dm_fake, void* ws, size_t ws_bytes, kccot_stream_t stream) {
    if (!g3 || !real || !fake) return fail(KCCOT_EINVAL, "pairwise_cost3_bwd: null pointer");
    if (row_begin < 0 || row_count <= 0 || row_begin + row_count > B)
        return fail(KCCOT_EINVAL, "pairwise_cost3_bwd: bad row range [%d, %d) of %d", row_begin, row_begin + row_count, B);
    if (B <= 0 || K <= 0 || T < 1 || J < 1)
        return fail(KCCOT_EINVAL, "pairwise_cost3_bwd: bad shape B=%d K=%lld T=%d J=%d", B, (long long)K, T, J);
    if ((dh_fake || dh_real || dm_real || dm_fake) && (!h_fake || !h_real || !m_real || !m_fake))
        return fail(KCCOT_EINVAL, "pairwise_cost3_bwd: feature gradients requested without features");
    hipStream_t st = (hipStream_t)stream;
    const int64_t bb = (int64_t)B * B;
    const float *gxy = g3, *gxx = g3 + bb, *gyy = g3 + 2 * bb;
    int rc;
    // gan_utils.py:221-223: h_fake rows of xy (cols m_real) and of yy (cols m_fake); h_real rows of xx
    // (cols m_real); m_real cols of xy (rows h_fake) and of xx (rows h_real); m_fake cols of yy (rows h_fake)
    CausalGradBatch cg{};
    if (dh_fake) cg.job[cg.njobs++] = CausalGradJob{dh_fake, CG_H, row_count, B, row_begin, B, {gxy, gyy}, {m_real, m_fake}};
    if (dh_real) cg.job[cg.njobs++] = CausalGradJob{dh_real, CG_H, row_count, B, row_begin, B, {gxx, nullptr}, {m_real, nullptr}};
    if (dm_real) cg.job[cg.njobs++] = CausalGradJob{dm_real, CG_M, row_count, B, row_begin, B, {gxy, gxx}, {h_fake, h_real}};
    if (dm_fake) cg.job[cg.njobs++] = CausalGradJob{dm_fake, CG_M, row_count, B, row_begin, B, {gyy, nullptr}, {h_fake, nullptr}};
    if (!dfake) return launch_causal_grads(cg, T, J, sc, st, gscale);
    const size_t need = kccot_pairwise_cost3_bwd_workspace_bytes(B, K);
    if (!ws || ws_bytes < need)
        return fail(KCCOT_EWORKSPACE, "pairwise_cost3_bwd: workspace %zu < required %zu", ws_bytes, need);
    // B <= 64, whole batch, aligned rows: ONE launch (W built by the consumer waves, feature gradients in spare workgroups)
    if (B <= 64 && B % 8 == 0 && row_begin == 0 && row_count == B && K % 4 == 0 && (uintptr_t)real % 16 == 0 &&
        (uintptr_t)fake % 16 == 0 && !opt(OPT_APPLY_F32) && opt(OPT_APPLY_ONE_LAUNCH)) {
        Loss3Apply la{};
        la.gxy = gxy; la.gyy = gyy; la.gscale = gscale; la.real = real; la.fake = fake; la.out = dfake;
        la.B = B; la.sc = sc; la.K = K; la.ntiles = (K + AM_COLS - 1) / AM_COLS;
        la.cg = cg; la.T = T; la.J = J; la.gx = (T * J + 15) / 16; la.gy = (row_count + 15) / 16;
        // 101 KB of LDS: one workgroup per CU; 240 (not 256) when that makes the persistent tile loops even (configs[1]: 1920 tiles)
        const int64_t cap = (la.ntiles % 240 == 0 && la.ntiles % 256 != 0) ? 240 : 256;
        la.nmain = (int)(la.ntiles < cap ? la.ntiles : cap);
#ifdef KCCOT_DIAG
        // timing ablations of the diagnostic twin only (results wrong): KCCOT_L3_ABLATE bit 0: no feature-gradient tasks
        if (const char* e = getenv("KCCOT_L3_ABLATE")) { if (atoi(e) & 1) la.cg.njobs = 0; }
#endif
        const bool tasks = la.cg.njobs > 0;
        if (B == 64 && tasks) hipLaunchKernelGGL((apply_coeffs_x3_loss3<true, true>), dim3(la.nmain), dim3(512), 0, st, la);
        else if (B == 64) hipLaunchKernelGGL((apply_coeffs_x3_loss3<true, false>), dim3(la.nmain), dim3(512), 0, st, la);
        else if (tasks) hipLaunchKernelGGL((apply_coeffs_x3_loss3<false, true>), dim3(la.nmain), dim3(512), 0, st, la);
        else hipLaunchKernelGGL((apply_coeffs_x3_loss3<false, false>), dim3(la.nmain), dim3(512), 0, st, la);
        return launch_status("apply_coeffs_x3_loss3");
    }
    float* Wt = static_cast<float*>(ws);
    unsigned short* W3 = reinterpret_cast<unsigned short*>(static_cast<char*>(ws) + align_up((size_t)2 * B * B * sizeof(float), 256));
    if (cg.njobs == 0) {
        hipLaunchKernelGGL(build_coeffs, dim3(B), dim3(256), 0, st, (int)CO_LOSS3_DFAKE, gxy, gyy, B, B, sc, Wt, gscale);
        if ((rc = launch_status("build_coeffs"))) return rc;
        hipLaunchKernelGGL(split_coeffs, dim3((2 * B * B + 255) / 256), dim3(256), 0, st, (const float*)Wt, B, 2 * B, B, W3);
        if ((rc = launch_status("split_coeffs"))) return rc;
    } else {
        const int gx = (T * J + 15) / 16, gy = (row_count + 15) / 16;
        hipLaunchKernelGGL(coeffs_and_causal_grads, dim3(B + gx * gy * cg.njobs), dim3(256), 0, st, (int)CO_LOSS3_DFAKE,
                           gxy, gyy, B, B, sc, Wt, B, cg, T, J, gx, gy, W3, gscale);
        if ((rc = launch_status("coeffs_and_causal_grads"))) return rc;
    }
    unsigned short* Wt3 = reinterpret_cast<unsigned short*>(reinterpret_cast<char*>(W3) + align_up((size_t)3 * B * 2 * B * sizeof(unsigned short), 256));
    return launch_apply(Wt + row_begin, B, real, B, fake, B, row_count, K, dfake, st, W3 + (int64_t)row_begin * 2 * B, B, 2 * B,
                        W3, (B % 32 == 0 && row_begin % 32 == 0) ? Wt3 : nullptr);
}

extern "C" int kccot_pairwise_cost3_bwd_rows_f32(const float* g3, const float* real, const float* fake, int B,
                                                 int64_t K, float sc, const float* h_fake, const float* h_real,
                                                 const float* m_real, const float* m_fake, int T, int J,
                                                 int row_begin, int row_count,
                                                 float* dfake, float* dh_fake, float* dh_real, float* dm_real,
                                                 float* dm_fake, void* ws, size_t ws_bytes, kccot_stream_t stream) {
    return cost3_bwd_rows_impl(g3, nullptr, real, fake, B, K, sc, h_fake, h_real, m_real, m_fake, T, J, row_begin, row_count,
                               dfake, dh_fake, dh_real, dm_real, dm_fake, ws, ws_bytes, stream);
}

// g3 = d loss / d C3 at dLoss = 1 (kccot_sinkhorn_divergence_fused_f32), gscale = ONE device float dLoss/dloss
extern "C" int kccot_pairwise_cost3_bwd_scaled_f32(const float* g3, const float* gscale, const float* real, const float* fake,
                                                   int B, int64_t K, float sc, const float* h_fake, const float* h_real,
                                                   const float* m_real, const float* m_fake, int T, int J,
                                                   float* dfake, float* dh_fake, float* dh_real, float* dm_real,
                                                   float* dm_fake, void* ws, size_t ws_bytes, kccot_stream_t stream) {
    if (!gscale) return fail(KCCOT_EINVAL, "pairwise_cost3_bwd_scaled: null gscale");
    return cost3_bwd_rows_impl(g3, gscale, real, fake, B, K, sc, h_fake, h_real, m_real, m_fake, T, J, 0, B,
                               dfake, dh_fake, dh_real, dm_real, dm_fake, ws, ws_bytes, stream);
}

extern "C" int kccot_pairwise_cost3_bwd_f32(const float* g3, const float* real, const float* fake, int B,
                                            int64_t K, float sc, const float* h_fake, const float* h_real,
                                            const float* m_real, const float* m_fake, int T, int J,
                                            float* dfake, float* dh_fake, float* dh_real, float* dm_real,
                                            float* dm_fake, void* ws, size_t ws_bytes, kccot_stream_t stream) {
    return kccot_pairwise_cost3_bwd_rows_f32(g3, real, fake, B, K, sc, h_fake, h_real, m_real, m_fake, T, J, 0, B,
                                             dfake, dh_fake, dh_real, dm_real, dm_fake, ws, ws_bytes, stream);
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
            hipLaunchKernelGGL(build_coeffs, dim3(Bx), dim3(256), 0, st, (int)CO_SAME, g,
                               (const float*)nullptr, Bx, Bx, sc, W1, (const float*)nullptr);
            if ((rc = launch_status("build_coeffs"))) return rc;
            if ((rc = launch_apply(W1, Bx, x, Bx, x, 0, Bx, K, dx, st))) return rc;
        }
    } else {
        if (dx) {
            hipLaunchKernelGGL(build_coeffs, dim3(Bx), dim3(256), 0, st, (int)CO_DX, g,
                               (const float*)nullptr, Bx, By, sc, W1, (const float*)nullptr);
            if ((rc = launch_status("build_coeffs"))) return rc;
            if ((rc = launch_apply(W1, Bx, x, Bx, y, By, Bx, K, dx, st))) return rc;
        }
        if (dy) {
            hipLaunchKernelGGL(build_coeffs, dim3(By), dim3(256), 0, st, (int)CO_DY, g,
                               (const float*)nullptr, Bx, By, sc, W2, (const float*)nullptr);
            if ((rc = launch_status("build_coeffs"))) return rc;
            if ((rc = launch_apply(W2, By, x, Bx, y, By, By, K, dy, st))) return rc;
        }
    }
    CausalGradBatch cg{};
    if (dh) cg.job[cg.njobs++] = CausalGradJob{dh, CG_H, Bx, By, 0, By, {g, nullptr}, {M, nullptr}};
    if (dM) cg.job[cg.njobs++] = CausalGradJob{dM, CG_M, By, Bx, 0, By, {g, nullptr}, {h, nullptr}};
    return launch_causal_grads(cg, T, J, sc, st);
}
