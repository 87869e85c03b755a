/*
 * kccot.h -- C ABI of the MI355X-native causal-OT (Sinkhorn) + kernel-smoothing loss path.
 *
 * This is the drop-in boundary for the hot path of neuripss2020/kccotgan.
 * The reference has no FFI: its boundary is a set of Python call signatures in gan_utils.py /
 * data_utils.py.  Every entry point below cites the reference function whose arithmetic it
 * replaces; the Python host mirror (kccotgan_amd/gan_utils.py, kccotgan_amd/data_utils.py)
 * keeps those signatures and binds these symbols through ctypes.
 *
 * Conventions
 *   - all pointers are DEVICE pointers (HBM) unless named host_*; fp32, row-major, contiguous;
 *     video tensors are viewed as [B, K] with K = T*H*W*C (the reference's
 *     transpose(0,2,1,3,4) at gan_utils.py:217-220 does not change a sum over all of T,H,W,C,
 *     so the native [B,H,T,W,C] buffer is read as-is);
 *   - every call is asynchronous on `stream` (a hipStream_t passed as void*; NULL = default
 *     stream); nothing allocates, frees or synchronises: the caller owns outputs and the
 *     workspace (size from the matching *_workspace_bytes query);
 *   - inputs are never written; NaN/Inf propagate unchanged (the caller's np.isfinite guard,
 *     kernel_train.py:323, keeps working);
 *   - return value: 0 = ok; KCCOT_E* < 0 = rejected before any launch; > 0 = a hipError_t
 *     from the launch.  kccot_last_error() returns a thread-local message.
 */
#ifndef KCCOT_H
#define KCCOT_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define KCCOT_VERSION 301          /* 0.3.1 */
#define KCCOT_EINVAL (-1)          /* bad shape / null pointer / inconsistent arguments      */
#define KCCOT_EUNSUPPORTED (-2)    /* valid request outside what this build implements       */
#define KCCOT_EWORKSPACE (-3)      /* workspace too small                                    */
#define KCCOT_EABORTED (-4)        /* a multi-CU Sinkhorn solve gave up (kccot_sinkhorn_status) */

/* cost flags */
#define KCCOT_COST_SAME 1u         /* x and y are the same tensor: upper triangle computed,   */
                                   /* mirrored, l2 diagonal exactly 0 (as gan_utils.py:16)    */
#define KCCOT_COST_FORCE_DIRECT 2u /* use the direct-difference kernel (exact (x-y)^2 form)   */
#define KCCOT_COST_FORCE_MFMA 4u   /* use the stacked-Gram f32-MFMA kernel                    */
#define KCCOT_COST_PARTIAL_ONLY 8u /* profiling aid: launch only the K-split partial kernel   */
                                   /* (the dominant one); C_out is NOT written                */
#define KCCOT_COST_GRAM_SUMS_ONLY 16u /* kccot_pairwise_cost3_f32: stop after the fp64 Gram sums (and the causal sums)  */
                                      /* are in the workspace -- nothing is written to C3                               */
#define KCCOT_COST_FROM_GRAM_SUMS 32u /* kccot_pairwise_cost3_f32: only the finalize step, from the sums left in the    */
                                      /* SAME workspace by a GRAM_SUMS_ONLY call (real / fake are not read)             */

/* Sinkhorn stop modes */
#define KCCOT_STOP_COUNT 0         /* compute_sinkhorn: stop when err<thresh && nits >= Lmin  */
                                   /*   (gan_utils.py:157-160)                                */
#define KCCOT_STOP_INDEX 1         /* benchmark_sinkhorn: ... && loop index i >= Lmin (:116)  */

typedef void* kccot_stream_t;

int kccot_version(void);
const char* kccot_last_error(void);

/* ---------------------------------------------------------------------------------------------
 * Options.  Everything the library decides at run time beyond its arguments is in this table (the reference has no
 * counterpart: it has one code path).  Process-wide, thread-safe (one atomic int each; a call in flight on another thread
 * sees the old or the new value), read with plain loads on the call paths -- NO environment variable is consulted by any
 * entry point.  kccot_set_option returns 0 or KCCOT_EINVAL (unknown name / value out of range).
 * The environment variable KCCOT_OPTIONS="name=value,name=value" seeds the table ONCE, at the first use of the library
 * (command-line tools: bench.py, tools/); a binding calls kccot_set_option.
 *
 *   name                      default  meaning
 *   gram_f32                  0        1: cost matrices on the f32-input MFMA kernels (v_mfma_f32_32x32x2_f32) instead of the
 *                                      exact bf16 three-way split -- same fp32 arithmetic, 1/16 of the matrix rate; parity runs
 *   apply_f32                 0        1: the video gradient dfake = W [X;Y] on the f32-input MFMA kernel, likewise
 *   cost_tiled                1        0: no tiled Gram kernels for B % 128 == 0 (the blocked or direct path serves instead)
 *   cost_tile256              1        0: B % 256 == 0 runs the 128-row tiles of cost_tiled.hip instead of the 256-row ones
 *   cost_blocked              1        0: no 64 x 64-block MFMA path for B % 64 == 0 (the direct VALU kernel serves instead)
 *   apply_m256                1        0: the video gradient of B % 256 == 0 in 64-row blocks instead of 256-row tiles
 *   apply_one_launch          1        0: the loss's video gradient at B <= 64 as coefficient build + apply (two launches);
 *                                      1: one launch -- the apply kernel's consumer waves form their coefficient fragments from dC
 *                                      themselves, the feature gradients run in its spare workgroups
 *   apply_q256                1        0: the video gradient of B % 256 == 0 on 256 x 64 / 128 tiles with the coefficient fragments
 *                                      streamed from L2; 1: from 512 tiles on, 256 x 256 output tiles with the coefficient panel
 *                                      staged through LDS (csrc/cost_bwd_q256.hip; bit-identical, 13.7 -> 11.1 ms at configs[4])
 *   sinkhorn_shortcut         1        0: execute every Sinkhorn iteration; 1: skip iterations EXACTLY once the fp32 state is
 *                                      bit-for-bit periodic (identical results; see kccot_sinkhorn_fwd_f32)
 *   sinkhorn_fused            1        0: kccot_sinkhorn_fused_eligible reports 0 (solve and reverse sweep as two launches)
 *   sinkhorn_fused_max_n      64       largest n the fused solve + sweep launch accepts (<= 128; above 64 it spills registers)
 *   sinkhorn_lanes_per_line   0        4 / 8 / 16 lanes per matrix line for 32 < n <= 64 (0: forward 8, reverse sweep 16)
 *   sinkhorn_coop             1        0: 128 < n <= 1024 on the one-workgroup streaming solver instead of the multi-CU one
 *   sinkhorn_coop_xcd         1        1: the multi-CU solver lays one problem out per XCD (1-D grid dealt round-robin) and, after
 *                                      an in-kernel check that all workgroups of a problem really share an XCD, exchanges the
 *                                      duals through that XCD's L2 (n = 256: 4.5 -> 2.4 us per iteration; same bits); else, and
 *                                      with 0, the agent-scope exchange; 2 (tests): the check on the 2-D grid, which must fail
 *   sinkhorn_coop_max_wg      0        > 0: workgroups the multi-CU solver may assume co-resident (a caller that runs in a
 *                                      partition or under a CU mask); 0: queried from the device (CU count x occupancy, 3/4)
 *                                      (384 on a whole MI355X also lets n = 512 use the per-XCD layout of sinkhorn_coop_xcd:
 *                                      32 workgroups = every CU of an XCD; by default only n <= 256 = half an XCD)
 *   smooth_stream             1        0: KernelSmoothing on the per-axis global stencils (any radius) instead of the
 *                                      streaming kernels (radius 3 / 4)
 *   smooth_generic            0        1: the any-length / any-alignment streaming kernels even where the register-line ones apply
 *   smooth_fused_tw           1        0: T and W stage of the 3-D smoothing as two launches instead of one
 *   smooth_bwd_fold           1        0: the smoothing backward computes the normalisation's two sums in a pass of its own
 *                                      first; 1: from 4 M elements (temporal) / 32 M (3-D) on its first adjoint stage gathers
 *                                      them and a sparse fix-up applies them (two tensor reads fewer; data with > 32 arg-max
 *                                      elements -- saturated still regions -- then pays the two-pass chain on top, decided on
 *                                      the device: set 0 for such data); 2: at every size (tests)
 *   smooth_fused3             1        0: the 3-D smoothing as the chain of per-axis stages (five to seven tensor moves forward, seven
 *                                      backward); 1: T, W and H stage in ONE pass per call phase (three moves + halo; forward: same
 *                                      bits, backward: to rounding) where that was measured faster: forward three channels and
 *                                      >= 20 M elements (radius 3 / 4), backward >= 3.5 M elements (radius 3; statistics folded:
 *                                      smooth_bwd_fold != 0, or handed in: kccot_smooth_bwd_sharded_f32); 2: wherever it can run
 *                                      (C = 1 or 3; tests)
 * ------------------------------------------------------------------------------------------- */
int kccot_set_option(const char* name, int value);
int kccot_get_option(const char* name, int* value);
int kccot_option_count(void);
const char* kccot_option_name(int index);      /* 0 <= index < kccot_option_count(), else NULL */

/* ---------------------------------------------------------------------------------------------
 * Pairwise cost.  Replaces cost_xy (gan_utils.py:6-18), modified_cost (:21-43) and
 * bi_causal_modified_cost (:46-72):
 *     C[i,j] = sc * sum_k (x[i,k]-y[j,k])^2
 *            + sc * sum_{t<T-1,q} h1[i,t,q] * (M1[j,t+1,q]-M1[j,t,q])      if h1 != NULL
 *            + sc * sum_{t<T-1,q} h2[i,t,q] * (M2[j,t+1,q]-M2[j,t,q])      if h2 != NULL
 * h* index ROWS ([Bx,T,J]), M* index COLUMNS ([By,T,J]) -- gan_utils.py:37.
 * C_out is [Bx,By].
 * ------------------------------------------------------------------------------------------- */
size_t kccot_pairwise_cost_workspace_bytes(int Bx, int By, int64_t K);
int kccot_pairwise_cost_f32(const float* x, const float* y, int Bx, int By, int64_t K, float sc,
                            const float* h1, const float* M1, const float* h2, const float* M2,
                            int T, int J, unsigned flags, float* C_out,
                            void* ws, size_t ws_bytes, kccot_stream_t stream);

/* The three cost matrices of compute_sinkhorn_loss (gan_utils.py:221-223) in one pass that
 * reads `real` and `fake` once:  C3[0] = xy: modified_cost(real, fake, h_fake, m_real)
 *                                C3[1] = xx: modified_cost(real, real, h_real, m_real)
 *                                C3[2] = yy: modified_cost(fake, fake, h_fake, m_fake)
 * C3 is [3,B,B].  The four feature pointers may ALL be NULL: plain scaled squared distances. */
size_t kccot_pairwise_cost3_workspace_bytes(int B, int64_t K);
int kccot_pairwise_cost3_f32(const float* real, const float* fake, int B, int64_t K, float sc,
                             const float* h_fake, const float* h_real,
                             const float* m_real, const float* m_fake, int T, int J,
                             unsigned flags, float* C3,
                             void* ws, size_t ws_bytes, kccot_stream_t stream);

/* The two flags above split the call for a caller that shards the CONTRACTION over ranks (every rank holds all B
 * samples but only a slice [B, Ks] of the features): each rank runs GRAM_SUMS_ONLY on its slice, the callers
 * all-reduce(SUM) the fp64 Gram sums in place, and FROM_GRAM_SUMS turns them into the three cost matrices.
 * kccot_pairwise_cost3_gram_sums_span reports where the sums sit in the workspace (byte offset, number of doubles);
 * 0 doubles = this (B, K) has no Gram path (use the other protocol).  Both calls must use the same B, K, workspace. */
int kccot_pairwise_cost3_gram_sums_span(int B, int64_t K, size_t* byte_offset, size_t* n_doubles);

/* Row blocks [row_count, B] of the same three matrices for the batch-sharded caller (kccotgan_amd/dist.py: rank g
 * owns samples [row_begin, row_begin+row_count) of the gathered batch): one launch of the exact direct-difference
 * kernel over the three problems.  real / fake: gathered [B,K]; features: gathered [B,T,J]; C3_rows: [3,row_count,B].
 * xx rows keep the reference's value for the i == j entries as computed ((x - x)^2 summed = 0 exactly). */
size_t kccot_pairwise_cost3_rows_workspace_bytes(int row_count, int B, int64_t K);
int kccot_pairwise_cost3_rows_f32(const float* real, const float* fake, int B, int64_t K, float sc,
                                  const float* h_fake, const float* h_real, const float* m_real,
                                  const float* m_fake, int T, int J, int row_begin, int row_count,
                                  float* C3_rows, void* ws, size_t ws_bytes, kccot_stream_t stream);

/* The same row blocks on the matrix pipe (round 3; what the batch-sharded caller runs for B > 64): the rank forms the Gram
 * row block [X_I ; E_I] [X ; E]^T (E = fake - real) with the exact bf16 split of the single-GPU kernels and combines it
 * in fp64 with the pair-difference identities.  Those need x_j.x_j, e_j.e_j, x_j.e_j of EVERY sample j:
 *   kccot_row_norms_f64 writes them for `rows` rows as [rows,3] doubles (each rank: its own rows, from its local shard,
 *   before the all-gather of the videos; with a workspace of kccot_row_norms_workspace_bytes(rows) a row is split over
 *   several workgroups, with ws = NULL one workgroup sums a whole row); the caller all-gathers the 3 B doubles and passes them as `norms` [B,3].
 *   norms == NULL: the call computes them for all B rows itself (one more pass over both videos).
 * Supported (kccot_pairwise_cost3_rows_gram_supported): row_count 32 or 64, B % 128 == 0, K % 4 == 0, 256 <= K <= 2^22;
 * otherwise use kccot_pairwise_cost3_rows_f32.  Results agree with it to fp32 rounding (1e-5 of max|C|). */
int kccot_pairwise_cost3_rows_gram_supported(int row_count, int B, int64_t K);
size_t kccot_pairwise_cost3_rows_gram_workspace_bytes(int row_count, int B, int64_t K);
size_t kccot_row_norms_workspace_bytes(int rows);
int kccot_row_norms_f64(const float* real, const float* fake, int rows, int64_t K, double* norms_out,
                        void* ws, size_t ws_bytes, kccot_stream_t stream);
int kccot_pairwise_cost3_rows_gram_f32(const float* real, const float* fake, int B, int64_t K, float sc,
                                       const float* h_fake, const float* h_real, const float* m_real,
                                       const float* m_fake, int T, int J, int row_begin, int row_count,
                                       const double* norms, float* C3_rows, void* ws, size_t ws_bytes,
                                       kccot_stream_t stream);

/* The same in two stages, for a caller that receives the COLUMNS in pieces (kccotgan_amd/dist.py with
 * KCCOT_DIST_GATHER_CHUNKS = N: the videos are all-gathered in N column ranges and each range's Gram sums are formed while
 * the next ranges are still in flight -- SURVEY.md section 8(e) "chunk the gather along K"):
 *   _sums_f64: real / fake are [B, K] arrays holding ONE column range of all samples (K = its width); the fp64 Gram sums of
 *   the row block (kccot_pairwise_cost3_rows_gram_sums_count doubles) are written (accumulate = 0) or added to
 *   (accumulate != 0); stream order fixes the order of the additions, so a fixed range order gives a reproducible result.
 *   Workspace: kccot_pairwise_cost3_rows_gram_workspace_bytes(row_count, B, K) of that range.
 *   _from_sums_f32: the three row blocks from the complete sums and the norms over ALL columns (kccot_row_norms_f64). */
size_t kccot_pairwise_cost3_rows_gram_sums_count(int row_count, int B);
int kccot_pairwise_cost3_rows_gram_sums_f64(const float* real, const float* fake, int B, int64_t K, int row_begin,
                                            int row_count, double* gsum, int accumulate, void* ws, size_t ws_bytes,
                                            kccot_stream_t stream);
int kccot_pairwise_cost3_rows_gram_from_sums_f32(const double* gsum, int B, float sc, const float* h_fake,
                                                 const float* h_real, const float* m_real, const float* m_fake, int T,
                                                 int J, int row_begin, int row_count, const double* norms,
                                                 float* C3_rows, kccot_stream_t stream);

/* Backward of the three cost matrices: given g3 = dLoss/dC3 [3,B,B] writes
 *   dfake [B,K] (may be NULL), dh_fake, dh_real, dm_real, dm_fake [B,T,J] (each may be NULL).
 * real never receives a gradient (kernel_train.py:252,289). */
size_t kccot_pairwise_cost3_bwd_workspace_bytes(int B, int64_t K);
int kccot_pairwise_cost3_bwd_f32(const float* g3, const float* real, const float* fake, int B,
                                 int64_t K, float sc, const float* h_fake, const float* h_real,
                                 const float* m_real, const float* m_fake, int T, int J,
                                 float* dfake, float* dh_fake, float* dh_real, float* dm_real,
                                 float* dm_fake, void* ws, size_t ws_bytes, kccot_stream_t stream);

/* Same for the batch rows [row_begin, row_begin+row_count) only (the batch-sharded caller: every
 * rank holds the full, replicated g3 and all videos after the all-gather and needs the gradient
 * of ITS samples; outputs are [row_count, ...]). */
int kccot_pairwise_cost3_bwd_rows_f32(const float* g3, const float* real, const float* fake, int B,
                                      int64_t K, float sc, const float* h_fake, const float* h_real,
                                      const float* m_real, const float* m_fake, int T, int J,
                                      int row_begin, int row_count,
                                      float* dfake, float* dh_fake, float* dh_real, float* dm_real,
                                      float* dm_fake, void* ws, size_t ws_bytes, kccot_stream_t stream);

/* Backward of one general cost matrix (cost_xy / modified_cost): given g = dLoss/dC [Bx,By]
 * writes dx [Bx,K], dy [By,K], dh [Bx,T,J], dM [By,T,J]; any of them may be NULL.  With
 * KCCOT_COST_SAME (x is y) pass dy = NULL: dx receives both contributions. */
size_t kccot_pairwise_cost_bwd_workspace_bytes(int Bx, int By);
int kccot_pairwise_cost_bwd_f32(const float* g, const float* x, const float* y, int Bx, int By,
                                int64_t K, float sc, const float* h, const float* M, int T, int J,
                                unsigned flags, float* dx, float* dy, float* dh, float* dM,
                                void* ws, size_t ws_bytes, kccot_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Sinkhorn.  Replaces the loop and final cost of compute_sinkhorn (gan_utils.py:138-165) and
 * benchmark_sinkhorn (:87-121) for `nprob` independent n x n problems (one workgroup each):
 *   u = v = 0; repeat up to L times:
 *     u += eps*(log(1/n) - LSE_j((-C+u+v^T)/eps));  v += eps*(log(1/n) - LSE_i((-C+u+v^T)/eps))
 *     err = sum|u - u_prev|; stop per stop_mode once err < thresh
 *   cost = sum(exp((-C+u+v^T)/eps) * C)
 * C is [nprob,n,n].  cost_out [nprob]; nits_out is device int32 [2*nprob]: nits_out[p] = the
 * iteration count of the reference's loop (what its `actual_nits` would be), nits_out[nprob+p] = the
 * iterations the kernel actually executed -- fewer when the fp32 state became bit-for-bit periodic
 * and the remaining iterations were skipped EXACTLY (option "sinkhorn_shortcut" = 0 disables that).
 * u_hist / v_hist [nprob,L,n] receive u and v after every executed iteration (needed by the
 * backward; pass NULL for a forward-only evaluation).  pi_out [nprob,n,n] optional.
 * ------------------------------------------------------------------------------------------- */
size_t kccot_sinkhorn_workspace_bytes(int nprob, int n);
int kccot_sinkhorn_fwd_f32(const float* C, int nprob, int n, float eps, int L, int Lmin,
                           float thresh, int stop_mode, float* u_hist, float* v_hist,
                           float* cost_out, int32_t* nits_out, float* pi_out,
                           void* ws, size_t ws_bytes, kccot_stream_t stream);

/* Status of finished solves.  The multi-CU solver (128 < n <= 1024) spreads a problem over several workgroups
 * that exchange duals by polling; it is only launched when the device can hold all of them at once (CU count x
 * occupancy, queried at run time; otherwise the one-workgroup streaming solver runs), and its polling is bounded:
 * should a workgroup never show up, the solve drains within about a second, writes NaN to cost_out[p] and a
 * NEGATIVE count to nits_out[p], and the reverse sweep of that problem writes NaN gradients -- never a plausible
 * number.  This call makes that visible to a host that does not want to look for NaN: it WAITS for `stream`,
 * copies nits[0..nprob) to the host and returns 0 or KCCOT_EABORTED.  (The reference has no counterpart: its loop
 * is a host loop, gan_utils.py:151-160.)  Synchronising by design -- call it where a NaN guard would sit
 * (kernel_train.py:323), not per step. */
int kccot_sinkhorn_status(const int32_t* nits, int nprob, kccot_stream_t stream);

/* Reverse sweep through the executed iterations (what tf.GradientTape does through the unrolled
 * loop, kernel_train.py:221,252,262,289): dC_out[p] = gcost[p] * dcost[p]/dC[p].  gcost is a
 * DEVICE array [nprob]; nits is the device array written by the forward. */
int kccot_sinkhorn_bwd_f32(const float* C, const float* u_hist, const float* v_hist,
                           const int32_t* nits, int nprob, int n, float eps, int L,
                           const float* gcost, float* dC_out,
                           void* ws, size_t ws_bytes, kccot_stream_t stream);

/* The three solves of compute_sinkhorn_loss AND their combination in ONE launch each way
 * (gan_utils.py:221-225): C3 = [xy, xx, yy] as [3,n,n]; loss_out = 2*cost3[0] - cost3[1] - cost3[2]
 * is written by the last workgroup to finish.  `ticket` is one device int32 that must be ZERO on
 * entry; the kernel leaves it zero.  Backward: gloss is ONE device float (dLoss/dloss). */
int kccot_sinkhorn_divergence_fwd_f32(const float* C3, int n, float eps, int L, int Lmin, float thresh,
                                      float* u_hist, float* v_hist, float* cost3_out, int32_t* nits_out,
                                      float* loss_out, int32_t* ticket, void* ws, size_t ws_bytes,
                                      kccot_stream_t stream);
int kccot_sinkhorn_divergence_bwd_f32(const float* C3, const float* u_hist, const float* v_hist,
                                      const int32_t* nits, int n, float eps, int L, const float* gloss,
                                      float* dC3_out, void* ws, size_t ws_bytes, kccot_stream_t stream);

/* compute_sinkhorn_loss (gan_utils.py:204-227) in ONE call each way -- what a binding of the
 * reference's loss function calls.  These only sequence the stage entry points above (cost3 ->
 * sinkhorn_divergence_fwd; sinkhorn_divergence_bwd -> cost3_bwd) so that the caller pays one FFI
 * crossing per direction.
 *   forward : writes C3 [3,B,B], u_hist / v_hist [3,max(L,1),B], cost3_out [3], nits_out [6]
 *             (see kccot_sinkhorn_fwd_f32), loss_out [1].  `ticket`: one device int32, zero on entry.
 *   backward: gloss = ONE device float; C3 / u_hist / v_hist / nits as written by the forward;
 *             dfake [B,K], dh_fake, dh_real, dm_real, dm_fake [B,T,J] -- each may be NULL.
 * Workspace (both directions): kccot_sinkhorn_loss_workspace_bytes(B, K). */
size_t kccot_sinkhorn_loss_workspace_bytes(int B, int64_t K);
int kccot_sinkhorn_loss_fwd_f32(const float* real, const float* fake, int B, int64_t K, float sc,
                                const float* h_fake, const float* h_real, const float* m_real,
                                const float* m_fake, int T, int J, float eps, int L, int Lmin,
                                float thresh, unsigned flags, float* C3, float* u_hist, float* v_hist,
                                float* cost3_out, int32_t* nits_out, float* loss_out, int32_t* ticket,
                                void* ws, size_t ws_bytes, kccot_stream_t stream);
int kccot_sinkhorn_loss_bwd_f32(const float* gloss, const float* real, const float* fake, int B,
                                int64_t K, float sc, const float* h_fake, const float* h_real,
                                const float* m_real, const float* m_fake, int T, int J, float eps,
                                int L, const float* C3, const float* u_hist, const float* v_hist,
                                const int32_t* nits, float* dfake, float* dh_fake, float* dh_real,
                                float* dm_real, float* dm_fake, void* ws, size_t ws_bytes,
                                kccot_stream_t stream);

/* The same loss with the solves AND the reverse sweep in one persistent launch (the dual history stays in LDS; what
 * tf.GradientTape replays through the unrolled loop, kernel_train.py:287-289, is computed before the kernel leaves
 * the CU).  Eligible -- ask kccot_sinkhorn_fused_eligible, do not size from a rule -- when n <= "sinkhorn_fused_max_n"
 * (default 64: above that the kernel spills) and the dual history, 2 (L+1) NS floats with NS = n rounded up to the
 * kernel's lane grid (entries per lane x lanes per line), fits 144 KB of LDS; configs[0] and configs[1] are.
 * dC3_unit [3,n,n] = d loss / d C3 at dLoss = 1; the backward multiplies by the upstream scalar `gloss` (one device
 * float) while building its coefficients.  Costs, iteration counts and loss are bit-identical to the two-launch form.
 * Option "sinkhorn_fused" = 0 reports "not eligible". */
int kccot_sinkhorn_fused_eligible(int n, int L);
int kccot_sinkhorn_divergence_fused_f32(const float* C3, int n, float eps, int L, int Lmin, float thresh,
                                        float* cost3_out, int32_t* nits_out, float* loss_out, int32_t* ticket,
                                        float* dC3_unit, kccot_stream_t stream);
int kccot_pairwise_cost3_bwd_scaled_f32(const float* g3, const float* gscale, const float* real, const float* fake,
                                        int B, int64_t K, float sc, const float* h_fake, const float* h_real,
                                        const float* m_real, const float* m_fake, int T, int J,
                                        float* dfake, float* dh_fake, float* dh_real, float* dm_real,
                                        float* dm_fake, void* ws, size_t ws_bytes, kccot_stream_t stream);
int kccot_sinkhorn_loss_fused_fwd_f32(const float* real, const float* fake, int B, int64_t K, float sc,
                                      const float* h_fake, const float* h_real, const float* m_real,
                                      const float* m_fake, int T, int J, float eps, int L, int Lmin,
                                      float thresh, unsigned flags, float* C3, float* dC3_unit,
                                      float* cost3_out, int32_t* nits_out, float* loss_out, int32_t* ticket,
                                      void* ws, size_t ws_bytes, kccot_stream_t stream);
int kccot_sinkhorn_loss_fused_bwd_f32(const float* gloss, const float* dC3_unit, const float* real,
                                      const float* fake, int B, int64_t K, float sc, const float* h_fake,
                                      const float* h_real, const float* m_real, const float* m_fake, int T, int J,
                                      float* dfake, float* dh_fake, float* dh_real, float* dm_real, float* dm_fake,
                                      void* ws, size_t ws_bytes, kccot_stream_t stream);

/* Mixed Sinkhorn divergence (gan_utils.py:225): loss = 2*cost3[0] - cost3[1] - cost3[2] for
 * cost3 = [W(real,fake), W(real,real), W(fake,fake)], and its backward gcost3 = gloss*[2,-1,-1].
 * All arguments are device pointers (one launch each, no host round trip). */
int kccot_mixed_divergence_fwd_f32(const float* cost3, float* loss_out, kccot_stream_t stream);
int kccot_mixed_divergence_bwd_f32(const float* gloss, float* gcost3_out, kccot_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Martingale penalty.  Replaces scale_invariante_martingale_regularization (gan_utils.py:179-201):
 *   pM = lam * sc * sum_{t<T-1,q} | (1/B) sum_b (M[b,t+1,q]-M[b,t,q]) / (std_{b,t}(M[:,:,q]) + 1e-6) |
 * (population std).  pm_out is one device float.  Backward: dM = gpm * dpM/dM with gpm a device
 * float. */
int kccot_martingale_fwd_f32(const float* M, int B, int T, int J, float lam, float sc,
                             float* pm_out, kccot_stream_t stream);
int kccot_martingale_bwd_f32(const float* M, int B, int T, int J, float lam, float sc,
                             const float* gpm, float* dM, kccot_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * One time step of the ConvLSTM cells of the generator (the reference
 * uses tf.keras.layers.ConvLSTM2D: gan.py:50-85 encoder, gan.py:194-266 decoder), everything behind
 * the two convolutions in ONE launch: with g = gx + gh the four gate pre-activations [B,4F,HW]
 * (gate order i, f, c, o as in Keras; gx = input convolution, gh = recurrent convolution of h_prev),
 *   i = hard_sigmoid(g_i), f = hard_sigmoid(g_f), cc = tanh(g_c), o = hard_sigmoid(g_o)
 *   c = f * c_prev + i * cc;  h = o * tanh(c)            hard_sigmoid(x) = clip(0.2 x + 0.5, 0, 1)
 * c_prev, c_out, h_out: [B,F,HW].  Backward: dh, dc_out (either may be NULL = zero) -> dg [B,4F,HW]
 * (the gradient of gx AND of gh) and dc_prev; the gates are recomputed from gx + gh. */
int kccot_convlstm_cell_fwd_f32(const float* gx, const float* gh, const float* c_prev, int B, int F, int HW,
                                float* c_out, float* h_out, kccot_stream_t stream);
int kccot_convlstm_cell_bwd_f32(const float* gx, const float* gh, const float* c_prev, const float* c_out,
                                const float* dh, const float* dc_out, int B, int F, int HW, float* dg,
                                float* dc_prev, kccot_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * LayerNormalization over the CHANNELS of an NCHW tensor (the
 * tf.keras.layers.LayerNormalization(axis=-1) behind every ConvLSTM2D / Conv2DTranspose of the
 * generator, gan.py:60-85,216-266, applied there to channels-last data): for every (n, pixel)
 *   y[n,c,p] = (x[n,c,p] - mean) * rstd * gamma[c] + beta[c],   mean / biased variance over c,
 *   rstd = 1 / sqrt(var + eps)
 * on the NCHW layout the convolutions produce (no permute, no copy).  x, y: [N,C,HW]; gamma, beta: [C];
 * mean, rstd: [N,HW] (saved for the backward).  Backward: dx [N,C,HW]; the parameter gradients come as
 * per-chunk partial sums partials[chunk][2][C] (dgamma, dbeta) that the caller adds up in chunk order
 * (deterministic); kccot_channel_layernorm_chunks(N, C, HW) = number of chunks. */
int kccot_channel_layernorm_chunks(int N, int C, int HW);
int kccot_channel_layernorm_fwd_f32(const float* x, const float* gamma, const float* beta, int N, int C, int HW,
                                    float eps, float* y, float* mean, float* rstd, kccot_stream_t stream);
int kccot_channel_layernorm_bwd_f32(const float* dy, const float* x, const float* gamma, const float* mean,
                                    const float* rstd, int N, int C, int HW, float* dx, float* partials,
                                    kccot_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * EXTENSION, no reference behaviour (BASELINE.json north_star names a "batch-vs-batch Gaussian
 * kernel / MMD matrix"; the reference only imports sklearn's rbf_kernel, data_utils.py:16, and never
 * calls it).  Defined with sklearn.metrics.pairwise.rbf_kernel semantics: K = exp(-gamma * D) on
 * the plain squared distances D3 = [xy, xx, yy] ([3,B,B], from kccot_pairwise_cost3_f32 with sc = 1
 * and no features); mmd_out = mean(Kxx) + mean(Kyy) - 2 mean(Kxy).  K3_out [3,B,B] optional.
 * ------------------------------------------------------------------------------------------- */
int kccot_rbf_mmd_f32(const float* D3, int B, float gamma, float* K3_out, float* mmd_out,
                      kccot_stream_t stream);
/* Backward of the estimate w.r.t. the distances: gD3 [3,B,B] = gmmd * d mmd / d D3 from the kernel
 * matrices K3 of the forward; gmmd is ONE device float.  Feed gD3 to kccot_pairwise_cost3_bwd_f32
 * (sc = 1, no features) for the gradient w.r.t. `fake`. */
int kccot_rbf_mmd_bwd_f32(const float* K3, int B, float gamma, const float* gmmd, float* gD3,
                          kccot_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Kernel smoothing.  Replaces KernelSmoothing.temporal_convolution (data_utils.py:503-521,
 * axes = KCCOT_SMOOTH_T) and gaussian_convolution3D (:552-582, axes = T|H|W) on the native
 * [B,H,T,W,C] layout: normalised (2r+1)-tap Gaussian exp(-d^2/(2 sigma^2)) along each selected
 * axis (the 7x7x7 kernel of data_utils.py:493-501 is the outer product of the 1-D one), REFLECT
 * borders, then division by the maximum of the WHOLE smoothed tensor (data_utils.py:520,573,581).
 *   out       [B,H,T,W,C]
 *   max_inout one device float: written with the tensor maximum.  With
 *             KCCOT_SMOOTH_EXTERNAL_MAX it is READ instead (the batch-sharded caller has
 *             all-reduced(MAX) it across ranks) and `out` is divided by it.
 *   KCCOT_SMOOTH_NO_DIVIDE leaves `out` un-normalised and only writes the local maximum
 *             (first phase of the sharded protocol).
 * Backward: din = d(out)/d(in)^T gout, including the arg-max path of the global maximum.
 * ------------------------------------------------------------------------------------------- */
#define KCCOT_SMOOTH_T 1u
#define KCCOT_SMOOTH_H 2u
#define KCCOT_SMOOTH_W 4u
#define KCCOT_SMOOTH_NO_DIVIDE 16u
#define KCCOT_SMOOTH_EXTERNAL_MAX 32u
#define KCCOT_SMOOTH_STATS_ONLY 64u      /* kccot_smooth_bwd_sharded_f32 only */
#define KCCOT_SMOOTH_EXTERNAL_STATS 128u /* kccot_smooth_bwd_sharded_f32 only */
size_t kccot_smooth_workspace_bytes(int B, int H, int T, int W, int C);
int kccot_smooth_fwd_f32(const float* in, int B, int H, int T, int W, int C, float sigma,
                         int radius, unsigned axes_flags, float* out, float* max_inout,
                         void* ws, size_t ws_bytes, kccot_stream_t stream);
int kccot_smooth_bwd_f32(const float* gout, const float* out, const float* max_in,
                         int B, int H, int T, int W, int C, float sigma, int radius,
                         unsigned axes_flags, float* din,
                         void* ws, size_t ws_bytes, kccot_stream_t stream);
/* Backward for the batch-sharded caller (every rank holds B/G samples, `max_in` is the all-reduced(MAX) global maximum
 * the forward divided by).  The adjoint of the division by the GLOBAL maximum needs two sums over the whole batch,
 * stats = {sum(gout * out), number of elements with out == 1}:
 *   flags | KCCOT_SMOOTH_STATS_ONLY      writes this rank's two sums to stats_inout[0..1] (din is not touched);
 *   the caller all-reduces(SUM) them;
 *   flags | KCCOT_SMOOTH_EXTERNAL_STATS  reads the global sums from stats_inout and writes din. */
int kccot_smooth_bwd_sharded_f32(const float* gout, const float* out, const float* max_in, float* stats_inout,
                                 int B, int H, int T, int W, int C, float sigma, int radius,
                                 unsigned axes_flags, float* din,
                                 void* ws, size_t ws_bytes, kccot_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* KCCOT_H */
