// compute_sinkhorn_loss (gan_utils.py:204-227) as ONE host call each way: the launch sequence
// cost assembly -> three Sinkhorn solves + combination (forward) and reverse sweep -> cost backward
// (backward) is issued from C, so a caller pays one FFI crossing and one workspace per direction
// instead of one per stage.  No new kernels: these entry points only sequence the stage functions.
#include "common.h"

namespace kccot {
static size_t up256(size_t b) { return (b + 255) & ~(size_t)255; }
static size_t max3(size_t a, size_t b, size_t c) { return a > b ? (a > c ? a : c) : (b > c ? b : c); }
}  // namespace kccot
using namespace kccot;

extern "C" size_t kccot_sinkhorn_loss_workspace_bytes(int B, int64_t K) {
    if (B <= 0 || K <= 0) return 0;
    // forward: cost stage | Sinkhorn stage;  backward: dC3 [3,B,B] + 3 floats, then Sinkhorn | cost-backward stage
    const size_t stage = max3(kccot_pairwise_cost3_workspace_bytes(B, K), kccot_sinkhorn_workspace_bytes(3, B),
                              kccot_pairwise_cost3_bwd_workspace_bytes(B, K));
    return up256((size_t)3 * B * B * sizeof(float)) + 256 + up256(stage);
}

extern "C" int kccot_sinkhorn_loss_fwd_f32(const float* real, const float* fake, int B, int64_t K, float sc,
                                           const float* h_fake, const float* h_real, const float* m_real,
                                           const float* m_fake, int T, int J, float eps, int L, int Lmin,
                                           float thresh, unsigned flags, float* C3, float* u_hist, float* v_hist,
                                           float* cost3_out, int32_t* nits_out, float* loss_out, int32_t* ticket,
                                           void* ws, size_t ws_bytes, kccot_stream_t stream) {
    if (!C3 || !cost3_out || !nits_out || !loss_out || !ticket)
        return fail(KCCOT_EINVAL, "sinkhorn_loss_fwd: null output pointer");
    if (ws_bytes < kccot_sinkhorn_loss_workspace_bytes(B, K) || (!ws && ws_bytes))
        return fail(KCCOT_EWORKSPACE, "sinkhorn_loss_fwd: workspace %zu < %zu bytes", ws_bytes,
                    kccot_sinkhorn_loss_workspace_bytes(B, K));
    int rc = kccot_pairwise_cost3_f32(real, fake, B, K, sc, h_fake, h_real, m_real, m_fake, T, J, flags, C3, ws, ws_bytes,
                                      stream);
    if (rc) return rc;
    return kccot_sinkhorn_divergence_fwd_f32(C3, B, eps, L, Lmin, thresh, u_hist, v_hist, cost3_out, nits_out, loss_out,
                                             ticket, ws, ws_bytes, stream);
}

extern "C" int kccot_sinkhorn_loss_bwd_f32(const float* gloss, const float* real, const float* fake, int B, int64_t K,
                                           float sc, const float* h_fake, const float* h_real, const float* m_real,
                                           const float* m_fake, int T, int J, float eps, int L, const float* C3,
                                           const float* u_hist, const float* v_hist, const int32_t* nits,
                                           float* dfake, float* dh_fake, float* dh_real, float* dm_real, float* dm_fake,
                                           void* ws, size_t ws_bytes, kccot_stream_t stream) {
    if (!gloss || !C3 || !u_hist || !v_hist || !nits) return fail(KCCOT_EINVAL, "sinkhorn_loss_bwd: null pointer");
    if (!ws || ws_bytes < kccot_sinkhorn_loss_workspace_bytes(B, K))
        return fail(KCCOT_EWORKSPACE, "sinkhorn_loss_bwd: workspace %zu < %zu bytes", ws_bytes,
                    kccot_sinkhorn_loss_workspace_bytes(B, K));
    char* base = static_cast<char*>(ws);
    float* dC3 = reinterpret_cast<float*>(base);
    const size_t off_gc = up256((size_t)3 * B * B * sizeof(float));
    float* gc = reinterpret_cast<float*>(base + off_gc);
    void* stage = base + off_gc + 256;
    const size_t stage_bytes = ws_bytes - off_gc - 256;
    int rc;
    if (kccot_sinkhorn_workspace_bytes(3, B) > 0) {
        // streaming solver (n > 128): weights {2,-1,-1} * gloss first, then the generic reverse sweep
        rc = kccot_mixed_divergence_bwd_f32(gloss, gc, stream);
        if (rc) return rc;
        rc = kccot_sinkhorn_bwd_f32(C3, u_hist, v_hist, nits, 3, B, eps, L, gc, dC3, stage, stage_bytes, stream);
    } else {
        rc = kccot_sinkhorn_divergence_bwd_f32(C3, u_hist, v_hist, nits, B, eps, L, gloss, dC3, stage, stage_bytes, stream);
    }
    if (rc) return rc;
    return kccot_pairwise_cost3_bwd_f32(dC3, real, fake, B, K, sc, h_fake, h_real, m_real, m_fake, T, J, dfake, dh_fake,
                                        dh_real, dm_real, dm_fake, stage, stage_bytes, stream);
}

// ---- the same with the fused solve + sweep (kccot_sinkhorn_divergence_fused_f32) ----------------------------------
// forward = cost assembly -> ONE launch (three solves, combination, reverse sweep at dLoss = 1): C3 is scratch for the
// caller, dC3_unit [3,B,B] is what the backward needs; backward = coefficient build (x gloss) -> video gradient.
extern "C" int kccot_sinkhorn_loss_fused_fwd_f32(const float* real, const float* fake, int B, int64_t K, float sc,
                                                 const float* h_fake, const float* h_real, const float* m_real,
                                                 const float* m_fake, int T, int J, float eps, int L, int Lmin,
                                                 float thresh, unsigned flags, float* C3, float* dC3_unit,
                                                 float* cost3_out, int32_t* nits_out, float* loss_out, int32_t* ticket,
                                                 void* ws, size_t ws_bytes, kccot_stream_t stream) {
    if (!C3 || !dC3_unit || !cost3_out || !nits_out || !loss_out || !ticket)
        return fail(KCCOT_EINVAL, "sinkhorn_loss_fused_fwd: null output pointer");
    if (ws_bytes < kccot_sinkhorn_loss_workspace_bytes(B, K) || (!ws && ws_bytes))
        return fail(KCCOT_EWORKSPACE, "sinkhorn_loss_fused_fwd: workspace %zu < %zu bytes", ws_bytes,
                    kccot_sinkhorn_loss_workspace_bytes(B, K));
    int rc = kccot_pairwise_cost3_f32(real, fake, B, K, sc, h_fake, h_real, m_real, m_fake, T, J, flags, C3, ws, ws_bytes,
                                      stream);
    if (rc) return rc;
    return kccot_sinkhorn_divergence_fused_f32(C3, B, eps, L, Lmin, thresh, cost3_out, nits_out, loss_out, ticket, dC3_unit,
                                               stream);
}

extern "C" int kccot_sinkhorn_loss_fused_bwd_f32(const float* gloss, const float* dC3_unit, const float* real,
                                                 const float* fake, int B, int64_t K, float sc, const float* h_fake,
                                                 const float* h_real, const float* m_real, const float* m_fake, int T, int J,
                                                 float* dfake, float* dh_fake, float* dh_real, float* dm_real, float* dm_fake,
                                                 void* ws, size_t ws_bytes, kccot_stream_t stream) {
    if (!gloss || !dC3_unit) return fail(KCCOT_EINVAL, "sinkhorn_loss_fused_bwd: null pointer");
    return kccot_pairwise_cost3_bwd_scaled_f32(dC3_unit, gloss, real, fake, B, K, sc, h_fake, h_real, m_real, m_fake, T, J,
                                               dfake, dh_fake, dh_real, dm_real, dm_fake, ws, ws_bytes, stream);
}
