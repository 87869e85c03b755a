"""Batch-sharded evaluation of compute_sinkhorn_loss over the GPUs of one node (one process per
GPU, torch.distributed over RCCL/xGMI).  The reference has no distributed code at all
(SURVEY.md section 5); this is the new design of SURVEY.md section 8(e):

  1. every rank owns B/G samples of real / fake and of the four feature tensors;
  2. ALL-GATHER the video shards and the (KB-sized) features -> every rank holds [B,K] of both;
  3. rank g builds the ROW BLOCKS C_xy[I_g,:], C_xx[I_g,:], C_yy[I_g,:] ([B/G, B] each) -- on the matrix pipe
     (csrc/cost_rows.hip) when it owns 32 or 64 samples and B % 128 == 0: the Gram row block [X_I ; E_I][X ; E]^T plus the
     all-gathered row norms x.x, e.e, x.e (3 doubles per sample); on the direct-difference kernel otherwise;
  4. ALL-GATHER the row blocks (3*B*B*4 bytes in total) -> the full cost matrices, replicated;
  5. every rank runs the identical (deterministic) Sinkhorn forward and reverse sweep: the loss
     and dLoss/dC are bitwise the same everywhere, no communication;
  6. rank g forms the gradients of ITS samples from the replicated dLoss/dC and the gathered
     videos (kccot_pairwise_cost3_bwd_rows_f32): NO reduce-scatter of video-sized gradients.

The returned loss is the GLOBAL-batch loss (identical on every rank).  Parameter gradients that
flow back through a rank's local samples are therefore partial sums: combine them with an
all-reduce SUM (not the mean DistributedDataParallel applies by default).

The arithmetic is done by the HIP library (``HipOps``).  ``ops`` is injectable so that the
sharding / collective logic can be exercised over gloo on a CPU-only box by the tests (which plug
in the CPU oracle there); the product never does.
"""
import os

import torch
import torch.distributed as dist

from . import _lib
from . import gan_utils
from ._lib import lib, check, ptr, stream_of, workspace

_THRESH = 10 ** (-2)
_LMIN = 100

last_info = {}   # executed Sinkhorn iteration counts of the latest sharded evaluation (device tensor)


class _Phases:
    """Optional per-phase device timing of the sharded step (bench.py's N > 1 blocks: gathers, row block, Sinkhorn, gradient
    rows -- DESIGN.md section 6's table, measured).  Off by default: `phase_timing(True)` makes the autograd functions below
    drop a HIP event on the current stream at every phase boundary (a collective issued without async_op makes the current
    stream wait for it, so the interval that ENDS at a boundary holds the phase's transfers and kernels in stream order);
    `phase_ms()` synchronises and returns {phase: ms since the previous boundary}, summed over the steps recorded since the
    last call, plus "steps"."""
    enabled = False
    marks = []

    @classmethod
    def mark(cls, name):
        if cls.enabled and torch.cuda.is_available():
            e = torch.cuda.Event(enable_timing=True)
            e.record()
            cls.marks.append((name, e))


def phase_timing(on):
    _Phases.enabled = bool(on)
    _Phases.marks = []


def phase_ms():
    torch.cuda.synchronize()
    out, steps, prev = {}, 0, None
    for name, e in _Phases.marks:
        if name == "start":
            steps += 1
        elif prev is not None:
            out[name] = out.get(name, 0.0) + prev.elapsed_time(e)
        prev = e
    _Phases.marks = []
    out["steps"] = steps
    return out


_mark = _Phases.mark


class HipOps:
    """The four device operations of the sharded path, through the C-ABI."""

    @staticmethod
    def cost_rows(x_rows, y_full, h_rows, M_full, sc):
        Bx, K = x_rows.shape
        By = y_full.shape[0]
        T, J = h_rows.shape[1], h_rows.shape[2]
        C = _lib.empty((Bx, By), torch.float32, x_rows.device)
        ws, wsb = workspace(lib.kccot_pairwise_cost_workspace_bytes(Bx, By, K), x_rows)
        # row blocks pair sample i with ALL samples j, so the pair-difference stack of the fused
        # single-GPU kernel does not apply; the direct-difference kernel keeps the small distances of
        # the GAN regime exact (a rank only builds B/G rows, the VALU rate is ample)
        check(lib.kccot_pairwise_cost_f32(ptr(x_rows), ptr(y_full), Bx, By, K, sc, ptr(h_rows), ptr(M_full), None, None,
                                          T, J, _lib.COST_FORCE_DIRECT, ptr(C), ws, wsb, stream_of(x_rows)),
              "pairwise_cost")
        return C

    @staticmethod
    def rows_gram_supported(row_count, B, K):
        """The row block on the matrix pipe (kccot_pairwise_cost3_rows_gram_f32): 32 or 64 rows per rank, B % 128 == 0.
        Depends on (row_count, B, K) and the process-wide options only, so all ranks agree."""
        return bool(lib.kccot_pairwise_cost3_rows_gram_supported(int(row_count), int(B), int(K)))

    @staticmethod
    def row_norms(real_l, fake_l):
        """x.x, e.e, x.e (e = fake - real) of this rank's rows, [Bl,3] float64 -- the column-side diagonal Gram entries every
        other rank's row block needs (all-gathered by the caller: 24 bytes per sample)."""
        Bl, K = real_l.shape
        out = _lib.empty((Bl, 3), torch.float64, real_l.device)
        ws, wsb = workspace(lib.kccot_row_norms_workspace_bytes(Bl), real_l)
        check(lib.kccot_row_norms_f64(ptr(real_l), ptr(fake_l), Bl, K, _lib.ptr_f64(out), ws, wsb, stream_of(real_l)), "row_norms")
        return out

    @staticmethod
    def cost3_rows(real, fake, h_fake, h_real, m_real, m_fake, sc, row_begin, row_count, norms=None):
        """Row blocks [3, row_count, B] of (xy, xx, yy): on the matrix pipe when the gathered `norms` [B,3] are given
        (kccot_pairwise_cost3_rows_gram_f32), else one launch of the direct-difference kernel
        (kccot_pairwise_cost3_rows_f32)."""
        B, K = real.shape
        T, J = h_fake.shape[1], h_fake.shape[2]
        out = _lib.empty((3, row_count, B), torch.float32, real.device)
        if norms is not None:
            ws, wsb = workspace(lib.kccot_pairwise_cost3_rows_gram_workspace_bytes(row_count, B, K), real)
            check(lib.kccot_pairwise_cost3_rows_gram_f32(ptr(real), ptr(fake), B, K, sc, ptr(h_fake), ptr(h_real), ptr(m_real),
                                                         ptr(m_fake), T, J, row_begin, row_count, _lib.ptr_f64(norms.contiguous()),
                                                         ptr(out), ws, wsb, stream_of(real)), "pairwise_cost3_rows_gram")
            return out
        ws, wsb = workspace(lib.kccot_pairwise_cost3_rows_workspace_bytes(row_count, B, K), real)
        check(lib.kccot_pairwise_cost3_rows_f32(ptr(real), ptr(fake), B, K, sc, ptr(h_fake), ptr(h_real), ptr(m_real),
                                                ptr(m_fake), T, J, row_begin, row_count, ptr(out), ws, wsb,
                                                stream_of(real)), "pairwise_cost3_rows")
        return out

    @staticmethod
    def rows_gram_sums(real_c, fake_c, row_begin, row_count, gsum, accumulate):
        """fp64 Gram sums of the rank's row block over ONE column range (real_c, fake_c: [B, Kc] of all samples), written to
        or added to `gsum` (kccot_pairwise_cost3_rows_gram_sums_f64)."""
        B, Kc = real_c.shape
        ws, wsb = workspace(lib.kccot_pairwise_cost3_rows_gram_workspace_bytes(row_count, B, Kc), real_c)
        check(lib.kccot_pairwise_cost3_rows_gram_sums_f64(ptr(real_c), ptr(fake_c), B, Kc, row_begin, row_count, _lib.ptr_f64(gsum),
                                                          1 if accumulate else 0, ws, wsb, stream_of(real_c)), "rows_gram_sums")

    @staticmethod
    def rows_gram_from_sums(gsum, B, h_fake, h_real, m_real, m_fake, sc, row_begin, row_count, norms):
        T, J = h_fake.shape[1], h_fake.shape[2]
        out = _lib.empty((3, row_count, B), torch.float32, gsum.device)
        check(lib.kccot_pairwise_cost3_rows_gram_from_sums_f32(_lib.ptr_f64(gsum), B, sc, ptr(h_fake), ptr(h_real), ptr(m_real), ptr(m_fake),
                                                               T, J, row_begin, row_count, _lib.ptr_f64(norms.contiguous()), ptr(out),
                                                               stream_of(gsum)), "rows_gram_from_sums")
        return out

    @staticmethod
    def replicate_costs(B, K):
        """Batches of at most 64: every rank assembles the WHOLE [3,B,B] with the one-pass MFMA kernels (31 us at
        configs[1]) instead of its row block on the direct kernel (46-100 us) followed by another all-gather --
        cheaper, one collective fewer, and the same pair-difference arithmetic as the single-GPU loss.  Larger batches
        keep the row blocks (cost ~ B^2 K / G).  KCCOT_DIST_ROW_BLOCKS=1 forces the row-block protocol (tests, A/B).
        The choice depends on (B, K) only, so all ranks agree."""
        return B <= 64 and K % 4 == 0 and K >= 256 and os.environ.get("KCCOT_DIST_ROW_BLOCKS") != "1"

    @staticmethod
    def cost3_full(real, fake, h_fake, h_real, m_real, m_fake, sc):
        B, K = real.shape
        T, J = h_fake.shape[1], h_fake.shape[2]
        C3 = _lib.empty((3, B, B), torch.float32, real.device)
        ws, wsb = workspace(lib.kccot_pairwise_cost3_workspace_bytes(B, K), real)
        check(lib.kccot_pairwise_cost3_f32(ptr(real), ptr(fake), B, K, sc, ptr(h_fake), ptr(h_real), ptr(m_real),
                                           ptr(m_fake), T, J, 0, ptr(C3), ws, wsb, stream_of(real)), "pairwise_cost3")
        return C3

    @staticmethod
    def sinkhorn3_fwd(C3, eps, L):
        nprob, n, _ = C3.shape
        dev = C3.device
        Lh = max(int(L), 1)
        u_hist = _lib.empty((nprob, Lh, n), torch.float32, dev)
        v_hist = _lib.empty((nprob, Lh, n), torch.float32, dev)
        cost = _lib.empty((nprob,), torch.float32, dev)
        nits = _lib.empty((2 * nprob,), torch.int32, dev)
        ws, wsb = workspace(lib.kccot_sinkhorn_workspace_bytes(nprob, n), C3)
        check(lib.kccot_sinkhorn_fwd_f32(ptr(C3), nprob, n, float(eps), int(L), _LMIN, _THRESH, _lib.STOP_COUNT,
                                         ptr(u_hist), ptr(v_hist), ptr(cost), ptr(nits), None, ws, wsb,
                                         stream_of(C3)), "sinkhorn_fwd")
        return cost, (C3, u_hist, v_hist, nits, float(eps), Lh)

    @staticmethod
    def sinkhorn3_bwd(saved, gcost3):
        C3, u_hist, v_hist, nits, eps, Lh = saved
        nprob, n, _ = C3.shape
        dC3 = _lib.empty_like(C3)
        ws, wsb = workspace(lib.kccot_sinkhorn_workspace_bytes(nprob, n), C3)
        check(lib.kccot_sinkhorn_bwd_f32(ptr(C3), ptr(u_hist), ptr(v_hist), ptr(nits), nprob, n, eps, Lh,
                                         ptr(gcost3.contiguous()), ptr(dC3), ws, wsb, stream_of(C3)), "sinkhorn_bwd")
        return dC3

    @staticmethod
    def divergence_fwd(C3, eps, L):
        """The three solves AND 2 xy - xx - yy in one launch (n <= 128; larger n: two launches inside the library)."""
        from .gan_utils import _ticket
        _, n, _ = C3.shape
        dev = C3.device
        Lh = max(int(L), 1)
        u_hist = _lib.empty((3, Lh, n), torch.float32, dev)
        v_hist = _lib.empty((3, Lh, n), torch.float32, dev)
        small = _lib.empty((4,), torch.float32, dev)             # cost3 | loss
        nits = _lib.empty((6,), torch.int32, dev)
        ws, wsb = workspace(lib.kccot_sinkhorn_workspace_bytes(3, n), C3)
        check(lib.kccot_sinkhorn_divergence_fwd_f32(ptr(C3), n, float(eps), int(L), _LMIN, _THRESH, ptr(u_hist), ptr(v_hist),
                                                    ptr(small), ptr(nits), ptr(small[3:]), ptr(_ticket(dev)), ws, wsb,
                                                    stream_of(C3)), "sinkhorn_divergence_fwd")
        return small[3:].reshape(()), (C3, u_hist, v_hist, nits, float(eps), Lh)

    @staticmethod
    def divergence_bwd(saved, g):
        C3, u_hist, v_hist, nits, eps, Lh = saved
        _, n, _ = C3.shape
        g = g.reshape(1).contiguous().float()
        dC3 = _lib.empty_like(C3)
        ws, wsb = workspace(lib.kccot_sinkhorn_workspace_bytes(3, n), C3)
        if n > 128:      # streaming / cooperative solvers: weights first, then the generic reverse sweep
            gc = _lib.empty((3,), torch.float32, g.device)
            check(lib.kccot_mixed_divergence_bwd_f32(ptr(g), ptr(gc), stream_of(g)), "mixed_divergence_bwd")
            check(lib.kccot_sinkhorn_bwd_f32(ptr(C3), ptr(u_hist), ptr(v_hist), ptr(nits), 3, n, eps, Lh, ptr(gc), ptr(dC3),
                                             ws, wsb, stream_of(C3)), "sinkhorn_bwd")
        else:
            check(lib.kccot_sinkhorn_divergence_bwd_f32(ptr(C3), ptr(u_hist), ptr(v_hist), ptr(nits), n, eps, Lh, ptr(g),
                                                        ptr(dC3), ws, wsb, stream_of(C3)), "sinkhorn_divergence_bwd")
        return dC3

    @staticmethod
    def cost3_bwd_rows(dC3, real, fake, h_fake, h_real, m_real, m_fake, sc, row_begin, row_count):
        B, K = real.shape
        T, J = h_fake.shape[1], h_fake.shape[2]
        dev = real.device
        dfake = _lib.empty((row_count, K), torch.float32, dev)
        dhf, dhr, dmr, dmf = (_lib.empty((row_count, T, J), torch.float32, dev) for _ in range(4))
        ws, wsb = workspace(lib.kccot_pairwise_cost3_bwd_workspace_bytes(B, K), real)
        check(lib.kccot_pairwise_cost3_bwd_rows_f32(ptr(dC3), ptr(real), ptr(fake), B, K, sc, ptr(h_fake), ptr(h_real),
                                                    ptr(m_real), ptr(m_fake), T, J, row_begin, row_count, ptr(dfake),
                                                    ptr(dhf), ptr(dhr), ptr(dmr), ptr(dmf), ws, wsb, stream_of(real)),
              "pairwise_cost3_bwd_rows")
        return dfake, dhf, dhr, dmr, dmf


# ---- contraction-sharded protocol ("ksplit") ------------------------------------------------------------------
# Instead of gathering the whole batch on every rank, the [B/G, K] shards are all-to-all'ed into [B, K/G] slices:
# every rank holds ALL samples but 1/G of the features.  The Gram kernels split K anyway, so rank g forms the fp64
# Gram sums of its slice (KCCOT_COST_GRAM_SUMS_ONLY), the 80 KB (B <= 64) of sums are all-reduced(SUM), and the
# finalize step (KCCOT_COST_FROM_GRAM_SUMS) gives every rank the full cost matrices -- the same arithmetic as one GPU,
# with the K-chunks summed in a different grouping.  Backward: the video gradient of ALL samples on the rank's slice,
# all-to-all back.  7/8 of a shard leaves a rank per tensor instead of 7 shards arriving, and the two HBM-heavy
# kernels shrink with G.  Opt-in (protocol="ksplit" / KCCOT_DIST_PROTOCOL=ksplit) until it has been timed on a
# multi-GPU node; needs K % G == 0, K/G % 4 == 0, K/G >= 256 and a Gram path for B (B <= 64 or B % 128 == 0).
def ksplit_supported(B, K, world):
    if world < 1 or K % world:
        return False
    Ks = K // world
    if Ks % 4 or Ks < 256:
        return False
    import ctypes
    off, cnt = ctypes.c_size_t(0), ctypes.c_size_t(0)
    check(lib.kccot_pairwise_cost3_gram_sums_span(B, Ks, ctypes.byref(off), ctypes.byref(cnt)), "gram_sums_span")
    return cnt.value > 0


def all_to_all_slices(t, group=None):
    """[B/G, K] shard -> [B, K/G] slice: rows in global sample order, columns = this rank's K-range."""
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    Bl, K = t.shape
    Ks = K // world
    if dist.get_backend(group) != "nccl":        # gloo has no all_to_all: rehearsal through an all-gather
        return t.contiguous() if world == 1 else all_gather_cat(t, group)[:, rank * Ks:(rank + 1) * Ks].contiguous()
    send = t.reshape(Bl, world, Ks).transpose(0, 1).contiguous()          # [G, Bl, Ks]: chunk g goes to rank g
    recv = torch.empty_like(send)
    dist.all_to_all_single(recv, send, group=group)
    return recv.reshape(world * Bl, Ks)


def all_to_all_rows(t, group=None):
    """Inverse of all_to_all_slices: [B, K/G] slice (all samples) -> [B/G, K] rows of this rank's samples."""
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    B, Ks = t.shape
    Bl = B // world
    if dist.get_backend(group) != "nccl":
        if world == 1:
            return t.contiguous()
        full = all_gather_cat(t.contiguous(), group).reshape(world, B, Ks)   # [source rank = K-range][sample][Ks]
        return full[:, rank * Bl:(rank + 1) * Bl].transpose(0, 1).reshape(Bl, world * Ks).contiguous()
    send = t.reshape(world, Bl, Ks).contiguous()                          # chunk g = the rows of rank g's samples
    recv = torch.empty_like(send)
    dist.all_to_all_single(recv, send, group=group)                       # recv[g] = K-range g of my samples
    return recv.transpose(0, 1).reshape(Bl, world * Ks).contiguous()


def _all_reduce_sum(t, group):
    if dist.get_world_size(group) == 1:
        return
    if dist.get_backend(group) == "gloo" and t.is_cuda:
        h = t.cpu()
        dist.all_reduce(h, op=dist.ReduceOp.SUM, group=group)
        t.copy_(h)
    else:
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)


class _KSplitLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, real_l, fake_l, h_fake_l, h_real_l, m_real_l, m_fake_l, sc, eps, L, group):
        import ctypes
        rank, world = dist.get_rank(group), dist.get_world_size(group)
        Bl = real_l.shape[0]
        B = Bl * world
        _mark("start")
        real_s = all_to_all_slices(real_l, group)
        fake_s = all_to_all_slices(fake_l, group)
        Ks = real_s.shape[1]
        feats = all_gather_cat(torch.stack([h_fake_l, h_real_l, m_real_l, m_fake_l], dim=1), group)
        h_fake, h_real, m_real, m_fake = (feats[:, i].contiguous() for i in range(4))
        _mark("exchange_inputs")
        T, J = h_fake.shape[1], h_fake.shape[2]
        dev = real_s.device
        # a workspace of its own: the Gram sums must survive between the two calls (the shared scratch is reused)
        wsb = int(lib.kccot_pairwise_cost3_workspace_bytes(B, Ks))
        ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
        off, cnt = ctypes.c_size_t(0), ctypes.c_size_t(0)
        check(lib.kccot_pairwise_cost3_gram_sums_span(B, Ks, ctypes.byref(off), ctypes.byref(cnt)), "gram_sums_span")
        if cnt.value == 0:
            raise NotImplementedError("ksplit protocol: no Gram path for B=%d, K/G=%d" % (B, Ks))
        C3 = _lib.empty((3, B, B), torch.float32, dev)
        args = (ptr(real_s), ptr(fake_s), B, Ks, sc, ptr(h_fake), ptr(h_real), ptr(m_real), ptr(m_fake), T, J)
        check(lib.kccot_pairwise_cost3_f32(*args, _lib.COST_GRAM_SUMS_ONLY, ptr(C3), ws.data_ptr(), wsb, stream_of(real_s)),
              "pairwise_cost3(gram sums)")
        gsum = ws[off.value:off.value + 8 * cnt.value].view(torch.float64)
        _mark("cost_gram_sums")
        _all_reduce_sum(gsum, group)
        _mark("exchange_costs")
        check(lib.kccot_pairwise_cost3_f32(*args, _lib.COST_FROM_GRAM_SUMS, ptr(C3), ws.data_ptr(), wsb, stream_of(real_s)),
              "pairwise_cost3(from gram sums)")
        _mark("cost_finalize")
        loss, saved = HipOps.divergence_fwd(C3, eps, L)
        _mark("sinkhorn_fwd")
        last_info["nits"], last_info["nits_executed"] = saved[3][:3], saved[3][3:]
        gan_utils.last_info["compute_sinkhorn_loss"] = saved[3][:3]      # raise_if_solver_aborted() covers the sharded loss too
        ctx.saved_state = (saved, real_s, fake_s, h_fake, h_real, m_real, m_fake)
        ctx.cfg = (sc, rank * Bl, Bl, group)
        return loss

    @staticmethod
    def backward(ctx, g):
        saved, real_s, fake_s, h_fake, h_real, m_real, m_fake = ctx.saved_state
        sc, row_begin, Bl, group = ctx.cfg
        if ctx.needs_input_grad[0]:
            raise NotImplementedError("the loss path never differentiates w.r.t. real (kernel_train.py:252,289)")
        _mark("between_fwd_and_bwd")
        dC3 = HipOps.divergence_bwd(saved, g.reshape(()))
        _mark("sinkhorn_bwd")
        B, Ks = real_s.shape
        T, J = h_fake.shape[1], h_fake.shape[2]
        # video gradient of ALL samples on this rank's K-slice, then back to the sample-sharded layout
        dfake_s = _lib.empty((B, Ks), torch.float32, real_s.device)
        ws, wsb = workspace(lib.kccot_pairwise_cost3_bwd_workspace_bytes(B, Ks), real_s)
        check(lib.kccot_pairwise_cost3_bwd_f32(ptr(dC3), ptr(real_s), ptr(fake_s), B, Ks, sc, None, None, None, None, 1, 1,
                                               ptr(dfake_s), None, None, None, None, ws, wsb, stream_of(real_s)),
              "pairwise_cost3_bwd")
        _mark("gradient")
        dfake = all_to_all_rows(dfake_s, group)
        _mark("exchange_gradient")
        # feature gradients of this rank's samples (KB-sized products of dC3 with the gathered features)
        dhf, dhr, dmr, dmf = (_lib.empty((Bl, T, J), torch.float32, real_s.device) for _ in range(4))
        check(lib.kccot_pairwise_cost3_bwd_rows_f32(ptr(dC3), ptr(real_s), ptr(fake_s), B, Ks, sc, ptr(h_fake), ptr(h_real),
                                                    ptr(m_real), ptr(m_fake), T, J, row_begin, Bl, None, ptr(dhf), ptr(dhr),
                                                    ptr(dmr), ptr(dmf), None, 0, stream_of(real_s)), "pairwise_cost3_bwd_rows")
        _mark("gradient")
        return None, dfake, dhf, dhr, dmr, dmf, None, None, None, None


def all_gather_cat(t, group=None):
    """Concatenate the ranks' equally shaped tensors along dim 0.  RCCL handles device tensors
    directly; the gloo backend (CPU rehearsal) is staged through host memory."""
    world = dist.get_world_size(group)
    if world == 1:
        return t
    t = t.contiguous()
    if dist.get_backend(group) == "gloo" and t.is_cuda:
        parts = [torch.empty(t.shape, dtype=t.dtype) for _ in range(world)]
        dist.all_gather(parts, t.cpu(), group=group)
        return torch.cat(parts, 0).to(t.device)
    out = torch.empty((world * t.shape[0],) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
    dist.all_gather_into_tensor(out, t, group=group)
    return out


def gather_chunk_bounds(K, nchunks):
    """Column ranges of the chunked video all-gather (KCCOT_DIST_GATHER_CHUNKS): `nchunks` ranges of whole 32-column load
    granules (the last one takes the remainder), every range at least 256 columns wide.  A function of (K, nchunks) only, so
    all ranks cut alike."""
    gran = (K + 31) // 32
    n = max(1, min(int(nchunks), gran // 8))
    per = (gran + n - 1) // n
    bounds, a = [], 0
    while a < K:
        b = min(K, a + per * 32)
        if K - b < 256:
            b = K
        bounds.append((a, b))
        a = b
    return bounds


def _gather_columns_async(t_l, bounds, group, force_collective=False):
    """all-gather the column ranges of a [Bl, K] shard as separate collectives: [(work or None, [B, Kc] tensor)] in range
    order.  RCCL: async_op -- the collectives queue up on the communicator's stream and `work.wait()` makes the compute
    stream wait for ONE of them, so what is computed on range c overlaps the transfers of ranges c + 1, ...; gloo (CPU
    rehearsal with device tensors): staged through the host, no overlap.
    Coverage: the RCCL branch (async_op + work.wait()) has run at world size 1 only (tools/nccl_selftest.py forces the
    collective there); with two or more ranks it has not executed anywhere yet -- a one-GPU box cannot hold two RCCL ranks,
    and the two-rank gloo test takes the synchronous branch.  bench.py times it as `gather_chunks` on the first multi-GPU run."""
    world = dist.get_world_size(group)
    out = []
    for a, b in bounds:
        piece = t_l[:, a:b].contiguous()
        if world == 1 and not force_collective:
            out.append((None, piece))
        elif dist.get_backend(group) == "gloo":
            out.append((None, all_gather_cat(piece, group)))
        else:
            full = torch.empty((world * piece.shape[0], b - a), dtype=piece.dtype, device=piece.device)
            out.append((dist.all_gather_into_tensor(full, piece, group=group, async_op=True), full))
    return out


class _AllGatherLocalGrad(torch.autograd.Function):
    """all-gather whose backward hands back the LOCAL slice of the incoming gradient.  For a
    quantity every rank computes identically from the gathered tensor (replicated loss) that slice
    is the partial derivative through this rank's samples; the parameter-gradient all-reduce SUM
    completes it."""

    @staticmethod
    def forward(ctx, t, group):
        ctx.cfg = (dist.get_rank(group), t.shape[0])
        return all_gather_cat(t, group)

    @staticmethod
    def backward(ctx, g):
        rank, Bl = ctx.cfg
        return g[rank * Bl:(rank + 1) * Bl].contiguous(), None


def all_gather_local_grad(t, group=None):
    return _AllGatherLocalGrad.apply(t, group)


class _ShardedLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, real_l, fake_l, h_fake_l, h_real_l, m_real_l, m_fake_l, sc, eps, L, group, ops):
        rank, world = dist.get_rank(group), dist.get_world_size(group)
        Bl = real_l.shape[0]
        _mark("start")
        # B > 64 on the HIP ops: the row block runs on the matrix pipe and needs x.x, e.e, x.e of every sample -- each rank
        # computes its own rows' from its local shard and the 24 bytes per sample are gathered ahead of the videos
        norms = None
        if (hasattr(ops, "row_norms") and not ops.replicate_costs(Bl * world, real_l.shape[1])
                and ops.rows_gram_supported(Bl, Bl * world, real_l.shape[1]) and os.environ.get("KCCOT_DIST_ROWS") != "direct"):
            norms = all_gather_cat(ops.row_norms(real_l, fake_l), group)
        # KCCOT_DIST_GATHER_CHUNKS = N > 1 (matrix-pipe row blocks only): the videos travel as N column ranges and the Gram
        # sums of a range are formed while the later ranges are still in flight (opt-in: it has not
        # been timed on more than one GPU).  The ranges are kept as they arrive -- the backward works range by range too.
        nchunks = int(os.environ.get("KCCOT_DIST_GATHER_CHUNKS", "0") or 0)
        chunked = norms is not None and nchunks > 1 and hasattr(ops, "rows_gram_sums")
        bounds = gather_chunk_bounds(real_l.shape[1], nchunks) if chunked else None
        chunked = chunked and len(bounds) > 1
        if chunked:
            pieces_r = _gather_columns_async(real_l, bounds, group)
            pieces_f = _gather_columns_async(fake_l, bounds, group)
            real = fake = None
        else:
            real = all_gather_cat(real_l, group)
            fake = all_gather_cat(fake_l, group)
        # the four [Bl,T,J] feature shards travel as one message
        feats = all_gather_cat(torch.stack([h_fake_l, h_real_l, m_real_l, m_fake_l], dim=1), group)
        h_fake, h_real, m_real, m_fake = (feats[:, i].contiguous() for i in range(4))
        _mark("exchange_inputs")       # (chunked gather: only the small messages -- the video ranges are waited for below)
        if chunked:
            B = Bl * world
            gsum = _lib.empty((int(lib.kccot_pairwise_cost3_rows_gram_sums_count(Bl, B)),), torch.float64, real_l.device)
            real, fake = [], []
            for c, ((wr, r_c), (wf, f_c)) in enumerate(zip(pieces_r, pieces_f)):      # fixed range order: reproducible sums
                for w in (wr, wf):
                    if w is not None:
                        w.wait()
                ops.rows_gram_sums(r_c, f_c, rank * Bl, Bl, gsum, c > 0)
                real.append(r_c)
                fake.append(f_c)
            blk = ops.rows_gram_from_sums(gsum, B, h_fake, h_real, m_real, m_fake, sc, rank * Bl, Bl, norms)
            _mark("cost_rows_overlapping_the_gather")
            C3 = all_gather_cat(blk.transpose(0, 1).contiguous(), group).transpose(0, 1).contiguous()  # [3,B,B]
            _mark("exchange_costs")
        elif hasattr(ops, "cost3_full") and ops.replicate_costs(real.shape[0], real.shape[1]):
            C3 = ops.cost3_full(real, fake, h_fake, h_real, m_real, m_fake, sc)     # small batch: replicated assembly
            _mark("cost_replicated")
        else:
            # row blocks of the three cost matrices (gan_utils.py:221-223)
            if norms is not None:                # the Gram row block on the matrix pipe
                blk = ops.cost3_rows(real, fake, h_fake, h_real, m_real, m_fake, sc, rank * Bl, Bl, norms)
            elif hasattr(ops, "cost3_rows"):     # one launch for the three row blocks
                blk = ops.cost3_rows(real, fake, h_fake, h_real, m_real, m_fake, sc, rank * Bl, Bl)
            else:
                blk = torch.stack([ops.cost_rows(real_l, fake, h_fake_l, m_real, sc),
                                   ops.cost_rows(real_l, real, h_real_l, m_real, sc),
                                   ops.cost_rows(fake_l, fake, h_fake_l, m_fake, sc)], dim=0)        # [3,Bl,B]
            _mark("cost_rows")
            C3 = all_gather_cat(blk.transpose(0, 1).contiguous(), group).transpose(0, 1).contiguous()  # [3,B,B]
            _mark("exchange_costs")
        if hasattr(ops, "divergence_fwd"):       # solves + combination in one launch
            loss, saved = ops.divergence_fwd(C3, eps, L)
        else:
            cost3, saved = ops.sinkhorn3_fwd(C3, eps, L)
            loss = (2.0 * cost3[0] - cost3[1]) - cost3[2]       # gan_utils.py:225
        _mark("sinkhorn_fwd")
        if ops is HipOps:
            last_info["nits"], last_info["nits_executed"] = saved[3][:3], saved[3][3:]
            gan_utils.last_info["compute_sinkhorn_loss"] = saved[3][:3]  # raise_if_solver_aborted() covers the sharded loss too
        ctx.saved_state = (saved, real, fake, h_fake, h_real, m_real, m_fake)
        ctx.cfg = (sc, rank * Bl, Bl, ops)
        return loss

    @staticmethod
    def backward(ctx, g):
        saved, real, fake, h_fake, h_real, m_real, m_fake = ctx.saved_state
        sc, row_begin, Bl, ops = ctx.cfg
        if ctx.needs_input_grad[0]:
            raise NotImplementedError("the loss path never differentiates w.r.t. real (kernel_train.py:252,289)")
        g = g.reshape(())
        _mark("between_fwd_and_bwd")
        if hasattr(ops, "divergence_bwd"):
            dC3 = ops.divergence_bwd(saved, g)
        else:
            gcost3 = torch.stack([2.0 * g, -g, -g])             # d(2 xy - xx - yy)
            dC3 = ops.sinkhorn3_bwd(saved, gcost3)
        _mark("sinkhorn_bwd")
        if isinstance(real, list):       # chunked gather: the video gradient is separable in the columns, range by range
            parts = [ops.cost3_bwd_rows(dC3, r_c, f_c, h_fake, h_real, m_real, m_fake, sc, row_begin, Bl)
                     for r_c, f_c in zip(real, fake)]
            dfake = torch.cat([p[0] for p in parts], dim=1)
            dhf, dhr, dmr, dmf = parts[0][1:]                    # the feature gradients do not depend on the videos
        else:
            dfake, dhf, dhr, dmr, dmf = ops.cost3_bwd_rows(dC3, real, fake, h_fake, h_real, m_real, m_fake, sc, row_begin, Bl)
        _mark("gradient")
        return None, dfake, dhf, dhr, dmr, dmf, None, None, None, None, None


def sharded_sinkhorn_loss(f_real_l, f_fake_l, scaling_coef, h_fake_l, m_real_l, h_real_l, m_fake_l, group=None,
                          ops=None, epsilon=1.0, L=100, protocol=None):
    """compute_sinkhorn_loss (gan_utils.py:204-227) of the GLOBAL batch from per-rank shards.
    Arguments are this rank's [B/G, ...] slices, in the reference's order h_fake, m_real, h_real,
    m_fake.  epsilon / L default to what the reference effectively runs (1.0, 100)."""
    ops = ops or HipOps
    Bl = f_real_l.shape[0]
    cast = (lambda v: v.float()) if ops is HipOps else (lambda v: v)   # the HIP kernels are fp32
    flat = lambda v: cast(v.reshape(Bl, -1)).contiguous()
    feat = lambda v: cast(v).contiguous()
    # protocol: "gather" (the default, and what BASELINE.json's north star prescribes: all-gather the batch; replicated
    # assembly at B <= 64, row blocks above), "ksplit" (opt-in: shard the contraction -- all-to-all into K-slices,
    # all-reduced fp64 Gram sums; by byte counts and per-rank kernel times it should win above B = 64, DESIGN.md
    # section 6, but it has never been timed on more than one GPU, so it stays opt-in until a SCALE record exists),
    # or "auto" (ksplit above B = 64 when the shape allows it, gather otherwise).
    protocol = protocol or os.environ.get("KCCOT_DIST_PROTOCOL", "gather")
    if protocol not in ("auto", "gather", "ksplit"):
        raise ValueError("unknown protocol %r" % (protocol,))
    if protocol == "auto":
        world = dist.get_world_size(group)
        K = f_real_l.reshape(Bl, -1).shape[1]
        protocol = "ksplit" if (ops is HipOps and Bl * world > 64 and ksplit_supported(Bl * world, K, world)) else "gather"
    if protocol == "ksplit":
        if ops is not HipOps:
            raise ValueError("the ksplit protocol runs on the HIP ops only")
        world = dist.get_world_size(group)
        K = f_real_l.reshape(Bl, -1).shape[1]
        if not ksplit_supported(Bl * world, K, world):
            raise NotImplementedError("ksplit protocol: unsupported shape B=%d K=%d on %d ranks" % (Bl * world, K, world))
        return _KSplitLoss.apply(flat(f_real_l), flat(f_fake_l), feat(h_fake_l), feat(h_real_l), feat(m_real_l),
                                 feat(m_fake_l), float(scaling_coef), float(epsilon), int(L), group).reshape(())
    return _ShardedLoss.apply(flat(f_real_l), flat(f_fake_l), feat(h_fake_l), feat(h_real_l), feat(m_real_l),
                              feat(m_fake_l), float(scaling_coef), float(epsilon), int(L), group, ops).reshape(())


# ---- helpers used by bench.py ---------------------------------------------------------------------
def shard_batch(t, rank, world):
    """Slice the leading (batch) axis of every tensor of a dict into this rank's shard."""
    out = {}
    for k, v in t.items():
        B = v.shape[0]
        if B % world:
            raise ValueError("batch %d is not divisible by %d ranks" % (B, world))
        s = v.detach()[rank * (B // world):(rank + 1) * (B // world)].contiguous()
        out[k] = s.requires_grad_(k != "real")
    return out


def sharded_loss_step(shard, sc, group=None, epsilon=1.0, L=100, protocol=None):
    loss = sharded_sinkhorn_loss(shard["real"], shard["fake"], sc, shard["h_fake"], shard["m_real"], shard["h_real"],
                                 shard["m_fake"], group, epsilon=epsilon, L=L, protocol=protocol)
    grads = torch.autograd.grad(loss, [shard[k] for k in ("fake", "h_fake", "h_real", "m_real", "m_fake")])
    return loss, grads
