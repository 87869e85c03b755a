"""HIP-graph capture of the loss path (PyTorch-ROCm ``torch.cuda.CUDAGraph`` = hipGraph).

One ``compute_sinkhorn_loss`` forward + backward is eight short kernels (22 + 5 + 12 us of cost
assembly, the two Sinkhorn kernels, 5 + 28 + 9 us of cost backward); issued eagerly from Python the
host needs 0.1-0.4 ms to launch them, which is as long as or longer than they run.  Captured once
into a graph over static buffers, a step is ONE ``hipGraphLaunch`` and the path becomes GPU-bound.
Every kernel of the eager path runs, with the same arguments, in the same order: results are
bit-identical (tests/test_gpu_parity.py::test_graphed_loss_is_bit_identical).

``GraphedLossStep``  -- forward + backward in one graph; for callers that need the loss and its
                        gradients w.r.t. (fake, h_fake, h_real, m_real, m_fake) at dLoss = 1
                        (the generator step, kernel_train.py:287-289; bench.py).
``graphed_loss``     -- ``torch.cuda.make_graphed_callables`` around ``compute_sinkhorn_loss``:
                        forward and backward captured separately, usable inside a larger autograd
                        graph (discriminators / generator around it run eagerly).

``GraphedShardedStep`` -- the batch-sharded step (kccotgan_amd.dist): the collectives stay ordinary RCCL calls,
                        the two compute segments between them (row blocks; solves + reverse sweep + cost backward
                        of this rank's rows -- the backward needs no communication) are one graph each.  Issued
                        eagerly from Python the sharded step costs 0.36 ms of host time for 0.22 ms of kernels.

All require fixed shapes; inputs are copied into the static buffers on every call unless the
caller writes into ``.static`` directly.
"""
import os

import torch

from . import gan_utils

_NAMES = ("real", "fake", "h_fake", "m_real", "h_real", "m_fake")
_WRT = ("fake", "h_fake", "h_real", "m_real", "m_fake")


def _loss(t, sc, sinkhorn_eps, sinkhorn_l, honor_eps_l=False):
    return gan_utils.compute_sinkhorn_loss(t["real"], t["fake"], sc, sinkhorn_eps, sinkhorn_l, t["h_fake"], t["m_real"],
                                           t["h_real"], t["m_fake"], video=True, honor_eps_l=honor_eps_l)


class GraphedLossStep:
    def __init__(self, sample, scaling_coef, sinkhorn_eps=0.8, sinkhorn_l=100, warmup=3, honor_eps_l=False, clone=True):
        """``sample``: dict with the six tensors of compute_sinkhorn_loss (shapes and device are what
        gets captured; values are copied -- ``clone=False`` captures the given tensors themselves: a caller that owns
        static buffers already, or a batch too large to hold twice).  ``honor_eps_l``: the keyword-only opt-in of
        ``compute_sinkhorn_loss`` (the reference ignores ``sinkhorn_eps`` / ``sinkhorn_l``, gan_utils.py:221-223)."""
        self.static = {k: (sample[k].detach().clone() if clone else sample[k].detach()) for k in _NAMES}
        for k in _WRT:
            self.static[k].requires_grad_(True)
        self._cfg = (float(scaling_coef), sinkhorn_eps, sinkhorn_l, bool(honor_eps_l))
        dev = self.static["real"].device
        self._one = torch.ones((), device=dev)       # dLoss = 1 held in a static buffer: no fill kernel per replay
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):            # warm-up off the capture: workspaces, ticket, allocator
            for _ in range(warmup):
                self._eager()
        torch.cuda.current_stream(dev).wait_stream(side)
        torch.cuda.synchronize(dev)
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self.loss, self.grads = self._eager()
        # iteration counts written by the captured forward (device tensors, refreshed by every replay)
        self.nits = gan_utils.last_info["compute_sinkhorn_loss"]
        self.nits_executed = gan_utils.last_info["compute_sinkhorn_loss_executed"]

    def _eager(self):
        t = self.static
        loss = _loss(t, *self._cfg)
        grads = torch.autograd.grad(loss, [t[k] for k in _WRT], grad_outputs=self._one)
        return loss.detach(), dict(zip(_WRT, grads))

    def __call__(self, **inputs):
        """Copy the given inputs (any subset of the six names) into the static buffers and replay.
        Returns (loss, grads): static tensors, overwritten by the next call."""
        with torch.no_grad():
            for k, v in inputs.items():
                if v is not self.static[k]:
                    self.static[k].copy_(v)
        self.graph.replay()
        return self.loss, self.grads


def graphed_loss(sample, scaling_coef, sinkhorn_eps=0.8, sinkhorn_l=100):
    """A callable ``f(real, fake, h_fake, m_real, h_real, m_fake) -> loss`` whose forward and backward
    are graph replays, differentiable w.r.t. the arguments that require grad in ``sample``."""
    sc = float(scaling_coef)

    def fn(real, fake, h_fake, m_real, h_real, m_fake):
        return gan_utils.compute_sinkhorn_loss(real, fake, sc, sinkhorn_eps, sinkhorn_l, h_fake, m_real, h_real, m_fake,
                                               video=True)

    args = tuple(sample[k].detach().clone().requires_grad_(sample[k].requires_grad) for k in _NAMES)
    return torch.cuda.make_graphed_callables(fn, args)


_FEATS = ("h_fake", "h_real", "m_real", "m_fake")


class GraphedShardedStep:
    """Forward + backward of ``dist.sharded_sinkhorn_loss`` at dLoss = 1 for one rank, as
    all-gather x3 -> graph A (row blocks of the three cost matrices) -> all-gather -> graph B (the replicated solves,
    the reverse sweep, the gradients of this rank's rows); for batches of at most 64 (``HipOps.replicate_costs``) the
    costs are assembled whole on every rank and everything after the input gathers is ONE graph.  Same kernels, arguments and order as the eager path
    (``dist._ShardedLoss``): bit-identical results (tests/test_dist_gloo.py).  ``step(fake=..., h_fake=...)`` copies the
    given LOCAL shards ([B/G, ...]) into the static buffers first; returns (loss, grads) as static tensors."""

    def __init__(self, shard, scaling_coef, group=None, epsilon=1.0, L=100, warmup=2):
        import torch.distributed as dist
        from . import dist as kd
        self._kd, self._dist, self.group = kd, dist, group
        self.rank, self.world = dist.get_rank(group), dist.get_world_size(group)
        self._nccl = dist.get_backend(group) == "nccl"
        Bl = shard["real"].shape[0]
        dev = shard["real"].device
        flat = lambda v: v.detach().reshape(Bl, -1).float().contiguous().clone()
        self.local = {"real": flat(shard["real"]), "fake": flat(shard["fake"]),
                      "feats": torch.stack([shard[k].detach().float() for k in _FEATS], dim=1).contiguous()}
        self._shapes = {k: tuple(shard[k].shape) for k in ("fake",) + _FEATS}
        B, K = Bl * self.world, self.local["real"].shape[1]
        T, J = self.local["feats"].shape[2], self.local["feats"].shape[3]
        new = lambda *shape: torch.empty(shape, dtype=torch.float32, device=dev)
        self.full = {"real": new(B, K), "fake": new(B, K), "feats": new(B, 4, T, J)}
        self._f = [new(B, T, J) for _ in range(4)]          # h_fake, h_real, m_real, m_fake of the whole batch
        self._blk_t, self._C3g = new(Bl, 3, B), new(B, 3, B)
        self._one = torch.ones((), device=dev)
        self._cfg = (float(scaling_coef), float(epsilon), int(L), self.rank * Bl, Bl)
        # small batches: replicated one-pass cost assembly, no row-block exchange -- ONE graph after the input gathers
        self.replicated = kd.HipOps.replicate_costs(B, K)
        # B > 64: the row block runs on the matrix pipe when the shape allows it (as dist._ShardedLoss does) and needs the
        # row norms of every sample: each rank's own rows are computed in front of the gathers and travel with them
        self.use_norms = (not self.replicated and kd.HipOps.rows_gram_supported(Bl, B, K)
                          and os.environ.get("KCCOT_DIST_ROWS") != "direct")
        if self.use_norms:
            self.local["norms"] = torch.zeros((Bl, 3), dtype=torch.float64, device=dev)
            self.full["norms"] = torch.zeros((B, 3), dtype=torch.float64, device=dev)
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):                        # warm-up off the capture: workspaces, ticket, allocator
            for _ in range(warmup):
                self._gather_inputs()
                self._seg_a()
                if not self.replicated:
                    self._gather(self._C3g, self._blk_t)
                self._seg_b()
        torch.cuda.current_stream(dev).wait_stream(side)
        torch.cuda.synchronize(dev)
        # thread-local error mode: the process group's watchdog thread may touch the device during the capture
        self.graph_a = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph_a, capture_error_mode="thread_local"):
            self._seg_a()
            if self.replicated:
                self.loss, self.grads, nits = self._seg_b()
        if not self.replicated:
            self.graph_b = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.graph_b, capture_error_mode="thread_local"):
                self.loss, self.grads, nits = self._seg_b()
        self.nits, self.nits_executed = nits[:3], nits[3:]

    def _gather(self, out, local):
        if self._nccl:                                       # also at world size 1: the same call sequence
            self._dist.all_gather_into_tensor(out, local, group=self.group)
        elif self.world == 1:
            out.copy_(local)
        else:                                                # gloo rehearsal: staged through the host
            out.copy_(self._kd.all_gather_cat(local, self.group))

    def _gather_inputs(self):
        if self.use_norms:
            self.local["norms"].copy_(self._kd.HipOps.row_norms(self.local["real"], self.local["fake"]))
            self._gather(self.full["norms"], self.local["norms"])
        # RCCL: the three all-gathers as ONE coalesced group (one launch, one set of fixed latencies); decided at the
        # first call, sequential calls if this torch build has no coalesced all-gather
        if self._nccl and getattr(self, "_coalesce", True):
            try:
                with self._dist._coalescing_manager(group=self.group):
                    for k in ("real", "fake", "feats"):
                        self._dist.all_gather_into_tensor(self.full[k], self.local[k], group=self.group)
                self._coalesce = True
                return
            except Exception:
                if getattr(self, "_coalesce", None) is True:
                    raise                                   # it worked before: a real failure
                self._coalesce = False
        for k in ("real", "fake", "feats"):
            self._gather(self.full[k], self.local[k])

    def _seg_a(self):
        sc, _, _, row_begin, Bl = self._cfg
        for i in range(4):
            self._f[i].copy_(self.full["feats"][:, i])
        if self.replicated:
            self._C3 = self._kd.HipOps.cost3_full(self.full["real"], self.full["fake"], *self._f, sc)
            return
        blk = self._kd.HipOps.cost3_rows(self.full["real"], self.full["fake"], *self._f, sc, row_begin, Bl,
                                         self.full["norms"] if self.use_norms else None)
        self._blk_t.copy_(blk.transpose(0, 1))

    def _seg_b(self):
        sc, eps, L, row_begin, Bl = self._cfg
        H = self._kd.HipOps
        C3 = self._C3 if self.replicated else self._C3g.transpose(0, 1).contiguous()
        loss, saved = H.divergence_fwd(C3, eps, L)
        dC3 = H.divergence_bwd(saved, self._one)
        g = H.cost3_bwd_rows(dC3, self.full["real"], self.full["fake"], *self._f, sc, row_begin, Bl)
        names = ("fake",) + _FEATS
        return loss, {k: v.reshape(self._shapes[k]) for k, v in zip(names, g)}, saved[3]

    def __call__(self, **inputs):
        with torch.no_grad():
            for k, v in inputs.items():
                if k in ("real", "fake"):
                    self.local[k].copy_(v.reshape(self.local[k].shape))
                else:
                    self.local["feats"][:, _FEATS.index(k)].copy_(v)
            self._gather_inputs()
            self.graph_a.replay()
            if not self.replicated:
                self._gather(self._C3g, self._blk_t)
                self.graph_b.replay()
        return self.loss, self.grads


class GraphedKSplitStep:
    """Forward + backward of the contraction-sharded protocol (``dist._KSplitLoss``) at dLoss = 1 for one rank:
    all-to-all x2 + all-gather (features) -> graph A (fp64 Gram sums of the rank's K-slice) -> all-reduce(SUM) of the
    sums -> graph B (finalize, solves, reverse sweep, video gradient of ALL samples on the slice, feature gradients
    of the rank's samples) -> all-to-all back.  Same calls and order as the eager Function: bit-identical results."""

    def __init__(self, shard, scaling_coef, group=None, epsilon=1.0, L=100, warmup=2):
        import ctypes
        import torch.distributed as dist
        from . import dist as kd
        from . import _lib
        from ._lib import lib, check
        self._kd, self._dist, self.group = kd, dist, group
        self.rank, self.world = dist.get_rank(group), dist.get_world_size(group)
        Bl = shard["real"].shape[0]
        dev = shard["real"].device
        flat = lambda v: v.detach().reshape(Bl, -1).float().contiguous().clone()
        self.local = {"real": flat(shard["real"]), "fake": flat(shard["fake"]),
                      "feats": torch.stack([shard[k].detach().float() for k in _FEATS], dim=1).contiguous()}
        self._shapes = {k: tuple(shard[k].shape) for k in ("fake",) + _FEATS}
        B, K = Bl * self.world, self.local["real"].shape[1]
        if not kd.ksplit_supported(B, K, self.world):
            raise NotImplementedError("ksplit protocol: unsupported shape B=%d K=%d on %d ranks" % (B, K, self.world))
        Ks = K // self.world
        T, J = self.local["feats"].shape[2], self.local["feats"].shape[3]
        new = lambda *shape: torch.empty(shape, dtype=torch.float32, device=dev)
        self._f = [new(B, T, J) for _ in range(4)]
        self._C3 = new(3, B, B)
        self._dfake_s = new(B, Ks)
        self._fg = [new(Bl, T, J) for _ in range(4)]
        self._one = torch.ones((), device=dev)
        self._wsb = int(lib.kccot_pairwise_cost3_workspace_bytes(B, Ks))
        self._ws = torch.empty(self._wsb, dtype=torch.uint8, device=dev)
        off, cnt = ctypes.c_size_t(0), ctypes.c_size_t(0)
        check(lib.kccot_pairwise_cost3_gram_sums_span(B, Ks, ctypes.byref(off), ctypes.byref(cnt)), "gram_sums_span")
        self._gsum = self._ws[off.value:off.value + 8 * cnt.value].view(torch.float64)
        self._cfg = (float(scaling_coef), float(epsilon), int(L), self.rank * Bl, Bl, B, Ks, T, J)
        self._exchange_in()
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):
            for _ in range(warmup):
                self._exchange_in()
                self._seg_a()
                kd._all_reduce_sum(self._gsum, group)
                self._seg_b()
                kd.all_to_all_rows(self._dfake_s, group)
        torch.cuda.current_stream(dev).wait_stream(side)
        torch.cuda.synchronize(dev)
        self.graph_a = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph_a, capture_error_mode="thread_local"):
            self._seg_a()
        self.graph_b = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph_b, capture_error_mode="thread_local"):
            self.loss, nits = self._seg_b()
        self.nits, self.nits_executed = nits[:3], nits[3:]
        self.grads = None

    def _exchange_in(self):
        kd = self._kd
        rs, fs = kd.all_to_all_slices(self.local["real"], self.group), kd.all_to_all_slices(self.local["fake"], self.group)
        ft = kd.all_gather_cat(self.local["feats"], self.group)
        if not hasattr(self, "_real_s"):
            self._real_s, self._fake_s, self._feats = rs.clone(), fs.clone(), ft.clone()       # static homes
        else:
            self._real_s.copy_(rs); self._fake_s.copy_(fs); self._feats.copy_(ft)

    def _cost3(self, flags):
        from . import _lib
        from ._lib import lib, check, ptr, stream_of
        sc, _, _, _, _, B, Ks, T, J = self._cfg
        check(lib.kccot_pairwise_cost3_f32(ptr(self._real_s), ptr(self._fake_s), B, Ks, sc, ptr(self._f[0]), ptr(self._f[1]),
                                           ptr(self._f[2]), ptr(self._f[3]), T, J, flags, ptr(self._C3), self._ws.data_ptr(),
                                           self._wsb, stream_of(self._real_s)), "pairwise_cost3")

    def _seg_a(self):
        from . import _lib
        for i in range(4):
            self._f[i].copy_(self._feats[:, i])
        self._cost3(_lib.COST_GRAM_SUMS_ONLY)

    def _seg_b(self):
        from . import _lib
        from ._lib import lib, check, ptr, stream_of, workspace
        sc, eps, L, row_begin, Bl, B, Ks, T, J = self._cfg
        H = self._kd.HipOps
        self._cost3(_lib.COST_FROM_GRAM_SUMS)
        loss, saved = H.divergence_fwd(self._C3, eps, L)
        dC3 = H.divergence_bwd(saved, self._one)
        ws, wsb = workspace(lib.kccot_pairwise_cost3_bwd_workspace_bytes(B, Ks), self._real_s)
        check(lib.kccot_pairwise_cost3_bwd_f32(ptr(dC3), ptr(self._real_s), ptr(self._fake_s), B, Ks, sc, None, None, None, None,
                                               1, 1, ptr(self._dfake_s), None, None, None, None, ws, wsb,
                                               stream_of(self._real_s)), "pairwise_cost3_bwd")
        check(lib.kccot_pairwise_cost3_bwd_rows_f32(ptr(dC3), ptr(self._real_s), ptr(self._fake_s), B, Ks, sc, ptr(self._f[0]),
                                                    ptr(self._f[1]), ptr(self._f[2]), ptr(self._f[3]), T, J, row_begin, Bl, None,
                                                    ptr(self._fg[0]), ptr(self._fg[1]), ptr(self._fg[2]), ptr(self._fg[3]),
                                                    None, 0, stream_of(self._real_s)), "pairwise_cost3_bwd_rows")
        return loss, saved[3]

    def __call__(self, **inputs):
        with torch.no_grad():
            for k, v in inputs.items():
                if k in ("real", "fake"):
                    self.local[k].copy_(v.reshape(self.local[k].shape))
                else:
                    self.local["feats"][:, _FEATS.index(k)].copy_(v)
            self._exchange_in()
            self.graph_a.replay()
            self._kd._all_reduce_sum(self._gsum, self.group)
            self.graph_b.replay()
            dfake = self._kd.all_to_all_rows(self._dfake_s, self.group)
        self.grads = {"fake": dfake.reshape(self._shapes["fake"])}
        for k, v in zip(_FEATS, self._fg):
            self.grads[k] = v.reshape(self._shapes[k])
        return self.loss, self.grads
