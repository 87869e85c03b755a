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

Both require fixed shapes; inputs are copied into the static buffers on every call unless the
caller writes into ``.static`` directly.
"""
import torch

from . import gan_utils

_NAMES = ("real", "fake", "h_fake", "m_real", "h_real", "m_fake")
_WRT = ("fake", "h_fake", "h_real", "m_real", "m_fake")


def _loss(t, sc, sinkhorn_eps, sinkhorn_l):
    return gan_utils.compute_sinkhorn_loss(t["real"], t["fake"], sc, sinkhorn_eps, sinkhorn_l, t["h_fake"], t["m_real"],
                                           t["h_real"], t["m_fake"], video=True)


class GraphedLossStep:
    def __init__(self, sample, scaling_coef, sinkhorn_eps=0.8, sinkhorn_l=100, warmup=3):
        """``sample``: dict with the six tensors of compute_sinkhorn_loss (shapes and device are what
        gets captured; values are copied)."""
        self.static = {k: sample[k].detach().clone() for k in _NAMES}
        for k in _WRT:
            self.static[k].requires_grad_(True)
        self._cfg = (float(scaling_coef), sinkhorn_eps, sinkhorn_l)
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
