"""Worker + test-only ops for the batch-sharded path (spawned by tests/test_dist_*.py)."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests", "golden")):
    if p not in sys.path:
        sys.path.insert(0, p)

import cases  # noqa: E402
from oracle import gan_utils_torch as ot  # noqa: E402


class OracleOps:
    """CPU stand-in for the four device operations, built on the torch oracle (fp64 autograd).
    TEST ONLY: lets the gloo tests exercise gather / row-block / index logic without a GPU."""

    @staticmethod
    def _cost(x, y, h, M, sc):
        return ot.cost_xy(x.unsqueeze(1), y.unsqueeze(1), sc) + ot.causal_term(h, M, sc)

    @staticmethod
    def cost_rows(x_rows, y_full, h_rows, M_full, sc):
        return OracleOps._cost(x_rows, y_full, h_rows, M_full, sc)

    @staticmethod
    def sinkhorn3_fwd(C3, eps, L):
        with torch.enable_grad():
            leaf = C3.detach().clone().requires_grad_(True)
            costs = torch.stack([ot.sinkhorn_from_cost(leaf[p], eps, L)[0] for p in range(3)])
        return costs.detach(), (leaf, costs)

    @staticmethod
    def sinkhorn3_bwd(saved, gcost3):
        leaf, costs = saved
        return torch.autograd.grad(costs, leaf, gcost3.to(costs.dtype))[0]

    @staticmethod
    def cost3_bwd_rows(dC3, real, fake, h_fake, h_real, m_real, m_fake, sc, row_begin, row_count):
        with torch.enable_grad():
            v = [t.detach().clone().requires_grad_(True) for t in (fake, h_fake, h_real, m_real, m_fake)]
            f, hf, hr, mr, mf = v
            C3 = torch.stack([OracleOps._cost(real, f, hf, mr, sc), OracleOps._cost(real, real, hr, mr, sc),
                              OracleOps._cost(f, f, hf, mf, sc)])
            grads = torch.autograd.grad(C3, v, dC3)
        return tuple(g[row_begin:row_begin + row_count].contiguous() for g in grads)


def run(rank, world, port, shape, seed, regime, device, use_hip, out_path):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from kccotgan_amd import dist as kd
    inp = cases.gen_inputs(shape, seed, regime)
    dtype = torch.float32 if use_hip else torch.float64
    t = {k: torch.from_numpy(v).to(dtype).to(device) for k, v in inp.items()}
    shard = kd.shard_batch(t, rank, world)
    loss = kd.sharded_sinkhorn_loss(shard["real"], shard["fake"], cases.SC, shard["h_fake"], shard["m_real"],
                                    shard["h_real"], shard["m_fake"], ops=None if use_hip else OracleOps)
    names = ("fake", "h_fake", "h_real", "m_real", "m_fake")
    grads = torch.autograd.grad(loss, [shard[k] for k in names])
    res = {"loss": np.array(float(loss))}
    for k, g in zip(names, grads):
        res["d" + k] = g.detach().cpu().double().numpy()
    if use_hip:      # the graph-captured form of the same step (kccotgan_amd.graph): bit-identical
        from kccotgan_amd.graph import GraphedShardedStep, GraphedKSplitStep
        from kccotgan_amd import dist as _kd
        B_, K_ = inp["real"].shape[0], int(np.prod(inp["real"].shape[1:]))
        proto = os.environ.get("KCCOT_DIST_PROTOCOL", "gather")
        if proto == "auto":
            proto = "ksplit" if (B_ > 64 and _kd.ksplit_supported(B_, K_, world)) else "gather"
        cls = GraphedKSplitStep if proto == "ksplit" else GraphedShardedStep
        step = cls(shard, cases.SC, L=100)
        for _ in range(2):
            gl, gg = step()
        res["graphed_loss_equal"] = np.array(bool(torch.equal(gl.reshape(()), loss.detach().reshape(()))))
        res["graphed_grads_equal"] = np.array(all(bool(torch.equal(gg[k], g)) for k, g in zip(names, grads)))
        # new local inputs through the call: the fake shard scaled by 0.5 must change the loss
        gl2, _ = step(fake=shard["fake"].detach() * 0.5)
        res["graphed_sees_new_inputs"] = np.array(not bool(torch.equal(gl2, loss.detach().reshape(()))))
    np.savez(out_path % rank, **res)
    dist.barrier()
    dist.destroy_process_group()


def smooth_case(seed):
    """Videos whose global maximum sits in ONE shard, and an upstream gradient, both functions of the global index."""
    rng = np.random.default_rng(100 + seed)
    x = rng.random((4, 16, 12, 16, 1), dtype=np.float32)
    x[3] *= 1.5                               # the arg-max lives in the last sample (rank world-1)
    g = rng.standard_normal(x.shape).astype(np.float32)
    return x, g


def run_smooth(rank, world, port, seed, device, out_path):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from kccotgan_amd.data_utils import KernelSmoothing
    x, g = smooth_case(seed)
    Bl = x.shape[0] // world
    sl = slice(rank * Bl, (rank + 1) * Bl)
    ks = KernelSmoothing(6, 6, sharded=True)
    res = {}
    for name, fn in (("t", ks.temporal_convolution), ("3d", ks.gaussian_convolution3D)):
        xt = torch.from_numpy(x[sl]).to(device).requires_grad_(True)
        out = fn(xt, 1.7)
        out.backward(torch.from_numpy(g[sl]).to(device))
        res["out_" + name] = out.detach().cpu().numpy()
        res["din_" + name] = xt.grad.cpu().numpy()
    np.savez(out_path % rank, **res)
    dist.barrier()
    dist.destroy_process_group()


def run_train(rank, world, port, seed, device, out_path):
    """One data-parallel training iteration (disc + gen step) of the small trainer configuration on `world` ranks."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from kccotgan_amd import gan
    from kccotgan_amd.kernel_train import KCCOTTrainer
    gan._NATIVE = {"convlstm", "deconv", "dconv"}            # conservative convolution mode, as the single-GPU trainer test
    Bl, H, W, C, T, iT = 2, 64, 64, 1, 6, 2
    tr = KCCOTTrainer(Bl, total_time_steps=T, int_time_steps=iT, x_height=H, x_width=W, channels=C, kernel="1d", warmup=10,
                      device=device, seed=seed + rank)       # different initial weights per rank: the constructor broadcasts rank 0's
    x = torch.from_numpy(np.random.default_rng(7).random((Bl * world, H, T, W, C), dtype=np.float32))[rank * Bl:(rank + 1) * Bl]
    p0 = torch.cat([p.detach().reshape(-1) for p in tr.g_params + tr.d_params]).cpu().numpy()
    pm, loss = tr.train_iteration(x.to(device), 4.0)
    p1 = torch.cat([p.detach().reshape(-1) for p in tr.g_params + tr.d_params]).cpu().numpy()
    np.savez(out_path % rank, p0=p0, p1=p1, pm=np.array(float(pm)), loss=np.array(float(loss)))
    dist.barrier()
    dist.destroy_process_group()


def run_noise(rank, world, port, device, out_path):
    """Data-parallel trainer built with the DEFAULT seed on every rank: identical weights, different generator noise."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from kccotgan_amd.kernel_train import KCCOTTrainer
    tr = KCCOTTrainer(2, total_time_steps=4, int_time_steps=2, x_height=16, x_width=16, channels=1, g_state_size=2,
                      d_state_size=2, g_filter_size=2, d_filter_size=2, z_channels=3, kernel="none", device=device)
    z = torch.randn(tr.z_shape, device=device)           # what _forward draws (kernel_train.py:220,260)
    p = torch.cat([q.detach().reshape(-1) for q in tr.g_params + tr.d_params])
    np.savez(out_path % rank, z=z.cpu().numpy(), p=p.cpu().numpy())
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    a = sys.argv
    if a[8] == "noise":
        run_noise(int(a[1]), int(a[2]), int(a[3]), a[7], a[9])
    elif a[8] == "train":
        run_train(int(a[1]), int(a[2]), int(a[3]), int(a[5]), a[7], a[9])
    elif a[8] == "smooth":
        run_smooth(int(a[1]), int(a[2]), int(a[3]), int(a[5]), a[7], a[9])
    else:
        run(int(a[1]), int(a[2]), int(a[3]), a[4], int(a[5]), a[6], a[7], a[8] == "hip", a[9])
