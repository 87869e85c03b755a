"""The reference's training-step API (kernel_train.py:219-292) on PyTorch-ROCm + the HIP loss path.

``KCCOTTrainer.disc_training_step(real_in, real_pred, sigma) -> pM`` and
``gen_training_step(real_in, real_pred, sigma) -> loss`` keep the closures' names, arguments and
return values; the body is the reference's: sample z, run encoder/decoder, concatenate context and
prediction along time (kernel_train.py:222-227), optional kernel smoothing of real and fake
(:229-239), the two discriminators on both (:241-245), ``compute_sinkhorn_loss`` with the
reference's argument order (:247-248), ``disc_loss = -loss + pM`` (:250), Adam(beta1=0.5,
beta2=0.9) on (D_h, D_m) resp. (encoder, decoder) (:62-63,252-255,289-291) with the WarmUp +
staircase ExponentialDecay schedule (:54-59, data_utils.py:589-621).

Data-parallel use (one process per GPU, torch.distributed over RCCL): pass ``group``; each rank
feeds its shard of the batch, the loss is the GLOBAL-batch divergence assembled by
``kccotgan_amd.dist`` and parameter gradients are all-reduced with SUM (the loss is replicated,
each rank holds the partial derivative through its own samples).

``sample`` is the test-time autoregressive loop (kernel_train.py:340-347), ``fit`` the loop body around the two
steps (:295-330): per-iteration sigma (annealed or fixed, :308-311), the scalar log ``pM`` / ``Sinkhorn Loss``
(:318-321), the non-finite-loss guard (:323-329) and the periodic sample (:331,338-355).  Batches come from
``kccotgan_amd.datasets``.

Out of scope: TFRecord readers, CLI, TensorBoard writers and weight files of ``train(args)``.
"""
import math

import torch
import torch.distributed as dist

from . import gan, gan_utils
from .data_utils import KernelSmoothing


def warmup_exponential_decay(step, base_lr, warmup_steps=10000, decay_steps=5000, decay_rate=0.975):
    """data_utils.py:589-621 WarmUp around kernel_train.py:57 ExponentialDecay(staircase=True).  Pinned against the
    reference's own WarmUp class (tests/golden/lr_schedule.npz, tests/test_smoothing_golden.py)."""
    if step < warmup_steps:
        return base_lr * (step / warmup_steps)
    return base_lr * decay_rate ** ((step - warmup_steps) // decay_steps)


class KerasSharedAdam:
    """``tf.keras.optimizers.Adam(schedule, beta_1=0.5, beta_2=0.9)`` as the reference uses it: ONE optimiser
    object per step kind whose ``apply_gradients`` is called TWICE per training step, once per network
    (kernel_train.py:62-63,254-255,290-291).  Keras semantics reproduced:

    * ``iterations`` counts ``apply_gradients`` calls, so it advances by 2 per training step;
    * the schedule is evaluated at the PRE-increment count (the very first update runs at lr(0) = 0 under WarmUp),
      the first network of a step sees lr(2k), the second lr(2k+1);
    * the bias correction uses t = iterations + 1 for every variable of that call, although each variable's moments
      have only been updated k+1 times;
    * update rule ``var -= lr_t * sqrt(1 - b2^t) / (1 - b1^t) * m / (sqrt(v) + epsilon)``, epsilon = 1e-7 outside
      the bias correction (Keras' "epsilon hat").
    """

    def __init__(self, schedule, beta_1=0.5, beta_2=0.9, epsilon=1e-7):
        self.schedule, self.beta_1, self.beta_2, self.epsilon = schedule, beta_1, beta_2, epsilon
        self.iterations = 0
        self.state = {}
        self.last_lr = None

    @torch.no_grad()
    def apply_gradients(self, grads_and_vars):
        pairs = [(g, p) for g, p in grads_and_vars if g is not None]
        t = self.iterations + 1
        lr_t = float(self.schedule(self.iterations))
        self.last_lr = lr_t
        alpha = lr_t * math.sqrt(1.0 - self.beta_2 ** t) / (1.0 - self.beta_1 ** t)
        if pairs:
            ps = [p for _, p in pairs]
            gs = [g for g, _ in pairs]
            for p in ps:
                if p not in self.state:
                    self.state[p] = (torch.zeros_like(p), torch.zeros_like(p))
            ms = [self.state[p][0] for p in ps]
            vs = [self.state[p][1] for p in ps]
            torch._foreach_mul_(ms, self.beta_1)
            torch._foreach_add_(ms, gs, alpha=1.0 - self.beta_1)
            torch._foreach_mul_(vs, self.beta_2)
            torch._foreach_addcmul_(vs, gs, gs, value=1.0 - self.beta_2)
            denom = torch._foreach_sqrt(vs)
            torch._foreach_add_(denom, self.epsilon)
            torch._foreach_addcdiv_(ps, ms, denom, value=-alpha)
        self.iterations += 1


class KCCOTTrainer:
    def __init__(self, batch_size, total_time_steps=15, int_time_steps=5, x_height=64, x_width=64, channels=3,
                 g_state_size=8, d_state_size=8, g_filter_size=8, d_filter_size=8, z_channels=128, bn=True,
                 lr=5e-4, warmup=10000, sinkhorn_eps=0.8, sinkhorn_l=100, scaling_coef=15.0, reg_penalty=1.0,
                 kernel="none", device="cuda", seed=1, group=None):
        # defaults = kernel_train.py:363-409
        torch.manual_seed(seed)
        self.batch_size, self.device, self.group = batch_size, torch.device(device), group
        self.int_time_steps = int_time_steps
        self.pred_time_steps = total_time_steps - int_time_steps
        # z joins the encoder's level-4 feature map: 4 x 4 at the reference's 64 x 64 frames (kernel_train.py:135-136,220)
        self.z_shape = (batch_size, self.pred_time_steps, x_height // 16, x_width // 16, z_channels)
        self.scaling_coef = 1.0 / scaling_coef                                  # kernel_train.py:71
        self.sinkhorn_eps, self.sinkhorn_l, self.reg_penalty = sinkhorn_eps, sinkhorn_l, reg_penalty
        self.kernel_choice = kernel
        self.convolution_mode = gan.convolution_mode()      # 'miopen' or 'native:...' (the slow fallback; see gan.py)
        data_parallel = group is not None or (dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1)
        # :216; data-parallel: the division is by the maximum of the GLOBAL batch, as in the single-process run
        self.gaussian_kernel = KernelSmoothing(temporal_kernel_size=6, spatial_kernel_size=6, group=group, sharded=data_parallel)
        mk = dict(z_width=4, z_height=4, bn=bn, nchannel=channels)
        self.context_encoder = gan.VideoEncoderConvLSTM(batch_size, int_time_steps, self.pred_time_steps, g_state_size,
                                                        x_width, x_height, filter_size=g_filter_size, **mk).to(self.device)
        self.decoder = gan.VideoDecoderConvLSTM(batch_size, int_time_steps, self.pred_time_steps, g_state_size, x_width,
                                                x_height, filter_size=g_filter_size, z_channels=z_channels,
                                                **mk).to(self.device)
        self.discriminator_h = gan.VideoDiscriminator(batch_size, total_time_steps, d_state_size, x_width, x_height,
                                                      filter_size=d_filter_size, **mk).to(self.device)
        self.discriminator_m = gan.VideoDiscriminator(batch_size, total_time_steps, d_state_size, x_width, x_height,
                                                      filter_size=d_filter_size, **mk).to(self.device)
        # (first network, second network) of each shared optimiser, in the reference's apply order (:254-255,290-291)
        self.g_nets = (list(self.context_encoder.parameters()), list(self.decoder.parameters()))
        self.d_nets = (list(self.discriminator_h.parameters()), list(self.discriminator_m.parameters()))
        self.g_params = self.g_nets[0] + self.g_nets[1]
        self.d_params = self.d_nets[0] + self.d_nets[1]
        self.base_lr, self.warmup = lr, warmup
        schedule = lambda it: warmup_exponential_decay(it, lr, warmup)           # :54-59
        self.gen_optimiser = KerasSharedAdam(schedule, beta_1=0.5, beta_2=0.9)    # :62
        self.dischm_optimiser = KerasSharedAdam(schedule, beta_1=0.5, beta_2=0.9)  # :63
        if data_parallel:
            for p in self.g_params + self.d_params:                              # identical replicas
                dist.broadcast(p.data, src=0, group=group)
            # ... but independent generator noise: every rank has consumed the RNG identically up to here, so
            # without this each shard of the global batch would be generated from the same z
            torch.manual_seed(seed + 1000 * (1 + dist.get_rank(group)))

    # ------------------------------------------------------------------ the shared forward
    def _forward(self, real_in, real_pred, sigma, generator_grad=True):
        hidden_z = torch.randn(self.z_shape, device=self.device)                 # kernel_train.py:220,260
        real_inp = torch.cat((real_in, real_pred), dim=2)                        # :222
        # the discriminator step differentiates w.r.t. D_h / D_m only (:252): no generator tape there
        with torch.set_grad_enabled(generator_grad):
            preds_features = self.context_encoder(real_inp)                      # :223
            fake_pred = self.decoder(preds_features, hidden_z)                   # :224
        real = torch.cat((real_in, real_pred), dim=2)                            # :226
        fake = torch.cat((real_in, fake_pred), dim=2)                            # :227
        if self.kernel_choice == "1d":                                           # :229-239
            real = self.gaussian_kernel.temporal_convolution(real, sigma)
            fake = self.gaussian_kernel.temporal_convolution(fake, sigma)
        elif self.kernel_choice == "2d":
            real = self.gaussian_kernel.spatial_convolution(real, sigma)
            fake = self.gaussian_kernel.spatial_convolution(fake, sigma)
        elif self.kernel_choice == "3d":
            real = self.gaussian_kernel.gaussian_convolution3D(real, sigma)
            fake = self.gaussian_kernel.gaussian_convolution3D(fake, sigma)
        h_fake = self.discriminator_h(fake)                                      # :241-245
        h_real = self.discriminator_h(real)
        m_real = self.discriminator_m(real)
        m_fake = self.discriminator_m(fake)
        if self._world() > 1:
            from . import dist as kd
            loss = kd.sharded_sinkhorn_loss(real.detach(), fake, self.scaling_coef, h_fake, m_real, h_real, m_fake,
                                            group=self.group)
        else:
            loss = gan_utils.compute_sinkhorn_loss(real.detach(), fake, self.scaling_coef, self.sinkhorn_eps,
                                                   self.sinkhorn_l, h_fake, m_real, h_real, m_fake, video=True)  # :247
        return loss, m_real

    def _world(self):
        return dist.get_world_size(self.group) if (dist.is_available() and dist.is_initialized()) else 1

    def _apply(self, optimiser, nets, grads):
        """``optimiser.apply_gradients`` once per network, first then second (kernel_train.py:254-255 / :290-291);
        data-parallel: the loss is replicated and every rank holds the partial derivative through its own samples,
        so the gradients are all-reduced with SUM first (one flat buffer)."""
        grads = list(grads)
        params = nets[0] + nets[1]
        if self._world() > 1:
            flat = torch.cat([g.reshape(-1) if g is not None else p.new_zeros(p.numel()) for g, p in zip(grads, params)])
            dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group)
            off = 0
            for i, p in enumerate(params):
                n = p.numel()
                if grads[i] is not None:
                    grads[i] = flat[off:off + n].view_as(p)
                off += n
        n0 = len(nets[0])
        optimiser.apply_gradients(zip(grads[:n0], nets[0]))
        optimiser.apply_gradients(zip(grads[n0:], nets[1]))

    # ------------------------------------------------------------------ kernel_train.py:219-256
    def disc_training_step(self, real_in, real_pred, sigma):
        with gan.conv_guard():                   # forward AND backward off MIOpen when native convolutions are selected
            return self._disc_training_step(real_in, real_pred, sigma)

    def gen_training_step(self, real_in, real_pred, sigma):
        with gan.conv_guard():
            return self._gen_training_step(real_in, real_pred, sigma)

    def _disc_training_step(self, real_in, real_pred, sigma):
        loss, m_real = self._forward(real_in, real_pred, sigma, generator_grad=False)
        if self._world() > 1:               # pM couples the whole batch (std and mean over b): use the global M
            from . import dist as kd
            m_real = kd.all_gather_local_grad(m_real, self.group)
        pm1 = gan_utils.scale_invariante_martingale_regularization(m_real, self.reg_penalty, self.scaling_coef)   # :249
        disc_loss = -loss + pm1                                                  # :250
        grads = torch.autograd.grad(disc_loss, self.d_params, allow_unused=True)  # :252
        self._apply(self.dischm_optimiser, self.d_nets, grads)                   # :254-255
        return pm1.detach()                                                      # :256

    # ------------------------------------------------------------------ kernel_train.py:259-292
    def _gen_training_step(self, real_in, real_pred, sigma):
        loss, _ = self._forward(real_in, real_pred, sigma)
        grads = torch.autograd.grad(loss, self.g_params, allow_unused=True)      # :289
        self._apply(self.gen_optimiser, self.g_nets, grads)                      # :290-291
        return loss.detach()                                                     # :292

    # ------------------------------------------------------------------ kernel_train.py:338-355
    @torch.no_grad()
    def sample(self, test_data):
        """Autoregressive roll-out (kernel_train.py:340-347): keep the first ``int_time_steps`` frames of
        ``test_data`` [B,H,T,W,C], then ``pred_time_steps`` times encode everything generated so far, draw ONE
        latent frame z ~ N(0,1) of shape [B,1,z_h,z_w,z_channels] and decode with ``training=False`` (the decoder
        reads only the last encoded frame, gan.py:269-272) and append the frame.  Returns [B,H,T_total,W,C]."""
        test_inputs = test_data[:, :, :self.int_time_steps].to(self.device, torch.float32)
        z_shape = (self.batch_size, 1) + tuple(self.z_shape[2:])
        with gan.conv_guard():
            for _ in range(self.pred_time_steps):
                preds_features = self.context_encoder(test_inputs, training=False)
                hidden_z = torch.randn(z_shape, device=self.device)
                preds = self.decoder(preds_features, hidden_z, training=False)
                test_inputs = torch.cat((test_inputs, preds), dim=2)
        return test_inputs

    @staticmethod
    def sample_image(videos, max_rows=10):
        """kernel_train.py:349-351: [B,H,T,W,C] -> one image [1, min(10,B)*H, W*T, C], a row of frames per sample."""
        B, H, T, W, C = videos.shape
        images = videos.reshape(B, H, W * T, C)
        return torch.cat(list(images[:min(max_rows, B)]), dim=0)[None]

    def fit(self, batched_x, test_x=None, init_sigma=5.0, decaying_sigma=False, save_freq=500, log=None,
            max_iterations=None):
        """The loop body of kernel_train.py:295-355 over an iterable of [B,H,T,W,C] batches.

        ``log(name, value, step)`` receives 'pM' and 'Sinkhorn Loss' every iteration (:318-321) and 'Training data'
        (the sample image) at iteration 1 and every ``save_freq`` iterations when ``test_x`` (one [B,H,T,W,C] batch)
        is given (:331,338-355).  A non-finite generator loss ends the run (:323-329).  Returns a dict with the
        iteration count, the scalar history and ``exploded``."""
        history = {"pM": [], "Sinkhorn Loss": []}
        it_counts, exploded = 0, False
        for x in batched_x:
            if x.shape[0] != self.batch_size:                                    # :298-299
                continue
            it_counts += 1
            real_data = x.to(self.device, torch.float32)
            sig = (self.gaussian_kernel.annealing_sigma(init_sigma, it_counts) if decaying_sigma else init_sigma)  # :308-311
            pm, loss = self.train_iteration(real_data, sig)
            pm, loss = float(pm), float(loss)
            history["pM"].append(pm)
            history["Sinkhorn Loss"].append(loss)
            if log is not None:
                log("pM", pm, it_counts)
                log("Sinkhorn Loss", loss, it_counts)
            if not math.isfinite(loss):                                          # :323
                gan_utils.raise_if_solver_aborted()        # an aborted multi-CU solve is an error, not an exploded loss
                exploded = True
                break
            if test_x is not None and (it_counts % save_freq == 0 or it_counts == 1) and log is not None:   # :331
                log("Training data", self.sample_image(self.sample(test_x)), it_counts)
            if max_iterations is not None and it_counts >= max_iterations:
                break
        return {"iterations": it_counts, "history": history, "exploded": exploded}

    def train_iteration(self, real_data, sigma=5.0):
        """One pass of the loop body kernel_train.py:301-314 on a [B,H,T,W,C] batch."""
        real_inputs = real_data[:, :, :self.int_time_steps]
        real_preds = real_data[:, :, self.int_time_steps:]
        pm = self.disc_training_step(real_inputs, real_preds, sigma)
        loss = self.gen_training_step(real_inputs, real_preds, sigma)
        return pm, loss
