"""Generator (ConvLSTM encoder / decoder) and discriminators of KCCOT-GAN on stock PyTorch-ROCm
ops -- SURVEY.md section 8 row f1.  The north star assigns these to PyTorch (MIOpen convolutions,
nn.LSTM); only the loss path is hand-written HIP.

Mirrors the layer stack of the reference's gan.py (class names, constructor arguments, tensor
layouts at the boundary):
  VideoEncoderConvLSTM  gan.py:9-113   four strided ConvLSTM2D (6,6,5,5 kernels, stride 2, filters
                                       fs*{4,8,16,32}, tanh, no bias) each followed by a LayerNorm over
                                       channels; returns the frames int_T-1.. of the input and of
                                       every level
  VideoDecoderConvLSTM  gan.py:116-364 Conv2DTranspose <-> ConvLSTM2D ladder with skip-concats from
                                       the encoder levels, noise z joined at the 4x4 level, sigmoid out
  VideoDiscriminator    gan.py:367-428 per-frame 3x (Conv 5x5 s2 + BN + LeakyReLU) -> LSTM fs*8 ->
                                       BN -> LSTM fs*4 -> BN -> LSTM J with sigmoid cell activation

Videos cross the boundary in the reference's [B, H, T, W, C] layout.  Numerics against Keras are
PARITY UNPINNED (no TensorFlow, no weights, no tests in the reference): padding follows TF 'same'
arithmetic, recurrent activation is Keras' hard_sigmoid, initialisation is PyTorch's default.
"""
import contextlib
import math
import os

import torch
import torch.nn as nn
import torch.nn.functional as F

# The MIOpen solver switch that keeps these models' backward off the faulting `igemm_bwd_gtcx35_nhwc_fp32_*` kernel is
# set by the package's __init__ (it must precede the first convolution of the process).  Where that cannot be
# guaranteed -- the variable was set to something else by the user, or the GPU was already in use when the package
# was imported -- the default falls back to the conservative mode below (no MIOpen kernel at all).
from . import MIOPEN_SWITCH, MIOPEN_WORKAROUND_GUARANTEED

# Layer families listed in KCCOT_NATIVE_CONV (comma separated: convlstm, deconv, dconv) run on the native ATen
# convolution kernels instead of MIOpen's (the conservative mode: no MIOpen kernel at all when all three are listed;
# the forward contexts below cover the forward calls, `conv_guard` the backward).  Default: none when the MIOpen
# workaround is in effect, all three otherwise.
_NATIVE_DEFAULT = "" if MIOPEN_WORKAROUND_GUARANTEED else "convlstm,deconv,dconv"
_NATIVE = set(filter(None, os.environ.get("KCCOT_NATIVE_CONV", _NATIVE_DEFAULT).split(",")))


def convolution_mode():
    """Which kernels the G/D convolutions run on: 'miopen' (the fast default), or 'native:<families>' when layer families
    were moved to the ATen kernels (KCCOT_NATIVE_CONV, or the import-order fallback kccotgan_amd/__init__.py warns about).
    Reported by KCCOTTrainer.convolution_mode, tools/bench_train.py and bench.py's train_steps_per_sec block."""
    return "native:" + ",".join(sorted(_NATIVE)) if _NATIVE else "miopen"


def _backend(kind):
    return torch.backends.cudnn.flags(enabled=False) if kind in _NATIVE else contextlib.nullcontext()


def conv_guard():
    """Context for everything that runs these models' BACKWARD (``loss.backward()`` / ``torch.autograd.grad``).
    The per-family contexts above only cover the forward calls: ``convolution_backward`` picks its backend again,
    from the process-wide flag, when the autograd engine runs it -- outside any forward-time context.  (Found with
    AMD_LOG_LEVEL=3 + blocking launches: the faulting kernel of a 'native' run was MIOpen's
    ``igemm_bwd_gtcx35_nhwc_fp32`` in the generator's backward.)  With any family listed in KCCOT_NATIVE_CONV the
    guard switches MIOpen off for the whole region; KernelTrainer wraps both training steps and ``sample`` in it."""
    return torch.backends.cudnn.flags(enabled=False) if _NATIVE else contextlib.nullcontext()


def _same_pad(size, k, s):
    """TF 'same' padding along one axis: (before, after)."""
    out = math.ceil(size / s)
    total = max((out - 1) * s + k - size, 0)
    return total // 2, total - total // 2


def hard_sigmoid(x):
    """Keras recurrent_activation default of ConvLSTM2D: clip(0.2 x + 0.5, 0, 1)."""
    return torch.clamp(0.2 * x + 0.5, 0.0, 1.0)


class _ChannelLayerNormHIP(torch.autograd.Function):
    """LayerNorm over the channels of an NCHW tensor on its native layout (kccot_channel_layernorm_{fwd,bwd}_f32)."""

    @staticmethod
    def forward(ctx, x, gamma, beta, eps):
        from ._lib import lib, ptr, check, stream_of, empty_like, empty
        x = x.contiguous()
        N, C, HW = x.shape[0], x.shape[1], x.shape[2] * x.shape[3]
        y = empty_like(x)
        mean, rstd = empty((N, HW), torch.float32, x.device), empty((N, HW), torch.float32, x.device)
        check(lib.kccot_channel_layernorm_fwd_f32(ptr(x), ptr(gamma), ptr(beta), N, C, HW, eps, ptr(y), ptr(mean), ptr(rstd),
                                                  stream_of(x)), "channel_layernorm_fwd")
        ctx.save_for_backward(x, gamma, mean, rstd)
        return y

    @staticmethod
    def backward(ctx, dy):
        from ._lib import lib, ptr, check, stream_of, empty_like, empty
        x, gamma, mean, rstd = ctx.saved_tensors
        dy = dy.contiguous()
        N, C, HW = x.shape[0], x.shape[1], x.shape[2] * x.shape[3]
        dx = empty_like(x)
        parts = empty((int(lib.kccot_channel_layernorm_chunks(N, C, HW)), 2, C), torch.float32, x.device)
        check(lib.kccot_channel_layernorm_bwd_f32(ptr(dy), ptr(x), ptr(gamma), ptr(mean), ptr(rstd), N, C, HW, ptr(dx), ptr(parts),
                                                  stream_of(x)), "channel_layernorm_bwd")
        sums = parts.sum(0)
        return dx, sums[0], sums[1], None


# KCCOT_CHANNEL_LN=torch: nn.LayerNorm on the permuted tensor (also what CPU tensors take)
_LN_HIP = os.environ.get("KCCOT_CHANNEL_LN", "hip") != "torch"


class ChannelLayerNorm(nn.Module):
    """tf.keras.layers.LayerNormalization(axis=[-1]) on a channels-last tensor, applied to NCHW data."""

    def __init__(self, channels):
        super().__init__()
        self.ln = nn.LayerNorm(channels, eps=1e-3)   # Keras epsilon default; holds gamma / beta

    def forward(self, x):                             # [N, C, H, W]
        if _LN_HIP and x.is_cuda and x.dtype == torch.float32:
            return _ChannelLayerNormHIP.apply(x, self.ln.weight, self.ln.bias, self.ln.eps)
        return self.ln(x.permute(0, 2, 3, 1)).permute(0, 3, 1, 2)


class _ConvLSTMCellHIP(torch.autograd.Function):
    """c, h = cell(gx_t, gh, c_prev): the gate arithmetic of one ConvLSTM step as ONE HIP launch each way
    (kccot_convlstm_cell_{fwd,bwd}_f32) instead of a dozen elementwise kernels forward and twice that backward."""

    @staticmethod
    def forward(ctx, gx, gh, c_prev):
        from ._lib import lib, ptr, check, stream_of, empty_like
        gx, gh, c_prev = gx.contiguous(), gh.contiguous(), c_prev.contiguous()
        B, F4 = gx.shape[0], gx.shape[1]
        Fn, HW = F4 // 4, gx.shape[2] * gx.shape[3]
        c, h = empty_like(c_prev), empty_like(c_prev)
        check(lib.kccot_convlstm_cell_fwd_f32(ptr(gx), ptr(gh), ptr(c_prev), B, Fn, HW, ptr(c), ptr(h), stream_of(gx)), "convlstm_cell_fwd")
        ctx.save_for_backward(gx, gh, c_prev, c)
        return c, h

    @staticmethod
    def backward(ctx, dc, dh):
        from ._lib import lib, ptr, check, stream_of, empty_like
        gx, gh, c_prev, c = ctx.saved_tensors
        B, F4 = gx.shape[0], gx.shape[1]
        Fn, HW = F4 // 4, gx.shape[2] * gx.shape[3]
        dc = dc.contiguous() if dc is not None else None
        dh = dh.contiguous() if dh is not None else None
        dg, dcp = empty_like(gx), empty_like(c_prev)
        check(lib.kccot_convlstm_cell_bwd_f32(ptr(gx), ptr(gh), ptr(c_prev), ptr(c), ptr(dh) if dh is not None else None,
                                              ptr(dc) if dc is not None else None, B, Fn, HW, ptr(dg), ptr(dcp), stream_of(gx)),
              "convlstm_cell_bwd")
        return dg, dg, dcp


def _cell_torch(g, c):
    gi, gf, gc, go = torch.chunk(g, 4, dim=1)
    c = hard_sigmoid(gf) * c + hard_sigmoid(gi) * torch.tanh(gc)
    return c, hard_sigmoid(go) * torch.tanh(c)


# KCCOT_CONVLSTM_CELL=torch: the stock tensor-op cell (also what CPU tensors and double precision take)
_CELL_HIP = os.environ.get("KCCOT_CONVLSTM_CELL", "hip") != "torch"


class ConvLSTM2D(nn.Module):
    """Keras ConvLSTM2D(filters, kernel, strides, padding='same', return_sequences=True):
    the input convolution is strided, the recurrent one runs on the hidden state at stride 1;
    gate order i, f, c, o; h = o * act(c)."""

    def __init__(self, in_ch, filters, kernel, stride, in_hw, bias=False):
        super().__init__()
        self.filters, self.k, self.s = filters, kernel, stride
        self.in_hw = in_hw
        self.out_hw = (math.ceil(in_hw[0] / stride), math.ceil(in_hw[1] / stride))
        self.wx = nn.Conv2d(in_ch, 4 * filters, kernel, stride, bias=bias)
        self.wh = nn.Conv2d(filters, 4 * filters, kernel, 1, bias=False)
        ph, pw = _same_pad(in_hw[0], kernel, stride), _same_pad(in_hw[1], kernel, stride)
        self.pad_x = (pw[0], pw[1], ph[0], ph[1])
        qh, qw = _same_pad(self.out_hw[0], kernel, 1), _same_pad(self.out_hw[1], kernel, 1)
        self.pad_h = (qw[0], qw[1], qh[0], qh[1])

    def forward(self, x):                             # [B, T, C, H, W] -> [B, T, F, H', W']
        B, T = x.shape[:2]
        # all frames in one conv, TIME-MAJOR: the per-step slices below are then whole contiguous tensors taken with
        # unbind (its backward is ONE stack) -- slicing a batch-major [B, T, ...] result per step costs a zero-filled
        # full-size gradient tensor and an add per step in the backward (7 % + 2 % of the training iteration's GPU time)
        xt = x.transpose(0, 1).reshape((T * B,) + x.shape[2:])
        with _backend("convlstm"):
            gx = self.wx(F.pad(xt, self.pad_x))
        gxs = gx.reshape(T, B, 4 * self.filters, *self.out_hw).unbind(0)
        h = x.new_zeros(B, self.filters, *self.out_hw)
        c = torch.zeros_like(h)
        outs = []
        for t in range(T):
            with _backend("convlstm"):
                gh = self.wh(F.pad(h, self.pad_h))
            if _CELL_HIP and gh.is_cuda and gh.dtype == torch.float32:
                c, h = _ConvLSTMCellHIP.apply(gxs[t], gh, c)
            else:
                c, h = _cell_torch(gxs[t] + gh, c)
            outs.append(h)
        return torch.stack(outs, dim=1)


class _SameConvTranspose(nn.Module):
    """Conv2DTranspose(filters, k, strides=s, padding='same', use_bias=False): output = input * s."""

    def __init__(self, in_ch, filters, k, s):
        super().__init__()
        self.s = s
        if s == 1:
            self.pad = _same_pad(1 << 20, k, 1)
            self.conv = nn.Conv2d(in_ch, filters, k, 1, bias=False)
        else:
            assert (k - s) % 2 == 0, "even kernel/stride difference expected (gan.py:160-168)"
            self.conv = nn.ConvTranspose2d(in_ch, filters, k, s, padding=(k - s) // 2, bias=False)

    def forward(self, x):
        with _backend("deconv"):
            if self.s == 1:
                return self.conv(F.pad(x, (self.pad[0], self.pad[1], self.pad[0], self.pad[1])))
            return self.conv(x)


def _to_frames(video):
    """[B, H, T, W, C] -> [B, T, C, H, W]   (gan.py:88 transposes to [B,T,H,W,C], channels last)."""
    return video.permute(0, 2, 4, 1, 3)


def _to_video(frames):
    """[B, T, C, H, W] -> [B, H, T, W, C]   (gan.py:359-360)."""
    return frames.permute(0, 3, 1, 4, 2)


class VideoEncoderConvLSTM(nn.Module):
    """gan.py:9-113."""

    def __init__(self, batch_size, int_time_steps, pred_time_steps, state_size, x_width, x_height, z_width=5,
                 z_height=5, filter_size=64, bn=False, nlstm=1, cat=False, nchannel=3, dropout=0.0, rnn_dropout=0.0,
                 reg=False, cw=False, period=(1, 2, 4)):
        super().__init__()
        self.int_time_steps, self.rnn_bn = int_time_steps, bn
        fs, hw = filter_size, (x_height, x_width)
        specs = [(nchannel, fs * 4, 6), (fs * 4, fs * 8, 6), (fs * 8, fs * 16, 5), (fs * 16, fs * 32, 5)]
        self.enc, self.norms = nn.ModuleList(), nn.ModuleList()
        for cin, f, k in specs:
            layer = ConvLSTM2D(cin, f, k, 2, hw, bias=False)
            hw = layer.out_hw
            self.enc.append(layer)
            self.norms.append(ChannelLayerNorm(f))

    def forward(self, inputs_real, training=True):
        """inputs_real [B,H,T,W,C] -> list of 5 tensors [B, T-int_T+1, C_l, H_l, W_l]: the input
        frames and the four encoder levels from frame int_T-1 on (gan.py:87-110)."""
        x = _to_frames(inputs_real)
        k = self.int_time_steps - 1
        feats = [x[:, k:]]
        for layer, norm in zip(self.enc, self.norms):
            x = layer(x)
            if self.rnn_bn:
                B, T = x.shape[:2]
                x = norm(x.reshape((B * T,) + x.shape[2:])).reshape(x.shape)
            feats.append(x[:, k:])
        return feats

    call = forward


class VideoDecoderConvLSTM(nn.Module):
    """gan.py:116-364 (square frames: kernel / stride table of gan.py:160-168)."""

    def __init__(self, batch_size, int_time_steps, pred_time_steps, state_size, x_width, x_height, z_width=5,
                 z_height=5, filter_size=64, bn=False, output_activation="sigmoid", nlstm=1, cat=False, nchannel=3,
                 dropout=0.0, reg=False, rnn_dropout=0.0, cw=False, period=(1, 2, 4), z_channels=128):
        super().__init__()
        if x_height != x_width:
            raise NotImplementedError("only the square-frame branch of gan.py:160-168 is built")
        fs, self.rnn_bn, self.nchannel = filter_size, bn, nchannel
        self.output_activation = output_activation
        h4 = x_height // 16                                   # encoder level 4 resolution
        self.ct1 = _SameConvTranspose(fs * 32 + z_channels, fs * 32, 2, 2)
        self.n1 = ChannelLayerNorm(fs * 32)
        self.dec2 = ConvLSTM2D(fs * 16 + fs * 32, fs * 16, 4, 1, (2 * h4, 2 * h4), bias=False)
        self.n5 = ChannelLayerNorm(fs * 16)
        self.ct2 = _SameConvTranspose(fs * 16, fs * 16, 4, 2)
        self.n2 = ChannelLayerNorm(fs * 16)
        self.dec3 = ConvLSTM2D(fs * 8 + fs * 16, fs * 8, 6, 1, (4 * h4, 4 * h4), bias=False)
        self.n6 = ChannelLayerNorm(fs * 8)
        self.ct3 = _SameConvTranspose(fs * 8, fs * 8, 6, 2)
        self.n3 = ChannelLayerNorm(fs * 8)
        self.dec4 = ConvLSTM2D(fs * 4 + fs * 8, fs * 4, 8, 1, (8 * h4, 8 * h4), bias=True)
        self.n7 = ChannelLayerNorm(fs * 4)
        self.ct4 = _SameConvTranspose(fs * 4, fs * 2, 6, 2)
        self.n4 = ChannelLayerNorm(fs * 2)
        self.dec5 = ConvLSTM2D(nchannel + fs * 2, fs, 8, 1, (16 * h4, 16 * h4), bias=True)
        self.n8 = ChannelLayerNorm(fs)
        self.ct5 = _SameConvTranspose(fs, nchannel, 8, 1)

    def _per_frame(self, x, conv, norm):
        B, T = x.shape[:2]
        y = torch.tanh(conv(x.reshape((B * T,) + x.shape[2:])))
        if self.rnn_bn:
            y = norm(y)
        return y.reshape((B, T) + y.shape[1:])

    def _lstm(self, x, lstm, norm):
        y = lstm(x)
        if self.rnn_bn:
            B, T = y.shape[:2]
            y = norm(y.reshape((B * T,) + y.shape[2:])).reshape(y.shape)
        return y

    def forward(self, predictions, inputs_z, training=True):
        """predictions: the encoder's 5-level list; inputs_z [B, T', 4, 4, z_channels] (channels
        last, as dist_z.sample gives it, kernel_train.py:220).  Returns [B,H,T',W,C]."""
        sel = (lambda f: f[:, :-1]) if training else (lambda f: f[:, -1:])     # gan.py:269-272
        z = inputs_z.permute(0, 1, 4, 2, 3)
        x = torch.cat((sel(predictions[4]), z), dim=2)
        x = self._per_frame(x, self.ct1, self.n1)
        x = self._lstm(torch.cat((sel(predictions[3]), x), dim=2), self.dec2, self.n5)
        x = self._per_frame(x, self.ct2, self.n2)
        x = self._lstm(torch.cat((sel(predictions[2]), x), dim=2), self.dec3, self.n6)
        x = self._per_frame(x, self.ct3, self.n3)
        x = self._lstm(torch.cat((sel(predictions[1]), x), dim=2), self.dec4, self.n7)
        x = self._per_frame(x, self.ct4, self.n4)
        x = self._lstm(torch.cat((sel(predictions[0]), x), dim=2), self.dec5, self.n8)
        B, T = x.shape[:2]
        y = self.ct5(x.reshape((B * T,) + x.shape[2:]))
        y = torch.sigmoid(y) if self.output_activation == "sigmoid" else y
        return _to_video(y.reshape((B, T) + y.shape[1:]))

    call = forward


class _SigmoidLSTM(nn.Module):
    """Keras LSTM(units, activation='sigmoid', return_sequences=True) (gan.py:418): the cell
    candidate and the output use sigmoid instead of tanh, which nn.LSTM cannot express."""

    def __init__(self, in_features, units):
        super().__init__()
        self.units = units
        self.wx = nn.Linear(in_features, 4 * units)
        self.wh = nn.Linear(units, 4 * units, bias=False)

    def forward(self, x):                              # [B, T, F]
        B, T, _ = x.shape
        gx = self.wx(x)
        h = x.new_zeros(B, self.units)
        c = torch.zeros_like(h)
        outs = []
        for t in range(T):
            gi, gf, gc, go = torch.chunk(gx[:, t] + self.wh(h), 4, dim=1)
            c = torch.sigmoid(gf) * c + torch.sigmoid(gi) * torch.sigmoid(gc)
            h = torch.sigmoid(go) * torch.sigmoid(c)
            outs.append(h)
        return torch.stack(outs, dim=1)


class VideoDiscriminator(nn.Module):
    """gan.py:367-428: h or M features [B, T, J] from a [B,H,T,W,C] video."""

    def __init__(self, batch_size, time_steps, state_size, x_width, x_height, z_width=5, z_height=5, filter_size=64,
                 bn=False, output_activation="sigmoid", nlstm=1, cat=False, nchannel=3):
        super().__init__()
        fs, self.bn = filter_size, bn
        hw = (x_height, x_width)
        convs, cin = [], nchannel
        self.pads = []
        for f in (fs * 4, fs * 8, fs * 16):
            ph, pw = _same_pad(hw[0], 5, 2), _same_pad(hw[1], 5, 2)
            self.pads.append((pw[0], pw[1], ph[0], ph[1]))
            convs.append(nn.Conv2d(cin, f, 5, 2))
            hw = (math.ceil(hw[0] / 2), math.ceil(hw[1] / 2))
            cin = f
        self.convs = nn.ModuleList(convs)
        self.conv_bn = nn.ModuleList([nn.BatchNorm2d(c.out_channels, eps=1e-3, momentum=0.01) for c in convs])
        feat = cin * hw[0] * hw[1]
        self.rnn1 = nn.LSTM(feat, fs * 8, batch_first=True)
        self.rnn2 = nn.LSTM(fs * 8, fs * 4, batch_first=True)
        self.rnn_bn = nn.ModuleList([nn.BatchNorm1d(fs * 8, eps=1e-3, momentum=0.01),
                                     nn.BatchNorm1d(fs * 4, eps=1e-3, momentum=0.01)])
        self.rnn3 = _SigmoidLSTM(fs * 4, state_size) if output_activation == "sigmoid" else nn.LSTM(fs * 4, state_size, batch_first=True)

    def forward(self, inputs, training=True, mask=None):
        x = _to_frames(inputs)
        B, T = x.shape[:2]
        z = x.reshape((B * T,) + x.shape[2:])
        for conv, bn, pad in zip(self.convs, self.conv_bn, self.pads):
            with _backend("dconv"):
                z = conv(F.pad(z, pad))
            if self.bn:
                z = bn(z)
            z = F.leaky_relu(z, 0.3)                   # Keras LeakyReLU default alpha
        # Keras flattens channels-last: [B*T, H, W, C] -> [B, T, H*W*C]
        z = z.permute(0, 2, 3, 1).reshape(B, T, -1)
        z, _ = self.rnn1(z)
        if self.bn:
            z = self.rnn_bn[0](z.reshape(B * T, -1)).reshape(B, T, -1)
        z, _ = self.rnn2(z)
        if self.bn:
            z = self.rnn_bn[1](z.reshape(B * T, -1)).reshape(B, T, -1)
        z = self.rnn3(z)
        return z[0] if isinstance(z, tuple) else z

    call = forward
