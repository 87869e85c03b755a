"""CPU checks of the PyTorch G/D models and the LR schedule (f2): layer
stack, tensor layouts and shapes of the reference's gan.py; Keras numerics are parity-unpinned."""
import torch

from kccotgan_amd import gan
from kccotgan_amd.kernel_train import warmup_exponential_decay


def test_same_padding_matches_tf_arithmetic():
    assert gan._same_pad(64, 6, 2) == (2, 2) and gan._same_pad(16, 5, 2) == (1, 2) and gan._same_pad(8, 4, 1) == (1, 2)


def test_encoder_decoder_discriminator_shapes_and_gradients():
    B, H, W, C, T, iT, fs, zc, J = 2, 16, 16, 1, 5, 2, 1, 3, 4
    enc = gan.VideoEncoderConvLSTM(B, iT, T - iT, 8, W, H, z_width=1, z_height=1, filter_size=fs, bn=True, nchannel=C)
    dec = gan.VideoDecoderConvLSTM(B, iT, T - iT, 8, W, H, z_width=1, z_height=1, filter_size=fs, bn=True, nchannel=C,
                                   z_channels=zc)
    dis = gan.VideoDiscriminator(B, T, J, W, H, filter_size=fs, bn=True, nchannel=C)
    x = torch.rand(B, H, T, W, C)
    feats = enc(x)
    # gan.py:87-110: frames int_T-1.. of the input and of the four levels (stride 2 each)
    assert [tuple(f.shape) for f in feats] == [(B, T - iT + 1, C, 16, 16), (B, 4, 4 * fs, 8, 8), (B, 4, 8 * fs, 4, 4),
                                               (B, 4, 16 * fs, 2, 2), (B, 4, 32 * fs, 1, 1)]
    z = torch.randn(B, T - iT, 1, 1, zc)
    fake_pred = dec(feats, z)
    assert tuple(fake_pred.shape) == (B, H, T - iT, W, C)          # gan.py:359-360
    assert float(fake_pred.min()) >= 0 and float(fake_pred.max()) <= 1   # sigmoid output
    fake = torch.cat((x[:, :, :iT], fake_pred), dim=2)
    h = dis(fake)
    assert tuple(h.shape) == (B, T, J) and float(h.min()) >= 0 and float(h.max()) <= 1   # gan.py:418 sigmoid cell
    (h.sum() + fake_pred.mean()).backward()
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in list(enc.parameters()) + list(dec.parameters()))
    # inference path feeds only the last encoded frame (gan.py:269-272)
    one = dec(feats, torch.randn(B, 1, 1, 1, zc), training=False)
    assert tuple(one.shape) == (B, H, 1, W, C)


def test_default_layer_stack_of_the_reference():
    """filter_size 8 (kernel_train.py:370,372): channel widths and kernel sizes of gan.py:50-85,194-266,392-418."""
    enc = gan.VideoEncoderConvLSTM(2, 5, 10, 8, 64, 64, filter_size=8, bn=True, nchannel=1)
    assert [(l.wx.in_channels, l.filters, l.k, l.s) for l in enc.enc] == [(1, 32, 6, 2), (32, 64, 6, 2), (64, 128, 5, 2),
                                                                         (128, 256, 5, 2)]
    assert all(l.wx.bias is None for l in enc.enc)
    dec = gan.VideoDecoderConvLSTM(2, 5, 10, 8, 64, 64, filter_size=8, bn=True, nchannel=1, z_channels=128)
    assert [(l.wx.in_channels, l.filters, l.k) for l in (dec.dec2, dec.dec3, dec.dec4, dec.dec5)] == [
        (384, 128, 4), (192, 64, 6), (96, 32, 8), (17, 8, 8)]
    assert dec.dec2.wx.bias is None and dec.dec4.wx.bias is not None      # use_bias=False only on decoder2/3
    dis = gan.VideoDiscriminator(2, 15, 8, 64, 64, filter_size=8, bn=True, nchannel=1)
    assert [c.out_channels for c in dis.convs] == [32, 64, 128]
    assert (dis.rnn1.input_size, dis.rnn1.hidden_size, dis.rnn2.hidden_size, dis.rnn3.units) == (8 * 8 * 128, 64, 32, 8)


def test_lr_schedule():
    # data_utils.py:599-612 linear warm-up, then kernel_train.py:57 staircase decay 0.975 every 5000 steps
    assert warmup_exponential_decay(0, 5e-4) == 0.0
    assert abs(warmup_exponential_decay(5000, 5e-4) - 2.5e-4) < 1e-12
    assert abs(warmup_exponential_decay(10000, 5e-4) - 5e-4) < 1e-12
    assert abs(warmup_exponential_decay(14999, 5e-4) - 5e-4) < 1e-12
    assert abs(warmup_exponential_decay(15000, 5e-4) - 5e-4 * 0.975) < 1e-12
