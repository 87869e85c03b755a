"""CPU checks of the PyTorch G/D models and the LR schedule (f2): layer
stack, tensor layouts and shapes of the reference's gan.py; Keras numerics are parity-unpinned."""
import os
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


# ---------------------------------------------------------------- SURVEY.md section 8 f3 / f4
def _tiny_trainer(B=2, T=5, iT=2):
    from kccotgan_amd.kernel_train import KCCOTTrainer
    return KCCOTTrainer(B, total_time_steps=T, int_time_steps=iT, x_height=16, x_width=16, channels=1, g_state_size=4,
                        d_state_size=4, g_filter_size=1, d_filter_size=1, z_channels=3, device="cpu", seed=3)


def test_autoregressive_sampling_loop():
    """kernel_train.py:340-351: the context frames are kept, one frame is appended per step, output in [0,1]."""
    import numpy as np
    tr = _tiny_trainer()
    x = torch.rand(2, 16, 5, 16, 1)
    out = tr.sample(x)
    assert tuple(out.shape) == (2, 16, 5, 16, 1)
    assert torch.equal(out[:, :, :2], x[:, :, :2])
    assert float(out.min()) >= 0 and float(out.max()) <= 1
    # each appended frame depends only on the frames before it: re-running the first step alone reproduces frame 2
    torch.manual_seed(11)
    a = tr.sample(x)
    torch.manual_seed(11)
    feats = tr.context_encoder(x[:, :, :2], training=False)
    z = torch.randn(2, 1, 1, 1, 3)
    first = tr.decoder(feats, z, training=False)
    np.testing.assert_allclose(a[:, :, 2:3].numpy(), first.detach().numpy(), rtol=0, atol=1e-6)
    img = tr.sample_image(out)
    assert tuple(img.shape) == (1, 2 * 16, 16 * 5, 1)
    # row b of the image is sample b with its frames side by side (tf.reshape of [B,H,T,W,C] to [B,H,W*T,C])
    assert torch.equal(img[0, 16:32, 16:32, 0], out[1, :, 1, :, 0])


def test_dataset_adapters_layouts():
    import numpy as np
    from kccotgan_amd import datasets as ds
    rng = np.random.default_rng(0)
    mm = rng.integers(0, 256, size=(7, 5, 8, 6), dtype=np.uint8)            # [T_all, N, H, W]
    v = ds.mmnist_videos(mm, 4)
    assert v.shape == (5, 8, 4, 6) and v.dtype == np.float64
    for n, h, t, w in ((0, 0, 0, 0), (4, 7, 3, 5), (2, 3, 1, 4)):
        assert v[n, h, t, w] == mm[t, n, h, w] / 255.0
    mz = rng.random((3, 8, 9, 8, 3))
    assert np.array_equal(ds.mazes_test_videos(mz, 4), mz[:, :, :4])
    fr = rng.integers(0, 256, size=(6, 8, 8, 3), dtype=np.uint8)            # decoded frames [T,H,W,C]
    vid = ds.frames_to_video(fr, 5)
    assert vid.shape == (8, 5, 8, 3) and vid[3, 2, 4, 1] == fr[2, 3, 4, 1] / 255.0
    got = list(ds.batches(v, 2, 8, 4, 6, 1, epochs=2))
    assert len(got) == 4                                                      # 5 videos -> 2 full batches per epoch
    assert all(tuple(b.shape) == (2, 8, 4, 6, 1) and b.dtype == torch.float32 for b in got)
    assert torch.equal(got[1][1, :, :, :, 0], torch.from_numpy(v[3].astype(np.float32)))
    assert torch.equal(got[0], got[2])
    rgba = rng.random((2, 8, 4, 6, 4))
    b = next(ds.batches(rgba, 2, 8, 4, 6, 3))
    assert tuple(b.shape) == (2, 8, 4, 6, 3) and torch.equal(b, torch.from_numpy(rgba[..., :3].astype(np.float32)))
    sq = ds.synthetic_moving_squares(3, 16, 5, 16, seed=1)
    assert sq.shape == (5, 3, 16, 16) and sq.dtype == np.uint8 and set(np.unique(sq)) == {0, 255}
    assert all(int((sq[t, i] == 255).sum()) == 36 for t in range(5) for i in range(3))
    import pytest
    with pytest.raises(ValueError):
        ds.mmnist_videos(mm, 8)


def test_shared_adam_counts_two_apply_gradients_per_step_like_keras():
    """kernel_train.py:62-63,254-255,290-291: ONE Keras Adam per step kind, apply_gradients called once per
    network -> `iterations` advances by 2 per training step, the schedule is read at the pre-increment count
    (first update at lr(0) = 0), the bias correction uses t = iterations + 1.  The LR sequence is compared with the
    reference's own WarmUp class (tests/golden/lr_schedule.npz, run 'short': lr 1e-3, warmup 6, decay 4 / 0.5)."""
    import os
    import numpy as np
    import smooth_cases
    from kccotgan_amd.kernel_train import KerasSharedAdam, warmup_exponential_decay
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "lr_schedule.npz"))
    lr, warmup, decay_steps, rate = smooth_cases.LR_RUNS["short"]
    gold = dict(zip(g["steps"].tolist(), g["lr_short"].tolist()))
    opt = KerasSharedAdam(lambda it: warmup_exponential_decay(it, lr, warmup, decay_steps, rate), beta_1=0.5, beta_2=0.9)
    torch.manual_seed(0)
    a, b = torch.randn(5, dtype=torch.float64), torch.randn(3, dtype=torch.float64)
    a0, b0 = a.clone(), b.clone()
    ma = va = np.zeros(5); mb = vb = np.zeros(3)
    ra, rb = a0.numpy().copy(), b0.numpy().copy()
    for step in range(4):                  # iterations 0..7 (all in the fixture; warm-up ends at 6)
        ga, gb = torch.randn(5, dtype=torch.float64), torch.randn(3, dtype=torch.float64)
        opt.apply_gradients([(ga, a)])
        lr_a = opt.last_lr
        opt.apply_gradients([(gb, b)])
        lr_b = opt.last_lr
        assert opt.iterations == 2 * (step + 1)
        np.testing.assert_allclose([lr_a, lr_b], [gold[2 * step], gold[2 * step + 1]], rtol=2e-6)
        # Keras Adam._resource_apply_dense, written out
        for (gr, t, lr_t, which) in ((ga.numpy(), 2 * step + 1, lr_a, "a"), (gb.numpy(), 2 * step + 2, lr_b, "b")):
            alpha = lr_t * np.sqrt(1 - 0.9 ** t) / (1 - 0.5 ** t)
            if which == "a":
                ma = 0.5 * ma + 0.5 * gr; va = 0.9 * va + 0.1 * gr * gr; ra = ra - alpha * ma / (np.sqrt(va) + 1e-7)
            else:
                mb = 0.5 * mb + 0.5 * gr; vb = 0.9 * vb + 0.1 * gr * gr; rb = rb - alpha * mb / (np.sqrt(vb) + 1e-7)
        np.testing.assert_allclose(a.numpy(), ra, rtol=1e-12, atol=1e-15)
        np.testing.assert_allclose(b.numpy(), rb, rtol=1e-12, atol=1e-15)
        if step == 0:
            assert torch.equal(a, a0)          # lr(0) = 0: the first network's first update is a no-op
            assert not torch.equal(b, b0)


def test_shipped_miopen_find_db_is_installed_for_the_user(tmp_path, monkeypatch):
    """kccotgan_amd ships MIOpen's solver choices for the G/D convolutions (miopen_db/*.ufdb.txt): at import they are
    copied to a per-user cache directory and MIOPEN_USER_DB_PATH is pointed there, unless the user has a database of
    their own or opts out."""
    import kccotgan_amd
    monkeypatch.setenv("HOME", str(tmp_path))
    monkeypatch.delenv("MIOPEN_USER_DB_PATH", raising=False)
    monkeypatch.delenv("KCCOT_NO_MIOPEN_DB", raising=False)
    d = kccotgan_amd._install_miopen_find_db(force=True)
    assert d and d.startswith(str(tmp_path)) and os.environ["MIOPEN_USER_DB_PATH"] == d
    assert any(f.endswith(".ufdb.txt") for f in os.listdir(d))
    # an explicit choice of the user is left alone
    monkeypatch.setenv("MIOPEN_USER_DB_PATH", "/somewhere/else")
    assert kccotgan_amd._install_miopen_find_db(force=True) is None and os.environ["MIOPEN_USER_DB_PATH"] == "/somewhere/else"
    monkeypatch.delenv("MIOPEN_USER_DB_PATH")
    monkeypatch.setenv("KCCOT_NO_MIOPEN_DB", "1")
    assert kccotgan_amd._install_miopen_find_db(force=True) is None
    # a tuned database in MIOpen's default place wins
    monkeypatch.delenv("KCCOT_NO_MIOPEN_DB")
    own = tmp_path / ".config" / "miopen"
    own.mkdir(parents=True)
    (own / "gfx950100.HIP.x.ufdb.txt").write_text("")
    assert kccotgan_amd._install_miopen_find_db(force=True) is None
