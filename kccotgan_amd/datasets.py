"""Dataset adapters of the reference's ``train(args)``: array layouts in, ``[B,H,T,W,C]``
fp32 batches out -- the tensor the training-step API takes (kernel_train.py:302-307).

What the reference does with each source, restated on numpy arrays:

* Moving-MNIST ``mnist_training_set.npy`` / ``mnist_test_set.npy`` (kernel_train.py:99-111): ``[T_all, N, H, W]``
  uint8 -> ``/255`` -> first ``total_time_steps`` frames -> ``transpose(1,0,2,3)`` -> ``transpose(0,2,1,3)`` =
  ``[N, H, T, W]``; the batch is then reshaped to ``[B, H, T, W, -1]`` (one channel).
* GQN mazes test set ``np_mazes_test.npy`` (kernel_train.py:118-121): ``[N, H, T_all, W, C]`` float -> ``[:, :, :T]``.
* BAIR robot push (data_utils.py:63-104) and the GQN training reader (data_utils.py:355-449) decode TFRecords
  into per-video frame stacks ``[T, H, W, C]`` and emit ``transpose(1,0,2,3)`` = ``[H, T, W, C]`` scaled to [0,1].
  TFRecord / JPEG decoding needs TensorFlow and is out of scope here; ``frames_to_video``
  is the layout step for frames decoded elsewhere.

``batches`` is the loop head of kernel_train.py:297-307: fixed-size batches (short ones are skipped), reshape to
``[B, H, T, W, -1]``, drop channels beyond ``channels`` (the alpha channel), cast to fp32, ``epochs`` repeats.
"""
import numpy as np
import torch


def mmnist_videos(arr, total_time_steps):
    """kernel_train.py:99-104.  ``arr`` [T_all, N, H, W] (uint8 or float, 0..255) -> float64 [N, H, T, W] in [0,1]
    (``np.load(path) / 255.0`` is float64 in the reference; the cast to fp32 happens per batch, :304)."""
    arr = np.asarray(arr)
    if arr.ndim != 4:
        raise ValueError("Moving-MNIST array must be [T, N, H, W], got shape %r" % (arr.shape,))
    if total_time_steps > arr.shape[0]:
        raise ValueError("total_time_steps %d exceeds the %d frames stored" % (total_time_steps, arr.shape[0]))
    data = arr[:total_time_steps] / 255.0
    return np.ascontiguousarray(data.transpose(1, 0, 2, 3).transpose(0, 2, 1, 3))


def mazes_test_videos(arr, total_time_steps):
    """kernel_train.py:118-119.  ``arr`` [N, H, T_all, W, C] -> [N, H, T, W, C]."""
    arr = np.asarray(arr)
    if arr.ndim != 5:
        raise ValueError("mazes test array must be [N, H, T, W, C], got shape %r" % (arr.shape,))
    return arr[:, :, :total_time_steps, :, :]


def frames_to_video(frames, total_time_steps, scale=255.0):
    """data_utils.py:103-104 (BAIR) / :449 (GQN reader): decoded frames [T_all, H, W, C] -> [H, T, W, C];
    ``scale`` divides uint8 frames into [0,1] (pass 1.0 for frames that are already float images)."""
    frames = np.asarray(frames)
    if frames.ndim != 4:
        raise ValueError("frames must be [T, H, W, C], got shape %r" % (frames.shape,))
    video = frames.transpose(1, 0, 2, 3) / scale
    return video[:, :total_time_steps, :, :]


def batches(videos, batch_size, x_height, total_time_steps, x_width, channels, epochs=1, device=None):
    """Yield ``[B, H, T, W, C]`` fp32 tensors from ``videos`` ([N, H, T, W] or [N, H, T, W, C'] array, or any
    iterable of per-video arrays), ``epochs`` times over (kernel_train.py:104,297-304).  A trailing batch with
    fewer than ``batch_size`` videos is skipped (:298-299); channels beyond ``channels`` are dropped (:304)."""
    for _ in range(epochs):
        buf = []
        for v in videos:
            buf.append(np.asarray(v))
            if len(buf) == batch_size:
                x = np.stack(buf)
                buf = []
                x = x.reshape(batch_size, x_height, total_time_steps, x_width, -1)[..., :channels]
                t = torch.from_numpy(np.ascontiguousarray(x, dtype=np.float32))
                yield t.to(device) if device is not None else t


def synthetic_moving_squares(n, x_height, total_time_steps, x_width, seed=0, size=6):
    """Stand-in videos of the Moving-MNIST layout ([T, N, H, W] uint8, bright squares bouncing off the frame
    edges) for smoke runs on a box without datasets; feed to ``mmnist_videos``."""
    rng = np.random.default_rng(seed)
    out = np.zeros((total_time_steps, n, x_height, x_width), dtype=np.uint8)
    pos = rng.integers(0, [x_height - size, x_width - size], size=(n, 2)).astype(np.int64)
    vel = rng.integers(1, 4, size=(n, 2)) * rng.choice([-1, 1], size=(n, 2))
    lim = np.array([x_height - size, x_width - size])
    for t in range(total_time_steps):
        for i in range(n):
            r, c = pos[i]
            out[t, i, r:r + size, c:c + size] = 255
        pos = pos + vel
        low, high = pos < 0, pos > lim
        pos = np.where(low, -pos, np.where(high, 2 * lim - pos, pos))
        vel = np.where(low | high, -vel, vel)
    return out
