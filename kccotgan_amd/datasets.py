"""Dataset adapters of the reference's ``train(args)``: array layouts in, ``[B,H,T,W,C]``
fp32 batches out -- the tensor the training-step API takes (kernel_train.py:302-307).

What the reference does with each source, restated on numpy arrays:

* Moving-MNIST ``mnist_training_set.npy`` / ``mnist_test_set.npy`` (kernel_train.py:99-111): ``[T_all, N, H, W]``
  uint8 -> ``/255`` -> first ``total_time_steps`` frames -> ``transpose(1,0,2,3)`` -> ``transpose(0,2,1,3)`` =
  ``[N, H, T, W]``; the batch is then reshaped to ``[B, H, T, W, -1]`` (one channel).
* GQN mazes test set ``np_mazes_test.npy`` (kernel_train.py:118-121): ``[N, H, T_all, W, C]`` float -> ``[:, :, :T]``.
* BAIR robot push (data_utils.py:63-104) and the GQN training reader (data_utils.py:355-449) decode TFRecords
  into per-video frame stacks ``[T, H, W, C]`` and emit ``transpose(1,0,2,3)`` = ``[H, T, W, C]`` scaled to [0,1]:
  ``robot_push_videos`` and ``gqn_videos`` do the same from the files themselves, with the TFRecord container and the
  ``tf.train`` messages read by ``kccotgan_amd/tfrecord.py`` (no TensorFlow) and the GQN JPEG frames decoded by PIL
  (libjpeg: may differ from TensorFlow's decoder by one grey level on some pixels).  ``frames_to_video`` is the layout
  step for frames decoded elsewhere.

``batches`` is the loop head of kernel_train.py:297-307: fixed-size batches (short ones are skipped), reshape to
``[B, H, T, W, -1]``, drop channels beyond ``channels`` (the alpha channel), cast to fp32, ``epochs`` repeats.
"""
import collections
import io
import os

import numpy as np
import torch

from . import tfrecord


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


def robot_push_videos(files, T=30, frames_per_video=30, img_shape=(64, 64, 3), camera="image_aux1"):
    """``robot_push_data`` (data_utils.py:63-104) over the given TFRecord files (the reference lists
    ``../data/softmotion30_44k/{train,test}/`` sorted by name): every record is a ``tf.train.SequenceExample`` whose CONTEXT
    holds, per frame i, ``"{i}/image_aux1/encoded"`` = the raw uint8 bytes of a 64 x 64 x 3 image (:95-98).  Yields one
    float64 ``[H, T, W, C]`` video in [0, 1] per record (``np.stack(frames).transpose(1, 0, 2, 3) / 255.0``, first T
    frames, :103-104)."""
    n = int(np.prod(img_shape))
    for path in files:
        for payload in tfrecord.records(path):
            context, _ = tfrecord.parse_sequence_example(payload)
            frames = []
            for i in range(frames_per_video):
                key = "%d/%s/encoded" % (i, camera)
                if key not in context or not context[key]:
                    raise ValueError("%s: record without the feature %r" % (path, key))
                img = np.frombuffer(context[key][0], dtype=np.uint8)
                if img.size != n:
                    raise ValueError("%s: %r holds %d bytes, expected %d" % (path, key, img.size, n))
                frames.append(img.reshape(img_shape))
            yield (np.stack(frames).transpose(1, 0, 2, 3) / 255.0)[:, :T, :, :]


# The GQN datasets of "Neural scene representation and rendering" as the reference's reader knows them
# (data_utils.py:281-330): directory, number of train / test files, frame size, views per scene.
GqnInfo = collections.namedtuple("GqnInfo", ["basepath", "train_size", "test_size", "frame_size", "sequence_size"])
GQN_DATASETS = {
    "jaco": GqnInfo("jaco", 3600, 400, 64, 11),
    "mazes": GqnInfo("mazes", 1080, 120, 84, 300),
    "rooms_free_camera_with_object_rotations": GqnInfo("rooms_free_camera_with_object_rotations", 2034, 226, 128, 10),
    "rooms_ring_camera": GqnInfo("rooms_ring_camera", 2160, 240, 64, 10),
    "rooms_free_camera_no_object_rotations": GqnInfo("rooms_free_camera_no_object_rotations", 2160, 240, 64, 10),
    "shepard_metzler_5_parts": GqnInfo("shepard_metzler_5_parts", 900, 100, 64, 15),
    "shepard_metzler_7_parts": GqnInfo("shepard_metzler_7_parts", 900, 100, 64, 15),
}


def gqn_files(dataset, root, mode="train"):
    """File list of one GQN dataset (data_utils.py:336-348): ``root/<basepath>/<mode>/0001-of-1080.tfrecord`` ..."""
    if dataset not in GQN_DATASETS:
        raise ValueError("Unrecognized dataset %s requested. Available datasets are %s" % (dataset, sorted(GQN_DATASETS)))
    if mode not in ("train", "test"):
        raise ValueError("Unsupported mode %s requested. Supported modes are ('train', 'test')" % mode)
    info = GQN_DATASETS[dataset]
    count = info.train_size if mode == "train" else info.test_size
    width = len(str(count))
    return [os.path.join(root, info.basepath, mode, "%0*d-of-%0*d.tfrecord" % (width, i + 1, width, count))
            for i in range(count)]


def gqn_videos(files, dataset, time_steps, custom_frame_size=None):
    """``DataReader.provide_dataset`` (data_utils.py:355-449) without its shuffling: every record is a
    ``tf.train.Example`` whose ``frames`` feature holds ``sequence_size`` JPEG strings (:431); each is decoded, converted
    to float32 in [0, 1] (``tf.image.convert_image_dtype``, :350-352), optionally resized bilinearly to
    ``custom_frame_size`` (``tf.image.resize``, half-pixel centres, :443-447), cut to ``time_steps`` views and emitted as
    ``[H, T, W, C]`` (:449).  Yields one float32 video per record."""
    from PIL import Image
    info = GQN_DATASETS[dataset]
    if time_steps > info.sequence_size:
        raise ValueError("Maximum support context size for dataset %s is %d, but was %d."
                         % (dataset, info.sequence_size, time_steps))
    for path in files:
        for payload in tfrecord.records(path):
            frames = tfrecord.parse_example(payload).get("frames", [])
            if len(frames) != info.sequence_size:
                raise ValueError("%s: record with %d frames, expected %d" % (path, len(frames), info.sequence_size))
            imgs = np.stack([np.asarray(Image.open(io.BytesIO(j)).convert("RGB")) for j in frames])
            if imgs.shape[1:] != (info.frame_size, info.frame_size, 3):
                raise ValueError("%s: frames of shape %r, expected %d x %d x 3" % (path, imgs.shape[1:], info.frame_size, info.frame_size))
            video = imgs.astype(np.float32) * np.float32(1.0 / 255.0)
            if custom_frame_size and custom_frame_size != info.frame_size:
                t = torch.from_numpy(video).permute(0, 3, 1, 2)
                t = torch.nn.functional.interpolate(t, size=(custom_frame_size, custom_frame_size), mode="bilinear",
                                                    align_corners=False, antialias=False)
                video = t.permute(0, 2, 3, 1).numpy()
            yield np.ascontiguousarray(video[:time_steps].transpose(1, 0, 2, 3))


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
