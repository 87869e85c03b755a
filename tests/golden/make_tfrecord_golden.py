#!/usr/bin/env python3
"""Writes the two TFRecord fixtures of tests/golden/tfrecord/ BYTE BY BYTE -- an independent writer for
tests/test_tfrecord_readers.py: nothing here imports kccotgan_amd, google.protobuf or TensorFlow.

What is assembled by hand, from the public format descriptions only:
  * the TFRecord framing  [u64 length][u32 masked crc32c(length)][payload][u32 masked crc32c(payload)]  with a BIT-AT-A-TIME
    CRC-32C (Castagnoli polynomial 0x1EDC6F41, reflected) that is first checked against the known answers of RFC 3720
    section B.4 and the "123456789" check value 0xE3069283 -- the reader under test uses a table-driven variant;
  * the protobuf wire format of tf.train.Example / Features / Feature / BytesList / FloatList / Int64List: tag varints
    (field << 3 | type), length-delimited submessages, packed AND unpacked repeated scalars, map entries written in BOTH field
    orders (key first / value first: both are legal on the wire), a negative int64 as a ten-byte varint, an empty Feature.

Fixtures (layouts: data_utils.py:63-104 and :355-449):
  bair_softmotion.tfrecord   two records in the layout of the BAIR robot-push files (softmotion30_44k): a tf.train.Example
                             whose features '<i>/image_aux1/encoded' and '<i>/image_main/encoded' hold RAW uint8 frames
                             (16 x 16 x 3 here instead of 64 x 64 x 3: the reader takes the frame shape as a parameter),
                             '<i>/endeffector_pos' 3 floats, '<i>/action' 4 floats, i = 0..29.  The reference parses these
                             records as SequenceExample and reads .context -- Example.features and SequenceExample.context
                             are both field 1, so the same bytes serve.
  001-of-900.tfrecord       two records in the layout of the GQN shepard_metzler_5_parts files: 'frames' = 15 JPEG strings
                             of 64 x 64 x 3 views, 'cameras' = 15 x 5 floats.  (JPEG ENcoding is PIL's here; the expected
                             pixels stored beside the file are PIL's decode of those very strings, so what is pinned is
                             the container and the layout, not a decoder.)
  tfrecord_expected.npz      the arrays a correct reader must produce.

usage: python tests/golden/make_tfrecord_golden.py      (rewrites tests/golden/tfrecord/)"""
import io
import os
import struct

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, "tfrecord")


# ---- CRC-32C, one bit at a time ----------------------------------------------------------------------------------------
def crc32c_bitwise(data):
    crc = 0xFFFFFFFF
    for byte in data:
        crc ^= byte
        for _ in range(8):
            crc = (crc >> 1) ^ (0x82F63B78 if crc & 1 else 0)     # 0x1EDC6F41 bit-reversed
    return crc ^ 0xFFFFFFFF


def check_crc():
    assert crc32c_bitwise(b"123456789") == 0xE3069283                         # the CRC catalogue's check value
    assert crc32c_bitwise(bytes(32)) == 0x8A9136AA                            # RFC 3720 B.4: 32 bytes of zeros
    assert crc32c_bitwise(b"\xff" * 32) == 0x62A8AB43                         #               32 bytes of ones
    assert crc32c_bitwise(bytes(range(32))) == 0x46DD794E                     #               incrementing
    assert crc32c_bitwise(bytes(range(31, -1, -1))) == 0x113FDB5C             #               decrementing


def masked(data):
    c = crc32c_bitwise(data)
    return (((c >> 15) | (c << 17)) + 0xA282EAD8) & 0xFFFFFFFF            # TFRecord: rotate right by 15, add the constant


def frame_record(payload):
    head = struct.pack("<Q", len(payload))
    return head + struct.pack("<I", masked(head)) + payload + struct.pack("<I", masked(payload))


# ---- protobuf wire format ----------------------------------------------------------------------------------------------
def varint(n):
    n &= (1 << 64) - 1                     # negative int64 -> two's complement, ten bytes
    out = bytearray()
    while True:
        b = n & 0x7F
        n >>= 7
        out.append(b | (0x80 if n else 0))
        if not n:
            return bytes(out)


def ld(field, payload):                    # length-delimited field (wire type 2)
    return varint((field << 3) | 2) + varint(len(payload)) + payload


def bytes_feature(values):                 # Feature { bytes_list = 1 { repeated bytes value = 1 } }
    return ld(1, b"".join(ld(1, v) for v in values))


def float_feature(values, packed=True):    # Feature { float_list = 2 { repeated float value = 1 } }
    if packed:
        body = ld(1, b"".join(struct.pack("<f", v) for v in values))
    else:                                  # one fixed32 field (wire type 5) per value
        body = b"".join(varint((1 << 3) | 5) + struct.pack("<f", v) for v in values)
    return ld(2, body)


def int64_feature(values, packed=True):    # Feature { int64_list = 3 { repeated int64 value = 1 } }
    if packed:
        body = ld(1, b"".join(varint(v) for v in values))
    else:
        body = b"".join(varint((1 << 3) | 0) + varint(v) for v in values)
    return ld(3, body)


def features(entries):
    """Features { map<string, Feature> feature = 1 }: every map entry is a submessage { key = 1, value = 2 }; entries
    flagged `swap` are written value first."""
    out = b""
    for key, feat, swap in entries:
        k, v = ld(1, key.encode("utf-8")), ld(2, feat)
        out += ld(1, v + k if swap else k + v)
    return out


def example(entries):                      # Example { Features features = 1 }
    return ld(1, features(entries))


# ---- the fixtures ------------------------------------------------------------------------------------------------------
def bair_records(rng, n=2, frames=30, shape=(16, 16, 3)):
    payloads, aux = [], []
    for r in range(n):
        entries, vid = [], []
        for i in range(frames):
            a = rng.integers(0, 256, size=shape, dtype=np.uint8)
            m = rng.integers(0, 256, size=shape, dtype=np.uint8)
            vid.append(a)
            entries.append(("%d/image_main/encoded" % i, bytes_feature([m.tobytes()]), False))
            entries.append(("%d/image_aux1/encoded" % i, bytes_feature([a.tobytes()]), (i % 3 == 1)))
            entries.append(("%d/endeffector_pos" % i, float_feature(rng.random(3).tolist(), packed=(i % 2 == 0)), False))
            entries.append(("%d/action" % i, float_feature(rng.random(4).tolist()), (i % 5 == 0)))
        entries.append(("meta/ids", int64_feature([r, -1, 1 << 40], packed=(r == 0)), False))
        entries.append(("meta/empty", b"", False))                       # a Feature with no list set
        order = rng.permutation(len(entries))                            # map entries come in no particular order
        payloads.append(example([entries[j] for j in order]))
        aux.append(np.stack(vid))
    return payloads, np.stack(aux)                                       # [n, frames, H, W, C] uint8


def gqn_records(rng, n=2, views=15, size=64):
    from PIL import Image
    payloads, pix = [], []
    yy, xx = np.mgrid[0:size, 0:size]
    for r in range(n):
        jpegs, dec = [], []
        for v in range(views):
            img = np.zeros((size, size, 3), np.uint8)                    # smooth content: JPEG keeps it nearly exactly
            img[..., 0] = (xx * 3 + 10 * v) % 256
            img[..., 1] = (yy * 2 + 40 * r) % 256
            img[..., 2] = ((xx + yy) * 2) % 256
            buf = io.BytesIO()
            Image.fromarray(img).save(buf, format="JPEG", quality=90)
            jpegs.append(buf.getvalue())
            dec.append(np.asarray(Image.open(io.BytesIO(jpegs[-1])).convert("RGB")))
        cams = rng.random(views * 5).tolist()
        payloads.append(example([("cameras", float_feature(cams), True), ("frames", bytes_feature(jpegs), False)]))
        pix.append(np.stack(dec))
    return payloads, np.stack(pix)                                       # [n, views, H, W, 3] uint8 (decoded)


def main():
    check_crc()
    os.makedirs(OUT, exist_ok=True)
    rng = np.random.default_rng(20261005)
    bair, bair_frames = bair_records(rng)
    with open(os.path.join(OUT, "bair_softmotion.tfrecord"), "wb") as f:
        for p in bair:
            f.write(frame_record(p))
    gqn, gqn_pixels = gqn_records(rng)
    with open(os.path.join(OUT, "001-of-900.tfrecord"), "wb") as f:
        for p in gqn:
            f.write(frame_record(p))
    np.savez_compressed(os.path.join(OUT, "tfrecord_expected.npz"), bair_frames=bair_frames, gqn_pixels=gqn_pixels,
                        bair_payload_bytes=np.array([len(p) for p in bair]), gqn_payload_bytes=np.array([len(p) for p in gqn]))
    for name in sorted(os.listdir(OUT)):
        print("%-28s %8d bytes" % (name, os.path.getsize(os.path.join(OUT, name))))


if __name__ == "__main__":
    main()
