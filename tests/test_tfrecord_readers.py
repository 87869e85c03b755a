"""the BAIR robot-push reader (data_utils.py:63-104) and the GQN reader (:355-449) without
TensorFlow.  The files under test are written by an INDEPENDENT encoder: the official ``google.protobuf`` runtime over
message types declared here from the public ``tf.train`` schema (example.proto / feature.proto), wrapped in the TFRecord
framing with ``zlib``-free CRC-32C from its published test vector -- so the hand-written wire parser of
``kccotgan_amd/tfrecord.py`` is checked against the library everyone else uses, not against itself."""
import io
import os
import struct

import numpy as np
import pytest

from kccotgan_amd import datasets, tfrecord


def _tf_train_messages():
    from google.protobuf import descriptor_pb2, descriptor_pool, message_factory
    T = descriptor_pb2.FieldDescriptorProto
    fd = descriptor_pb2.FileDescriptorProto(name="kccot_test_example.proto", package="kccot_test", syntax="proto3")

    def msg(name, fields, nested=()):
        m = fd.message_type.add(name=name)
        for fname, num, ftype, label, tname, oneof in fields:
            f = m.field.add(name=fname, number=num, type=ftype, label=label)
            if tname:
                f.type_name = tname
            if oneof is not None:
                f.oneof_index = oneof
        return m

    OPT, REP = T.LABEL_OPTIONAL, T.LABEL_REPEATED
    msg("BytesList", [("value", 1, T.TYPE_BYTES, REP, None, None)])
    msg("FloatList", [("value", 1, T.TYPE_FLOAT, REP, None, None)])
    msg("Int64List", [("value", 1, T.TYPE_INT64, REP, None, None)])
    feat = msg("Feature", [("bytes_list", 1, T.TYPE_MESSAGE, OPT, ".kccot_test.BytesList", 0),
                           ("float_list", 2, T.TYPE_MESSAGE, OPT, ".kccot_test.FloatList", 0),
                           ("int64_list", 3, T.TYPE_MESSAGE, OPT, ".kccot_test.Int64List", 0)])
    feat.oneof_decl.add(name="kind")
    # a proto map<string, V> is a repeated nested message {key = 1; value = 2} with the map_entry option
    for owner, vtype in (("Features", ".kccot_test.Feature"), ("FeatureLists", ".kccot_test.FeatureList")):
        if owner == "FeatureLists":
            msg("FeatureList", [("feature", 1, T.TYPE_MESSAGE, REP, ".kccot_test.Feature", None)])
        field = "feature" if owner == "Features" else "feature_list"
        entry = field.title().replace("_", "") + "Entry"
        m = msg(owner, [(field, 1, T.TYPE_MESSAGE, REP, ".kccot_test.%s.%s" % (owner, entry), None)])
        e = m.nested_type.add(name=entry)
        e.field.add(name="key", number=1, type=T.TYPE_STRING, label=OPT)
        e.field.add(name="value", number=2, type=T.TYPE_MESSAGE, label=OPT, type_name=vtype)
        e.options.map_entry = True
    msg("Example", [("features", 1, T.TYPE_MESSAGE, OPT, ".kccot_test.Features", None)])
    msg("SequenceExample", [("context", 1, T.TYPE_MESSAGE, OPT, ".kccot_test.Features", None),
                            ("feature_lists", 2, T.TYPE_MESSAGE, OPT, ".kccot_test.FeatureLists", None)])
    pool = descriptor_pool.DescriptorPool()
    pool.Add(fd)
    get = getattr(message_factory, "GetMessageClass", None)
    if get is None:
        fac = message_factory.MessageFactory(pool)
        get = fac.GetPrototype
    return {n: get(pool.FindMessageTypeByName("kccot_test." + n)) for n in ("Example", "SequenceExample")}


def _write_tfrecord(path, payloads):
    with open(path, "wb") as f:
        for p in payloads:
            head = struct.pack("<Q", len(p))
            f.write(head + struct.pack("<I", tfrecord.masked_crc32c(head)) + p + struct.pack("<I", tfrecord.masked_crc32c(p)))


def test_crc32c_known_answers():
    # RFC 3720 appendix B.4 test vectors
    assert tfrecord.crc32c(b"123456789") == 0xE3069283
    assert tfrecord.crc32c(bytes(32)) == 0x8A9136AA
    assert tfrecord.crc32c(bytes([0xFF] * 32)) == 0x62A8AB43
    assert tfrecord.crc32c(bytes(range(32))) == 0x46DD794E


def test_wire_parser_reads_what_the_protobuf_runtime_writes(tmp_path):
    M = _tf_train_messages()
    ex = M["Example"]()
    ex.features.feature["frames"].bytes_list.value.extend([b"abc", b"", b"\x00\xff" * 100])
    ex.features.feature["pos"].float_list.value.extend([1.5, -2.25, 3e-8])
    ex.features.feature["ids"].int64_list.value.extend([0, 1, -1, 2 ** 40, -2 ** 62])
    ex.features.feature["empty"].SetInParent()
    got = tfrecord.parse_example(ex.SerializeToString())
    assert got["frames"] == [b"abc", b"", b"\x00\xff" * 100]
    np.testing.assert_array_equal(got["pos"], np.array([1.5, -2.25, 3e-8], np.float32))
    np.testing.assert_array_equal(got["ids"], np.array([0, 1, -1, 2 ** 40, -2 ** 62], np.int64))
    assert got["empty"] == []

    se = M["SequenceExample"]()
    se.context.feature["0/action"].float_list.value.extend([0.25, 0.5])
    fl = se.feature_lists.feature_list["steps"]
    for i in range(3):
        fl.feature.add().int64_list.value.append(i * 7)
    ctx, lists = tfrecord.parse_sequence_example(se.SerializeToString())
    np.testing.assert_array_equal(ctx["0/action"], np.array([0.25, 0.5], np.float32))
    assert [int(s[0]) for s in lists["steps"]] == [0, 7, 14]

    path = tmp_path / "two.tfrecord"
    _write_tfrecord(path, [ex.SerializeToString(), se.SerializeToString()])
    recs = list(tfrecord.records(str(path), verify_payload=True))
    assert recs == [ex.SerializeToString(), se.SerializeToString()]
    raw = bytearray(path.read_bytes())
    raw[20] ^= 1                                        # a payload byte
    path.write_bytes(bytes(raw))
    assert len(list(tfrecord.records(str(path)))) == 2  # the default checks the length field only
    with pytest.raises(ValueError, match="payload checksum"):
        list(tfrecord.records(str(path), verify_payload=True))
    raw[3] ^= 1                                         # a length byte
    path.write_bytes(bytes(raw))
    with pytest.raises(ValueError, match="length checksum"):
        list(tfrecord.records(str(path)))


def test_robot_push_reader_layout(tmp_path):
    """data_utils.py:63-104: 30 frames of raw 64 x 64 x 3 uint8 per record under '<i>/image_aux1/encoded' in the CONTEXT,
    stacked, transposed to [H, T, W, C], / 255, first T frames."""
    M = _tf_train_messages()
    rng = np.random.default_rng(0)
    vids = rng.integers(0, 256, size=(3, 30, 64, 64, 3), dtype=np.uint8)
    payloads = []
    for v in vids:
        se = M["SequenceExample"]()
        for i in range(30):
            se.context.feature["%d/image_aux1/encoded" % i].bytes_list.value.append(v[i].tobytes())
            se.context.feature["%d/image_main/encoded" % i].bytes_list.value.append(bytes(64 * 64 * 3))
            se.context.feature["%d/action" % i].float_list.value.extend([0.1, 0.2, 0.3, 0.4])
        payloads.append(se.SerializeToString())
    _write_tfrecord(tmp_path / "a.tfrecord", payloads[:2])
    _write_tfrecord(tmp_path / "b.tfrecord", payloads[2:])
    out = list(datasets.robot_push_videos([str(tmp_path / "a.tfrecord"), str(tmp_path / "b.tfrecord")], T=12))
    assert len(out) == 3
    for v, o in zip(vids, out):
        assert o.shape == (64, 12, 64, 3) and o.dtype == np.float64
        np.testing.assert_array_equal(o, (v.transpose(1, 0, 2, 3) / 255.0)[:, :12])
    x = next(datasets.batches(out + out, 2, 64, 12, 64, 3))
    assert tuple(x.shape) == (2, 64, 12, 64, 3)


def test_gqn_reader_layout_and_file_names(tmp_path):
    """data_utils.py:355-449: 'frames' = sequence_size JPEG strings per record -> float32 [0, 1] -> [H, T, W, C]; file
    names '<i>-of-<n>.tfrecord' zero-padded to the width of n (:336-348)."""
    from PIL import Image
    M = _tf_train_messages()
    names = datasets.gqn_files("rooms_ring_camera", "/data", "test")
    assert names[0] == "/data/rooms_ring_camera/test/001-of-240.tfrecord" and names[-1].endswith("240-of-240.tfrecord")
    assert datasets.gqn_files("mazes", "/d")[9] == "/d/mazes/train/0010-of-1080.tfrecord"
    with pytest.raises(ValueError):
        datasets.gqn_files("nope", "/d")
    info = datasets.GQN_DATASETS["shepard_metzler_5_parts"]              # 15 views of 64 x 64
    yy, xx = np.mgrid[0:64, 0:64]
    payloads, frames_ref = [], []
    for r in range(2):
        ex = M["Example"]()
        ref = []
        for t in range(info.sequence_size):
            img = np.stack([(xx * 3 + t * 5 + r) % 256, (yy * 2 + t) % 256, ((xx + yy) * 2) % 256], -1).astype(np.uint8)
            buf = io.BytesIO()
            Image.fromarray(img).save(buf, format="JPEG", quality=95)
            ex.features.feature["frames"].bytes_list.value.append(buf.getvalue())
            ref.append(np.asarray(Image.open(io.BytesIO(buf.getvalue())).convert("RGB")))
        ex.features.feature["cameras"].float_list.value.extend(np.arange(info.sequence_size * 5, dtype=np.float32))
        payloads.append(ex.SerializeToString())
        frames_ref.append(np.stack(ref))
    path = tmp_path / "01-of-10.tfrecord"
    _write_tfrecord(path, payloads)
    out = list(datasets.gqn_videos([str(path)], "shepard_metzler_5_parts", time_steps=6))
    assert len(out) == 2
    for ref, o in zip(frames_ref, out):
        assert o.shape == (64, 6, 64, 3) and o.dtype == np.float32
        np.testing.assert_allclose(o, (ref[:6].astype(np.float32) / 255.0).transpose(1, 0, 2, 3), rtol=0, atol=1e-7)
    small = next(datasets.gqn_videos([str(path)], "shepard_metzler_5_parts", time_steps=4, custom_frame_size=32))
    assert small.shape == (32, 4, 32, 3)
    # bilinear, half-pixel centres, 2:1: every output pixel is the mean of a 2 x 2 block
    blk = (frames_ref[0][:4].astype(np.float32) / 255.0).reshape(4, 32, 2, 32, 2, 3).mean((2, 4))
    np.testing.assert_allclose(small, blk.transpose(1, 0, 2, 3), rtol=0, atol=1e-6)
    with pytest.raises(ValueError, match="Maximum support context size"):
        next(datasets.gqn_videos([str(path)], "shepard_metzler_5_parts", time_steps=16))


# ---- fixtures from an INDEPENDENT writer (tests/golden/make_tfrecord_golden.py: bytes assembled by hand, bit-at-a-time CRC-32C
# checked against RFC 3720's vectors; no protobuf runtime, none of this package's code) ------------------------------------
GOLD_TF = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "tfrecord")


def test_readers_on_hand_assembled_bair_records():
    """data_utils.py:63-104 on records in the layout of the real BAIR files: tf.train.Example messages (the reference parses
    them as SequenceExample and reads .context -- field 1 either way) with '<i>/image_aux1/encoded' raw uint8 frames among
    other features, map entries in arbitrary order and in both key/value orders, packed and unpacked floats, a negative
    int64, an empty Feature; both checksums of every record verified."""
    exp = np.load(os.path.join(GOLD_TF, "tfrecord_expected.npz"))
    path = os.path.join(GOLD_TF, "bair_softmotion.tfrecord")
    recs = list(tfrecord.records(path, verify_payload=True))
    assert [len(r) for r in recs] == exp["bair_payload_bytes"].tolist()
    ctx, lists = tfrecord.parse_sequence_example(recs[0])
    assert lists == {} and len(ctx) == 4 * 30 + 2
    assert ctx["meta/ids"].tolist() == [0, -1, 1 << 40] and ctx["meta/empty"] == []
    assert ctx["3/endeffector_pos"].shape == (3,) and ctx["3/action"].shape == (4,)          # unpacked / packed floats
    assert tfrecord.parse_example(recs[1])["meta/ids"].tolist() == [1, -1, 1 << 40]          # unpacked varints
    out = list(datasets.robot_push_videos([path], T=20, img_shape=(16, 16, 3)))
    assert len(out) == 2
    for frames, o in zip(exp["bair_frames"], out):
        assert o.shape == (16, 20, 16, 3) and o.dtype == np.float64
        np.testing.assert_array_equal(o, (frames.transpose(1, 0, 2, 3) / 255.0)[:, :20])
    with pytest.raises(ValueError, match="expected"):                                      # the default 64 x 64 x 3 does not fit
        next(datasets.robot_push_videos([path]))


def test_readers_on_hand_assembled_gqn_records():
    """data_utils.py:355-449 on records in the layout of the GQN shepard_metzler_5_parts files ('frames': 15 JPEG strings,
    'cameras': 75 floats, the map entry of 'cameras' written value first): float32 in [0, 1], [H, T, W, C], first T views,
    the file name the reference's template produces; bilinear resize to a custom frame size."""
    exp = np.load(os.path.join(GOLD_TF, "tfrecord_expected.npz"))
    names = datasets.gqn_files("shepard_metzler_5_parts", GOLD_TF, "train")
    assert os.path.basename(names[0]) == "001-of-900.tfrecord" and len(names) == 900      # data_utils.py:343-346
    path = os.path.join(GOLD_TF, "001-of-900.tfrecord")
    recs = list(tfrecord.records(path, verify_payload=True))
    assert [len(r) for r in recs] == exp["gqn_payload_bytes"].tolist()
    ex = tfrecord.parse_example(recs[0])
    assert len(ex["frames"]) == 15 and ex["cameras"].shape == (75,) and ex["frames"][0][:2] == b"\xff\xd8"   # JPEG SOI
    out = list(datasets.gqn_videos([path], "shepard_metzler_5_parts", 10))
    assert len(out) == 2
    for pix, o in zip(exp["gqn_pixels"], out):
        assert o.shape == (64, 10, 64, 3) and o.dtype == np.float32
        np.testing.assert_array_equal(o, (pix[:10].astype(np.float32) * np.float32(1 / 255.0)).transpose(1, 0, 2, 3))
    small = next(datasets.gqn_videos([path], "shepard_metzler_5_parts", 4, custom_frame_size=32))
    blk = (exp["gqn_pixels"][0][:4].astype(np.float32) / 255.0).reshape(4, 32, 2, 32, 2, 3).mean((2, 4))    # 2x2 box = bilinear at 1/2
    np.testing.assert_allclose(small, blk.transpose(1, 0, 2, 3), atol=1e-6)
