"""TFRecord files and ``tf.train.Example`` / ``tf.train.SequenceExample`` messages read WITHOUT TensorFlow
(the BAIR robot-push reader ``data_utils.py:63-104`` iterates
``tf.compat.v1.io.tf_record_iterator`` + ``tf.train.SequenceExample.FromString``, the GQN reader ``:355-449``
``tf.data.TFRecordDataset`` + ``tf.io.parse_example``).  Both formats are public and small:

* a TFRecord file is a sequence of ``[u64 length][u32 masked crc32c(length)][payload][u32 masked crc32c(payload)]``
  (little endian; mask = ``((crc >> 15 | crc << 17) + 0xa282ead8) mod 2^32``);
* ``Example { Features features = 1 }``, ``SequenceExample { Features context = 1; FeatureLists feature_lists = 2 }``,
  ``Features { map<string, Feature> feature = 1 }``, ``FeatureLists { map<string, FeatureList> feature_list = 1 }``,
  ``FeatureList { repeated Feature feature = 1 }``,
  ``Feature { oneof { BytesList bytes_list = 1; FloatList float_list = 2; Int64List int64_list = 3 } }``, the three lists
  ``{ repeated <bytes | float | int64> value = 1 }`` (floats and ints usually packed) -- protobuf wire format: a tag
  varint ``(field << 3) | type``, type 0 = varint, 1 = 8 bytes, 2 = length-delimited, 5 = 4 bytes.

Host-side data plumbing only (numpy); nothing here touches the GPU path.
"""
import struct

import numpy as np

__all__ = ["crc32c", "masked_crc32c", "records", "parse_example", "parse_sequence_example"]

_POLY = 0x82F63B78          # CRC-32C (Castagnoli), reflected
_TABLE = []
for _i in range(256):
    _c = _i
    for _ in range(8):
        _c = (_c >> 1) ^ _POLY if _c & 1 else _c >> 1
    _TABLE.append(_c)


def crc32c(data):
    """CRC-32C of ``data`` (bytes).  Byte-at-a-time table walk: ~1 MB/s in pure Python, which is why ``records`` only
    checks the 8-byte length field by default."""
    c = 0xFFFFFFFF
    tab = _TABLE
    for b in data:
        c = tab[(c ^ b) & 0xFF] ^ (c >> 8)
    return c ^ 0xFFFFFFFF


def masked_crc32c(data):
    c = crc32c(data)
    return (((c >> 15) | (c << 17)) + 0xA282EAD8) & 0xFFFFFFFF


def records(path, verify_payload=False):
    """Yield the payloads of a TFRecord file in order.  The length checksum is always verified (a wrong one means the
    file is not a TFRecord or is truncated mid-header); the payload checksum only on request."""
    with open(path, "rb") as f:
        while True:
            head = f.read(12)
            if not head:
                return
            if len(head) < 12:
                raise ValueError("%s: truncated record header" % path)
            (n,), (crc,) = struct.unpack("<Q", head[:8]), struct.unpack("<I", head[8:])
            if masked_crc32c(head[:8]) != crc:
                raise ValueError("%s: length checksum mismatch (not a TFRecord file?)" % path)
            body = f.read(n + 4)
            if len(body) < n + 4:
                raise ValueError("%s: truncated record (%d of %d bytes)" % (path, len(body), n + 4))
            if verify_payload and masked_crc32c(body[:n]) != struct.unpack("<I", body[n:])[0]:
                raise ValueError("%s: payload checksum mismatch" % path)
            yield body[:n]


# ---- protobuf wire format ----------------------------------------------------------------------------------------------
def _varint(buf, pos):
    out = shift = 0
    while True:
        b = buf[pos]
        pos += 1
        out |= (b & 0x7F) << shift
        if not b & 0x80:
            return out, pos
        shift += 7
        if shift > 63:
            raise ValueError("varint longer than 10 bytes")


def _fields(buf):
    """(field number, wire type, value) of every field of one message; value = int (types 0, 1, 5 as raw little-endian
    integers) or a memoryview slice (type 2)."""
    buf = memoryview(buf)
    pos, end = 0, len(buf)
    while pos < end:
        tag, pos = _varint(buf, pos)
        field, wt = tag >> 3, tag & 7
        if wt == 0:
            val, pos = _varint(buf, pos)
        elif wt == 1:
            val, pos = int.from_bytes(buf[pos:pos + 8], "little"), pos + 8
        elif wt == 2:
            n, pos = _varint(buf, pos)
            val, pos = buf[pos:pos + n], pos + n
        elif wt == 5:
            val, pos = int.from_bytes(buf[pos:pos + 4], "little"), pos + 4
        else:
            raise ValueError("unsupported protobuf wire type %d" % wt)
        if pos > end:
            raise ValueError("field runs past the end of its message")
        yield field, wt, val


def _feature(buf):
    """One ``Feature`` -> list of bytes | float32 array | int64 array (an empty Feature -> empty list)."""
    for field, wt, val in _fields(buf):
        if wt != 2:
            raise ValueError("Feature: unexpected wire type %d" % wt)
        if field == 1:                                   # BytesList
            return [bytes(v) for f, w, v in _fields(val) if f == 1]
        if field == 2:                                   # FloatList: packed (type 2) and / or single values (type 5)
            parts = []
            for f, w, v in _fields(val):
                if f != 1:
                    continue
                parts.append(np.frombuffer(v, dtype="<f4") if w == 2 else
                             np.frombuffer(struct.pack("<I", v), dtype="<f4"))
            return np.concatenate(parts).astype(np.float32) if parts else np.zeros(0, np.float32)
        if field == 3:                                   # Int64List: packed varints and / or single varints
            out = []
            for f, w, v in _fields(val):
                if f != 1:
                    continue
                if w == 2:
                    p = 0
                    while p < len(v):
                        x, p = _varint(v, p)
                        out.append(x)
                else:
                    out.append(v)
            a = np.array(out, dtype=np.uint64)
            return a.astype(np.int64)                    # two's complement: negative values are 10-byte varints
    return []


def _features(buf):
    """``Features`` -> {name: feature value}."""
    out = {}
    for field, wt, entry in _fields(buf):
        if field != 1 or wt != 2:
            continue
        key, value = None, []
        for f, w, v in _fields(entry):                   # map entry: key = 1, value = 2
            if f == 1:
                key = bytes(v).decode("utf-8")
            elif f == 2:
                value = _feature(v)
        if key is not None:
            out[key] = value
    return out


def parse_example(payload):
    """``tf.train.Example`` -> {name: list of bytes | float32 array | int64 array}."""
    out = {}
    for field, wt, val in _fields(payload):
        if field == 1 and wt == 2:
            out.update(_features(val))
    return out


def parse_sequence_example(payload):
    """``tf.train.SequenceExample`` -> (context {name: value}, feature_lists {name: [value per step]})."""
    context, lists = {}, {}
    for field, wt, val in _fields(payload):
        if wt != 2:
            continue
        if field == 1:
            context.update(_features(val))
        elif field == 2:
            for f, w, entry in _fields(val):
                if f != 1 or w != 2:
                    continue
                key, steps = None, []
                for f2, w2, v2 in _fields(entry):
                    if f2 == 1:
                        key = bytes(v2).decode("utf-8")
                    elif f2 == 2:
                        steps = [_feature(v3) for f3, w3, v3 in _fields(v2) if f3 == 1]
                if key is not None:
                    lists[key] = steps
    return context, lists
