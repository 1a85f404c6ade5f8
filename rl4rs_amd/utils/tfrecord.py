"""TFRecord files of ``tf.train.Example`` records without TensorFlow.

The reference stores its supervised training set as TFRecords (``FeatureUtil.to_tfrecord`` / ``read_tfrecord``,
rl4rs/utils/datautil.py:71-230): one ``Example`` per sample with the features ``dense_feature`` (FloatList),
``category_feature``, ``slate_label``, ``label`` and ``sequence_id_<i>`` (Int64List).  This module reads and writes that
format from its public definitions:

* TFRecord framing: ``uint64 length | uint32 masked_crc32c(length) | bytes data | uint32 masked_crc32c(data)``, little
  endian, ``masked(c) = ((c >> 15 | c << 17) + 0xa282ead8) mod 2^32`` (CRC-32C, Castagnoli polynomial).
* protobuf wire format of ``Example { Features features = 1 }``, ``Features { map<string, Feature> feature = 1 }``,
  ``Feature { oneof { BytesList bytes_list = 1; FloatList float_list = 2; Int64List int64_list = 3 } }``,
  ``FloatList { repeated float value = 1 [packed] }``, ``Int64List { repeated int64 value = 1 [packed] }``
  (packed and unpacked repeated fields are both accepted when reading).

Host-side format conversion only (no model arithmetic).
"""
import struct

import numpy as np

_MASK_DELTA = 0xa282ead8


def _make_table():
    poly = 0x82F63B78                       # reversed Castagnoli polynomial
    tab = np.zeros(256, dtype=np.uint32)
    for i in range(256):
        c = i
        for _ in range(8):
            c = (c >> 1) ^ poly if c & 1 else c >> 1
        tab[i] = c
    return [int(x) for x in tab]


_TABLE = _make_table()


def _crc32c_py(data):
    c = 0xFFFFFFFF
    tab = _TABLE
    for b in data:
        c = tab[(c ^ b) & 0xFF] ^ (c >> 8)
    return c ^ 0xFFFFFFFF


def crc32c(data):
    """CRC-32C of a bytes-like; records go through the native ``rl4rs_crc32c`` (a training set is ~GBs of records)."""
    if len(data) < 64:
        return _crc32c_py(data)
    import ctypes
    from .. import _lib
    data = bytes(data)
    return int(_lib.load().rl4rs_crc32c(ctypes.cast(ctypes.c_char_p(data), ctypes.c_void_p), len(data), 0))


def masked_crc(data):
    c = crc32c(data)
    return (((c >> 15) | (c << 17)) + _MASK_DELTA) & 0xFFFFFFFF


# ------------------------------------------------------------------------------------------------ protobuf wire format
def _varint(n):
    n &= (1 << 64) - 1                      # int64 as two's complement
    out = bytearray()
    while True:
        b = n & 0x7F
        n >>= 7
        if n:
            out.append(b | 0x80)
        else:
            out.append(b)
            return bytes(out)


def _read_varint(buf, pos):
    shift = result = 0
    while True:
        b = buf[pos]
        pos += 1
        result |= (b & 0x7F) << shift
        if not b & 0x80:
            return result, pos
        shift += 7


def _ld(field, payload):                    # length-delimited field
    return _varint((field << 3) | 2) + _varint(len(payload)) + payload


def encode_example(features):
    """features: dict name -> 1-D numpy array / list; float dtypes become a FloatList, integer dtypes an Int64List.
    Entries are written in sorted key order (a map has no defined order on the wire)."""
    body = b''
    for name in sorted(features):
        v = np.asarray(features[name])
        if v.dtype.kind == 'f':
            payload = _ld(1, v.astype('<f4').tobytes())                     # FloatList.value, packed
            feat = _ld(2, payload)
        else:
            payload = _ld(1, b''.join(_varint(int(x)) for x in v.reshape(-1)))   # Int64List.value, packed
            feat = _ld(3, payload)
        entry = _ld(1, name.encode()) + _ld(2, feat)
        body += _ld(1, entry)
    return _ld(1, body)


def _fields(buf):
    """Yield (field number, wire type, value) of one message; value = int (varint / fixed) or bytes."""
    pos, n = 0, len(buf)
    while pos < n:
        key, pos = _read_varint(buf, pos)
        field, wt = key >> 3, key & 7
        if wt == 0:
            v, pos = _read_varint(buf, pos)
        elif wt == 2:
            ln, pos = _read_varint(buf, pos)
            v = buf[pos:pos + ln]
            pos += ln
        elif wt == 5:
            v = buf[pos:pos + 4]
            pos += 4
        elif wt == 1:
            v = buf[pos:pos + 8]
            pos += 8
        else:
            raise ValueError('unsupported protobuf wire type %d' % wt)
        yield field, wt, v


def _decode_feature(buf):
    for field, wt, v in _fields(buf):
        if field == 2:                       # FloatList
            vals = []
            for f2, w2, x in _fields(v):
                if f2 == 1:
                    vals.append(np.frombuffer(x, dtype='<f4'))       # packed run or one fixed32
            return np.concatenate(vals).astype(np.float32) if vals else np.zeros(0, np.float32)
        if field == 3:                       # Int64List
            vals = []
            for f2, w2, x in _fields(v):
                if f2 != 1:
                    continue
                if w2 == 0:
                    vals.append(x)
                else:
                    p = 0
                    while p < len(x):
                        y, p = _read_varint(x, p)
                        vals.append(y)
            a = np.array(vals, dtype=np.uint64).astype(np.int64)     # two's complement
            return a
        if field == 1:                       # BytesList
            return [x for f2, w2, x in _fields(v) if f2 == 1]
    return np.zeros(0, np.int64)


def decode_example(buf):
    out = {}
    for field, wt, features in _fields(buf):
        if field != 1:
            continue
        for f1, w1, entry in _fields(features):
            if f1 != 1:
                continue
            key = val = None
            for f2, w2, x in _fields(entry):
                if f2 == 1:
                    key = bytes(x).decode()
                elif f2 == 2:
                    val = _decode_feature(x)
            out[key] = val
    return out


# ------------------------------------------------------------------------------------------------------------ files
class TFRecordWriter(object):
    def __init__(self, path):
        self.f = open(path, 'wb')

    def write(self, data):
        head = struct.pack('<Q', len(data))
        self.f.write(head + struct.pack('<I', masked_crc(head)) + data + struct.pack('<I', masked_crc(data)))

    def close(self):
        self.f.close()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()


def read_records(path, verify=False):
    """Yield the raw record payloads of a TFRecord file; verify=True also checks both CRCs."""
    with open(path, 'rb') as f:
        while True:
            head = f.read(8)
            if not head:
                return
            if len(head) != 8:
                raise ValueError('%s: truncated record header' % path)
            (n,) = struct.unpack('<Q', head)
            (c1,) = struct.unpack('<I', f.read(4))
            data = f.read(n)
            tail = f.read(4)
            if len(data) != n or len(tail) != 4:
                raise ValueError('%s: truncated record' % path)
            if verify:
                if masked_crc(head) != c1:
                    raise ValueError('%s: corrupt record length' % path)
                if masked_crc(data) != struct.unpack('<I', tail)[0]:
                    raise ValueError('%s: corrupt record data' % path)
            yield data
