"""TensorFlow checkpoints (``tf.train.Saver`` "V2" tensor bundles) without TensorFlow.

The reference keeps its simulator weights in TF1 checkpoints: ``supervised_train.py:44-46`` writes one with
``tf.train.Saver().save(sess, model_file)`` and every env restores it (``RecSimBase.__init__`` / ``reload_model``,
rl4rs/env/base.py:129,148-151).  ``config['model_file']`` is therefore a checkpoint *prefix*; this module reads (and
writes) the files behind it from their public format so that such a ``model_file`` loads here unchanged:

* ``<prefix>.index`` - an SSTable in the LevelDB table format (tensorflow/core/lib/io/table*): data blocks of
  prefix-compressed ``key -> value`` entries (``varint32 shared | varint32 non_shared | varint32 value_len | key suffix |
  value``, restart offsets ``uint32[n] | uint32 n`` at the end), each block followed by a 5-byte trailer
  (``uint8 compression`` 0 = none / 1 = snappy, ``uint32 masked crc32c(block + type)``); an index block mapping
  separator keys to ``BlockHandle(varint64 offset, varint64 size)``; a 48-byte footer
  (``metaindex handle | index handle | zero padding to 40 bytes | magic 0xdb4775248b80fb57`` little endian).
  Key ``""`` holds a ``BundleHeaderProto {num_shards=1, endianness=2, version=3}``, every other key is a variable name
  holding a ``BundleEntryProto {dtype=1, shape=2, shard_id=3, offset=4, size=5, crc32c=6 (fixed32, masked), slices=7}``
  (tensorflow/core/protobuf/tensor_bundle.proto).
* ``<prefix>.data-SSSSS-of-NNNNN`` - the raw little-endian tensor bytes, addressed by (shard_id, offset, size).

``variable_names`` lists the graph variable names the reference's four simulator families get from Keras / deepctr
auto-naming (creation order of rl4rs/nets/{dien,dnn,widedeep,lstm}.py and utils.py), mapped to this package's weight
names (``rl4rs_amd.nets.dien.dien_spec`` / ``nets.simnets.simnet_spec``).  There is no TensorFlow or deepctr in this
image and no checkpoint in the reference tree, so the FORMAT is tested against its published constants and by round
trip, and the NAME TABLE is unpinned: ``load_simulator_weights`` fails loudly, listing both sides, when a name is
missing, and takes an explicit ``name_map`` override.

Host-side format conversion only (no model arithmetic).
"""
import os
import re
import struct
from collections import OrderedDict

import numpy as np

from .tfrecord import _fields, _ld, _read_varint, _varint

TABLE_MAGIC = 0xdb4775248b80fb57
_MASK_DELTA = 0xa282ead8
_FOOTER_LEN = 48
_BLOCK_SIZE = 4096                  # table::Options::block_size
_RESTART_INTERVAL = 16              # table::Options::block_restart_interval

# tensorflow/core/framework/types.proto
_DTYPES = {1: '<f4', 2: '<f8', 3: '<i4', 4: 'u1', 5: '<i2', 6: 'i1', 9: '<i8', 10: '?', 17: '<u2', 19: '<f2',
           22: '<u4', 23: '<u8'}
_DTYPE_ENUM = dict((np.dtype(v).str, k) for k, v in _DTYPES.items())


def crc32c(data, crc=0):
    """CRC-32C through the native helper (``rl4rs_crc32c``; embedding tables are 51 MB each)."""
    from .. import _lib
    import ctypes
    if isinstance(data, np.ndarray):
        arr = np.ascontiguousarray(data)
        return int(_lib.load().rl4rs_crc32c(arr.ctypes.data_as(ctypes.c_void_p), arr.nbytes, crc))
    data = bytes(data)
    return int(_lib.load().rl4rs_crc32c(ctypes.cast(ctypes.c_char_p(data), ctypes.c_void_p), len(data), crc))


def mask_crc(c):
    return (((c >> 15) | (c << 17)) + _MASK_DELTA) & 0xFFFFFFFF


def unmask_crc(m):
    rot = (m - _MASK_DELTA) & 0xFFFFFFFF
    return ((rot >> 17) | (rot << 15)) & 0xFFFFFFFF


# ------------------------------------------------------------------------------------------------------------ snappy
def snappy_uncompress(buf):
    """Raw snappy block format (a table block may carry compression type 1; TF's bundle writer uses none)."""
    n, pos = _read_varint(buf, 0)
    out = bytearray()
    end = len(buf)
    while pos < end:
        tag = buf[pos]
        pos += 1
        kind = tag & 3
        if kind == 0:                                   # literal
            ln = tag >> 2
            if ln >= 60:
                nb = ln - 59
                ln = int.from_bytes(buf[pos:pos + nb], 'little')
                pos += nb
            ln += 1
            out += buf[pos:pos + ln]
            pos += ln
            continue
        if kind == 1:
            ln = ((tag >> 2) & 7) + 4
            off = ((tag >> 5) << 8) | buf[pos]
            pos += 1
        elif kind == 2:
            ln = (tag >> 2) + 1
            off = buf[pos] | (buf[pos + 1] << 8)
            pos += 2
        else:
            ln = (tag >> 2) + 1
            off = int.from_bytes(buf[pos:pos + 4], 'little')
            pos += 4
        if off == 0 or off > len(out):
            raise ValueError('corrupt snappy block (copy offset %d at output %d)' % (off, len(out)))
        start = len(out) - off
        for i in range(ln):                             # copies may overlap their own output
            out.append(out[start + i])
    if len(out) != n:
        raise ValueError('corrupt snappy block (%d bytes, header says %d)' % (len(out), n))
    return bytes(out)


# ------------------------------------------------------------------------------------------------------ table reading
def _read_block(buf, offset, size, verify):
    body = buf[offset:offset + size]
    trailer = buf[offset + size:offset + size + 5]
    if len(body) != size or len(trailer) != 5:
        raise ValueError('table block [%d, +%d) runs past the end of the file' % (offset, size))
    ctype = trailer[0]
    if verify:
        want = unmask_crc(struct.unpack('<I', trailer[1:5])[0])
        got = crc32c(bytes(body) + bytes(trailer[:1]))
        if want != got:
            raise ValueError('table block at %d: crc32c %08x, stored %08x' % (offset, got, want))
    if ctype == 1:
        body = snappy_uncompress(bytes(body))
    elif ctype != 0:
        raise ValueError('table block at %d: unknown compression type %d' % (offset, ctype))
    return bytes(body)


def _block_entries(block):
    """(key, value) pairs of one block, undoing the prefix compression."""
    if len(block) < 4:
        raise ValueError('table block shorter than its restart count')
    n_restarts = struct.unpack('<I', block[-4:])[0]
    limit = len(block) - 4 - 4 * n_restarts
    if limit < 0:
        raise ValueError('table block: %d restarts do not fit %d bytes' % (n_restarts, len(block)))
    pos, key = 0, b''
    while pos < limit:
        shared, pos = _read_varint(block, pos)
        non_shared, pos = _read_varint(block, pos)
        vlen, pos = _read_varint(block, pos)
        if shared > len(key) or pos + non_shared + vlen > limit:
            raise ValueError('corrupt table block entry at %d' % pos)
        key = key[:shared] + block[pos:pos + non_shared]
        pos += non_shared
        yield key, block[pos:pos + vlen]
        pos += vlen


def read_table(path, verify=True):
    """All (key, value) pairs of an SSTable file, in key order."""
    with open(path, 'rb') as f:
        buf = f.read()
    if len(buf) < _FOOTER_LEN:
        raise ValueError('%s: %d bytes is shorter than a table footer' % (path, len(buf)))
    footer = buf[-_FOOTER_LEN:]
    if struct.unpack('<Q', footer[40:48])[0] != TABLE_MAGIC:
        raise ValueError('%s: not a TensorFlow checkpoint index (bad table magic)' % path)
    pos = 0
    _, pos = _read_varint(footer, pos)                  # metaindex handle (unused by the bundle format)
    _, pos = _read_varint(footer, pos)
    idx_off, pos = _read_varint(footer, pos)
    idx_size, pos = _read_varint(footer, pos)
    out = []
    for _, handle in _block_entries(_read_block(buf, idx_off, idx_size, verify)):
        off, p = _read_varint(handle, 0)
        size, p = _read_varint(handle, p)
        out.extend(_block_entries(_read_block(buf, off, size, verify)))
    return out


# ------------------------------------------------------------------------------------------------------ table writing
class _BlockBuilder(object):
    def __init__(self, restart_interval):
        self.interval = restart_interval
        self.reset()

    def reset(self):
        self.buf = bytearray()
        self.restarts = [0]
        self.count = 0
        self.last = b''

    def add(self, key, value):
        shared = 0
        if self.count < self.interval:
            m = min(len(key), len(self.last))
            while shared < m and key[shared] == self.last[shared]:
                shared += 1
        else:
            self.restarts.append(len(self.buf))
            self.count = 0
        self.buf += _varint(shared) + _varint(len(key) - shared) + _varint(len(value)) + key[shared:] + value
        self.last = key
        self.count += 1

    def size(self):
        return len(self.buf) + 4 * len(self.restarts) + 4

    def empty(self):
        return not self.buf

    def finish(self):
        return bytes(self.buf) + b''.join(struct.pack('<I', r) for r in self.restarts) + struct.pack('<I', len(self.restarts))


def _handle(offset, size):
    return _varint(offset) + _varint(size)


def write_table(path, items):
    """items: iterable of (key bytes, value bytes) in strictly increasing key order."""
    out = bytearray()

    def emit(block):
        off = len(out)
        out.extend(block)
        out.append(0)                                   # kNoCompression
        out.extend(struct.pack('<I', mask_crc(crc32c(block + b'\x00'))))
        return off, len(block)

    data, index = _BlockBuilder(_RESTART_INTERVAL), _BlockBuilder(1)
    prev = None
    for key, value in items:
        if prev is not None and key <= prev:
            raise ValueError('table keys must be strictly increasing (%r after %r)' % (key, prev))
        data.add(key, value)
        prev = key
        if data.size() >= _BLOCK_SIZE:
            off, size = emit(data.finish())
            index.add(prev, _handle(off, size))         # separator = the block's last key (any key in [last, next) is valid)
            data.reset()
    if not data.empty():
        off, size = emit(data.finish())
        index.add(prev, _handle(off, size))
    meta_off, meta_size = emit(_BlockBuilder(_RESTART_INTERVAL).finish())
    idx_off, idx_size = emit(index.finish())
    footer = _handle(meta_off, meta_size) + _handle(idx_off, idx_size)
    footer += b'\x00' * (40 - len(footer)) + struct.pack('<Q', TABLE_MAGIC)
    out.extend(footer)
    with open(path, 'wb') as f:
        f.write(bytes(out))


# ------------------------------------------------------------------------------------------------------- bundle protos
def _encode_shape(shape):
    return b''.join(_ld(2, _varint((1 << 3) | 0) + _varint(int(d))) for d in shape)     # TensorShapeProto.dim.size


def _decode_shape(buf):
    dims = []
    for field, _, v in _fields(buf):
        if field == 2:
            size = 0
            for f2, _, v2 in _fields(v):
                if f2 == 1:
                    size = v2 - (1 << 64) if v2 >= (1 << 63) else v2
            dims.append(size)
        elif field == 3 and v:
            raise ValueError('checkpoint tensor of unknown rank')
    return tuple(dims)


def _encode_entry(dtype_enum, shape, shard, offset, size, crc):
    out = _varint((1 << 3) | 0) + _varint(dtype_enum)
    out += _ld(2, _encode_shape(shape))
    if shard:
        out += _varint((3 << 3) | 0) + _varint(shard)
    if offset:
        out += _varint((4 << 3) | 0) + _varint(offset)
    if size:
        out += _varint((5 << 3) | 0) + _varint(size)
    out += _varint((6 << 3) | 5) + struct.pack('<I', crc)
    return out


def _decode_entry(buf):
    e = dict(dtype=0, shape=(), shard_id=0, offset=0, size=0, crc32c=None, slices=False)
    for field, _, v in _fields(buf):
        if field == 1:
            e['dtype'] = v
        elif field == 2:
            e['shape'] = _decode_shape(v)
        elif field == 3:
            e['shard_id'] = v
        elif field == 4:
            e['offset'] = v
        elif field == 5:
            e['size'] = v
        elif field == 6:
            e['crc32c'] = struct.unpack('<I', v)[0]
        elif field == 7:
            e['slices'] = True
    return e


def _data_path(prefix, shard, num_shards):
    return '%s.data-%05d-of-%05d' % (prefix, shard, num_shards)


def resolve_prefix(model_file):
    """``model_file`` as the reference passes it (a Saver prefix); also accepts the ``.index`` path, or a directory
    holding a ``checkpoint`` state file (``model_checkpoint_path: "..."``, what ``tf.train.latest_checkpoint`` reads)."""
    p = str(model_file)
    if p.endswith('.index'):
        p = p[:-len('.index')]
    if os.path.isdir(p):
        state = os.path.join(p, 'checkpoint')
        if os.path.exists(state):
            with open(state) as f:
                m = re.search(r'^model_checkpoint_path:\s*"(.*)"', f.read(), re.M)
            if m:
                q = m.group(1)
                p = q if os.path.isabs(q) else os.path.join(p, q)
    return p


def is_checkpoint(model_file):
    return os.path.exists(resolve_prefix(model_file) + '.index')


def list_variables(model_file, verify=False):
    """[(name, shape, numpy dtype str)] of a checkpoint, like ``tf.train.list_variables``."""
    prefix = resolve_prefix(model_file)
    out = []
    for key, value in read_table(prefix + '.index', verify=verify):
        if key == b'':
            continue
        e = _decode_entry(value)
        out.append((key.decode(), e['shape'], _DTYPES.get(e['dtype'], 'enum %d' % e['dtype'])))
    return out


def read_checkpoint(model_file, names=None, verify=True):
    """name -> ndarray for the variables of a checkpoint (only ``names`` when given; dtypes this module does not
    know - strings, resources - are skipped unless explicitly asked for)."""
    prefix = resolve_prefix(model_file)
    entries = read_table(prefix + '.index', verify=verify)
    num_shards = 1
    table = OrderedDict()
    for key, value in entries:
        if key == b'':
            for field, _, v in _fields(value):
                if field == 1:
                    num_shards = v
                elif field == 2 and v != 0:
                    raise ValueError('%s: big-endian checkpoint' % prefix)
            continue
        table[key.decode()] = _decode_entry(value)
    want = list(table) if names is None else list(names)
    files, out = {}, OrderedDict()
    try:
        for name in want:
            if name not in table:
                raise KeyError('variable %r is not in checkpoint %s' % (name, prefix))
            e = table[name]
            if e['dtype'] not in _DTYPES or e['slices']:
                if names is None:
                    continue
                raise ValueError('variable %r: unsupported dtype enum %d / sliced tensor' % (name, e['dtype']))
            dt = np.dtype(_DTYPES[e['dtype']])
            count = int(np.prod(e['shape'], dtype=np.int64)) if e['shape'] else 1
            if count * dt.itemsize != e['size']:
                raise ValueError('variable %r: %d bytes for shape %r of %s' % (name, e['size'], e['shape'], dt))
            f = files.get(e['shard_id'])
            if f is None:
                f = files[e['shard_id']] = open(_data_path(prefix, e['shard_id'], num_shards), 'rb')
            f.seek(e['offset'])
            arr = np.fromfile(f, dtype=dt, count=count)
            if arr.size != count:
                raise ValueError('variable %r: data file truncated' % name)
            if verify and e['crc32c'] is not None and crc32c(arr) != unmask_crc(e['crc32c']):
                raise ValueError('variable %r: crc32c mismatch in %s' % (name, prefix))
            out[name] = arr.reshape(e['shape'])
    finally:
        for f in files.values():
            f.close()
    return out


def write_checkpoint(prefix, variables, write_state=True):
    """Write name -> ndarray as a single-shard V2 checkpoint (``<prefix>.index`` + ``.data-00000-of-00001``) and, like
    ``Saver.save``, the ``checkpoint`` state file next to it."""
    prefix = str(prefix)
    items = []
    offset = 0
    with open(_data_path(prefix, 0, 1), 'wb') as f:
        for name in sorted(variables, key=lambda s: s.encode()):
            shape = np.shape(variables[name])                   # (ascontiguousarray turns a scalar into shape (1,))
            arr = np.ascontiguousarray(variables[name])
            arr = arr.astype(arr.dtype.newbyteorder('<'), copy=False)
            key = arr.dtype.str
            if key not in _DTYPE_ENUM:
                raise ValueError('variable %r: dtype %s cannot be stored' % (name, arr.dtype))
            f.write(arr.tobytes())
            items.append((name.encode(), _encode_entry(_DTYPE_ENUM[key], shape, 0, offset, arr.nbytes,
                                                       mask_crc(crc32c(arr)))))
            offset += arr.nbytes
    header = _varint((1 << 3) | 0) + _varint(1)                             # num_shards = 1, endianness LITTLE (default)
    header += _ld(3, _varint((1 << 3) | 0) + _varint(1))                    # version { producer: 1 }
    write_table(prefix + '.index', [(b'', header)] + items)
    if write_state:
        with open(os.path.join(os.path.dirname(prefix) or '.', 'checkpoint'), 'w') as f:
            base = os.path.basename(prefix)
            f.write('model_checkpoint_path: "%s"\nall_model_checkpoint_paths: "%s"\n' % (base, base))


# -------------------------------------------------------------------------------------------------- variable name table
def _uid(counts, prefix):
    """Keras auto-naming: the n-th layer of a class in a graph is ``prefix`` then ``prefix_1``, ``prefix_2`` ..."""
    n = counts.get(prefix, 0)
    counts[prefix] = n + 1
    return prefix if n == 0 else '%s_%d' % (prefix, n)


def variable_names(config, algo='dien'):
    """OrderedDict graph variable name -> weight name of this package, following the layer creation order of
    rl4rs/nets/<algo>.py (category branch, dense tower, sequence branch, head).  Entries whose value is None are
    variables the reference's graph holds but the scoring path never reads (dnn.py:33 builds a sequence branch it
    does not connect)."""
    S = config['seq_num']
    c = {}
    names = OrderedDict()
    names[_uid(c, 'embedding') + '/embeddings'] = 'cat_emb'            # utils.py:10 / :20 / :32 / :42
    if algo == 'lstm':
        g = _uid(c, 'gru')                                             # utils.py:34
        names[g + '/kernel'] = 'cat_gru_kernel'
        names[g + '/recurrent_kernel'] = 'cat_gru_recurrent'
        names[g + '/bias'] = 'cat_gru_bias'
    for i in (1, 2):                                                   # utils.py:50,52
        d = _uid(c, 'dense')
        names[d + '/kernel'] = 'dense_w%d' % i
        names[d + '/bias'] = 'dense_b%d' % i
    names[_uid(c, 'embedding') + '/embeddings'] = None if algo == 'dnn' else 'seq_emb'    # utils.py:64 / :86 / :113
    if algo == 'dien':
        for i in range(S):                                             # utils.py:117-125
            g = _uid(c, 'dynamic_gru')
            names[g + '/gru_cell/gates/kernel'] = 'gru%d_gate_w' % i
            names[g + '/gru_cell/gates/bias'] = 'gru%d_gate_b' % i
            names[g + '/gru_cell/candidate/kernel'] = 'gru%d_cand_w' % i
            names[g + '/gru_cell/candidate/bias'] = 'gru%d_cand_b' % i
            a = '%s/%s' % (_uid(c, 'attention_sequence_pooling_layer'), _uid(c, 'local_activation_unit'))
            dn = a + '/' + _uid(c, 'dnn')
            names[dn + '/kernel0'] = 'att%d_w1' % i
            names[dn + '/bias0'] = 'att%d_b1' % i
            names[dn + '/kernel1'] = 'att%d_w2' % i
            names[dn + '/bias1'] = 'att%d_b2' % i
            names[a + '/kernel'] = 'att%d_w3' % i
            names[a + '/bias'] = 'att%d_b3' % i
            g = _uid(c, 'dynamic_gru')
            names[g + '/vec_att_gru_cell/gates/kernel'] = 'augru%d_gate_w' % i
            names[g + '/vec_att_gru_cell/gates/bias'] = 'augru%d_gate_b' % i
            names[g + '/vec_att_gru_cell/candidate/kernel'] = 'augru%d_cand_w' % i
            names[g + '/vec_att_gru_cell/candidate/bias'] = 'augru%d_cand_b' % i
    elif algo == 'lstm':
        for i in range(S):                                             # utils.py:92
            g = _uid(c, 'gru')
            names[g + '/kernel'] = 'seq%d_gru_kernel' % i
            names[g + '/recurrent_kernel'] = 'seq%d_gru_recurrent' % i
            names[g + '/bias'] = 'seq%d_gru_bias' % i
    elif algo in ('dnn', 'widedeep'):
        d = _uid(c, 'dense')                                           # dnn.py:35 / widedeep.py:34
        names[d + '/kernel'] = 'fc_w'
        names[d + '/bias'] = 'fc_b'
    else:
        raise ValueError('unknown simulator family %r' % (algo,))
    if algo != 'widedeep':                                             # widedeep's simulator_obs is a Concatenate
        names['simulator_obs/kernel'] = 'obs_w'
        names['simulator_obs/bias'] = 'obs_b'
    names['simulator_reward/kernel'] = 'out_w'
    names['simulator_reward/bias'] = 'out_b'
    return names


_OPTIMIZER_SLOT = re.compile(r'Adam|beta\d_power|training/|/(m|v)$')


def _spec(config, algo):
    if algo == 'dien':
        from ..nets import dien
        return dien.dien_spec(config)
    from ..nets import simnets
    return simnets.simnet_spec(config, algo)


def load_simulator_weights(model_file, config, algo='dien', name_map=None, verify=True):
    """Weights of one simulator family from a TF checkpoint, under this package's names and shapes.

    ``name_map`` ({weight name: graph variable name}) overrides single entries of ``variable_names``.  Optimiser
    slots, metric accumulators and unconnected variables in the checkpoint are ignored.  Raises ``KeyError`` with
    the unmatched names of BOTH sides when the checkpoint does not hold every weight the family needs."""
    spec = _spec(config, algo)
    table = dict((w, v) for v, w in variable_names(config, algo).items() if w is not None)
    table.update(name_map or {})
    present = dict((n, (shape, dt)) for n, shape, dt in list_variables(model_file))
    for w, shape in spec.items():
        # a differing inner scope (other deepctr / TF release) is tolerated when layer (first component), leaf (last
        # component) and shape identify exactly one model variable of the checkpoint
        var = table.get(w)
        if var is None or var in present or (name_map and w in name_map):
            continue
        head, leaf = var.split('/')[0], var.split('/')[-1]
        cand = [n for n, (sh, _) in present.items() if n.split('/')[0] == head and n.split('/')[-1] == leaf
                and tuple(sh) == tuple(shape) and not _OPTIMIZER_SLOT.search(n)]
        if len(cand) == 1:
            table[w] = cand[0]
    missing = [(w, table.get(w)) for w in spec if table.get(w) not in present]
    if missing:
        used = set(table.get(w) for w in spec)
        spare = sorted(n for n in present if n not in used and not _OPTIMIZER_SLOT.search(n))
        raise KeyError('checkpoint %s lacks %s; its unmatched variables are %s - pass name_map={weight: variable}'
                       % (resolve_prefix(model_file), ', '.join('%s (expected %r)' % m for m in missing),
                          ', '.join('%s%r' % (n, tuple(present[n][0])) for n in spare) or 'none'))
    raw = read_checkpoint(model_file, names=[table[w] for w in spec], verify=verify)
    out = OrderedDict()
    for w, shape in spec.items():
        arr = raw[table[w]]
        if tuple(arr.shape) != tuple(shape):
            raise ValueError('checkpoint variable %r has shape %r, %s weight %r needs %r'
                             % (table[w], tuple(arr.shape), algo, w, tuple(shape)))
        out[w] = np.ascontiguousarray(arr, dtype=np.float32)
    return out


def save_simulator_weights(prefix, weights, config, algo='dien'):
    """Write weights (this package's names) as a checkpoint under the reference's graph variable names - the
    counterpart of ``saver.save(sess, model_file)`` (supervised_train.py:44-46) for the model variables.  Variables
    of the reference's graph that carry no model state (the unconnected embedding of dnn.py, metric accumulators of
    ``model.compile``) are written as zeros where their shape is known and otherwise left out."""
    spec = _spec(config, algo)
    out = {}
    for var, w in variable_names(config, algo).items():
        if w is None:
            out[var] = np.zeros((config['category_hash_size'], config['emb_size']), np.float32)
            continue
        arr = np.asarray(weights[w], dtype=np.float32)
        if tuple(arr.shape) != tuple(spec[w]):
            raise ValueError('weight %r has shape %r, expected %r' % (w, tuple(arr.shape), tuple(spec[w])))
        out[var] = arr
    write_checkpoint(prefix, out)


if __name__ == '__main__':                              # python -m rl4rs_amd.utils.tfckpt <prefix>: list the variables
    import sys
    for n, shape, dt in list_variables(sys.argv[1]):
        print('%-72s %-18r %s' % (n, tuple(shape), dt))
