"""TF checkpoint (tensor bundle) reader / writer of rl4rs_amd.utils.tfckpt - CPU only.

The reference restores its simulator from a ``tf.train.Saver`` checkpoint (rl4rs/env/base.py:129,148-151;
written by script/supervised_train.py:44-46).  No TensorFlow here: the format is checked against its published
constants (table magic, CRC-32C known answers, masking, block layout) and by round trips; a hand-assembled index
with prefix compression, several restart points and a snappy block exercises the reader on bytes the writer did not
produce.
"""
import os
import struct

import numpy as np
import pytest

from rl4rs_amd.nets import dien, simnets
from rl4rs_amd.utils import tfckpt, tfrecord

CFG = dict(category_hash_size=300, emb_size=128, hidden_units=128, dense_feature_num=432, category_feature_num=21,
           seq_num=2, class_num=2, maxlen=64)


def test_crc32c_known_answers_and_streaming():
    assert tfckpt.crc32c(b'123456789') == 0xE3069283                    # CRC-32C check value (RFC 3720 B.4)
    assert tfckpt.crc32c(b'\x00' * 32) == 0x8A9136AA
    assert tfckpt.crc32c(b'\xff' * 32) == 0x62A8AB43
    assert tfckpt.crc32c(bytes(range(32))) == 0x46DD794E
    blob = np.random.RandomState(0).randint(0, 256, 100003).astype(np.uint8).tobytes()
    assert tfckpt.crc32c(blob) == tfrecord._crc32c_py(blob) == tfrecord.crc32c(blob)    # slicing-by-8 == bytewise table
    for cut in (0, 1, 7, 8, 9, 4096, len(blob)):
        assert tfckpt.crc32c(blob[cut:], tfckpt.crc32c(blob[:cut])) == tfckpt.crc32c(blob)
    assert tfckpt.crc32c(np.frombuffer(blob, np.uint8)) == tfckpt.crc32c(blob)
    for c in (0, 1, 0x12345678, 0xFFFFFFFF, tfckpt.crc32c(blob)):
        assert tfckpt.unmask_crc(tfckpt.mask_crc(c)) == c
        assert tfckpt.mask_crc(c) == (((c >> 15) | (c << 17)) + 0xa282ead8) & 0xFFFFFFFF


def test_round_trip_dtypes_scalars_and_many_blocks(tmp_path):
    rs = np.random.RandomState(1)
    v = {'a/kernel': rs.randn(3, 4).astype('f4'), 'a/bias': rs.randn(4).astype('f4'), 'global_step': np.int64(7),
         'flag': np.array([True, False]), 'emb': rs.randn(1000, 128).astype('f4'), 'empty': np.zeros((0, 5), 'f4'),
         'half': rs.randn(5).astype('f2'), 'i32': rs.randint(-9, 9, (2, 3, 4)).astype('i4'),
         'big_endian': rs.randn(6).astype('>f8')}
    for i in range(400):                                                # index spans many 4 KB blocks
        v['layer_%d/some/long/variable/name/kernel' % i] = rs.randn(2, 2).astype('f8')
    prefix = str(tmp_path / 'model.ckpt')
    tfckpt.write_checkpoint(prefix, v)
    assert sorted(os.listdir(str(tmp_path))) == ['checkpoint', 'model.ckpt.data-00000-of-00001', 'model.ckpt.index']
    for target in (prefix, prefix + '.index', str(tmp_path)):          # prefix, index path, directory + state file
        got = tfckpt.read_checkpoint(target)
        assert list(got) == sorted(v, key=lambda s: s.encode())
        for k in v:
            want = np.asarray(v[k])
            assert got[k].shape == want.shape and got[k].dtype == want.dtype.newbyteorder('<'), k
            assert np.array_equal(got[k], want), k
    listed = tfckpt.list_variables(prefix)
    assert ('emb', (1000, 128), '<f4') in listed and ('global_step', (), '<i8') in listed
    only = tfckpt.read_checkpoint(prefix, names=['a/bias'])
    assert list(only) == ['a/bias']
    with pytest.raises(KeyError):
        tfckpt.read_checkpoint(prefix, names=['nope'])


def test_file_layout_constants(tmp_path):
    prefix = str(tmp_path / 'm')
    tfckpt.write_checkpoint(prefix, {'w': np.arange(6, dtype='f4').reshape(2, 3)})
    idx = open(prefix + '.index', 'rb').read()
    assert struct.unpack('<Q', idx[-8:])[0] == 0xdb4775248b80fb57       # table magic
    footer, pos = idx[-48:], 0
    for _ in range(4):                                                  # metaindex + index handles, then zero padding
        _, pos = tfrecord._read_varint(footer, pos)
    assert pos <= 40 and footer[pos:40] == b'\x00' * (40 - pos)
    assert open(prefix + '.data-00000-of-00001', 'rb').read() == np.arange(6, dtype='<f4').tobytes()
    entries = tfckpt.read_table(prefix + '.index')
    assert [k for k, _ in entries] == [b'', b'w']
    e = tfckpt._decode_entry(entries[1][1])
    assert e['dtype'] == 1 and e['shape'] == (2, 3) and e['offset'] == 0 and e['size'] == 24
    assert tfckpt.unmask_crc(e['crc32c']) == tfckpt.crc32c(np.arange(6, dtype='<f4').tobytes())
    header = dict((f, v) for f, _, v in tfrecord._fields(entries[0][1]))
    assert header[1] == 1 and 2 not in header                           # one shard, little endian


def test_corruption_is_detected(tmp_path):
    prefix = str(tmp_path / 'm')
    tfckpt.write_checkpoint(prefix, {'w': np.arange(64, dtype='f4'), 'x': np.ones(3, 'f4')})
    data = bytearray(open(prefix + '.data-00000-of-00001', 'rb').read())
    data[10] ^= 0x40
    open(prefix + '.data-00000-of-00001', 'wb').write(bytes(data))
    with pytest.raises(ValueError, match='crc32c'):
        tfckpt.read_checkpoint(prefix)
    assert tfckpt.read_checkpoint(prefix, verify=False)['w'].shape == (64,)
    idx = bytearray(open(prefix + '.index', 'rb').read())
    idx[3] ^= 0x01
    open(prefix + '.index', 'wb').write(bytes(idx))
    with pytest.raises(ValueError):
        tfckpt.read_checkpoint(prefix)
    open(prefix + '.index', 'wb').write(b'not a table at all' * 4)
    with pytest.raises(ValueError, match='magic'):
        tfckpt.read_checkpoint(prefix)
    assert not tfckpt.is_checkpoint(str(tmp_path / 'absent'))


def _snappy_literal_only(raw):
    out = tfrecord._varint(len(raw))
    for i in range(0, len(raw), 60):
        chunk = raw[i:i + 60]
        out += bytes([(len(chunk) - 1) << 2]) + chunk
    return out


def test_snappy_decoder():
    # literals, a 1-byte-offset copy that overlaps its own output (run-length), a 2-byte-offset copy
    comp = bytes([18]) + bytes([2 << 2]) + b'abc' + bytes([((7 - 4) << 2) | 1 | (0 << 5), 3]) \
        + bytes([(8 - 1) << 2 | 2, 10, 0])
    assert tfckpt.snappy_uncompress(comp) == b'abcabcabca' + b'abcabcab'
    long_lit = bytes(range(256)) * 2
    comp = tfrecord._varint(512) + bytes([61 << 2]) + struct.pack('<H', 511) + long_lit      # 2-byte literal length
    assert tfckpt.snappy_uncompress(comp) == long_lit
    with pytest.raises(ValueError):
        tfckpt.snappy_uncompress(bytes([4]) + bytes([(4 - 4) << 2 | 1, 9]))                   # copy before any output


def test_reader_on_a_hand_assembled_index(tmp_path):
    """Blocks built here byte by byte: shared-prefix entries, restart interval 2, one snappy block, a non-empty
    metaindex handle - the reader must not depend on choices our writer makes."""
    def block(pairs, interval):
        buf, restarts, last = b'', [], b''
        for i, (k, v) in enumerate(pairs):
            shared = 0
            if i % interval == 0:
                restarts.append(len(buf))
            else:
                while shared < min(len(k), len(last)) and k[shared] == last[shared]:
                    shared += 1
            buf += tfrecord._varint(shared) + tfrecord._varint(len(k) - shared) + tfrecord._varint(len(v)) + k[shared:] + v
            last = k
        return buf + b''.join(struct.pack('<I', r) for r in restarts) + struct.pack('<I', len(restarts))

    out = bytearray()

    def emit(raw, snappy=False):
        body = _snappy_literal_only(raw) if snappy else raw
        off = len(out)
        t = bytes([1 if snappy else 0])
        out.extend(body + t + struct.pack('<I', tfckpt.mask_crc(tfckpt.crc32c(body + t))))
        return tfrecord._varint(off) + tfrecord._varint(len(body))

    tensors = {'dense/bias': np.arange(4, dtype='<f4'), 'dense/kernel': np.arange(8, dtype='<f4').reshape(2, 4),
               'dense_1/bias': np.ones(2, '<f4'), 'dense_1/kernel': np.full((4, 2), 2.5, '<f4'),
               'step': np.array(3, '<i8')}
    data, entries, off = b'', [], 0
    for name in sorted(tensors):
        a = tensors[name]
        entries.append((name.encode(), tfckpt._encode_entry(9 if a.dtype.kind == 'i' else 1, a.shape, 0, off, a.nbytes,
                                                            tfckpt.mask_crc(tfckpt.crc32c(a.tobytes())))))
        data += a.tobytes()
        off += a.nbytes
    header = (b'', b'\x08\x01\x1a\x02\x08\x01')                          # num_shards 1, version{producer 1}
    h1 = emit(block([header] + entries[:2], 2))
    h2 = emit(block(entries[2:], 2), snappy=True)
    meta = emit(block([], 16))
    index = emit(block([(b'dense/kernel', h1), (b'zzzz', h2)], 1))       # separator key past the last entry
    footer = meta + index
    out.extend(footer + b'\x00' * (40 - len(footer)) + struct.pack('<Q', tfckpt.TABLE_MAGIC))
    prefix = str(tmp_path / 'hand')
    open(prefix + '.index', 'wb').write(bytes(out))
    open(prefix + '.data-00000-of-00001', 'wb').write(data)
    got = tfckpt.read_checkpoint(prefix)
    assert list(got) == sorted(tensors)
    for k, a in tensors.items():
        assert np.array_equal(got[k], a) and got[k].shape == a.shape and got[k].dtype == a.dtype


@pytest.mark.parametrize('algo', ['dien', 'dnn', 'widedeep', 'lstm'])
def test_simulator_weights_through_the_reference_variable_names(tmp_path, algo):
    w = dien.init_dien_weights(CFG, seed=3, bias_noise=0.1) if algo == 'dien' \
        else simnets.init_simnet_weights(CFG, algo, seed=3, bias_noise=0.1)
    prefix = str(tmp_path / ('simulator_' + algo))
    tfckpt.save_simulator_weights(prefix, w, CFG, algo)
    names = [n for n, _, _ in tfckpt.list_variables(prefix)]
    assert 'simulator_reward/kernel' in names and 'embedding/embeddings' in names and 'dense_1/bias' in names
    assert ('simulator_obs/kernel' in names) == (algo != 'widedeep')     # widedeep.py:35: simulator_obs is a Concatenate
    if algo == 'dien':
        assert 'dynamic_gru_3/vec_att_gru_cell/gates/kernel' in names
        assert 'attention_sequence_pooling_layer_1/local_activation_unit_1/dnn_1/kernel0' in names
    if algo == 'dnn':
        assert 'embedding_1/embeddings' in names                         # dnn.py:33 builds it, nothing reads it
    back = tfckpt.load_simulator_weights(prefix, CFG, algo)
    assert list(back) == list(w)
    for k in w:
        assert back[k].dtype == np.float32 and np.array_equal(back[k], np.asarray(w[k], np.float32)), k


def test_name_table_mismatch_is_loud_and_overridable(tmp_path):
    w = simnets.init_simnet_weights(CFG, 'dnn', seed=3)
    prefix = str(tmp_path / 'm')
    tfckpt.save_simulator_weights(prefix, w, CFG, 'dnn')
    raw = tfckpt.read_checkpoint(prefix)
    # an optimiser slot and a metric accumulator, as a training-time Saver would add; renamed head kernel
    raw['training/Adam/dense/kernel/m'] = np.zeros_like(raw['dense/kernel'])
    raw['auc/true_positives'] = np.zeros(200, 'f4')
    raw['final_scores/kernel'] = raw.pop('simulator_reward/kernel')
    # a different inner scope for one layer: still identified by layer + leaf + shape
    raw['dense_2/extra_scope/kernel'] = raw.pop('dense_2/kernel')
    tfckpt.write_checkpoint(prefix, raw)
    with pytest.raises(KeyError) as ei:
        tfckpt.load_simulator_weights(prefix, CFG, 'dnn')
    msg = str(ei.value)
    assert 'out_w' in msg and 'final_scores/kernel' in msg and 'Adam' not in msg
    back = tfckpt.load_simulator_weights(prefix, CFG, 'dnn', name_map={'out_w': 'final_scores/kernel'})
    for k in w:
        assert np.array_equal(back[k], np.asarray(w[k], np.float32)), k
    with pytest.raises(ValueError, match='shape'):
        tfckpt.load_simulator_weights(prefix, CFG, 'dnn', name_map={'out_w': 'dense/kernel'})
