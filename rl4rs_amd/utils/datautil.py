"""FeatureUtil mirror (``rl4rs/utils/datautil.py:8-69``).

``record_split`` parses one ``@``-record (datautil.py:20-32).  ``feature_extraction`` turns state rows
into the four fixed-width arrays the simulator net takes (datautil.py:34-69).  When handed the
device-backed rows of ``SlateState.state`` / ``get_complete_states`` it only copies the tensors the GPU
already built; when handed the reference's nested python lists it pads/truncates on the host
(format conversion only; no model arithmetic happens on the CPU).
"""
import numpy as np


class FeatureUtil(object):

    def __init__(self, config):
        self.config = config
        self.maxlen = config['maxlen']
        self.batch_size = config['batch_size']
        self.class_num = config['class_num']
        self.dense_feature_num = config['dense_feature_num']
        self.category_feature_num = config['category_feature_num']
        self.category_hash_size = config['category_hash_size']
        self.seq_num = self.config['seq_num']

    @classmethod
    def record_split(cls, record):
        f = record.split('@')
        if len(f) != 9:
            raise ValueError('not enough values to unpack (expected 9, got %d)' % len(f))
        ints = lambda s: [int(x) for x in s.split(',')]
        floats = lambda s: [float(x) for x in s.split(',')]
        return (int(f[0]), int(f[1]), int(f[2]), ints(f[3]), ints(f[4]), ints(f[5]), floats(f[6]),
                floats(f[7].replace(';', ',')), int(f[8]))

    @staticmethod
    def _fit(rows, width, dtype, pre):
        out = np.zeros((len(rows), width), dtype=dtype)
        for i, r in enumerate(rows):
            r = list(r)
            if not r:
                continue
            if pre:                      # pad_sequences defaults: keep the last `width`, right-align
                r = r[-width:]
                out[i, width - len(r):] = r
            else:                        # padding='post', truncating='post'
                r = r[:width]
                out[i, :len(r)] = r
        return out

    def feature_extraction(self, data):
        if hasattr(data, 'numpy_features'):
            seq, dense, cat = data.numpy_features()
            n = seq.shape[0]
            return (seq[:, :self.seq_num], dense, cat, np.zeros((n, 9), dtype=np.int64)), [0] * n
        seqs, dense, cat, slate_labels, labels = [], [], [], [], []
        for record in data:
            _, sequence_feature, dense_feature, category_feature, slate_label, label = record
            seqs.append([self._fit([xx], self.maxlen, np.int32, True)[0] for xx in sequence_feature[:self.seq_num]])
            dense.append(dense_feature)
            cat.append([int(x) for x in category_feature])
            slate_labels.append(slate_label)
            labels.append(label)
        return (np.array(seqs), self._fit(dense, self.dense_feature_num, np.float32, False),
                self._fit(cat, self.category_feature_num, np.int32, False), np.array(slate_labels)), labels

    # ------------------------------------------------------------------ TFRecord training sets (datautil.py:71-230)
    def to_tfrecord(self, data, filename):
        """Samples (nested lists as ``feature_extraction`` takes them) -> one ``tf.train.Example`` per sample with the
        reference's feature names (datautil.py:177-230)."""
        from .tfrecord import TFRecordWriter, encode_example
        with TFRecordWriter(filename) as wr:
            for jj in range(0, len(data), 10000):
                (seqs, dense, cat, slate_labels), labels = self.feature_extraction(data[jj:jj + 10000])
                for i in range(len(seqs)):
                    feat = {"dense_feature": np.asarray(dense[i], dtype=np.float32),
                            "category_feature": np.asarray(cat[i], dtype=np.int64),
                            "slate_label": np.asarray(slate_labels[i], dtype=np.int64),
                            "label": np.asarray([labels[i]], dtype=np.int64)}
                    for s in range(self.seq_num):
                        feat['sequence_id_%d' % s] = np.asarray(seqs[i][s], dtype=np.int64)
                    wr.write(encode_example(feat))

    def load_tfrecord(self, filenames, verify=False):
        """All samples of the files as columnar arrays: (sequence_id [n, seq_num, maxlen] i32, dense [n, Dn] f32,
        category [n, Cn] i32, slate_label [n, 9] i64), label [n] i64 - ``_parse_exmp`` + the padded batching of
        datautil.py:74-120 (short features are zero-padded on the right, long ones cut)."""
        from .tfrecord import read_records, decode_example
        if isinstance(filenames, str):
            filenames = [filenames]
        seqs, dense, cat, slate, label = [], [], [], [], []

        def fit(v, width, dtype):
            out = np.zeros(width, dtype=dtype)
            v = np.asarray(v)[:width]
            out[:len(v)] = v
            return out

        for fn in filenames:
            for rec in read_records(fn, verify=verify):
                ex = decode_example(rec)
                seqs.append(np.stack([fit(ex.get('sequence_id_%d' % s, []), self.maxlen, np.int32) for s in range(self.seq_num)]))
                dense.append(fit(ex['dense_feature'], self.dense_feature_num, np.float32))
                cat.append(fit(ex['category_feature'], self.category_feature_num, np.int32))
                slate.append(fit(ex.get('slate_label', []), 9, np.int64))
                label.append(int(ex['label'][0]))
        n = len(label)
        return ((np.stack(seqs) if n else np.zeros((0, self.seq_num, self.maxlen), np.int32),
                 np.stack(dense) if n else np.zeros((0, self.dense_feature_num), np.float32),
                 np.stack(cat) if n else np.zeros((0, self.category_feature_num), np.int32),
                 np.stack(slate) if n else np.zeros((0, 9), np.int64)), np.asarray(label, dtype=np.int64))

    def read_tfrecord(self, filename, is_pred=False, is_slate_label=False, seed=0):
        """Generator of batches like the reference's dataset iterator (datautil.py:71-175):
        ``((sequence_id, dense, category, slate_label), target)`` with target = one_hot(label) or, with ``is_slate_label``,
        the slate labels.  Training mode (``is_pred=False``): shuffled, remainder dropped, repeats forever; prediction mode:
        file order, last batch kept, one pass."""
        (seqs, dense, cat, slate), label = self.load_tfrecord(filename)
        n, B = len(label), self.batch_size
        rs = np.random.RandomState(seed)

        def batch(idx):
            x = (seqs[idx], dense[idx], cat[idx], slate[idx])
            return x, (slate[idx] if is_slate_label else np.eye(self.class_num, dtype=np.float32)[label[idx]])

        if is_pred:
            for lo in range(0, n, B):
                yield batch(np.arange(lo, min(lo + B, n)))
            return
        while n >= B:
            perm = rs.permutation(n)
            for lo in range(0, n - B + 1, B):
                yield batch(perm[lo:lo + B])
