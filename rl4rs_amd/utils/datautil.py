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
