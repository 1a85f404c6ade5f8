"""Log-record parsing.  ORACLE — test infrastructure only (see oracle/__init__.py).

Follows ``FeatureUtil.record_split`` (``rl4rs/utils/datautil.py:20-32``) and
``SlateState.records_to_state`` (``rl4rs/env/slate.py:67-83``).
"""
import numpy as np


def pad_sequences(seqs, maxlen, dtype='int32', padding='pre', truncating='pre', value=0.):
    """Keras-preprocessing 1.1.2 ``pad_sequences`` semantics (call sites datautil.py:43-65)."""
    out = np.full((len(seqs), maxlen), value, dtype=dtype)
    for i, s in enumerate(seqs):
        s = list(s)
        if len(s) == 0:
            continue
        t = s[-maxlen:] if truncating == 'pre' else s[:maxlen]
        t = np.asarray(t, dtype=dtype)
        if padding == 'post':
            out[i, :len(t)] = t
        else:
            out[i, -len(t):] = t
    return out


class ParsedRecords(object):
    """Columnar view of a batch of ``@``-records.

    exposed[B,n] int64, feedback[B,n] int64 (rectangular: the reference builds np.array of them,
    slate.py:169-171), history (ragged list of int lists), user_dense[B,32] f64, user_cat[B,10] int64.
    """

    def __init__(self, records):
        self.records = list(records)
        exposed, feedback, hist, udense, ucat, users = [], [], [], [], [], []
        for rec in self.records:
            f = rec.split('@')
            # datautil.py:22-32
            exposed.append(list(map(int, f[3].split(','))))
            feedback.append(list(map(int, f[4].split(','))))
            hist.append(list(map(int, f[5].split(','))))
            portrait = list(map(float, f[6].split(',')))
            # slate.py:78-79: dense = portrait[10:], category = portrait[:10]
            udense.append(portrait[10:])
            # datautil.py:49: list(map(int, category_feature)) truncates the float ids
            ucat.append([int(x) for x in portrait[:10]])
            users.append(f[1])
        self.exposed = exposed
        self.feedback = feedback
        self.history = hist
        self.user_dense = np.array(udense, dtype=np.float64)
        self.user_cat = np.array(ucat, dtype=np.int64)
        self.users = users
