"""Host-side parsing of the reference's two file formats into columnar arrays.

Runs once per catalogue / log file (not on the step path): the parsed columns are uploaded to HBM and
every ``reset`` is a device-side row gather.

* catalogue: ``SlateState.get_iteminfo_from_file`` / ``get_mask_from_file`` (rl4rs/env/slate.py:28-65)
* records:   ``FeatureUtil.record_split`` (rl4rs/utils/datautil.py:20-32) +
             ``SlateState.records_to_state`` (rl4rs/env/slate.py:67-83) +
             ``pad_sequences`` of the history (datautil.py:43-46; pre-pad, pre-truncate)
"""
import numpy as np


class CatalogTables(object):
    """Host tables handed to ``rl4rs_env_set_catalog`` (see include/rl4rs_hip.h)."""

    def __init__(self, iteminfo_file, action_size, action_emb_size=32, onehot_action=False):
        # slate.py:30-31 — the file has a header line and NO trailing newline
        text = open(iteminfo_file, 'r').read().split('\n')[1:]
        rows = [x.split(' ') for x in text]
        ids = [int(r[0]) for r in rows]
        vecs = np.array([list(map(float, r[1].split(','))) for r in rows], dtype=np.float64)
        self.action_size = action_size
        self.item_dim = vecs.shape[1]
        size = max(action_size, max(ids) + 1)
        item_vec = np.zeros((size, self.item_dim), dtype=np.float64)    # id 0 = zeros (slate.py:42-46)
        price = np.zeros((size,), dtype=np.float64)
        special = np.zeros((size,), dtype=np.uint8)
        for r, i, v in zip(rows, ids, vecs):
            item_vec[i] = v
            price[i] = float(r[2])
            special[i] = 1 if int(r[4]) == 2 else 0
        self.item_info_d = None       # built lazily for API parity (get_iteminfo_from_file)
        self._rows = rows
        self.item_vec64 = item_vec
        # the env emits float32 features: pad_sequences(dtype='float32') casts the float64 parse
        self.item_vec = np.ascontiguousarray(item_vec[:action_size].astype(np.float32))
        self.price = np.ascontiguousarray(price[:action_size])
        self.is_special = np.ascontiguousarray(special[:action_size])
        self.special_items = [i for r, i in zip(rows, ids) if int(r[4]) == 2]
        # slate.py:47-52 — rows 1.. are the L2-normalised last E dims, in file order
        action_emb = np.zeros((action_size, action_emb_size))
        tail = vecs[:, -action_emb_size:]
        action_emb[1:] = np.einsum('ij,i->ij', tail, 1.0 / np.linalg.norm(tail, axis=1))
        if onehot_action:             # slate.py:22-25
            action_emb = np.eye(action_size)
        self.action_emb = np.ascontiguousarray(action_emb, dtype=np.float64)
        # slate.py:59-64 — hard-coded layer boundaries
        loc = np.zeros((4, action_size), dtype=np.int64)
        loc[0, 1:40] = 1
        loc[1, 40:148] = 1
        loc[2, 148:] = 1
        loc[3, 0] = 1
        self.location_mask = loc

    def item_info_dict(self):
        """The dict ``get_iteminfo_from_file`` returns (slate.py:32-46)."""
        if self.item_info_d is None:
            d = dict((str(r[0]), {'item_vec': list(map(float, r[1].split(','))), 'price': float(r[2]),
                                  'location': int(r[3])}) for r in self._rows)
            d['0'] = {'item_vec': [0] * self.item_dim, 'price': float(0), 'location': int(0)}
            self.item_info_d = d
        return self.item_info_d


def pad_history(history, maxlen):
    """pad_sequences([h], maxlen) with the Keras defaults: pre-pad zeros, keep the LAST maxlen ids."""
    out = np.zeros((len(history), maxlen), dtype=np.int32)
    for i, h in enumerate(history):
        t = h[-maxlen:]
        if len(t):
            out[i, maxlen - len(t):] = t
    return out


class RecordColumns(object):
    """Columnar form of a list of ``@``-records (what ``rl4rs_env_load_batch`` takes)."""

    def __init__(self, records, maxlen, log_steps=None):
        n = len(records)
        exposed, feedback, hist, users = [], [], [], []
        portrait = []
        for rec in records:
            f = rec.split('@')
            if len(f) != 9:
                raise ValueError('record has %d fields, expected 9 (datautil.py:22-23)' % len(f))
            exposed.append(np.array(f[3].split(','), dtype=np.int64))
            feedback.append(np.array(f[4].split(','), dtype=np.int64))
            hist.append(np.array(f[5].split(','), dtype=np.int64))
            portrait.append(np.array(f[6].split(','), dtype=np.float64))
            users.append(f[1])
        self.n = n
        self.users = users
        self.exposed_len = np.array([len(x) for x in exposed], dtype=np.int64)
        width = int(self.exposed_len.max()) if n else 0
        if log_steps is not None:
            width = log_steps
        self.log_steps = width
        self.exposed = np.zeros((n, width), dtype=np.int32)
        self.feedback = np.zeros((n, width), dtype=np.int32)
        for i in range(n):
            k = min(len(exposed[i]), width)
            self.exposed[i, :k] = exposed[i][:k]
            k2 = min(len(feedback[i]), width)
            self.feedback[i, :k2] = feedback[i][:k2]
        self.history = pad_history(hist, maxlen)
        p = np.stack(portrait) if n else np.zeros((0, 42))
        # slate.py:78-79: category = portrait[:10] (ids stored as floats; int() truncates, datautil.py:49),
        # dense = portrait[10:] (float64 parse, cast to float32 by pad_sequences datautil.py:52-58)
        self.user_cat = np.ascontiguousarray(p[:, :10].astype(np.int64).astype(np.int32))
        self.user_dense = np.ascontiguousarray(p[:, 10:].astype(np.float32))


def parse_records_native(records, maxlen, log_steps=None):
    """Same columns as ``RecordColumns`` through the C parser of librl4rs_hip.so (``rl4rs_parse_records``): ~100x
    faster than the Python loop, used to load a whole sample file into HBM once."""
    import ctypes as C
    from . import _lib
    lib = _lib.load()
    n = len(records)
    if log_steps is None:
        log_steps = len(records[0].split('@')[3].split(',')) if n else 1
    text = '\n'.join(records).encode()
    out = RecordColumns.__new__(RecordColumns)
    out.n = n
    out.log_steps = log_steps
    out.exposed = np.zeros((n, log_steps), dtype=np.int32)
    out.feedback = np.zeros((n, log_steps), dtype=np.int32)
    out.history = np.zeros((n, maxlen), dtype=np.int32)
    out.user_dense = np.zeros((n, 32), dtype=np.float32)
    out.user_cat = np.zeros((n, 10), dtype=np.int32)
    elen = np.zeros((n,), dtype=np.int32)
    got = C.c_int32()
    p = lambda a: a.ctypes.data_as(C.c_void_p)
    _lib.check(lib.rl4rs_parse_records(text, len(text), n, maxlen, log_steps, 32, 10, p(out.exposed), p(out.feedback),
                                       p(out.history), p(out.user_dense), p(out.user_cat), p(elen), C.byref(got)))
    if got.value != n:
        raise ValueError('parsed %d records, expected %d (blank lines inside the batch?)' % (got.value, n))
    out.exposed_len = elen.astype(np.int64)
    out.users = None
    return out
