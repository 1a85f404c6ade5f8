"""Shared helpers for the parity tests (tests only)."""
import json
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


def manifest():
    with open(os.path.join(GOLDEN, 'manifest.json')) as f:
        return json.load(f)


def load_scenario(name):
    m = manifest()[name]
    g = np.load(os.path.join(GOLDEN, name + '.npz'))
    cfg = dict(m['config'])
    cfg['iteminfo_file'] = os.path.join(GOLDEN, m['catalog'])
    cfg['support_conti_env'] = bool(m['conti'])
    if m['mask_flag']:
        cfg[m['mask_flag']] = True
    with open(os.path.join(GOLDEN, m['records'])) as f:
        records = [x for x in f.read().split('\n') if x]
    return m, cfg, records, g


def golden_equal(g, key, arr):
    """Bit-exact comparison of ``arr`` with golden array ``key``: stored whole, or (large float arrays of the batch-256
    scenario, make_golden.py::compact) as sha1 digest + shape + dtype + first rows."""
    import hashlib
    if key in g.files:
        return np.array_equal(arr, g[key])
    a = np.ascontiguousarray(arr)
    head = g[key + '__head']
    return (tuple(a.shape) == tuple(g[key + '__shape'].tolist()) and str(a.dtype) == str(g[key + '__dtype'])
            and np.array_equal(a[:len(head)], head)
            and hashlib.sha1(a.tobytes()).digest() == g[key + '__sha1'].tobytes())


def golden_has(g, key):
    return key in g.files or (key + '__sha1') in g.files


SCENARIOS = ['slate_discrete', 'slate_conti', 'seq36_discrete', 'seq36_conti', 'seq32_discrete',
             'seq32_conti', 'real_discrete', 'real_conti', 'slate256_discrete', 'seq36_b64_discrete', 'slate_onehot']
