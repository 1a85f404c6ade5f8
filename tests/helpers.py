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


SCENARIOS = ['slate_discrete', 'slate_conti', 'seq36_discrete', 'seq36_conti', 'seq32_discrete',
             'seq32_conti', 'real_discrete', 'real_conti']
