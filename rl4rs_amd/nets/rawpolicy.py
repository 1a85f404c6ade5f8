"""Parameter set of the raw-state policy encoder (``rl4rs/nets/rllib/rllib_rawstate_model.py:25-86``).

=====================  ====================  ================================================================
name                   shape                 reference variable
=====================  ====================  ================================================================
cat_emb                [H, E]                Embedding of utils.id_input_processing (utils.py:7-14)
dense_w1/b1, w2/b2     [Dn,U]/[U],[U,U]/[U]  utils.dense_input_processing (utils.py:48-54)
seq_emb                [H, E]                the one Embedding of utils.sequence_input_concat (utils.py:57-77)
ctx_w/b                [S*E+U+E, 256]/[256]  Dense(256, ELU) on [sequence | dense | category] (rllib_rawstate_model.py:52-53)
out_w/b                [256, A] / [A]        'fc_out' (normc_initializer(0.01), linear)
value_w/b              [256, 1] / [1]        'value_out'
=====================  ====================  ================================================================
"""
from collections import OrderedDict

import numpy as np


def rawpolicy_spec(config):
    H, E, U = config['category_hash_size'], config['emb_size'], config['hidden_units']
    Dn, S, A = config['dense_feature_num'], config['seq_num'], config['action_size']
    spec = OrderedDict()
    spec['cat_emb'] = (H, E)
    spec['seq_emb'] = (H, E)
    spec['dense_w1'] = (Dn, U)
    spec['dense_b1'] = (U,)
    spec['dense_w2'] = (U, U)
    spec['dense_b2'] = (U,)
    spec['ctx_w'] = (S * E + U + E, 256)
    spec['ctx_b'] = (256,)
    spec['out_w'] = (256, A)
    spec['out_b'] = (A,)
    spec['value_w'] = (256, 1)
    spec['value_b'] = (1,)
    return spec


def init_rawpolicy_weights(config, seed=0, emb_scale=0.05, head_std=0.01, bias_noise=0.0):
    """Keras defaults (glorot-uniform kernels, uniform embeddings); the two heads use RLlib's normc_initializer(0.01):
    column-normalised gaussians."""
    rs = np.random.RandomState(seed)
    out = OrderedDict()
    for name, shape in rawpolicy_spec(config).items():
        if name.endswith('_emb'):
            w = rs.uniform(-emb_scale, emb_scale, size=shape)
        elif name in ('out_w', 'value_w'):
            w = rs.randn(*shape)
            w *= head_std / np.sqrt(np.square(w).sum(axis=0, keepdims=True))
        elif len(shape) == 2:
            lim = np.sqrt(6.0 / (shape[0] + shape[1]))
            w = rs.uniform(-lim, lim, size=shape)
        else:
            w = np.zeros(shape)
            if bias_noise:
                w = w + rs.uniform(-bias_noise, bias_noise, size=shape)
        out[name] = np.ascontiguousarray(w, dtype=np.float32)
    return out
