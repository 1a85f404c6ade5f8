"""DIEN simulator-net parameter set (host side).

Topology follows the reference's ``rl4rs/nets/dien.py:8-45`` and ``rl4rs/nets/utils.py:16-25,48-54,
100-129``; variable shapes follow the Keras / TF-1.15 ``GRUCell`` / deepctr-0.9.0 ``VecAttGRUCell`` /
``LocalActivationUnit`` variables those call sites create:

===================  =====================  ==========================================================
name                 shape                  reference variable
===================  =====================  ==========================================================
cat_emb              [H, E]                 Embedding in id_input_processing_attn (utils.py:20)
dense_w1/b1          [Dn, U] / [U]          Dense(hidden, ELU) (utils.py:50)
dense_w2/b2          [U, U] / [U]           Dense(hidden, ELU) (utils.py:52)
seq_emb              [H, E]                 Embedding shared by both sequences + query (utils.py:113)
gru{i}_gate_w/b      [2E, 2E] / [2E]        GRUCell gates, rows = [x ; h], cols = [r | u], bias init 1
gru{i}_cand_w/b      [2E, E] / [E]          GRUCell candidate, rows = [x ; r*h]
att{i}_w1/b1         [4E, 64] / [64]        LocalActivationUnit DNN layer 1 (sigmoid), rows=[q;k;q-k;q*k]
att{i}_w2/b2         [64, 16] / [16]        DNN layer 2 (sigmoid)
att{i}_w3/b3         [16, 1] / [1]          final linear score
augru{i}_gate_w/b    [E+2E, 4E] / [4E]      VecAttGRUCell gates (num_units = 2E), rows=[x ; h]
augru{i}_cand_w/b    [E+2E, 2E] / [2E]      VecAttGRUCell candidate
obs_w/b              [2*2E+U+(Cn+1)E, 256]  Dense(256, ELU) 'simulator_obs' (dien.py:35)
out_w/b              [256, class_num]       Dense(class_num, softmax) 'simulator_reward' (dien.py:36)
===================  =====================  ==========================================================

There is no checkpoint in the reference tree (README.md:124-135 are external downloads) and no TF here,
so weights are seeded synthetic (Keras default initialisers) or loaded from an ``.npz`` with these names.
"""
from collections import OrderedDict

import numpy as np

OBS_DIM = 256
ATT_H1 = 64
ATT_H2 = 16


def dien_spec(config):
    H = config['category_hash_size']
    E = config['emb_size']
    U = config['hidden_units']
    Dn = config['dense_feature_num']
    Cn = config['category_feature_num']
    S = config['seq_num']
    K = config['class_num']
    spec = OrderedDict()
    spec['cat_emb'] = (H, E)
    spec['dense_w1'] = (Dn, U)
    spec['dense_b1'] = (U,)
    spec['dense_w2'] = (U, U)
    spec['dense_b2'] = (U,)
    spec['seq_emb'] = (H, E)
    for i in range(S):
        spec['gru%d_gate_w' % i] = (2 * E, 2 * E)
        spec['gru%d_gate_b' % i] = (2 * E,)
        spec['gru%d_cand_w' % i] = (2 * E, E)
        spec['gru%d_cand_b' % i] = (E,)
        spec['att%d_w1' % i] = (4 * E, ATT_H1)
        spec['att%d_b1' % i] = (ATT_H1,)
        spec['att%d_w2' % i] = (ATT_H1, ATT_H2)
        spec['att%d_b2' % i] = (ATT_H2,)
        spec['att%d_w3' % i] = (ATT_H2, 1)
        spec['att%d_b3' % i] = (1,)
        spec['augru%d_gate_w' % i] = (3 * E, 4 * E)
        spec['augru%d_gate_b' % i] = (4 * E,)
        spec['augru%d_cand_w' % i] = (3 * E, 2 * E)
        spec['augru%d_cand_b' % i] = (2 * E,)
    spec['obs_w'] = (S * 2 * E + U + (Cn + 1) * E, OBS_DIM)
    spec['obs_b'] = (OBS_DIM,)
    spec['out_w'] = (OBS_DIM, K)
    spec['out_b'] = (K,)
    return spec


def init_dien_weights(config, seed=7, emb_scale=0.05, gain=1.0, bias_noise=0.0):
    """Seeded synthetic weights with the Keras/TF default initialisers.

    Dense/GRU kernels: glorot-uniform (x ``gain``); embeddings: U(-emb_scale, emb_scale); GRU gate
    biases 1.0, everything else 0 (+ optional ``bias_noise`` so parity tests exercise every bias).
    """
    rs = np.random.RandomState(seed)
    out = OrderedDict()
    for name, shape in dien_spec(config).items():
        if name.endswith('_emb'):
            w = rs.uniform(-emb_scale, emb_scale, size=shape)
        elif name.endswith('_w3'):
            # raw (un-normalised) attention scores scale the AUGRU update gate as (1 - a_t) * u
            # (deepctr VecAttGRUCell); a trained net keeps a_t in a sane range, random signed weights
            # do not (u > 1 => exponential blow-up over 64 steps), so synthetic scores live in (0, 1).
            w = rs.uniform(0.0, 1.0 / shape[0], size=shape)
        elif len(shape) == 2:
            lim = gain * np.sqrt(6.0 / (shape[0] + shape[1]))
            w = rs.uniform(-lim, lim, size=shape)
        else:
            w = np.zeros(shape)
            if 'gate_b' in name:
                w += 1.0
            if bias_noise:
                w = w + rs.uniform(-bias_noise, bias_noise, size=shape)
        out[name] = np.ascontiguousarray(w, dtype=np.float32)
    return out


def save_weights(path, weights):
    np.savez(path, **weights)


def load_weights(path, config=None):
    z = np.load(path)
    w = OrderedDict((k, np.ascontiguousarray(z[k], dtype=np.float32)) for k in z.files)
    if config is not None:
        check_weights(w, config)
    return w


def check_weights(weights, config):
    for name, shape in dien_spec(config).items():
        if name not in weights:
            raise KeyError('DIEN weight %r missing' % name)
        if tuple(weights[name].shape) != tuple(shape):
            raise ValueError('DIEN weight %r has shape %r, expected %r'
                             % (name, tuple(weights[name].shape), tuple(shape)))
