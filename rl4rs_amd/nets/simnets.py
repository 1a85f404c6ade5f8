"""Parameter sets of the reference's other simulator families (config['algo'] = 'dnn' | 'widedeep' | 'lstm').

Topology follows ``rl4rs/nets/dnn.py:31-37``, ``rl4rs/nets/widedeep.py:31-38``, ``rl4rs/nets/lstm.py:31-37`` and
the helpers of ``rl4rs/nets/utils.py`` (line numbers below).  All three keep the 'simulator_obs' /
'simulator_reward' contract of the DIEN model, so the env facade treats them alike.

=====================  ==========================  ===========================================================
name                   shape                       reference variable
=====================  ==========================  ===========================================================
cat_emb                [H, E]                      Embedding of id_input_processing / _concat / _lstm (utils.py:7-45)
dense_w1/b1, w2/b2     [Dn,U]/[U], [U,U]/[U]       dense_input_processing (utils.py:48-54)
seq_emb                [H, E]                      the ONE Embedding shared by all sequences (utils.py:64, 87)
                                                   (widedeep, lstm; the dnn model builds it too but never uses it)
fc_w/b                 dnn: [E+U, 256]             Dense(256, ELU) on [category ‖ dense]            (dnn.py:35)
                       widedeep: [S*E, 256]        Dense(256, ELU) on the pooled sequences          (widedeep.py:34)
obs_w/b                dnn: [256, 256]             Dense(256, ELU) 'simulator_obs'                  (dnn.py:36)
                       lstm: [S*U+U+U+Cn*E, 256]   Dense(256, ELU) 'simulator_obs' on
                                                   [sequences ‖ dense ‖ category GRU ‖ flatten]     (lstm.py:35-36)
                       widedeep: none — 'simulator_obs' is the Concatenate itself, 256+U+Cn*E wide (widedeep.py:35-37)
out_w/b                [obs_dim, class_num]        Dense(class_num, softmax) 'simulator_reward'
cat_gru_kernel/        [E,3U] / [U,3U] / [3U]      keras GRU(units=U) over the category embeddings (utils.py:34), gate
 recurrent/bias                                    order z | r | h
seq{i}_gru_kernel/...  same shapes                 keras GRU(units=U) per sequence (utils.py:91)
=====================  ==========================  ===========================================================

keras GRU semantics **[from memory of TF 1.15 ``tf.keras.layers.GRU`` = recurrent.GRU v1 defaults; parity unpinned]**:
recurrent_activation = hard_sigmoid, reset_after = False, one bias vector:
    z = hs(x Wz + h Uz + bz); r = hs(x Wr + h Ur + br); hh = tanh(x Wh + (r*h) Uh + bh); h' = z*h + (1-z)*hh.
"""
from collections import OrderedDict

import numpy as np

ALGOS = ('dnn', 'widedeep', 'lstm')
FC_DIM = 256


def obs_dim(config, algo):
    if algo == 'widedeep':
        return FC_DIM + config['hidden_units'] + config['category_feature_num'] * config['emb_size']
    return 256


def simnet_spec(config, algo):
    if algo not in ALGOS:
        raise ValueError('algo must be one of %r (got %r)' % (ALGOS, algo))
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
    if algo in ('widedeep', 'lstm'):
        spec['seq_emb'] = (H, E)
    if algo == 'dnn':
        spec['fc_w'] = (E + U, FC_DIM)
        spec['fc_b'] = (FC_DIM,)
        spec['obs_w'] = (FC_DIM, 256)
        spec['obs_b'] = (256,)
    elif algo == 'widedeep':
        spec['fc_w'] = (S * E, FC_DIM)
        spec['fc_b'] = (FC_DIM,)
    else:
        spec['cat_gru_kernel'] = (E, 3 * U)
        spec['cat_gru_recurrent'] = (U, 3 * U)
        spec['cat_gru_bias'] = (3 * U,)
        for i in range(S):
            spec['seq%d_gru_kernel' % i] = (E, 3 * U)
            spec['seq%d_gru_recurrent' % i] = (U, 3 * U)
            spec['seq%d_gru_bias' % i] = (3 * U,)
        spec['obs_w'] = (S * U + U + U + Cn * E, 256)
        spec['obs_b'] = (256,)
    spec['out_w'] = (obs_dim(config, algo), K)
    spec['out_b'] = (K,)
    return spec


def init_simnet_weights(config, algo, seed=7, emb_scale=0.05, bias_noise=0.0):
    """Seeded synthetic weights: glorot-uniform kernels, U(-emb_scale, emb_scale) embeddings, zero biases
    (+ optional noise so parity tests exercise every bias)."""
    rs = np.random.RandomState(seed)
    out = OrderedDict()
    for name, shape in simnet_spec(config, algo).items():
        if name.endswith('_emb'):
            w = rs.uniform(-emb_scale, emb_scale, size=shape)
        elif len(shape) == 2:
            lim = np.sqrt(6.0 / (shape[0] + shape[1]))
            w = rs.uniform(-lim, lim, size=shape)
        else:
            w = np.zeros(shape)
            if bias_noise:
                w = w + rs.uniform(-bias_noise, bias_noise, size=shape)
        out[name] = np.ascontiguousarray(w, dtype=np.float32)
    return out


def load_weights(path, config, algo):
    z = np.load(path)
    w = OrderedDict((k, np.ascontiguousarray(z[k], dtype=np.float32)) for k in z.files)
    check_weights(w, config, algo)
    return w


def check_weights(weights, config, algo):
    for name, shape in simnet_spec(config, algo).items():
        if name not in weights:
            raise KeyError('%s weight %r missing' % (algo, name))
        if tuple(weights[name].shape) != tuple(shape):
            raise ValueError('%s weight %r has shape %r, expected %r'
                             % (algo, name, tuple(weights[name].shape), tuple(shape)))
