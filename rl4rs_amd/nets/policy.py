"""Host side of the action-masked policy net (``rl4rs/nets/rllib/rllib_mask_model.py:7-64``).

Flat parameter layout shared with ``rl4rs_amd/csrc/policy.hip``:
``[ W1 (obs_dim x hidden) | b1 (hidden) | W2e (hidden x (A+1)) | b2e (A+1) ]`` where column ``A`` of layer 2 is
the value head (``vf_share_layers=True``, rllib_mask_model.py:34).  34 973 parameters at 256/64/284.
"""
import numpy as np


def param_count(obs_dim, hidden, action_size):
    return obs_dim * hidden + hidden + hidden * (action_size + 1) + (action_size + 1)


def split(flat, obs_dim, hidden, action_size):
    """Views (W1, b1, W2e, b2e) of a flat parameter / gradient vector (numpy or torch)."""
    ae = action_size + 1
    o = 0
    W1 = flat[o:o + obs_dim * hidden].reshape(obs_dim, hidden)
    o += obs_dim * hidden
    b1 = flat[o:o + hidden]
    o += hidden
    W2 = flat[o:o + hidden * ae].reshape(hidden, ae)
    o += hidden * ae
    b2 = flat[o:o + ae]
    return W1, b1, W2, b2


def init_policy_params(obs_dim=256, hidden=64, action_size=284, seed=0):
    """RLlib FullyConnectedNetwork initialisation: normc(1.0) hidden layer, normc(0.01) output layers, zero biases."""
    rs = np.random.RandomState(seed)

    def normc(shape, std):
        w = rs.randn(*shape)
        return w * std / np.sqrt(np.square(w).sum(axis=0, keepdims=True))

    W1 = normc((obs_dim, hidden), 1.0)
    W2 = np.concatenate([normc((hidden, action_size), 0.01), normc((hidden, 1), 0.01)], axis=1)
    flat = np.concatenate([W1.ravel(), np.zeros(hidden), W2.ravel(), np.zeros(action_size + 1)])
    return flat.astype(np.float32)
