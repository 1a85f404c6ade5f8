"""dnn / widedeep / lstm simulator restatement (numpy).  ORACLE — test infrastructure only (see oracle/__init__.py).

PARITY UNPINNED: like the DIEN scorer these models are keras graphs on tensorflow-gpu==1.15.0
(environment.yml:214), which is not installable here, and the reference holds no checkpoint or output vector for
them.  Topology follows the reference call sites:

* dnn       rl4rs/nets/dnn.py:31-37        [mean-pooled category emb ‖ dense tower] -> Dense256 ELU -> Dense256 ELU
                                            ('simulator_obs') -> softmax.  sequence_input_concat is built (dnn.py:33)
                                            but its output is never connected.
* widedeep  rl4rs/nets/widedeep.py:31-38   'simulator_obs' = [Dense256 ELU(mean-pooled sequence embs) ‖ dense tower ‖
                                            Flatten(category emb)] (a Concatenate, no activation) -> softmax
* lstm      rl4rs/nets/lstm.py:31-37       [GRU(seq_i) final states ‖ dense tower ‖ GRU(category emb) final ‖
                                            Flatten(category emb)] -> Dense256 ELU ('simulator_obs') -> softmax
* helpers   rl4rs/nets/utils.py:7-14 (GlobalAveragePooling1D over ALL positions, padding id 0 included — keras
            Embedding without mask_zero), :28-45, :48-54, :57-97

keras ``layers.GRU(units)`` **[from memory]** under TF 1.15 (``tf.keras.layers.GRU`` is the v1 class there):
    z = hard_sigmoid(x Wz + h Uz + bz);  r = hard_sigmoid(x Wr + h Ur + br)
    hh = tanh(x Wh + (r*h) Uh + bh);     h' = z*h + (1-z)*hh          (reset_after=False, one bias vector)
    hard_sigmoid(x) = clip(0.2 x + 0.5, 0, 1);  kernel column order z | r | h;  h0 = 0;  returns the last state.
"""
import numpy as np


def _elu(x):
    return np.where(x > 0, x, np.expm1(np.minimum(x, 0)))


def _softmax(x, axis=-1):
    x = x - x.max(axis=axis, keepdims=True)
    e = np.exp(x)
    return e / e.sum(axis=axis, keepdims=True)


def _hard_sigmoid(x):
    return np.clip(0.2 * x + 0.5, 0.0, 1.0)


def keras_gru_last(X, kernel, recurrent, bias):
    """X [R, L, E] -> last hidden state [R, U]."""
    R, L, _ = X.shape
    U = recurrent.shape[0]
    h = np.zeros((R, U), dtype=X.dtype)
    for t in range(L):
        xp = X[:, t] @ kernel + bias
        hz = h @ recurrent[:, :2 * U]
        z = _hard_sigmoid(xp[:, :U] + hz[:, :U])
        r = _hard_sigmoid(xp[:, U:2 * U] + hz[:, U:])
        hh = np.tanh(xp[:, 2 * U:] + (r * h) @ recurrent[:, 2 * U:])
        h = z * h + (1.0 - z) * hh
    return h


class OracleSimnet(object):
    """Same calling interface as ``OracleDien`` (obs / reward_probs / prob)."""

    def __init__(self, algo, weights, config, dtype=np.float64):
        assert algo in ('dnn', 'widedeep', 'lstm')
        self.algo = algo
        self.dtype = dtype
        self.config = config
        self.w = dict((k, np.asarray(v, dtype=dtype)) for k, v in weights.items())
        self.seq_num = config['seq_num']

    def dense_tower(self, dense):
        w = self.w
        h = _elu(dense.astype(self.dtype) @ w['dense_w1'] + w['dense_b1'])
        return _elu(h @ w['dense_w2'] + w['dense_b2'])

    def obs(self, seq, dense, cat):
        w = self.w
        cat = np.asarray(cat).astype(np.int64)
        seq = np.asarray(seq).astype(np.int64)
        cat_emb = w['cat_emb'][cat]                                  # [R, Cn, E]
        d = self.dense_tower(np.asarray(dense))
        R = cat.shape[0]
        if self.algo == 'dnn':
            feat = np.concatenate([cat_emb.mean(axis=1), d], axis=1)
            a = _elu(feat @ w['fc_w'] + w['fc_b'])
            return _elu(a @ w['obs_w'] + w['obs_b'])
        if self.algo == 'widedeep':
            pooled = np.concatenate([w['seq_emb'][seq[:, i]].mean(axis=1) for i in range(self.seq_num)], axis=1)
            s = _elu(pooled @ w['fc_w'] + w['fc_b'])
            return np.concatenate([s, d, cat_emb.reshape(R, -1)], axis=1)
        finals = [keras_gru_last(w['seq_emb'][seq[:, i]], w['seq%d_gru_kernel' % i], w['seq%d_gru_recurrent' % i],
                                 w['seq%d_gru_bias' % i]) for i in range(self.seq_num)]
        cg = keras_gru_last(cat_emb, w['cat_gru_kernel'], w['cat_gru_recurrent'], w['cat_gru_bias'])
        feat = np.concatenate(finals + [d, cg, cat_emb.reshape(R, -1)], axis=1)
        return _elu(feat @ w['obs_w'] + w['obs_b'])

    def reward_probs(self, seq, dense, cat):
        return _softmax(self.obs(seq, dense, cat) @ self.w['out_w'] + self.w['out_b'])

    def prob(self, seq, dense, cat):
        return self.reward_probs(seq, dense, cat)[:, 1]


def loss_and_grad(algo, weights, dense, cat, labels, seqs=None, mask1=None, mask2=None, rate=0.0, class_num=2):
    """Training-mode forward + keras binary_crossentropy + gradients of the dnn / widedeep / lstm model by torch float64 autograd -
    the checker for the hand-written HIP backward (rl4rs/nets/dnn.py:31-37, widedeep.py:31-38 +
    model.compile(loss='binary_crossentropy'); Dropout(0.2) after each dense-tower layer, utils.py:51,53).
    mask1 / mask2: the keep masks [N, U] the device drew (None = no dropout).  -> mean loss, dict of gradients."""
    import torch
    w = dict((k, torch.tensor(np.asarray(v, dtype=np.float64), requires_grad=True)) for k, v in weights.items())
    elu = torch.nn.functional.elu
    x = torch.tensor(np.asarray(dense, dtype=np.float64))
    ids = torch.tensor(np.asarray(cat, dtype=np.int64))
    h = elu(x @ w['dense_w1'] + w['dense_b1'])
    if mask1 is not None:
        h = h * torch.tensor(np.asarray(mask1, dtype=np.float64)) / (1.0 - rate)
    h = elu(h @ w['dense_w2'] + w['dense_b2'])
    if mask2 is not None:
        h = h * torch.tensor(np.asarray(mask2, dtype=np.float64)) / (1.0 - rate)
    def gru_last(X, i):      # keras GRU (see keras_gru_last above) in torch
        k, r, b = w[i + '_kernel'], w[i + '_recurrent'], w[i + '_bias']
        U = r.shape[0]
        hs = lambda v: torch.clamp(0.2 * v + 0.5, 0.0, 1.0)
        hst = torch.zeros((X.shape[0], U), dtype=torch.float64)
        for t in range(X.shape[1]):
            xp = X[:, t] @ k + b
            hz = hst @ r[:, :2 * U]
            z = hs(xp[:, :U] + hz[:, :U])
            rr = hs(xp[:, U:2 * U] + hz[:, U:])
            hh = torch.tanh(xp[:, 2 * U:] + (rr * hst) @ r[:, 2 * U:])
            hst = z * hst + (1.0 - z) * hh
        return hst

    if algo == 'dnn':
        feat = torch.cat([w['cat_emb'][ids].mean(dim=1), h], dim=1)
        a = elu(feat @ w['fc_w'] + w['fc_b'])
        obs = elu(a @ w['obs_w'] + w['obs_b'])
    elif algo == 'lstm':
        finals = [gru_last(w['seq_emb'][torch.tensor(np.asarray(q, dtype=np.int64))], 'seq%d_gru' % i) for i, q in enumerate(seqs)]
        cg = gru_last(w['cat_emb'][ids], 'cat_gru')
        feat = torch.cat(finals + [h, cg, w['cat_emb'][ids].reshape(ids.shape[0], -1)], dim=1)
        obs = elu(feat @ w['obs_w'] + w['obs_b'])
    else:
        pooled = torch.cat([w['seq_emb'][torch.tensor(np.asarray(q, dtype=np.int64))].mean(dim=1) for q in seqs], dim=1)
        obs = torch.cat([elu(pooled @ w['fc_w'] + w['fc_b']), h, w['cat_emb'][ids].reshape(ids.shape[0], -1)], dim=1)
    p = torch.softmax(obs @ w['out_w'] + w['out_b'], dim=1)
    y = torch.nn.functional.one_hot(torch.tensor(np.asarray(labels, dtype=np.int64)), class_num).double()
    pc = torch.clamp(p, 1e-7, 1.0 - 1e-7)
    loss = (-(y * torch.log(pc) + (1.0 - y) * torch.log(1.0 - pc)).mean(dim=1)).mean()
    loss.backward()
    return float(loss.item()), dict((k, v.grad.numpy()) for k, v in w.items() if v.grad is not None)
