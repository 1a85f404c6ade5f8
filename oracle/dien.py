"""DIEN scorer restatement (numpy).  ORACLE — test infrastructure only (see oracle/__init__.py).

PARITY UNPINNED: the reference delegates this arithmetic to deepctr==0.9.0 (environment.yml:147) on
tensorflow-gpu==1.15.0 (environment.yml:214); neither is vendored / installable here and the reference
holds no checkpoint or output vector for it.  Topology is taken from the reference call sites; the
cell equations are the published ones of the pinned libraries:

* topology                     rl4rs/nets/dien.py:8-45
* category self-attention      rl4rs/nets/utils.py:16-25   (keras Attention(): softmax(Q K^T) V, no scale)
* dense tower                  rl4rs/nets/utils.py:48-54   (Dense+ELU x2; Dropout inactive at inference)
* sequence branch              rl4rs/nets/utils.py:100-129
    - TF1.15 ``GRUCell``:  [r,u] = sigmoid([x,h] Wg + bg); c = tanh([x, r*h] Wc + bc);
                           h' = u*h + (1-u)*c
    - deepctr ``LocalActivationUnit`` (att_hidden_units=(64,16), sigmoid, no BN/dropout) on
      [q, k, q-k, q*k] -> linear score; ``AttentionSequencePoolingLayer(return_score=True,
      weight_normalization=False)`` => raw scores, key mask all-true because sequence_length == maxlen
      (utils.py:111)
    - deepctr ``VecAttGRUCell`` (AUGRU): as GRUCell but u <- (1 - a_t) * u before the update
* head                         rl4rs/nets/dien.py:34-36
"""
import numpy as np


def _sigmoid(x):
    return 1.0 / (1.0 + np.exp(-x))


def _elu(x):
    return np.where(x > 0, x, np.expm1(np.minimum(x, 0)))


def _softmax(x, axis=-1):
    x = x - x.max(axis=axis, keepdims=True)
    e = np.exp(x)
    return e / e.sum(axis=axis, keepdims=True)


class OracleDien(object):
    def __init__(self, weights, config, dtype=np.float64):
        self.dtype = dtype
        self.config = config
        self.w = dict((k, np.asarray(v, dtype=dtype)) for k, v in weights.items())
        self.seq_num = config['seq_num']

    # ------------------------------------------------------------ pieces
    def gru(self, X, i):
        """X [R,L,E] -> all hidden states [R,L,E]."""
        w = self.w
        Wg, bg = w['gru%d_gate_w' % i], w['gru%d_gate_b' % i]
        Wc, bc = w['gru%d_cand_w' % i], w['gru%d_cand_b' % i]
        R, L, E = X.shape
        h = np.zeros((R, E), dtype=self.dtype)
        out = np.empty((R, L, E), dtype=self.dtype)
        for t in range(L):
            x = X[:, t]
            g = _sigmoid(np.concatenate([x, h], axis=1) @ Wg + bg)
            r, u = g[:, :E], g[:, E:]
            c = np.tanh(np.concatenate([x, r * h], axis=1) @ Wc + bc)
            h = u * h + (1.0 - u) * c
            out[:, t] = h
        return out

    def att_scores(self, q, keys, i):
        """q [R,E], keys [R,L,E] -> raw scores [R,L]."""
        w = self.w
        qq = np.broadcast_to(q[:, None, :], keys.shape)
        a = np.concatenate([qq, keys, qq - keys, qq * keys], axis=-1)
        h1 = _sigmoid(a @ w['att%d_w1' % i] + w['att%d_b1' % i])
        h2 = _sigmoid(h1 @ w['att%d_w2' % i] + w['att%d_b2' % i])
        return (h2 @ w['att%d_w3' % i] + w['att%d_b3' % i])[..., 0]

    def augru(self, X, att, i):
        """X [R,L,E], att [R,L] -> final state [R,2E]."""
        w = self.w
        Wg, bg = w['augru%d_gate_w' % i], w['augru%d_gate_b' % i]
        Wc, bc = w['augru%d_cand_w' % i], w['augru%d_cand_b' % i]
        R, L, E = X.shape
        N = Wc.shape[1]
        h = np.zeros((R, N), dtype=self.dtype)
        for t in range(L):
            x = X[:, t]
            g = _sigmoid(np.concatenate([x, h], axis=1) @ Wg + bg)
            r, u = g[:, :N], g[:, N:]
            c = np.tanh(np.concatenate([x, r * h], axis=1) @ Wc + bc)
            u = (1.0 - att[:, t:t + 1]) * u
            h = u * h + (1.0 - u) * c
        return h

    # ------------------------------------------------------------- model
    def features(self, seq, dense, cat, return_parts=False):
        w = self.w
        seq = np.asarray(seq).astype(np.int64)
        cat = np.asarray(cat).astype(np.int64)
        dense = np.asarray(dense, dtype=self.dtype)
        R = cat.shape[0]
        # category branch (utils.py:16-25)
        E = w['cat_emb'][cat]                                   # [R,Cn,E]
        att = _softmax(E @ E.transpose(0, 2, 1), axis=-1) @ E
        c = np.concatenate([att.mean(axis=1), E.reshape(R, -1)], axis=1)
        # dense tower (utils.py:48-54)
        d = _elu(_elu(dense @ w['dense_w1'] + w['dense_b1']) @ w['dense_w2'] + w['dense_b2'])
        # sequence branch (dien.py:29-32, utils.py:100-129)
        q = w['seq_emb'][cat[:, -10:]].mean(axis=1)             # [R,E]
        finals, parts = [], {}
        for i in range(self.seq_num):
            X = w['seq_emb'][seq[:, i, :]]
            H1 = self.gru(X, i)
            s = self.att_scores(q, H1, i)
            h2 = self.augru(H1, s, i)
            finals.append(h2)
            if return_parts:
                parts['h1_%d' % i], parts['score_%d' % i], parts['h2_%d' % i] = H1, s, h2
        allf = np.concatenate(finals + [d, c], axis=1)          # dien.py:34
        if return_parts:
            parts.update(cat_feat=c, dense_feat=d, query=q, all=allf)
            return allf, parts
        return allf

    def obs(self, seq, dense, cat):
        """'simulator_obs' activations [R,256] (dien.py:35)."""
        allf = self.features(seq, dense, cat)
        return _elu(allf @ self.w['obs_w'] + self.w['obs_b'])

    def reward_probs(self, seq, dense, cat):
        """'simulator_reward' softmax [R,class_num] (dien.py:36)."""
        return _softmax(self.obs(seq, dense, cat) @ self.w['out_w'] + self.w['out_b'])

    def prob(self, seq, dense, cat):
        """res[:,1] as float32, the dtype keras would hand back (slate.py:298)."""
        return self.reward_probs(seq, dense, cat)[:, 1].astype(np.float32)


def loss_and_grad(weights, config, seq, dense, cat, labels, mask1=None, mask2=None, rate=0.0):
    """Training-mode DIEN forward + keras binary_crossentropy + gradients by torch float64 autograd: the checker for the
    hand-written HIP backward (rl4rs_dientrain_*).  Same restatement as OracleDien above (PARITY UNPINNED, see the module
    header) with Dropout(rate) after each dense-tower layer (utils.py:51,53) applied through the keep masks the device drew.
    seq [N, S, L] ids.  -> mean loss, dict of gradients."""
    import torch
    w = dict((k, torch.tensor(np.asarray(v, dtype=np.float64), requires_grad=True)) for k, v in weights.items())
    S, K = config['seq_num'], config['class_num']
    elu = torch.nn.functional.elu
    sig = torch.sigmoid
    ids = torch.tensor(np.asarray(cat, dtype=np.int64))
    sq = torch.tensor(np.asarray(seq, dtype=np.int64))
    x = torch.tensor(np.asarray(dense, dtype=np.float64))
    N = ids.shape[0]
    Ec = w['cat_emb'][ids]
    att = torch.softmax(Ec @ Ec.transpose(1, 2), dim=-1) @ Ec
    c = torch.cat([att.mean(dim=1), Ec.reshape(N, -1)], dim=1)
    h = elu(x @ w['dense_w1'] + w['dense_b1'])
    if mask1 is not None:
        h = h * torch.tensor(np.asarray(mask1, dtype=np.float64)) / (1.0 - rate)
    h = elu(h @ w['dense_w2'] + w['dense_b2'])
    if mask2 is not None:
        h = h * torch.tensor(np.asarray(mask2, dtype=np.float64)) / (1.0 - rate)
    q = w['seq_emb'][ids[:, -10:]].mean(dim=1)

    def cell(xt, hs, Wg, bg, Wc, bc, a=None):
        n = Wc.shape[1]
        g = sig(torch.cat([xt, hs], dim=1) @ Wg + bg)
        r, u = g[:, :n], g[:, n:]
        cc = torch.tanh(torch.cat([xt, r * hs], dim=1) @ Wc + bc)
        if a is not None:
            u = (1.0 - a) * u
        return u * hs + (1.0 - u) * cc

    finals = []
    for i in range(S):
        X = w['seq_emb'][sq[:, i, :]]
        L, E = X.shape[1], X.shape[2]
        hs = torch.zeros((N, E), dtype=torch.float64)
        keys = []
        for t in range(L):
            hs = cell(X[:, t], hs, w['gru%d_gate_w' % i], w['gru%d_gate_b' % i], w['gru%d_cand_w' % i], w['gru%d_cand_b' % i])
            keys.append(hs)
        Kk = torch.stack(keys, dim=1)
        qq = q[:, None, :].expand_as(Kk)
        a = torch.cat([qq, Kk, qq - Kk, qq * Kk], dim=-1)
        s1 = sig(a @ w['att%d_w1' % i] + w['att%d_b1' % i])
        s2 = sig(s1 @ w['att%d_w2' % i] + w['att%d_b2' % i])
        score = (s2 @ w['att%d_w3' % i] + w['att%d_b3' % i])[..., 0]
        h2 = torch.zeros((N, 2 * E), dtype=torch.float64)
        for t in range(L):
            h2 = cell(Kk[:, t], h2, w['augru%d_gate_w' % i], w['augru%d_gate_b' % i], w['augru%d_cand_w' % i],
                      w['augru%d_cand_b' % i], score[:, t:t + 1])
        finals.append(h2)
    allf = torch.cat(finals + [h, c], dim=1)
    obs = elu(allf @ w['obs_w'] + w['obs_b'])
    p = torch.softmax(obs @ w['out_w'] + w['out_b'], dim=1)
    y = torch.nn.functional.one_hot(torch.tensor(np.asarray(labels, dtype=np.int64)), K).double()
    pc = torch.clamp(p, 1e-7, 1.0 - 1e-7)
    loss = (-(y * torch.log(pc) + (1.0 - y) * torch.log(1.0 - pc)).mean(dim=1)).mean()
    loss.backward()
    return float(loss.item()), dict((k, v.grad.numpy()) for k, v in w.items() if v.grad is not None)
