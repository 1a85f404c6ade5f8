"""TEST INFRASTRUCTURE ONLY - CPU restatement (torch float64, autograd) of the reference's offline-RL networks and of the
d3rlpy 0.91 losses they are trained with.  Only tests/ may import this module.

Networks: ``CustomVectorEncoder.forward`` (rl4rs/nets/cql/encoder.py:42-67, with_q=True, hidden_units=[256], relu) and
d3rlpy's plain ``VectorEncoder`` ([256, 256], relu), each followed by the ``nn.Linear(feature_size, action_size)`` head that
d3rlpy's ``DiscreteMeanQFunction`` / ``DiscreteImitator`` put on an encoder.  Losses (d3rlpy 0.91 is absent from this image -
``environment.yml:146`` - so they follow the published algorithms; PARITY UNPINNED):
  DiscreteImitator.compute_error   nll_loss(log_softmax(logits), a) + beta * (logits ** 2).mean()
  DiscreteMeanQFunction.compute_error + DoubleDQN target   huber(r + gamma * Q_targ(s')[a*] * (1 - ter) - Q(s)[a]).mean()
  DiscreteBCQ._predict_best_action   a* = argmax (Q - min Q) * [log pi - max log pi > log(action_flexibility)]
  DiscreteCQL conservative loss      (logsumexp Q(s) - Q(s)[a]).mean()
Parameters use the product's [in, out] storage so the same dict feeds both sides.
"""
import numpy as np
import torch


def mask_from_tail(x, location_mask, special_items, mask_size):
    """encoder.py:44-49,61-66 - returns a bool [B, A] array, True where the encoder keeps the value."""
    x = np.asarray(x)
    B = x.shape[0]
    prev = x[:, -mask_size:-1].astype(np.int64)
    cur = x[:, -1].astype(np.int64)
    layer = cur % 9 // 3
    mask = np.asarray(location_mask)[layer].copy()
    for i in range(mask_size - 1):
        mask[np.arange(B), prev[:, i]] = 0
    keep = ~(mask < 0.01)
    for i in range(B):
        if len(np.intersect1d(prev[i], special_items)) > 0:
            keep[i, special_items] = False
    return keep


class OracleQNet(object):
    def __init__(self, params, mask_size=0, location_mask=None, special_items=None, dtype=torch.float64):
        self.p = dict((k, torch.tensor(np.asarray(v), dtype=dtype, requires_grad=True)) for k, v in params.items())
        self.M = int(mask_size)
        self.location_mask = location_mask
        self.special_items = list(special_items) if special_items is not None else None
        self.dtype = dtype

    def forward(self, x):
        p = self.p
        xt = torch.as_tensor(np.asarray(x), dtype=self.dtype)
        h = torch.relu(xt @ p['fc1_w'] + p['fc1_b'])
        if self.M > 0:
            ids = torch.as_tensor(np.asarray(x)[:, -self.M:].astype(np.int64))
            emb = p['emb'][ids].reshape(xt.shape[0], -1)
            enc = torch.cat([h, emb], dim=1) @ p['fc2_w'] + p['fc2_b']
            keep = torch.as_tensor(mask_from_tail(x, self.location_mask, self.special_items, self.M))
            enc = torch.where(keep, enc, torch.zeros_like(enc))          # h[action_mask] = 0: no gradient through masked entries
        else:
            enc = torch.relu(h @ p['fc2_w'] + p['fc2_b'])
        return enc @ p['head_w'] + p['head_b']

    def grads(self):
        return dict((k, (v.grad.numpy() if v.grad is not None else np.zeros(tuple(v.shape)))) for k, v in self.p.items())

    def zero_grad(self):
        for v in self.p.values():
            v.grad = None


def imitation_loss(logits, actions, beta):
    logp = torch.log_softmax(logits, dim=1)
    a = torch.as_tensor(np.asarray(actions), dtype=torch.int64)
    return torch.nn.functional.nll_loss(logp, a) + beta * (logits ** 2).mean()


def best_action(q, imitator_logits=None, action_flexibility=0.3):
    q = q.detach()
    if imitator_logits is None:
        return q.argmax(dim=1)
    logp = torch.log_softmax(imitator_logits.detach(), dim=1)
    ratio = logp - logp.max(dim=1, keepdim=True).values
    mask = (ratio > np.log(action_flexibility)).to(q.dtype)
    value = q - q.min(dim=1, keepdim=True).values
    return (value * mask).argmax(dim=1)


def huber(y, target, beta=1.0):
    diff = target - y
    cond = diff.detach().abs() < beta
    return torch.where(cond, 0.5 * diff ** 2, beta * (diff.abs() - 0.5 * beta))


def dqn_loss(q_t, actions, rewards, terminals, q_next, q_next_target, imitator_next=None, action_flexibility=0.3, gamma=0.99,
             cql_alpha=0.0):
    """returns (td, conservative, best next action); total = td + cql_alpha * conservative."""
    a = torch.as_tensor(np.asarray(actions), dtype=torch.int64)
    best = best_action(q_next, imitator_next, action_flexibility)
    q_tp1 = q_next_target.detach().gather(1, best[:, None])
    r = torch.as_tensor(np.asarray(rewards), dtype=q_t.dtype)[:, None]
    ter = torch.as_tensor(np.asarray(terminals), dtype=q_t.dtype)[:, None]
    y = r + gamma * q_tp1 * (1 - ter)
    qa = q_t.gather(1, a[:, None])
    td = huber(qa, y).mean()
    cons = (torch.logsumexp(q_t, dim=1, keepdim=True) - qa).mean()
    return td, cons, best


def torch_adam(params, grads, m, v, t, lr, beta1=0.9, beta2=0.999, eps=1e-8):
    """torch.optim.Adam step in float64 numpy (t = 1 for the first step)."""
    out = {}
    for k in params:
        m[k] = beta1 * m[k] + (1 - beta1) * grads[k]
        v[k] = beta2 * v[k] + (1 - beta2) * grads[k] ** 2
        mh = m[k] / (1 - beta1 ** t)
        vh = v[k] / (1 - beta2 ** t)
        out[k] = params[k] - lr * mh / (np.sqrt(vh) + eps)
    return out
