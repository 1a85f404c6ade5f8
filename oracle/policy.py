"""Policy-net restatement.  ORACLE — test infrastructure only (see oracle/__init__.py).

PARITY UNPINNED: the model is RLlib 1.5.1's ``FullyConnectedNetwork`` + ``ParametricActionsModel`` and the
losses are RLlib's ``a3c_tf_policy`` / ``ppo_tf_policy`` (ray==1.5.1, environment.yml:199, not vendored); only the
masking formula is in-tree (rl4rs/nets/rllib/rllib_mask_model.py:61-62).  Forward in numpy; loss + gradients by
torch float64 autograd (the checker for the hand-written HIP backward)."""
import numpy as np

F32_MIN = float(np.finfo(np.float32).min)


def _split(flat, od, hid, A):
    ae = A + 1
    o = 0
    W1 = flat[o:o + od * hid].reshape(od, hid); o += od * hid
    b1 = flat[o:o + hid]; o += hid
    W2 = flat[o:o + hid * ae].reshape(hid, ae); o += hid * ae
    b2 = flat[o:o + ae]
    return W1, b1, W2, b2


def forward(flat, obs, mask, od=256, hid=64, A=284):
    """-> masked logits [N,A] (float64), value [N].  mask: [N,A] in {0,1} or None."""
    W1, b1, W2, b2 = _split(np.asarray(flat, dtype=np.float64), od, hid, A)
    h = np.tanh(np.asarray(obs, dtype=np.float64) @ W1 + b1)
    out = h @ W2 + b2
    logits = out[:, :A]
    if mask is not None:
        with np.errstate(divide='ignore'):
            logits = logits + np.maximum(np.log(np.asarray(mask, dtype=np.float64)), F32_MIN)   # rllib_mask_model.py:61-62
    return logits, out[:, A]


def log_softmax(l):
    m = l.max(axis=1, keepdims=True)
    return l - (m + np.log(np.exp(l - m).sum(axis=1, keepdims=True)))


def loss_and_grad(algo, flat, obs, mask, actions, adv, ret, old_logp=None, old_value=None, old_logits=None,
                  vf_coeff=0.5, ent_coeff=0.01, clip=0.3, vf_clip=500.0, kl_coeff=0.2, od=256, hid=64, A=284):
    """float64 autograd of the A2C (algo 0) / PPO (algo 1) loss -> (grad flat, stats[4] sums)."""
    import torch
    t = lambda x: torch.as_tensor(np.asarray(x), dtype=torch.float64)
    p = t(flat).clone().requires_grad_(True)
    ae = A + 1
    o = 0
    W1 = p[o:o + od * hid].reshape(od, hid); o += od * hid
    b1 = p[o:o + hid]; o += hid
    W2 = p[o:o + hid * ae].reshape(hid, ae); o += hid * ae
    b2 = p[o:o + ae]
    h = torch.tanh(t(obs) @ W1 + b1)
    out = h @ W2 + b2
    logits, v = out[:, :A], out[:, A]
    if mask is not None:
        logits = logits + torch.clamp(torch.log(t(mask)), min=F32_MIN)
    lsm = torch.log_softmax(logits, dim=1)
    pr = torch.exp(lsm)
    ent = -(torch.where(pr > 0, pr * lsm, torch.zeros_like(pr))).sum(1)
    a = torch.as_tensor(np.asarray(actions), dtype=torch.int64)
    lp = lsm.gather(1, a[:, None])[:, 0]
    adv_t, ret_t = t(adv), t(ret)
    N = lp.shape[0]
    if algo == 0:
        pi = -(lp * adv_t)
        vf = 0.5 * (v - ret_t) ** 2
        kl = torch.zeros_like(lp)
        total = pi.sum() + vf_coeff * vf.sum() - ent_coeff * ent.sum()
    else:
        ratio = torch.exp(lp - t(old_logp))
        pi = -torch.minimum(adv_t * ratio, adv_t * torch.clamp(ratio, 1 - clip, 1 + clip))
        pv = t(old_value)
        l1 = (v - ret_t) ** 2
        vc = pv + torch.clamp(v - pv, -vf_clip, vf_clip)
        vf = torch.maximum(l1, (vc - ret_t) ** 2)
        olsm = torch.log_softmax(t(old_logits), dim=1)
        q = torch.exp(olsm)
        kl = torch.where(q > 0, q * (olsm - lsm), torch.zeros_like(q)).sum(1)
        total = (pi + kl_coeff * kl + vf_coeff * vf - ent_coeff * ent).mean()
    total.backward()
    stats = np.array([pi.sum().item(), vf.sum().item(), ent.sum().item(), kl.sum().item()])
    return p.grad.numpy(), stats


def predict_with_mask(scores, obs, location_mask, special_items, page_items=9):
    """policy_model.predict_with_mask for a d3rlpy policy (rl4rs/policy/policy_model.py:17-41), with ``scores``
    standing in for ``action_probs(obs)``.  obs: [N, 256 + page_items + 1]."""
    obs = np.array(obs)
    action_probs = np.array(scores, dtype=np.float64)
    batch_size = len(obs)
    mask_size = page_items + 1
    prev_actions = obs[:, -mask_size:-1].astype(int)
    cur_step = obs[:, -1].astype(int)
    mask = np.array(location_mask)[(cur_step % page_items // 3).astype(int)]
    for i in range(mask_size - 1):
        mask[range(batch_size), prev_actions[:, i]] = 0
    action_probs[mask < 0.01] = -2 ** 15
    for i in range(batch_size):
        if len(np.intersect1d(prev_actions[i], special_items)) > 0:
            action_probs[i][special_items] = -2 ** 15
    return action_probs.argmax(axis=1)


def rawstate_forward(weights, cat, dense, seqs, mask=None):
    """Raw-state policy encoder (rl4rs/nets/rllib/rllib_rawstate_model.py:49-76; PARITY UNPINNED like the rest of this
    module): context = ELU([mean seq embs | dense tower | mean category emb] @ ctx_w + ctx_b); logits = context @ out_w +
    out_b (+ the mask term of rllib_mask_model.py:61-62); value = context @ value_w + value_b.
    cat [N,Cn] ids, dense [N,Dn], seqs = list of [N,L] ids.  -> masked logits [N,A] float64, value [N]."""
    w = dict((k, np.asarray(v, dtype=np.float64)) for k, v in weights.items())
    elu = lambda x: np.where(x > 0, x, np.expm1(np.minimum(x, 0)))
    seq_feat = [w['seq_emb'][np.asarray(s, dtype=np.int64)].mean(axis=1) for s in seqs]     # utils.py:57-77
    d = elu(elu(np.asarray(dense, dtype=np.float64) @ w['dense_w1'] + w['dense_b1']) @ w['dense_w2'] + w['dense_b2'])
    c = w['cat_emb'][np.asarray(cat, dtype=np.int64)].mean(axis=1)                          # utils.py:7-14
    ctx = elu(np.concatenate(seq_feat + [d, c], axis=1) @ w['ctx_w'] + w['ctx_b'])
    logits = ctx @ w['out_w'] + w['out_b']
    if mask is not None:
        with np.errstate(divide='ignore'):
            logits = logits + np.maximum(np.log(np.asarray(mask, dtype=np.float64)), F32_MIN)
    return logits, (ctx @ w['value_w'] + w['value_b'])[:, 0]


def rawstate_loss_and_grad(algo, weights, cat, dense, seqs, mask, actions, adv, ret, old_logp=None, old_value=None,
                           old_logits=None, vf_coeff=0.5, ent_coeff=0.01, clip=0.3, vf_clip=500.0, kl_coeff=0.2):
    """float64 autograd of RLlib's A2C (algo 0) / PPO (algo 1) loss on the raw-state policy (rawstate_forward above) -> dict of
    gradients (with head_w = [out_w | value_w], head_b = [out_b | value_b]), stats[4] sums."""
    import torch
    t = lambda x: torch.as_tensor(np.asarray(x), dtype=torch.float64)
    w = dict((k, t(v).clone().requires_grad_(True)) for k, v in weights.items())
    elu = torch.nn.functional.elu
    seq_feat = [w['seq_emb'][torch.as_tensor(np.asarray(s), dtype=torch.int64)].mean(dim=1) for s in seqs]
    d = elu(elu(t(dense) @ w['dense_w1'] + w['dense_b1']) @ w['dense_w2'] + w['dense_b2'])
    c = w['cat_emb'][torch.as_tensor(np.asarray(cat), dtype=torch.int64)].mean(dim=1)
    ctx = elu(torch.cat(seq_feat + [d, c], dim=1) @ w['ctx_w'] + w['ctx_b'])
    logits = ctx @ w['out_w'] + w['out_b']
    v = (ctx @ w['value_w'] + w['value_b'])[:, 0]
    if mask is not None:
        logits = logits + torch.clamp(torch.log(t(mask)), min=F32_MIN)
    lsm = torch.log_softmax(logits, dim=1)
    pr = torch.exp(lsm)
    ent = -(torch.where(pr > 0, pr * lsm, torch.zeros_like(pr))).sum(1)
    a = torch.as_tensor(np.asarray(actions), dtype=torch.int64)
    lp = lsm.gather(1, a[:, None])[:, 0]
    adv_t, ret_t = t(adv), t(ret)
    if algo == 0:
        pi = -(lp * adv_t)
        vf = 0.5 * (v - ret_t) ** 2
        kl = torch.zeros_like(lp)
        total = pi.sum() + vf_coeff * vf.sum() - ent_coeff * ent.sum()
    else:
        ratio = torch.exp(lp - t(old_logp))
        pi = -torch.minimum(adv_t * ratio, adv_t * torch.clamp(ratio, 1 - clip, 1 + clip))
        pv = t(old_value)
        l1 = (v - ret_t) ** 2
        vc = pv + torch.clamp(v - pv, -vf_clip, vf_clip)
        vf = torch.maximum(l1, (vc - ret_t) ** 2)
        olsm = torch.log_softmax(t(old_logits), dim=1)
        q = torch.exp(olsm)
        kl = torch.where(q > 0, q * (olsm - lsm), torch.zeros_like(q)).sum(1)
        total = (pi + kl_coeff * kl + vf_coeff * vf - ent_coeff * ent).mean()
    total.backward()
    g = dict((k, x.grad.numpy()) for k, x in w.items())
    g['head_w'] = np.concatenate([g.pop('out_w'), g.pop('value_w')], axis=1)
    g['head_b'] = np.concatenate([g.pop('out_b'), g.pop('value_b')])
    stats = np.array([pi.sum().item(), vf.sum().item(), ent.sum().item(), kl.sum().item()])
    return g, stats


# ---------------------------------------------------------------------------------------------------------------------
# PPO train calls restated in float64 (what rl4rs_amd.train.Trainer must track): RLlib 1.5.1's minibatch SGD over one
# shuffled train batch (script/modelfree_train.py:179-216: sgd_minibatch_size 256, num_sgd_iter 1, lr 1e-4, kl_coeff 0.2,
# kl_target 0.01), tf.train.AdamOptimizer, and KLCoeffMixin.update_kl.  PARITY UNPINNED like the rest of this module
# (ray is absent); the update rule below is RLlib's published one.
def adam_update(flat, m, v, t, grad, lr, beta1=0.9, beta2=0.999, eps=1e-8):
    """One tf.train.AdamOptimizer step in float64 -> (flat, m, v, t)."""
    t = t + 1
    m = beta1 * m + (1.0 - beta1) * grad
    v = beta2 * v + (1.0 - beta2) * grad * grad
    lr_t = lr * np.sqrt(1.0 - beta2 ** t) / (1.0 - beta1 ** t)
    return flat - lr_t * m / (np.sqrt(v) + eps), m, v, t


def update_kl_coeff(kl_coeff, sampled_kl, kl_target):
    """ray/rllib/agents/ppo/ppo_tf_policy.py KLCoeffMixin.update_kl (1.5.1)."""
    if sampled_kl > 2.0 * kl_target:
        kl_coeff *= 1.5
    elif sampled_kl < 0.5 * kl_target:
        kl_coeff *= 0.5
    return kl_coeff


def a2c_train_call(state, batch, lr, mask_fn, od=256, hid=64, A=284, vf_coeff=0.5, ent_coeff=0.01, grad_clip=10.0):
    """One A2C train call (script/modelfree_train.py:248-304: RLlib A2C, ONE gradient over the whole rollout batch, summed
    losses, tf.clip_by_global_norm(40 -> the script's grad_clip 10), AdamOptimizer).  state = (flat, m, v, t) float64; batch =
    dict of numpy arrays obs, act, mask, adv, ret.  -> (state, sums [pi, vf, ent, 0], gradient norm before clipping, clipped
    gradient)."""
    flat, m, v, t = state
    mask = mask_fn(batch['mask']) if batch.get('mask') is not None else None
    g, s = loss_and_grad(0, flat, batch['obs'], mask, batch['act'], batch['adv'], batch['ret'], vf_coeff=vf_coeff, ent_coeff=ent_coeff,
                         od=od, hid=hid, A=A)
    norm = float(np.sqrt((g * g).sum()))
    if grad_clip and norm > grad_clip:
        g = g * (grad_clip / norm)
    return adam_update(flat, m, v, t, g, lr), s, norm, g


def ppo_train_call(state, batch, minibatch, lr, kl_coeff, kl_target, mask_fn, od=256, hid=64, A=284, vf_coeff=0.5, clip=0.3,
                   vf_clip=500.0):
    """One PPO train call on an already shuffled batch.  state = (flat, m, v, t) float64; batch = dict of numpy arrays
    obs, act, mask (dense [N, A] 0/1 or None via mask_fn), adv, ret, logp, val, logits.  Minibatches are consecutive rows,
    the trailing N % minibatch rows are dropped.  -> (state, stats) with stats = dict(kl_mean, kl_coeff (updated),
    last = [pi, vf, ent, kl] MEANS of the last minibatch)."""
    flat, m, v, t = state
    N = batch['obs'].shape[0]
    kl_sum, count, last = 0.0, 0, None
    for lo in range(0, N - minibatch + 1, minibatch):
        hi = lo + minibatch
        mask = mask_fn(batch['mask'][lo:hi]) if batch.get('mask') is not None else None
        g, s = loss_and_grad(1, flat, batch['obs'][lo:hi], mask, batch['act'][lo:hi], batch['adv'][lo:hi], batch['ret'][lo:hi],
                             old_logp=batch['logp'][lo:hi], old_value=batch['val'][lo:hi], old_logits=batch['logits'][lo:hi],
                             vf_coeff=vf_coeff, ent_coeff=0.0, clip=clip, vf_clip=vf_clip, kl_coeff=kl_coeff, od=od, hid=hid, A=A)
        flat, m, v, t = adam_update(flat, m, v, t, g, lr)
        kl_sum += s[3]
        count += minibatch
        last = s / minibatch
    kl_mean = kl_sum / max(count, 1)
    return (flat, m, v, t), dict(kl_mean=kl_mean, kl_coeff=update_kl_coeff(kl_coeff, kl_mean, kl_target), last=last)
