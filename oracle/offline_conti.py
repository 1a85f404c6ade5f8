"""TEST INFRASTRUCTURE ONLY - CPU restatement (torch float64, autograd) of the CONTINUOUS-action offline learners the
reference instantiates as 'BCQ-conti' / 'CQL-conti' (script/batchrl_trainer.py:61-73, :91-107): d3rlpy.algos.BCQ / CQL on
d3rlpy's default encoders.  Only tests/ may import this module.

PARITY UNPINNED: d3rlpy 0.91 (environment.yml:146) is a third-party dependency absent from this image and the reference holds no
vector for these learners; the algorithms below restate d3rlpy 0.91 as published -
  d3rlpy/models/torch/encoders.py   VectorEncoderWithAction: relu MLP [256, 256] on torch.cat([x, action], dim=1)
  d3rlpy/models/torch/imitators.py  ConditionalVAE: encode -> Normal(mu, exp(clamp(logstd, -20, 2))), decode -> tanh(fc(h)),
                                    compute_error = mse(decode(x, rsample), a) + beta * kl_divergence(dist, N(0, 1)).mean()
  d3rlpy/models/torch/policies.py   DeterministicResidualPolicy: (a + scale * tanh(fc(h))).clamp(-1, 1)
  d3rlpy/models/torch/q_functions   ContinuousMeanQFunction (mse), compute_max_with_n_actions (lam-weighted min / max)
  d3rlpy/algos/torch/bcq_impl.py    BCQImpl: latents clamp(randn, -0.5, 0.5); target over n sampled actions through the TARGET
                                    policy and TARGET critics; actor loss -Q_1(s, pi(s, decode(s, z))).mean()
Parameters use the product's [in, out] storage (fc1_w rows: observation first, then action) so the same dict feeds both sides.
"""
import numpy as np
import torch


class OracleAMLP(object):
    def __init__(self, params, head_act='none', dtype=torch.float64):
        self.p = dict((k, torch.tensor(np.asarray(v), dtype=dtype, requires_grad=True)) for k, v in params.items())
        self.head_act = head_act
        self.dtype = dtype

    def __call__(self, x, a=None):
        p = self.p
        x = torch.as_tensor(x, dtype=self.dtype)
        if a is not None:
            x = torch.cat([x, torch.as_tensor(a, dtype=self.dtype)], dim=1)
        h = torch.relu(x @ p['fc1_w'] + p['fc1_b'])
        h = torch.relu(h @ p['fc2_w'] + p['fc2_b'])
        out = h @ p['head_w'] + p['head_b']
        return torch.tanh(out) if self.head_act == 'tanh' else out

    def grads(self):
        return dict((k, (v.grad.numpy() if v.grad is not None else np.zeros(tuple(v.shape)))) for k, v in self.p.items())

    def zero_grad(self):
        for v in self.p.values():
            v.grad = None

    def numpy_params(self):
        return dict((k, v.detach().numpy().copy()) for k, v in self.p.items())


def _t(x):
    return torch.as_tensor(np.asarray(x), dtype=torch.float64) if not isinstance(x, torch.Tensor) else x.to(torch.float64)


def cvae_loss(enc, dec, obs, act, eps, beta, min_logstd=-20.0, max_logstd=2.0):
    """ConditionalVAE.compute_error with the rsample noise supplied."""
    obs, act, eps = _t(obs), _t(act), _t(eps)
    L = eps.shape[1]
    e = enc(obs, act)
    mu, logstd = e[:, :L], e[:, L:].clamp(min_logstd, max_logstd)
    dist = torch.distributions.Normal(mu, logstd.exp())
    kl = torch.distributions.kl.kl_divergence(dist, torch.distributions.Normal(0.0, 1.0)).mean()
    y = dec(obs, mu + logstd.exp() * eps)
    return torch.nn.functional.mse_loss(y, act) + beta * kl


def residual_policy(policy, obs, action, scale):
    return (action + scale * policy(obs, action)).clamp(-1.0, 1.0)


def sample_actions(dec, policy, obs, z, n, scale):
    """BCQImpl._sample_repeated_action: [B * n, E] for latents z [B * n, L] (row b * n + j = j-th sample of observation b)."""
    obs = _t(obs)
    rep = obs.repeat_interleave(n, dim=0)
    sampled = dec(rep, _t(z).clamp(-0.5, 0.5))
    return rep, residual_policy(policy, rep, sampled, scale)


def bcq_target(dec, policy_targ, q_targs, nxt, z, n, scale, lam, rewards, terminals, gamma):
    with torch.no_grad():
        rep, a = sample_actions(dec, policy_targ, nxt, z, n, scale)
        vals = torch.stack([q(rep, a).reshape(-1, n) for q in q_targs])           # [critics, B, n]
        mix = (1.0 - lam) * vals.max(dim=0).values + lam * vals.min(dim=0).values
        v = mix.max(dim=1).values
        return _t(rewards) + gamma * v * (1.0 - _t(terminals))


def critic_loss(qs, obs, act, y):
    """EnsembleContinuousQFunction.compute_error: sum over critics of the mean squared error."""
    return sum(((q(_t(obs), _t(act))[:, 0] - y) ** 2).mean() for q in qs)


def actor_loss(dec, policy, q1, obs, z, scale):
    obs = _t(obs)
    with torch.no_grad():
        sampled = dec(obs, _t(z).clamp(-0.5, 0.5))
    a = residual_policy(policy, obs, sampled, scale)
    return -q1(obs, a).mean()


def predict_best_action(dec, policy, q1, obs, z, n, scale):
    with torch.no_grad():
        rep, a = sample_actions(dec, policy, obs, z, n, scale)
        v = q1(rep, a).reshape(-1, n)
        idx = v.argmax(dim=1)
        return a.reshape(-1, n, a.shape[1])[torch.arange(v.shape[0]), idx], idx, v


def soft_sync(targ, src, tau):
    return dict((k, (1.0 - tau) * targ[k] + tau * src[k]) for k in targ)


# ---- continuous CQL (d3rlpy.algos.CQL = SAC + conservative critic loss), d3rlpy 0.91 as published; PARITY UNPINNED --------------
#   d3rlpy/models/torch/policies.py   SquashedNormalPolicy: Normal(mu, exp(clamp(logstd, -20, 2))), _squash_action:
#                                     tanh(u), log_prob = sum(dist.log_prob(u) - 2 (log 2 - u - softplus(-2u)))
#   d3rlpy/algos/torch/sac_impl.py    actor (exp(log_temp) * logp - min_c Q_c(s, a)).mean(); temp -(exp(log_temp) * (logp - A)).mean()
#   d3rlpy/algos/torch/cql_impl.py    conservative loss over [pi(s) samples | pi(s') samples | uniform], deterministic target

def squashed_sample(policy, obs, eps, min_logstd=-20.0, max_logstd=2.0):
    """(tanh(u), log-prob [rows]) for eps [rows, A]; rows = n_obs * rep with the observation of row i = i // rep."""
    head = policy(_t(obs))
    A = head.shape[1] // 2
    eps = _t(eps)
    rep = eps.shape[0] // head.shape[0]
    mu = head[:, :A].repeat_interleave(rep, dim=0)
    logstd = head[:, A:].clamp(min_logstd, max_logstd).repeat_interleave(rep, dim=0)
    dist = torch.distributions.Normal(mu, logstd.exp())
    u = mu + logstd.exp() * eps
    jacob = 2.0 * (np.log(2.0) - u - torch.nn.functional.softplus(-2.0 * u))
    return torch.tanh(u), (dist.log_prob(u) - jacob).sum(dim=1)


def best_action(policy, obs):
    head = policy(_t(obs))
    return torch.tanh(head[:, :head.shape[1] // 2])


def temp_loss(policy, log_temp, obs, eps):
    with torch.no_grad():
        _, logp = squashed_sample(policy, obs, eps)
        targ = logp - eps.shape[1]
    return -(log_temp.exp() * targ).mean()


def conservative_loss(policy, qs, log_alpha, obs, act, nxt, eps_t, eps_tp1, uniform, n, weight, threshold):
    """CQLImpl._compute_conservative_loss: clipped_alpha * (weight * (logsumexp.mean - data.mean) - threshold)"""
    obs, act = _t(obs), _t(act)
    B, A = act.shape
    with torch.no_grad():
        a_t, lp_t = squashed_sample(policy, obs, eps_t)
        a_tp1, lp_tp1 = squashed_sample(policy, nxt, eps_tp1)
    rep = obs.repeat_interleave(n, dim=0)
    uni = _t(uniform).reshape(B * n, A)
    vals = []
    for q in qs:
        v = torch.cat([(q(rep, a_t)[:, 0] - lp_t).reshape(B, n), (q(rep, a_tp1)[:, 0] - lp_tp1).reshape(B, n),
                       (q(rep, uni)[:, 0] - np.log(0.5 ** A)).reshape(B, n)], dim=1)
        vals.append(v)
    lse = torch.stack([torch.logsumexp(v, dim=1) for v in vals])               # [critics, B]
    data = torch.stack([q(obs, act)[:, 0] for q in qs])
    scaled = weight * (lse.mean(dim=0).mean() - data.mean(dim=0).mean())
    return log_alpha.exp().clamp(0, 1e6) * (scaled - threshold)


def cql_target(policy, q_targs, nxt, rewards, terminals, gamma):
    with torch.no_grad():
        a = best_action(policy, nxt)
        v = torch.stack([q(_t(nxt), a)[:, 0] for q in q_targs]).min(dim=0).values
        return _t(rewards) + gamma * v * (1.0 - _t(terminals))


def sac_actor_loss(policy, qs, log_temp, obs, eps):
    a, logp = squashed_sample(policy, obs, eps)
    qmin = torch.stack([q(_t(obs), a)[:, 0] for q in qs]).min(dim=0).values
    return (log_temp.detach().exp() * logp - qmin).mean()
