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
