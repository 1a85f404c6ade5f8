"""Offline (batch) RL learners on the device - SURVEY section 8 row f1, BASELINE configs[4].

The reference trains d3rlpy's ``DiscreteBC`` / ``DiscreteBCQ`` / ``DiscreteCQL`` on the logged-policy ``MDPDataset``
(script/batchrl_trainer.py:34-90, :395-...), the first two with its own ``CustomVectorEncoder`` that re-derives the action
mask from the observation tail (rl4rs/nets/cql/encoder.py:9-67).  Here the same networks, losses and update rules run
through ``librl4rs_hip`` (``rl4rs_qnet_*``, ``rl4rs_qloss_*``); this module is the thin loop around them:

* ``transitions_from_mdp`` - d3rlpy 0.91's episode -> transition rule (the reward of an action is stored with the NEXT
  observation, script/batchrl_trainer.py:186-197; the terminal row yields a transition into a zero observation).
* ``DiscreteBC``  - imitator only, ``nll + beta * mean(logits^2)`` (the script passes beta = 0), Adam lr 1e-3.
* ``DiscreteBCQ`` - Q network + target + imitator; next action by the BCQ rule (action_flexibility 0.3), imitator penalty
  beta 0.5, Adam lr 6.25e-5, hard target sync every ``target_update_interval`` (8000) updates.
* ``DiscreteCQL`` - DoubleDQN + ``alpha * (logsumexp Q - Q[a])`` (alpha 1.0) on d3rlpy's default two-layer encoder, as the
  script leaves the custom factory commented out for CQL.

d3rlpy is absent from this image (parity unpinned): the hyper-parameter defaults above are d3rlpy 0.91's as published;
every gradient is checked against torch autograd of the restated model in ``tests/test_gpu_offline_rl.py``.  With
``torch.distributed`` initialised each rank trains on its own minibatches and the flat gradient is mean-all-reduced
before Adam (data parallel, one collective per network per update).
"""
import numpy as np
import torch

from . import device as D
from . import dist as rdist


def transitions_from_mdp(observations, actions, rewards, terminals):
    """(obs, act, next_reward, next_obs, terminal) tensors from MDPDataset-style arrays (rows in time order, an episode
    ends at ``terminals == 1``).  Row t of an episode gives the transition (o_t, a_t, r_{t+1}, o_{t+1}, 0); the terminal
    row gives (o_T, a_T, 0, zeros, 1); a trailing episode without terminal flag drops its last row."""
    obs = torch.as_tensor(observations, dtype=torch.float32)
    act = torch.as_tensor(actions).reshape(obs.shape[0], -1)[:, 0].to(torch.int32)
    rew = torch.as_tensor(rewards, dtype=torch.float32).reshape(-1)
    ter = torch.as_tensor(terminals, dtype=torch.float32).reshape(-1)
    n = obs.shape[0]
    is_end = ter > 0.5
    nxt = torch.roll(obs, -1, dims=0)
    nrew = torch.roll(rew, -1, dims=0)
    nxt = torch.where(is_end[:, None], torch.zeros_like(nxt), nxt)
    nrew = torch.where(is_end, torch.zeros_like(nrew), nrew)
    keep = torch.ones(n, dtype=torch.bool, device=obs.device)
    if n and not bool(is_end[-1]):
        keep[-1] = False                     # the last row has no successor and is not terminal
    return obs[keep], act[keep], nrew[keep], nxt[keep], ter[keep]


def init_qnet_params(obs_dim, action_size, mask_size=0, emb_size=32, hidden1=256, hidden2=256, seed=0):
    """torch's default initialisers (Linear: U(+-1/sqrt(fan_in)) for weight and bias; Embedding: N(0, 1)), stored [in, out]."""
    rs = np.random.RandomState(seed)

    def linear(fan_in, fan_out):
        k = 1.0 / np.sqrt(fan_in)
        return (rs.uniform(-k, k, size=(fan_in, fan_out)).astype(np.float32), rs.uniform(-k, k, size=(fan_out,)).astype(np.float32))

    p = {}
    p['fc1_w'], p['fc1_b'] = linear(obs_dim, hidden1)
    if mask_size > 0:
        p['emb'] = rs.normal(size=(action_size, emb_size)).astype(np.float32)
        p['fc2_w'], p['fc2_b'] = linear(hidden1 + mask_size * emb_size, action_size)
        p['head_w'], p['head_b'] = linear(action_size, action_size)
    else:
        p['fc2_w'], p['fc2_b'] = linear(hidden1, hidden2)
        p['head_w'], p['head_b'] = linear(hidden2, action_size)
    return p


class _Learner(object):
    def __init__(self, config, obs_dim, batch_size, lr, seed, custom_encoder, device=None):
        self.config = config
        self.A = int(config['action_size'])
        self.D = int(obs_dim)
        self.batch_size = int(batch_size)
        self.lr = float(lr)
        self.seed = int(seed)
        self.custom = bool(custom_encoder)
        self.device = device
        self.total_step = 0

    def _net(self, seed):
        kw = {}
        M = 0
        if self.custom:                                   # batchrl_trainer.py:35-41: mask_size = page_items + 1
            from .data import CatalogTables
            M = int(self.config['page_items']) + 1
            if 'location_mask' in self.config and 'special_items' in self.config:
                loc, special = self.config['location_mask'], self.config['special_items']
            else:
                tab = CatalogTables(self.config['iteminfo_file'], self.A, int(self.config.get('action_emb_size', 32)))
                loc, special = tab.location_mask, tab.special_items
            kw = dict(location_mask=loc, special_items=special)
        params = init_qnet_params(self.D, self.A, M, seed=seed)
        return D.DeviceQNet(self.D, self.A, params, mask_size=M, max_rows=self.batch_size, device=self.device, **kw)

    def _apply(self, net):
        if rdist.world_size() > 1:
            g = net.flat_gradient()
            rdist.allreduce_mean_(g)                      # data parallel: the one collective of an update (per network)
            net.set_flat_gradient(g)
        net.adam_step(self.lr)

    def fit(self, transitions, n_steps, shuffle_seed=None):
        """``n_steps`` minibatch updates over ``transitions`` (tuple of tensors from ``transitions_from_mdp``), epoch-wise
        random permutation like d3rlpy's ``fit``; returns the list of losses."""
        obs, act, rew, nxt, ter = [t.to(self.nets[0].device) for t in transitions]
        n = obs.shape[0]
        assert n >= self.batch_size, 'dataset smaller than one minibatch'
        rs = np.random.RandomState(self.seed if shuffle_seed is None else shuffle_seed)
        losses, perm, pos = [], None, n
        for _ in range(n_steps):
            if pos + self.batch_size > n:
                perm = torch.from_numpy(rs.permutation(n)).to(obs.device)
                pos = 0
            idx = perm[pos:pos + self.batch_size]
            pos += self.batch_size
            losses.append(self.update(obs[idx].contiguous(), act[idx].contiguous(), rew[idx].contiguous(),
                                      nxt[idx].contiguous(), ter[idx].contiguous()))
        out = [float(x) for x in torch.stack(losses).cpu()]
        for net in self.nets:
            net.check_status()
        return out

    def close(self):
        for net in self.nets:
            net.close()


class DiscreteBC(_Learner):
    """d3rlpy.algos.DiscreteBC(batch_size=256, beta=0, encoder_factory=CustomVectorEncoderFactory(with_q=True))
    (batchrl_trainer.py:34-47)."""

    def __init__(self, config, obs_dim, batch_size=256, learning_rate=1e-3, beta=0.0, seed=0, device=None):
        super(DiscreteBC, self).__init__(config, obs_dim, batch_size, learning_rate, seed, True, device)
        self.beta = float(beta)
        self.imitator = self._net(seed)
        self.nets = [self.imitator]

    def update(self, obs, act, rew=None, nxt=None, ter=None):
        logits = self.imitator.forward(obs)
        loss2, d = self.imitator.imitation_loss(logits, act, self.beta)
        self.imitator.backward(obs, d)
        self._apply(self.imitator)
        self.total_step += 1
        return loss2[0] + self.beta * loss2[1] / self.A

    def predict(self, obs):
        return self.imitator.best_action(self.imitator.forward(obs))


class _QLearner(_Learner):
    def __init__(self, config, obs_dim, batch_size, lr, seed, custom, gamma, target_update_interval, device):
        super(_QLearner, self).__init__(config, obs_dim, batch_size, lr, seed, custom, device)
        self.gamma = float(gamma)
        self.target_update_interval = int(target_update_interval)
        self.q = self._net(seed)
        self.q_target = self._net(seed)
        self.q_target.copy_from(self.q)

    def predict_value(self, obs, actions):
        q = self.q.forward(obs)
        return q.gather(1, torch.as_tensor(actions, device=q.device).long().reshape(-1, 1))[:, 0]

    def _sync_target(self):
        self.total_step += 1
        if self.total_step % self.target_update_interval == 0:
            self.q_target.copy_from(self.q)


class DiscreteBCQ(_QLearner):
    """d3rlpy.algos.DiscreteBCQ(batch_size=256, encoder_factory=CustomVectorEncoderFactory(with_q=True))
    (batchrl_trainer.py:48-60)."""

    def __init__(self, config, obs_dim, batch_size=256, learning_rate=6.25e-5, gamma=0.99, action_flexibility=0.3, beta=0.5,
                 target_update_interval=8000, seed=0, device=None):
        super(DiscreteBCQ, self).__init__(config, obs_dim, batch_size, learning_rate, seed, True, gamma, target_update_interval, device)
        self.action_flexibility, self.beta = float(action_flexibility), float(beta)
        self.imitator = self._net(seed + 1)
        self.nets = [self.q, self.q_target, self.imitator]

    def update(self, obs, act, rew, nxt, ter):
        # the no-gradient forwards first: a handle keeps the activations of its LAST forward for the backward
        imit_next = self.imitator.forward(nxt)
        q_next = self.q.forward(nxt)
        q_next_t = self.q_target.forward(nxt)
        q_t = self.q.forward(obs)
        td2, dq, _ = self.q.dqn_loss(q_t, act, rew, ter, q_next, q_next_t, imitator_next=imit_next,
                                     action_flexibility=self.action_flexibility, gamma=self.gamma)
        self.q.backward(obs, dq)
        logits = self.imitator.forward(obs)
        im2, dl = self.imitator.imitation_loss(logits, act, self.beta)
        self.imitator.backward(obs, dl)
        self._apply(self.q)
        self._apply(self.imitator)
        self._sync_target()
        return td2[0] + im2[0] + self.beta * im2[1] / self.A

    def predict(self, obs):
        return self.q.best_action(self.q.forward(obs), self.imitator.forward(obs), self.action_flexibility)


class DiscreteCQL(_QLearner):
    """d3rlpy.algos.DiscreteCQL(batch_size=256) on d3rlpy's default encoder (batchrl_trainer.py:74-90)."""

    def __init__(self, config, obs_dim, batch_size=256, learning_rate=6.25e-5, gamma=0.99, alpha=1.0, target_update_interval=8000,
                 seed=0, device=None):
        super(DiscreteCQL, self).__init__(config, obs_dim, batch_size, learning_rate, seed, False, gamma, target_update_interval, device)
        self.alpha = float(alpha)
        self.nets = [self.q, self.q_target]

    def update(self, obs, act, rew, nxt, ter):
        q_next = self.q.forward(nxt)
        q_next_t = self.q_target.forward(nxt)
        q_t = self.q.forward(obs)
        l2, dq, _ = self.q.dqn_loss(q_t, act, rew, ter, q_next, q_next_t, gamma=self.gamma, cql_alpha=self.alpha)
        self.q.backward(obs, dq)
        self._apply(self.q)
        self._sync_target()
        return l2[0] + self.alpha * l2[1]

    def predict(self, obs):
        return self.q.best_action(self.q.forward(obs))
