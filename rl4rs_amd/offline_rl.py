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
* ``BCQ`` - the CONTINUOUS-action learner of ``'BCQ-conti'`` (script/batchrl_trainer.py:61-73, ``d3rlpy.algos.BCQ(batch_size=256)``
  on default encoders): what BASELINE configs[4] is quoted on.  Trains on the dataset ``generate_offline_dataset`` emits for
  ``support_conti_env`` (actions = 32-d item embeddings, :220-270): conditional-VAE imitator, residual perturbation actor, twin
  critics with the lam-weighted min/max target over 100 sampled actions, soft target updates; ``predict`` returns the embedding
  the env's K-NN resolves (``env.step(policy.predict(obs))``, batchrl_trainer.py:398-399 with ``predict_with_mask`` ->
  ``predict`` for continuous envs, rl4rs/policy/policy_model.py:17-19).
* ``CQL`` - the continuous-action learner of ``'CQL-conti'`` (script/batchrl_trainer.py:91-107, ``d3rlpy.algos.CQL(batch_size=256,
  gamma=1.0, reward_scaler='standard')``; the ``alpha=`` keyword the script passes is not a parameter of the continuous CQL and
  is swallowed by d3rlpy's ``**kwargs``): SAC (squashed-Gaussian actor, learned temperature) with the conservative critic term
  over 10 policy samples at s, 10 at s' and 10 uniform actions per row, learned log-alpha with threshold 10, weight 5.

d3rlpy is absent from this image (parity unpinned): the hyper-parameter defaults above are d3rlpy 0.91's as published;
every gradient is checked against torch autograd of the restated model in ``tests/test_gpu_offline_rl.py``.  With
``torch.distributed`` initialised each rank trains on its own minibatches and the flat gradient is mean-all-reduced
before Adam (data parallel, one collective per network per update).
"""
import ctypes as C

import numpy as np
import torch

from . import device as D
from . import device as D_
from . import dist as rdist


def transitions_from_mdp(observations, actions, rewards, terminals, discrete_action=True):
    """(obs, act, next_reward, next_obs, terminal) tensors from MDPDataset-style arrays (rows in time order, an episode
    ends at ``terminals == 1``).  Row t of an episode gives the transition (o_t, a_t, r_{t+1}, o_{t+1}, 0); the terminal
    row gives (o_T, a_T, 0, zeros, 1); a trailing episode without terminal flag drops its last row.
    ``discrete_action=False`` (``MDPDataset(..., discrete_action=False)``, batchrl_trainer.py:266): actions stay float [N, E]."""
    obs = torch.as_tensor(observations, dtype=torch.float32)
    if discrete_action:
        act = torch.as_tensor(actions).reshape(obs.shape[0], -1)[:, 0].to(torch.int32)
    else:
        act = torch.as_tensor(actions, dtype=torch.float32).reshape(obs.shape[0], -1)
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



class _ModelIO(object):
    """``save_model`` / ``load_model`` / ``fit_mdp`` shared by every learner of this module: what ``script/batchrl_train.py`` calls
    on a d3rlpy algo (``model.fit(dataset, n_epochs=...)`` then ``model.save_model(path)``, :127-132; ``model.load_model(path)``
    in the eval / ope stages, :140-142).  The file is this package's own (``numpy.savez``: every network's flat parameter vector,
    Adam moments and step count, the learned scalars, the update counter) - d3rlpy's torch ``state_dict`` pickles need d3rlpy."""

    def _io_nets(self):
        seen, out = set(), []
        for name, v in sorted(vars(self).items()):
            if hasattr(v, 'flat_params') and hasattr(v, 'set_flat_params') and id(v) not in seen:
                seen.add(id(v))
                out.append((name, v))
        return out

    def save_model(self, fname):
        if hasattr(self, 'check_status'):
            self.check_status()
        blob = {'__class__': np.array(type(self).__name__), 'total_step': np.array(self.total_step, dtype=np.int64)}
        for name, net in self._io_nets():
            blob['net.' + name] = net.flat_params().cpu().numpy()
            m, v, t = net.adam_state()
            blob['adam_m.' + name], blob['adam_v.' + name] = m.cpu().numpy(), v.cpu().numpy()
            blob['adam_t.' + name] = np.array(t, dtype=np.int64)
        for name in ('log_temp', 'log_alpha'):
            sp = getattr(self, name, None)
            if sp is not None:
                blob['scalar.' + name] = np.concatenate([sp.p.cpu().numpy(), sp.m.cpu().numpy(), sp.v.cpu().numpy(), [float(sp.t)]])
        rs = getattr(self, 'reward_scaler', None)
        if isinstance(rs, str):
            # a scaler given by name ('standard', as batchrl_trainer builds CQL-conti) has no statistics until fit_mdp fits it on the
            # dataset: nothing to save - a learner that never trained on scaled rewards is saved as one without a scaler
            rs = None
        if rs is not None:
            blob['reward_scaler'] = np.array([rs.mean, rs.std, rs.eps], dtype=np.float64)
        with open(fname, 'wb') as f:
            np.savez(f, **blob)

    def load_model(self, fname, legacy=True):
        with np.load(fname) as z:
            cls = str(z['__class__'])
            if cls != type(self).__name__:
                raise ValueError('%s holds a %s, this learner is a %s' % (fname, cls, type(self).__name__))
            self.total_step = int(z['total_step'])
            for name, net in self._io_nets():
                flat = z['net.' + name]
                if flat.size != net.n_params:
                    raise ValueError('%s: network %r has %d parameters in the file, %d here' % (fname, name, flat.size, net.n_params))
                dev = net.device
                net.set_flat_params(torch.from_numpy(np.ascontiguousarray(flat, dtype=np.float32)).to(dev))
                net.set_adam_state(torch.from_numpy(z['adam_m.' + name]).to(dev), torch.from_numpy(z['adam_v.' + name]).to(dev),
                                   int(z['adam_t.' + name]))
            for name in ('log_temp', 'log_alpha'):
                sp = getattr(self, name, None)
                if sp is not None:
                    a = z['scalar.' + name]
                    sp.p.fill_(float(a[0])); sp.m.fill_(float(a[1])); sp.v.fill_(float(a[2])); sp.t = int(a[3])
            # the reward scale the critics were trained on travels with them (d3rlpy keeps it in params.json): a learner restored
            # onto a different scale would keep training / report predict_value in other units without any error
            if 'reward_scaler' in z.files:
                if not hasattr(self, 'reward_scaler'):
                    raise ValueError('%s carries a reward scaler, %s has none' % (fname, type(self).__name__))
                self.reward_scaler = StandardRewardScaler.from_stats(*[float(x) for x in z['reward_scaler']])
            elif getattr(self, 'reward_scaler', None) is not None and not isinstance(self.reward_scaler, str):
                # files written before the scaler travelled with the model (or by a learner whose named scaler was never fitted): the
                # learner keeps ITS scaler - loud, not fatal (legacy=False turns it back into an error)
                if not legacy:
                    raise ValueError('%s was saved without a reward scaler, this learner has one' % fname)
                import warnings
                warnings.warn('%s was saved without a reward scaler; keeping this learner\'s (mean %.6g, std %.6g) - make sure it is the '
                              'scale the file was trained on' % (fname, self.reward_scaler.mean, self.reward_scaler.std))

    def fit_mdp(self, data, n_epochs=1, discrete_action=None, **kw):
        """``fit`` on MDPDataset-style arrays (the dict ``offline.generate_offline_dataset`` returns) for ``n_epochs`` passes over its
        transitions - d3rlpy's ``fit(dataset, n_epochs=...)``."""
        if discrete_action is None:
            discrete_action = isinstance(self, _Learner)
        tr = transitions_from_mdp(data['observations'], data['actions'], data['rewards'], data['terminals'], discrete_action=discrete_action)
        if getattr(self, 'reward_scaler', None) == 'standard':
            # d3rlpy: a scaler given by name is fitted on the dataset when fit() builds the algorithm
            self.reward_scaler = StandardRewardScaler(tr[2])
        steps = int(n_epochs) * (tr[0].shape[0] // self.batch_size)
        return self.fit(tr, steps, **kw)


class _Learner(_ModelIO):
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
        if rdist.collectives_active():
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
        losses, ep, pos = [], None, n
        for _ in range(n_steps):
            if pos + self.batch_size > n:
                # the epoch's permutation is applied to the five arrays ONCE: minibatches are then contiguous row ranges (views),
                # not five gathers per update
                perm = torch.from_numpy(rs.permutation(n)).to(obs.device)
                ep = [t[perm] for t in (obs, act, rew, nxt, ter)]
                pos = 0
            lo, hi = pos, pos + self.batch_size
            pos = hi
            losses.append(self.update(*[t[lo:hi] for t in ep]))
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


def init_amlp_params(obs_dim, act_dim, out_dim, hidden1=256, hidden2=256, seed=0, heads=1):
    """torch's default Linear initialiser (U(+-1/sqrt(fan_in)) for weight and bias), stored [in, out]; ``heads`` Linear layers
    of ``out_dim // heads`` outputs side by side (the mu / logstd pair of a ConditionalVAE or a SquashedNormalPolicy)."""
    rs = np.random.RandomState(seed)

    def linear(fan_in, fan_out):
        k = 1.0 / np.sqrt(fan_in)
        return (rs.uniform(-k, k, size=(fan_in, fan_out)).astype(np.float32), rs.uniform(-k, k, size=(fan_out,)).astype(np.float32))

    p = {}
    p['fc1_w'], p['fc1_b'] = linear(obs_dim + act_dim, hidden1)
    p['fc2_w'], p['fc2_b'] = linear(hidden1, hidden2)
    hw = [linear(hidden2, out_dim // heads) for _ in range(heads)]
    p['head_w'] = np.ascontiguousarray(np.concatenate([w for w, _ in hw], axis=1))
    p['head_b'] = np.ascontiguousarray(np.concatenate([b for _, b in hw]))
    return p


def _check_transitions(device, D, E, obs, act, rew, nxt, ter):
    """The one-call updates hand raw pointers to the library (which only knows the row count): everything DeviceAMLP._rows asserts
    on the per-phase path is asserted here - device, float32, shapes [B, D] / [B, E] / [B] - before any data_ptr() is taken."""
    B = obs.shape[0]
    for name, t, shape in (('observations', obs, (B, D)), ('actions', act, (B, E)), ('rewards', rew, (B,)),
                           ('next observations', nxt, (B, D)), ('terminals', ter, (B,))):
        assert isinstance(t, torch.Tensor) and t.is_cuda and (device.index is None or t.device.index == device.index), \
            '%s must be a tensor on %s' % (name, device)
        assert t.dtype == torch.float32, '%s must be float32 (got %s)' % (name, t.dtype)
        assert tuple(t.shape) == shape or (len(shape) == 1 and tuple(t.shape) == (B, 1)), \
            '%s must have shape %s (got %s)' % (name, shape, tuple(t.shape))
    return [t if t.is_contiguous() else t.contiguous() for t in (obs, act, rew, nxt, ter)]


def _allreduce_group(nets):
    """Data parallel: ONE mean all-reduce for the flat gradients of a group of networks that are stepped together."""
    if not rdist.collectives_active():
        return
    flats = [net.flat_gradient() for net in nets]
    g = torch.cat(flats)
    rdist.allreduce_mean_(g)
    o = 0
    for net, f in zip(nets, flats):
        net.set_flat_gradient(g[o:o + f.numel()].contiguous())
        o += f.numel()


class BCQ(_ModelIO):
    """d3rlpy.algos.BCQ(batch_size=256) as 'BCQ-conti' instantiates it (script/batchrl_trainer.py:61-73): default
    ``VectorEncoderWithAction([256, 256])`` everywhere, Adam 1e-3 for actor / critic / imitator, gamma 0.99, tau 0.005, two
    critics, lam 0.75, 100 sampled actions, action_flexibility 0.05, latent_size 32, beta 0.5, update_actor_interval 1,
    rl_start_step 0 (d3rlpy 0.91 defaults; d3rlpy is absent here - PARITY UNPINNED).

    One ``update`` = d3rlpy's ``BCQ._update``: imitator step, critic step, actor step, soft target updates.  All network
    arithmetic runs through ``librl4rs_hip`` (``rl4rs_amlp_*`` and the cvae / residual / bcq_target / critic_mse entry points);
    torch supplies device memory and the Gaussian noise.  ``noise`` (tests): dict with ``eps`` [B, L], ``z_target`` [B * n, L]
    and ``z_actor`` [B, L] replacing the three ``torch.randn`` draws (the latter two before clamping to +-0.5).
    With ``torch.distributed`` initialised each rank trains on its own minibatches; the flat gradients of the networks of a
    phase are mean-all-reduced together before Adam (3 collectives per update; the phases depend on each other)."""

    def __init__(self, config, obs_dim, action_size=None, batch_size=256, actor_learning_rate=1e-3, critic_learning_rate=1e-3,
                 imitator_learning_rate=1e-3, gamma=0.99, tau=0.005, update_actor_interval=1, lam=0.75, n_action_samples=100,
                 action_flexibility=0.05, rl_start_step=0, latent_size=32, beta=0.5, predict_rows=512, seed=0, device=None,
                 nograd_precision='fp16x2'):
        self.config = config
        self.D = int(obs_dim)
        self.E = int(action_size if action_size is not None else config['action_emb_size'])
        self.L = int(latent_size)
        self.batch_size = int(batch_size)
        self.n = int(n_action_samples)
        self.actor_lr, self.critic_lr, self.imitator_lr = float(actor_learning_rate), float(critic_learning_rate), float(imitator_learning_rate)
        self.gamma, self.tau, self.lam = float(gamma), float(tau), float(lam)
        self.update_actor_interval, self.rl_start_step = int(update_actor_interval), int(rl_start_step)
        self.scale, self.beta = float(action_flexibility), float(beta)
        self.predict_rows = max(int(predict_rows), 1)
        # arithmetic of the forwards that never see a backward (the batch x n_action_samples rows of compute_target and of
        # _predict_best_action): 'fp16x2' = the scorer's split form, one fused launch per network (DeviceAMLP.forward), or 'fp32'
        assert nograd_precision in ('fp16x2', 'fp32')
        self.nograd = nograd_precision
        self.seed = int(seed)
        self.total_step = 0
        B, n, D, E, L = self.batch_size, self.n, self.D, self.E, self.L
        big = max(B, self.predict_rows) * n

        def net(act_dim, out_dim, seed_off, max_rows, grad_rows, head_act='none', heads=1):
            return D_.DeviceAMLP(D, act_dim, out_dim, init_amlp_params(D, act_dim, out_dim, seed=seed + seed_off, heads=heads),
                                 head_act=head_act, max_rows=max_rows, max_grad_rows=grad_rows, device=device)

        self.imit_enc = net(E, 2 * L, 0, B, B, heads=2)                 # ConditionalVAE._encoder_encoder + _mu | _logstd
        self.imit_dec = net(L, E, 1, big, B, head_act='tanh')           # ConditionalVAE._decoder_encoder + _fc, tanh
        self.policy = net(E, E, 2, big, B, head_act='tanh')             # DeterministicResidualPolicy
        self.policy_targ = net(E, E, 2, B * n, 0, head_act='tanh')
        self.q1 = net(E, 1, 3, big, B)                                  # ContinuousMeanQFunction x 2
        self.q2 = net(E, 1, 4, big, B)
        self.q1_targ = net(E, 1, 3, B * n, 0)
        self.q2_targ = net(E, 1, 4, B * n, 0)
        self.policy_targ.copy_from(self.policy)
        self.q1_targ.copy_from(self.q1)
        self.q2_targ.copy_from(self.q2)
        self.nets = [self.imit_enc, self.imit_dec, self.policy, self.policy_targ, self.q1, self.q2, self.q1_targ, self.q2_targ]
        self.device = self.q1.device
        self._gen = torch.Generator(device=self.device)
        self._gen.manual_seed(self.seed + 1000003 * rdist.rank())
        self._minus_inv_b = None
        self._imit_w = torch.tensor([1.0 / self.E, self.beta / self.L], dtype=torch.float32, device=self.device)
        self.one_call = True             # update() as one library call on a single rank (False: the per-phase calls; tests compare the two)
        self._ws = {}                    # minibatch size -> workspace of rl4rs_bcq_update

    def _randn(self, rows, noise, key):
        if noise is not None and key in noise:
            return noise[key].to(device=self.device, dtype=torch.float32).contiguous()
        return torch.randn((rows, self.L), generator=self._gen, device=self.device, dtype=torch.float32)

    def _clipped_latent(self, rows, noise, key):
        """clamp(randn, -0.5, 0.5) (BCQImpl: the decoder's latent is clipped when actions are sampled); a caller-supplied noise
        tensor is never modified"""
        own = noise is None or key not in noise
        z = self._randn(rows, noise, key)
        return z.clamp_(-0.5, 0.5) if own else z.clamp(-0.5, 0.5)

    def _sample_actions(self, obs, z, policy, rep, nograd=False):
        """imitator.decode(x, clip(z)) -> policy residual: [rows, E] actions for ``rep`` latents per observation."""
        sampled = self.imit_dec.forward(obs, z, rep=rep, nograd=nograd)
        t = policy.forward(obs, sampled, rep=rep, nograd=nograd)
        return sampled, t, D_.residual_action(sampled, t, self.scale)

    def _update_one_call(self, obs, act, rew, nxt, ter, noise):
        """The whole update as ONE library call (rl4rs_bcq_update: the same launches with the same arguments as ``update`` below
        issues one by one - after the fused kernels the update was bound by its ~55 Python -> C calls, not by the GPU)."""
        from . import _lib
        B, n, E, L = obs.shape[0], self.n, self.E, self.L
        lib = _lib.load()
        ws = self._ws.get(B)
        if ws is None:
            ws = self._ws[B] = torch.empty(int(lib.rl4rs_bcq_workspace_floats(B, n, E, L)), dtype=torch.float32, device=self.device)
        total = (B + B * n + B) * L
        if noise is None:
            nz = torch.randn(total, generator=self._gen, device=self.device, dtype=torch.float32)      # eps | z_target | z_actor: one draw
        else:
            parts = [self._randn(B, noise, 'eps'), self._randn(B * n, noise, 'z_target'), self._randn(B, noise, 'z_actor')]
            nz = torch.cat([p.reshape(-1) for p in parts])                                           # (a copy: the caller's noise is never modified)
        metrics = torch.zeros(3, dtype=torch.float32, device=self.device)
        do_rl = self.total_step >= self.rl_start_step
        do_actor = do_rl and self.total_step % self.update_actor_interval == 0
        cont = _check_transitions(self.device, self.D, self.E, obs, act, rew, nxt, ter)
        st = _lib.BcqStep(*[net.h.value for net in (self.imit_enc, self.imit_dec, self.policy, self.policy_targ, self.q1, self.q2, self.q1_targ, self.q2_targ)],
                          B, n, E, L, self.beta, self.scale, self.lam, self.gamma, self.tau, self.imitator_lr, self.critic_lr, self.actor_lr,
                          1 if do_rl else 0, 1 if do_actor else 0, 1 if self.nograd == 'fp16x2' else 0, int(self.q1.H16_MIN_ROWS),
                          *[t.data_ptr() for t in cont], nz.data_ptr(), ws.data_ptr(), metrics.data_ptr())
        _lib.check(lib.rl4rs_bcq_update(C.byref(st), D_._stream()))
        self.total_step += 1
        out = {'imitator_loss': metrics[0]}
        if do_rl:
            out['critic_loss'] = metrics[1]
        if do_actor:
            out['actor_loss'] = metrics[2]
        return out

    def update(self, obs, act, rew, nxt, ter, noise=None):
        if self.one_call and not rdist.collectives_active():
            return self._update_one_call(obs, act, rew, nxt, ter, noise)
        B, n = obs.shape[0], self.n
        metrics = {}
        # --- imitator (BCQImpl.update_imitator: ConditionalVAE.compute_error) ---
        eps = self._randn(B, noise, 'eps')
        enc_out = self.imit_enc.forward(obs, act)
        z = D_.cvae_sample(enc_out, eps)
        y = self.imit_dec.forward(obs, z)
        loss2, d_dec = D_.cvae_loss(y, act, enc_out)
        dz = self.imit_dec.backward(obs, z, d_dec, want_dact=True)
        d_enc = D_.cvae_encoder_grad(enc_out, eps, dz, self.beta)
        self.imit_enc.backward(obs, act, d_enc)
        _allreduce_group([self.imit_enc, self.imit_dec])
        D_.amlp_adam_multi([self.imit_enc, self.imit_dec], [self.imitator_lr] * 2)            # one optimiser launch per phase
        metrics['imitator_loss'] = torch.dot(loss2, self._imit_w)          # mse / E + beta * kl / L as ONE launch
        if self.total_step >= self.rl_start_step:
            # --- critic (DDPGBaseImpl.update_critic with BCQImpl.compute_target) ---
            zt = self._clipped_latent(B * n, noise, 'z_target')
            _, _, a_next = self._sample_actions(nxt, zt, self.policy_targ, n, nograd=self.nograd)
            q1n = self.q1_targ.forward(nxt, a_next, rep=n, nograd=self.nograd)
            q2n = self.q2_targ.forward(nxt, a_next, rep=n, nograd=self.nograd)
            yq, _ = D_.bcq_target(q1n, q2n, n, self.lam, rew, ter, self.gamma)
            q1v, q2v = D_.amlp_forward_multi([self.q1, self.q2], obs, act)              # the twin critics as one launch each way
            closs2, dq1, dq2 = D_.critic_mse(q1v, q2v, yq)
            D_.amlp_backward_multi([self.q1, self.q2], obs, act, [dq1, dq2])
            _allreduce_group([self.q1, self.q2])
            D_.amlp_adam_multi([self.q1, self.q2], [self.critic_lr] * 2)
            metrics['critic_loss'] = closs2.sum()
            if self.total_step % self.update_actor_interval == 0:
                # --- actor (BCQImpl.compute_actor_loss: -Q_1(s, pi(s, decode(s, z))).mean()) ---
                za = self._clipped_latent(B, noise, 'z_actor')
                sampled, t, a_pi = self._sample_actions(obs, za, self.policy, 1)
                qv = self.q1.forward(obs, a_pi)
                if self._minus_inv_b is None or self._minus_inv_b.numel() != B:
                    self._minus_inv_b = torch.full((B, 1), -1.0 / B, dtype=torch.float32, device=self.device)
                da = self.q1.backward(obs, a_pi, self._minus_inv_b, want_dact=True, want_param_grad=False)
                d_pre = D_.residual_grad(sampled, t, self.scale, da)
                self.policy.backward(obs, sampled, d_pre)
                _allreduce_group([self.policy])
                # Adam of the actor + the three soft target updates (the critics were stepped in their own phase): one launch
                D_.amlp_adam_multi([self.policy, self.q1, self.q2], [self.actor_lr, 0.0, 0.0], targets=[self.policy_targ, self.q1_targ, self.q2_targ],
                                   tau=self.tau, step=[True, False, False])
                metrics['actor_loss'] = torch.dot(qv.view(-1), self._minus_inv_b.view(-1))       # -mean(q) as ONE launch
        self.total_step += 1
        return metrics

    def fit(self, transitions, n_steps, shuffle_seed=None, to_host=True):
        """``n_steps`` updates over ``transitions`` (``transitions_from_mdp(..., discrete_action=False)``), epoch-wise random
        permutation like d3rlpy's ``fit``; returns dict of per-step loss lists (``to_host=False``: of stacked device tensors,
        nothing waits for the GPU)."""
        obs, act, rew, nxt, ter = [t.to(self.device) for t in transitions]
        n = obs.shape[0]
        assert n >= self.batch_size, 'dataset smaller than one minibatch'
        E = getattr(self, 'E', None) or self.A
        assert act.dim() == 2 and act.shape[1] == E and act.dtype == torch.float32, 'continuous actions [N, %d] needed' % E
        rs = np.random.RandomState(self.seed if shuffle_seed is None else shuffle_seed)
        hist, ep, pos = [], None, n
        for _ in range(n_steps):
            if pos + self.batch_size > n:
                perm = torch.from_numpy(rs.permutation(n)).to(obs.device)          # applied once per epoch: minibatches are views
                ep = [t[perm] for t in (obs, act, rew, nxt, ter)]
                pos = 0
            lo, hi = pos, pos + self.batch_size
            pos = hi
            hist.append(self.update(*[t[lo:hi] for t in ep]))
        out = {}
        for k in self._LOSS_KEYS:
            vals = [h[k] for h in hist if k in h]
            if not vals:
                out[k] = []
                continue
            st = torch.stack(vals)
            # a non-finite loss poisons Adam for good (the fused fp16x2 no-grad forward turns a row whose hidden activation
            # leaves the fp16 range into NaN): the flag stays on the device, check_status() / to_host reads it
            bad = ~torch.isfinite(st).all()
            self._nonfinite = bad if getattr(self, '_nonfinite', None) is None else (self._nonfinite | bad)
            out[k] = [float(x) for x in st.cpu()] if to_host else st
        if to_host:
            self.check_status()
        return out

    NONFINITE_MESSAGE = ("a training loss of this learner became non-finite; with nograd_precision='fp16x2' a hidden activation that "
                         "leaves the fp16 range does that - rebuild the learner with nograd_precision='fp32'")

    def check_status(self):
        """Raise if any loss of a ``fit`` so far was NaN / inf (one host read; ``fit(to_host=True)`` and ``save_model`` call it)."""
        flag = getattr(self, '_nonfinite', None)
        if flag is not None and bool(flag.item()):
            self._nonfinite = None
            raise RuntimeError(self.NONFINITE_MESSAGE)

    _LOSS_KEYS = ('imitator_loss', 'critic_loss', 'actor_loss')

    def predict(self, obs, noise=None):
        """BCQImpl._predict_best_action: 100 sampled actions per observation, the one the FIRST critic values highest.
        obs [B, D] (device tensor or array) -> actions [B, E] on the device: the embedding the env's K-NN resolves."""
        obs = D_._dev_tensor(obs, torch.float32, self.device)
        out = torch.empty((obs.shape[0], self.E), dtype=torch.float32, device=self.device)
        n = self.n
        for lo in range(0, obs.shape[0], self.predict_rows):
            x = obs[lo:lo + self.predict_rows].contiguous()
            b = x.shape[0]
            if noise is not None:
                z = noise[lo * n:(lo + b) * n].to(device=self.device, dtype=torch.float32)
            else:
                z = torch.randn((b * n, self.L), generator=self._gen, device=self.device, dtype=torch.float32)
            _, _, a = self._sample_actions(x, z.clamp(-0.5, 0.5).contiguous(), self.policy, n, nograd=self.nograd)
            q = self.q1.forward(x, a, rep=n, nograd=self.nograd)
            _, best = D_.bcq_target(q, None, n, 0.0, want_best=True)
            out[lo:lo + b] = D_.pick_rows(a, best, n)
        return out

    def predict_value(self, obs, actions):
        """AlgoBase.predict_value: the mean of the critics' values of (obs, action)."""
        obs = D_._dev_tensor(obs, torch.float32, self.device)
        actions = D_._dev_tensor(actions, torch.float32, self.device)
        out = torch.empty(obs.shape[0], dtype=torch.float32, device=self.device)
        rows = max(self.batch_size, self.predict_rows) * self.n
        for lo in range(0, obs.shape[0], rows):
            x, a = obs[lo:lo + rows].contiguous(), actions[lo:lo + rows].contiguous()
            out[lo:lo + x.shape[0]] = 0.5 * (self.q1.forward(x, a, nograd=self.nograd)[:, 0] + self.q2.forward(x, a, nograd=self.nograd)[:, 0])
        return out

    def close(self):
        for net in self.nets:
            net.close()


class StandardRewardScaler(object):
    """d3rlpy 0.91 ``reward_scaler='standard'``: (r - mean) / (std + 1e-3) with the statistics of the dataset's rewards."""

    def __init__(self, rewards, eps=1e-3):
        r = torch.as_tensor(rewards, dtype=torch.float64)
        self.mean, self.std, self.eps = float(r.mean()), float(r.std(unbiased=False)), float(eps)

    def transform(self, r):
        return ((r - self.mean) / (self.std + self.eps)).to(torch.float32)

    def reverse_transform(self, q):
        """d3rlpy ``StandardRewardScaler.reverse_transform``: q * (std + eps) + mean (policy_model.predict_q, policy_model.py:55-61)"""
        return q * (self.std + self.eps) + self.mean

    @classmethod
    def from_stats(cls, mean, std, eps=1e-3):
        s = cls.__new__(cls)
        s.mean, s.std, s.eps = float(mean), float(std), float(eps)
        return s


class _ScalarParam(object):
    """One learned scalar (SAC's log-temperature, CQL's log-alpha) with torch-style Adam, kept on the device: a handful of
    one-element tensor ops per update (plumbing; no network arithmetic)."""

    def __init__(self, value, device):
        self.state = torch.zeros(4, dtype=torch.float32, device=device)         # {value, Adam m, Adam v, pad}: what rl4rs_cql_update steps in place
        self.p, self.m, self.v = self.state[0:1], self.state[1:2], self.state[2:3]
        self.p.fill_(float(value))
        self.t = 0

    def adam_step(self, grad, lr, beta1=0.9, beta2=0.999, eps=1e-8):
        if rdist.collectives_active():
            grad = rdist.allreduce_mean_(grad.clone())
        self.t += 1
        self.m.mul_(beta1).add_(grad, alpha=1.0 - beta1)
        self.v.mul_(beta2).addcmul_(grad, grad, value=1.0 - beta2)
        denom = (self.v / (1.0 - beta2 ** self.t)).sqrt_().add_(eps)
        self.p.addcdiv_(self.m, denom, value=-lr / (1.0 - beta1 ** self.t))


class CQL(_ModelIO):
    """d3rlpy.algos.CQL (continuous) as 'CQL-conti' instantiates it (script/batchrl_trainer.py:91-107): default encoders,
    actor lr 1e-4, critic lr 3e-4, temperature / alpha lr 1e-4, tau 0.005, two critics, initial temperature 1, initial alpha 1,
    alpha threshold 10, conservative weight 5, 10 action samples, deterministic (non-soft) backup; the script sets gamma = 1 and
    ``reward_scaler='standard'`` (pass ``gamma=1.0`` and ``reward_scaler=StandardRewardScaler(dataset rewards)``).
    d3rlpy 0.91 defaults as published; PARITY UNPINNED.

    One ``update`` = d3rlpy's ``CQL._update``: temperature step, alpha step, critic step (TD on the min of the target critics at
    tanh(mu(s')) + the conservative term), actor step, soft critic-target update.  Network arithmetic through ``librl4rs_hip``
    (``rl4rs_amlp_*``, ``rl4rs_squashed_sample``, ``rl4rs_cql_critic_loss``, ``rl4rs_sac_actor_grad``, ``rl4rs_twin_min``); torch
    supplies device memory, the noise and the two learned scalars' Adam.  The 1 + 3n rows of an observation (dataset action | n
    samples of pi(s) | n of pi(s') | n uniform) go through each critic as ONE forward / backward with the observation side of
    the first layer computed once per observation (rep = 31).  ``noise`` (tests): dict with ``eps_temp`` [B, A], ``alpha`` and
    ``critic`` = (eps_t [B*n, A], eps_tp1 [B*n, A], uniform [B, n, A]), ``eps_actor`` [B, A]."""

    def __init__(self, config, obs_dim, action_size=None, batch_size=256, actor_learning_rate=1e-4, critic_learning_rate=3e-4,
                 temp_learning_rate=1e-4, alpha_learning_rate=1e-4, gamma=0.99, tau=0.005, initial_temperature=1.0, initial_alpha=1.0,
                 alpha_threshold=10.0, conservative_weight=5.0, n_action_samples=10, reward_scaler=None, predict_rows=4096, seed=0,
                 device=None, nograd_precision='fp16x2'):
        assert nograd_precision in ('fp16x2', 'fp32')
        self.nograd = nograd_precision               # the alpha step's critic forwards (never differentiated): see BCQ
        self.config = config
        self.D = int(obs_dim)
        self.A = int(action_size if action_size is not None else config['action_emb_size'])
        self.batch_size, self.n = int(batch_size), int(n_action_samples)
        self.m = 1 + 3 * self.n
        self.actor_lr, self.critic_lr = float(actor_learning_rate), float(critic_learning_rate)
        self.temp_lr, self.alpha_lr = float(temp_learning_rate), float(alpha_learning_rate)
        self.gamma, self.tau = float(gamma), float(tau)
        self.alpha_threshold, self.conservative_weight = float(alpha_threshold), float(conservative_weight)
        self.reward_scaler = reward_scaler
        self.predict_rows = max(int(predict_rows), 1)
        self.seed = int(seed)
        self.total_step = 0
        B, D, A, m = self.batch_size, self.D, self.A, self.m
        rows = max(B * m, self.predict_rows)

        def net(act_dim, out_dim, seed_off, max_rows, grad_rows, heads=1):
            return D_.DeviceAMLP(D, act_dim, out_dim, init_amlp_params(D, act_dim, out_dim, seed=seed + seed_off, heads=heads),
                                 max_rows=max_rows, max_grad_rows=grad_rows, device=device)

        self.policy = net(0, 2 * A, 0, max(B, self.predict_rows), B, heads=2)        # SquashedNormalPolicy: encoder + _mu | _logstd
        self.q1 = net(A, 1, 1, rows, B * m)
        self.q2 = net(A, 1, 2, rows, B * m)
        self.q1_targ = net(A, 1, 1, B, 0)
        self.q2_targ = net(A, 1, 2, B, 0)
        self.q1_targ.copy_from(self.q1)
        self.q2_targ.copy_from(self.q2)
        self.nets = [self.policy, self.q1, self.q2, self.q1_targ, self.q2_targ]
        self.device = self.q1.device
        self.log_temp = _ScalarParam(np.log(initial_temperature), self.device)
        self.log_alpha = _ScalarParam(np.log(initial_alpha), self.device)
        self._gen = torch.Generator(device=self.device)
        self._gen.manual_seed(self.seed + 1000003 * rdist.rank())
        self._acts = torch.zeros((B, m, A), dtype=torch.float32, device=self.device)
        self._offs = torch.zeros((B, m), dtype=torch.float32, device=self.device)
        self._offs[:, 1 + 2 * self.n:] = float(A * np.log(0.5))                       # log of the uniform density on [-1, 1]^A
        self.one_call = True             # update() as one library call on a single rank (False: the per-phase calls; tests compare the two)
        self._ws = {}

    def _randn(self, shape, given):
        if given is not None:
            return given.to(device=self.device, dtype=torch.float32).contiguous()
        return torch.randn(shape, generator=self._gen, device=self.device, dtype=torch.float32)

    def _conservative_rows(self, head_obs, head_nxt, act, given):
        """[B, m, A] actions and [B, m] importance offsets of the conservative term: column 0 the dataset action, then n samples of
        pi(. | s), n of pi(. | s'), n uniform on [-1, 1]^A (CQLImpl._compute_policy_is_values / _compute_random_is_values)."""
        B, n, A = act.shape[0], self.n, self.A
        acts, offs = self._acts[:B], self._offs[:B]
        acts[:, 0] = act
        e_t, e_tp1, uni = given if given is not None else (None, None, None)
        flat_a, flat_o = acts.view(B * self.m, A), offs.view(B * self.m)
        D_.squashed_sample(head_obs, self._randn((B * n, A), e_t), rep=n, act_out=flat_a, logp_out=flat_o, out_rep=self.m, out_off=1)
        D_.squashed_sample(head_nxt, self._randn((B * n, A), e_tp1), rep=n, act_out=flat_a, logp_out=flat_o, out_rep=self.m, out_off=1 + n)
        if uni is not None:
            acts[:, 1 + 2 * n:] = uni.to(device=self.device, dtype=torch.float32)
        else:
            acts[:, 1 + 2 * n:].uniform_(-1.0, 1.0, generator=self._gen)
        return flat_a, flat_o

    def _conservative_value(self, sums, B):
        """conservative_weight * (mean_c mean_b logsumexp - mean_c mean_b Q(s, a))"""
        return self.conservative_weight * ((sums[2] + sums[3]) - (sums[4] + sums[5])) / (2.0 * B)

    def _update_one_call(self, obs, act, rew, nxt, ter, noise):
        """The whole update as ONE library call (rl4rs_cql_update: the launches of ``update`` below, the two learned scalars' Adam on
        the device)."""
        from . import _lib
        B, n, A = obs.shape[0], self.n, self.A
        lib = _lib.load()
        ws = self._ws.get(B)
        if ws is None:
            ws = self._ws[B] = torch.empty(int(lib.rl4rs_cql_workspace_floats(B, n, A)), dtype=torch.float32, device=self.device)
        if noise:
            # any subset of the keys, like the per-phase path's noise.get(...): what is not given is drawn from the generator
            def normal_part(x, rows):
                assert x is None or x.numel() == rows * A, 'noise tensor of %d values where %d x %d are needed' % (x.numel(), rows, A)
                return self._randn((rows, A), x).reshape(-1)

            def uniform_part(x):
                if x is None:
                    return torch.empty(B * n * A, dtype=torch.float32, device=self.device).uniform_(-1.0, 1.0, generator=self._gen)
                assert x.numel() == B * n * A, 'uniform noise of %d values where %d are needed' % (x.numel(), B * n * A)
                return x.to(device=self.device, dtype=torch.float32).reshape(-1)

            a_t, a_tp1, a_u = noise.get('alpha') or (None, None, None)
            c_t, c_tp1, c_u = noise.get('critic') or (None, None, None)
            normal = torch.cat([normal_part(noise.get('eps_temp'), B), normal_part(a_t, B * n), normal_part(a_tp1, B * n),
                                normal_part(c_t, B * n), normal_part(c_tp1, B * n), normal_part(noise.get('eps_actor'), B)])
            uniform = torch.cat([uniform_part(a_u), uniform_part(c_u)])
        else:
            normal = torch.randn((2 * B + 4 * B * n) * A, generator=self._gen, device=self.device, dtype=torch.float32)
            uniform = torch.empty(2 * B * n * A, dtype=torch.float32, device=self.device).uniform_(-1.0, 1.0, generator=self._gen)
        metrics = torch.zeros(4, dtype=torch.float32, device=self.device)
        cont = _check_transitions(self.device, self.D, self.A, obs, act, rew, nxt, ter)
        st = _lib.CqlStep(*[net.h.value for net in (self.policy, self.q1, self.q2, self.q1_targ, self.q2_targ)], B, n, A,
                          self.gamma, self.tau, self.actor_lr, self.critic_lr, self.temp_lr, self.alpha_lr, self.alpha_threshold, self.conservative_weight,
                          1 if self.nograd == 'fp16x2' else 0, int(self.q1.H16_MIN_ROWS), self.log_temp.t, self.log_alpha.t,
                          self.log_temp.state.data_ptr(), self.log_alpha.state.data_ptr(), *[t.data_ptr() for t in cont],
                          normal.data_ptr(), uniform.data_ptr(), ws.data_ptr(), metrics.data_ptr())
        _lib.check(lib.rl4rs_cql_update(C.byref(st), D_._stream()))
        out = {'critic_loss': metrics[0], 'actor_loss': metrics[1]}
        if self.temp_lr > 0:
            self.log_temp.t += 1
            out['temp_loss'] = metrics[2]
        if self.alpha_lr > 0:
            self.log_alpha.t += 1
            out['alpha_loss'] = metrics[3]
        self.total_step += 1
        return out

    def update(self, obs, act, rew, nxt, ter, noise=None):
        noise = noise or {}
        B, A, m = obs.shape[0], self.A, self.m
        if self.reward_scaler is not None:
            if isinstance(self.reward_scaler, str):
                raise ValueError("reward_scaler=%r is fitted by fit_mdp(dataset); pass StandardRewardScaler(rewards) to use update / fit "
                                 "directly" % self.reward_scaler)
            rew = self.reward_scaler.transform(rew)
        if self.one_call and not rdist.collectives_active():
            return self._update_one_call(obs, act, rew, nxt, ter, noise)
        metrics = {}
        # the policy does not change until the actor step: its heads on s' and s are computed once (s last: the handle keeps the
        # activations of s for the actor's backward)
        head_nxt = self.policy.forward(nxt)
        head_obs = self.policy.forward(obs)
        # --- temperature (SACImpl.update_temp): -(exp(log_temp) * (logp - A)).mean(), gradient wrt log_temp
        if self.temp_lr > 0:
            _, logp = D_.squashed_sample(head_obs, self._randn((B, A), noise.get('eps_temp')))
            targ = (logp - A).mean()
            temp = self.log_temp.p.exp()
            metrics['temp_loss'] = -(temp * targ)[0]
            self.log_temp.adam_step(-(temp * targ), self.temp_lr)
        # --- alpha (CQLImpl.update_alpha): -conservative loss, gradient wrt log_alpha
        if self.alpha_lr > 0:
            fa, fo = self._conservative_rows(head_obs, head_nxt, act, noise.get('alpha'))
            sums, _, _ = D_.cql_critic_loss(self.q1.forward(obs, fa, rep=m, nograd=self.nograd), self.q2.forward(obs, fa, rep=m, nograd=self.nograd), fo, m)
            gap = self._conservative_value(sums, B) - self.alpha_threshold
            ea = self.log_alpha.p.exp()
            metrics['alpha_loss'] = -(ea.clamp(0.0, 1e6) * gap)[0]
            self.log_alpha.adam_step(torch.where(ea <= 1e6, -(ea * gap), torch.zeros_like(ea)), self.alpha_lr)
        # --- critic (DDPGBaseImpl.update_critic with CQLImpl.compute_critic_loss / _compute_deterministic_target)
        a_next, _ = D_.squashed_sample(head_nxt, None)
        yq, _ = D_.bcq_target(self.q1_targ.forward(nxt, a_next), self.q2_targ.forward(nxt, a_next), 1, 1.0, rew, ter, self.gamma)
        fa, fo = self._conservative_rows(head_obs, head_nxt, act, noise.get('critic'))
        clipped_alpha = self.log_alpha.p.exp().clamp(0.0, 1e6)
        aw = (clipped_alpha * self.conservative_weight).contiguous()
        q1v = self.q1.forward(obs, fa, rep=m)
        q2v = self.q2.forward(obs, fa, rep=m)
        sums, dq1, dq2 = D_.cql_critic_loss(q1v, q2v, fo, m, y=yq, alpha_w=aw)
        self.q1.backward(obs, fa, dq1, rep=m)
        self.q2.backward(obs, fa, dq2, rep=m)
        _allreduce_group([self.q1, self.q2])
        D_.amlp_adam_multi([self.q1, self.q2], [self.critic_lr] * 2)
        metrics['critic_loss'] = (sums[0] + sums[1]) / B + (clipped_alpha * (self._conservative_value(sums, B) - self.alpha_threshold))[0]
        # --- actor (SACImpl.compute_actor_loss): (exp(log_temp) * logp - min_c Q_c(s, a)).mean()
        eps = self._randn((B, A), noise.get('eps_actor'))
        a_pi, logp = D_.squashed_sample(head_obs, eps)
        q1p, q2p = D_.amlp_forward_multi([self.q1, self.q2], obs, a_pi)
        qmin, dq1, dq2 = D_.twin_min(q1p, q2p, want_grad=True)
        g1, g2 = D_.amlp_backward_multi([self.q1, self.q2], obs, a_pi, [dq1.view(B, 1), dq2.view(B, 1)], want_dact=True, want_param_grad=False)
        g_a = g1.add_(g2)
        d_head = D_.sac_actor_grad(head_obs, eps, a_pi, g_a, self.log_temp.p)
        self.policy.backward(obs, None, d_head)
        _allreduce_group([self.policy])
        D_.amlp_adam_multi([self.policy, self.q1, self.q2], [self.actor_lr, 0.0, 0.0], targets=[None, self.q1_targ, self.q2_targ], tau=self.tau,
                           step=[True, False, False])
        metrics['actor_loss'] = (self.log_temp.p.exp() * logp - qmin).mean()
        self.total_step += 1
        return metrics

    fit = BCQ.fit
    check_status = BCQ.check_status
    NONFINITE_MESSAGE = BCQ.NONFINITE_MESSAGE
    _LOSS_KEYS = ('critic_loss', 'actor_loss', 'temp_loss', 'alpha_loss')

    def predict(self, obs):
        """SquashedNormalPolicy.best_action: tanh(mu(s)) - the embedding the env's K-NN resolves."""
        obs = D_._dev_tensor(obs, torch.float32, self.device)
        out = torch.empty((obs.shape[0], self.A), dtype=torch.float32, device=self.device)
        rows = self.policy.max_rows
        for lo in range(0, obs.shape[0], rows):
            x = obs[lo:lo + rows].contiguous()
            D_.squashed_sample(self.policy.forward(x), None, act_out=out[lo:lo + x.shape[0]])
        return out

    def predict_value(self, obs, actions):
        obs = D_._dev_tensor(obs, torch.float32, self.device)
        actions = D_._dev_tensor(actions, torch.float32, self.device)
        out = torch.empty(obs.shape[0], dtype=torch.float32, device=self.device)
        rows = self.q1.max_rows
        for lo in range(0, obs.shape[0], rows):
            x, a = obs[lo:lo + rows].contiguous(), actions[lo:lo + rows].contiguous()
            out[lo:lo + x.shape[0]] = 0.5 * (self.q1.forward(x, a)[:, 0] + self.q2.forward(x, a)[:, 0])
        return out

    def close(self):
        for net in self.nets:
            net.close()
