"""On-device policy training loop over the GPU env (A2C / PPO with the reference's RLlib hyper-parameters,
script/modelfree_train.py:179-304: gamma = 1, GAE lambda = 1, lr 1e-4, vf_loss_coeff 0.5; A2C entropy 0.01 and
grad_clip 10; PPO clip 0.3, kl_coeff 0.2 adapted towards kl_target 0.01, vf_clip 500, one SGD pass over minibatches of 256;
train_batch_size = min(B * T, 1024) timesteps per train call, :409).

Rollout, policy forward/backward and Adam run through librl4rs_hip.so; torch is used for buffers, the reversed
cumulative sum of rewards and (data-parallel) the RCCL all-reduce of the flat gradient buffer.

Data parallelism: every rank owns its own env batch and sampling stream (``seed`` differs per rank) and a replica
of the policy that is IDENTICAL on all ranks: initialised from the shared ``init_seed``, rank 0's parameters / Adam state are
broadcast at construction, and every optimiser step applies the rank-mean gradient.  A2C: one all-reduce per iteration.
PPO: one per minibatch (synchronous SGD over the global minibatch of world_size x 256 samples): phase A + the gradient
tiles of the persistent pass run as ONE launch per minibatch (rl4rs_policy_ppo_minibatch_grad), then all-reduce, then Adam.
"""
import math

import numpy as np
import torch

from . import dist as rdist
from . import device as D


def update_kl_coeff(kl_coeff, sampled_kl, kl_target):
    """RLlib's adaptive KL penalty (ray 1.5.1 ``KLCoeffMixin.update_kl``, configured by modelfree_train.py:189 kl_coeff 0.2 and
    :216 kl_target 0.01): x1.5 when the pass-mean KL exceeds 2 x target, x0.5 when it is below target / 2."""
    if sampled_kl > 2.0 * kl_target:
        return kl_coeff * 1.5
    if sampled_kl < 0.5 * kl_target:
        return kl_coeff * 0.5
    return kl_coeff


def rollouts_per_train_call(batch_size, max_steps, train_batch_size=None):
    """modelfree_train.py:403-409: rollout_fragment_length = max_steps, batch_mode complete_episodes, train_batch_size =
    min(B * T, 1024).  One vector-env rollout yields B * T timesteps, RLlib keeps sampling until it holds at least
    train_batch_size, so the reference's value always means exactly ONE rollout per train call; a larger explicit value asks
    for ceil(train_batch_size / (B * T)) rollouts."""
    per = int(batch_size) * int(max_steps)
    if train_batch_size is None:
        train_batch_size = min(per, 1024)
    return max(1, int(math.ceil(float(train_batch_size) / per)))


def _returns(rew, R, T, B):
    """gamma = 1, lambda = 1: reward-to-go within each rollout ([R, T, B] layout of the flat buffers)."""
    r = rew.view(R, T, B)
    return torch.flip(torch.cumsum(torch.flip(r, dims=[1]), dim=1), dims=[1]).reshape(-1).to(torch.float32)


class LazyStats(object):
    """Mapping over the statistics of one train call.  The numbers live on the device until somebody looks: the train call
    only enqueues an asynchronous copy into pinned memory + an event, so a loop that does not read its statistics never
    drains the GPU queue (the next rollout's host work - sampling, de-duplication, launches - overlaps the update pass).
    Any access (``stats['kl']``, ``dict(stats)``, ``repr``) waits for that one event."""

    def __init__(self, trainer, token):
        self._trainer, self._token, self._values = trainer, token, None

    def _resolve(self):
        if self._values is None:
            self._values = self._trainer._resolve(self._token)
        return self._values

    def __getitem__(self, k):
        return self._resolve()[k]

    def __iter__(self):
        return iter(self._resolve())

    def __len__(self):
        return len(self._resolve())

    def __contains__(self, k):
        return k in self._resolve()

    def keys(self):
        return self._resolve().keys()

    def items(self):
        return self._resolve().items()

    def values(self):
        return self._resolve().values()

    def get(self, k, default=None):
        return self._resolve().get(k, default)

    def __repr__(self):
        return repr(self._resolve())

    def __eq__(self, other):
        return dict(self._resolve()) == (dict(other._resolve()) if isinstance(other, LazyStats) else other)

    def __reduce__(self):                   # pickles / torch.save as the plain dict of numbers
        return (dict, (dict(self._resolve()),))


class _DeferredStats(object):
    """Shared by the trainers: statistics of a train call are copied asynchronously and settled one call later (LazyStats)."""
    @property
    def kl_coeff(self):
        """PPO's adaptive KL coefficient AFTER every finished train call (waits for the last call's statistics)."""
        self._settle()
        return self._kl_coeff

    @kl_coeff.setter
    def kl_coeff(self, v):
        self._settle()
        self._kl_coeff = float(v)

    def _submit(self, mean_reward, stats, kl_div, extra):
        """Enqueue the device -> pinned copy of one train call's numbers [mean_reward, stats..., status word] and an event."""
        dev = stats.device
        status = (self.policy.status_words()[1:2].to(torch.float64) if hasattr(self.policy, 'status_words')
                  else torch.zeros(1, dtype=torch.float64, device=dev))
        overflow = rdist.take_row_overflow(dev)
        if overflow is not None:
            status = status + 1000.0 * overflow.to(torch.float64)
        vec = torch.cat([mean_reward.reshape(1).to(torch.float64), stats.reshape(-1).to(torch.float64), status])
        pin = torch.empty(vec.shape, dtype=torch.float64, pin_memory=True)
        pin.copy_(vec, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(dev))
        token = dict(pin=pin, ev=ev, kl_div=kl_div, values=None, iteration=self.iteration, **extra)
        self._pending.append(token)
        return token

    def _settle(self):
        """Finish every outstanding train call in order: wait for its numbers, validate the pass, apply RLlib's KL rule."""
        while self._pending:
            tok = self._pending.pop(0)
            tok['ev'].synchronize()
            v = tok['pin'].numpy().copy()
            if v[-1] >= 1000.0:
                raise RuntimeError(rdist.ROW_OVERFLOW_MESSAGE)
            if v[-1] != 0.0:
                self.policy.check_status()                  # clears the flag and raises
                raise RuntimeError(D.DevicePolicy.PASS_TIMEOUT_MESSAGE)
            s = v[1:-1]
            kl_mean = 0.0
            if tok['ppo']:
                kl_mean = float(s[7]) / tok['kl_div'] if tok['kl_mean'] is None else tok['kl_mean']
                if tok['kl_mean'] is None:
                    self._kl_coeff = update_kl_coeff(self._kl_coeff, kl_mean, self.kl_target)
            tok['values'] = {'episode_reward_mean': float(v[0]), 'policy_loss': float(s[0]), 'vf_loss': float(s[1]),
                             'entropy': float(s[2]), 'kl': float(s[3]), 'kl_mean': kl_mean, 'kl_coeff': self._kl_coeff,
                             'iteration': tok['iteration']}

    def _resolve(self, token):
        if token['values'] is None:
            self._settle()
        return token['values']

class RawStateTrainer(_DeferredStats):
    """The same loop for an env with config['rawstate_as_obs'] (+ support_rllib_mask, return_tensors): the policy is the
    raw-state encoder (rllib_rawstate_model.py; modelfree_train.py 'rawstate' variants) acting on the device tensors of the
    raw features; loss, backward and Adam run through rl4rs_rawtrain_*.

    Data-parallel: the gradient lives in the handle's flat buffer [cat_emb | seq_emb | dense layers]; the dense tail
    (~280 K floats) is mean-all-reduced in place, the two 100000 x 128 embedding tables exchange only the rows the minibatch
    touched (``dist.allreduce_rows_mean_``)."""

    def __init__(self, env, algo='A2C', seed=0, lr=1e-4, minibatch=256, weights=None, init_seed=0, kl_coeff=0.2, kl_target=0.01):
        from .nets.rawpolicy import init_rawpolicy_weights
        cfg = env.config
        assert cfg.get('return_tensors', False) and cfg.get('rawstate_as_obs', False) and cfg.get('support_rllib_mask', False), \
            "RawStateTrainer needs config rawstate_as_obs + support_rllib_mask + return_tensors"
        self.env = env
        self.algo = {'A2C': D.DeviceRawTrainer.A2C, 'PPO': D.DeviceRawTrainer.PPO}[algo]
        self.B, self.T, self.A = cfg['batch_size'], cfg['max_steps'], cfg['action_size']
        self.S, self.L = cfg['seq_num'], cfg['maxlen']
        self.seed, self.lr, self.minibatch = seed, lr, minibatch
        self._kl_coeff, self.kl_target = float(kl_coeff), float(kl_target)
        self._pending = []
        if weights is None:
            weights = init_rawpolicy_weights(cfg, seed=init_seed)
        N = self.B * self.T
        if algo == 'PPO' and N < minibatch:
            raise ValueError("PPO needs at least one full minibatch per train call: batch_size * max_steps = %d < minibatch = %d"
                             % (N, minibatch))
        self.policy = D.DeviceRawTrainer(cfg, weights, max_rows=max(N, self.B))
        self.iteration = 0
        dev = self.policy.device
        W = self.policy.W
        self.buf = dict(cat=torch.empty((N, cfg['category_feature_num']), dtype=torch.int32, device=dev),
                        dense=torch.empty((N, cfg['dense_feature_num']), dtype=torch.float32, device=dev),
                        seqs=[torch.empty((N, self.L), dtype=torch.int32, device=dev) for _ in range(self.S)],
                        mask=torch.empty((N, W), dtype=torch.int32, device=dev),
                        act=torch.empty(N, dtype=torch.int32, device=dev), logp=torch.empty(N, dtype=torch.float32, device=dev),
                        val=torch.empty(N, dtype=torch.float32, device=dev), rew=torch.empty(N, dtype=torch.float64, device=dev),
                        logits=torch.empty((N, self.A), dtype=torch.float32, device=dev))
        self._grad = None
        self._row_caps = None            # static distinct-row bounds of the sparse table exchange (dist.calibrate_row_cap)
        if rdist.collectives_active():
            rdist.broadcast_(self.policy.flat_view('params'), src=0)      # replicas start identical (Adam state is zero)
            self._grad = self.policy.flat_view('grad')

    def _mask_bits(self, mask):
        W = self.policy.W
        m = mask.to(torch.int32)
        pad = W * 32 - self.A
        if pad:
            m = torch.nn.functional.pad(m, (0, pad))
        w = (m.view(self.B, W, 32) << torch.arange(32, device=m.device, dtype=torch.int32)).sum(dim=2, dtype=torch.int64)
        return (w & 0xffffffff).to(torch.int32).contiguous()

    def rollout(self):
        B, T = self.B, self.T
        obs = self.env.reset()
        b = self.buf
        for t in range(T):
            sl = slice(t * B, (t + 1) * B)
            bits = self._mask_bits(obs['action_mask'])
            seqs = [q.contiguous() for q in obs['sequence_feature']]
            a, lp, v, ent, lg = self.policy.act(obs['category_feature'], obs['dense_feature'], seqs, bits, seed=self.seed,
                                                step=self.iteration * T + t, want_logits=self.algo == D.DeviceRawTrainer.PPO)
            b['cat'][sl] = obs['category_feature']
            b['dense'][sl] = obs['dense_feature']
            for s in range(self.S):
                b['seqs'][s][sl] = seqs[s]
            b['mask'][sl] = bits
            b['act'][sl] = a
            b['logp'][sl] = lp
            b['val'][sl] = v
            if lg is not None:
                b['logits'][sl] = lg
            obs, reward, done, info = self.env.step(a)
            b['rew'][sl] = reward
        ret = _returns(b['rew'], 1, T, B)
        return ret, ret - b['val'], b['rew'].view(T, B).sum(dim=0).mean()         # stays on the device (LazyStats)

    def _allreduce_gradient(self, cat, seqs):
        """Mean over ranks of the gradient the last loss_grad left in the handle (no-op on one rank)."""
        if self._grad is None:
            return
        g = self._grad
        tables = self.policy.table_rows()
        # ids clamped the way the kernels clamp them; allreduce_rows_mean_ finds the distinct rows itself with fixed-shape
        # device ops (torch.unique would drain the queue once per minibatch)
        flat = [cat.reshape(-1).to(torch.int64), torch.cat([q.reshape(-1) for q in seqs]).to(torch.int64)]
        if self._row_caps is None:
            # static distinct-row bounds, measured on the first minibatch and agreed across the ranks (start-up only)
            self._row_caps = [rdist.calibrate_row_cap(i.clamp(0, H - 1), H) for (off, H, E), i in zip(tables, flat)]
        for (off, H, E), i, cap in zip(tables, flat, self._row_caps):
            rdist.allreduce_rows_mean_(g[off:off + H * E].view(H, E), i.clamp(0, H - 1), cap=cap)
        tail = tables[-1][0] + tables[-1][1] * tables[-1][2]
        rdist.allreduce_mean_(g[tail:])

    def train_iteration(self):
        """One train call; returns a ``LazyStats`` mapping (see Trainer.train_iteration)."""
        ret, adv, mean_reward = self.rollout()
        self._settle()
        b = self.buf
        N = self.B * self.T
        self.iteration += 1
        if self.algo == D.DeviceRawTrainer.A2C:
            stats = self.policy.loss_grad(self.algo, b['cat'], b['dense'], b['seqs'], b['act'], adv, ret, mask_bits=b['mask'],
                                          vf_coeff=0.5, ent_coeff=0.01)
            self._allreduce_gradient(b['cat'], b['seqs'])
            self.policy.adam_step(lr=self.lr, grad_clip=10.0)
            return LazyStats(self, self._submit(mean_reward, stats[:4], 1, dict(ppo=False, kl_mean=None)))
        adv_n = (adv - adv.mean()) / adv.std().clamp_min(1e-4)
        perm = torch.from_numpy(np.random.RandomState(self.seed + self.iteration - 1).permutation(N)).to(b['cat'].device)
        sh = dict((k, b[k][perm]) for k in ('cat', 'dense', 'mask', 'act', 'logp', 'val', 'logits'))
        shs = [q[perm] for q in b['seqs']]
        adv_s, ret_s = adv_n[perm], ret[perm]
        kl_sum = torch.zeros((), dtype=torch.float32, device=b['cat'].device)
        nmb = 0
        for lo in range(0, N - self.minibatch + 1, self.minibatch):
            hi = lo + self.minibatch
            seq_mb = [q[lo:hi] for q in shs]
            stats = self.policy.loss_grad(self.algo, sh['cat'][lo:hi], sh['dense'][lo:hi], seq_mb, sh['act'][lo:hi],
                                          adv_s[lo:hi], ret_s[lo:hi], mask_bits=sh['mask'][lo:hi], old_logp=sh['logp'][lo:hi],
                                          old_value=sh['val'][lo:hi], old_logits=sh['logits'][lo:hi], vf_coeff=0.5,
                                          ent_coeff=0.0, clip=0.3, vf_clip=500.0, kl_coeff=self._kl_coeff)
            self._allreduce_gradient(sh['cat'][lo:hi], seq_mb)
            self.policy.adam_step(lr=self.lr)
            kl_sum += stats[3]
            nmb += 1
        stats8 = torch.cat([stats.reshape(-1)[:4].to(torch.float32), torch.zeros(3, dtype=torch.float32, device=kl_sum.device),
                            kl_sum.reshape(1)])
        if rdist.collectives_active():
            # every rank must take the same kl_coeff decision: the rule sees the mean over ALL ranks' samples.  The sum is
            # all-reduced ON THE DEVICE and travels with the deferred statistics: no rank reads a device scalar in a train call
            rdist.allreduce_sum_(stats8[7:8])
            return LazyStats(self, self._submit(mean_reward, stats8, max(nmb * self.minibatch, 1) * rdist.world_size(),
                                                dict(ppo=True, kl_mean=None)))
        return LazyStats(self, self._submit(mean_reward, stats8, max(nmb * self.minibatch, 1), dict(ppo=True, kl_mean=None)))


class Trainer(_DeferredStats):
    """A2C / PPO on the action-masked FC policy (rllib_mask_model.py:7-64) over the zero-copy discrete-action env.

    seed       sampling stream of this rank (Gumbel noise, PPO shuffle): pass a different value per rank
    init_seed  parameter initialisation, shared by all ranks (and rank 0's parameters are broadcast anyway)
    kl_coeff / kl_target   PPO's adaptive KL penalty (update_kl_coeff above)
    train_batch_size       timesteps per train call (rollouts_per_train_call above; the reference's value = one rollout)
    keep_last_batch        keep the shuffled tensors of the last iteration in ``last_batch`` (tests)"""

    def __init__(self, env, algo='A2C', hidden=64, seed=0, lr=1e-4, minibatch=256, init_seed=0, kl_coeff=0.2, kl_target=0.01,
                 train_batch_size=None, keep_last_batch=False):
        cfg = env.config
        assert cfg.get('return_tensors', False) and not cfg.get('support_conti_env', False), \
            "Trainer needs the zero-copy discrete-action env (config['return_tensors'] = True)"
        self.env = env
        self.algo = {'A2C': D.DevicePolicy.A2C, 'PPO': D.DevicePolicy.PPO}[algo]
        self.B, self.T, self.A = cfg['batch_size'], cfg['max_steps'], cfg['action_size']
        self.R = rollouts_per_train_call(self.B, self.T, train_batch_size)
        self.seed, self.lr, self.minibatch = seed, lr, minibatch
        self._kl_coeff, self.kl_target = float(kl_coeff), float(kl_target)
        self._pending = []          # train calls whose statistics have not been looked at yet (oldest first)
        self.keep_last_batch = keep_last_batch
        self.last_batch = None
        N = self.R * self.B * self.T
        if algo == 'PPO' and N < minibatch:
            # (N // minibatch == 0 would leave the data-parallel pass without a single step: undefined statistics, a division
            # by zero, ranks failing at different points around a collective)
            raise ValueError("PPO needs at least one full minibatch per train call: %d rollouts x batch_size x max_steps = %d < "
                             "minibatch = %d" % (self.R, N, minibatch))
        self.policy = D.DevicePolicy(256, hidden, self.A, max_rows=N, seed=init_seed)
        share = rdist.ranks_sharing_device()
        if share > 1:
            # several ranks on ONE GPU (gloo dry runs only): the persistent PPO pass may use its grid barrier only while the grids of
            # all the processes on the device fit together, one workgroup per CU; beyond that the per-minibatch kernels run
            n_cu = torch.cuda.get_device_properties(self.policy.device).multi_processor_count
            self.policy.set_option('resident_wgs', n_cu // share)
        self.iteration = 0
        self._rollouts = 0
        dev = self.policy.device
        self.buf = dict(obs=torch.empty((N, 256), dtype=torch.float32, device=dev),
                        mask=torch.empty((N, self.policy.W), dtype=torch.int32, device=dev),
                        act=torch.empty(N, dtype=torch.int32, device=dev),
                        logp=torch.empty(N, dtype=torch.float32, device=dev),
                        val=torch.empty(N, dtype=torch.float32, device=dev),
                        rew=torch.empty(N, dtype=torch.float64, device=dev),
                        logits=torch.empty((N, self.A), dtype=torch.float32, device=dev))
        self.grad = torch.empty(self.policy.n_params, dtype=torch.float32, device=dev)
        self._mb_stats = torch.empty(4, dtype=torch.float32, device=dev)
        if rdist.collectives_active():
            self.sync_replicas()

    def sync_replicas(self, src=0):
        """Make every rank's parameters, Adam moments and step counter those of rank ``src``."""
        p = self.policy.params()
        m, v, t = self.policy.adam_state()
        rdist.broadcast_(p, src)
        rdist.broadcast_(m, src)
        rdist.broadcast_(v, src)
        step = torch.tensor([t], dtype=torch.int64)
        if rdist.collectives_active():
            import torch.distributed as dist
            if dist.get_backend() != 'gloo':
                step = step.to(p.device)
            dist.broadcast(step, src=src)
        self.policy.set_params(p)
        self.policy.set_adam_state(m, v, int(step.item()))

    def params(self):
        """Flat parameters (validated: raises if a persistent pass was cut short)."""
        self._settle()
        self.policy.check_status()
        return self.policy.params()

    def close(self):
        try:
            self._settle()
            self.policy.check_status()
        finally:
            self.policy.close()

    def _mask_bits(self):
        """Packed obs-side action mask (action_mask & location_mask[layer] & special_mask, slate.py:92-97)."""
        return self.env.samples._live().obs_mask_bits()          # packed on the device: no dense [B, A] round trip

    def rollout(self):
        B, T, R = self.B, self.T, self.R
        b = self.buf
        ppo = self.algo == D.DevicePolicy.PPO
        for r in range(R):
            obs = self.env.reset()
            for t in range(T):
                obs_t = obs['obs'] if isinstance(obs, dict) else obs
                sl = slice((r * T + t) * B, (r * T + t + 1) * B)
                # mask, sampled actions, log-probs, values (and PPO's logits) are written straight into the rollout buffers
                mask = self.env.samples._live().obs_mask_bits(out=b['mask'][sl])
                b['obs'][sl] = obs_t
                a = self.policy.act(b['obs'][sl], mask, seed=self.seed, step=self._rollouts * T + t, want_logits=ppo,
                                    out=(b['act'][sl], b['logp'][sl], b['val'][sl], b['logits'][sl] if ppo else None))[0]
                obs, reward, done, info = self.env.step(a)
                b['rew'][sl] = reward
            self._rollouts += 1
        # gamma = 1, lambda = 1: advantage = (sum of future rewards) - V
        ret = _returns(b['rew'], R, T, B)
        adv = ret - b['val']
        return ret, adv, b['rew'].view(R, T, B).sum(dim=1).mean()        # mean episode reward stays on the device

    def train_iteration(self):
        """One train call.  Returns a ``LazyStats`` mapping (episode_reward_mean, policy_loss, vf_loss, entropy, kl, kl_mean,
        kl_coeff, iteration): nothing in here waits for the GPU on one rank - the previous call's numbers (needed for PPO's KL
        rule and to validate its pass) are collected after this call's rollout has been enqueued, when they are long there."""
        ret, adv, mean_reward = self.rollout()
        self._settle()                       # previous call: statistics, pass status, kl_coeff (the GPU is busy with the rollout)
        b = self.buf
        N = self.R * self.B * self.T
        world = rdist.world_size()
        self.iteration += 1
        if self.algo == D.DevicePolicy.A2C:
            g, stats = self.policy.loss_grad(self.algo, b['obs'], b['act'], adv, ret, mask_bits=b['mask'],
                                             vf_coeff=0.5, ent_coeff=0.01, grad_out=self.grad)
            if self.keep_last_batch:
                self.last_batch = dict(obs=b['obs'].clone(), act=b['act'].clone(), mask=b['mask'].clone(), adv=adv.clone(), ret=ret.clone())
            rdist.allreduce_mean_(g)                         # the ONE collective: flat policy gradient over RCCL
            self.policy.adam_step(g, lr=self.lr, grad_clip=10.0)
            return LazyStats(self, self._submit(mean_reward, stats[:4], 1, dict(ppo=False, kl_mean=None)))
        adv_n = (adv - adv.mean()) / adv.std().clamp_min(1e-4)      # RLlib standardises PPO advantages
        perm = torch.from_numpy(np.random.RandomState(self.seed + self.iteration - 1).permutation(N)).to(b['obs'].device)
        # shuffle every buffer ONCE (7 gathers per iteration instead of 7 per minibatch): minibatches are then
        # contiguous row ranges that go to the library as plain pointers
        sh = dict((k, b[k][perm]) for k in ('obs', 'act', 'mask', 'logp', 'val', 'logits'))
        adv_s, ret_s = adv_n[perm].contiguous(), ret[perm].contiguous()
        if self.keep_last_batch:
            self.last_batch = dict(sh, adv=adv_s, ret=ret_s, kl_coeff=self._kl_coeff)
        MB = self.minibatch
        nmb = N // MB
        if not rdist.collectives_active():
            # single GPU: the whole pass is one library call (no per-minibatch collective to interleave)
            stats = self.policy.ppo_epoch(sh['obs'], sh['act'], adv_s, ret_s, sh['mask'], sh['logp'], sh['val'],
                                          sh['logits'], minibatch=MB, vf_coeff=0.5, ent_coeff=0.0,
                                          clip=0.3, vf_clip=500.0, kl_coeff=self._kl_coeff, lr=self.lr, grad_out=self.grad)
            return LazyStats(self, self._submit(mean_reward, stats, nmb * MB, dict(ppo=True, kl_mean=None)))
        # data parallel: per minibatch ONE fused gradient launch, ONE all-reduce, ONE Adam launch
        kl_sum = torch.zeros((), dtype=torch.float32, device=b['obs'].device)
        for mb in range(nmb):
            g, stats = self.policy.ppo_minibatch_grad(mb, sh['obs'], sh['act'], adv_s, ret_s, sh['mask'], sh['logp'],
                                                      sh['val'], sh['logits'], minibatch=MB, vf_coeff=0.5, ent_coeff=0.0,
                                                      clip=0.3, vf_clip=500.0, kl_coeff=self._kl_coeff,
                                                      grad_out=self.grad, stats_out=self._mb_stats)
            rdist.allreduce_mean_(g)
            self.policy.adam_step(g, lr=self.lr)
            kl_sum += stats[3]
        # every rank must take the same kl_coeff decision: the rule sees the mean over ALL ranks' samples.  The KL sum is
        # all-reduced on the device and settles with the deferred statistics like the single-GPU path (ADVICE / VERDICT r3: this
        # path used to read kl_sum on the host - one queue drain per train call and rank)
        stats8 = torch.cat([stats.reshape(-1)[:4], torch.zeros(3, dtype=stats.dtype, device=stats.device), kl_sum.reshape(1).to(stats.dtype)])
        rdist.allreduce_sum_(stats8[7:8])
        return LazyStats(self, self._submit(mean_reward, stats8, nmb * MB * world, dict(ppo=True, kl_mean=None)))
