"""On-device policy training loop over the GPU env (A2C / PPO with the reference's RLlib hyper-parameters,
script/modelfree_train.py:179-304: gamma = 1, GAE lambda = 1, lr 1e-4, vf_loss_coeff 0.5; A2C entropy 0.01 and
grad_clip 10; PPO clip 0.3, kl_coeff 0.2, vf_clip 500, one SGD pass over minibatches of 256).

Rollout, policy forward/backward and Adam run through librl4rs_hip.so; torch is used for buffers, the reversed
cumulative sum of rewards and (data-parallel) the single RCCL all-reduce of the flat gradient buffer."""
import numpy as np
import torch

from . import dist as rdist
from . import device as D


class RawStateTrainer(object):
    """The same loop for an env with config['rawstate_as_obs'] (+ support_rllib_mask, return_tensors): the policy is the
    raw-state encoder (rllib_rawstate_model.py; modelfree_train.py 'rawstate' variants) acting on the device tensors of the
    raw features; loss, backward and Adam run through rl4rs_rawtrain_*."""

    def __init__(self, env, algo='A2C', seed=0, lr=1e-4, minibatch=256, weights=None):
        from .nets.rawpolicy import init_rawpolicy_weights
        cfg = env.config
        assert cfg.get('return_tensors', False) and cfg.get('rawstate_as_obs', False) and cfg.get('support_rllib_mask', False), \
            "RawStateTrainer needs config rawstate_as_obs + support_rllib_mask + return_tensors"
        self.env = env
        self.algo = {'A2C': D.DeviceRawTrainer.A2C, 'PPO': D.DeviceRawTrainer.PPO}[algo]
        self.B, self.T, self.A = cfg['batch_size'], cfg['max_steps'], cfg['action_size']
        self.S, self.L = cfg['seq_num'], cfg['maxlen']
        self.seed, self.lr, self.minibatch = seed, lr, minibatch
        if weights is None:
            weights = init_rawpolicy_weights(cfg, seed=seed)
        N = self.B * self.T
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
        rew = b['rew'].view(T, B)
        ret = torch.flip(torch.cumsum(torch.flip(rew, dims=[0]), dim=0), dims=[0]).reshape(-1).to(torch.float32)
        return ret, ret - b['val'], float(rew.sum(dim=0).mean().item())

    def train_iteration(self):
        ret, adv, mean_reward = self.rollout()
        b = self.buf
        N = self.B * self.T
        if self.algo == D.DeviceRawTrainer.A2C:
            stats = self.policy.loss_grad(self.algo, b['cat'], b['dense'], b['seqs'], b['act'], adv, ret, mask_bits=b['mask'],
                                          vf_coeff=0.5, ent_coeff=0.01)
            self.policy.adam_step(lr=self.lr, grad_clip=10.0)
        else:
            adv_n = (adv - adv.mean()) / adv.std().clamp_min(1e-4)
            perm = torch.from_numpy(np.random.RandomState(self.seed + self.iteration).permutation(N)).to(b['cat'].device)
            sh = dict((k, b[k][perm]) for k in ('cat', 'dense', 'mask', 'act', 'logp', 'val', 'logits'))
            shs = [q[perm] for q in b['seqs']]
            adv_s, ret_s = adv_n[perm], ret[perm]
            for lo in range(0, N - self.minibatch + 1, self.minibatch):
                hi = lo + self.minibatch
                stats = self.policy.loss_grad(self.algo, sh['cat'][lo:hi], sh['dense'][lo:hi], [q[lo:hi] for q in shs], sh['act'][lo:hi],
                                              adv_s[lo:hi], ret_s[lo:hi], mask_bits=sh['mask'][lo:hi], old_logp=sh['logp'][lo:hi],
                                              old_value=sh['val'][lo:hi], old_logits=sh['logits'][lo:hi], vf_coeff=0.5,
                                              ent_coeff=0.0, clip=0.3, vf_clip=500.0, kl_coeff=0.2)
                self.policy.adam_step(lr=self.lr)
        self.iteration += 1
        s = stats.cpu().numpy()
        return {'episode_reward_mean': mean_reward, 'policy_loss': float(s[0]), 'vf_loss': float(s[1]), 'entropy': float(s[2]),
                'kl': float(s[3]), 'iteration': self.iteration}


class Trainer(object):
    def __init__(self, env, algo='A2C', hidden=64, seed=0, lr=1e-4, minibatch=256):
        cfg = env.config
        assert cfg.get('return_tensors', False) and not cfg.get('support_conti_env', False), \
            "Trainer needs the zero-copy discrete-action env (config['return_tensors'] = True)"
        self.env = env
        self.algo = {'A2C': D.DevicePolicy.A2C, 'PPO': D.DevicePolicy.PPO}[algo]
        self.B, self.T, self.A = cfg['batch_size'], cfg['max_steps'], cfg['action_size']
        self.seed, self.lr, self.minibatch = seed, lr, minibatch
        self.policy = D.DevicePolicy(256, hidden, self.A, max_rows=self.B * self.T, seed=seed)
        self.iteration = 0
        dev = self.policy.device
        N = self.B * self.T
        self.buf = dict(obs=torch.empty((N, 256), dtype=torch.float32, device=dev),
                        mask=torch.empty((N, self.policy.W), dtype=torch.int32, device=dev),
                        act=torch.empty(N, dtype=torch.int32, device=dev),
                        logp=torch.empty(N, dtype=torch.float32, device=dev),
                        val=torch.empty(N, dtype=torch.float32, device=dev),
                        rew=torch.empty(N, dtype=torch.float64, device=dev),
                        logits=torch.empty((N, self.A), dtype=torch.float32, device=dev))
        self.grad = torch.empty(self.policy.n_params, dtype=torch.float32, device=dev)

    def _mask_bits(self):
        """Packed obs-side action mask (action_mask & location_mask[layer] & special_mask, slate.py:92-97)."""
        return self.env.samples._live().obs_mask_bits()          # packed on the device: no dense [B, A] round trip

    def rollout(self):
        B, T = self.B, self.T
        obs = self.env.reset()
        b = self.buf
        for t in range(T):
            obs_t = obs['obs'] if isinstance(obs, dict) else obs
            sl = slice(t * B, (t + 1) * B)
            # mask, sampled actions, log-probs, values (and PPO's logits) are written straight into the rollout buffers
            mask = self.env.samples._live().obs_mask_bits(out=b['mask'][sl])
            b['obs'][sl] = obs_t
            ppo = self.algo == D.DevicePolicy.PPO
            a = self.policy.act(b['obs'][sl], mask, seed=self.seed, step=self.iteration * T + t, want_logits=ppo,
                                out=(b['act'][sl], b['logp'][sl], b['val'][sl], b['logits'][sl] if ppo else None))[0]
            obs, reward, done, info = self.env.step(a)
            b['rew'][sl] = reward
        # gamma = 1, lambda = 1: advantage = (sum of future rewards) - V
        rew = b['rew'].view(T, B)
        ret = torch.flip(torch.cumsum(torch.flip(rew, dims=[0]), dim=0), dims=[0]).reshape(-1).to(torch.float32)
        adv = ret - b['val']
        return ret, adv, float(rew.sum(dim=0).mean().item())

    def train_iteration(self):
        ret, adv, mean_reward = self.rollout()
        b = self.buf
        N = self.B * self.T
        stats_out = None
        if self.algo == D.DevicePolicy.A2C:
            g, stats = self.policy.loss_grad(self.algo, b['obs'], b['act'], adv, ret, mask_bits=b['mask'],
                                             vf_coeff=0.5, ent_coeff=0.01, grad_out=self.grad)
            rdist.allreduce_mean_(g)                         # the ONE collective: flat policy gradient over RCCL
            self.policy.adam_step(g, lr=self.lr, grad_clip=10.0)
            stats_out = stats
        else:
            adv_n = (adv - adv.mean()) / adv.std().clamp_min(1e-4)      # RLlib standardises PPO advantages
            perm = torch.from_numpy(np.random.RandomState(self.seed + self.iteration).permutation(N)).to(b['obs'].device)
            # shuffle every buffer ONCE (7 gathers per iteration instead of 7 per minibatch): minibatches are then
            # contiguous row ranges that go to the library as plain pointers
            sh = dict((k, b[k][perm]) for k in ('obs', 'act', 'mask', 'logp', 'val', 'logits'))
            adv_s, ret_s = adv_n[perm], ret[perm]
            if rdist.world_size() == 1:
                # single GPU: the whole pass is one library call (no per-minibatch collective to interleave)
                stats_out = self.policy.ppo_epoch(sh['obs'], sh['act'], adv_s, ret_s, sh['mask'], sh['logp'], sh['val'],
                                                  sh['logits'], minibatch=self.minibatch, vf_coeff=0.5, ent_coeff=0.0,
                                                  clip=0.3, vf_clip=500.0, kl_coeff=0.2, lr=self.lr, grad_out=self.grad)
            for lo in (range(0, N - self.minibatch + 1, self.minibatch) if rdist.world_size() > 1 else ()):
                hi = lo + self.minibatch
                g, stats = self.policy.loss_grad(self.algo, sh['obs'][lo:hi], sh['act'][lo:hi], adv_s[lo:hi], ret_s[lo:hi],
                                                 mask_bits=sh['mask'][lo:hi], old_logp=sh['logp'][lo:hi],
                                                 old_value=sh['val'][lo:hi], old_logits=sh['logits'][lo:hi],
                                                 vf_coeff=0.5, ent_coeff=0.0, clip=0.3, vf_clip=500.0, kl_coeff=0.2,
                                                 grad_out=self.grad)
                rdist.allreduce_mean_(g)
                self.policy.adam_step(g, lr=self.lr)
                stats_out = stats
        self.iteration += 1
        s = stats_out.cpu().numpy()
        return {'episode_reward_mean': mean_reward, 'policy_loss': float(s[0]), 'vf_loss': float(s[1]),
                'entropy': float(s[2]), 'kl': float(s[3]), 'iteration': self.iteration}
