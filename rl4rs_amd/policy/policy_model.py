"""``policy_model`` of the reference (rl4rs/policy/policy_model.py:8-92) over this package's device learners: the object
``script/batchrl_trainer.py:377-411`` (`evaluate`) and ``script/batchrl_train.py:131-134`` drive an env with.

    policy = policy_model(model, config=config)          # model: offline_rl.DiscreteBC / DiscreteBCQ / DiscreteCQL / BCQ
    action = policy.predict_with_mask(obs)               # obs: the d3rl-mode observation [B, 256 + page_items + 1]
    obs, reward, done, info = env.step(action)

* continuous env (``config['support_conti_env']``): ``predict_with_mask`` IS ``predict`` (policy_model.py:18-19) - the learner's
  32-d embedding, which the env's K-NN resolves under its own masks;
* discrete learners: ``action_probs`` (imitator logits for DiscreteBC, softmax of the Q values otherwise, policy_model.py:75-83)
  masked by the rule the observation tail encodes (previous actions | cur_step -> location_mask row, chosen items, special items;
  :22-33, fill value -2**15) and arg-maxed - on the device (``rl4rs_env_predict_with_mask``), first maximum like numpy.
Observations and results stay device tensors when they come in as device tensors; numpy in -> numpy out like the reference.
The RLlib-trainer branch of the reference (``compute_actions``) has no counterpart here: the on-device trainers act through
``train.Trainer``."""
import numpy as np
import torch

from .. import device as D
from .. import offline_rl as R


class policy_model(object):
    def __init__(self, model, config={}, env=None):
        self.policy = model
        self.config = config
        self.page_items = int(config.get('page_items', 9))
        self.mask_size = self.page_items + 1
        self.location_mask = config.get('location_mask', None)
        self.special_items = config.get('special_items', None)
        self._env = env            # a RecEnvBase of this package: its device handle carries the catalogue masks
        self._mask_handle = None

    def _handle(self):
        if self._mask_handle is None:
            if self._env is not None:
                self._mask_handle = self._env.samples._live() if hasattr(self._env, 'samples') else self._env
            else:
                # the masks come from the catalogue file like SlateState.get_mask_from_file (batchrl_train.py:34-38)
                from ..data import CatalogTables
                cfg = dict(self.config, batch_size=1)
                tab = CatalogTables(cfg['iteminfo_file'], int(cfg['action_size']), int(cfg.get('action_emb_size', 32)))
                self._mask_handle = D.DeviceEnv(cfg, tab, False, int(cfg.get('max_steps', 9)), True)
        return self._mask_handle

    @staticmethod
    def _out(t, like):
        return t if isinstance(like, torch.Tensor) and like.is_cuda else t.cpu().numpy()

    def _obs(self, obs):
        dev = getattr(self.policy, 'device', None) or self.policy.nets[0].device
        return D._dev_tensor(np.asarray(obs) if not isinstance(obs, torch.Tensor) else obs, torch.float32, dev)

    def predict_with_mask(self, obs):
        if self.config.get('support_conti_env', False):
            return self.predict(obs)
        if not isinstance(self.policy, R._Learner):
            raise NotImplementedError('policy_model: %r is not one of this package\'s offline learners' % type(self.policy).__name__)
        x = self._obs(obs)
        scores = self._action_probs(x)
        a = self._handle().predict_with_mask(scores, x[:, -self.mask_size:])
        return self._out(a.to(torch.int64), obs)

    def _rows(self):
        """rows one call of the learner takes (the discrete learners' networks are sized for a minibatch)"""
        return self.policy.batch_size if isinstance(self.policy, R._Learner) else 1 << 30

    def predict(self, obs):
        x = self._obs(obs)
        n = self._rows()
        out = [self.policy.predict(x[lo:lo + n].contiguous()) for lo in range(0, x.shape[0], n)]
        return self._out(out[0] if len(out) == 1 else torch.cat(out), obs)

    def predict_q(self, obs, action):
        """AlgoBase.predict_value(obs, action), back on the reward's own scale when the learner has a reward scaler
        (policy_model.py:55-61; 'CQL-conti' is built with reward_scaler='standard', batchrl_trainer.py:91-107)."""
        x = self._obs(obs)
        a = torch.as_tensor(np.asarray(action) if not isinstance(action, torch.Tensor) else action).to(x.device)
        n = self._rows()
        out = [self.policy.predict_value(x[lo:lo + n].contiguous(), a[lo:lo + n].contiguous()) for lo in range(0, x.shape[0], n)]
        q = out[0] if len(out) == 1 else torch.cat(out)
        scaler = getattr(self.policy, 'reward_scaler', None)
        if isinstance(scaler, str):
            raise ValueError("reward_scaler=%r has not been fitted (fit_mdp fits a scaler given by name on its dataset): the critics "
                             "have no reward scale to undo yet" % scaler)
        if scaler is not None:
            q = scaler.reverse_transform(q)
        return self._out(q, obs)

    def _action_probs(self, x):
        if isinstance(self.policy, R.DiscreteBC):
            return self._chunks(self.policy.imitator, x)
        return torch.softmax(self._chunks(self.policy.q, x), dim=1)

    @staticmethod
    def _chunks(net, x):
        out = [net.forward(x[lo:lo + net.max_rows].contiguous()) for lo in range(0, x.shape[0], net.max_rows)]
        return out[0] if len(out) == 1 else torch.cat(out)

    def action_probs(self, obs):
        return self._out(self._action_probs(self._obs(obs)), obs)
