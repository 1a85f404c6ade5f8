"""Step order and reward rules.  ORACLE — test infrastructure only (see oracle/__init__.py).

``reward_from_probs`` follows ``SlateRecEnv.forward`` (rl4rs/env/slate.py:281-308) and
``SeqSlateRecEnv.forward`` (rl4rs/env/seqslate.py:136-160); ``OracleEnv.step`` follows
``RecSimBase._step`` (rl4rs/env/base.py:157-170) and ``RecEnvBase.step/reset`` (base.py:256-269).
"""
import numpy as np

from .state import OracleState


def is_reward_step(state):
    if state.seq:
        return state.cur_steps % state.page_items == 0        # seqslate.py:138
    return state.cur_steps >= state.max_steps                 # slate.py:283


def reward_from_probs(state, probs):
    """probs: [B, P] float32 click probabilities (res[:,1]) for the rows of complete_features()."""
    cfg = state.config
    B = state.batch_size
    if not is_reward_step(state):
        return [0] * B
    if state.seq:
        P = state.page_items
        price = state.get_price(state.prev_actions[:, :state.cur_steps])[:, -P:]   # seqslate.py:147
        reward = np.sum(price * np.asarray(probs).reshape(B, P), axis=1)           # seqslate.py:151-153
        if cfg.get("support_rllib_mask", False) or cfg.get("support_d3rl_mask", False):
            reward[state.get_violation() < 0.5] = 0                               # seqslate.py:154-157
    else:
        price = state.get_price(state.prev_actions)                                # slate.py:294
        reward = np.sum(price * np.asarray(probs).reshape(state.prev_actions.shape), axis=1)
        reward[state.get_violation() < 0.5] = 0                                   # slate.py:303-307
    return reward.tolist()


class OracleEnv(object):
    """Batched env with a pluggable scorer: scorer.obs(seq,dense,cat)->[R,256], scorer.prob(...)->[R]."""

    def __init__(self, config, records, scorer, seq=False, catalog=None):
        self.config = config
        self.seq = seq
        self.scorer = scorer
        self.catalog = catalog
        self.records = list(records)
        self.batch_size = config["batch_size"]
        self.max_steps = config["max_steps"]
        self.reset()

    def reset(self, records=None):
        if records is not None:
            self.records = list(records)
        self.cur_step = 0
        self.samples = OracleState(self.config, self.records, seq=self.seq, catalog=self.catalog)
        self.catalog = self.samples.cat
        return self._obs()

    def _obs(self):
        """SlateRecEnv.obs_fn (slate.py:244-279) without the list-of-dict packaging."""
        seq, dense, cat = self.samples.features()
        out = {}
        if self.config.get("rawstate_as_obs", False):
            out.update(sequence_feature=seq, dense_feature=dense, category_feature=cat)
        else:
            out['obs'] = self.scorer.obs(seq, dense, cat)
        if self.config.get("support_rllib_mask", False):
            out['action_mask'] = self.samples.obs_action_mask()
        elif self.config.get("support_d3rl_mask", False):
            out['masked_actions'], out['cur_steps'] = self.samples.masked_actions()
        return out

    def step(self, action):
        st = self.samples
        chosen = st.act(action)
        obs = self._obs()
        if is_reward_step(st):
            seq, dense, cat = st.complete_features()
            probs = self.scorer.prob(seq, dense, cat)
            reward = reward_from_probs(st, probs)
        else:
            reward = [0] * self.batch_size
        done = [0 if self.cur_step < self.max_steps - 1 else 1] * self.batch_size
        self.cur_step += 1
        return obs, reward, done, chosen
