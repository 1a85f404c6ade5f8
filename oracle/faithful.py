"""The reference's env loop at its own granularity - one python iteration per sample - for bench.py's ``cpu_baseline``
"faithful" leg ("faithful mode mirroring the reference's per-sample loop structure on 1 core").
ORACLE - test infrastructure only (see oracle/__init__.py).

``oracle/state.py`` is a VECTORISED restatement (what a careful numpy port would be); the reference itself walks the batch
in python: ``SlateState.act`` loops ``for i in range(batch_size)`` twice (masks, then the per-sample state rebuild,
rl4rs/env/slate.py:198-213), ``get_complete_states`` loops slots x samples with a fresh copy of the init state per slot
(:117-131), ``get_violation`` loops samples (:133-147), and ``FeatureUtil.feature_extraction`` pads every row on its own
(rl4rs/utils/datautil.py:34-69).  This module restates exactly that control flow (own code, same order of operations and
python-level data structures: nested lists per sample) so that its single-core rate is comparable with the reference's
measured 1.9-3.0 k env-steps/s without the net; it is checked bit for bit against the reference's golden
vectors in tests/test_oracle_golden.py.  Slate only (the bench workload); the scorer is pluggable as in oracle/env.py."""
import copy

import numpy as np

from .catalog import Catalog
from .records import ParsedRecords, pad_sequences


class FaithfulSlateEnv(object):
    def __init__(self, config, records, scorer, catalog=None):
        self.config = config
        self.records = list(records)
        self.scorer = scorer
        self.B = config['batch_size']
        self.A = config['action_size']
        self.T = config['max_steps']
        self.cat = catalog if catalog is not None else Catalog(config['iteminfo_file'], self.A)
        self.reset()

    # ---- SlateState.__init__ + records_to_state (slate.py:9-26, 67-83): nested python lists per sample
    def reset(self):
        p = ParsedRecords(self.records)
        self.parsed = p
        self.init_state = []
        for i in range(self.B):
            self.init_state.append([0, [list(p.history[i]), [0]], list(p.user_dense[i]), list(p.user_cat[i]), [0] * 9, 0])
        self.state = copy.deepcopy(self.init_state)
        self.prev_actions = np.full((self.B, self.T), 0)
        self.action_mask = np.full((self.B, self.A), 1, dtype=np.int64)
        self.special_mask = np.full((self.B, self.A), 1, dtype=np.int64)
        self.cur_steps = 0
        self.cur_step = 0
        return self._obs()

    # ---- feature_extraction (datautil.py:34-69): every row padded on its own
    def _features(self, state):
        L, Dn, Cn = self.config['maxlen'], self.config['dense_feature_num'], self.config['category_feature_num']
        seqs, dense, cat = [], [], []
        for row in state:
            seqs.append([pad_sequences([s], L)[0] for s in row[1]])
            d = np.zeros((Dn,), dtype=np.float32)
            n = min(len(row[2]), Dn)
            d[:n] = np.asarray(row[2][:n], dtype=np.float32)
            c = np.zeros((Cn,), dtype=np.int32)
            n = min(len(row[3]), Cn)
            c[:n] = np.asarray(row[3][:n], dtype=np.int32)
            dense.append(d)
            cat.append(c)
        return np.asarray(seqs, dtype=np.int32), np.asarray(dense), np.asarray(cat)

    def _obs(self):
        seq, dense, cat = self._features(self.state)
        return self.scorer.obs(seq, dense, cat)

    # ---- SlateState.act (slate.py:193-214): per-sample loops
    def act(self, actions):
        item_vec = self.cat.item_vec
        for i in range(self.B):
            a = int(actions[i])
            self.prev_actions[i][self.cur_steps] = a
            self.action_mask[i][a] = 0
            if len(np.intersect1d(self.prev_actions[i], self.cat.special_items)) > 0:
                self.special_mask[i][self.cat.special_items] = 0
        state = copy.deepcopy(self.init_state)
        for i in range(self.B):
            a = int(actions[i])
            row = state[i]
            feats = []
            for j in range(self.T):
                feats = feats + list(item_vec[self.prev_actions[i][j]])
            row[2] = row[2] + feats + list(item_vec[a])
            row[3] = row[3] + [1] + [int(x) for x in self.prev_actions[i]] + [a]
        self.state = state
        self.cur_steps += 1

    # ---- get_complete_states (slate.py:117-131): slots x samples, a fresh copy of the init state per slot
    def complete_states(self):
        item_vec = self.cat.item_vec
        rows = []
        for j in range(self.T):
            state = copy.deepcopy(self.init_state)
            for i in range(self.B):
                a = int(self.prev_actions[i][j])
                row = state[i]
                feats = []
                for k in range(self.T):
                    feats = feats + list(item_vec[self.prev_actions[i][k]])
                row[2] = row[2] + feats + list(item_vec[a])
                row[3] = row[3] + [1] + [int(x) for x in self.prev_actions[i]] + [a]
            rows.append(state)
        # env-major [B * T] as SlateRecEnv.forward reshapes them (slate.py:289-293)
        return [rows[j][i] for i in range(self.B) for j in range(self.T)]

    # ---- get_violation (slate.py:133-147)
    def violation(self):
        out = np.ones((self.B,), dtype=np.int64)
        loc = self.cat.location_mask
        for i in range(self.B):
            pa = self.prev_actions[i]
            ok = 1
            for step in range(self.cur_steps):
                ok = ok & int(loc[step // 3][pa[step]])
            for step in range(max(self.cur_steps - 1, 1)):
                ok = ok & int(pa[step] != pa[step + 1])
            for step in range(max(self.cur_steps - 2, 1)):
                ok = ok & int(pa[step] != pa[step + 2])
            if len(np.unique(pa[self.cat.is_special[pa]])) > 1:
                ok = 0
            out[i] = ok
        return out

    @property
    def offline_action(self):
        if self.cur_steps < self.T:
            return [x[self.cur_steps] for x in self.parsed.exposed]
        return [0] * self.B

    # ---- RecSimBase._step + SlateRecEnv.forward (base.py:157-170, slate.py:281-308)
    def step(self, actions):
        self.act(actions)
        obs = self._obs()
        if self.cur_steps >= self.T:
            seq, dense, cat = self._features(self.complete_states())
            probs = np.asarray(self.scorer.prob(seq, dense, cat)).reshape(self.B, self.T)
            price = self.cat.price[self.prev_actions]
            reward = np.sum(price * probs, axis=1)
            reward[self.violation() < 0.5] = 0
            reward = reward.tolist()
        else:
            reward = [0] * self.B
        done = [0 if self.cur_step < self.T - 1 else 1] * self.B
        self.cur_step += 1
        return obs, reward, done, {}
