"""Vectorised numpy restatement of SlateState / SeqSlateState.

ORACLE — test infrastructure only (see oracle/__init__.py).  Pinned against golden vectors captured
from the reference classes themselves (tests/golden/make_golden.py).

Reference: ``rl4rs/env/slate.py:8-214`` (SlateState), ``rl4rs/env/seqslate.py:8-126`` (SeqSlateState),
``rl4rs/utils/datautil.py:34-69`` (feature_extraction).
"""
import numpy as np

from .catalog import Catalog
from .records import ParsedRecords, pad_sequences


def nearest_neighbor(actions, action_emb):
    """slate.py:180-184"""
    score = np.einsum('ij,kj->ik', np.array(actions), action_emb)
    return np.argmax(score, axis=1)


def nearest_neighbor_with_mask(actions, action_emb, action_mask):
    """slate.py:186-191: float64 scores, masked entries = -2**31, first max wins."""
    score = np.einsum('ij,kj->ik', np.array(actions), action_emb)
    score[action_mask < 0.5] = -2 ** 31
    return np.argmax(score, axis=1)


class OracleState(object):
    """One object for both envs; ``seq=True`` selects the SeqSlateState rules."""

    def __init__(self, config, records, seq=False, catalog=None):
        self.config = config
        self.seq = seq
        self.records = list(records)
        self.batch_size = config["batch_size"]
        self.action_size = config["action_size"]
        self.action_emb_size = config.get("action_emb_size", 32)
        self.max_steps = config["max_steps"]
        self.page_items = config.get("page_items", 9)
        self.maxlen = config["maxlen"]
        self.dense_feature_num = config["dense_feature_num"]
        self.category_feature_num = config["category_feature_num"]
        B, A, T = self.batch_size, self.action_size, self.max_steps
        self.cat = catalog if catalog is not None else Catalog(config["iteminfo_file"], A)
        self.action_emb = self.cat.action_emb
        if config.get('support_onehot_action', False):        # slate.py:22-25
            self.action_emb_size = A
            self.action_emb = np.eye(A)
        self.location_mask = self.cat.location_mask
        self.special_items = self.cat.special_items
        self.parsed = ParsedRecords(self.records)
        assert len(self.records) == B
        # slate.py:16-19
        self.prev_actions = np.full((B, T), 0)
        self.action_mask = np.full((B, A), 1, dtype=np.int64)
        self.special_mask = np.full((B, A), 1, dtype=np.int64)
        self.cur_steps = 0
        self.infos = [{} for _ in range(B)]
        # _state = copy(_init_state) (base.py:30-31): nothing appended yet
        self._acted = False
        self._last_action = None
        self._act_step = 0
        # history sequence is constant (slate.py:77): pre-pad / pre-truncate to maxlen (datautil.py:43-46)
        self._seq0 = pad_sequences(self.parsed.history, self.maxlen)

    # ------------------------------------------------------------------ act
    def _layer(self):
        if self.seq:
            return self.cur_steps % self.page_items // 3     # seqslate.py:94-95
        return self.cur_steps // 3                           # slate.py:195

    def act(self, actions):
        B = self.batch_size
        if self.config.get("support_conti_env", False):
            location_mask = self.location_mask[self._layer()][np.newaxis, :]
            mask = self.action_mask & location_mask & self.special_mask
            actions = nearest_neighbor_with_mask(actions, self.action_emb, mask)
        actions = np.asarray(actions)
        self.prev_actions[:, self.cur_steps] = actions       # slate.py:198
        self.action_mask[np.arange(B), actions] = 0          # slate.py:199
        # slate.py:200-202 / seqslate.py:100-102: whole prev_actions row is tested
        hit = self.cat.is_special[self.prev_actions].any(axis=1)
        if len(self.special_items) > 0:
            rows = np.nonzero(hit)[0]
            self.special_mask[np.ix_(rows, self.special_items)] = 0
        self._acted = True
        self._last_action = actions.copy()
        self._act_step = self.cur_steps
        self.cur_steps += 1
        if self.seq and self.cur_steps % self.page_items == 0:   # seqslate.py:124-126
            self.action_mask = np.full((B, self.action_size), 1, dtype=np.int64)
            self.special_mask = np.full((B, self.action_size), 1, dtype=np.int64)
        return actions

    # ------------------------------------------------------------- features
    def _rows(self, page_slice, action, sequence_id):
        """dense/category rows before pad/truncate (slate.py:205-212 / seqslate.py:111-122)."""
        B = page_slice.shape[0]
        iv = self.cat.item_vec
        dense = np.concatenate([
            self.parsed.user_dense,
            iv[page_slice].reshape(B, -1),
            iv[action],
        ], axis=1)
        cat = np.concatenate([
            self.parsed.user_cat,
            np.full((B, 1), sequence_id, dtype=np.int64),
            page_slice.astype(np.int64),
            np.asarray(action, dtype=np.int64)[:, None],
        ], axis=1)
        return dense, cat

    def _fit(self, dense, cat):
        """post-pad / post-truncate to the configured widths (datautil.py:52-65)."""
        B = dense.shape[0]
        d = np.zeros((B, self.dense_feature_num), dtype=np.float32)
        n = min(dense.shape[1], self.dense_feature_num)
        d[:, :n] = dense[:, :n].astype(np.float32)
        c = np.zeros((B, self.category_feature_num), dtype=np.int32)
        n = min(cat.shape[1], self.category_feature_num)
        c[:, :n] = cat[:, :n].astype(np.int32)
        return d, c

    def _seq1(self, page_init):
        """second sequence: [0] in Slate (slate.py:77); items of previous pages in Seq (seqslate.py:107-108)."""
        B = self.batch_size
        if self.seq and page_init > 0:
            return pad_sequences(list(self.prev_actions[:, :page_init]), self.maxlen)
        return np.zeros((B, self.maxlen), dtype=np.int32)

    def _state_rows(self, step, action):
        if self.seq:
            P = self.page_items
            page_init = step // P * P
            page = self.prev_actions[:, page_init:page_init + P]
            dense, cat = self._rows(page, action, step // P + 1)
            seq1 = self._seq1(page_init)
        else:
            dense, cat = self._rows(self.prev_actions, action, 1)
            seq1 = self._seq1(0)
        d, c = self._fit(dense, cat)
        seq = np.stack([self._seq0, seq1], axis=1).astype(np.int32)
        return seq, d, c

    def features(self):
        """feature_extraction(self._state) -> (seq[B,2,L] i32, dense[B,Dn] f32, cat[B,Cn] i32)."""
        if not self._acted:
            d, c = self._fit(self.parsed.user_dense, self.parsed.user_cat)
            seq = np.stack([self._seq0, self._seq1(0)], axis=1).astype(np.int32)
            return seq, d, c
        return self._state_rows(self._act_step, self._last_action)

    def complete_features(self):
        """Rows fed to the reward net, env-major [B*P] (slate.py:117-131,289-294; seqslate.py:27-50,141-146)."""
        B = self.batch_size
        if self.seq:
            js = list(range(self.cur_steps))[-self.page_items:]
        else:
            js = list(range(self.max_steps))
        per = [self._state_rows(j, self.prev_actions[:, j]) for j in js]
        n = len(js)
        seq = np.stack([p[0] for p in per], axis=1).reshape(B * n, 2, self.maxlen)
        dense = np.stack([p[1] for p in per], axis=1).reshape(B * n, -1)
        cat = np.stack([p[2] for p in per], axis=1).reshape(B * n, -1)
        return seq, dense, cat

    # ---------------------------------------------------------------- masks
    def obs_action_mask(self):
        """state['action_mask'] (slate.py:92-97 / seqslate.py:15-17): uses POST-increment cur_steps."""
        layer = self.cur_steps % self.page_items // 3 if self.seq else self.cur_steps // 3
        return self.action_mask & self.location_mask[layer][np.newaxis, :] & self.special_mask

    def masked_actions(self):
        """state['masked_actions'], state['cur_steps'] for d3rl mode (slate.py:98-104 / seqslate.py:18-23)."""
        cur = np.full((self.batch_size, 1), self.cur_steps)
        if self.seq:
            P = self.page_items
            page_init = self.cur_steps // P * P
            page_end = min(page_init + P - 1, self.max_steps - 1)
            return self.prev_actions[:, page_end + 1 - P:page_end + 1], cur
        return self.prev_actions, cur

    # ------------------------------------------------------ reward helpers
    def get_price(self, actions):
        return self.cat.price[np.asarray(actions)]          # slate.py:112-115

    def get_violation(self):
        """slate.py:133-147 / seqslate.py:52-69."""
        B = self.batch_size
        P = self.page_items
        tmp = np.ones((B,), dtype=np.int64)
        pa = self.prev_actions
        for step in range(self.cur_steps):
            layer = step % P // 3 if self.seq else step // 3
            tmp = tmp & self.location_mask[layer][pa[:, step]]
        for step in range(max(self.cur_steps - 1, 1)):
            tmp = tmp & (pa[:, step] != pa[:, step + 1])
        for step in range(max(self.cur_steps - 2, 1)):
            tmp = tmp & (pa[:, step] != pa[:, step + 2])
        sp = self.cat.is_special
        for i in range(B):
            if self.seq:
                for j in range(self.cur_steps % P + 1):
                    acts = pa[i][P * j:P * (j + 1)]
                    if len(np.unique(acts[sp[acts]])) > 1:
                        tmp[i] = 0
            else:
                if len(np.unique(pa[i][sp[pa[i]]])) > 1:
                    tmp[i] = 0
        return tmp

    # ----------------------------------------------------- logged policy
    @property
    def offline_action(self):
        """slate.py:149-162"""
        cur = self.cur_steps
        conti = self.config.get("support_conti_env", False)
        if cur < self.max_steps:
            ids = [x[cur] for x in self.parsed.exposed]
            return [self.action_emb[i] for i in ids] if conti else ids
        return [self.action_emb[0]] * self.batch_size if conti else [0] * self.batch_size

    @property
    def offline_reward(self):
        """slate.py:164-174 / seqslate.py:71-86"""
        cur = self.cur_steps
        if self.seq:
            if cur % 9 != 0:
                return [0] * self.batch_size
            P = self.page_items
            action = np.array([x[:cur] for x in self.parsed.exposed])
            price = self.get_price(action.astype(np.int64).reshape(self.batch_size, -1))[:, -P:]
            label = np.array(self.parsed.feedback)[:, cur - P:cur]
            return np.sum(price * label, axis=1)
        if cur < self.max_steps:
            return [0] * self.batch_size
        price = self.get_price(np.array(self.parsed.exposed))
        label = np.array(self.parsed.feedback)
        return [sum([xx * yy for (xx, yy) in zip(x, y)]) for (x, y) in zip(price, label)]

    @property
    def user(self):
        return self.parsed.users
