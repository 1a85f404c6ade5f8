"""SlateState / SlateRecEnv with the reference's names and behaviour, executing on the GPU.

Mirrors ``rl4rs/env/slate.py``: ``SlateState`` (slate.py:8-218) keeps its per-batch state machine in
HBM behind ``librl4rs_hip.so``; ``SlateRecEnv`` (slate.py:220-308) scores with the HIP DIEN.  Public
attributes scripts read (``prev_actions``, ``action_mask``, ``special_mask``, ``cur_steps``,
``action_emb``, ``records``, ``item_info_d``, ``location_mask``, ``special_items``) are materialised
from device memory on access.

Config keys are the reference's; two additions select the zero-copy mode:
``return_tensors`` (bool, default False): return torch CUDA tensors instead of numpy/list objects;
``model_seed`` (int): seed for synthetic DIEN weights when ``model_file`` is empty.
"""
import numpy as np

from .base import RecSimBase, RecState, RecordBatch
from ..data import CatalogTables, RecordColumns
from ..utils.datautil import FeatureUtil
from .. import device as D

_CATALOG_CACHE = {}


def _catalog(iteminfo_file, action_size, action_emb_size, onehot):
    key = (iteminfo_file, action_size, action_emb_size, bool(onehot))
    if key not in _CATALOG_CACHE:
        _CATALOG_CACHE[key] = CatalogTables(iteminfo_file, action_size, action_emb_size, onehot)
    return _CATALOG_CACHE[key]


class StateRows(object):
    """What ``SlateState.state`` / ``get_complete_states`` hand to ``obs_fn`` / ``feature_extraction``:
    a view of R feature rows that live in the env's device buffers (``kind`` 'state' or 'complete')."""

    def __init__(self, owner, kind):
        self.owner = owner
        self.kind = kind

    @property
    def rows(self):
        return self.owner.batch_size * (1 if self.kind == 'state' else self.owner._env.n_complete)

    def __len__(self):
        return self.rows

    def numpy_features(self):
        env = self.owner._env
        B, L = env.B, env.L
        s0 = env.snapshot(D.BUF_SEQ0).cpu().numpy()
        s1 = env.snapshot(D.BUF_SEQ1).cpu().numpy()
        if self.kind == 'state':
            dense = env.snapshot(D.BUF_DENSE).cpu().numpy()
            cat = env.snapshot(D.BUF_CATEGORY).cpu().numpy()
            seq = np.stack([s0, s1], axis=1)
        else:
            n = env.n_complete
            dense = env.snapshot(D.BUF_C_DENSE).cpu().numpy()
            cat = env.snapshot(D.BUF_C_CATEGORY).cpu().numpy()
            seq = np.repeat(np.stack([s0, s1], axis=1), n, axis=0)
        return seq, dense, cat


def _unique_lines(rows, n):
    """np.unique(rows, return_inverse=True) for line numbers in [0, n): a presence table instead of a sort (the batch is drawn
    from a window of a few thousand lines, base.py:92-100) -> (sorted distinct lines, index of every row's line in them)."""
    if n > 4 * len(rows) + 65536 or rows.min() < 0:
        return np.unique(rows, return_inverse=True)
    present = np.zeros(n, dtype=bool)
    present[rows] = True
    uniq = np.flatnonzero(present)
    slot = np.cumsum(present, dtype=np.int64) - 1
    return uniq, slot[rows]


class SlateState(RecState):
    is_seq = False

    def __init__(self, config, records, _ctx=None):
        RecState.__init__(self, config, records)
        self.batch_size = self.config["batch_size"]
        self.action_size = self.config["action_size"]
        self.action_emb_size = self.config.get("action_emb_size", 32)
        self.max_steps = config['max_steps']
        self.page_items = config.get("page_items", 9)
        self._infos = None              # the per-env info dicts (slate.py:50) are made on first use: `infos` below
        onehot = config.get('support_onehot_action', False)
        # the reference always takes the last 32 dims (get_iteminfo_from_file default, slate.py:21,29); config's
        # action_emb_size only shapes the action space
        self._catalog = _catalog(config["iteminfo_file"], self.action_size, 32, onehot)
        if onehot:                      # slate.py:22-25
            config['action_emb_size'] = self.action_size
            self.action_emb_size = self.action_size
        self.action_emb = self._catalog.action_emb
        self.location_mask = self._catalog.location_mask
        self.special_items = self._catalog.special_items
        if len(records) != self.batch_size:
            raise ValueError('got %d records for batch_size=%d' % (len(records), self.batch_size))
        self._ctx = _ctx if _ctx is not None else {}
        self._bind_device(records)

    # ------------------------------------------------------------------ device plumbing
    def _violation_zeroes_reward(self):
        return True                     # slate.py:303-307: unconditional

    def _bind_device(self, records):
        import torch
        store = getattr(records, 'store', None)
        fused = None
        if store is not None:
            dev = torch.device('cuda', torch.cuda.current_device())
            rows = np.asarray(records.rows, dtype=np.int64)
            store.ensure(rows, dev)
            log_steps = store.log_steps
            self._exposed_len_min = int(store.exposed_len[rows].min())
            self._exposed_host = None            # host copy of the logged ids of these rows: gathered on first use (_offline_from_host)
            self._users = None
            self._store_rows = (store, rows)
            # RecDataBase.sample draws the batch WITH replacement from a cache window (base.py:92-100; 4096 envs from
            # 2048 lines at the bench config), so many envs share one user history: keep the distinct histories and
            # the env -> history map, the scorer encodes each distinct sequence once
            uniq, inv = _unique_lines(rows, store.n)
            dedup = len(uniq) < len(rows)
            # ONE pinned block, one host-to-device copy: [line of every env | distinct lines | env -> history slot | envs sorted
            # by slot]; ONE gather launch (rl4rs_env_load_lines) instead of a dozen index_select / copy / memset launches
            B, U = len(rows), (len(uniq) if dedup else 0)
            from .base import h2d_async
            packed = np.empty(B + U + (2 * B if dedup else 0), dtype=np.int32)
            packed[:B] = rows
            if dedup:
                packed[B:B + U] = uniq
                packed[B + U:2 * B + U] = inv
                # envs sorted by their history's slot: the scorer processes duplicates next to each other (L2 locality)
                packed[2 * B + U:] = np.argsort(inv.astype(np.uint16) if U <= 65536 else inv, kind='stable')   # 16-bit keys: radix sort
            pk = h2d_async(packed, dev)
            fused = (store._dev, store.n, pk[:B], pk[B:B + U] if dedup else None)
            if dedup:
                self._hist_unique = (torch.empty((U, self.config['maxlen']), dtype=torch.int32, device=dev), pk[B + U:2 * B + U])
                self._row_order = pk[2 * B + U:]
        else:
            rc = RecordColumns(list(records), self.config['maxlen'])
            cols = dict(exposed=rc.exposed, feedback=rc.feedback, history=rc.history,
                        user_dense=rc.user_dense, user_cat=rc.user_cat)
            log_steps = rc.log_steps
            self._exposed_len_min = int(rc.exposed_len.min())
            self._exposed_host = np.ascontiguousarray(rc.exposed, dtype=np.int32)
            self._users = rc.users
            self._feedback_cols = cols['feedback']
        key = ('env', self.is_seq, log_steps, self.batch_size, self.max_steps, self._violation_zeroes_reward())
        env = self._ctx.get(key)
        if env is None:
            env = D.DeviceEnv(self.config, self._catalog, self.is_seq, log_steps, self._violation_zeroes_reward())
            self._ctx[key] = env
        self._env = env
        self._ctx['owner'] = self
        if fused is not None:
            hu = getattr(self, '_hist_unique', None)
            env.load_lines(fused[0], fused[1], fused[2], fused[3], hu[0] if hu is not None else None)
        else:
            env.load_batch(cols['exposed'], cols['feedback'], cols['history'], cols['user_dense'], cols['user_cat'])
        env.reset()
        self._seq1_version = 0
        self._batch_version = self._ctx.get('batch_version', 0) + 1
        self._ctx['batch_version'] = self._batch_version

    @property
    def _feedback(self):
        """Logged click labels of the batch, int32 [B, log_steps] on the device (simulator training, rl4rs_amd/simtrain.py)."""
        fb = getattr(self, '_feedback_cols', None)
        if fb is None:
            import torch
            from .base import h2d_async
            store, rows = self._store_rows
            fb = store._dev['feedback'].index_select(0, h2d_async(rows, store._dev['feedback'].device))
            self._feedback_cols = fb
        return fb

    def _live(self):
        if self._ctx.get('owner') is not self:
            raise RuntimeError('this SlateState was replaced by a newer sample() on the same device env')
        return self._env

    # ------------------------------------------------------------------ reference attributes
    @property
    def cur_steps(self):
        return self._live().cur_steps

    @property
    def prev_actions(self):
        return self._live().snapshot(D.BUF_PREV_ACTIONS).cpu().numpy().astype(np.int64)

    @property
    def action_mask(self):
        env = self._live()
        return env.bits_to_mask(env.snapshot(D.BUF_ACTION_MASK))

    @property
    def special_mask(self):
        env = self._live()
        return env.bits_to_mask(env.snapshot(D.BUF_SPECIAL_MASK))

    @property
    def item_info_d(self):
        return self._catalog.item_info_dict()

    @staticmethod
    def get_iteminfo_from_file(iteminfo_file, action_size, action_emb_size=32):
        t = _catalog(iteminfo_file, action_size, action_emb_size, False)
        return t.item_info_dict(), t.action_emb

    @staticmethod
    def get_mask_from_file(iteminfo_file, action_size):
        t = _catalog(iteminfo_file, action_size, 32, False)
        return t.location_mask, t.special_items

    @staticmethod
    def records_to_state(records):
        """The reference's nested-list state (slate.py:67-83), kept for API parity."""
        out = []
        for rec in records:
            parts = FeatureUtil.record_split(rec)
            hist, portrait = parts[5], parts[6]
            out.append([0, [hist, [0]], portrait[10:], portrait[:10], [0] * 9, 0])
        return out

    def get_location_mask(self, location_mask, cur_layer):
        return np.repeat(location_mask[cur_layer][np.newaxis, :], self.batch_size, 0)

    # ------------------------------------------------------------------ state / obs views
    def _tensor_mode(self):
        return bool(self.config.get('return_tensors', False))

    def _obs_mask(self):
        import torch
        m = self._live().obs_mask(torch.int64)
        return m if self._tensor_mode() else D.to_host(m)

    def _masked_actions(self):
        """slate.py:98-104"""
        import torch
        env = self._live()
        pa = env.snapshot(D.BUF_PREV_ACTIONS).to(torch.int64)
        cur = torch.full((self.batch_size, 1), env.cur_steps, dtype=torch.int64, device=pa.device)
        return pa, cur

    @property
    def state(self):
        rows = StateRows(self, 'state')
        if self.config.get("support_rllib_mask", False):
            return {"state": rows, "action_mask": self._obs_mask()}
        elif self.config.get("support_d3rl_mask", False):
            pa, cur = self._masked_actions()
            if not self._tensor_mode():
                pa, cur = pa.cpu().numpy(), cur.cpu().numpy()
            return {"state": rows, "masked_actions": pa, "cur_steps": cur}
        return rows

    @property
    def user(self):
        if self._users is None:
            self._users = [x.split('@')[1] for x in self.records]
        return self._users

    @property
    def infos(self):
        if self._infos is None:
            self._infos = [{} for _ in range(self.batch_size)]
        return self._infos

    @infos.setter
    def infos(self, value):
        self._infos = value

    @property
    def info(self):
        return self.infos

    def to_string(self):
        return '\n'.join(self.records)

    # ------------------------------------------------------------------ dynamics
    def get_price(self, actions):
        return self._catalog.price[np.asarray(actions)]

    def get_complete_states(self):
        self._live().build_complete()
        return StateRows(self, 'complete')

    def get_violation(self):
        return self._live().violation().cpu().numpy().astype(np.int64)

    @property
    def offline_action(self):
        env = self._live()
        if env.cur_steps < self.max_steps and env.cur_steps >= self._exposed_len_min:
            raise IndexError('list index out of range')       # exposed_items[cur_step], slate.py:154-156
        conti = bool(self.config.get("support_conti_env", False))
        if self._tensor_mode():
            return env.offline_action(conti=conti)
        nxt = getattr(self, '_next_offline', None)
        if nxt is not None and nxt[0] == (self._batch_version, env.cur_steps):
            # built while the last transition's kernels ran (SlateRecEnv._step).  Handed out ONCE: the reference builds a fresh
            # list on every access (slate.py:150-161), so a caller that mutates what it got must not see its own edits on the
            # next read - later reads of the same step rebuild the list (remembering the same device copy of the ids)
            lst = nxt[1]
            if lst is None:
                return self._offline_from_host(env.cur_steps, conti, None if conti else nxt[2])
            self._next_offline = (nxt[0], None, getattr(lst, '_dev', None))
            return lst
        return self._offline_from_host(env.cur_steps, conti, None)

    def _offline_from_host(self, cur, conti, dev_ids):
        """The logged action of step ``cur`` in the reference's shape (slate.py:152-161) from the HOST copy of the logged item
        ids: a python list of ids, or of action-embedding rows for a continuous-action env.  ``dev_ids``: device tensor holding
        the same ids (part of the last transition record), remembered by the list so that handing it back to ``step`` uploads
        nothing."""
        env = self._live()
        ex = getattr(self, '_exposed_host', None)
        if ex is None and not self._tensor_mode() and getattr(self, '_store_rows', None) is not None:
            store, rows = self._store_rows
            ex = self._exposed_host = store.exposed_host[rows]
        if ex is None:                                      # (defensive: no host copy)
            out = D.to_host(env.offline_action(conti=conti))
            return [row for row in out] if conti else out.tolist()
        ids = ex[:, cur] if (cur < self.max_steps and cur < ex.shape[1]) else np.zeros(self.batch_size, dtype=np.int32)
        if conti:
            if ids.min() < 0 or ids.max() >= self.action_size:      # the device kernel flags this and plays item 0: let it
                out = D.to_host(env.offline_action(conti=True))
                return [row for row in out]
            return [row for row in self._catalog.action_emb[ids]]
        if dev_ids is not None and self.batch_size > 1:
            return D.OfflineActionList(ids.tolist(), dev=dev_ids, tag=(self._batch_version, cur))
        return ids.tolist()

    @property
    def offline_reward(self):
        env = self._live()
        if env.cur_steps < self.max_steps:
            return [0, ] * self.batch_size
        r = env.offline_reward()
        return r if self._tensor_mode() else r.cpu().numpy().tolist()

    @staticmethod
    def get_nearest_neighbor(actions, action_emb, temperature=None):
        import torch
        emb = torch.from_numpy(np.ascontiguousarray(action_emb, dtype=np.float64)).cuda()
        return D.knn(actions, emb).cpu().numpy().astype(np.int64)

    @staticmethod
    def get_nearest_neighbor_with_mask(actions, action_emb, action_mask, temperature=None):
        import torch
        emb = torch.from_numpy(np.ascontiguousarray(action_emb, dtype=np.float64)).cuda()
        return D.knn(actions, emb, mask=np.asarray(action_mask)).cpu().numpy().astype(np.int64)

    def act(self, actions):
        env = self._live()
        if self.config.get("support_conti_env", False):
            self.last_actions = env.act_conti(actions)
        else:
            self.last_actions = env.act_discrete(actions)
        if not self._tensor_mode():
            env.check_error_flag()


class DienModel(object):
    """What ``SlateRecEnv.get_model`` returns: host weights + (lazily) the device scorer."""

    def __init__(self, config, weights=None):
        self.config = config
        self.weights = weights
        self.device_net = None
        self.max_rows = 0
        self.max_slots = 0

    def ensure_device(self, max_rows, max_slots):
        if self.device_net is None or max_rows > self.max_rows or max_slots > self.max_slots:
            if self.device_net is not None:
                self.device_net.close()
            if self.weights is None:
                raise RuntimeError("no simulator weights: set config['model_file'] to an .npz with the "
                                   "rl4rs_amd.nets.dien.dien_spec arrays or config['model_seed'] for synthetic ones")
            self.max_rows, self.max_slots = max(max_rows, self.max_rows), max(max_slots, self.max_slots)
            self.device_net = self._make_device_net()
            self.zero_slot_ready = False
        return self.device_net

    def _make_device_net(self):
        return D.DeviceDien(self.config, self.weights, self.max_rows, self.max_slots)


class SimnetModel(DienModel):
    """The dnn / widedeep / lstm simulators (rl4rs/nets/dnn.py, widedeep.py, lstm.py) on the device."""

    def __init__(self, config, algo, weights=None):
        DienModel.__init__(self, config, weights)
        self.algo = algo

    def _make_device_net(self):
        return D.DeviceSimnet(self.config, self.weights, self.max_rows, self.max_slots, algo=self.algo)


class SlateRecEnv(RecSimBase):
    """Core simulator (slate.py:220-308) on the GPU."""
    default_state_seq = False

    def __init__(self, config, state_cls):
        if not (isinstance(state_cls, type) and issubclass(state_cls, SlateState)):
            raise NotImplementedError(
                "state_cls must be rl4rs_amd SlateState/SeqSlateState (or a subclass): the env step runs on "
                "the GPU and there is no CPU fallback for arbitrary RecState plugins")
        self.max_steps = config['max_steps']
        self.batch_size = config['batch_size']
        self.FeatureUtil = FeatureUtil(config)
        self._ctx = {}
        RecSimBase.__init__(self, config, state_cls)
        self._recData.state_kwargs = {'_ctx': self._ctx}
        self._encoded_batch = None
        self._encoded_seq1 = None
        self._slots = None

    # -- model ---------------------------------------------------------------------------------
    def get_model(self, config):
        model_type = config.get('algo', 'dien')           # slate.py:228-241
        from ..nets import dien, simnets
        if model_type == 'dien':
            model = DienModel(config)
            if not config.get('model_file', None):
                model.weights = dien.init_dien_weights(config, seed=config.get('model_seed', 7))
        elif model_type in simnets.ALGOS:
            model = SimnetModel(config, model_type)
            if not config.get('model_file', None):
                model.weights = simnets.init_simnet_weights(config, model_type, seed=config.get('model_seed', 7))
        else:
            raise NotImplementedError("config['algo'] must be 'dien', 'dnn', 'widedeep' or 'lstm' (got %r)" % (model_type,))
        return model

    def reload_model(self, model_file):
        """base.py:148-151 (``saver.restore``).  ``model_file`` is either the TF checkpoint prefix the reference takes
        (read by ``rl4rs_amd.utils.tfckpt`` - optional ``config['model_name_map']`` overrides its variable-name
        table) or an ``.npz`` of this package's weight names."""
        from ..nets import dien
        from ..utils import tfckpt
        algo = self.config.get('algo', 'dien')
        if str(model_file).endswith('.npz'):
            if algo == 'dien':
                self.model.weights = dien.load_weights(model_file, self.config)
            else:
                from ..nets import simnets
                self.model.weights = simnets.load_weights(model_file, self.config, algo)
        elif tfckpt.is_checkpoint(model_file):
            self.model.weights = tfckpt.load_simulator_weights(model_file, self.config, algo,
                                                               name_map=self.config.get('model_name_map'))
        else:
            raise FileNotFoundError(
                "model_file=%r is neither an .npz (rl4rs_amd.nets.dien.dien_spec / nets.simnets.simnet_spec names) "
                "nor the prefix of a TF checkpoint (<prefix>.index + <prefix>.data-*)" % (model_file,))
        if self.model.device_net is not None:
            self.model.device_net.close()
            self.model.device_net = None

    # -- sequence cache ------------------------------------------------------------------------
    def _net_for(self, samples):
        import torch
        B = self.batch_size
        env = samples._live()
        net = self.model.ensure_device(B * env.n_complete, B + 1)
        hu = None if self.config.get('no_history_dedup', False) else getattr(samples, '_hist_unique', None)
        if self._encoded_batch != (id(net), samples._batch_version):
            if hu is not None:
                net.encode(0, hu[0], 0)                     # history: one slot per DISTINCT sampled log line
            else:
                p0, _ = env.buffer_ptr(D.BUF_SEQ0)
                net.encode(0, (p0, B), 0)                   # history: one slot per env
            self._encoded_batch = (id(net), samples._batch_version)
            self._encoded_seq1 = None
            self._slots_hist = None
            if hasattr(net, 'set_row_order'):
                net.set_row_order(None if self.config.get('no_row_order', False) else getattr(samples, '_row_order', None))
        if samples.is_seq:
            if self._slots is None or self._slots[0] != 'seq':
                self._slots = ('seq', torch.arange(B, dtype=torch.int32, device=env.device).repeat(net.S, 1).contiguous())
                self._slots_hist = None
                self._seq_tab_own = True
            if samples._seq1_version == 0 and not self.config.get('no_seq_zero_slot', False):
                # the whole first page the second sequence input is the constant [0] (seqslate.py:107: prev_actions[:0]): ONE shared
                # slot like SlateState's, instead of encoding B identical all-zero rows at every reset
                self._ensure_zero_slot(net, env, B)
                if self._seq_tab_own:
                    self._slots[1][1:].fill_(B)
                    self._seq_tab_own = False
            else:
                if self._encoded_seq1 != samples._seq1_version:
                    p1, _ = env.buffer_ptr(D.BUF_SEQ1)
                    for s in range(1, net.S):
                        net.encode(s, (p1, B), 0)
                    self._encoded_seq1 = samples._seq1_version
                self._seq_slots_own(env, B)
        else:
            self._ensure_zero_slot(net, env, B)
            if self._slots is None or self._slots[0] != 'slate':
                sl = torch.full((net.S, B), B, dtype=torch.int32, device=env.device)
                sl[0] = torch.arange(B, dtype=torch.int32, device=env.device)
                self._slots = ('slate', sl.contiguous())
                self._slots_hist = None
        if getattr(self, '_slots_hist', None) is None:      # row 0 of the slot table: env -> history slot of this batch
            self._slots[1][0].copy_(hu[1] if hu is not None else torch.arange(B, dtype=torch.int32, device=env.device))
            self._slots_hist = samples._batch_version
        return net, self._slots[1]

    def _ensure_zero_slot(self, net, env, B):
        import torch
        if not getattr(self.model, 'zero_slot_ready', False):
            z = torch.zeros((1, net.L), dtype=torch.int32, device=env.device)
            for s in range(1, net.S):
                net.encode(s, z, B)                         # the constant [0] sequence: ONE shared slot (index B)
            self.model.zero_slot_ready = True

    def _seq_slots_own(self, env, B):
        """SeqSlate from the second page on: every env's second sequence input has a slot of its own (rows 1.. of the slot table,
        in place - the fused step holds the table's address)."""
        import torch
        if not getattr(self, '_seq_tab_own', True):
            self._slots[1][1:] = torch.arange(B, dtype=torch.int32, device=env.device)
            self._seq_tab_own = True

    # -- obs -----------------------------------------------------------------------------------
    def obs_fn(self, state):
        masked = self.config.get("support_rllib_mask", False)
        d3rl = self.config.get("support_d3rl_mask", False)
        rows = state["state"] if (masked or d3rl) else state
        samples = rows.owner
        tensor_mode = samples._tensor_mode()
        B = self.batch_size
        if self.config.get("rawstate_as_obs", False) and tensor_mode:
            # zero-copy form of the raw-state observation (slate.py:250-262): device tensors of the whole batch, what
            # DeviceRawPolicy.act takes
            env = samples._live()
            seq0, seq1 = env.snapshot(D.BUF_SEQ0), env.snapshot(D.BUF_SEQ1)
            obs = {"category_feature": env.snapshot(D.BUF_CATEGORY), "dense_feature": env.snapshot(D.BUF_DENSE),
                   "sequence_feature": [seq0] + [seq1] * (self.config['seq_num'] - 1)}
            if masked:
                obs["action_mask"] = state["action_mask"]
            return obs
        if self.config.get("rawstate_as_obs", False):
            feat, _ = self.FeatureUtil.feature_extraction(rows)
            obs = [{"category_feature": feat[2][i], "dense_feature": feat[1][i], "sequence_feature": feat[0][i]}
                   for i in range(B)]
            if masked:
                am = state["action_mask"]
                am = am.cpu().numpy() if hasattr(am, 'cpu') else am
                return [dict(action_mask=am[i], **obs[i]) for i in range(B)]
            return obs
        env = samples._live()
        net, slots = self._net_for(samples)
        dp, _ = env.buffer_ptr(D.BUF_DENSE)
        cp, _ = env.buffer_ptr(D.BUF_CATEGORY)
        obs, _ = net.forward(B, 1, dp, cp, slots, want_obs=True, want_prob=False)
        # the state row just scored IS the last complete-state row of the reward forward (same prev_actions,
        # action = the item just played): keep its activations so forward() scores one row less per env
        self._last_obs = (samples._batch_version, env.cur_steps, obs if not tensor_mode else obs.clone())
        if tensor_mode:
            if masked:
                return {"action_mask": state["action_mask"], "obs": obs}
            if d3rl:
                import torch
                return torch.cat([obs.double(), state["masked_actions"].double(), state["cur_steps"].double()], dim=-1)
            return obs
        obs = D.to_host(obs)
        if masked:
            return [{"action_mask": m, "obs": o} for m, o in zip(state["action_mask"], obs)]
        if d3rl:
            return np.concatenate([obs, state["masked_actions"], state["cur_steps"]], axis=-1)
        return obs

    # -- fused transition ----------------------------------------------------------------------
    def _stock_methods(self, samples):
        """The fused entry points re-implement act / obs_fn / forward / _reward_due of THIS package's classes inside the
        library.  A subclass that overrides one of them (state_cls and the env class are the reference's extension points) must
        see its override run: such an env takes the composed path (RecSimBase._step)."""
        from .seqslate import SeqSlateState, SeqSlateRecEnv
        st, ev = type(samples), type(self)
        stock_state = SeqSlateState if samples.is_seq else SlateState
        stock_env = SeqSlateRecEnv if samples.is_seq else SlateRecEnv
        return (st.act is stock_state.act and st.state is stock_state.state and st.get_complete_states is SlateState.get_complete_states
                and st._masked_actions is stock_state._masked_actions
                and ev.obs_fn is SlateRecEnv.obs_fn and ev.forward is SlateRecEnv.forward and ev._reward_due is stock_env._reward_due)

    def _fused_ok(self, samples):
        cfg = self.config
        return (not cfg.get("rawstate_as_obs", False) and not cfg.get('no_state_row_reuse', False)
                and not cfg.get('no_fused_step', False) and samples._env.n_complete > 1
                and not (samples._tensor_mode() and (cfg.get("support_d3rl_mask", False) or cfg.get("simulator_info_fetch", False)))
                and self._stock_methods(samples))

    def _stepper_for(self, samples):
        env = samples._live()
        net, slots = self._net_for(samples)                 # history encoded for this batch, slot table current
        key = (id(env), id(net), slots.data_ptr())
        if getattr(self, '_stepper_key', None) != key:
            if getattr(self, '_stepper', None) is not None:
                self._stepper.close()
            self._stepper = D.DeviceStepper(env, net, slots)
            self._stepper_key = key
        return env, net, self._stepper

    def _mask_dicts(self, rec):
        """The support_rllib_mask observation (slate.py:262-266): a list of B ``{"action_mask", "obs"}`` dicts over row views of the
        record's pinned block.  Called while the transition's kernels run, so it also takes the costs that would otherwise
        fall between two steps, with the GPU idle: the lists of the last TWO calls (4096 dicts + 8192 row views each at the
        bench size) are kept alive, and the older one is dropped here - a loop that holds ``obs`` until the next step has
        returned (``obs, r, d, info = env.step(a)``) would otherwise tear it down when it rebinds its variable, between two
        steps; this way the 12 k objects go while the kernels run.  The cycle collector is paused for the build (thousands
        of container allocations would trigger it several times, and dicts of arrays cannot form cycles)."""
        import gc
        keep = self.__dict__.setdefault('_obs_keep', [])
        while len(keep) >= 2:
            keep.pop(0)
        was_on = gc.isenabled()
        gc.disable()
        try:
            out = [{"action_mask": m, "obs": o} for m, o in zip(rec.mask, rec.obs)]
        finally:
            if was_on:
                gc.enable()
        keep.append(out)
        return out

    def _copied_obs(self, rec, masked):
        """config['copy_outputs']: the observation as PAGEABLE copies.  By default every array a reference-shaped step returns
        is a view into ONE page-locked block per step (13 MB at B=4096 in the rllib-mask mode): a caller that keeps a single
        row - a replay buffer appending ``obs[i]`` - keeps the whole block alive, and thousands of kept steps pin gigabytes of
        host memory.  Keep the default for the act -> step loop; set ``copy_outputs`` when observations are stored."""
        if not masked:
            return np.array(rec.obs)
        m, o = np.array(rec.mask), np.array(rec.obs)
        return [{"action_mask": mi, "obs": oi} for mi, oi in zip(m, o)]

    def sample(self, batch_size):
        """base.py:172-175: ``samples = recData.sample(B); obs = obs_fn(samples.state)``.  In the reference's host-returning
        modes the observation of the fresh batch comes back the way a transition's does (``_step`` below): ONE library call
        (rl4rs_env_observe_record_host) whose record - obs, the int64 mask or the float64 d3rl rows, the first logged action -
        is in pinned memory after one wait, the python objects around it built while the scorer runs."""
        samples = self._recData.sample(batch_size)
        if samples._tensor_mode() or not self._fused_ok(samples):
            return samples, self.obs_fn(samples.state)
        env, net, stepper = self._stepper_for(samples)
        conti = bool(self.config.get("support_conti_env", False))
        masked = self.config.get("support_rllib_mask", False)
        d3rl = (not masked) and self.config.get("support_d3rl_mask", False)
        want = ['offline_action'] + (['mask_i64'] if masked else []) + (['d3rl_obs'] if d3rl else [])
        built = {}

        def in_the_gpu_shadow(rec):
            if masked:
                built['obs'] = self._mask_dicts(rec)
            samples.infos                                   # the per-env info dicts
            if env.cur_steps < self.max_steps and env.cur_steps < samples._exposed_len_min:
                samples._next_offline = ((samples._batch_version, env.cur_steps),
                                         samples._offline_from_host(env.cur_steps, conti, None if conti else stepper.offline_action_view()))

        r = stepper.observe_record(conti=conti, want=want, shadow=in_the_gpu_shadow)
        samples._range_seen = getattr(samples, '_range_seen', 0) | int(r.status[1])
        self._last_obs = None
        if self.config.get('copy_outputs', False):
            return samples, self._copied_obs(r, masked)
        return samples, (built['obs'] if masked else r.obs)

    def _step(self, samples, action, **kwargs):
        """base.py:157-170.  The whole transition is ONE library call whenever this package's own act / obs_fn / forward are in
        charge: rl4rs_env_step_discrete / _conti in zero-copy mode (device tensors in and out, no host round trip),
        rl4rs_env_step_record in the reference's own list / ndarray modes (plain, support_rllib_mask, support_d3rl_mask,
        simulator_info_fetch): every output lands in one device record that ONE copy into pinned memory and ONE wait bring
        back.  Anything else (raw-state observations, overridden plug-in methods) composes it call by call."""
        if not self._fused_ok(samples):
            return RecSimBase._step(self, samples, action, **kwargs)
        env, net, stepper = self._stepper_for(samples)
        first_of_page = samples.is_seq and env.cur_steps > 0 and env.cur_steps % samples.page_items == 0      # (the first page starts with the [0] it had at reset)
        if first_of_page:
            self._seq_slots_own(env, self.batch_size)       # the library re-encodes every env's second input into its own slot
        conti = bool(self.config.get("support_conti_env", False))
        last = kwargs['step'] >= self.max_steps - 1
        masked = self.config.get("support_rllib_mask", False)
        if samples._tensor_mode():
            obs, reward, _, chosen = stepper.step(action, conti=conti)
            samples.last_actions = chosen
            if masked:
                obs = {"action_mask": samples._obs_mask(), "obs": obs}
        else:
            d3rl = (not masked) and self.config.get("support_d3rl_mask", False)
            fetch = bool(self.config.get("simulator_info_fetch", False))
            want = ['offline_action']
            if masked:
                want.append('mask_i64')
            if d3rl:
                want.append('d3rl_obs')
            if fetch:
                want.append('click_p')
            built = {}

            cur_after = env.cur_steps + 1

            def in_the_gpu_shadow(rec):
                # host work that needs no result of this transition, done while its kernels run: the python objects around the
                # (still empty) views, and the NEXT step's logged-action list (from the host copy of the log; the device copy
                # of the same ids is part of the record)
                if masked:
                    built['obs'] = self._mask_dicts(rec)
                built['done'] = [1 if last else 0] * self.batch_size
                built['zero'] = [0] * self.batch_size            # the reward of a step on which none is due (slate.py:284)
                if cur_after < self.max_steps and cur_after < samples._exposed_len_min:
                    samples._next_offline = ((samples._batch_version, cur_after),
                                             samples._offline_from_host(cur_after, conti, None if conti else stepper.offline_action_view()))

            if (isinstance(action, D.OfflineActionList) and action._dev is not None
                    and action._tag == (samples._batch_version, env.cur_steps)):
                action = action._dev                        # offline_action handed back unchanged: its device-side copy
            r = stepper.step_record(action, conti=conti, want=want, shadow=in_the_gpu_shadow)
            if r.status[0]:
                raise IndexError("an action id outside [0, action_size) was passed to act() "
                                 "(numpy would raise at rl4rs/env/slate.py:199)")
            import torch
            samples.last_actions = torch.from_numpy(r.chosen)
            due = self._reward_due(samples)
            samples._range_seen = getattr(samples, '_range_seen', 0) | int(r.status[1])      # the record read-and-cleared the flag
            if due and samples._range_seen:
                raise D._lib.Rl4rsHipError(
                    "fp16x2 scorer: a recurrent state left the fp16 range (|h| >= 6e4 or NaN); the affected forwards are "
                    "invalid - use config['scorer_precision'] = 'fp32' for this model")
            if fetch and due:
                click_p = np.array(r.click_p)               # a small pageable copy: info dicts outlive the step's pinned block
                for i in range(self.batch_size):
                    samples.info[i].update({'click_p': click_p[i]})
            obs = built['obs'] if masked else r.obs
            if self.config.get('copy_outputs', False):
                obs = self._copied_obs(r, masked)
            reward = r.reward.tolist() if due else built['zero']
        if first_of_page:                                   # the library re-encoded the second sequence input
            samples._seq1_version += 1
            self._encoded_seq1 = samples._seq1_version
        self._last_obs = None
        return obs, reward, built['done'] if not samples._tensor_mode() else [1 if last else 0] * self.batch_size, samples.info

    # -- reward --------------------------------------------------------------------------------
    def _reward_due(self, samples):
        return samples.cur_steps >= self.max_steps        # slate.py:283

    def forward(self, model, samples):
        import torch
        B = self.batch_size
        tensor_mode = samples._tensor_mode()
        if not self._reward_due(samples):
            if tensor_mode:
                return torch.zeros(B, dtype=torch.float64, device=samples._env.device)
            return [0] * B
        env = samples._live()
        n = env.n_complete
        net, slots = self._net_for(samples)
        last = getattr(self, '_last_obs', None)
        reuse = (last is not None and last[0] == samples._batch_version and last[1] == env.cur_steps and n > 1
                 and not self.config.get('no_state_row_reuse', False))
        m = n - 1 if reuse else n
        env.build_complete(m)
        dp, _ = env.buffer_ptr(D.BUF_C_DENSE)
        cp, _ = env.buffer_ptr(D.BUF_C_CATEGORY)
        _, probs = net.forward(B * m, m, dp, cp, slots, want_obs=False, want_prob=True)
        p_last = net.head_prob(last[2]) if reuse else None
        if self.config.get("simulator_info_fetch", False):
            pr = probs.reshape(B, m)
            if reuse:
                pr = torch.cat([pr, p_last[:, None]], dim=1)
            pr = pr.cpu().numpy()
            for i in range(B):
                samples.info[i].update({'click_p': pr[i]})
        reward = env.reward(probs, p_last)
        if tensor_mode:
            return reward          # zero-copy mode never synchronises; callers may poll model.device_net.check_status()
        out = reward.cpu().numpy().tolist()
        net.check_status()         # already synchronised by the copy: did the fp16x2 scorer stay in range?
        return out
