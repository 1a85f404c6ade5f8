"""Host-side mirror of the reference's env plugin surface, backed by the HIP hot path.

Same names, argument meaning and error behaviour as ``rl4rs/env/base.py``:

* ``single_elem_support``  base.py:9-23
* ``RecState`` (ABC)       base.py:26-57
* ``RecDataBase``          base.py:60-108   (file cache; sampling uses the global numpy RNG exactly like the
                                             reference so seeded runs draw the same records)
* ``RecSimBase`` (ABC)     base.py:111-176  (``_step`` order: act -> state -> obs_fn -> forward -> info -> done)
* ``RecEnvBase``           base.py:178-273  (gym facade; gym itself is optional here)

What differs, by design: the log file is parsed ONCE into columnar tensors resident in HBM
(``LogStore``) and every ``sample`` is a device-side row gather; the simulator net and the state
machine run on the GPU through librl4rs_hip.so.  There is no CPU execution path.
"""
from abc import ABC, abstractmethod
from operator import itemgetter

import numpy as np

try:                                   # gym is optional (not installed in the build image)
    import gym as _gym
    _Env = _gym.Env
    _spaces = _gym.spaces
except Exception:                      # pragma: no cover - depends on the image
    _gym = None

    class _Env(object):
        metadata = {}

    class _Space(object):
        def __init__(self, *args, **kw):
            self.args, self.kw = args, kw
            self.shape = kw.get('shape', None)

    class _Box(_Space):
        def __init__(self, low, high, shape=None, dtype=np.float32):
            _Space.__init__(self, low, high, shape=shape)
            self.low, self.high, self.dtype = low, high, dtype

        def contains(self, x):                      # gym.spaces.Box.contains: shape and bounds
            try:
                x = np.asarray(x, dtype=np.float64)
            except (TypeError, ValueError):
                return False
            return bool(tuple(x.shape) == tuple(self.shape) and np.all(x >= self.low) and np.all(x <= self.high))

    class _Discrete(_Space):
        def __init__(self, n):
            _Space.__init__(self, n)
            self.n = n

        def contains(self, x):                      # gym.spaces.Discrete.contains
            if isinstance(x, (int, np.integer)):
                v = int(x)
            elif isinstance(x, np.ndarray) and x.shape == () and np.issubdtype(x.dtype, np.integer):
                v = int(x)
            else:
                return False
            return 0 <= v < self.n

    class _Dict(_Space):
        def __init__(self, spaces):
            _Space.__init__(self, spaces)
            self.spaces = dict(spaces)

        def contains(self, x):
            return isinstance(x, dict) and set(x) == set(self.spaces) and all(sp.contains(x[k]) for k, sp in self.spaces.items())

    class _spaces(object):
        Box, Discrete, Dict = _Box, _Discrete, _Dict


_SEQ_TYPES = (list, tuple, np.ndarray)


def _unwrap_singleton(res):
    """Batch-of-one results are handed back without the batch axis (base.py:9-23)."""
    if type(res) in _SEQ_TYPES and len(res) == 1:
        return res[0]
    if type(res) not in _SEQ_TYPES:          # zero-copy mode: torch tensors / dicts of tensors pass through
        return res
    if type(res[0]) in _SEQ_TYPES and len(res[0]) == 1:
        return [part[0] for part in res]
    return res


def single_elem_support(func):
    """Decorator form of ``_unwrap_singleton`` (same name as the reference's aop helper)."""
    def wrapper(*args, **kwargs):
        return _unwrap_singleton(func(*args, **kwargs))
    return wrapper


class RecState(ABC):
    """State plugin ABC (base.py:26-57)."""

    def __init__(self, config, records):
        self.config = config
        self.records = records

    @staticmethod
    def records_to_state(records):
        pass

    @property
    def state(self):
        return self._state

    @property
    @abstractmethod
    def user(self):
        pass

    @property
    @abstractmethod
    def info(self):
        pass

    @abstractmethod
    def act(self, actions):
        pass

    @abstractmethod
    def to_string(self):
        pass


class RecordBatch(list):
    """The ``records`` handed to ``state_cls(config, records)``: the record strings (what the reference
    passes) plus their row numbers in the device-resident ``LogStore`` so no text is re-parsed."""

    def __init__(self, strings, rows=None, store=None):
        list.__init__(self, strings)
        self.rows = rows
        self.store = store


class LogStore(object):
    """A sample file parsed once into columnar tensors in HBM (one row per text line).

    Lines are parsed lazily the first time a cache window touches them and memoised, so an epoch over
    the file pays the text parsing once; ``preload()`` parses everything up front.
    """

    def __init__(self, path, maxlen, log_steps=None):
        with open(path, 'r') as f:
            self.lines = f.read().split('\n')
        self.maxlen = maxlen
        self.n = len(self.lines)
        self.log_steps = log_steps
        self._parsed = np.zeros(self.n, dtype=bool)
        self._dev = None
        self._min_len = None
        self._stripped = None
        self._stripped_arr = None
        self._all_parsed = False
        # prefix count of blank lines (a line whose rstrip() is empty reads as EOF for the reference's loop)
        self._blank_prefix = np.concatenate([[0], np.cumsum([0 if l.rstrip() else 1 for l in self.lines])])

    def nonblank_run(self, start, num):
        """True when lines [start, start + num) hold no blank line."""
        return int(self._blank_prefix[start + num] - self._blank_prefix[start]) == 0

    def nonblank_len(self, start):
        """Length of the run of non-blank lines that starts at ``start`` (0 at a blank line or at the end of the file)."""
        if start >= self.n:
            return 0
        return int(np.searchsorted(self._blank_prefix, self._blank_prefix[start], side='right')) - 1 - start

    @property
    def stripped_array(self):
        """The same lines as an object ndarray (built once): a sampled batch's record strings are one fancy-index gather."""
        if self._stripped_arr is None:
            arr = np.empty(self.n, dtype=object)
            arr[:] = self.stripped
            self._stripped_arr = arr
        return self._stripped_arr

    @property
    def stripped(self):
        """``line.rstrip()`` of every line (what ``fp.readline().rstrip()`` yields), built once: a cache window is a slice."""
        if self._stripped is None:
            self._stripped = [l.rstrip() for l in self.lines]
        return self._stripped

    def _ensure_width(self, rows):
        if self.log_steps is None:
            # exposed_items / user_feedback column count of the table = the LONGEST record of the file (one cheap pre-scan of
            # the fourth '@' field): a later, longer record must not be truncated - offline_reward sums over every logged
            # item (slate.py:164-174) - and shorter ones are zero padded (their true length stays in exposed_len)
            width = 0
            for line in self.lines:
                if line:
                    parts = line.split('@', 4)
                    if len(parts) > 3:
                        width = max(width, parts[3].count(',') + 1)
            self.log_steps = max(width, 1)

    def _alloc(self, device):
        import torch
        n, L, W = self.n, self.maxlen, self.log_steps
        self._dev = {
            'exposed': torch.zeros((n, W), dtype=torch.int32, device=device),
            'feedback': torch.zeros((n, W), dtype=torch.int32, device=device),
            'history': torch.zeros((n, L), dtype=torch.int32, device=device),
            'user_dense': torch.zeros((n, 32), dtype=torch.float32, device=device),
            'user_cat': torch.zeros((n, 10), dtype=torch.int32, device=device),
        }
        self.exposed_len = np.zeros(n, dtype=np.int64)
        # host copy of the logged item ids (a few hundred KB): the reference-shaped offline_action hands out python lists of
        # them, which then need no device round trip
        self.exposed_host = np.zeros((n, W), dtype=np.int32)

    def ensure(self, rows, device):
        """Make sure the given line numbers are parsed and resident on ``device``."""
        if self._all_parsed:
            return
        import torch
        from ..data import parse_records_native
        rows = np.unique(np.asarray(rows, dtype=np.int64))
        self._ensure_width(rows)
        if self._dev is None:
            self._alloc(device)
        todo = rows[~self._parsed[rows]]
        if len(todo) == 0:
            return
        cols = parse_records_native([self.lines[r] for r in todo], self.maxlen, self.log_steps)
        idx = torch.from_numpy(todo).to(device)
        for name in ('exposed', 'feedback', 'history', 'user_dense', 'user_cat'):
            src = torch.from_numpy(getattr(cols, name)).to(device)
            self._dev[name].index_copy_(0, idx, src)
        self.exposed_len[todo] = cols.exposed_len
        self.exposed_host[todo] = cols.exposed
        self._parsed[todo] = True

    def preload(self, device):
        rows = [i for i, l in enumerate(self.lines) if l.strip()]
        self.ensure(rows, device)
        self._all_parsed = True          # blank lines are never handed to a state (they read as EOF, sample_cache wraps there)

    def gather(self, rows, device):
        self.ensure(rows, device)
        idx = h2d_async(np.asarray(rows, dtype=np.int64), device)
        return dict((k, v.index_select(0, idx)) for k, v in self._dev.items())


def h2d_async(array, device):
    """Host array -> device tensor WITHOUT synchronising the stream: through pinned memory (torch's caching host allocator
    keeps the staging block alive until the copy has run).  A pageable ``.to(device)`` blocks the host until every kernel
    already queued has finished - once per reset that serialised the host's sampling work with the GPU (~1.2 ms per
    episode-batch of the bench workload)."""
    import torch
    t = torch.from_numpy(np.ascontiguousarray(array))
    if device is None or str(device) == 'cpu' or not torch.cuda.is_available():
        return t
    return t.pin_memory().to(device, non_blocking=True)


class RecDataBase(object):
    """File-based data source (base.py:60-108) over a device-resident ``LogStore``."""

    def __init__(self, config, state_cls):
        self.config = config
        self.sample_list = []
        self.sample_rows = []
        self._rows_arr = None
        self.state_cls = state_cls
        self.state_kwargs = {}
        self.is_eval = config.get('is_eval', False)
        self.cache_size = config.get('cache_size', 2048)
        self.store = LogStore(config['sample_file'], config['maxlen'])
        self._cursor = 0

    @staticmethod
    def seed(seed):
        np.random.seed(seed)

    def _readline(self):
        """fp.readline().rstrip() over the in-memory lines (returns '' at EOF like a file does)."""
        if self._cursor >= self.store.n:
            return '', -1
        row = self._cursor
        self._cursor += 1
        return self.store.lines[row].rstrip(), row

    def sample_cache(self, f, num):
        """base.py:82-90: on a blank/EOF read, seek to the start, skip one line, take the next."""
        stripped = self.store.stripped
        while num > 0:
            # a run without a blank line and without EOF: the reference's loop degenerates to a slice
            run = min(num, self.store.nonblank_len(self._cursor))
            if run > 0:
                c = self._cursor
                self.sample_list.extend(stripped[c:c + run])
                self.sample_rows.extend(range(c, c + run))
                self._cursor = c + run
                num -= run
                continue
            tmp, row = self._readline()                 # blank or EOF: wrap, skip one line, take the next whatever it is
            if len(tmp) < 1:
                self._cursor = 0
                self._readline()
                tmp, row = self._readline()
            self.sample_list.append(tmp)
            self.sample_rows.append(row)
            num -= 1

    def sample(self, batch_size):
        if self.is_eval:
            assert self.cache_size == batch_size
            assert len(self.sample_list) == batch_size
            pick = np.arange(batch_size)
        else:
            # np.random.choice(self.sample_list, batch_size) draws randint(0, len, size) from the global RNG
            pick = np.random.choice(len(self.sample_list), batch_size)
        if self._rows_arr is None or len(self._rows_arr) != len(self.sample_rows):
            self._rows_arr = np.asarray(self.sample_rows, dtype=np.int64)
        rows = self._rows_arr[pick]
        if len(rows) > 1 and rows.min() >= 0:
            strings = self.store.stripped_array[rows].tolist()      # one gather from the object array of all stripped lines
        else:
            picked = pick.tolist()
            strings = list(itemgetter(*picked)(self.sample_list)) if len(picked) > 1 else [self.sample_list[i] for i in picked]
        records = RecordBatch(strings, rows=rows, store=self.store)
        return self.state_cls(self.config, records, **self.state_kwargs)

    def reset(self, reset_file=False):
        self.sample_list = []
        self.sample_rows = []
        self._rows_arr = None
        if reset_file:
            self._cursor = 0
        self.sample_cache(None, self.cache_size)


class RecSimBase(ABC):
    """Core simulator (base.py:111-176).  ``self.model`` is whatever ``get_model`` returns (a device
    scorer here instead of a keras model); there is no TF session."""

    def __init__(self, config, state_cls):
        self.config = config
        self.max_steps = config['max_steps']
        self.batch_size = config['batch_size']
        self.model = self.get_model(config)
        if config.get('model_file', None):
            self.reload_model(config['model_file'])
        self._recData = RecDataBase(config, state_cls)

    def reset(self, reset_file=False):
        self._recData.reset(reset_file)

    @abstractmethod
    def get_model(self, config):
        pass

    @abstractmethod
    def obs_fn(self, state):
        pass

    @abstractmethod
    def forward(self, model, samples):
        pass

    def reload_model(self, model_file):
        raise NotImplementedError

    def seed(self, sd=0):
        self._recData.seed(sd)
        np.random.seed(sd)

    def _step(self, samples, action, **kwargs):
        """One batched transition in the reference's order (base.py:157-170):
        act -> state -> obs_fn -> forward(reward) -> info -> done (from the FACADE's step counter)."""
        samples.act(action)
        next_obs = self.obs_fn(samples.state)
        reward = self.forward(self.model, samples)
        last = kwargs['step'] >= self.max_steps - 1
        return next_obs, reward, [1 if last else 0] * self.batch_size, samples.info

    def sample(self, batch_size):
        samples = self._recData.sample(batch_size)
        obs = self.obs_fn(samples.state)
        return samples, obs


def _feature_len(x):
    return int(x.shape[0]) if hasattr(x, 'shape') else len(x)


def _observation_space(config, first):
    """Spaces derived from the first observation, as base.py:188-213 does.

    plain obs -> Box(+-1e5, (obs_dim,)); rllib mask mode -> Dict(action_mask Box(0,1,(A,)), obs ...);
    raw-state mode -> Dict of the three feature boxes (+ action_mask)."""
    Box, Dict = _spaces.Box, _spaces.Dict
    masked = bool(config.get("support_rllib_mask", False))
    parts = {}
    if masked:
        parts["action_mask"] = Box(0, 1, shape=(_feature_len(first['action_mask']),))
    if config.get("rawstate_as_obs", False):
        for key in ("category_feature", "dense_feature"):
            parts[key] = Box(-1000000.0, 1000000.0, shape=(_feature_len(first[key]),))
        seq = first['sequence_feature']
        parts["sequence_feature"] = Box(-1000000.0, 1000000.0, shape=tuple(int(d) for d in seq.shape))
        return Dict(parts)
    if masked:
        parts["obs"] = Box(-100000.0, 100000.0, shape=(_feature_len(first["obs"]),))
        return Dict(parts)
    return Box(-100000.0, 100000.0, shape=(_feature_len(first),))


def _action_space(config):
    """base.py:214-217"""
    if config.get("support_conti_env", False):
        return _spaces.Box(-1, 1, shape=(config['action_emb_size'],))
    return _spaces.Discrete(config['action_size'])


class RecEnvBase(_Env):
    """gym facade (base.py:178-273)."""
    metadata = {'render.modes': ['human']}

    def __init__(self, recsim):
        self.config = recsim.config
        self.batch_size = self.config['batch_size']
        self.cur_step = 0
        self.sim = recsim
        self.sim.reset()
        self.samples, self.obs = self.sim.sample(self.batch_size)
        first = dict((k, v[0]) for k, v in self.obs.items()) if isinstance(self.obs, dict) else self.obs[0]
        self.observation_space = _observation_space(self.config, first)
        self.action_space = _action_space(self.config)
        self.reset()

    def seed(self, sd=0):
        self.sim.seed(sd)
        np.random.seed(sd)

    @property
    @single_elem_support
    def state(self):
        return self.obs

    @property
    @single_elem_support
    def user_id(self):
        return self.samples.user

    @property
    @single_elem_support
    def offline_action(self):
        return self.samples.offline_action

    @property
    @single_elem_support
    def offline_reward(self):
        return self.samples.offline_reward

    @single_elem_support
    def step(self, action):
        import torch
        if not isinstance(action, (list, np.ndarray, torch.Tensor)):
            action = [action]
        obs, reward, done, info = self.sim._step(self.samples, action, step=self.cur_step)
        self.cur_step += 1
        return obs, reward, done, info

    def reset(self, reset_file=False):
        self.cur_step = 0
        self.sim.reset(reset_file)
        self.samples, self.obs = self.sim.sample(self.batch_size)
        return self.state

    def render(self, mode='human', close=False):
        print('Current State:', '\n')
        print(self.samples.to_string())

    def close(self):
        pass
