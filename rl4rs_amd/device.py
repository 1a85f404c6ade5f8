"""Thin object wrappers over the C-ABI handles of librl4rs_hip.so.

PyTorch is used for device memory and stream handles only (``tensor.data_ptr()``,
``torch.cuda.current_stream().cuda_stream``); all compute goes through the C ABI.
"""
import ctypes as C
import os

import numpy as np
import torch

from . import _lib
from ._lib import check

# rl4rs_env_buffer ids (include/rl4rs_hip.h)
BUF_PREV_ACTIONS, BUF_ACTION_MASK, BUF_SPECIAL_MASK, BUF_DENSE, BUF_CATEGORY, BUF_SEQ0, BUF_SEQ1, \
    BUF_C_DENSE, BUF_C_CATEGORY, BUF_ERROR_FLAG = range(10)
DIEN_ALL_FEATURE, DIEN_SCORES, DIEN_QUERY, DIEN_H1 = range(4)

_MASK_DTYPES = {torch.uint8: 0, torch.int32: 1, torch.int64: 2, torch.float32: 3}


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


def _dev_tensor(x, dtype, device):
    if isinstance(x, torch.Tensor):
        return x.to(device=device, dtype=dtype).contiguous()
    return torch.from_numpy(np.ascontiguousarray(x)).to(device=device, dtype=dtype).contiguous()


def to_device_async(x, dtype, device):
    """Host data (list / ndarray / CPU tensor) -> device tensor of ``dtype`` through pinned memory, without draining the
    queue (a pageable ``.to(device)`` blocks until every queued kernel has finished); device tensors pass through."""
    if isinstance(x, torch.Tensor) and x.is_cuda:
        return x.to(device=device, dtype=dtype).contiguous()
    np_dtype = {torch.int32: np.int32, torch.int64: np.int64, torch.float32: np.float32, torch.float64: np.float64}[dtype]
    if isinstance(x, list) and dtype == torch.int32 and x and type(x[0]) is int:
        import array                                # a python list of ints (what offline_action hands out): half the cost of np.asarray
        h = np.frombuffer(array.array('i', x), dtype=np.int32)
    else:
        h = x.detach().cpu().numpy() if isinstance(x, torch.Tensor) else x
        h = np.ascontiguousarray(np.asarray(h), dtype=np_dtype)
    p = torch.empty(h.shape, dtype=dtype, device='cpu', pin_memory=True)       # caching host allocator: no page pinning per call
    p.numpy()[...] = h
    return p.to(device, non_blocking=True)


def wait_stream():
    """Wait for everything queued on the current stream.  (Measured on the bench box: polling an event from python instead of
    hipStreamSynchronize - to shave its wake-up latency off every reference-shaped step - was no faster: 15.24 -> 15.26 ms per
    episode-batch; the runtime's own wait already spins before it parks.)"""
    torch.cuda.current_stream().synchronize()


class OfflineActionList(list):
    """The python list ``offline_action`` hands out in the reference-shaped modes, remembering the device-side copy of the
    same ids (part of the last transition record): handed back to ``env.step`` unchanged - the reference's canonical replay
    loop - the ids never cross PCIe again.  Any in-place change drops the device reference, so a modified list is converted
    and uploaded like any other."""
    __slots__ = ('_dev', '_tag')

    def __init__(self, values, dev=None, tag=None):
        list.__init__(self, values)
        self._dev, self._tag = dev, tag

    def _drop(self):
        self._dev = None

    def __reduce__(self):                       # pickles / copies as the plain list it is
        return (list, (list(self),))


def _mutator(name):
    base = getattr(list, name)

    def f(self, *a, **k):
        self._drop()
        return base(self, *a, **k)
    f.__name__ = name
    return f


for _n in ('__setitem__', '__delitem__', '__iadd__', '__imul__', 'append', 'extend', 'insert', 'pop', 'remove', 'reverse', 'sort', 'clear'):
    setattr(OfflineActionList, _n, _mutator(_n))


def to_host(t):
    """Device tensor -> numpy array through ONE pinned staging block of torch's caching host allocator (a pageable ``.cpu()``
    goes through an internal bounce buffer at a fraction of the PCIe rate).  The array aliases the pinned block, which goes
    back to the allocator's cache when the array is released."""
    h = torch.empty(t.shape, dtype=t.dtype, device='cpu', pin_memory=True)
    h.copy_(t, non_blocking=True)
    wait_stream()
    return h.numpy()


class DeviceEnv(object):
    """rl4rs_env handle: SlateState / SeqSlateState device state for one batch."""

    def __init__(self, config, catalog, is_seq, log_steps, violation_zeroes_reward, device=None):
        _lib.require_device()
        self.lib = _lib.load()
        self.device = torch.device('cuda', torch.cuda.current_device()) if device is None else torch.device(device)
        self.B = int(config['batch_size'])
        self.T = int(config['max_steps'])
        self.A = int(config['action_size'])
        self.E = int(catalog.action_emb.shape[1])
        self.P = int(config.get('page_items', 9))
        self.L = int(config['maxlen'])
        self.Dn = int(config['dense_feature_num'])
        self.Cn = int(config['category_feature_num'])
        self.W = (self.A + 31) // 32
        self.is_seq = bool(is_seq)
        self.log_steps = int(log_steps)
        cfg = _lib.EnvCfg(self.B, self.T, self.A, self.E, self.P, int(catalog.item_dim), 32, 10, self.L,
                          self.Dn, self.Cn, self.log_steps, 1 if is_seq else 0,
                          1 if violation_zeroes_reward else 0)
        self.user_dense_dim, self.user_cat_dim = 32, 10
        h = C.c_void_p()
        with torch.cuda.device(self.device):
            check(self.lib.rl4rs_env_create(C.byref(cfg), C.byref(h)))
            self.h = h
            loc = np.ascontiguousarray(catalog.location_mask.astype(np.uint8))
            check(self.lib.rl4rs_env_set_catalog(
                self.h, catalog.item_vec.ctypes.data_as(C.c_void_p), catalog.price.ctypes.data_as(C.c_void_p),
                catalog.action_emb.ctypes.data_as(C.c_void_p), catalog.is_special.ctypes.data_as(C.c_void_p),
                loc.ctypes.data_as(C.c_void_p), _stream()))
        self.n_complete = self.lib.rl4rs_env_complete_rows(self.h)
        self._keep = None
        if 'env_rows_variant' in config:                       # A/B runs: catalogue staged in LDS (0, default) or read through L1 / L2 (1)
            self.set_option('rows_variant', config['env_rows_variant'])

    def set_option(self, name, value):
        """Kernel-path selection of this handle (rl4rs_env_set_option)."""
        check(self.lib.rl4rs_env_set_option(self.h, _lib.ENV_OPTS[name], int(value)))

    def close(self):
        if getattr(self, 'h', None) is not None and self.h:
            self.lib.rl4rs_env_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---------------------------------------------------------------- batch / reset
    def load_batch(self, exposed, feedback, history, user_dense, user_cat):
        d = self.device
        ex = _dev_tensor(exposed, torch.int32, d)
        fb = _dev_tensor(feedback, torch.int32, d)
        hi = _dev_tensor(history, torch.int32, d)
        ud = _dev_tensor(user_dense, torch.float32, d)
        uc = _dev_tensor(user_cat, torch.int32, d)
        assert ex.shape == (self.B, self.log_steps) and fb.shape == ex.shape, (ex.shape, self.log_steps)
        assert hi.shape == (self.B, self.L) and ud.shape == (self.B, 32) and uc.shape == (self.B, 10)
        check(self.lib.rl4rs_env_load_batch(self.h, _ptr(ex), _ptr(fb), _ptr(hi), _ptr(ud), _ptr(uc), _stream()))
        self._keep = (ex, fb, hi, ud, uc)     # keep sources alive until the async copies ran

    def load_lines(self, tables, n_lines, line_idx, uniq_idx=None, hist_unique=None):
        """rl4rs_env_load_lines: the sampled lines of the resident log tables -> batch buffers + start-of-episode state, and the
        batch's distinct histories -> ``hist_unique`` (int32 [n_uniq, L]), in ONE launch.  ``tables``: dict of the LogStore's
        device tensors; ``line_idx`` / ``uniq_idx``: int32 device tensors."""
        assert line_idx.dtype == torch.int32 and line_idx.numel() == self.B and line_idx.is_contiguous()
        n_uniq = 0 if uniq_idx is None else int(uniq_idx.numel())
        if n_uniq:
            assert uniq_idx.dtype == torch.int32 and uniq_idx.is_contiguous()
            assert hist_unique.dtype == torch.int32 and tuple(hist_unique.shape) == (n_uniq, self.L) and hist_unique.is_contiguous()
        ex = tables['exposed']
        assert ex.shape[1] == self.log_steps and tables['history'].shape[1] == self.L
        check(self.lib.rl4rs_env_load_lines(self.h, _ptr(ex), _ptr(tables['feedback']), _ptr(tables['history']),
                                            _ptr(tables['user_dense']), _ptr(tables['user_cat']), int(n_lines), int(ex.shape[1]),
                                            _ptr(line_idx), _ptr(uniq_idx), n_uniq, _ptr(hist_unique), _stream()))
        self._keep = (line_idx, uniq_idx, hist_unique)

    def reset(self):
        check(self.lib.rl4rs_env_reset(self.h, _stream()))

    # ------------------------------------------------------------------------ act
    def act_discrete(self, actions):
        a = _dev_tensor(actions, torch.int32, self.device).reshape(-1)
        assert a.numel() == self.B, (a.shape, self.B)
        check(self.lib.rl4rs_env_act_discrete(self.h, _ptr(a), _stream()))
        return a

    def act_conti(self, actions):
        if isinstance(actions, torch.Tensor):
            a = actions.to(self.device)
            if a.dtype not in (torch.float32, torch.float64):
                a = a.to(torch.float64)
            a = a.contiguous()
        else:
            a = np.asarray(actions)
            a = a.astype(np.float32 if a.dtype == np.float32 else np.float64)
            a = torch.from_numpy(np.ascontiguousarray(a)).to(self.device)
        assert a.shape == (self.B, self.E), (a.shape, (self.B, self.E))
        chosen = torch.empty(self.B, dtype=torch.int32, device=self.device)
        check(self.lib.rl4rs_env_act_conti(self.h, _ptr(a), 1 if a.dtype == torch.float64 else 0,
                                           _ptr(chosen), _stream()))
        return chosen

    # -------------------------------------------------------------------- queries
    @property
    def cur_steps(self):
        return self.lib.rl4rs_env_cur_steps(self.h)

    def is_reward_step(self):
        return bool(self.lib.rl4rs_env_is_reward_step(self.h))

    def buffer_ptr(self, which):
        p = C.c_void_p()
        n = C.c_int64()
        check(self.lib.rl4rs_env_buffer(self.h, which, C.byref(p), C.byref(n)))
        return p, n.value

    def snapshot(self, which):
        """Copy an env-owned buffer into a fresh torch tensor (device)."""
        p, n = self.buffer_ptr(which)
        B, nc = self.B, self.n_complete
        shapes = {
            BUF_PREV_ACTIONS: ((B, self.T), torch.int32), BUF_ACTION_MASK: ((B, self.W), torch.int32),
            BUF_SPECIAL_MASK: ((B, self.W), torch.int32), BUF_DENSE: ((B, self.Dn), torch.float32),
            BUF_CATEGORY: ((B, self.Cn), torch.int32), BUF_SEQ0: ((B, self.L), torch.int32),
            BUF_SEQ1: ((B, self.L), torch.int32), BUF_C_DENSE: ((B * nc, self.Dn), torch.float32),
            BUF_C_CATEGORY: ((B * nc, self.Cn), torch.int32), BUF_ERROR_FLAG: ((1,), torch.int32),
        }
        shape, dt = shapes[which]
        out = torch.empty(shape, dtype=dt, device=self.device)
        assert out.numel() * out.element_size() == n, (which, shape, n)
        check(self.lib.rl4rs_copy_d2d(_ptr(out), p, n, _stream()))
        return out

    def bits_to_mask(self, bits):
        """uint32 bit rows [B,W] -> int64 [B,A] (numpy), the reference's mask layout."""
        b = bits.cpu().numpy().view(np.uint32)
        k = np.arange(self.A)
        return ((b[:, k >> 5] >> (k & 31).astype(np.uint32)) & 1).astype(np.int64)

    def build_complete(self, rows_per_env=None):
        if rows_per_env is None:
            check(self.lib.rl4rs_env_build_complete(self.h, _stream()))
        else:
            check(self.lib.rl4rs_env_build_complete_rows(self.h, rows_per_env, _stream()))

    def reward(self, probs, p_last=None):
        out = torch.empty(self.B, dtype=torch.float64, device=self.device)
        m = self.n_complete - (1 if p_last is not None else 0)
        assert probs.dtype == torch.float32 and probs.numel() == self.B * m
        assert p_last is None or (p_last.dtype == torch.float32 and p_last.numel() == self.B)
        check(self.lib.rl4rs_env_reward_split(self.h, _ptr(probs), _ptr(p_last), _ptr(out), _stream()))
        return out

    def violation(self):
        out = torch.empty(self.B, dtype=torch.int32, device=self.device)
        check(self.lib.rl4rs_env_violation(self.h, _ptr(out), _stream()))
        return out

    def obs_mask(self, dtype=torch.int64):
        out = torch.empty((self.B, self.A), dtype=dtype, device=self.device)
        check(self.lib.rl4rs_env_obs_mask(self.h, _ptr(out), _MASK_DTYPES[dtype], _stream()))
        return out

    def obs_mask_bits(self, out=None):
        """The same mask packed 32 actions per int32 word, [B, (A + 31) // 32] - the ``mask_bits`` the policy kernels take
        (``out``: a contiguous int32 [B, W] tensor to fill, e.g. a slice of a rollout buffer)."""
        if out is None:
            out = torch.empty((self.B, (self.A + 31) // 32), dtype=torch.int32, device=self.device)
        assert out.dtype == torch.int32 and tuple(out.shape) == (self.B, (self.A + 31) // 32) and out.is_contiguous()
        check(self.lib.rl4rs_env_obs_mask(self.h, _ptr(out), 4, _stream()))
        return out

    def offline_action(self, conti=False):
        ids = torch.empty(self.B, dtype=torch.int32, device=self.device)
        emb = torch.empty((self.B, self.E), dtype=torch.float64, device=self.device) if conti else None
        check(self.lib.rl4rs_env_offline_action(self.h, _ptr(ids), _ptr(emb), _stream()))
        return emb if conti else ids

    def offline_reward(self):
        out = torch.empty(self.B, dtype=torch.float64, device=self.device)
        check(self.lib.rl4rs_env_offline_reward(self.h, _ptr(out), _stream()))
        return out

    def predict_with_mask(self, scores, obs_tail):
        """policy_model.predict_with_mask: scores [N,A] f32, obs_tail [N, P+1] = previous actions | cur_step."""
        scores = scores.to(device=self.device, dtype=torch.float32).contiguous()
        tail = obs_tail.to(device=self.device)
        prev = tail[:, :-1].to(torch.int32).contiguous()
        cur = tail[:, -1].to(torch.int32).contiguous()
        N = scores.shape[0]
        out = torch.empty(N, dtype=torch.int32, device=self.device)
        check(self.lib.rl4rs_env_predict_with_mask(self.h, N, _ptr(scores), _ptr(prev), prev.shape[1], _ptr(cur),
                                                   _ptr(out), _stream()))
        return out

    def check_error_flag(self):
        flag = int(self.snapshot(BUF_ERROR_FLAG).item())
        if flag:
            raise IndexError("an action id outside [0, action_size) was passed to act() "
                             "(numpy would raise at rl4rs/env/slate.py:199)")


def knn(actions, action_emb_dev, mask=None):
    """SlateState.get_nearest_neighbor(_with_mask) on device: actions [n,E] (f32/f64) -> int32 [n]."""
    lib = _lib.load()
    dev = action_emb_dev.device
    a = actions if isinstance(actions, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(np.asarray(actions)))
    if a.dtype not in (torch.float32, torch.float64):
        a = a.to(torch.float64)
    a = a.to(dev).contiguous()
    n, E = a.shape
    out = torch.empty(n, dtype=torch.int32, device=dev)
    m = None
    if mask is not None:
        m = (torch.as_tensor(np.asarray(mask) >= 0.5) if not isinstance(mask, torch.Tensor) else (mask >= 0.5))
        m = m.to(device=dev, dtype=torch.uint8).contiguous()
        assert m.shape == (n, action_emb_dev.shape[0])
    check(lib.rl4rs_knn(_ptr(a), 1 if a.dtype == torch.float64 else 0, n, _ptr(action_emb_dev),
                        action_emb_dev.shape[0], E, _ptr(m), _ptr(out), _stream()))
    return out


SCORER_MODES = {'auto': 0, 'fp32': 1, 'fp16x2': 2}       # include/rl4rs_hip.h RL4RS_SCORER_*
class DeviceStepper(object):
    """rl4rs_stepper handle: an env bound to its scorer so that one call runs a whole transition
    (rl4rs_env_step_discrete / rl4rs_env_step_conti = RecSimBase._step, base.py:157-170)."""

    def __init__(self, env, net, slots):
        self.lib = _lib.load()
        self.env, self.net, self.slots = env, net, slots          # keep the handles and the slot table alive
        assert slots.dtype == torch.int32 and slots.is_contiguous() and slots.shape[1] == env.B
        h = C.c_void_p()
        attach = self.lib.rl4rs_env_attach_simnet if isinstance(net, DeviceSimnet) else self.lib.rl4rs_env_attach_scorer
        check(attach(env.h, net.h, _ptr(slots), int(slots.shape[0]), C.byref(h)))
        self.h = h
        self.B, self.A, self.E, self.W, self.device = env.B, env.A, env.E, env.W, env.device
        self.obs_dim = int(getattr(net, 'obs_dim', 256))         # 256 for DIEN / dnn / lstm, 256 + U + Cn*E for widedeep

    def close(self):
        if getattr(self, 'h', None) is not None and self.h:
            self.lib.rl4rs_stepper_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def step(self, actions, conti=False, want_reward=True, want_mask_bits=False):
        """-> (obs f32 [B, obs_dim], reward f64 [B] or None, mask_bits i32 [B, W] or None, chosen i32 [B])."""
        obs = torch.empty((self.B, self.obs_dim), dtype=torch.float32, device=self.device)
        reward = torch.empty(self.B, dtype=torch.float64, device=self.device) if want_reward else None
        bits = torch.empty((self.B, self.W), dtype=torch.int32, device=self.device) if want_mask_bits else None
        if conti:
            a = actions.to(self.device) if isinstance(actions, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(actions)).to(self.device)
            if a.dtype not in (torch.float32, torch.float64):
                a = a.to(torch.float64)
            a = a.contiguous()
            assert a.shape == (self.B, self.E), (a.shape, (self.B, self.E))
            chosen = torch.empty(self.B, dtype=torch.int32, device=self.device)
            check(self.lib.rl4rs_env_step_conti(self.h, _ptr(a), 1 if a.dtype == torch.float64 else 0, _ptr(chosen), _ptr(obs),
                                                _ptr(reward), None, _ptr(bits), _stream()))
        else:
            chosen = _dev_tensor(actions, torch.int32, self.device).reshape(-1)
            assert chosen.numel() == self.B, (chosen.shape, self.B)
            check(self.lib.rl4rs_env_step_discrete(self.h, _ptr(chosen), _ptr(obs), _ptr(reward), None, _ptr(bits), _stream()))
        return obs, reward, bits, chosen

    def offline_action_view(self):
        """Device view of the logged next-step actions the LAST step_record left in its record (int32 [B]; None if it wrote
        none).  Valid until the next step_record has read it: that call consumes its action ids before it rewrites this part."""
        last = getattr(self, '_last_record', None)
        if last is None or last[0].offline_action < 0:
            return None
        L, rec = last
        off = int(L.offline_action)
        return rec[off:off + self.B * 4].view(torch.int32)

    # ---- reference-shaped transitions: one library call, ONE device-to-host copy, one wait ----------------------------------
    def _layout(self, want, conti):
        key = (want, bool(conti))
        cache = self.__dict__.setdefault('_layouts', {})
        if key not in cache:
            L = _lib.StepRecord()
            check(self.lib.rl4rs_stepper_record_layout(self.h, want, 1 if conti else 0, C.byref(L)))
            rec = torch.empty(int(L.total_bytes), dtype=torch.uint8, device=self.device)
            cache[key] = (L, rec)
        return cache[key]

    def step_record(self, actions, conti=False, want=(), shadow=None):
        """rl4rs_env_step_record_host (the transition + its record's host part copied into a fresh pinned block: the int64
        mask of the rllib mode beside the scorer's kernels, the rest after the last one) -> ``StepResult`` of numpy
        views (obs float32 [B, obs_dim], or float64 [B, obs_dim + cols + 1] when 'd3rl_obs' is wanted; reward float64 [B];
        done uint8 [B]; chosen int32 [B]; mask int64 [B, A]; mask_bits uint32 [B, W]; click_p float32 [B, n]; offline_action
        int32 [B] / float64 [B, E]; status int32 [2]).  ``want``: names from rl4rs_amd._lib.STEP_WANT.
        ``shadow(result)`` (optional) runs after everything is enqueued and BEFORE the wait: the views exist (they alias the
        pinned block) but hold no data yet - the place for host work that only needs the objects, e.g. building the list of
        per-env dicts the reference's mask mode returns, while the GPU computes."""
        bits = 0
        for name in want:
            bits |= _lib.STEP_WANT[name]
        L, rec = self._layout(bits, conti)
        if actions is None:
            a = None                                        # observe_record: no transition
        elif conti:
            a = actions
            if not (isinstance(a, torch.Tensor) and a.is_cuda):
                a = np.asarray(a.detach().cpu().numpy() if isinstance(a, torch.Tensor) else a)
                a = to_device_async(a, torch.float32 if a.dtype == np.float32 else torch.float64, self.device)
            elif a.dtype not in (torch.float32, torch.float64):
                a = a.to(torch.float64)
            a = a.contiguous()
            assert tuple(a.shape) == (self.B, self.E), (a.shape, (self.B, self.E))
            kind = 2 if a.dtype == torch.float64 else 1
        else:
            a = to_device_async(actions, torch.int32, self.device).reshape(-1)
            assert a.numel() == self.B, (a.shape, self.B)
            kind = 0
        self._last_record = (L, rec)
        nb = int(L.host_bytes)
        # the page-locked block of this record: the one the PREVIOUS call set aside while its kernels ran (same size), else a fresh one
        spare = self.__dict__.pop('_spare_block', None)
        host = spare if (spare is not None and spare.numel() == nb) else torch.empty(nb, dtype=torch.uint8, device='cpu', pin_memory=True)
        if a is None:
            check(self.lib.rl4rs_env_observe_record_host(self.h, 1 if conti else 0, bits, _ptr(rec), host.data_ptr(), _stream()))
        else:
            check(self.lib.rl4rs_env_step_record_host(self.h, _ptr(a), kind, bits, _ptr(rec), host.data_ptr(), _stream()))
        raw = host.numpy()
        B = self.B

        def view(off, dtype, shape):
            if off < 0:
                return None
            n = int(np.prod(shape)) * np.dtype(dtype).itemsize
            return raw[off:off + n].view(dtype).reshape(shape)

        r = StepResult()
        r._block = host
        r.status = view(L.status, np.int32, (2,))
        r.reward = view(L.reward, np.float64, (B,))
        r.done = view(L.done, np.uint8, (B,))
        r.chosen = view(L.chosen, np.int32, (B,))
        r.obs = (view(L.obs_d3rl, np.float64, (B, int(L.d3rl_cols))) if L.obs_d3rl >= 0 else view(L.obs, np.float32, (B, int(L.obs_dim))))
        r.mask = view(L.mask_i64, np.int64, (B, self.A))
        r.mask_bits = view(L.mask_bits, np.uint32, (B, self.W))
        r.click_p = view(L.click_p, np.float32, (B, self.env.n_complete))
        r.offline_action = view(L.offline_action, np.float64, (B, self.E)) if conti else view(L.offline_action, np.int32, (B,))
        if shadow is not None:
            shadow(r)
        self._spare_block = torch.empty(nb, dtype=torch.uint8, device='cpu', pin_memory=True)      # in the GPU's shadow: the next record's block
        wait_stream()
        return r

    def observe_record(self, conti=False, want=(), shadow=None):
        """rl4rs_env_observe_record_host: the record of the state the env is in (no transition; what a reset returns) in the
        same form as ``step_record``; reward / done / chosen of the result are meaningless."""
        return self.step_record(None, conti=conti, want=want, shadow=shadow)


class StepResult(object):
    """Host views of one transition record (DeviceStepper.step_record): numpy arrays aliasing ONE pinned block."""
    __slots__ = ('status', 'reward', 'done', 'chosen', 'obs', 'mask', 'mask_bits', 'click_p', 'offline_action', '_block')


AUGRU_KERNELS = {1: 'k_recur<256,augru>', 2: 'k_augru_h16'}


def parse_dien_opts(spec):
    """config['scorer_kernels'] -> sorted tuple of option names (rl4rs_amd._lib.DIEN_OPTS)."""
    if spec is None:
        return ()
    names = [x.strip().lower() for x in spec.split(',')] if isinstance(spec, str) else [str(x).strip().lower() for x in spec]
    names = sorted(set(n for n in names if n))
    bad = [n for n in names if n not in _lib.DIEN_OPTS]
    if bad:
        raise ValueError("unknown scorer_kernels option(s) %s; known: %s" % (bad, sorted(_lib.DIEN_OPTS)))
    return tuple(names)


class DeviceDien(object):
    """rl4rs_dien handle: DIEN scorer with its sequence cache."""

    def __init__(self, config, weights, max_rows, max_slots, device=None):
        _lib.require_device()
        self.lib = _lib.load()
        self.device = torch.device('cuda', torch.cuda.current_device()) if device is None else torch.device(device)
        self.S = int(config['seq_num'])
        self.L = int(config['maxlen'])
        self.E = int(config['emb_size'])
        self.U = int(config['hidden_units'])
        self.Cn = int(config['category_feature_num'])
        self.Dn = int(config['dense_feature_num'])
        self.K = int(config['class_num'])
        self.max_rows, self.max_slots = int(max_rows), int(max_slots)
        self.F = self.S * 2 * self.E + self.U + (self.Cn + 1) * self.E
        # config['scorer_precision']: 'auto' (default: fp16x2 when the weights allow it; the RL4RS_SCORER environment variable
        # overrides the default for A/B runs - read HERE, the library itself never looks at the environment),
        # 'fp32' (exact-operand fp32 MFMA) or 'fp16x2' (fp16 hi+lo operand split, fp32 accumulate)
        precision = str(config.get('scorer_precision') or os.environ.get('RL4RS_SCORER') or 'auto').lower()
        if precision not in SCORER_MODES:
            raise ValueError("scorer_precision must be one of %s (got %r)" % (sorted(SCORER_MODES), precision))
        # config['scorer_kernels']: names of rl4rs_dien_cfg.kernel_opts bits (rl4rs_amd._lib.DIEN_OPTS; include/rl4rs_hip.h
        # RL4RS_DIEN_OPT_*), an iterable or a comma-separated string; the RL4RS_DIEN_OPTS environment variable is the default
        self.kernel_opts = parse_dien_opts(config.get('scorer_kernels', os.environ.get('RL4RS_DIEN_OPTS', '')))
        opt_bits = 0
        for name in self.kernel_opts:
            opt_bits |= _lib.DIEN_OPTS[name]
        cfg = _lib.DienCfg(self.L, self.E, self.U, self.Dn, self.Cn, int(config['category_hash_size']),
                           self.S, self.K, self.max_rows, self.max_slots, SCORER_MODES[precision], opt_bits)
        w = _lib.DienWeights()
        keep = []

        def fp(name):
            arr = np.ascontiguousarray(weights[name], dtype=np.float32)
            keep.append(arr)
            return arr.ctypes.data_as(_lib._FP)

        for name in ('cat_emb', 'dense_w1', 'dense_b1', 'dense_w2', 'dense_b2', 'seq_emb', 'obs_w', 'obs_b',
                     'out_w', 'out_b'):
            setattr(w, name, fp(name))
        for i in range(self.S):
            w.gru_gate_w[i] = fp('gru%d_gate_w' % i)
            w.gru_gate_b[i] = fp('gru%d_gate_b' % i)
            w.gru_cand_w[i] = fp('gru%d_cand_w' % i)
            w.gru_cand_b[i] = fp('gru%d_cand_b' % i)
            w.att_w1[i] = fp('att%d_w1' % i)
            w.att_b1[i] = fp('att%d_b1' % i)
            w.att_w2[i] = fp('att%d_w2' % i)
            w.att_b2[i] = fp('att%d_b2' % i)
            w.att_w3[i] = fp('att%d_w3' % i)
            w.att_b3[i] = fp('att%d_b3' % i)
            w.augru_gate_w[i] = fp('augru%d_gate_w' % i)
            w.augru_gate_b[i] = fp('augru%d_gate_b' % i)
            w.augru_cand_w[i] = fp('augru%d_cand_w' % i)
            w.augru_cand_b[i] = fp('augru%d_cand_b' % i)
        h = C.c_void_p()
        with torch.cuda.device(self.device):
            check(self.lib.rl4rs_dien_create(C.byref(cfg), C.byref(w), _stream(), C.byref(h)))
        self.h = h

    def close(self):
        if getattr(self, 'h', None) is not None and self.h:
            self.lib.rl4rs_dien_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def encode(self, s, ids, slot_base=0):
        """ids: int32 device tensor [n, L] (or a raw (ptr, n) pair)."""
        if isinstance(ids, tuple):
            ptr, n = ids
        else:
            ids = _dev_tensor(ids, torch.int32, self.device)
            assert ids.dim() == 2 and ids.shape[1] == self.L
            ptr, n = _ptr(ids), ids.shape[0]
            self._keep_ids = ids
        check(self.lib.rl4rs_dien_encode(self.h, s, ptr, n, slot_base, _stream()))

    def forward(self, R, group, dense, cat, slots, want_obs=True, want_prob=False, obs_out=None):
        """dense/cat: device tensors or raw c_void_p pointers; slots: int32 device tensor [S, R/group]."""
        dp = dense if isinstance(dense, C.c_void_p) else _ptr(dense)
        cp = cat if isinstance(cat, C.c_void_p) else _ptr(cat)
        assert slots.dtype == torch.int32 and slots.numel() == self.S * (R // group)
        obs = None
        if want_obs:
            obs = obs_out if obs_out is not None else torch.empty((R, 256), dtype=torch.float32, device=self.device)
        prob = torch.empty(R, dtype=torch.float32, device=self.device) if want_prob else None
        check(self.lib.rl4rs_dien_forward(self.h, R, group, dp, cp, _ptr(slots), _ptr(obs), _ptr(prob), _stream()))
        return obs, prob

    def head_prob(self, obs):
        R = obs.shape[0]
        assert obs.dtype == torch.float32 and obs.is_contiguous() and obs.shape[1] == 256
        prob = torch.empty(R, dtype=torch.float32, device=self.device)
        check(self.lib.rl4rs_dien_head_prob(self.h, R, _ptr(obs), _ptr(prob), _stream()))
        return prob

    def snapshot(self, which, rows):
        p = C.c_void_p()
        n = C.c_int64()
        check(self.lib.rl4rs_dien_buffer(self.h, which, C.byref(p), C.byref(n)))
        if which == DIEN_ALL_FEATURE:
            shape = (self.max_rows, n.value // (4 * self.max_rows))      # F, or the Kh columns of the table form
        elif which == DIEN_SCORES:
            shape = (self.S, self.max_rows, self.L)
        elif which == DIEN_QUERY:
            shape = (self.max_rows, self.E)
        else:
            shape = (self.max_slots, self.L, self.E)
        out = torch.empty(shape, dtype=torch.float32, device=self.device)
        assert out.numel() * 4 == n.value
        check(self.lib.rl4rs_copy_d2d(_ptr(out), p, n.value, _stream()))
        return out

    # profiling (bench.py roofline)
    def set_row_order(self, order):
        """Processing order of the env groups of the following forwards (int32 device tensor, a permutation; None = identity):
        rl4rs_dien_set_row_order - a locality hint, results unchanged."""
        if order is not None:
            assert order.dtype == torch.int32 and order.is_contiguous() and order.is_cuda
        self._row_order = order                     # keep it alive
        check(self.lib.rl4rs_dien_set_row_order(self.h, _ptr(order), 0 if order is None else int(order.numel())))

    def set_augru_rows(self, rows):
        """Row-tile form of k_augru_x for the following forwards: 0 automatic, 32 or 64 (rl4rs_dien_set_augru_rows)."""
        check(self.lib.rl4rs_dien_set_augru_rows(self.h, int(rows)))

    def set_profiling(self, on):
        """0 / False off, 1 / True every kernel class, 2 only the AUGRU recurrence (rl4rs_dien_set_profiling)."""
        check(self.lib.rl4rs_dien_set_profiling(self.h, 2 if on == 2 else (1 if on else 0)))

    def profile_reset(self):
        check(self.lib.rl4rs_dien_profile_reset(self.h))

    def profile(self):
        out = {}
        for k in range(self.lib.rl4rs_dien_kernel_count()):
            ms = C.c_double()
            n = C.c_int64()
            check(self.lib.rl4rs_dien_profile_read(self.h, k, C.byref(ms), C.byref(n)))
            name = self.lib.rl4rs_dien_kernel_name(k).decode()
            if name == 'augru':
                name = self.augru_kernel                # the key bench.py's roofline looks up
            else:
                buf = C.create_string_buffer(160)       # what this handle really launches (mode- and option-dependent)
                check(self.lib.rl4rs_dien_kernel_label(self.h, k, buf, 160))
                name = buf.value.decode()
            out[name] = (ms.value, n.value)
        return out

    @property
    def scorer_mode(self):
        """'fp32' or 'fp16x2': what the handle resolved config['scorer_precision'] to."""
        m = C.c_int32()
        check(self.lib.rl4rs_dien_scorer_mode(self.h, C.byref(m)))
        return 'fp16x2' if m.value == 2 else 'fp32'

    def check_status(self):
        """Synchronising check of the handle's status bits: raises if the fp16x2 scorer left the fp16 range."""
        f = C.c_int32()
        check(self.lib.rl4rs_dien_status(self.h, C.byref(f), _stream()))
        if f.value & 1:
            raise _lib.Rl4rsHipError(
                "fp16x2 scorer: a recurrent state left the fp16 range (|h| >= 6e4 or NaN); the affected forwards are "
                "invalid - use config['scorer_precision'] = 'fp32' for this model")

    @property
    def augru_kernel(self):
        if self.scorer_mode == 'fp16x2' and 'augru_h16' not in self.kernel_opts:
            return 'k_augru_x'                      # dien.hip: the default fp16x2 recurrence ('augru_h16' selects the first generation)
        return AUGRU_KERNELS[SCORER_MODES[self.scorer_mode]]


SIMNET_ALGOS = {'dnn': 1, 'widedeep': 2, 'lstm': 3}          # include/rl4rs_hip.h RL4RS_SIMNET_*


class DeviceSimnet(object):
    """rl4rs_simnet handle: the dnn / widedeep / lstm simulator families (same calling pattern as DeviceDien)."""

    def __init__(self, config, weights, max_rows, max_slots, algo=None, device=None):
        _lib.require_device()
        self.lib = _lib.load()
        self.device = torch.device('cuda', torch.cuda.current_device()) if device is None else torch.device(device)
        self.algo = str(algo if algo is not None else config.get('algo')).lower()
        if self.algo not in SIMNET_ALGOS:
            raise ValueError("algo must be one of %s (got %r)" % (sorted(SIMNET_ALGOS), self.algo))
        self.S = int(config['seq_num'])
        self.L = int(config['maxlen'])
        self.max_rows, self.max_slots = int(max_rows), int(max_slots)
        cfg = _lib.SimnetCfg(SIMNET_ALGOS[self.algo], self.L, int(config['emb_size']), int(config['hidden_units']),
                             int(config['dense_feature_num']), int(config['category_feature_num']),
                             int(config['category_hash_size']), self.S, int(config['class_num']), self.max_rows,
                             self.max_slots)
        w = _lib.SimnetWeights()
        keep = []

        def fp(name):
            arr = np.ascontiguousarray(weights[name], dtype=np.float32)
            keep.append(arr)
            return arr.ctypes.data_as(_lib._FP)

        for name in ('cat_emb', 'seq_emb', 'dense_w1', 'dense_b1', 'dense_w2', 'dense_b2', 'fc_w', 'fc_b', 'obs_w',
                     'obs_b', 'out_w', 'out_b', 'cat_gru_kernel', 'cat_gru_recurrent', 'cat_gru_bias'):
            if name in weights:
                setattr(w, name, fp(name))
        for i in range(self.S):
            if 'seq%d_gru_kernel' % i in weights:
                w.seq_gru_kernel[i] = fp('seq%d_gru_kernel' % i)
                w.seq_gru_recurrent[i] = fp('seq%d_gru_recurrent' % i)
                w.seq_gru_bias[i] = fp('seq%d_gru_bias' % i)
        h = C.c_void_p()
        with torch.cuda.device(self.device):
            check(self.lib.rl4rs_simnet_create(C.byref(cfg), C.byref(w), _stream(), C.byref(h)))
        self.h = h
        d = C.c_int32()
        check(self.lib.rl4rs_simnet_obs_dim(self.h, C.byref(d)))
        self.obs_dim = d.value

    def close(self):
        if getattr(self, 'h', None) is not None and self.h:
            self.lib.rl4rs_simnet_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def encode(self, s, ids, slot_base=0):
        if isinstance(ids, tuple):
            ptr, n = ids
        else:
            ids = _dev_tensor(ids, torch.int32, self.device)
            assert ids.dim() == 2 and ids.shape[1] == self.L
            ptr, n = _ptr(ids), ids.shape[0]
            self._keep_ids = ids
        check(self.lib.rl4rs_simnet_encode(self.h, s, ptr, n, slot_base, _stream()))

    def forward(self, R, group, dense, cat, slots, want_obs=True, want_prob=False, obs_out=None):
        dp = dense if isinstance(dense, C.c_void_p) else _ptr(dense)
        cp = cat if isinstance(cat, C.c_void_p) else _ptr(cat)
        assert slots.dtype == torch.int32 and slots.numel() == self.S * (R // group)
        obs = None
        if want_obs:
            obs = obs_out if obs_out is not None else torch.empty((R, self.obs_dim), dtype=torch.float32, device=self.device)
        prob = torch.empty(R, dtype=torch.float32, device=self.device) if want_prob else None
        check(self.lib.rl4rs_simnet_forward(self.h, R, group, dp, cp, _ptr(slots), _ptr(obs), _ptr(prob), _stream()))
        return obs, prob

    def head_prob(self, obs):
        R = obs.shape[0]
        assert obs.dtype == torch.float32 and obs.is_contiguous() and obs.shape[1] == self.obs_dim
        prob = torch.empty(R, dtype=torch.float32, device=self.device)
        check(self.lib.rl4rs_simnet_head_prob(self.h, R, _ptr(obs), _ptr(prob), _stream()))
        return prob

    def check_status(self):
        pass

    # no-op profiling hooks so bench / facade code can treat every scorer alike
    def set_profiling(self, on):
        pass

    def profile_reset(self):
        pass

    def profile(self):
        return {}


SIMTRAIN_ORDER = {
    'dnn': ('cat_emb', 'dense_w1', 'dense_b1', 'dense_w2', 'dense_b2', 'fc_w', 'fc_b', 'obs_w', 'obs_b', 'out_w', 'out_b'),
    'widedeep': ('cat_emb', 'seq_emb', 'dense_w1', 'dense_b1', 'dense_w2', 'dense_b2', 'fc_w', 'fc_b', 'out_w', 'out_b'),
    # lstm: + the keras GRUs (category GRU, then one per sequence input; the names carry the sequence index)
    'lstm': ('cat_emb', 'seq_emb', 'dense_w1', 'dense_b1', 'dense_w2', 'dense_b2', 'obs_w', 'obs_b', 'out_w', 'out_b',
             'cat_gru_kernel', 'cat_gru_recurrent', 'cat_gru_bias'),
}


def _simtrain_names(algo, seq_num):
    names = list(SIMTRAIN_ORDER[algo])
    if algo == 'lstm':
        for i in range(seq_num):
            names += ['seq%d_gru_kernel' % i, 'seq%d_gru_recurrent' % i, 'seq%d_gru_bias' % i]
    return names


class DeviceSimTrainer(object):
    """rl4rs_simtrain handle: supervised training of the 'dnn' / 'widedeep' / 'lstm' simulators on the device
    (script/supervised_train.py): forward with dropout, keras binary_crossentropy, backward, Adam."""

    def __init__(self, config, weights, max_batch=256, algo=None, device=None):
        _lib.require_device()
        self.lib = _lib.load()
        self.device = torch.device('cuda', torch.cuda.current_device()) if device is None else torch.device(device)
        self.config = dict(config)
        self.algo = str(algo if algo is not None else config.get('algo', 'dnn')).lower()
        if self.algo not in SIMTRAIN_ORDER:
            raise NotImplementedError("device-side simulator training exists for %s (got %r)" % (sorted(SIMTRAIN_ORDER), self.algo))
        self.Cn, self.Dn = int(config['category_feature_num']), int(config['dense_feature_num'])
        self.S, self.L = int(config['seq_num']), int(config['maxlen'])
        self.max_batch = int(max_batch)
        cfg = _lib.SimnetCfg(SIMNET_ALGOS[self.algo], self.L, int(config['emb_size']), int(config['hidden_units']),
                             self.Dn, self.Cn, int(config['category_hash_size']), self.S,
                             int(config['class_num']), self.max_batch, 1)
        w = _lib.SimnetWeights()
        keep = []
        self.shapes = []
        for name in _simtrain_names(self.algo, self.S):
            arr = np.ascontiguousarray(weights[name], dtype=np.float32)
            keep.append(arr)
            self.shapes.append((name, arr.shape))
            ptr = arr.ctypes.data_as(_lib._FP)
            if name.startswith('seq') and '_gru_' in name:
                getattr(w, 'seq_gru_' + name.split('_gru_')[1])[int(name[3:name.index('_')])] = ptr
            else:
                setattr(w, name, ptr)
        h = C.c_void_p()
        with torch.cuda.device(self.device):
            check(self.lib.rl4rs_simtrain_create(C.byref(cfg), C.byref(w), self.max_batch, _stream(), C.byref(h)))
        self.h = h
        self.iteration = 0
        self._fn = dict(destroy=self.lib.rl4rs_simtrain_destroy, params=self.lib.rl4rs_simtrain_params,
                        masks=self.lib.rl4rs_simtrain_masks, grad=self.lib.rl4rs_simtrain_grad, step=self.lib.rl4rs_simtrain_step)

    def close(self):
        if getattr(self, 'h', None) is not None and self.h:
            self._fn['destroy'](self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _flat(self, which):
        p, g, n = C.c_void_p(), C.c_void_p(), C.c_int64()
        check(self._fn['params'](self.h, C.byref(p), C.byref(g), C.byref(n)))
        out = torch.empty(n.value, dtype=torch.float32, device=self.device)
        check(self.lib.rl4rs_copy_d2d(_ptr(out), p if which == 'params' else g, n.value * 4, _stream()))
        return out

    def _split(self, flat):
        out, o = {}, 0
        for name, shape in self.shapes:
            k = int(np.prod(shape))
            out[name] = flat[o:o + k].reshape(shape)
            o += k
        return out

    def weights(self):
        """Current parameters as a dict of device tensors (same names as rl4rs_amd.nets.simnets.simnet_spec)."""
        return self._split(self._flat('params'))

    def gradients(self):
        return self._split(self._flat('grad'))

    def masks(self, N):
        """The dropout keep-masks [N, hidden_units] (uint8) of the last grad / step call."""
        m1, m2 = C.c_void_p(), C.c_void_p()
        check(self._fn['masks'](self.h, C.byref(m1), C.byref(m2)))
        U = int(self.config['hidden_units'])
        out = []
        for m in (m1, m2):
            t = torch.empty((N, U), dtype=torch.uint8, device=self.device)
            check(self.lib.rl4rs_copy_d2d(_ptr(t), m, N * U, _stream()))
            out.append(t)
        return out

    def _batch(self, dense, cat, labels, seqs):
        N = dense.shape[0]
        assert dense.dtype == torch.float32 and dense.shape == (N, self.Dn) and dense.is_contiguous()
        assert cat.dtype == torch.int32 and cat.shape == (N, self.Cn) and cat.is_contiguous()
        labels = labels.to(torch.int32).contiguous()
        assert labels.shape == (N,)
        sp = None
        if self.algo != 'dnn':
            assert seqs is not None and len(seqs) == self.S, "every family but dnn needs the seq_num sequence inputs"
            for q in seqs:
                assert q.dtype == torch.int32 and q.shape == (N, self.L) and q.is_contiguous()
            sp = (C.c_void_p * self.S)(*[_ptr(q) for q in seqs])
        return N, labels, sp

    def grad(self, dense, cat, labels, seqs=None, dropout_rate=0.2, seed=0, step=0):
        """Forward + loss + backward; returns the mean loss (device scalar tensor)."""
        N, labels, sp = self._batch(dense, cat, labels, seqs)
        loss = torch.empty(1, dtype=torch.float32, device=self.device)
        check(self._fn['grad'](self.h, N, _ptr(dense), _ptr(cat), sp, _ptr(labels), dropout_rate, seed, step,
                                           _ptr(loss), _stream()))
        return loss

    def step(self, dense, cat, labels, seqs=None, lr=1e-3, beta1=0.9, beta2=0.999, eps=1e-7, dropout_rate=0.2, seed=0):
        N, labels, sp = self._batch(dense, cat, labels, seqs)
        loss = torch.empty(1, dtype=torch.float32, device=self.device)
        check(self._fn['step'](self.h, N, _ptr(dense), _ptr(cat), sp, _ptr(labels), lr, beta1, beta2, eps,
                                           dropout_rate, seed, self.iteration, _ptr(loss), _stream()))
        self.iteration += 1
        return loss


DIENTRAIN_BASE = ('cat_emb', 'seq_emb', 'dense_w1', 'dense_b1', 'dense_w2', 'dense_b2', 'obs_w', 'obs_b', 'out_w', 'out_b')
DIENTRAIN_SEQ = ('gru%d_gate_w', 'gru%d_gate_b', 'gru%d_cand_w', 'gru%d_cand_b', 'att%d_w1', 'att%d_b1', 'att%d_w2', 'att%d_b2',
                 'att%d_w3', 'att%d_b3', 'augru%d_gate_w', 'augru%d_gate_b', 'augru%d_cand_w', 'augru%d_cand_b')


class DeviceDienTrainer(DeviceSimTrainer):
    """rl4rs_dientrain handle: supervised training of the DIEN simulator on the device (same interface as DeviceSimTrainer)."""

    def __init__(self, config, weights, max_batch=256, device=None):
        _lib.require_device()
        self.lib = _lib.load()
        self.device = torch.device('cuda', torch.cuda.current_device()) if device is None else torch.device(device)
        self.config = dict(config)
        self.algo = 'dien'
        self.Cn, self.Dn = int(config['category_feature_num']), int(config['dense_feature_num'])
        self.S, self.L = int(config['seq_num']), int(config['maxlen'])
        self.max_batch = int(max_batch)
        cfg = _lib.DienCfg(self.L, int(config['emb_size']), int(config['hidden_units']), self.Dn, self.Cn,
                           int(config['category_hash_size']), self.S, int(config['class_num']), self.max_batch, 1, 0)
        w = _lib.DienWeights()
        keep = []
        self.shapes = []
        names = list(DIENTRAIN_BASE) + [n % i for i in range(self.S) for n in DIENTRAIN_SEQ]
        for name in names:
            arr = np.ascontiguousarray(weights[name], dtype=np.float32)
            keep.append(arr)
            self.shapes.append((name, arr.shape))
            ptr = arr.ctypes.data_as(_lib._FP)
            if name in DIENTRAIN_BASE:
                setattr(w, name, ptr)
            else:                                   # 'gru0_gate_w' -> field gru_gate_w[0]
                head, tail = name.split('_', 1)
                idx = int(''.join(ch for ch in head if ch.isdigit()))
                getattr(w, ''.join(ch for ch in head if not ch.isdigit()) + '_' + tail)[idx] = ptr
        h = C.c_void_p()
        with torch.cuda.device(self.device):
            check(self.lib.rl4rs_dientrain_create(C.byref(cfg), C.byref(w), self.max_batch, _stream(), C.byref(h)))
        self.h = h
        self.iteration = 0
        self._fn = dict(destroy=self.lib.rl4rs_dientrain_destroy, params=self.lib.rl4rs_dientrain_params,
                        masks=self.lib.rl4rs_dientrain_masks, grad=self.lib.rl4rs_dientrain_grad, step=self.lib.rl4rs_dientrain_step)


class DeviceRawPolicy(object):
    """rl4rs_rawpolicy handle: the raw-state policy encoder (rllib_rawstate_model.py) with the action-mask rule,
    forward only.  Inputs are the raw feature tensors an env with config['rawstate_as_obs'] exposes."""

    def __init__(self, config, weights, max_rows, device=None):
        _lib.require_device()
        self.lib = _lib.load()
        self.device = torch.device('cuda', torch.cuda.current_device()) if device is None else torch.device(device)
        self.S, self.L, self.A = int(config['seq_num']), int(config['maxlen']), int(config['action_size'])
        self.Cn, self.Dn = int(config['category_feature_num']), int(config['dense_feature_num'])
        self.W = (self.A + 31) // 32
        self.max_rows = int(max_rows)
        cfg = _lib.RawPolicyCfg(self.L, int(config['emb_size']), int(config['hidden_units']), self.Dn, self.Cn,
                                int(config['category_hash_size']), self.S, self.A, self.max_rows)
        w = _lib.RawPolicyWeights()
        keep = []
        for name, _ in _lib.RawPolicyWeights._fields_:
            arr = np.ascontiguousarray(weights[name], dtype=np.float32)
            keep.append(arr)
            setattr(w, name, arr.ctypes.data_as(_lib._FP))
        h = C.c_void_p()
        with torch.cuda.device(self.device):
            check(self.lib.rl4rs_rawpolicy_create(C.byref(cfg), C.byref(w), _stream(), C.byref(h)))
        self.h = h

    def close(self):
        if getattr(self, 'h', None) is not None and self.h:
            self.lib.rl4rs_rawpolicy_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _inputs(self, cat, dense, seqs, mask_bits):
        N = cat.shape[0]
        assert cat.dtype == torch.int32 and cat.shape == (N, self.Cn) and cat.is_contiguous()
        assert dense.dtype == torch.float32 and dense.shape == (N, self.Dn) and dense.is_contiguous()
        assert len(seqs) == self.S
        for q in seqs:
            assert q.dtype == torch.int32 and q.shape == (N, self.L) and q.is_contiguous()
        if mask_bits is not None:
            assert mask_bits.dtype == torch.int32 and mask_bits.shape == (N, self.W) and mask_bits.is_contiguous()
        sp = (C.c_void_p * self.S)(*[_ptr(q) for q in seqs])
        return N, sp

    def _outs(self, N, want_logits):
        f = lambda: torch.empty(N, dtype=torch.float32, device=self.device)
        lg = torch.empty((N, self.A), dtype=torch.float32, device=self.device) if want_logits else None
        return f(), f(), f(), lg

    def act(self, cat, dense, seqs, mask_bits=None, seed=0, step=0, want_logits=False):
        """-> actions [N] int32, logp, value, entropy (float32 [N]), masked logits [N, A] or None."""
        N, sp = self._inputs(cat, dense, seqs, mask_bits)
        a = torch.empty(N, dtype=torch.int32, device=self.device)
        lp, v, ent, lg = self._outs(N, want_logits)
        check(self.lib.rl4rs_rawpolicy_act(self.h, N, _ptr(cat), _ptr(dense), sp, _ptr(mask_bits), seed, step, _ptr(a),
                                           _ptr(lp), _ptr(v), _ptr(ent), _ptr(lg), _stream()))
        return a, lp, v, ent, lg

    def evaluate(self, cat, dense, seqs, actions, mask_bits=None, want_logits=False):
        N, sp = self._inputs(cat, dense, seqs, mask_bits)
        actions = actions.to(torch.int32).contiguous()
        lp, v, ent, lg = self._outs(N, want_logits)
        check(self.lib.rl4rs_rawpolicy_evaluate(self.h, N, _ptr(cat), _ptr(dense), sp, _ptr(mask_bits), _ptr(actions),
                                                _ptr(lp), _ptr(v), _ptr(ent), _ptr(lg), _stream()))
        return lp, v, ent, lg


class _DevAlias(object):
    """__cuda_array_interface__ carrier: lets torch alias device memory owned by a library handle."""

    def __init__(self, ptr, count, owner):
        self.owner = owner
        self.__cuda_array_interface__ = {'shape': (int(count),), 'typestr': '<f4', 'data': (int(ptr), False), 'version': 2}


def _alias_f32(ptr, count, device, owner):
    with torch.cuda.device(device):
        t = torch.as_tensor(_DevAlias(ptr, count, owner), device=device)
    assert t.data_ptr() == int(ptr) and t.dtype == torch.float32
    return t


class DeviceRawTrainer(DeviceRawPolicy):
    """rl4rs_rawtrain handle: the raw-state policy with A2C / PPO loss, backward and Adam on the device."""
    A2C, PPO = 0, 1
    ORDER = ('cat_emb', 'seq_emb', 'dense_w1', 'dense_b1', 'dense_w2', 'dense_b2', 'ctx_w', 'ctx_b', 'head_w', 'head_b')

    def __init__(self, config, weights, max_rows, device=None):
        _lib.require_device()
        self.lib = _lib.load()
        self.device = torch.device('cuda', torch.cuda.current_device()) if device is None else torch.device(device)
        self.S, self.L, self.A = int(config['seq_num']), int(config['maxlen']), int(config['action_size'])
        self.Cn, self.Dn = int(config['category_feature_num']), int(config['dense_feature_num'])
        self.W = (self.A + 31) // 32
        self.max_rows = int(max_rows)
        E, U = int(config['emb_size']), int(config['hidden_units'])
        H = int(config['category_hash_size'])
        cfg = _lib.RawPolicyCfg(self.L, E, U, self.Dn, self.Cn, H, self.S, self.A, self.max_rows)
        w = _lib.RawPolicyWeights()
        keep = []
        for name, _ in _lib.RawPolicyWeights._fields_:
            arr = np.ascontiguousarray(weights[name], dtype=np.float32)
            keep.append(arr)
            setattr(w, name, arr.ctypes.data_as(_lib._FP))
        h = C.c_void_p()
        with torch.cuda.device(self.device):
            check(self.lib.rl4rs_rawtrain_create(C.byref(cfg), C.byref(w), _stream(), C.byref(h)))
        self.h = h
        self.shapes = [('cat_emb', (H, E)), ('seq_emb', (H, E)), ('dense_w1', (self.Dn, U)), ('dense_b1', (U,)), ('dense_w2', (U, U)),
                       ('dense_b2', (U,)), ('ctx_w', (self.S * E + U + E, 256)), ('ctx_b', (256,)), ('head_w', (256, self.A + 1)),
                       ('head_b', (self.A + 1,))]

    def close(self):
        if getattr(self, 'h', None) is not None and self.h:
            self.lib.rl4rs_rawtrain_destroy(self.h)
            self.h = None

    def _flat(self, which):
        p, g, n = C.c_void_p(), C.c_void_p(), C.c_int64()
        check(self.lib.rl4rs_rawtrain_params(self.h, C.byref(p), C.byref(g), C.byref(n)))
        out = torch.empty(n.value, dtype=torch.float32, device=self.device)
        check(self.lib.rl4rs_copy_d2d(_ptr(out), p if which == 'params' else g, n.value * 4, _stream()))
        return out

    def _split(self, flat):
        out, o = {}, 0
        for name, shape in self.shapes:
            k = int(np.prod(shape))
            out[name] = flat[o:o + k].reshape(shape)
            o += k
        return out

    def weights(self):
        """head_w = [out_w | value_w], head_b = [out_b | value_b]."""
        return self._split(self._flat('params'))

    def gradients(self):
        return self._split(self._flat('grad'))

    def flat_view(self, which):
        """ZERO-COPY torch view of the handle's flat parameter ('params') or gradient ('grad') buffer: a data-parallel trainer
        all-reduces the gradient in place between loss_grad and adam_step, and broadcasts the parameters at start."""
        p, g, n = C.c_void_p(), C.c_void_p(), C.c_int64()
        check(self.lib.rl4rs_rawtrain_params(self.h, C.byref(p), C.byref(g), C.byref(n)))
        return _alias_f32((p if which == 'params' else g).value, n.value, self.device, self)

    def table_rows(self):
        """(offset, rows, width) of the two embedding tables inside the flat buffers (cat_emb, seq_emb come first)."""
        (_, (H, E)) = self.shapes[0]
        return [(0, H, E), (H * E, H, E)]

    def act(self, cat, dense, seqs, mask_bits=None, seed=0, step=0, want_logits=False):
        N, sp = self._inputs(cat, dense, seqs, mask_bits)
        a = torch.empty(N, dtype=torch.int32, device=self.device)
        lp, v, ent, lg = self._outs(N, want_logits)
        check(self.lib.rl4rs_rawtrain_act(self.h, N, _ptr(cat), _ptr(dense), sp, _ptr(mask_bits), seed, step, _ptr(a), _ptr(lp),
                                          _ptr(v), _ptr(ent), _ptr(lg), _stream()))
        return a, lp, v, ent, lg

    def evaluate(self, cat, dense, seqs, actions, mask_bits=None, want_logits=False):
        N, sp = self._inputs(cat, dense, seqs, mask_bits)
        actions = actions.to(torch.int32).contiguous()
        lp, v, ent, lg = self._outs(N, want_logits)
        check(self.lib.rl4rs_rawtrain_evaluate(self.h, N, _ptr(cat), _ptr(dense), sp, _ptr(mask_bits), _ptr(actions), _ptr(lp),
                                               _ptr(v), _ptr(ent), _ptr(lg), _stream()))
        return lp, v, ent, lg

    def loss_grad(self, algo, cat, dense, seqs, actions, adv, ret, mask_bits=None, old_logp=None, old_value=None, old_logits=None,
                  vf_coeff=0.5, ent_coeff=0.01, clip=0.3, vf_clip=500.0, kl_coeff=0.2):
        """-> stats [pi_loss, vf_loss, entropy, kl] sums (device tensor); the gradient stays in the handle (gradients())."""
        N, sp = self._inputs(cat, dense, seqs, mask_bits)
        f = lambda t: None if t is None else t.to(torch.float32).contiguous()
        actions = actions.to(torch.int32).contiguous()
        adv, ret, old_logp, old_value, old_logits = f(adv), f(ret), f(old_logp), f(old_value), f(old_logits)
        stats = torch.empty(4, dtype=torch.float32, device=self.device)
        check(self.lib.rl4rs_rawtrain_loss_grad(self.h, algo, N, _ptr(cat), _ptr(dense), sp, _ptr(mask_bits), _ptr(actions), _ptr(adv),
                                                _ptr(ret), _ptr(old_logp), _ptr(old_value), _ptr(old_logits), vf_coeff, ent_coeff,
                                                clip, vf_clip, kl_coeff, _ptr(stats), _stream()))
        return stats

    def adam_step(self, lr=1e-4, beta1=0.9, beta2=0.999, eps=1e-8, grad_clip=0.0):
        check(self.lib.rl4rs_rawtrain_adam_step(self.h, lr, beta1, beta2, eps, grad_clip, _stream()))


def gemm_f32(a, w, bias=None, act=0):
    """C = act(a @ w + bias) through rl4rs_gemm_f32 (tests)."""
    lib = _lib.load()
    M, K = a.shape
    K2, N = w.shape
    assert K == K2
    c = torch.empty((M, N), dtype=torch.float32, device=a.device)
    check(lib.rl4rs_gemm_f32(_ptr(a), a.stride(0), _ptr(w), w.stride(0), _ptr(bias), _ptr(c), N, M, N, K, act, _stream()))
    return c


def gemm_f32_packed(a, w_host, bias=None, act=0):
    """Same through the packed-weight kernel (w_host: numpy [K,N] float32)."""
    lib = _lib.load()
    M, K = a.shape
    w_host = np.ascontiguousarray(w_host, dtype=np.float32)
    K2, N = w_host.shape
    assert K == K2
    c = torch.empty((M, N), dtype=torch.float32, device=a.device)
    check(lib.rl4rs_gemm_f32_packed(_ptr(a), a.stride(0), w_host.ctypes.data_as(C.c_void_p), N, _ptr(bias), _ptr(c), N,
                                    M, N, K, act, _stream()))
    return c


def gemm_h16_packed(a, w_host, bias=None, act=0):
    """The fp16x2 form (operands as fp16 hi + lo pairs, three f16 MFMAs per product) of the packed-weight GEMM."""
    lib = _lib.load()
    M, K = a.shape
    w_host = np.ascontiguousarray(w_host, dtype=np.float32)
    K2, N = w_host.shape
    assert K == K2
    c = torch.empty((M, N), dtype=torch.float32, device=a.device)
    check(lib.rl4rs_gemm_h16_packed(_ptr(a), a.stride(0), w_host.ctypes.data_as(C.c_void_p), N, _ptr(bias), _ptr(c), N,
                                    M, N, K, act, _stream()))
    return c


class DevicePolicy(object):
    """rl4rs_policy handle: action-masked policy net (rllib_mask_model.py:7-64) with flat parameters."""
    A2C, PPO = 0, 1

    def __init__(self, obs_dim, hidden, action_size, max_rows, params=None, seed=0, device=None):
        _lib.require_device()
        self.lib = _lib.load()
        self.device = torch.device('cuda', torch.cuda.current_device()) if device is None else torch.device(device)
        self.obs_dim, self.hidden, self.action_size, self.max_rows = int(obs_dim), int(hidden), int(action_size), int(max_rows)
        self.n_params = self.lib.rl4rs_policy_param_count(self.obs_dim, self.hidden, self.action_size)
        if params is None:
            from .nets.policy import init_policy_params
            params = init_policy_params(self.obs_dim, self.hidden, self.action_size, seed)
        params = np.ascontiguousarray(params, dtype=np.float32)
        assert params.shape == (self.n_params,), (params.shape, self.n_params)
        h = C.c_void_p()
        with torch.cuda.device(self.device):
            check(self.lib.rl4rs_policy_create(self.obs_dim, self.hidden, self.action_size, self.max_rows,
                                               params.ctypes.data_as(C.c_void_p), _stream(), C.byref(h)))
        self.h = h
        self.W = (self.action_size + 31) // 32
        # RL4RS_POLICY_OPTS="ppo_fused=0,ppo_rows=16": defaults for A/B runs, read here (the library never reads the environment)
        for item in filter(None, (x.strip() for x in os.environ.get('RL4RS_POLICY_OPTS', '').split(','))):
            k, _, v = item.partition('=')
            self.set_option(k.strip(), int(v))

    def set_option(self, name, value):
        """Kernel-path selection of this handle (rl4rs_policy_set_option): 'tile', 'ppo_fused', 'ppo_rows', 'resident_wgs', 'ppo_std'."""
        if name not in _lib.POLICY_OPTS:
            raise ValueError("unknown policy option %r; known: %s" % (name, sorted(_lib.POLICY_OPTS)))
        check(self.lib.rl4rs_policy_set_option(self.h, _lib.POLICY_OPTS[name], int(value)))

    def close(self):
        if getattr(self, 'h', None) is not None and self.h:
            self.lib.rl4rs_policy_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def params(self):
        """Copy of the flat parameter buffer (device tensor)."""
        p = C.c_void_p()
        n = C.c_int32()
        check(self.lib.rl4rs_policy_params(self.h, C.byref(p), C.byref(n)))
        out = torch.empty(n.value, dtype=torch.float32, device=self.device)
        check(self.lib.rl4rs_copy_d2d(_ptr(out), p, n.value * 4, _stream()))
        return out

    def set_params(self, flat):
        p = C.c_void_p()
        n = C.c_int32()
        check(self.lib.rl4rs_policy_params(self.h, C.byref(p), C.byref(n)))
        flat = flat.to(device=self.device, dtype=torch.float32).contiguous()
        assert flat.numel() == n.value
        check(self.lib.rl4rs_copy_d2d(p, _ptr(flat), n.value * 4, _stream()))
        self._keep = flat

    def _mask(self, mask_bits, N):
        if mask_bits is None:
            return None
        assert mask_bits.dtype == torch.int32 and mask_bits.shape == (N, self.W) and mask_bits.is_contiguous()
        return mask_bits

    def act(self, obs, mask_bits=None, seed=0, step=0, want_logits=False, out=None):
        """Sample actions.  ``out`` = (actions i32 [N], logp f32 [N], value f32 [N], logits f32 [N, A] or None): contiguous
        tensors to fill in place (slices of a rollout buffer) instead of fresh ones."""
        N = obs.shape[0]
        assert obs.dtype == torch.float32 and obs.shape == (N, self.obs_dim) and obs.is_contiguous()
        m = self._mask(mask_bits, N)
        if out is not None:
            a, lp, v, lg = out
            assert a.dtype == torch.int32 and lp.dtype == torch.float32 and v.dtype == torch.float32
            assert a.shape == (N,) and lp.shape == (N,) and v.shape == (N,) and a.is_contiguous() and lp.is_contiguous() and v.is_contiguous()
            assert lg is None or (lg.dtype == torch.float32 and tuple(lg.shape) == (N, self.action_size) and lg.is_contiguous())
        else:
            a = torch.empty(N, dtype=torch.int32, device=self.device)
            lp = torch.empty(N, dtype=torch.float32, device=self.device)
            v = torch.empty(N, dtype=torch.float32, device=self.device)
            lg = torch.empty((N, self.action_size), dtype=torch.float32, device=self.device) if want_logits else None
        ent = torch.empty(N, dtype=torch.float32, device=self.device)
        check(self.lib.rl4rs_policy_act(self.h, N, _ptr(obs), _ptr(m), seed & 0xffffffff, step & 0xffffffff, _ptr(a),
                                        _ptr(lp), _ptr(v), _ptr(ent), _ptr(lg), _stream()))
        return a, lp, v, ent, lg

    def evaluate(self, obs, actions, mask_bits=None, want_logits=False):
        N = obs.shape[0]
        m = self._mask(mask_bits, N)
        actions = actions.to(torch.int32).contiguous()
        lp = torch.empty(N, dtype=torch.float32, device=self.device)
        v = torch.empty(N, dtype=torch.float32, device=self.device)
        ent = torch.empty(N, dtype=torch.float32, device=self.device)
        lg = torch.empty((N, self.action_size), dtype=torch.float32, device=self.device) if want_logits else None
        check(self.lib.rl4rs_policy_evaluate(self.h, N, _ptr(obs), _ptr(m), _ptr(actions), _ptr(lp), _ptr(v), _ptr(ent),
                                             _ptr(lg), _stream()))
        return lp, v, ent, lg

    def loss_grad(self, algo, obs, actions, adv, ret, mask_bits=None, old_logp=None, old_value=None, old_logits=None,
                  vf_coeff=0.5, ent_coeff=0.01, clip=0.3, vf_clip=500.0, kl_coeff=0.2, grad_out=None):
        N = obs.shape[0]
        m = self._mask(mask_bits, N)
        g = grad_out if grad_out is not None else torch.empty(self.n_params, dtype=torch.float32, device=self.device)
        stats = torch.empty(4, dtype=torch.float32, device=self.device)
        f = lambda t: None if t is None else t.to(torch.float32).contiguous()
        actions = actions.to(torch.int32).contiguous()
        adv, ret, old_logp, old_value, old_logits = f(adv), f(ret), f(old_logp), f(old_value), f(old_logits)
        check(self.lib.rl4rs_policy_loss_grad(self.h, algo, N, _ptr(obs), _ptr(m), _ptr(actions), _ptr(adv), _ptr(ret),
                                              _ptr(old_logp), _ptr(old_value), _ptr(old_logits), vf_coeff, ent_coeff, clip,
                                              vf_clip, kl_coeff, _ptr(g), _ptr(stats), _stream()))
        return g, stats

    def ppo_epoch(self, obs, actions, adv, ret, mask_bits, old_logp, old_value, old_logits, minibatch=256, vf_coeff=0.5,
                  ent_coeff=0.0, clip=0.3, vf_clip=500.0, kl_coeff=0.2, lr=1e-4, beta1=0.9, beta2=0.999, eps=1e-8,
                  grad_clip=0.0, grad_out=None):
        """One SGD pass over already shuffled samples (rl4rs_policy_ppo_epoch).  Returns 8 floats: [0:4] sums of
        {pi_loss, vf_loss, entropy, kl} over the last minibatch, [4:8] the same sums over every sample of the pass."""
        N = obs.shape[0]
        m = self._mask(mask_bits, N)
        g = grad_out if grad_out is not None else torch.empty(self.n_params, dtype=torch.float32, device=self.device)
        stats = torch.empty(8, dtype=torch.float32, device=self.device)
        f = lambda t: t.to(torch.float32).contiguous()
        obs, adv, ret, old_logp, old_value, old_logits = f(obs), f(adv), f(ret), f(old_logp), f(old_value), f(old_logits)
        actions = actions.to(torch.int32).contiguous()
        check(self.lib.rl4rs_policy_ppo_epoch(self.h, N, minibatch, _ptr(obs), _ptr(m), _ptr(actions), _ptr(adv), _ptr(ret),
                                              _ptr(old_logp), _ptr(old_value), _ptr(old_logits), vf_coeff, ent_coeff, clip,
                                              vf_clip, kl_coeff, lr, beta1, beta2, eps, grad_clip, _ptr(g), _ptr(stats),
                                              _stream()))
        return stats

    def ppo_minibatch_grad(self, mb_index, obs, actions, adv, ret, mask_bits, old_logp, old_value, old_logits, minibatch=256,
                           vf_coeff=0.5, ent_coeff=0.0, clip=0.3, vf_clip=500.0, kl_coeff=0.2, grad_out=None, stats_out=None):
        """Data-parallel form of the pass: gradient of minibatch ``mb_index`` of the shuffled samples, parameters untouched
        (rl4rs_policy_ppo_minibatch_grad).  All inputs must already be contiguous float32 / int32 device tensors of the whole
        pass (no per-call conversions: this runs once per minibatch).  -> (grad, stats[4] sums)."""
        N = obs.shape[0]
        g = grad_out if grad_out is not None else torch.empty(self.n_params, dtype=torch.float32, device=self.device)
        stats = stats_out if stats_out is not None else torch.empty(4, dtype=torch.float32, device=self.device)
        for t in (obs, adv, ret, old_logp, old_value, old_logits):
            assert t.dtype == torch.float32 and t.is_contiguous()
        assert actions.dtype == torch.int32 and actions.is_contiguous()
        m = self._mask(mask_bits, N)
        check(self.lib.rl4rs_policy_ppo_minibatch_grad(self.h, N, minibatch, mb_index, _ptr(obs), _ptr(m), _ptr(actions), _ptr(adv),
                                                       _ptr(ret), _ptr(old_logp), _ptr(old_value), _ptr(old_logits), vf_coeff,
                                                       ent_coeff, clip, vf_clip, kl_coeff, _ptr(g), _ptr(stats), _stream()))
        return g, stats

    def adam_step(self, grad, lr=1e-4, beta1=0.9, beta2=0.999, eps=1e-8, grad_clip=0.0):
        check(self.lib.rl4rs_policy_adam_step(self.h, _ptr(grad), lr, beta1, beta2, eps, grad_clip, _stream()))

    def check_status(self):
        """Synchronises; raises if a persistent PPO pass gave up at a grid barrier (its workgroups were not co-resident)."""
        f = C.c_int32()
        check(self.lib.rl4rs_policy_status(self.h, C.byref(f), _stream()))
        if f.value & 1:
            raise RuntimeError("the persistent PPO pass timed out at a grid barrier (workgroups not co-resident: the GPU is shared or "
                               "partitioned); the pass is incomplete - use DevicePolicy.set_option('ppo_fused', 0) for the per-minibatch kernels")

    def status_words(self):
        """int32 [2] tensor aliasing the handle's status words (word 1 != 0: a persistent pass timed out); no synchronisation."""
        if getattr(self, '_status_words', None) is None:
            p = C.c_void_p()
            check(self.lib.rl4rs_policy_status_words(self.h, C.byref(p)))
            self._status_words = _alias_f32(p.value, 2, self.device, self).view(torch.int32)
        return self._status_words

    PASS_TIMEOUT_MESSAGE = ("the persistent PPO pass timed out at a grid barrier (workgroups not co-resident: the GPU is shared or "
                            "partitioned); the pass is incomplete - use DevicePolicy.set_option('ppo_fused', 0) for the per-minibatch kernels")

    def adam_state(self):
        """(m, v) copies of the Adam moments and the step counter."""
        m, v, t = C.c_void_p(), C.c_void_p(), C.c_int64()
        check(self.lib.rl4rs_policy_adam_state(self.h, C.byref(m), C.byref(v), C.byref(t)))
        om = torch.empty(self.n_params, dtype=torch.float32, device=self.device)
        ov = torch.empty(self.n_params, dtype=torch.float32, device=self.device)
        check(self.lib.rl4rs_copy_d2d(_ptr(om), m, self.n_params * 4, _stream()))
        check(self.lib.rl4rs_copy_d2d(_ptr(ov), v, self.n_params * 4, _stream()))
        return om, ov, int(t.value)

    def set_adam_state(self, m, v, step):
        pm, pv, t = C.c_void_p(), C.c_void_p(), C.c_int64()
        check(self.lib.rl4rs_policy_adam_state(self.h, C.byref(pm), C.byref(pv), C.byref(t)))
        m = m.to(device=self.device, dtype=torch.float32).contiguous()
        v = v.to(device=self.device, dtype=torch.float32).contiguous()
        assert m.numel() == self.n_params and v.numel() == self.n_params
        check(self.lib.rl4rs_copy_d2d(pm, _ptr(m), self.n_params * 4, _stream()))
        check(self.lib.rl4rs_copy_d2d(pv, _ptr(v), self.n_params * 4, _stream()))
        check(self.lib.rl4rs_policy_set_adam_step(self.h, int(step)))
        self._keep_adam = (m, v)


class DeviceQNet(object):
    """rl4rs_qnet handle: one offline-RL network (the reference's ``CustomVectorEncoder`` of rl4rs/nets/cql/encoder.py:9-67,
    or d3rlpy's plain ``VectorEncoder``, + d3rlpy's Linear head) with forward, backward and torch-style Adam on the device.
    ``params``: dict of float32 arrays stored [in, out]: fc1_w, fc1_b, (emb,) fc2_w, fc2_b, head_w, head_b."""

    def __init__(self, obs_dim, action_size, params, mask_size=0, emb_size=32, hidden1=256, hidden2=256, location_mask=None,
                 special_items=None, max_rows=256, device=None):
        _lib.require_device()
        self.lib = _lib.load()
        self.device = torch.device('cuda', torch.cuda.current_device()) if device is None else torch.device(device)
        self.D, self.A, self.M, self.ES = int(obs_dim), int(action_size), int(mask_size), int(emb_size)
        self.H1, self.H2 = int(hidden1), int(hidden2)
        self.custom = self.M > 0
        self.max_rows = int(max_rows)
        f2 = self.H1 + self.M * self.ES if self.custom else self.H1
        n2 = self.A if self.custom else self.H2
        self.shapes = [('fc1_w', (self.D, self.H1)), ('fc1_b', (self.H1,))]
        if self.custom:
            self.shapes.append(('emb', (self.A, self.ES)))
        self.shapes += [('fc2_w', (f2, n2)), ('fc2_b', (n2,)), ('head_w', (n2, self.A)), ('head_b', (self.A,))]
        flat = []
        for name, shape in self.shapes:
            arr = np.ascontiguousarray(params[name], dtype=np.float32)
            if tuple(arr.shape) != tuple(shape):
                raise ValueError('qnet parameter %r has shape %r, expected %r' % (name, tuple(arr.shape), tuple(shape)))
            flat.append(arr.reshape(-1))
        flat = np.ascontiguousarray(np.concatenate(flat))
        loc = sp = None
        n_layers = 0
        if self.custom:
            loc = np.ascontiguousarray(np.asarray(location_mask) >= 0.5, dtype=np.uint8)
            if loc.ndim != 2 or loc.shape[1] != self.A:
                raise ValueError('location_mask must be [n_layers, action_size]')
            n_layers = loc.shape[0]
            sp = np.zeros(self.A, dtype=np.uint8)
            sp[np.asarray(list(special_items), dtype=np.int64)] = 1
        cfg = _lib.QNetCfg(self.D, self.A, self.M, self.ES, self.H1, 0 if self.custom else self.H2, n_layers, self.max_rows)
        h = C.c_void_p()
        with torch.cuda.device(self.device):
            check(self.lib.rl4rs_qnet_create(C.byref(cfg), flat.ctypes.data_as(_lib._FP),
                                             loc.ctypes.data_as(C.c_void_p) if loc is not None else None,
                                             sp.ctypes.data_as(C.c_void_p) if sp is not None else None, _stream(), C.byref(h)))
        self.h = h
        self.n_params = int(flat.size)

    def close(self):
        if getattr(self, 'h', None) is not None and self.h:
            self.lib.rl4rs_qnet_destroy(self.h)
            self.h = None

    __del__ = close

    def _buffers(self):
        p, g, n = C.c_void_p(), C.c_void_p(), C.c_int64()
        check(self.lib.rl4rs_qnet_params(self.h, C.byref(p), C.byref(g), C.byref(n)))
        return p, g, n.value

    def _flat(self, which):
        p, g, n = self._buffers()
        out = torch.empty(n, dtype=torch.float32, device=self.device)
        check(self.lib.rl4rs_copy_d2d(_ptr(out), p if which == 'params' else g, n * 4, _stream()))
        return out

    def _split(self, flat):
        out, o = {}, 0
        for name, shape in self.shapes:
            k = int(np.prod(shape))
            out[name] = flat[o:o + k].reshape(shape)
            o += k
        return out

    def weights(self):
        return self._split(self._flat('params'))

    def gradients(self):
        return self._split(self._flat('grad'))

    def flat_gradient(self):
        return self._flat('grad')

    def set_flat_gradient(self, flat):
        _, g, n = self._buffers()
        assert flat.numel() == n and flat.dtype == torch.float32 and flat.is_contiguous()
        check(self.lib.rl4rs_copy_d2d(g, _ptr(flat), n * 4, _stream()))

    _PREFIX = 'rl4rs_qnet'

    def flat_params(self):
        return self._flat('params')

    def set_flat_params(self, flat):
        p, _, n = self._buffers()
        assert flat.numel() == n and flat.dtype == torch.float32 and flat.is_contiguous()
        check(self.lib.rl4rs_copy_d2d(p, _ptr(flat), n * 4, _stream()))

    def adam_state(self):
        """(m, v, step): copies of the Adam moments and the step count (checkpointing)."""
        pm, pv, t = C.c_void_p(), C.c_void_p(), C.c_int64()
        check(getattr(self.lib, self._PREFIX + '_adam_state')(self.h, C.byref(pm), C.byref(pv), C.byref(t)))
        n = self.n_params
        m = torch.empty(n, dtype=torch.float32, device=self.device)
        v = torch.empty(n, dtype=torch.float32, device=self.device)
        check(self.lib.rl4rs_copy_d2d(_ptr(m), pm, n * 4, _stream()))
        check(self.lib.rl4rs_copy_d2d(_ptr(v), pv, n * 4, _stream()))
        return m, v, int(t.value)

    def set_adam_state(self, m, v, step):
        pm, pv, t = C.c_void_p(), C.c_void_p(), C.c_int64()
        check(getattr(self.lib, self._PREFIX + '_adam_state')(self.h, C.byref(pm), C.byref(pv), C.byref(t)))
        n = self.n_params
        m = m.to(device=self.device, dtype=torch.float32).contiguous()
        v = v.to(device=self.device, dtype=torch.float32).contiguous()
        assert m.numel() == n and v.numel() == n
        check(self.lib.rl4rs_copy_d2d(pm, _ptr(m), n * 4, _stream()))
        check(self.lib.rl4rs_copy_d2d(pv, _ptr(v), n * 4, _stream()))
        check(getattr(self.lib, self._PREFIX + '_set_adam_step')(self.h, int(step)))

    def copy_from(self, other):
        check(self.lib.rl4rs_qnet_copy_params(self.h, other.h, _stream()))

    def check_status(self):
        flags = C.c_int32(0)
        check(self.lib.rl4rs_qnet_status(self.h, C.byref(flags), _stream()))
        if flags.value & 1:
            raise IndexError('offline-RL batch: an item id in an observation tail or an action is outside [0, action_size)')
        if flags.value & 2:
            raise IndexError('offline-RL batch: cur_step %% 9 // 3 selects a location_mask row that does not exist')

    def _obs(self, obs):
        obs = _dev_tensor(obs, torch.float32, self.device)
        if obs.dim() != 2 or obs.shape[1] != self.D or not 0 < obs.shape[0] <= self.max_rows:
            raise ValueError('obs must be [N <= %d, %d] (got %r)' % (self.max_rows, self.D, tuple(obs.shape)))
        return obs

    def forward(self, obs):
        """out [N, action_size]: Q values / imitator logits.  Keeps the activations for ``backward``."""
        obs = self._obs(obs)
        out = torch.empty((obs.shape[0], self.A), dtype=torch.float32, device=self.device)
        check(self.lib.rl4rs_qnet_forward(self.h, obs.shape[0], _ptr(obs), _ptr(out), _stream()))
        return out

    def backward(self, obs, dout):
        """Gradient of sum(out * dout) into the handle (after ``forward`` of the same rows)."""
        obs = self._obs(obs)
        dout = _dev_tensor(dout, torch.float32, self.device)
        assert tuple(dout.shape) == (obs.shape[0], self.A)
        check(self.lib.rl4rs_qnet_backward(self.h, obs.shape[0], _ptr(obs), _ptr(dout), _stream()))

    def adam_step(self, lr, beta1=0.9, beta2=0.999, eps=1e-8):
        check(self.lib.rl4rs_qnet_adam_step(self.h, lr, beta1, beta2, eps, _stream()))

    def imitation_loss(self, logits, actions, beta):
        """(loss2 = [mean nll, mean_n sum_k logits^2], dlogits) of ``nll + beta * mean(logits^2)``."""
        N = logits.shape[0]
        actions = _dev_tensor(actions, torch.int32, self.device)
        d = torch.empty_like(logits)
        rows = torch.empty((N, 2), dtype=torch.float32, device=self.device)
        loss2 = torch.empty(2, dtype=torch.float32, device=self.device)
        check(self.lib.rl4rs_qloss_imitation(self.h, N, _ptr(logits), _ptr(actions), beta, _ptr(d), _ptr(rows), _ptr(loss2), _stream()))
        return loss2, d

    def dqn_loss(self, q_t, actions, rewards, terminals, q_next, q_next_target, imitator_next=None, action_flexibility=0.3,
                 gamma=0.99, cql_alpha=0.0):
        """(loss2 = [mean huber TD, mean conservative], dq, best next action)."""
        N = q_t.shape[0]
        actions = _dev_tensor(actions, torch.int32, self.device)
        rewards = _dev_tensor(rewards, torch.float32, self.device)
        terminals = _dev_tensor(terminals, torch.float32, self.device)
        dq = torch.empty_like(q_t)
        rows = torch.empty((N, 2), dtype=torch.float32, device=self.device)
        loss2 = torch.empty(2, dtype=torch.float32, device=self.device)
        best = torch.empty(N, dtype=torch.int32, device=self.device)
        check(self.lib.rl4rs_qloss_dqn(self.h, N, _ptr(q_t), _ptr(actions), _ptr(rewards), _ptr(terminals), _ptr(q_next),
                                       _ptr(q_next_target), _ptr(imitator_next), action_flexibility, gamma, cql_alpha, _ptr(dq),
                                       _ptr(rows), _ptr(loss2), _ptr(best), _stream()))
        return loss2, dq, best

    def best_action(self, q, imitator_logits=None, action_flexibility=0.3):
        out = torch.empty(q.shape[0], dtype=torch.int32, device=self.device)
        check(self.lib.rl4rs_q_best_action(q.shape[0], self.A, _ptr(q), _ptr(imitator_logits), action_flexibility, _ptr(out), _stream()))
        return out


class DeviceAMLP(object):
    """rl4rs_amlp handle: d3rlpy's default continuous-control network - ``VectorEncoderWithAction([256, 256], relu)`` on
    ``cat([x, action])`` (``act_dim = 0``: plain ``VectorEncoder``) + one Linear head - with forward, backward (parameter and
    action-input gradients) and torch-style Adam on the device.  ``params``: dict of float32 arrays stored [in, out]:
    fc1_w [obs_dim + act_dim, hidden1] (observation rows first), fc1_b, fc2_w, fc2_b, head_w, head_b.
    ``head_act``: 'none' or 'tanh'.  The building block of the continuous BCQ / CQL learners (``offline_rl.BCQ`` / ``CQL``)."""

    HEAD_ACTS = {'none': 0, 'elu': 1, 'sigmoid': 2, 'tanh': 3, 'relu': 4}

    def __init__(self, obs_dim, act_dim, out_dim, params, hidden1=256, hidden2=256, head_act='none', max_rows=256, max_grad_rows=None,
                 device=None):
        _lib.require_device()
        self.lib = _lib.load()
        self.device = torch.device('cuda', torch.cuda.current_device()) if device is None else torch.device(device)
        self.D, self.E, self.K, self.H1, self.H2 = int(obs_dim), int(act_dim), int(out_dim), int(hidden1), int(hidden2)
        self.max_rows = int(max_rows)
        self.max_grad_rows = self.max_rows if max_grad_rows is None else int(max_grad_rows)
        self.shapes = [('fc1_w', (self.D + self.E, self.H1)), ('fc1_b', (self.H1,)), ('fc2_w', (self.H1, self.H2)), ('fc2_b', (self.H2,)),
                       ('head_w', (self.H2, self.K)), ('head_b', (self.K,))]
        flat = []
        for name, shape in self.shapes:
            arr = np.ascontiguousarray(params[name], dtype=np.float32)
            if tuple(arr.shape) != tuple(shape):
                raise ValueError('amlp parameter %r has shape %r, expected %r' % (name, tuple(arr.shape), tuple(shape)))
            flat.append(arr.reshape(-1))
        flat = np.ascontiguousarray(np.concatenate(flat))
        cfg = _lib.AmlpCfg(self.D, self.E, self.H1, self.H2, self.K, self.HEAD_ACTS[head_act], self.max_rows, self.max_grad_rows)
        h = C.c_void_p()
        with torch.cuda.device(self.device):
            check(self.lib.rl4rs_amlp_create(C.byref(cfg), flat.ctypes.data_as(_lib._FP), _stream(), C.byref(h)))
        self.h = h
        self.n_params = int(flat.size)
        self.h16_ok = bool(self.lib.rl4rs_amlp_h16_ok(self.h))

    def close(self):
        if getattr(self, 'h', None) is not None and self.h:
            self.lib.rl4rs_amlp_destroy(self.h)
            self.h = None

    __del__ = close

    def _buffers(self):
        p, g, n = C.c_void_p(), C.c_void_p(), C.c_int64()
        check(self.lib.rl4rs_amlp_params(self.h, C.byref(p), C.byref(g), C.byref(n)))
        return p, g, n.value

    def _flat(self, which):
        p, g, n = self._buffers()
        out = torch.empty(n, dtype=torch.float32, device=self.device)
        check(self.lib.rl4rs_copy_d2d(_ptr(out), p if which == 'params' else g, n * 4, _stream()))
        return out

    def _split(self, flat):
        out, o = {}, 0
        for name, shape in self.shapes:
            k = int(np.prod(shape))
            out[name] = flat[o:o + k].reshape(shape)
            o += k
        return out

    def weights(self):
        return self._split(self._flat('params'))

    def gradients(self):
        return self._split(self._flat('grad'))

    def flat_gradient(self):
        return self._flat('grad')

    def flat_params(self):
        return self._flat('params')

    def set_flat_gradient(self, flat):
        _, g, n = self._buffers()
        assert flat.numel() == n and flat.dtype == torch.float32 and flat.is_contiguous()
        check(self.lib.rl4rs_copy_d2d(g, _ptr(flat), n * 4, _stream()))

    def set_flat_params(self, flat):
        p, _, n = self._buffers()
        assert flat.numel() == n and flat.dtype == torch.float32 and flat.is_contiguous()
        check(self.lib.rl4rs_copy_d2d(p, _ptr(flat), n * 4, _stream()))

    _PREFIX = 'rl4rs_amlp'

    def adam_state(self):
        """(m, v, step): copies of the Adam moments and the step count (checkpointing)."""
        pm, pv, t = C.c_void_p(), C.c_void_p(), C.c_int64()
        check(getattr(self.lib, self._PREFIX + '_adam_state')(self.h, C.byref(pm), C.byref(pv), C.byref(t)))
        n = self.n_params
        m = torch.empty(n, dtype=torch.float32, device=self.device)
        v = torch.empty(n, dtype=torch.float32, device=self.device)
        check(self.lib.rl4rs_copy_d2d(_ptr(m), pm, n * 4, _stream()))
        check(self.lib.rl4rs_copy_d2d(_ptr(v), pv, n * 4, _stream()))
        return m, v, int(t.value)

    def set_adam_state(self, m, v, step):
        pm, pv, t = C.c_void_p(), C.c_void_p(), C.c_int64()
        check(getattr(self.lib, self._PREFIX + '_adam_state')(self.h, C.byref(pm), C.byref(pv), C.byref(t)))
        n = self.n_params
        m = m.to(device=self.device, dtype=torch.float32).contiguous()
        v = v.to(device=self.device, dtype=torch.float32).contiguous()
        assert m.numel() == n and v.numel() == n
        check(self.lib.rl4rs_copy_d2d(pm, _ptr(m), n * 4, _stream()))
        check(self.lib.rl4rs_copy_d2d(pv, _ptr(v), n * 4, _stream()))
        check(getattr(self.lib, self._PREFIX + '_set_adam_step')(self.h, int(step)))

    def copy_from(self, other):
        check(self.lib.rl4rs_amlp_copy_params(self.h, other.h, _stream()))

    def soft_update_from(self, other, tau):
        check(self.lib.rl4rs_amlp_soft_update(self.h, other.h, tau, _stream()))

    def check_status(self):
        pass

    def _rows(self, obs, act, rep):
        assert obs.is_cuda and obs.dtype == torch.float32 and obs.is_contiguous() and obs.shape[1] == self.D
        n = obs.shape[0] * rep
        if self.E:
            assert act.is_cuda and act.dtype == torch.float32 and act.is_contiguous() and tuple(act.shape) == (n, self.E), \
                (tuple(act.shape), n, self.E)
        return n

    H16_MIN_ROWS = 4096      # below this the fused fp16x2 forward has nothing to win over the small fp32 forms

    def forward(self, obs, act=None, rep=1, out=None, nograd=False):
        """out [N, out_dim] for obs [N / rep, obs_dim] (each observation shared by ``rep`` consecutive action rows) and act
        [N, act_dim].  Keeps the activations for ``backward``.  ``nograd='fp16x2'``: the rows will never see a backward and may be
        computed in the scorer's fp16x2 arithmetic - one fused launch for the three layers (rl4rs_amlp_forward_h16) when the
        network's shape has that form and there are at least ``H16_MIN_ROWS`` rows; anything else is the fp32 forward."""
        n = self._rows(obs, act, rep)
        if out is None:
            out = torch.empty((n, self.K), dtype=torch.float32, device=self.device)
        # (the fused kernel moves 16-byte vectors: an action view at an odd row offset takes the fp32 path, as documented)
        aligned = obs.data_ptr() % 16 == 0 and out.data_ptr() % 16 == 0 and (act is None or act.data_ptr() % 16 == 0)
        if nograd == 'fp16x2' and self.h16_ok and n >= self.H16_MIN_ROWS and aligned:
            check(self.lib.rl4rs_amlp_forward_h16(self.h, n, rep, _ptr(obs), _ptr(act), _ptr(out), _stream()))
        else:
            check(self.lib.rl4rs_amlp_forward(self.h, n, rep, _ptr(obs), _ptr(act), _ptr(out), _stream()))
        return out

    def backward(self, obs, act, dout, rep=1, want_dact=False, want_param_grad=True):
        """Given the gradient wrt the head's PRE-activation output: parameter gradients into the handle, returns the gradient
        wrt ``act`` when ``want_dact``."""
        n = self._rows(obs, act, rep)
        assert dout.is_cuda and dout.dtype == torch.float32 and dout.is_contiguous() and dout.numel() == n * self.K
        dact = torch.empty((n, self.E), dtype=torch.float32, device=self.device) if want_dact else None
        check(self.lib.rl4rs_amlp_backward(self.h, n, rep, _ptr(obs), _ptr(act), _ptr(dout), _ptr(dact), 1 if want_param_grad else 0,
                                           _stream()))
        return dact

    def adam_step(self, lr, beta1=0.9, beta2=0.999, eps=1e-8):
        check(self.lib.rl4rs_amlp_adam_step(self.h, lr, beta1, beta2, eps, _stream()))


def amlp_adam_multi(nets, lrs, targets=None, tau=0.0, step=None, beta1=0.9, beta2=0.999, eps=1e-8):
    """The optimiser of one phase of an update as ONE launch (rl4rs_amlp_adam_multi): torch Adam with learning rate ``lrs[i]`` for
    every ``nets[i]`` (``step[i]`` False: no step, soft update only) and, where ``targets[i]`` is given, the soft target update
    target = (1 - tau) target + tau net from the stepped parameters."""
    lib = _lib.load()
    n = len(nets)
    targets = list(targets) if targets is not None else [None] * n
    step = list(step) if step is not None else [True] * n
    H = (C.c_void_p * n)(*[net.h for net in nets])
    T = (C.c_void_p * n)(*[(t.h if t is not None else None) for t in targets])
    LR = (C.c_float * n)(*[float(x) for x in lrs])
    DO = (C.c_int32 * n)(*[1 if x else 0 for x in step])
    check(lib.rl4rs_amlp_adam_multi(n, H, LR, DO, T, beta1, beta2, eps, float(tau), _stream()))


def amlp_forward_multi(nets, obs, act=None):
    """``net.forward(obs, act)`` for several networks with the same input widths on the SAME rows - the twin critics - as ONE launch
    when the call has the fused form (rl4rs_amlp_forward_multi).  Returns the list of outputs; every network keeps its activations
    for a following backward."""
    lib = _lib.load()
    n0 = nets[0]
    N = n0._rows(obs, act, 1)
    outs = [torch.empty((N, net.K), dtype=torch.float32, device=net.device) for net in nets]
    H = (C.c_void_p * len(nets))(*[net.h for net in nets])
    O = (C.c_void_p * len(nets))(*[_ptr(o) for o in outs])
    check(lib.rl4rs_amlp_forward_multi(len(nets), H, N, _ptr(obs), _ptr(act), O, _stream()))
    return outs


def amlp_backward_multi(nets, obs, act, douts, want_dact=False, want_param_grad=True):
    """``net.backward(obs, act, dout)`` for the networks of ``amlp_forward_multi`` as one input-gradient launch + one parameter-
    gradient launch (rl4rs_amlp_backward_multi).  Returns the list of action-input gradients (or None)."""
    lib = _lib.load()
    n0 = nets[0]
    N = n0._rows(obs, act, 1)
    for net, d in zip(nets, douts):
        assert d.is_cuda and d.dtype == torch.float32 and d.is_contiguous() and d.numel() == N * net.K
    dacts = [torch.empty((N, net.E), dtype=torch.float32, device=net.device) for net in nets] if want_dact else None
    H = (C.c_void_p * len(nets))(*[net.h for net in nets])
    DO = (C.c_void_p * len(nets))(*[_ptr(d) for d in douts])
    DA = (C.c_void_p * len(nets))(*[_ptr(d) for d in dacts]) if want_dact else None
    check(lib.rl4rs_amlp_backward_multi(len(nets), H, N, _ptr(obs), _ptr(act), DO, DA, 1 if want_param_grad else 0, _stream()))
    return dacts


def amlp_set_fused(on):
    """Fused minibatch forward / backward of the amlp networks on (default), off, or 2 = their 8-rows-per-workgroup form
    (rl4rs_amlp_set_fused; tests and A/B runs)."""
    check(_lib.load().rl4rs_amlp_set_fused(2 if on == 2 else (1 if on else 0)))


def cvae_sample(enc_out, eps, min_logstd=-20.0, max_logstd=2.0):
    """z = mu + exp(clamp(logstd)) * eps for enc_out [N, 2L] = [mu | logstd]."""
    lib = _lib.load()
    N, L = eps.shape
    z = torch.empty_like(eps)
    check(lib.rl4rs_cvae_sample(N, L, _ptr(enc_out), _ptr(eps), min_logstd, max_logstd, _ptr(z), _stream()))
    return z


def cvae_loss(decoded, actions, enc_out, min_logstd=-20.0, max_logstd=2.0):
    """(loss2 = [mean_n sum_e (y - a)^2, mean_n sum_l KL], gradient of the mse term wrt the decoder's pre-tanh output)."""
    lib = _lib.load()
    N, E = decoded.shape
    L = enc_out.shape[1] // 2
    d = torch.empty_like(decoded)
    rows = torch.empty((N, 2), dtype=torch.float32, device=decoded.device)
    loss2 = torch.empty(2, dtype=torch.float32, device=decoded.device)
    check(lib.rl4rs_cvae_loss(N, E, L, _ptr(decoded), _ptr(actions), _ptr(enc_out), min_logstd, max_logstd, _ptr(d), _ptr(rows), _ptr(loss2),
                              _stream()))
    return loss2, d


def cvae_encoder_grad(enc_out, eps, dz, beta, min_logstd=-20.0, max_logstd=2.0):
    lib = _lib.load()
    N, L = eps.shape
    d = torch.empty_like(enc_out)
    check(lib.rl4rs_cvae_encoder_grad(N, L, _ptr(enc_out), _ptr(eps), _ptr(dz), beta, min_logstd, max_logstd, _ptr(d), _stream()))
    return d


def residual_action(action, tanh_out, scale):
    lib = _lib.load()
    N, E = action.shape
    out = torch.empty_like(action)
    check(lib.rl4rs_residual_action(N, E, _ptr(action), _ptr(tanh_out), scale, _ptr(out), _stream()))
    return out


def residual_grad(action, tanh_out, scale, d_out):
    lib = _lib.load()
    N, E = action.shape
    d = torch.empty_like(action)
    check(lib.rl4rs_residual_grad(N, E, _ptr(action), _ptr(tanh_out), scale, _ptr(d_out), _ptr(d), _stream()))
    return d


def bcq_target(q1, q2, n, lam, rewards=None, terminals=None, gamma=0.99, want_best=False):
    """(y [B], best [B] or None): max over the n sampled actions of the lam-weighted twin value (q2 None: of q1)."""
    lib = _lib.load()
    B = q1.numel() // n
    y = torch.empty(B, dtype=torch.float32, device=q1.device)
    best = torch.empty(B, dtype=torch.int32, device=q1.device) if want_best else None
    check(lib.rl4rs_bcq_target(B, n, _ptr(q1), _ptr(q2), lam, _ptr(rewards), _ptr(terminals), gamma, _ptr(y), _ptr(best), _stream()))
    return y, best


def pick_rows(rows, best, n):
    lib = _lib.load()
    B, E = best.numel(), rows.shape[1]
    out = torch.empty((B, E), dtype=torch.float32, device=rows.device)
    check(lib.rl4rs_pick_rows(B, n, E, _ptr(rows), _ptr(best), _ptr(out), _stream()))
    return out


def critic_mse(q1, q2, y):
    """(loss2 = [mean (q1 - y)^2, mean (q2 - y)^2], dq1, dq2)."""
    lib = _lib.load()
    N = y.numel()
    dq1, dq2 = torch.empty_like(q1), torch.empty_like(q2)
    loss2 = torch.empty(2, dtype=torch.float32, device=y.device)
    check(lib.rl4rs_critic_mse(N, _ptr(q1), _ptr(q2), _ptr(y), _ptr(dq1), _ptr(dq2), _ptr(loss2), _stream()))
    return loss2, dq1, dq2


def squashed_sample(head, eps, rep=1, act_out=None, logp_out=None, out_rep=None, out_off=0, min_logstd=-20.0, max_logstd=2.0):
    """d3rlpy SquashedNormalPolicy sampling from head [R, 2A] = [mu | logstd]: (actions, log-probs).  ``eps`` [R * rep, A] Gaussian
    noise, or None for the deterministic best_action tanh(mu).  ``act_out`` [rows, A] / ``logp_out`` [rows] with ``out_rep`` /
    ``out_off``: write sample i of observation r to row r * out_rep + out_off + i % rep of caller-owned buffers."""
    lib = _lib.load()
    R, A = head.shape[0], head.shape[1] // 2
    N = R * rep
    if out_rep is None:
        out_rep = rep
    if act_out is None:
        act_out = torch.empty((R * out_rep, A), dtype=torch.float32, device=head.device)
    if logp_out is None and eps is not None:
        logp_out = torch.empty(R * out_rep, dtype=torch.float32, device=head.device)
    if eps is not None:
        assert eps.is_contiguous() and eps.dtype == torch.float32 and eps.numel() == N * A
    check(lib.rl4rs_squashed_sample(N, rep, A, _ptr(head), _ptr(eps), min_logstd, max_logstd, out_rep, out_off, _ptr(act_out), _ptr(logp_out),
                                    _stream()))
    return act_out, logp_out


def sac_actor_grad(head, eps, act, g_act, log_temp, min_logstd=-20.0, max_logstd=2.0):
    lib = _lib.load()
    B, A = act.shape
    d = torch.empty_like(head)
    check(lib.rl4rs_sac_actor_grad(B, A, _ptr(head), _ptr(eps), _ptr(act), _ptr(g_act), _ptr(log_temp), min_logstd, max_logstd, _ptr(d), _stream()))
    return d


def twin_min(q1, q2, want_grad=False):
    """(min(q1, q2), dq1, dq2): the selector of -mean min (None without ``want_grad``)."""
    lib = _lib.load()
    B = q1.numel()
    qmin = torch.empty(B, dtype=torch.float32, device=q1.device)
    dq1 = torch.empty_like(q1) if want_grad else None
    dq2 = torch.empty_like(q2) if want_grad else None
    check(lib.rl4rs_twin_min(B, _ptr(q1), _ptr(q2), _ptr(qmin), _ptr(dq1), _ptr(dq2), _stream()))
    return qmin, dq1, dq2


def cql_critic_loss(q1, q2, offs, m, y=None, alpha_w=None):
    """(sums6, dq1, dq2) of the continuous CQL critic loss over rows [B][m] (column 0 = dataset action): see
    rl4rs_cql_critic_loss.  ``y`` None: sums only."""
    lib = _lib.load()
    B = q1.numel() // m
    dq1 = torch.empty_like(q1) if y is not None else None
    dq2 = torch.empty_like(q2) if y is not None else None
    rows = torch.empty((B, 6), dtype=torch.float32, device=q1.device)
    sums = torch.empty(6, dtype=torch.float32, device=q1.device)
    check(lib.rl4rs_cql_critic_loss(B, m, _ptr(q1), _ptr(q2), _ptr(offs), _ptr(y), _ptr(alpha_w), _ptr(dq1), _ptr(dq2), _ptr(rows), _ptr(sums),
                                    _stream()))
    return sums, dq1, dq2
