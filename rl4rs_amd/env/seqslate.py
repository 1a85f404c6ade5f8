"""SeqSlateState / SeqSlateRecEnv (multi-page variant) on the GPU.

Mirrors ``rl4rs/env/seqslate.py``: paging of ``prev_actions`` (seqslate.py:103-122), per-page mask reset
(:124-126), page-level reward (:136-160), page-0-only special check in ``get_violation`` (:63-68) and
the literal ``cur_step % 9`` in ``offline_reward`` (:74) are all reproduced by the device kernels
(rl4rs_amd/csrc/env.hip, ``is_seq`` branches).
"""
import numpy as np

from .slate import SlateState, SlateRecEnv
from .. import device as D


class SeqSlateState(SlateState):
    is_seq = True

    def __init__(self, config, records, _ctx=None):
        SlateState.__init__(self, config, records, _ctx=_ctx)
        self.page_items = config.get("page_items", 9)

    def _violation_zeroes_reward(self):
        # seqslate.py:154-157: only the mask modes zero the reward on violation
        return bool(self.config.get("support_rllib_mask", False) or self.config.get("support_d3rl_mask", False))

    def _masked_actions(self):
        """seqslate.py:18-23: the current page's columns of prev_actions."""
        import torch
        env = self._live()
        cur_steps = env.cur_steps
        P = self.page_items
        page_init = cur_steps // P * P
        page_end = min(page_init + P - 1, self.max_steps - 1)
        pa = env.snapshot(D.BUF_PREV_ACTIONS).to(torch.int64)[:, page_end + 1 - P:page_end + 1]
        cur = torch.full((self.batch_size, 1), cur_steps, dtype=torch.int64, device=pa.device)
        return pa, cur

    @property
    def offline_reward(self):
        env = self._live()
        if env.cur_steps % 9 != 0:               # literal 9 (seqslate.py:74)
            return [0, ] * self.batch_size
        r = env.offline_reward()
        return r if self._tensor_mode() else r.cpu().numpy()

    def act(self, actions):
        env = self._live()
        if env.cur_steps > 0 and env.cur_steps % self.page_items == 0:
            # the second sequence input (items of the previous pages, seqslate.py:107-108) changes on the first act of a page - of
            # every page but the first: `prev_actions[:0]` is the same [0] the episode started with
            self._seq1_version += 1
        SlateState.act(self, actions)


class SeqSlateRecEnv(SlateRecEnv):
    def __init__(self, config, state_cls):
        SlateRecEnv.__init__(self, config, state_cls)
        self.page_items = config.get("page_items", 9)

    def _reward_due(self, samples):
        return samples.cur_steps % self.page_items == 0      # seqslate.py:138
