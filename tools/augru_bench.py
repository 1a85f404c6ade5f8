#!/usr/bin/env python
"""Micro-benchmark + numerics check of the AUGRU recurrence kernels on the bench shapes (B = 4096 envs, 2 sequence inputs):
an obs-sized forward (R = B, one row per env: 8192 row-inputs = 256 row tiles) and a reward-sized one (R = 8 B, 8 rows per env).
Prints per kernel generation the HIP-event time of the AUGRU launch and the max |obs| difference against the exact-fp32
recurrence (k_recur<256,augru>) on the same weights and inputs.   usage: augru_bench.py [x h16 ...] [--reps N]"""
import os
import sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rl4rs_amd.nets.dien import init_dien_weights
from rl4rs_amd.device import DeviceDien

B = int(os.environ.get('AUGRU_BENCH_B', '4096'))
CFG = {"maxlen": 64, "batch_size": B, "action_size": 284, "class_num": 2, "dense_feature_num": 432,
       "category_feature_num": 21, "category_hash_size": 3000, "seq_num": 2, "emb_size": 128,
       "page_items": 9, "hidden_units": 128, "max_steps": 9, "action_emb_size": 32}
args = [a for a in sys.argv[1:] if not a.startswith('--')]
reps = int(sys.argv[sys.argv.index('--reps') + 1]) if '--reps' in sys.argv else 5
kinds = args or ['x', 'h16']
w = init_dien_weights(CFG, seed=3)
rs = np.random.RandomState(0)
seq = rs.randint(0, 284, size=(B, 2, 64)).astype(np.int32)
dense = {1: torch.from_numpy(np.abs(rs.randn(B, 432)).astype(np.float32)).cuda(),
         8: torch.from_numpy(np.abs(rs.randn(8 * B, 432)).astype(np.float32)).cuda()}
cat = {1: torch.from_numpy(rs.randint(0, 284, size=(B, 21)).astype(np.int32)).cuda(),
       8: torch.from_numpy(rs.randint(0, 284, size=(8 * B, 21)).astype(np.int32)).cuda()}
slots = torch.arange(B, dtype=torch.int32).repeat(2, 1).contiguous().cuda()
if os.environ.get('AUGRU_BENCH_SLOTS'):          # all rows share a few cache slots: the projection rows stay in L2
    slots = (slots % int(os.environ['AUGRU_BENCH_SLOTS'])).contiguous()


def run(kind):
    if kind == 'fp32':
        cfg = dict(CFG, scorer_precision='fp32')
    else:
        # kind: 'x' (k_augru_x, default), 'h16' (first generation), 'x32' / 'x64' (k_augru_x pinned to one row-tile form)
        opts = {'x': '', 'h16': 'augru_h16', 'x32': 'augru_rows32', 'x64': 'augru_rows64'}[kind]
        cfg = dict(CFG, scorer_precision='fp16x2', scorer_kernels=opts)
    net = DeviceDien(cfg, w, max_rows=8 * B, max_slots=B)
    for s in range(2):
        net.encode(s, torch.from_numpy(np.ascontiguousarray(seq[:, s])).cuda(), 0)
    out = {}
    for group in (1, 8):
        R = B * group
        obs, _ = net.forward(R, group, dense[group], cat[group], slots, want_obs=True, want_prob=False)
        out[group] = obs.clone()
        net.set_profiling(True)
        net.profile_reset()
        for _ in range(reps):
            net.forward(R, group, dense[group], cat[group], slots, want_obs=True, want_prob=False)
        torch.cuda.synchronize()
        ms, n = net.profile()[net.augru_kernel]
        net.set_profiling(False)
        out['ms%d' % group] = ms / max(n, 1)
    if not os.environ.get('AUGRU_BENCH_NOCHECK'):
        net.check_status()
    else:
        net.set_profiling(False)
    net.close()
    return out


ref = run('fp32')
print('%-6s obs-sized %8.3f ms   reward-sized %8.3f ms' % ('fp32', ref['ms1'], ref['ms8']))
flop_row = 64 * 256 * 768 * 2
for k in kinds:
    o = run(k)
    d1 = (o[1] - ref[1]).abs().max().item()
    d8 = (o[8] - ref[8]).abs().max().item()
    tf1 = B * 2 * flop_row / (o['ms1'] * 1e-3) / 1e12
    tf8 = 8 * B * 2 * flop_row / (o['ms8'] * 1e-3) / 1e12
    print('%-6s obs-sized %8.3f ms (%5.1f TF/s, %.3f of 833)   reward-sized %8.3f ms (%5.1f TF/s, %.3f of 833)   max|obs - fp32| %.2e / %.2e   nan %d'
          % (k, o['ms1'], tf1, tf1 / 833.3, o['ms8'], tf8, tf8 / 833.3, d1, d8, int(torch.isnan(o[1]).sum() + torch.isnan(o[8]).sum())))
