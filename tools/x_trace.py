#!/usr/bin/env python
"""Timing experiment (library built with -DRL4RS_X_TRACE): s_memtime marks of workgroup (0,0) of k_augru_x, steps 8..11, per wave.
marks: 0 step start | 1 R-late done | 2 U done (before barrier 1) | 3 after barrier 1 | 4 C done (before barrier a) |
5 after barrier a | 6 R-early(next) done (before barrier b) | 7 after barrier b.   usage: x_trace.py [R] [group]"""
import os
import sys
import numpy as np
os.environ['RL4RS_H16_TRACE_DUMP'] = '/tmp/x_trace.bin'
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rl4rs_amd.nets.dien import init_dien_weights
from rl4rs_amd.device import DeviceDien

B = 4096
CFG = {"maxlen": 64, "batch_size": B, "action_size": 284, "class_num": 2, "dense_feature_num": 432,
       "category_feature_num": 21, "category_hash_size": 3000, "seq_num": 2, "emb_size": 128,
       "page_items": 9, "hidden_units": 128, "max_steps": 9, "action_emb_size": 32, "scorer_precision": "fp16x2"}
group = int(sys.argv[1]) if len(sys.argv) > 1 else 1
R = B * group
w = init_dien_weights(CFG, seed=3)
rs = np.random.RandomState(0)
net = DeviceDien(CFG, w, max_rows=R, max_slots=B)
seq = rs.randint(0, 284, size=(B, 2, 64)).astype(np.int32)
for s in range(2):
    net.encode(s, torch.from_numpy(np.ascontiguousarray(seq[:, s])).cuda(), 0)
slots = torch.arange(B, dtype=torch.int32).repeat(2, 1).contiguous().cuda()
dense = torch.from_numpy(np.abs(rs.randn(R, 432)).astype(np.float32)).cuda()
cat = torch.from_numpy(rs.randint(0, 284, size=(R, 21)).astype(np.int32)).cuda()
for _ in range(3):      # the dump at launch k holds the marks of launch k-1
    net.forward(R, group, dense, cat, slots, want_obs=True, want_prob=False)
torch.cuda.synchronize()
tr = np.fromfile('/tmp/x_trace.bin', dtype=np.uint64).reshape(8, 4, 8).astype(np.int64)
names = ['R-late', 'U', 'bar1', 'C(+ep)', 'bar_a', 'R-early', 'bar_b']
for wv in range(8):
    for st in range(3):
        m = tr[wv, st]
        seg = [m[k + 1] - m[k] for k in range(7)]
        print('wave %d step %d: ' % (wv, st + 8) + '  '.join('%s %5d' % (n, v) for n, v in zip(names, seg)) +
              '   total %d' % (tr[wv, st + 1, 0] - m[0]))
