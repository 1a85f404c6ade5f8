#!/usr/bin/env python
"""Timing experiment (library built with tools/build_ab.py dinxtrace -DRL4RS_DINX_TRACE; RL4RS_LIB=tools/_ab/dinxtrace/librl4rs_hip.so):
s_memtime marks of workgroup (40, 0) of k_din_x on an obs-sized launch (B = 4096), per wave and 32-step tile.
marks: 0 tile start | 1 requests out, accumulators = AK + qa (the AK rows have arrived) | 2 first k-block's operand built |
3 k-block 4 | 4 k loop done | 5 epilogue done.   usage: dinx_trace.py [group]"""
import os
import sys
import numpy as np
os.environ['RL4RS_DINX_TRACE_DUMP'] = '/tmp/dinx_trace.bin'
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
# DISTINCT=1771 (default, the bench's share of distinct histories: duplicates adjacent, as the row-order hint arranges them) | DISTINCT=4096
distinct = int(os.environ.get('DISTINCT', '1771'))
slots = (torch.arange(B, dtype=torch.int64) * distinct // B).to(torch.int32).repeat(2, 1).contiguous().cuda()
dense = torch.from_numpy(np.abs(rs.randn(R, 432)).astype(np.float32)).cuda()
cat = torch.from_numpy(rs.randint(0, 284, size=(R, 21)).astype(np.int32)).cuda()
for _ in range(3):      # the dump at launch k holds the marks of launch k-1
    net.forward(R, group, dense, cat, slots, want_obs=True, want_prob=False)
torch.cuda.synchronize()
tr = np.fromfile('/tmp/dinx_trace.bin', dtype=np.uint64).reshape(8, 5, 8).astype(np.int64)
t0 = tr[:, 4, 0].min()
names = ['requests+AK', 'operand 0', 'kb 0-3', 'kb 4-7', 'epilogue']
for wv in range(8):
    print('wave %d: start %6d  staged %6d  end %6d' % (wv, tr[wv, 4, 0] - t0, tr[wv, 4, 1] - t0, tr[wv, 4, 2] - t0))
    for tile in range(4):
        m = tr[wv, tile]
        if m[5] == 0:
            continue
        print('   tile %d @%6d: ' % (tile, m[0] - t0) + '  '.join('%s %5d' % (n, m[k + 1] - m[k]) for k, n in enumerate(names)) + '   total %d' % (m[5] - m[0]))
