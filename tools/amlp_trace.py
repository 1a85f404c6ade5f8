#!/usr/bin/env python
"""s_memtime marks of k_amlp_fwd_h16 (library built with -DRL4RS_AMLP_TRACE=1: tools/build_ab.py trace -DRL4RS_AMLP_TRACE=1), two
workgroups (first / middle of the grid) x their waves.  Segments: top loads + staging | barrier | layer 1 | epilogue 1 | barrier |
layer 2 | barrier | epilogue 2 | barrier | head | reduce + store.
usage: RL4RS_LIB=tools/_ab/trace/librl4rs_hip.so python tools/amlp_trace.py [rows]"""
import os
import sys
import numpy as np
os.environ['RL4RS_AMLP_TRACE_DUMP'] = '/tmp/amlp_trace.bin'
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rl4rs_amd import device as D_
from rl4rs_amd.offline_rl import init_amlp_params

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 409600
D, E, K = 266, 32, 32
net = D_.DeviceAMLP(D, E, K, init_amlp_params(D, E, K, seed=1), head_act='tanh', max_rows=rows, max_grad_rows=0)
obs = torch.randn(rows // 100, D, device='cuda')
act = torch.rand(rows, E, device='cuda')
for _ in range(3):
    net.forward(obs, act, rep=100, nograd='fp16x2')
torch.cuda.synchronize()
tr = np.fromfile('/tmp/amlp_trace.bin', dtype=np.uint64).reshape(2, 8, 16).astype(np.int64)
names = ['load+stage', 'bar', 'L1', 'epi1', 'bar', 'L2', 'bar', 'epi2', 'bar', 'head', 'red+store']
for g in range(2):
    for w in range(8):
        m = tr[g, w]
        print('wg %s wave %d: ' % ('first' if g == 0 else 'mid  ', w) + '  '.join('%s %5d' % (n, m[k + 1] - m[k]) for k, n in enumerate(names)) +
              '   total %d' % (m[11] - m[0]))
