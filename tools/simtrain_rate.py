#!/usr/bin/env python
"""Time of one device-side simulator training step (batch 256, the reference's supervised_train.py batch) per family."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rl4rs_amd.device import DeviceSimTrainer, DeviceDienTrainer
from rl4rs_amd.nets.simnets import init_simnet_weights
from rl4rs_amd.nets.dien import init_dien_weights

CFG = {"maxlen": 64, "class_num": 2, "dense_feature_num": 432, "category_feature_num": 21, "category_hash_size": 100000,
       "seq_num": 2, "emb_size": 128, "hidden_units": 128}
N = 256
rs = np.random.RandomState(0)
t = lambda a: torch.from_numpy(a).cuda()
dense = t(np.abs(rs.randn(N, 432)).astype(np.float32))
cat = t(rs.randint(0, 284, size=(N, 21)).astype(np.int32))
labels = t(rs.randint(0, 2, size=N).astype(np.int32))
seqs = [t(rs.randint(0, 284, size=(N, 64)).astype(np.int32)) for _ in range(2)]
# RECUR_ROWS=8|32 pins the row-tile form of the persistent recurrences (default: automatic); ALGOS=lstm,dien restricts the families
from rl4rs_amd import _lib
_lib.check(_lib.load().rl4rs_recur_train_set_rows(int(os.environ.get('RECUR_ROWS', '0'))))
_lib.check(_lib.load().rl4rs_dientrain_set_fork(int(os.environ.get('DIEN_FORK', '1'))))        # DIEN_FORK=0: per-input chains on one stream
for algo in os.environ.get('ALGOS', 'dnn,widedeep,lstm,dien').split(','):
    if algo == 'dien':
        tr = DeviceDienTrainer(CFG, init_dien_weights(CFG, seed=1), max_batch=N)
    else:
        tr = DeviceSimTrainer(CFG, init_simnet_weights(CFG, algo, seed=1), max_batch=N, algo=algo)
    for _ in range(3):
        tr.step(dense, cat, labels, None if algo == 'dnn' else seqs)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    K = 20
    for _ in range(K):
        tr.step(dense, cat, labels, None if algo == 'dnn' else seqs)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / K
    print('%-9s %.2f ms / step of %d samples = %.0f samples/s' % (algo, dt * 1e3, N, N / dt))
    tr.close()
