import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from rl4rs_amd import offline_rl as R
D, E, n = 266, 32, 16384
rs = np.random.RandomState(0)
obs = rs.randn(n, D).astype(np.float32); act = rs.randn(n, E).astype(np.float32)
tr = tuple(torch.from_numpy(x).cuda() for x in (obs, act, rs.rand(n).astype(np.float32), np.roll(obs, -1, 0), (rs.rand(n) < 0.1).astype(np.float32)))
bcq = R.BCQ({'action_emb_size': E}, D, batch_size=256, seed=1)
bcq.fit(tr, n_steps=20, to_host=False)
torch.cuda.synchronize()
t0 = time.time()
bcq.fit(tr, n_steps=300, to_host=False)
t1 = time.time()
torch.cuda.synchronize()
t2 = time.time()
print('host issue %.3f ms/update, total %.3f ms/update' % ((t1 - t0) / 300 * 1e3, (t2 - t0) / 300 * 1e3))
