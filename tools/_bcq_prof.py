import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from rl4rs_amd import offline_rl as R
D, E, n = 266, 32, 16384
rs = np.random.RandomState(0)
obs = rs.randn(n, D).astype(np.float32)
act = rs.randn(n, E).astype(np.float32)
act /= np.linalg.norm(act, axis=1, keepdims=True)
tr = tuple(torch.from_numpy(x).cuda() for x in (obs, act, rs.rand(n).astype(np.float32), np.roll(obs, -1, 0), (rs.rand(n) < 0.1).astype(np.float32)))
bcq = R.BCQ({'action_emb_size': E}, D, batch_size=256, seed=1, predict_rows=4096)
mode = sys.argv[1]
if mode == 'update':
    bcq.fit(tr, n_steps=20)
else:
    x = tr[0][:4096].contiguous()
    for _ in range(10):
        bcq.predict(x)
torch.cuda.synchronize()
