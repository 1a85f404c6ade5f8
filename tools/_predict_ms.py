import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from rl4rs_amd import offline_rl as R
D, E = 266, 32
rs = np.random.RandomState(0)
x = torch.from_numpy(rs.randn(4096, D).astype(np.float32)).cuda()
bcq = R.BCQ({'action_emb_size': E}, D, batch_size=256, seed=1, predict_rows=4096)
for _ in range(3):
    bcq.predict(x)
torch.cuda.synchronize()
t0 = time.time()
for _ in range(20):
    bcq.predict(x)
torch.cuda.synchronize()
print(os.environ.get('RL4RS_LIB', 'head'), 'predict %.3f ms' % ((time.time() - t0) / 20 * 1e3))
