"""Episode rate of the reference-shaped API (numpy / list returns: obs [B,256] crosses PCIe every step)."""
import os, sys, tempfile, time
sys.path.insert(0, '.')
import numpy as np
import bench
class A: pass
a = A(); a.env = 'slate'; a.batch = 4096; a.horizon = 9; a.log_records = 8193; a.scorer = 'auto'
cfg, records = bench.make_config(a, tempfile.mkdtemp(), 0)
cfg['return_tensors'] = False
env = bench.build_env(cfg, False)
import torch
env.sim._recData.store.preload(torch.device('cuda', 0))
def ep():
    env.reset()
    for _ in range(9):
        obs, r, d, i = env.step(env.offline_action)
for _ in range(2): ep()
torch.cuda.synchronize(); t = time.perf_counter()
n = 5
for _ in range(n): ep()
torch.cuda.synchronize(); dt = time.perf_counter() - t
print('compat (numpy/list, PCIe-inclusive) env-steps/s: %.0f  (%.1f ms per episode-batch)' % (n * 4096 * 9 / dt, dt / n * 1e3))
