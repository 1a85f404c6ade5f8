"""Updates per second of the continuous-action CQL learner ('CQL-conti': batch 256, 10 action samples -> 31 rows per transition)
on synthetic transitions.  usage: python tools/cql_conti_rate.py [steps]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rl4rs_amd import offline_rl as R          # noqa: E402


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    D, E, n = 266, 32, 16384
    rs = np.random.RandomState(0)
    obs = rs.randn(n, D).astype(np.float32)
    act = rs.randn(n, E).astype(np.float32)
    act /= np.linalg.norm(act, axis=1, keepdims=True)
    rew = rs.rand(n).astype(np.float32)
    tr = tuple(torch.from_numpy(x).cuda() for x in (obs, act, rew, np.roll(obs, -1, 0), (rs.rand(n) < 0.1).astype(np.float32)))
    cql = R.CQL({'action_emb_size': E}, D, batch_size=256, gamma=1.0, reward_scaler=R.StandardRewardScaler(rew), seed=1,
                nograd_precision=os.environ.get('NOGRAD', 'fp16x2'))
    cql.fit(tr, n_steps=10)
    torch.cuda.synchronize()
    t0 = time.time()
    cql.fit(tr, n_steps=steps)
    torch.cuda.synchronize()
    dt = time.time() - t0
    print('CQL-conti %.3f ms / update of 256 transitions (31 rows each through the critics) = %.0f transitions/s' % (dt / steps * 1e3, steps * 256 / dt))
    cql.close()


if __name__ == '__main__':
    main()
