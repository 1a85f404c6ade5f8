"""Updates per second of the continuous-action BCQ learner ('BCQ-conti': batch 256, 100 sampled actions per target row) on
synthetic transitions, and the rate of its greedy prediction (100 sampled actions per observation).
usage: python tools/bcq_conti_rate.py [steps]"""
import sys
import time

import numpy as np
import torch

sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
from rl4rs_amd import offline_rl as R          # noqa: E402


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    D, E, n = 266, 32, 16384
    rs = np.random.RandomState(0)
    obs = rs.randn(n, D).astype(np.float32)
    act = rs.randn(n, E).astype(np.float32)
    act /= np.linalg.norm(act, axis=1, keepdims=True)
    tr = tuple(torch.from_numpy(x).cuda() for x in (obs, act, rs.rand(n).astype(np.float32), np.roll(obs, -1, 0),
                                                     (rs.rand(n) < 0.1).astype(np.float32)))
    from rl4rs_amd import device as Dv
    Dv.amlp_set_fused(int(__import__('os').environ.get('FUSED', '1')))          # 0: per-layer launches, 1: fused (default), 2: 8-row fused form
    prec = __import__('os').environ.get('NOGRAD', 'fp16x2')
    rows = int(__import__('os').environ.get('PREDICT_ROWS', '512'))
    bcq = R.BCQ({'action_emb_size': E}, D, batch_size=256, seed=1, nograd_precision=prec, predict_rows=rows)
    print('nograd_precision', prec, 'predict_rows', rows)
    bcq.fit(tr, n_steps=10)
    torch.cuda.synchronize()
    t0 = time.time()
    bcq.fit(tr, n_steps=steps)
    torch.cuda.synchronize()
    dt = time.time() - t0
    print('BCQ-conti %.3f ms / update of 256 transitions (100 sampled actions each) = %.0f transitions/s' % (dt / steps * 1e3, steps * 256 / dt))
    x = tr[0][:4096].contiguous()
    bcq.predict(x)
    torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(5):
        bcq.predict(x)
    torch.cuda.synchronize()
    dt = (time.time() - t0) / 5
    print('predict: %.2f ms per 4096 observations (409 600 sampled rows) = %.0f observations/s' % (dt * 1e3, 4096 / dt))
    bcq.close()


if __name__ == '__main__':
    main()
