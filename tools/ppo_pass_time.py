"""Time one PPO SGD pass (rl4rs_policy_ppo_epoch) on synthetic samples: ms per pass and us per minibatch.
RL4RS_POLICY_OPTS=ppo_fused=0 selects the per-minibatch kernel chain instead of the persistent k_ppo_pass."""
import sys
import time

import numpy as np
import torch

sys.path.insert(0, '.')
from rl4rs_amd.device import DevicePolicy            # noqa: E402
from rl4rs_amd.nets.policy import init_policy_params  # noqa: E402


def main():
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 36864
    MB = int(sys.argv[2]) if len(sys.argv) > 2 else 256
    rs = np.random.RandomState(0)
    A = 284
    obs = torch.from_numpy(rs.randn(N, 256).astype(np.float32)).cuda()
    mask = rs.rand(N, A) < 0.4
    mask[:, 1] = True
    bits = np.zeros((N, 9), np.uint32)
    for k in range(A):
        bits[:, k >> 5] |= (mask[:, k].astype(np.uint32) << np.uint32(k & 31))
    bits = torch.from_numpy(bits.view(np.int32)).cuda()
    pol = DevicePolicy(256, 64, A, max_rows=N, params=init_policy_params(seed=2))
    a, lp, v, ent, lg = pol.act(obs, mask_bits=bits, seed=1, step=0, want_logits=True)
    adv = torch.randn(N, device='cuda')
    ret = torch.randn(N, device='cuda') * 10
    for rep in range(3):
        torch.cuda.synchronize()
        t0 = time.time()
        pol.ppo_epoch(obs, a, adv, ret, bits, lp, v, lg, minibatch=MB, lr=1e-4)
        torch.cuda.synchronize()
        dt = time.time() - t0
        print('pass %d: %.2f ms, %.1f us per minibatch (%d minibatches)' % (rep, dt * 1e3, dt * 1e6 / (N // MB), N // MB))


if __name__ == '__main__':
    main()
