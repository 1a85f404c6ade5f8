"""Updates per second of the offline-RL learners (DiscreteBC / BCQ / CQL, 256-sample minibatches) on synthetic transitions."""
import os
import sys
import tempfile
import time

import numpy as np
import torch

sys.path.insert(0, '.')
from rl4rs_amd import offline_rl as R          # noqa: E402
from rl4rs_amd import synth                    # noqa: E402
from rl4rs_amd.data import CatalogTables       # noqa: E402


def main():
    A, D, n = 284, 266, 16384
    d = tempfile.mkdtemp()
    path = os.path.join(d, 'item_info.csv')
    synth.write_text(path, synth.make_catalog_text(seed=21))
    tab = CatalogTables(path, A, 32)
    cfg = {'action_size': A, 'page_items': 9, 'location_mask': tab.location_mask, 'special_items': tab.special_items}
    rs = np.random.RandomState(0)
    obs = np.zeros((n, D), np.float32)
    obs[:, :256] = rs.randn(n, 256)
    obs[:, -1] = rs.randint(0, 10, n)
    tr = tuple(torch.from_numpy(x).cuda() for x in (obs, rs.randint(1, A, n).astype(np.int32), rs.rand(n).astype(np.float32),
                                                     np.roll(obs, -1, 0), (rs.rand(n) < 0.1).astype(np.float32)))
    for name, cls in (('BC', R.DiscreteBC), ('BCQ', R.DiscreteBCQ), ('CQL', R.DiscreteCQL)):
        learner = cls(cfg, D, batch_size=256, seed=1)
        learner.fit(tr, n_steps=20)
        torch.cuda.synchronize()
        t0 = time.time()
        steps = 200
        learner.fit(tr, n_steps=steps)
        torch.cuda.synchronize()
        dt = time.time() - t0
        print('%-4s %.2f ms / update of 256 transitions = %.0f transitions/s' % (name, dt / steps * 1e3, steps * 256 / dt))
        learner.close()


if __name__ == '__main__':
    main()
