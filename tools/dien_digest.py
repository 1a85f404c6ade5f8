#!/usr/bin/env python
"""sha256 of two episode-batches (observations of every step + rewards) of the bench configuration at B = 1024: for bit-identity
A/Bs of library variants (tools/build_ab.py):   RL4RS_LIB=tools/_ab/<name>/librl4rs_hip.so python tools/dien_digest.py"""
import argparse
import hashlib
import os
import sys
import tempfile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


def main():
    args = argparse.Namespace(env='slate', log_records=4096, batch=1024, horizon=9, scorer=os.environ.get('SCORER', 'fp16x2'),
                              conti=False, train='none', algo='dien')
    with tempfile.TemporaryDirectory() as d:
        cfg, _ = bench.make_config(args, d, 0)
        env = bench.build_env(cfg, False)
        h = hashlib.sha256()
        for _ in range(2):
            obs = env.reset()
            h.update(obs.cpu().numpy().tobytes())
            for _ in range(9):
                obs, reward, done, info = env.step(env.offline_action)
                h.update(obs.cpu().numpy().tobytes())
                h.update(reward.cpu().numpy().tobytes())
        print(os.environ.get('RL4RS_LIB', 'head'), h.hexdigest())


if __name__ == '__main__':
    main()
