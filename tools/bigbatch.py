"""B = big (all distinct records) vs the same records in chunks of 4096: observations / rewards must be bit-identical."""
import os, sys, time, tempfile
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rl4rs_amd
from rl4rs_amd import synth
from rl4rs_amd.env.slate import SlateRecEnv, SlateState

BIG = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
CH = 4096
d = tempfile.mkdtemp()
text = synth.make_catalog_text(seed=1234)
synth.write_text(os.path.join(d, 'c.csv'), text)
t0 = time.time()
recs = synth.make_records(BIG, seed=1000, illegal_frac=0.05, special_ids=synth.special_ids_from_text(text))
print('records', time.time() - t0, flush=True)

def run(rs, tag):
    B = len(rs)
    p = os.path.join(d, 'log_%s.csv' % tag)
    synth.write_records(p, rs)
    cfg = {"maxlen": 64, "batch_size": B, "action_size": 284, "class_num": 2, "dense_feature_num": 432,
           "category_feature_num": 21, "category_hash_size": 100000, "seq_num": 2, "emb_size": 128, "page_items": 9,
           "hidden_units": 128, "max_steps": 9, "action_emb_size": 32, "sample_file": p,
           "iteminfo_file": os.path.join(d, 'c.csv'), "is_eval": True, "cache_size": B, "model_seed": 7,
           "return_tensors": True}
    env = rl4rs_amd.make('SlateRecEnv-v0', recsim=SlateRecEnv(cfg, state_cls=SlateState))
    obs = env.reset(reset_file=True)
    out = [obs.clone()]
    for t in range(9):
        a = env.offline_action
        obs, reward, done, info = env.step(a)
        out.append(obs.clone())
    torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(2):
        obs = env.reset(reset_file=True)
        for t in range(9):
            obs, reward, done, info = env.step(env.offline_action)
    torch.cuda.synchronize()
    dt = (time.time() - t0) / 2
    print(tag, 'B', B, 'ms per episode-batch %.2f' % (dt * 1e3), 'env-steps/s %.3f M' % (B * 9 / dt / 1e6), flush=True)
    res = torch.stack(out).cpu(), reward.clone().cpu()
    env.close() if hasattr(env, 'close') else None
    del env
    torch.cuda.empty_cache()
    return res

big_obs, big_r = run(recs, 'big')
assert torch.isfinite(big_obs).all() and torch.isfinite(big_r).all()
bad = 0
for k in range(0, BIG, CH):
    o, r = run(recs[k:k + CH], 'c%d' % (k // CH))
    eo = torch.equal(o, big_obs[:, k:k + CH]); er = torch.equal(r, big_r[k:k + CH])
    print('chunk', k // CH, 'obs identical', eo, 'reward identical', er, 'max |d obs|', float((o - big_obs[:, k:k + CH]).abs().max()), flush=True)
    bad += (not eo) + (not er)
print('RESULT', 'OK' if bad == 0 else 'MISMATCH %d' % bad)
