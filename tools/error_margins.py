"""Measured error of the HIP scorer against the fp64 oracle (and of the fp32 numpy oracle, for scale)."""
import sys, numpy as np, torch
sys.path.insert(0, '.')
from rl4rs_amd.nets.dien import init_dien_weights
from rl4rs_amd.device import DeviceDien
from oracle.dien import OracleDien
CFG = {"maxlen": 64, "batch_size": 8, "action_size": 284, "class_num": 2, "dense_feature_num": 432,
       "category_feature_num": 21, "category_hash_size": 3000, "seq_num": 2, "emb_size": 128,
       "page_items": 9, "hidden_units": 128, "max_steps": 9, "action_emb_size": 32}
for name, kw in (('keras-default init (emb 0.05)', dict(emb_scale=0.05)), ('stress init (emb 0.5, bias noise)', dict(emb_scale=0.5, bias_noise=0.2))):
    w = init_dien_weights(CFG, seed=3, **kw)
    rs = np.random.RandomState(1)
    R = 512
    seq = rs.randint(0, 284, size=(R, 2, 64)).astype(np.int32)
    dense = np.abs(rs.randn(R, 432) * 3).astype(np.float32)
    cat = rs.randint(0, 3000, size=(R, 21)).astype(np.int32)
    net = DeviceDien(CFG, w, max_rows=R, max_slots=R)
    for s in range(2):
        net.encode(s, torch.from_numpy(np.ascontiguousarray(seq[:, s])).cuda(), 0)
    slots = torch.arange(R, dtype=torch.int32).repeat(2, 1).contiguous().cuda()
    obs, prob = net.forward(R, 1, torch.from_numpy(dense).cuda(), torch.from_numpy(cat).cuda(), slots, True, True)
    o64 = OracleDien(w, CFG, np.float64); o32 = OracleDien(w, CFG, np.float32)
    ref, ref32 = o64.obs(seq, dense, cat), o32.obs(seq, dense, cat)
    p64, p32 = o64.reward_probs(seq, dense, cat)[:, 1], o32.reward_probs(seq, dense, cat)[:, 1]
    print('%s: |obs| max %.2f' % (name, np.abs(ref).max()))
    print('   HIP  vs fp64: obs max abs err %.2e, prob max abs err %.2e, prob max rel err %.2e'
          % (np.abs(obs.cpu().numpy() - ref).max(), np.abs(prob.cpu().numpy() - p64).max(), (np.abs(prob.cpu().numpy() - p64) / p64).max()))
    print('   numpy fp32 vs fp64: obs %.2e, prob %.2e' % (np.abs(ref32 - ref).max(), np.abs(p32 - p64).max()))
    net.close()
