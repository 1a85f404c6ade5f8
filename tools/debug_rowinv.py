import sys, numpy as np, torch
sys.path.insert(0, '.')
from rl4rs_amd.nets.dien import init_dien_weights
from rl4rs_amd.device import DeviceDien, DIEN_ALL_FEATURE, DIEN_SCORES, DIEN_QUERY
CFG = {"maxlen": 64, "batch_size": 8, "action_size": 284, "class_num": 2, "dense_feature_num": 432,
       "category_feature_num": 21, "category_hash_size": 3000, "seq_num": 2, "emb_size": 128,
       "page_items": 9, "hidden_units": 128, "max_steps": 9, "action_emb_size": 32}
B, G = 10, 8
w = init_dien_weights(CFG, seed=3, emb_scale=0.5, bias_noise=0.2)
rs = np.random.RandomState(0)
seq = rs.randint(0, 284, size=(B, 64)).astype(np.int32)
dense = np.abs(rs.randn(B, 432) * 3).astype(np.float32)
cat = rs.randint(0, 3000, size=(B, 21)).astype(np.int32)
net = DeviceDien(CFG, w, max_rows=B * G, max_slots=B + 1)
net.encode(0, torch.from_numpy(seq).cuda(), 0)
net.encode(1, torch.zeros((1, 64), dtype=torch.int32).cuda(), B)
sl = torch.full((2, B), B, dtype=torch.int32).cuda(); sl[0] = torch.arange(B, dtype=torch.int32).cuda()
d1, c1 = torch.from_numpy(dense).cuda(), torch.from_numpy(cat).cuda()
obs1, p1 = net.forward(B, 1, d1, c1, sl.contiguous(), True, True)
af1 = net.snapshot(DIEN_ALL_FEATURE, B).clone(); sc1 = net.snapshot(DIEN_SCORES, B).clone(); q1 = net.snapshot(DIEN_QUERY, B).clone()
dG = d1.repeat_interleave(G, dim=0).contiguous(); cG = c1.repeat_interleave(G, dim=0).contiguous()
obsG, pG = net.forward(B * G, G, dG, cG, sl.contiguous(), True, True)
afG = net.snapshot(DIEN_ALL_FEATURE, B * G); scG = net.snapshot(DIEN_SCORES, B * G); qG = net.snapshot(DIEN_QUERY, B * G)
idx = torch.arange(B).cuda() * G + (G - 1)
print('q', (q1[:B] - qG[idx]).abs().max().item())
print('scores s0', (sc1[0, :B] - scG[0, idx]).abs().max().item(), 's1', (sc1[1, :B] - scG[1, idx]).abs().max().item())
for name, lo, hi in (('h2_0', 0, 256), ('h2_1', 256, 512), ('dense', 512, 640), ('cat', 640, 3456)):
    print(name, (af1[:B, lo:hi] - afG[idx, lo:hi]).abs().max().item())
print('obs', (obs1 - obsG[idx]).abs().max().item(), 'p', (p1 - pG[idx]).abs().max().item())
print('--- determinism / position checks')
obs1b, p1b = net.forward(B, 1, d1, c1, sl.contiguous(), True, True)
print('run-to-run same call:', (obs1 - obs1b).abs().max().item())
obsG2, _ = net.forward(B * G, G, dG, cG, sl.contiguous(), True, True)
print('run-to-run G call:', (obsG - obsG2).abs().max().item())
afG = net.snapshot(DIEN_ALL_FEATURE, B * G)
rep = afG[:B * G, :512].reshape(B, G, 512)
print('within-call replicas (same env, G positions): max dev from pos 0 =', (rep - rep[:, :1]).abs().max().item())
dev = (rep - rep[:, :1]).abs().amax(dim=2)
print(dev[:4])
