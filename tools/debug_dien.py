import sys, numpy as np, torch
sys.path.insert(0, '.')
from rl4rs_amd.nets.dien import init_dien_weights
from rl4rs_amd.device import DeviceDien, DIEN_ALL_FEATURE, DIEN_SCORES, DIEN_QUERY, DIEN_H1
from oracle.dien import OracleDien
CFG = {"maxlen": 64, "batch_size": 8, "action_size": 284, "class_num": 2, "dense_feature_num": 432,
       "category_feature_num": 21, "category_hash_size": 3000, "seq_num": 2, "emb_size": 128,
       "page_items": 9, "hidden_units": 128, "max_steps": 9, "action_emb_size": 32}
R = 40
w = init_dien_weights(CFG, seed=3, emb_scale=0.5, bias_noise=0.2)
rs = np.random.RandomState(R)
seq = rs.randint(0, 284, size=(R, 2, 64)).astype(np.int32)
dense = np.abs(rs.randn(R, 432) * 3).astype(np.float32)
cat = rs.randint(0, 3000, size=(R, 21)).astype(np.int32)
net = DeviceDien(CFG, w, max_rows=R, max_slots=R)
for s in range(2):
    net.encode(s, torch.from_numpy(np.ascontiguousarray(seq[:, s])).cuda(), 0)
slots = torch.arange(R, dtype=torch.int32).repeat(2, 1).contiguous().cuda()
obs, prob = net.forward(R, 1, torch.from_numpy(dense).cuda(), torch.from_numpy(cat).cuda(), slots, True, True)
torch.cuda.synchronize()
orc = OracleDien(w, CFG, np.float64)
allf, parts = orc.features(seq, dense, cat, return_parts=True)
h1 = net.snapshot(DIEN_H1, R).cpu().numpy()[:R]
err = np.abs(h1 - parts['h1_0'])
print('h1 err by t (first 6):', err.max(axis=(0, 2))[:6])
print('h1 err by col block of 32 at t=0:', [float(err[:, 0, i*32:(i+1)*32].max()) for i in range(4)])
print('h1 err by row at t=0:', err[:, 0].max(axis=1)[:8], '...', err[:, 0].max(axis=1)[30:36])
af = net.snapshot(DIEN_ALL_FEATURE, R).cpu().numpy()[:R]
print('h2_0 err', np.abs(af[:, :256] - parts['h2_0']).max(), 'h2_1 err', np.abs(af[:, 256:512] - parts['h2_1']).max())
print('obs err', np.abs(obs.cpu().numpy() - orc.obs(seq, dense, cat)).max())
