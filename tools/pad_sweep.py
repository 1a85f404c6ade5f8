import sys, numpy as np, torch
sys.path.insert(0, '.')
from rl4rs_amd.nets.dien import init_dien_weights
from rl4rs_amd.device import DeviceDien, DIEN_H1
bad = 0
for seed in range(int(sys.argv[1]) if len(sys.argv) > 1 else 16):
    L = [64, 64, 33, 50][seed % 4]
    CFG = {"maxlen": L, "batch_size": 8, "action_size": 284, "class_num": 2, "dense_feature_num": 432,
           "category_feature_num": 21, "category_hash_size": 3000, "seq_num": 2, "emb_size": 128,
           "page_items": 9, "hidden_units": 128, "max_steps": 9, "action_emb_size": 32, 'scorer_precision': 'fp16x2'}
    w = init_dien_weights(CFG, seed=100 + seed, emb_scale=0.3 + 0.1 * (seed % 5), bias_noise=0.2)
    rs = np.random.RandomState(seed)
    R = 300
    seq = rs.randint(1, 284, size=(R, 2, L)).astype(np.int32)
    for b in range(0, R, 32):                          # every 32-row block its own common prefix + per-row extras
        base = rs.randint(0, L + 1)
        for r in range(b, min(R, b + 32)):
            seq[r, 0, :min(L, base + rs.randint(0, 4))] = 0
    seq[:, 1, :rs.randint(L // 2, L)] = 0
    dense = np.abs(rs.randn(R, 432) * 3).astype(np.float32)
    cat = rs.randint(0, 3000, size=(R, 21)).astype(np.int32)
    def run(kernels):
        net = DeviceDien(dict(CFG, scorer_kernels=kernels), w, max_rows=R, max_slots=R)
        for s in range(2):
            net.encode(s, torch.from_numpy(np.ascontiguousarray(seq[:, s])).cuda(), 0)
        h1 = net.snapshot(DIEN_H1, R)[:R].clone()
        sl = torch.arange(R, dtype=torch.int32).repeat(2, 1).contiguous().cuda()
        obs, p = net.forward(R, 1, torch.from_numpy(dense).cuda(), torch.from_numpy(cat).cuda(), sl, True, True)
        out = (h1, obs.clone(), p.clone())
        net.close()
        return out
    a, b = run(''), run('no_gru_pad')
    ok = all(torch.equal(x, y) for x, y in zip(a, b))
    bad += not ok
    print('seed', seed, 'L', L, 'identical', ok, flush=True)
print('SWEEP', 'OK' if bad == 0 else 'MISMATCH %d' % bad)
