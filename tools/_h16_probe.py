import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from rl4rs_amd.device import gemm_f32, gemm_h16_packed
for M, N, K in ((25600, 256, 256), (409600, 256, 256), (409600, 256, 32), (409600, 32, 256)):
    a = torch.randn(M, K, device='cuda')
    w = (np.random.randn(K, N) / np.sqrt(K)).astype(np.float32)
    b = torch.randn(N, device='cuda')
    for _ in range(3):
        c = gemm_h16_packed(a, w, b, 4)
    ref = gemm_f32(a, torch.from_numpy(w).cuda(), b, 4)
    print(M, N, K, float((c - ref).abs().max()))
