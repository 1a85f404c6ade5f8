"""TF/s of the fp32 GEMM forms on the shapes of the continuous learners' large forwards (M rows x 256 x K)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from rl4rs_amd.device import gemm_f32, gemm_f32_packed, gemm_h16_packed

for M, N, K in ((25600, 256, 256), (409600, 256, 256), (25600, 256, 32), (409600, 256, 32), (409600, 32, 256)):
    a = torch.randn(M, K, device='cuda')
    w = torch.randn(K, N, device='cuda') / np.sqrt(K)
    b = torch.randn(N, device='cuda')
    for name, fn in (('k_gemm_f32', lambda: gemm_f32(a, w, b, 4)),):
        fn(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            fn()
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 10
        print('%-12s M=%6d N=%3d K=%3d  %.1f us  %.1f TF/s  (%.2f of 157.3)  out+in bytes %.0f GB/s' % (
            name, M, N, K, ms * 1e3, 2.0 * M * N * K / ms / 1e9, 2.0 * M * N * K / ms / 1e9 / 157.3, (M * N + M * K) * 4 / ms / 1e6))
