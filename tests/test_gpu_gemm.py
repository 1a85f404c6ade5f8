"""fp32 MFMA GEMM kernel vs torch fp64 (transpose-detecting: asymmetric random operands)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('M,N,K,act', [(1, 2, 3, 0), (33, 65, 17, 1), (128, 64, 32, 0), (300, 284, 64, 3),
                                       (257, 128, 432, 1), (1000, 256, 3456, 1), (4096, 832, 128, 0),
                                       (77, 2, 256, 2)])
def test_gemm_f32(M, N, K, act):
    import torch
    from rl4rs_amd.device import gemm_f32
    g = torch.Generator().manual_seed(M * 31 + N)
    a = torch.randn(M, K, generator=g).cuda()
    w = (torch.randn(K, N, generator=g) / np.sqrt(K)).cuda()
    b = torch.randn(N, generator=g).cuda()
    c = gemm_f32(a, w, b, act)
    ref = a.double() @ w.double() + b.double()
    if act == 1:
        ref = torch.where(ref > 0, ref, torch.expm1(ref))
    elif act == 2:
        ref = torch.sigmoid(ref)
    elif act == 3:
        ref = torch.tanh(ref)
    err = (c.double() - ref).abs().max().item()
    assert err < 2e-5, err
    from rl4rs_amd.device import gemm_f32_packed
    cp = gemm_f32_packed(a, w.cpu().numpy(), b, act)
    assert (cp.double() - ref).abs().max().item() < 2e-5
    # strided A (a column slice of a wider matrix) and no bias
    wide = torch.randn(M, K + 8, generator=g).cuda()
    c2 = gemm_f32(wide[:, 4:4 + K], w, None, 0)
    ref2 = wide[:, 4:4 + K].double() @ w.double()
    assert (c2.double() - ref2).abs().max().item() < 2e-5
