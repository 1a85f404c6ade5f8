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
    # rows that are not 16-byte aligned (the scalar-load path of the small-M form, the per-element path of the big one)
    c3 = gemm_f32(wide[:, 3:3 + K], w, None, 0)
    ref3 = wide[:, 3:3 + K].double() @ w.double()
    assert (c3.double() - ref3).abs().max().item() < 2e-5


@pytest.mark.parametrize('M,N,K,act', [(1, 2, 3, 0), (33, 65, 17, 1), (64, 64, 64, 0), (300, 284, 70, 3), (4096, 128, 432, 1),
                                       (4096, 256, 768, 0), (32768, 256, 768, 0), (5000, 832, 128, 0), (77, 2, 256, 2),
                                       (9001, 832, 128, 0), (8300, 300, 128, 1)])      # K = 128, many rows: k_gemm_h16_wres
def test_gemm_h16(M, N, K, act):
    """fp16x2 GEMM (the scorer's GEMMs in scorer_mode fp16x2) vs fp64: same bar as the exact-fp32 kernel, on operands that
    exercise the split: large and tiny magnitudes next to each other, strided A, row / column / k tails."""
    import torch
    from rl4rs_amd.device import gemm_h16_packed, gemm_f32_packed
    g = torch.Generator().manual_seed(M * 17 + N + K)
    a = torch.randn(M, K, generator=g)
    a[:, ::3] *= 30.0                       # activations O(30) next to O(1)
    a[:, 1::5] *= 1e-3                      # and values whose lo part is an fp16 subnormal
    w = torch.randn(K, N, generator=g) / np.sqrt(K)
    w[::7] *= 1e-3
    b = torch.randn(N, generator=g)
    wide = torch.zeros(M, K + 8)
    wide[:, 4:4 + K] = a
    wide = wide.cuda()
    av = wide[:, 4:4 + K]                   # a column slice of a wider matrix (lda != K; 16-byte aligned rows)
    ref = a.double() @ w.double() + b.double()
    if act == 1:
        ref = torch.where(ref > 0, ref, torch.expm1(ref))
    elif act == 2:
        ref = torch.sigmoid(ref)
    elif act == 3:
        ref = torch.tanh(ref)
    c = gemm_h16_packed(av, w.numpy(), b.cuda(), act)
    err = (c.double().cpu() - ref).abs().max().item()
    c32 = gemm_f32_packed(av, w.numpy(), b.cuda(), act)
    err32 = (c32.double().cpu() - ref).abs().max().item()
    scale = ref.abs().max().item()
    assert err < max(4e-6 * max(scale, 1.0), 2.0 * err32), (err, err32, scale)
    # unaligned A (scalar load path) gives the same bits
    odd = torch.zeros(M, K + 9)
    odd[:, 1:1 + K] = a
    c_odd = gemm_h16_packed(odd.cuda()[:, 1:1 + K], w.numpy(), b.cuda(), act)
    assert torch.equal(c_odd, c)


def test_gemm_h16_poisons_out_of_range_rows():
    import torch
    from rl4rs_amd.device import gemm_h16_packed
    g = torch.Generator().manual_seed(5)
    a = torch.randn(100, 96, generator=g)
    a[7, 13] = 7.0e4                         # beyond the fp16 range
    a[55, 80] = float('nan')
    w = torch.randn(96, 64, generator=g) / 10
    c = gemm_h16_packed(a.cuda(), w.numpy(), None, 0).cpu()
    bad = torch.isnan(c).any(dim=1)
    assert bad[7] and bad[55] and int(bad.sum()) == 2
    assert torch.isnan(c[7]).all() and torch.isnan(c[55]).all()
    ref = a.double() @ w.double()
    ok = ~bad
    assert (c[ok].double() - ref[ok]).abs().max().item() < 1e-5


@pytest.mark.parametrize('k', [-12, 0, 12, 17, 30])
def test_gemm_h16_over_weight_scales(k):
    """The packed fp16x2 GEMM has no weight-range condition: a column whose weights would not fit fp16 is stored times a power
    of two (pack_gemm_weight_h16) and the accumulator divided by it.  Weights of 2.4e-5 ... 1e8: same bar relative to the
    result's scale as at the natural scale, and the scaled-DOWN cases reproduce each other bit for bit (same hi / lo bit
    patterns, exact power-of-two factors)."""
    import torch
    from rl4rs_amd.device import gemm_h16_packed
    g = torch.Generator().manual_seed(3)
    a = torch.randn(300, 200, generator=g)
    w = torch.randn(200, 96, generator=g) / 10
    ws = (w * float(2.0 ** k)).numpy().astype(np.float32)
    c = gemm_h16_packed(a.cuda(), ws, None, 0).cpu()
    ref = a.double() @ torch.from_numpy(ws).double()
    assert torch.isfinite(c).all()
    assert ((c.double() - ref).abs().max() / ref.abs().max()).item() < 4e-6
    if k >= 17:
        c2 = gemm_h16_packed(a.cuda(), (ws * np.float32(8.0)).astype(np.float32), None, 0).cpu()
        assert torch.equal(c2, c * 8.0)


def test_gemm_h16_takes_weights_beyond_the_fp16_range():
    """|w| >= 6.5e4 used to force the exact-fp32 GEMM; with the prescale it is just another scale: same bar against fp64."""
    import torch
    from rl4rs_amd.device import gemm_h16_packed
    g = torch.Generator().manual_seed(4)
    a = torch.randn(257, 128, generator=g)
    w = torch.randn(128, 64, generator=g) / 10
    w[5, 7] = 7.0e4
    w[100, 3] = -2.5e5
    c = gemm_h16_packed(a.cuda(), w.numpy(), None, 0).cpu()
    ref = a.double() @ w.double()
    assert torch.isfinite(c).all()
    from rl4rs_amd.device import gemm_f32_packed
    c32 = gemm_f32_packed(a.cuda(), w.numpy(), None, 0).cpu()
    err = (c.double() - ref).abs().max(dim=0).values                    # per column: the outlier columns are O(1e5), the rest O(1)
    err32 = (c32.double() - ref).abs().max(dim=0).values
    scale = ref.abs().max(dim=0).values
    assert (err <= torch.maximum(4e-6 * scale, 2.0 * err32)).all(), (err / scale).max().item()
    assert scale[7] > 1e4 and scale[3] > 1e4 and (err / scale)[[3, 7]].max().item() < 1e-6
