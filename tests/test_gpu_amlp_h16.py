"""rl4rs_amlp_forward_h16 (k_amlp_fwd_h16: the three layers of a continuous learner's network as ONE launch in fp16x2 arithmetic,
for rows that never see a backward) and the device-side fragment packer in front of it.  Checker: a float64 restatement of the
network (oracle-side arithmetic only: numpy), tolerance 2e-5 absolute + relative - the fp16 hi + lo operands carry 22 of the 24
significand bits, the products accumulate in fp32 like the fp32 forward they stand in for."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _ref(params, obs, act, rep, head_act):
    p = {k: np.asarray(v, dtype=np.float64) for k, v in params.items()}
    x = np.concatenate([np.repeat(np.asarray(obs, np.float64), rep, axis=0), np.asarray(act, np.float64)], axis=1)
    h = np.maximum(x @ p['fc1_w'] + p['fc1_b'], 0.0)
    h = np.maximum(h @ p['fc2_w'] + p['fc2_b'], 0.0)
    o = h @ p['head_w'] + p['head_b']
    return np.tanh(o) if head_act == 'tanh' else o


def _net(D, E, K, head_act, max_rows, seed, heads=1):
    from rl4rs_amd import device as D_
    from rl4rs_amd.offline_rl import init_amlp_params
    params = init_amlp_params(D, E, K, seed=seed, heads=heads)
    # biases away from zero and a few large / tiny weight columns: the per-column prescale and the trailer matter
    rs = np.random.RandomState(seed)
    for k in ('fc1_b', 'fc2_b', 'head_b'):
        params[k] = (0.1 * rs.standard_normal(params[k].shape)).astype(np.float32)
    params['fc2_w'][:, 3] *= 300.0
    params['fc2_w'][:, 7] *= 1e-4
    params['head_w'][:, 0] *= 50.0
    net = D_.DeviceAMLP(D, E, K, params, head_act=head_act, max_rows=max_rows, max_grad_rows=256)
    net.H16_MIN_ROWS = 0
    return net, params


@pytest.mark.parametrize('K,N', [(300, 256), (32, 256), (256, 32), (256, 1), (64, 70), (17, 33)])
def test_device_pack_is_bit_identical_to_the_host_pack(K, N):
    import torch
    from rl4rs_amd import _lib
    from rl4rs_amd.device import check, _stream
    lib = _lib.load()
    rs = np.random.RandomState(K * 1000 + N)
    w = (rs.standard_normal((K, N)) / np.sqrt(K)).astype(np.float32)
    w[:, 0] *= 1e3
    if N > 2:
        w[:, 2] *= 1e-5
        w[:, 1] = 0.0
    w[rs.randint(K), rs.randint(N)] = 3.0e4
    bad = C.c_int64(-1)
    with torch.cuda.device(0):
        check(lib.rl4rs_pack_h16_selftest(w.ctypes.data_as(C.c_void_p), N, K, N, C.byref(bad), _stream()))
    assert bad.value == 0


@pytest.mark.parametrize('E,K,head_act,rep,rows', [(32, 32, 'tanh', 100, 41 * 100), (32, 1, 'none', 100, 64 * 100), (32, 32, 'tanh', 1, 4099),
                                                   (64, 64, 'none', 10, 330), (8, 33, 'none', 1, 65), (32, 1, 'none', 1, 1)])
def test_fused_forward_matches_float64(E, K, head_act, rep, rows):
    import torch
    D = 266
    net, params = _net(D, E, K, head_act, max_rows=max(rows, 256), seed=E + K)
    assert net.h16_ok
    rs = np.random.RandomState(rows)
    obs = rs.standard_normal((rows // rep, D)).astype(np.float32)
    act = rs.uniform(-1, 1, (rows, E)).astype(np.float32)
    o, a = torch.from_numpy(obs).cuda(), torch.from_numpy(act).cuda()
    got = net.forward(o, a, rep=rep, nograd='fp16x2').cpu().numpy()
    f32 = net.forward(o, a, rep=rep).cpu().numpy()
    want = _ref(params, obs, act, rep, head_act)
    scale = np.abs(want).max() + 1.0
    print('max |err|: fp16x2 %.3g, fp32 %.3g (scale %.3g)' % (np.abs(got - want).max(), np.abs(f32 - want).max(), scale))
    assert np.abs(got - want).max() <= 5e-5 * scale, (np.abs(got - want).max(), scale)      # weight columns scaled by 300 / 50 amplify every rounding
    # no further from float64 than the fp32 forward it replaces, up to a small factor
    assert np.abs(got - want).max() <= 4 * np.abs(f32 - want).max() + 2e-6 * scale
    # rows are independent of their position in the launch (same rows, other offset)
    if rep == 1 and rows > 200:
        again = net.forward(o[70:].contiguous(), a[70:].contiguous(), nograd='fp16x2').cpu().numpy()
        assert np.array_equal(again, got[70:])
    net.close()


def test_backward_after_the_fused_forward_is_refused_and_shapes_without_the_form_fall_back():
    import torch
    from rl4rs_amd import device as D_
    from rl4rs_amd.offline_rl import init_amlp_params
    net, _ = _net(266, 32, 1, 'none', 256, 5)
    o = torch.randn(200, 266, device='cuda')
    a = torch.rand(200, 32, device='cuda')
    net.forward(o, a, nograd='fp16x2')
    with pytest.raises(Exception):
        net.backward(o, a, torch.ones(200, 1, device='cuda'))
    net.forward(o, a)
    net.backward(o, a, torch.ones(200, 1, device='cuda'))          # the fp32 forward keeps its activations
    net.close()
    plain = D_.DeviceAMLP(266, 0, 64, init_amlp_params(266, 0, 64, seed=1, heads=2), max_rows=256)      # CQL's policy: no action input
    assert not plain.h16_ok
    plain.H16_MIN_ROWS = 0
    out = plain.forward(o, None, nograd='fp16x2')
    assert torch.isfinite(out).all()
    plain.close()


def test_out_of_range_rows_come_back_nan():
    import torch
    net, params = _net(266, 32, 32, 'tanh', 512, 9)
    rs = np.random.RandomState(1)
    obs = rs.standard_normal((300, 266)).astype(np.float32)
    act = rs.uniform(-1, 1, (300, 32)).astype(np.float32)
    act[17, 5] = 1.0e5            # beyond the largest finite fp16 number
    got = net.forward(torch.from_numpy(obs).cuda(), torch.from_numpy(act).cuda(), nograd='fp16x2').cpu().numpy()
    assert np.isnan(got[17]).all()
    ok = np.delete(np.arange(300), 17)
    assert np.isfinite(got[ok]).all()
    want = _ref(params, obs, act, 1, 'tanh')
    assert np.abs(got[ok] - want[ok]).max() <= 4e-5
    net.close()
