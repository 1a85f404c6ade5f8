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


def trained_like(steps=300, R=512, lr=1e-3, verbose=True):
    """margins on a TRAINED-like model, not Glorot-synthetic weights: a few hundred Adam steps of the device
    DIEN trainer (rl4rs_dientrain_step) on synthetic logs, then both recurrences against the fp64 oracle on those weights.
    Training pushes the raw DIN attention scores away from zero (both signs), which is what decides whether (1 - a_t) u
    stays in a range the fp16 planes can carry (rl4rs/nets/utils.py:121-124: no softmax on the scores)."""
    import os, tempfile
    from rl4rs_amd import synth
    from rl4rs_amd.simtrain import SimulatorTrainer
    from rl4rs_amd.env.slate import SlateRecEnv, SlateState
    d = tempfile.mkdtemp(prefix='rl4rs_margins_')
    text = synth.make_catalog_text(seed=21)
    synth.write_text(os.path.join(d, 'c.csv'), text)
    synth.write_records(os.path.join(d, 'log.csv'), synth.make_records(1024, pages=1, seed=8, illegal_frac=0.0, hash_size=3000,
                                                                       special_ids=synth.special_ids_from_text(text)))
    cfg = dict(CFG, batch_size=256, sample_file=os.path.join(d, 'log.csv'), iteminfo_file=os.path.join(d, 'c.csv'), is_eval=False,
               cache_size=512, model_seed=3, return_tensors=True, algo='dien')
    sim = SlateRecEnv(cfg, state_cls=SlateState)
    tr = SimulatorTrainer(sim, minibatch=256, seed=2, lr=lr, dropout_rate=0.2)
    losses = []
    while len(losses) < steps:
        losses += tr.fit(windows=1, epochs=1)
    w = tr.export_weights()
    dense, cat, labels, seqs = tr.dataset_from_logs()
    seq = np.stack([s[:R].cpu().numpy() for s in seqs], axis=1)
    dense, cat = dense[:R].cpu().numpy(), cat[:R].cpu().numpy()
    out = dict(loss_first=float(np.mean(losses[:10])), loss_last=float(np.mean(losses[-10:])), steps=len(losses), variants={})
    # the trained weights as they are, and the same model with its attention head re-centred and widened to a score spread of
    # 0.1 (mean removed, output layer scaled; at a spread of 0.3 the recurrence itself is chaotic - |h2| reaches 4e10 in fp64
    # and the exact-fp32 kernel is off by 1e5 too; the fp16x2 kernel then NaN-poisons the rows it cannot carry): raw scores of BOTH signs, (1 - a_t) on both sides of 1 - a DIN head that
    # discriminates, which synthetic logs cannot teach in a few hundred steps
    o_tr = OracleDien(w, cfg, np.float64)
    _, parts_tr = o_tr.features(seq, dense, cat, return_parts=True)
    wide = dict(w)
    for i in range(2):
        sc_i = parts_tr['score_%d' % i]
        f = 0.1 / max(float(sc_i.std()), 1e-9)
        wide['att%d_w3' % i] = (w['att%d_w3' % i] * f).astype(np.float32)
        wide['att%d_b3' % i] = ((w['att%d_b3' % i] - sc_i.mean()) * f).astype(np.float32)
    for vname, wv in (('trained', w), ('trained, attention head centred and widened to std 0.1', wide)):
        o64 = OracleDien(wv, cfg, np.float64)
        allf, parts = o64.features(seq, dense, cat, return_parts=True)
        ref = o64.obs(seq, dense, cat)
        p64 = o64.reward_probs(seq, dense, cat)[:, 1]
        sc = np.concatenate([parts['score_0'].ravel(), parts['score_1'].ravel()])
        v = dict(score_min=float(sc.min()), score_max=float(sc.max()), h2_max=float(max(np.abs(parts['h2_0']).max(), np.abs(parts['h2_1']).max())))
        for mode in ('fp32', 'fp16x2'):
            net = DeviceDien(dict(cfg, scorer_precision=mode), wv, max_rows=R, max_slots=R)
            for s in range(2):
                net.encode(s, torch.from_numpy(np.ascontiguousarray(seq[:, s])).cuda(), 0)
            slots = torch.arange(R, dtype=torch.int32).repeat(2, 1).contiguous().cuda()
            obs, prob = net.forward(R, 1, torch.from_numpy(dense).cuda(), torch.from_numpy(cat).cuda(), slots, True, True)
            try:
                net.check_status()
                flagged = False
            except Exception:
                flagged = True                       # the fp16x2 recurrence left the fp16 range for some row: those rows are NaN
            o, p_ = obs.cpu().numpy(), prob.cpu().numpy()
            ok = ~np.isnan(o).any(axis=1)
            v[mode] = dict(flagged=flagged, poisoned_rows=int((~ok).sum()),
                           obs_abs=float((np.abs(o[ok] - ref[ok]) / np.maximum(1.0, np.abs(ref[ok]))).max()) if ok.any() else 0.0,
                           prob_abs=float(np.abs(p_[ok] - p64[ok]).max()) if ok.any() else 0.0,
                           prob_rel=float((np.abs(p_[ok] - p64[ok]) / np.maximum(p64[ok], 1e-30)).max()) if ok.any() else 0.0,
                           ref_h2_max_poisoned=float(np.abs(allf[~ok, :512]).max()) if (~ok).any() else 0.0)
            net.close()
        out['variants'][vname] = v
        if verbose:
            print('%s: raw attention scores in [%.2f, %.2f], max |h2| %.2f' % (vname, v['score_min'], v['score_max'], v['h2_max']))
            for mode in ('fp32', 'fp16x2'):
                print('   %-6s vs fp64: obs max err %.2e (abs, relative above 1), prob max abs err %.2e, prob max rel err %.2e%s'
                      % (mode, v[mode]['obs_abs'], v[mode]['prob_abs'], v[mode]['prob_rel'],
                         '' if not v[mode]['poisoned_rows'] else '; %d of %d rows left the fp16 range -> NaN-poisoned + status bit (fp64 |h2| of those rows up to %.3g)'
                         % (v[mode]['poisoned_rows'], R, v[mode]['ref_h2_max_poisoned'])))
    if verbose:
        print('(%d Adam steps of rl4rs_dientrain_step on synthetic logs, BCE %.4f -> %.4f)' % (out['steps'], out['loss_first'], out['loss_last']))
    return out


if __name__ == '__main__':
    trained_like()
