#!/usr/bin/env python
"""bench.py — env-steps/s of the batched SlateRecEnv env.step() hot path on MI355X.

Contract: ``python bench.py --gpus N --steps K --warmup W`` (N>1: launched by torch.distributed.run, one
rank per GPU).  A "step" is ONE episode-batch of the workload BASELINE.json's metric is quoted on
(configs[1]: SlateRecEnv-v0, batch 4096, 284-item catalogue, 9-slot slate, DIEN simulator scorer):
``env.reset()`` + 9 x ``env.step(offline_action)`` including the reward forward, i.e. B*T = 36 864
env-steps.  Inputs (parsed log + catalogue + weights) are resident in HBM before the timed region.
Ranks run independent env batches (weak scaling, no collective on the data path); rank 0 prints ONE
JSON line with the whole-job env-steps/s, the roofline of the dominant kernel (the AUGRU recurrence,
MFMA-bound, timed live with HIP events on the launch stream) and a CPU baseline (the numpy oracle port
timed on this box's host cores, rank 0 / N=1 only).
"""
import argparse
import json
import os
import sys
import tempfile
import time

REPO = os.path.dirname(os.path.abspath(__file__))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

MFMA_F32_PEAK_TFLOPS = 157.3        # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense
# HBM bytes per AUGRU launch of the default workload, from the PMC passes (None until measured for a mode)
TRAFFIC_B_PER_LAUNCH = {'fp32': 8.96e8, 'fp16x2': 7.34e8}
MFMA_F16_PEAK_TFLOPS = 2500.0       # MI355X_MICROARCH.md: v_mfma_f32_32x32x16_f16, dense (no sparsity)
HBM_PEAK_GBS = 8000.0               # MI355X_MICROARCH.md: HBM3E spec


def make_config(args, workdir, rank):
    from rl4rs_amd import synth
    cat_path = os.path.join(workdir, 'item_info.csv')
    log_path = os.path.join(workdir, 'log_rank%d.csv' % rank)
    cat_text = synth.make_catalog_text(seed=1234)
    synth.write_text(cat_path, cat_text)
    seq = args.env == 'seq'
    records = synth.make_records(args.log_records, pages=4 if seq else 1, seed=1000 + rank, illegal_frac=0.05,   # rdist.shard_seed(1000, rank)
                                 special_ids=synth.special_ids_from_text(cat_text))
    synth.write_records(log_path, records)
    cfg = {"maxlen": 64, "batch_size": args.batch, "action_size": 284, "class_num": 2, "dense_feature_num": 432,
           "category_feature_num": 21, "category_hash_size": 100000, "seq_num": 2, "emb_size": 128,
           "page_items": 9, "hidden_units": 128, "max_steps": args.horizon, "action_emb_size": 32,
           "sample_file": log_path, "iteminfo_file": cat_path, "is_eval": False, "cache_size": 2048,
           "model_seed": 7, "return_tensors": True, "scorer_precision": args.scorer,
           "algo": getattr(args, 'algo', 'dien')}
    if getattr(args, 'conti', False):
        cfg["support_conti_env"] = True       # configs[4]: continuous 32-d actions resolved by the masked K-NN
    return cfg, records


def build_env(cfg, seq):
    import rl4rs_amd
    if seq:
        from rl4rs_amd.env.seqslate import SeqSlateRecEnv, SeqSlateState
        return rl4rs_amd.make('SeqSlateRecEnv-v0', recsim=SeqSlateRecEnv(cfg, state_cls=SeqSlateState))
    from rl4rs_amd.env.slate import SlateRecEnv, SlateState
    return rl4rs_amd.make('SlateRecEnv-v0', recsim=SlateRecEnv(cfg, state_cls=SlateState))


def episode(env, T):
    """reset + T steps of offline_action replay (the reference's canonical loop, simulator_eval.py:34-48)."""
    env.reset()
    total = None
    for _ in range(T):
        a = env.offline_action
        obs, reward, done, info = env.step(a)
        total = reward if total is None else total + reward
    return obs, total


def cpu_baseline(cfg, records, seq, sample_batch):
    """The oracle port (vectorised numpy state machine + float32 numpy DIEN) on the host cores."""
    import numpy as np
    from rl4rs_amd.nets.dien import init_dien_weights
    from oracle.dien import OracleDien
    from oracle.env import OracleEnv
    c = dict(cfg, batch_size=sample_batch)
    algo = c.get('algo', 'dien')
    if algo == 'dien':
        w = init_dien_weights(c, seed=c.get('model_seed', 7))
        scorer = OracleDien(w, c, np.float32)
    else:
        from rl4rs_amd.nets.simnets import init_simnet_weights
        from oracle.simnets import OracleSimnet
        scorer = OracleSimnet(algo, init_simnet_weights(c, algo, seed=c.get('model_seed', 7)), c, np.float32)
    env = OracleEnv(c, records[:sample_batch], scorer, seq=seq)
    T = c['max_steps']
    t0 = time.time()
    env.reset()
    for _ in range(T):
        env.step(np.asarray(env.samples.offline_action))
    dt = time.time() - t0
    try:
        from threadpoolctl import threadpool_info
        cores = max([p.get('num_threads', 1) for p in threadpool_info()] or [1])
    except Exception:
        cores = os.cpu_count() or 1
    return {"value": sample_batch * T / dt, "unit": "env-steps/s", "cores": int(cores), "kind": "port",
            "sample": "1 episode-batch (reset + %d steps incl. reward forward) of %d envs, numpy oracle "
                      "(float32 %s scorer), %.1f s" % (T, sample_batch, algo.upper() if algo == 'dien' else algo, dt)}


def fp32_leg(args, cfg, seq, rank, steps=3):
    """The same workload with the exact-operand fp32 MFMA recurrence (scorer_precision='fp32'), reported next to the
    default so that both arithmetic forms are on the line: value, ms per episode-batch, AUGRU roofline vs the fp32 peak."""
    import torch
    cfg32 = dict(cfg, scorer_precision='fp32')
    env = build_env(cfg32, seq)
    env.seed(1000 + rank)
    env.sim._recData.store.preload(torch.device('cuda', torch.cuda.current_device()))
    B, T = cfg['batch_size'], cfg['max_steps']
    episode(env, T)
    net = env.sim.model.device_net
    net.set_profiling(True)
    net.profile_reset()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        episode(env, T)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    ms, launches = net.profile()[net.augru_kernel]
    n_complete = T if not seq else cfg['page_items']
    reward_calls = 1 if not seq else T // cfg['page_items']
    rows = (T + 1) * B + reward_calls * (n_complete - 1) * B
    flops = steps * rows * cfg['seq_num'] * cfg['maxlen'] * (2 * cfg['emb_size']) * (6 * cfg['emb_size']) * 2
    tf = flops / (ms * 1e-3) / 1e12 if ms > 0 else 0.0
    return {"value": B * T * steps / dt, "unit": "env-steps/s", "ms_per_step": dt / steps * 1e3, "steps": steps,
            "dtype": "f32", "roofline": {"bound": "mfma", "kernel": net.augru_kernel, "achieved": tf,
                                         "peak": MFMA_F32_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": tf / MFMA_F32_PEAK_TFLOPS}}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=10)
    ap.add_argument('--warmup', type=int, default=2)
    ap.add_argument('--batch', type=int, default=4096)
    ap.add_argument('--env', choices=['slate', 'seq'], default='slate')
    ap.add_argument('--horizon', type=int, default=None)
    ap.add_argument('--log-records', type=int, default=8193)
    ap.add_argument('--cpu-batch', type=int, default=256)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-fp32-leg', action='store_true',
                    help='skip the short extra run of the same workload with the exact-fp32 recurrence (N=1, default mode only)')
    ap.add_argument('--conti', action='store_true',
                    help="continuous-action env (support_conti_env): actions are 32-d embeddings resolved by the masked K-NN")
    ap.add_argument('--algo', choices=['dien', 'dnn', 'widedeep', 'lstm'], default='dien',
                    help="simulator family (config['algo']); the headline metric is quoted on dien")
    ap.add_argument('--scorer', choices=['auto', 'fp32', 'fp16x2'], default='auto',
                    help='arithmetic of the AUGRU recurrence (config scorer_precision); auto = fp16x2 operand split, '
                         'fp32 accumulate, same measured error as the exact fp32 MFMA kernel')
    ap.add_argument('--train', choices=['none', 'a2c', 'ppo'], default='none',
                    help='none: offline_action replay (BASELINE configs[1]); a2c/ppo: policy rollout + update with the '
                         'flat-gradient all-reduce over RCCL (configs[2]/[3])')
    args = ap.parse_args()
    if args.horizon is None:
        args.horizon = 9 if args.env == 'slate' else 32
    seq = args.env == 'seq'

    import torch
    import torch.distributed as dist
    from rl4rs_amd import dist as rdist
    rank, local_rank, world = rdist.dist_env()
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs a GPU: the hot path has no CPU fallback')
    torch.cuda.set_device(local_rank)
    rdist.init('nccl')

    workdir = tempfile.mkdtemp(prefix='rl4rs_bench_')
    cfg, records = make_config(args, workdir, rank)     # log shard / RNG stream of this rank: seed 1000 + rank
    env = build_env(cfg, seq)
    env.seed(1000 + rank)
    # inputs resident in HBM before the timed region: parse the whole log once
    env.sim._recData.store.preload(torch.device('cuda', local_rank))
    B, T = args.batch, args.horizon
    trainer = None
    if args.train != 'none':
        from rl4rs_amd.train import Trainer
        trainer = Trainer(env, algo=args.train.upper(), seed=1000 + rank)
    run_step = (lambda: trainer.train_iteration()) if trainer else (lambda: episode(env, T))
    for _ in range(args.warmup):
        run_step()
    net = env.sim.model.device_net
    net.set_profiling(not os.environ.get('RL4RS_BENCH_NOPROF'))
    net.profile_reset()

    rdist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        run_step()
    rdist.barrier()
    elapsed = rdist.max_over_ranks(time.perf_counter() - t0, device='cuda')
    prof = net.profile()
    net.set_profiling(False)

    if rank == 0:
        env_steps = world * B * T * args.steps
        is_dien = cfg.get('algo', 'dien') == 'dien'
        roofline = None
        if is_dien:
            # dominant kernel: AUGRU recurrence.  Executed (= algorithmic after the exact input-projection hoist)
            # FLOPs per row per sequence input: L steps x (2E x 6E) MACs x 2.
            L, E = cfg['maxlen'], cfg['emb_size']
            flop_row_seq = L * (2 * E) * (6 * E) * 2
            kname = net.augru_kernel
            ms, launches = prof[kname]
            n_complete = T if not seq else cfg['page_items']
            reward_calls = 1 if not seq else T // cfg['page_items']
            # reset obs + T step obs + reward rows (the last reward row of an env reuses the state row just scored)
            rows_per_episode = (T + 1) * B + reward_calls * (n_complete - 1) * B
            flops = args.steps * rows_per_episode * cfg['seq_num'] * flop_row_seq
            achieved = flops / (ms * 1e-3) / 1e12 if ms > 0 else 0.0
            if net.scorer_mode == 'fp16x2':
                # each fp32-class product costs 3 f16 MFMAs (hi*hi + hi*lo + lo*hi): peak in algorithmic FLOPs = f16 dense / 3
                peak, peak_note = MFMA_F16_PEAK_TFLOPS / 3.0, "v_mfma_f32_32x32x16_f16 dense 2500 TF/s / 3 MFMAs per product"
            else:
                peak, peak_note = MFMA_F32_PEAK_TFLOPS, "v_mfma_f32_32x32x2_f32 dense"
            roofline = {"bound": "mfma", "kernel": kname, "achieved": achieved,
                        "peak": peak, "unit": "TFLOP/s", "frac": achieved / peak, "peak_note": peak_note,
                        "traffic": None, "launches": int(launches), "avg_launch_ms": ms / max(launches, 1),
                        "kernel_ms_share": ms / (elapsed * 1e3)}
            if not seq and B == 4096 and T == 9 and not trainer:
                # HBM bytes per launch from the PMC passes of this exact workload (rocprofv3 FETCH_SIZE x 2 + WRITE_SIZE,
                # corrected as MI355X_MICROARCH.md prescribes): launch-weighted mean of the 10 obs-sized and the 1
                # reward-sized launch of an episode (fp32: 897 / 883 MB, fp16x2: 721 / 863 MB)
                roofline["traffic"] = TRAFFIC_B_PER_LAUNCH[net.scorer_mode]
                roofline["traffic_unit"] = "B/launch (PMC: profiles/r01c_pmc.md fp32, profiles/r01f_pmc.md / r01g_pmc.md fp16x2)"
        kernels = dict((k, {"ms": round(v[0], 3), "launches": int(v[1])}) for k, v in prof.items())
        # the HBM-bound gather kernel in isolation (complete-state rows, 9B rows x 2016 algorithmic bytes)
        samples = env.samples
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 50
        samples._env.build_complete()
        ev0.record()
        for _ in range(reps):
            samples._env.build_complete()
        ev1.record()
        torch.cuda.synchronize()
        g_ms = ev0.elapsed_time(ev1) / reps
        g_rows = B * samples._env.n_complete
        g_bytes = g_rows * (cfg['dense_feature_num'] * 4 + cfg['category_feature_num'] * 4 + 32 * 4 + 10 * 4 + 36)
        g_gbs = g_bytes / (g_ms * 1e-3) / 1e9
        gather = {"bound": "hbm", "kernel": "k_env_rows<complete>", "achieved": g_gbs, "peak": HBM_PEAK_GBS,
                  "unit": "GB/s", "frac": g_gbs / HBM_PEAK_GBS, "traffic": None, "avg_launch_ms": g_ms,
                  "rows": g_rows}
        out = {
            "metric": "env-steps/s (batch=%d, %d-slot slate)" % (B, 9),
            "value": env_steps / elapsed,
            "unit": "env-steps/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32" if (not is_dien or net.scorer_mode == 'fp32') else "f32 (AUGRU operands as fp16 hi+lo pairs, fp32 accumulate)",
            "data": "synthetic",
            "config": {"workload": "%s batch=%d per GPU, 284-item catalogue, 9-slot slate, %d-step horizon, "
                                   "%s simulator scorer, offline_action replay"
                                   % ('SeqSlateRecEnv-v0' if seq else 'SlateRecEnv-v0', B, T, cfg.get('algo', 'dien').upper() if is_dien else cfg['algo']),
                       "step": "one episode-batch = reset + %d env.step incl. reward forward" % T +
                               ("" if not trainer else " with %s policy sampling + update (gradient all-reduce)" % args.train.upper()) +
                               ("" if not args.conti else "; continuous actions -> masked K-NN over the catalogue"),
                       "parallelism": "independent env batches per GPU (no data-path collective)"},
            # dien: the AUGRU recurrence; the GEMM-only families (dnn / widedeep / lstm) have no dominant matrix kernel,
            # their line carries the HBM roofline of the feature-gather kernel
            "roofline": roofline if roofline is not None else gather,
            "roofline_gather": gather,
            "kernels": kernels,
        }
        if (world == 1 and is_dien and not trainer and net.scorer_mode == 'fp16x2' and not args.no_fp32_leg):
            out["exact_fp32_scorer"] = fp32_leg(args, cfg, seq, rank)
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(cfg, records, seq, args.cpu_batch)
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
