#!/usr/bin/env python
"""bench.py — env-steps/s of the batched SlateRecEnv env.step() hot path on MI355X.

Contract: ``python bench.py --gpus N --steps K --warmup W`` (N>1: one rank per GPU - launched by
torch.distributed.run, or, when started plainly without WORLD_SIZE in the environment, bench.py starts its own
N ranks through torch.distributed.run; WORLD_SIZE != --gpus or fewer visible GPUs than ranks is an error).  A "step" is ONE episode-batch of the workload BASELINE.json's metric is quoted on
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
# HBM bytes per AUGRU launch of the default workload, from the round's PMC passes (profiles/r06zzd_pmc.md, arithmetic in its header).
# fp16x2 (k_augru_x): raw FETCH_SIZE 142.7 MB per obs-sized / 214.3 MB per reward-sized launch (the obs-sized form reads a row's front
# padding from ONE shared cache slot since round 6: 185.0 MB before), x the factor calibrated on this kernel's own LDS-DMA stream
# against a known byte count (profiles/r02_fetch_calibration.md: 1.71 for the 32-row form, 2.00 for the 64-row form), + WRITE_SIZE
# 8.2 / 65.5 MB: (10 x 252.2 + 494.1) / 11.  fp32 (k_recur): not re-measured since round 1.
TRAFFIC_SOURCE_FILE = 'profiles/r06zzd_pmc.md'
TRAFFIC_B_PER_LAUNCH = {'fp32': None, 'fp16x2': 2.74e8}      # fp32 kernel: not re-measured since the row-order hint (r01e: 8.96e8)
TRAFFIC_NOTE = ("B/launch, launch-weighted over the 10 obs-sized (252.2 MB) + 1 reward-sized (494.1 MB) launches of an episode-batch; "
                "rocprofv3 FETCH_SIZE x the factor calibrated on this kernel's stream (profiles/r02_fetch_calibration.md) + WRITE_SIZE, "
                "arithmetic in the header of profiles/r06zzd_pmc.md; algorithmic bytes of an obs-sized launch = 1746 distinct histories x "
                "~48 non-padding steps x 768 f32 = 259 MB read (the front padding of all rows comes from one shared slot; 343 MB with "
                "every row's own padding) + 8.4 MB written: with the row-order hint the duplicate env rows of one history hit in L2")
GATHER_TRAFFIC_B = 6.48e7           # k_env_rows<2>: WRITE_SIZE 64.15 MB + FETCH_SIZE 0.65 MB per launch (profiles/r04f_pmc.md; r05q_pmc.md, r06p_pmc.md, r06zzd_pmc.md: unchanged)
MFMA_F16_PEAK_TFLOPS = 2500.0       # MI355X_MICROARCH.md: v_mfma_f32_32x32x16_f16, dense (no sparsity)
HBM_PEAK_GBS = 8000.0               # MI355X_MICROARCH.md: HBM3E spec


class SclkSampler(object):
    """Shader clock while the timed region runs, sampled from sysfs (`pp_dpm_sclk`: the active DPM level is starred) every 10 ms
    by a side thread: MI355X lowers its clock under the MFMA load of the recurrence (1.8-1.9 GHz in the PMC passes against the
    2.4 GHz the peak is quoted at), so `roofline.frac` is also reported against the measured clock.  Only the card whose PCI address
    is this process' HIP device is sampled, and only a plausible mean (0.5 - 3 GHz) is used; None otherwise - then the PMC figure
    under profiles/ (GRBM_GUI_ACTIVE / duration) is the reference."""

    def __init__(self, device_index=0):
        import glob
        import threading
        self.paths = []
        try:
            # the card of THIS process' HIP device, by PCI address (a box shows every GPU of the node in sysfs, most of them
            # other tenants')
            import torch
            pr = torch.cuda.get_device_properties(device_index)
            addr = '%04x:%02x:%02x.0' % (pr.pci_domain_id, pr.pci_bus_id, pr.pci_device_id)
            for p in sorted(glob.glob('/sys/class/drm/card*/device/pp_dpm_sclk')):
                if os.path.basename(os.path.realpath(os.path.dirname(p))) == addr:
                    self.paths = [p]
        except Exception:
            self.paths = []
        self.samples = dict((p, []) for p in self.paths)
        self._stop = threading.Event()
        self._thread = threading.Thread(target=self._run, daemon=True) if self.paths else None

    @staticmethod
    def _read(path):
        try:
            for line in open(path).read().splitlines():
                if line.strip().endswith('*'):
                    return float(line.split(':')[1].strip().lower().replace('mhz', '').replace('*', '').strip())
        except Exception:
            return None
        return None

    def _run(self):
        while not self._stop.is_set():
            for p in self.paths:
                v = self._read(p)
                if v:
                    self.samples[p].append(v)
            self._stop.wait(0.01)

    def __enter__(self):
        if self._thread:
            self._thread.start()
        return self

    def __exit__(self, *a):
        self._stop.set()
        if self._thread:
            self._thread.join(timeout=1.0)

    def mean_mhz(self):
        means = [sum(v) / len(v) for v in self.samples.values() if v]
        best = max(means) if means else None
        return best if best is not None and 500.0 <= best <= 3000.0 else None

    def n_samples(self):
        return max([len(v) for v in self.samples.values()] or [0])


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
    if os.environ.get('RL4RS_NO_ORDER'):
        cfg['no_row_order'] = True            # A/B: without the slot-sorted processing order of the scorer
    if getattr(args, 'conti', False):
        cfg["support_conti_env"] = True       # configs[4]: continuous 32-d actions resolved by the masked K-NN
    if getattr(args, 'train', 'none') == 'bcq':
        # configs[4]: the continuous env in the d3rlpy observation mode (obs | previous actions | cur_step, batchrl_train.py:25-27)
        cfg["support_conti_env"] = True
        cfg["support_d3rl_mask"] = True
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


CPU_ROW_WORKERS = 16      # measured on the bench box at R = 4096: 437 GFLOP/s with 16 row-parallel workers, 234 with 16 intra-op
                          # threads, 60-110 with 64-256 of either (python GIL / thread-pool overhead of the small per-step ops)


def _blas_threads():
    try:
        from threadpoolctl import threadpool_info
        return max([p.get('num_threads', 1) for p in threadpool_info()] or [1])
    except Exception:
        return os.cpu_count() or 1


def cpu_baseline(cfg, records, seq, sample_batch, faithful_batch=64):
    """SURVEY.md 8(d) CPU baselines, both on a bounded sample of the same workload (rank 0, N = 1 only):

    value            "vectorised": the numpy oracle state machine (oracle/state.py: whole-batch array ops) + a torch-CPU
                     float32 DIEN (oracle/dien_torch.py) on ALL host cores, one episode-batch of ``sample_batch`` envs
    faithful_1core   the reference's own control flow (oracle/faithful.py: one python iteration per sample, nested-list
                     state, per-row padding) + the numpy float32 DIEN, pinned to ONE core, one episode-batch of
                     ``faithful_batch`` envs
    The reference itself (TF1 / deepctr) cannot run here or on the GPU box; kind stays "port"."""
    import numpy as np
    import torch
    from rl4rs_amd.nets.dien import init_dien_weights
    from oracle.dien import OracleDien
    from oracle.env import OracleEnv
    algo = cfg.get('algo', 'dien')
    T = cfg['max_steps']

    def scorer_for(c, kind):
        if algo == 'dien':
            w = init_dien_weights(c, seed=c.get('model_seed', 7))
            if kind == 'torch':
                from oracle.dien_torch import TorchDien
                return TorchDien(w, c, workers=CPU_ROW_WORKERS)
            return OracleDien(w, c, np.float32)
        from rl4rs_amd.nets.simnets import init_simnet_weights
        from oracle.simnets import OracleSimnet
        return OracleSimnet(algo, init_simnet_weights(c, algo, seed=c.get('model_seed', 7)), c, np.float32)

    # ---- vectorised, all cores
    c = dict(cfg, batch_size=sample_batch)
    env = OracleEnv(c, records[:sample_batch], scorer_for(c, 'torch'), seq=seq)
    t0 = time.time()
    env.reset()
    for _ in range(T):
        env.step(np.asarray(env.samples.offline_action))
    dt = time.time() - t0
    threads = CPU_ROW_WORKERS if algo == 'dien' else int(torch.get_num_threads())
    threaded = {"value": sample_batch * T / dt, "unit": "env-steps/s", "cores": threads, "kind": "port",
                "sample": "1 episode-batch (reset + %d steps incl. reward forward) of %d envs: vectorised numpy state machine + "
                          "torch-CPU float32 %s on %d row-parallel worker THREADS of one process (the rate of one process peaks "
                          "there on this host: the 64-step recurrences are chains of small matmuls), %.1f s"
                          % (T, sample_batch, algo.upper() if algo == 'dien' else algo, threads, dt)}
    # ---- the same port on the WHOLE machine: env rows are independent, so row blocks go to single-threaded worker processes,
    # one per physical core (oracle/cpu_pool.py); timed from "every worker has built its env" to "last worker done"
    # Two geometries of the same 4096-env sample are timed and the faster one is `value` (both are on the line): one worker
    # per PHYSICAL core with 32 rows each, and a quarter of that with 128 rows each - on the round-3 bench box (128 cores /
    # 256 threads) the rate FELL with the worker count: 32 x 128 rows 4.96 k, 64 x 64 3.65 k, 128 x 32 2.72 k, 256 x 16 1.64 k
    # env-steps/s (tools/cpu_pool_sweep.py; the per-step matmuls of a block are small, and 128 of them thrash the shared
    # caches), and more than one thread per worker was slower still.
    from oracle.cpu_pool import run_pool
    logical = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1)
    physical = max(1, logical // 2) if logical >= 32 else logical
    total = 4096 if physical >= 32 else physical * 32
    recs = list(records)
    while len(recs) < total:
        recs += list(records)
    runs = []
    for workers in sorted(set([physical, max(1, physical // 4)])):
        rows_per_worker = total // workers
        pool = run_pool(dict(cfg), recs[:workers * rows_per_worker], seq, workers, rows_per_worker)
        runs.append({"value": pool['env_steps'] / pool['seconds'], "unit": "env-steps/s", "cores": workers, "kind": "port",
                     "sample": "1 episode-batch (reset + %d steps incl. reward forward) of %d envs cut into %d row blocks of %d, one "
                               "single-threaded process per block (vectorised numpy state machine + torch-CPU float32 %s per block; "
                               "%d logical cores, os.cpu_count() = %d), %.1f s wall, slowest worker %.1f s"
                               % (T, workers * rows_per_worker, workers, rows_per_worker, algo.upper() if algo == 'dien' else algo,
                                  logical, os.cpu_count() or 0, pool['seconds'], pool['slowest_worker_s'])})
    out = dict(max(runs, key=lambda r: r['value']))
    out["all_pool_geometries"] = [{"cores": r['cores'], "value": r['value'], "sample": r['sample']} for r in runs]
    out["one_process_%d_threads" % threads] = threaded
    # ---- faithful per-sample loop, one core (Slate only: the bench workload)
    if not seq and faithful_batch > 0:
        from oracle.faithful import FaithfulSlateEnv
        c1 = dict(cfg, batch_size=faithful_batch)
        try:
            from threadpoolctl import threadpool_limits
            limit = threadpool_limits(limits=1)
        except Exception:
            limit = None
        nt = torch.get_num_threads()
        torch.set_num_threads(1)
        try:
            fenv = FaithfulSlateEnv(c1, records[:faithful_batch], scorer_for(c1, 'numpy'))
            t0 = time.time()
            fenv.reset()
            for _ in range(T):
                fenv.step(fenv.offline_action)
            fdt = time.time() - t0
            # the same loop without the net (what BASELINE.md section 2 measured for the reference itself: 1.9-3.0 k env-steps/s)
            class _NoNet(object):
                def obs(self, seq_, dense, cat):
                    return np.zeros((len(cat), 256), dtype=np.float32)

                def prob(self, seq_, dense, cat):
                    return np.full((len(cat),), 0.5, dtype=np.float32)
            fenv2 = FaithfulSlateEnv(c1, records[:faithful_batch], _NoNet())
            t0 = time.time()
            fenv2.reset()
            for _ in range(T):
                fenv2.step(fenv2.offline_action)
            fdt2 = time.time() - t0
        finally:
            torch.set_num_threads(nt)
            if limit is not None:
                limit.restore_original_limits()
        out["faithful_1core"] = {"value": faithful_batch * T / fdt, "unit": "env-steps/s", "cores": 1, "kind": "port",
                                 "without_net": faithful_batch * T / fdt2,
                                 "sample": "1 episode-batch of %d envs: per-sample python loop (oracle/faithful.py) + numpy float32 "
                                           "DIEN, 1 thread, %.1f s (%.2f s without the net)" % (faithful_batch, fdt, fdt2)}
    return out


class BcqWorkload(object):
    """BASELINE configs[4] on one rank: the continuous-action env + K-NN with the BCQ learner of 'BCQ-conti'
    (script/batchrl_trainer.py:61-73).  Set-up (untimed): the rank's logged-policy dataset is generated on the device
    (data_generate_rl4rs_a_conti, :220-270).  One step = ``updates`` learner updates of 256 transitions (data parallel: the
    flat gradients of each phase all-reduced over the ranks) + one episode-batch of the learned policy driving the env:
    ``predict`` (100 sampled actions per env, the first critic's pick) -> ``env.step`` (masked K-NN over the catalogue,
    scorer, reward) - the `evaluate` loop of batchrl_trainer.py:377-411."""

    def __init__(self, env, cfg, rank, updates=16, epochs=2):
        import torch
        from rl4rs_amd.offline import generate_offline_dataset
        from rl4rs_amd.offline_rl import BCQ, transitions_from_mdp
        self.env, self.cfg, self.updates = env, cfg, int(updates)
        data = generate_offline_dataset(env, epochs=epochs, shuffle=False)
        self.tr = transitions_from_mdp(data['observations'], data['actions'], data['rewards'], data['terminals'], discrete_action=False)
        # predict_rows = the env batch: one pass of 100 x B sampled rows per step (h1 / h2 scratch: 3 x B x 100 x 256 floats per network)
        self.bcq = BCQ(cfg, self.tr[0].shape[1], batch_size=256, seed=7, predict_rows=cfg['batch_size'])       # the same initial replica on every rank
        self.bcq._gen.manual_seed(1000 + rank)                                  # its own noise / minibatch stream
        self.rank = rank
        self.calls = 0
        self.last_losses = None

    def step(self):
        T = self.cfg['max_steps']
        self.last_losses = self.bcq.fit(self.tr, self.updates, shuffle_seed=1000 + self.rank + 7919 * self.calls, to_host=False)
        self.calls += 1
        obs = self.env.reset()
        for _ in range(T):
            obs, reward, done, info = self.env.step(self.bcq.predict(obs))
        return reward


def episode_host(env, T):
    """The same loop through the reference-shaped API (config without return_tensors): lists / ndarrays / dicts come back to
    the host every step (obs [B, 256] float32, the int64 action mask in rllib-mask mode, rewards as a python list), and the
    logged actions go in as the python list the reference's offline_action hands out."""
    env.reset()
    for _ in range(T):
        obs, reward, done, info = env.step(env.offline_action)
    return obs, reward


def timed_episodes(env, T, steps, warmup=1, fn=None):
    import torch
    fn = fn or episode
    for _ in range(warmup):
        fn(env, T)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn(env, T)
    torch.cuda.synchronize()
    return time.perf_counter() - t0


def extra_leg(args, workdir, rank, name, steps=3):
    """One more workload timed the same way (K episode-batches between two synchronisations, inputs resident), N = 1 only:
    the BASELINE.json configs the headline does not cover and the variants the verdict asked to disclose."""
    import copy
    import torch
    a = copy.copy(args)
    a.conti, a.env, a.horizon = False, 'slate', 9
    train = None
    extra_cfg = {}
    if name == 'seq_t32':
        a.env, a.horizon = 'seq', 32
    elif name == 'seq_t32_ppo':
        a.env, a.horizon, train = 'seq', 32, 'PPO'
    elif name == 'seq_t32_a2c':
        a.env, a.horizon, train = 'seq', 32, 'A2C'            # configs[3]'s per-GPU shard (modelfree_train.py:42-44,248-304)
    elif name == 'bcq_conti':
        a.train = 'bcq'
    elif name == 'conti':
        a.conti = True
    elif name == 'all_distinct':
        extra_cfg = {'is_eval': True, 'cache_size': a.batch}           # eval mode: the first B lines, no duplicate histories
    elif name == 'fp32':
        a.scorer = 'fp32'
    elif name == 'compat_numpy':
        extra_cfg = {'return_tensors': False}
    elif name == 'compat_rllib_mask':
        extra_cfg = {'return_tensors': False, 'support_rllib_mask': True}
    elif name == 'compat_d3rl_mask':
        extra_cfg = {'return_tensors': False, 'support_d3rl_mask': True}
    compat = name.startswith('compat_')
    cfg, _ = make_config(a, workdir, rank)
    cfg.update(extra_cfg)
    seq = a.env == 'seq'
    env = build_env(cfg, seq)
    env.seed(1000 + rank)
    env.sim._recData.store.preload(torch.device('cuda', torch.cuda.current_device()))
    B, T = cfg['batch_size'], cfg['max_steps']
    if name == 'bcq_conti':
        return bcq_leg(env, cfg, rank)
    if train:
        from rl4rs_amd.train import Trainer
        tr = Trainer(env, algo=train, seed=1000 + rank)
        tr.train_iteration()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            tr.train_iteration()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        tr.close()
    else:
        if compat:
            # host-bound legs settle over the first episodes (page-locked blocks of the caching host allocator, python's allocator): the
            # metric is a steady-state mean (after 3 warm-ups), so these legs warm up 3 and time 10 episode-batches
            steps = max(steps, 10)
            dt = timed_episodes(env, T, steps, warmup=3, fn=episode_host)
        else:
            dt = timed_episodes(env, T, steps)
    out = {"value": B * T * steps / dt, "unit": "env-steps/s", "ms_per_step": dt / steps * 1e3, "steps": steps,
           "workload": "%s B=%d T=%d%s%s" % ('SeqSlateRecEnv-v0' if seq else 'SlateRecEnv-v0', B, T,
                                             ', continuous actions -> masked K-NN' if a.conti else '',
                                             ', %s rollout + update (configs[%d])' % (train, 2 if train == 'PPO' else 3) if train
                                             else ', offline_action replay')}
    if compat:
        out["workload"] += (" through the REFERENCE-SHAPED API (no return_tensors%s): one rl4rs_env_step_record call + one pinned "
                            "device-to-host copy per step; PCIe-inclusive, python list / dict construction included"
                            % ({'compat_numpy': '', 'compat_rllib_mask': ', support_rllib_mask: list of B {action_mask int64[284], obs} dicts',
                                'compat_d3rl_mask': ', support_d3rl_mask: float64 [B, 266] observations'}[name]))
    hu = getattr(env.samples, '_hist_unique', None)
    out["distinct_histories"] = int(hu[0].shape[0]) if hu is not None else B
    if name == 'fp32':
        net = env.sim.model.device_net
        net.set_profiling(2)
        net.profile_reset()
        episode(env, T)
        ms, launches = net.profile()[net.augru_kernel]
        net.set_profiling(0)
        rows = (T + 1) * B + (T - 1) * B
        tf = rows * cfg['seq_num'] * cfg['maxlen'] * (2 * cfg['emb_size']) * (6 * cfg['emb_size']) * 2 / (ms * 1e-3) / 1e12
        out["dtype"] = "f32"
        out["roofline"] = {"bound": "mfma", "kernel": net.augru_kernel, "achieved": tf, "peak": MFMA_F32_PEAK_TFLOPS,
                           "unit": "TFLOP/s", "frac": tf / MFMA_F32_PEAK_TFLOPS}
    return out


def bcq_leg(env, cfg, rank, update_steps=200, rollouts=3):
    """configs[4] on one GPU, its two halves timed separately: learner updates/s (256 transitions each, 100 sampled actions per
    target row) and the policy -> K-NN rollout's env-steps/s."""
    import torch
    wl = BcqWorkload(env, cfg, rank)
    B, T = cfg['batch_size'], cfg['max_steps']
    wl.bcq.fit(wl.tr, 10, to_host=False)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    hist = wl.bcq.fit(wl.tr, update_steps, to_host=False)
    torch.cuda.synchronize()
    dt_u = time.perf_counter() - t0
    wl.updates = 0
    wl.step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(rollouts):
        wl.step()
    torch.cuda.synchronize()
    dt_r = time.perf_counter() - t0
    out = {"value": B * T * rollouts / dt_r, "unit": "env-steps/s", "ms_per_step": dt_r / rollouts * 1e3, "steps": rollouts,
           "workload": "SlateRecEnv-v0 B=%d T=%d, continuous actions: BCQ policy predict (100 sampled 32-d actions per env, first critic's "
                       "pick) -> masked K-NN -> scorer (configs[4] evaluation rollout)" % (B, T),
           "learner": {"updates_per_s": update_steps / dt_u, "transitions_per_s": update_steps * 256 / dt_u, "ms_per_update": dt_u / update_steps * 1e3,
                       "updates": update_steps,
                       "what": "d3rlpy.algos.BCQ(batch_size=256) restated on the device: conditional-VAE imitator, residual actor, twin "
                               "critics, lam-weighted target over 100 sampled actions per row (25 600-row forward through 4 networks), soft "
                               "target updates; dataset = %d transitions generated on the device" % wl.tr[0].shape[0],
                       "dtype": "f32; the no-grad forwards (25 600 target rows per update, 409 600 sampled rows per rollout step) in %s"
                                % ("fp16x2 (fp16 hi + lo operands, three f16 MFMAs per product, fp32 accumulate: k_amlp_fwd_h16)"
                                   if wl.bcq.nograd == 'fp16x2' else "f32"),
                       "last_losses": dict((k, float(v[-1])) for k, v in hist.items() if len(v))}}
    # the other continuous learner the reference offers for this config ('CQL-conti', batchrl_trainer.py:91-107): updates/s only
    from rl4rs_amd.offline_rl import CQL, StandardRewardScaler
    cql = CQL(cfg, wl.tr[0].shape[1], batch_size=256, gamma=1.0, reward_scaler=StandardRewardScaler(wl.tr[2]), seed=7)
    cql.fit(wl.tr, 10, to_host=False)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    cql.fit(wl.tr, update_steps, to_host=False)
    torch.cuda.synchronize()
    dt_c = time.perf_counter() - t0
    out["learner_cql"] = {"updates_per_s": update_steps / dt_c, "transitions_per_s": update_steps * 256 / dt_c, "ms_per_update": dt_c / update_steps * 1e3,
                          "what": "d3rlpy.algos.CQL(batch_size=256, gamma=1, reward_scaler='standard') restated on the device: squashed-Gaussian "
                                  "actor, learned temperature and alpha, twin critics with the conservative term over 31 rows per transition"}
    cql.close()
    wl.bcq.close()
    return out


def _free_port():
    import socket
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    return port


def launch_ranks(n_gpus, argv):
    """``python bench.py --gpus N`` with N > 1 and no torch.distributed.run environment: start the N ranks ourselves (the command
    the contract names: ``python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P
    bench.py <the same arguments>``), let rank 0's one JSON line through on stdout and hand back the launcher's exit code - a
    plain ``--gpus 8`` can then never run as one silent rank."""
    import subprocess
    env = dict(os.environ, MASTER_ADDR='127.0.0.1')
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(n_gpus), '--master-addr', '127.0.0.1',
           '--master-port', str(_free_port()), os.path.abspath(__file__)] + list(argv)
    return subprocess.call(cmd, env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=10)
    ap.add_argument('--warmup', type=int, default=2)
    ap.add_argument('--batch', type=int, default=4096)
    ap.add_argument('--env', choices=['slate', 'seq'], default='slate')
    ap.add_argument('--horizon', type=int, default=None)
    ap.add_argument('--log-records', type=int, default=8193)
    ap.add_argument('--cpu-batch', type=int, default=2048,
                    help='envs of the vectorised cpu_baseline sample (one episode-batch; the faithful 1-core leg uses 64)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-fp32-leg', action='store_true',
                    help='skip the short extra run of the same workload with the exact-fp32 recurrence (N=1, default mode only)')
    ap.add_argument('--no-extra-legs', action='store_true',
                    help='skip the extra workloads of the default N=1 run (seq_t32, seq_t32_ppo, conti, all_distinct)')
    ap.add_argument('--conti', action='store_true',
                    help="continuous-action env (support_conti_env): actions are 32-d embeddings resolved by the masked K-NN")
    ap.add_argument('--algo', choices=['dien', 'dnn', 'widedeep', 'lstm'], default='dien',
                    help="simulator family (config['algo']); the headline metric is quoted on dien")
    ap.add_argument('--scorer', choices=['auto', 'fp32', 'fp16x2'], default='auto',
                    help='arithmetic of the AUGRU recurrence (config scorer_precision); auto = fp16x2 operand split, '
                         'fp32 accumulate, same measured error as the exact fp32 MFMA kernel')
    ap.add_argument('--train', choices=['none', 'a2c', 'ppo', 'bcq'], default='none',
                    help='none: offline_action replay (BASELINE configs[1]); a2c/ppo: policy rollout + update with the '
                         'flat-gradient all-reduce over RCCL (configs[2]/[3]); bcq: continuous-action env, a step = --bcq-updates '
                         'BCQ updates (gradient all-reduce) + one policy -> K-NN episode-batch (configs[4])')
    ap.add_argument('--bcq-updates', type=int, default=16, help='--train bcq: learner updates per step')
    ap.add_argument('--minibatch', type=int, default=256,
                    help='--train ppo: SGD minibatch per rank (RLlib: 256).  With N > 1 every minibatch costs one gradient '
                         'all-reduce (synchronous SGD over N x minibatch samples): 512 / 1024 halve / quarter the collectives '
                         'per pass at the price of a larger effective minibatch (DESIGN.md 6)')
    args = ap.parse_args()
    if args.horizon is None:
        args.horizon = 9 if args.env == 'slate' else 32
    seq = args.env == 'seq'
    if 'WORLD_SIZE' in os.environ and int(os.environ['WORLD_SIZE']) != args.gpus:
        raise SystemExit('bench.py: launched with WORLD_SIZE=%s but --gpus %d: the two must agree (the JSON line reports n_gpus = '
                         'the ranks that actually ran)' % (os.environ['WORLD_SIZE'], args.gpus))

    import torch
    import torch.distributed as dist
    from rl4rs_amd import dist as rdist
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs a GPU: the hot path has no CPU fallback')
    # RL4RS_DIST_BACKEND=gloo: dry run of the N > 1 control flow on a box with fewer GPUs than ranks (ranks share devices);
    # the driver's runs use the default: one rank per GPU over RCCL
    backend = os.environ.get('RL4RS_DIST_BACKEND', 'nccl')
    if args.gpus < 1:
        raise SystemExit('bench.py: --gpus must be >= 1')
    if backend == 'nccl' and torch.cuda.device_count() < args.gpus:
        raise SystemExit('bench.py: --gpus %d over RCCL needs %d visible GPUs, this box shows %d (one rank per GPU; '
                         'RL4RS_DIST_BACKEND=gloo is the dry-run mode in which ranks share devices)'
                         % (args.gpus, args.gpus, torch.cuda.device_count()))
    if 'WORLD_SIZE' not in os.environ:
        if args.gpus > 1:
            sys.exit(launch_ranks(args.gpus, sys.argv[1:]))          # no launcher around us: start the N ranks ourselves
    rank, local_rank, world = rdist.dist_env()
    if backend != 'nccl':
        local_rank = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    rdist.init(backend)

    workdir = tempfile.mkdtemp(prefix='rl4rs_bench_')
    cfg, records = make_config(args, workdir, rank)     # log shard / RNG stream of this rank: seed 1000 + rank
    env = build_env(cfg, seq)
    env.seed(1000 + rank)
    # inputs resident in HBM before the timed region: parse the whole log once
    env.sim._recData.store.preload(torch.device('cuda', local_rank))
    B, T = args.batch, args.horizon
    trainer = None
    if args.train == 'bcq':
        trainer = BcqWorkload(env, cfg, rank, updates=args.bcq_updates)
        run_step = trainer.step
    elif args.train != 'none':
        from rl4rs_amd.train import Trainer
        trainer = Trainer(env, algo=args.train.upper(), seed=1000 + rank, minibatch=args.minibatch)
        run_step = trainer.train_iteration
    else:
        run_step = lambda: episode(env, T)
    for _ in range(args.warmup):
        run_step()
    net = env.sim.model.device_net
    # the timed region carries event pairs around the dominant kernel only (two records per forward); the per-kernel
    # breakdown comes from a separate short pass afterwards (a pair around EVERY kernel costs ~0.4 ms per episode-batch)
    net.set_profiling(0 if os.environ.get('RL4RS_BENCH_NOPROF') else 2)
    net.profile_reset()

    rdist.barrier()
    sclk = SclkSampler(local_rank)
    t0 = time.perf_counter()
    with sclk:
        for _ in range(args.steps):
            run_step()
        torch.cuda.synchronize()
    my_elapsed = time.perf_counter() - t0            # this rank's own clock up to ITS last step (before the closing barrier)
    rdist.barrier()
    elapsed = rdist.max_over_ranks(time.perf_counter() - t0, device='cuda')
    # what the collective layer actually saw: the number of ranks that took part (an all-reduced count - equals --gpus only
    # if every rank joined the same process group) and every rank's own env-steps/s
    ranks_seen = int(round(rdist.sum_over_ranks(1.0, device='cuda')))
    if ranks_seen != args.gpus:
        raise SystemExit('bench.py: %d ranks joined the process group but --gpus is %d: nothing is reported' % (ranks_seen, args.gpus))
    per_rank = rdist.gather_floats(B * T * args.steps / my_elapsed, device='cuda')
    # data-parallel training: every rank must hold the SAME learner after the same all-reduced steps - a position-weighted float64
    # digest of each rank's parameters, gathered (equal digests on all ranks = replicas in step; the first N > 1 run on real
    # hardware cannot be debugged from here, so the line carries the evidence)
    param_digests = None
    if trainer is not None:
        nets = [trainer.policy] if hasattr(trainer, 'policy') else list(trainer.bcq.nets)
        dig = 0.0
        for net_ in nets:
            p_ = (net_.params() if hasattr(net_, 'params') else net_.flat_params()).double().reshape(-1)
            w_ = (torch.arange(p_.numel(), device=p_.device, dtype=torch.float64) % 977.0) + 1.0
            dig += float(torch.dot(p_, w_))
        param_digests = rdist.gather_floats(dig, device='cuda')
    prof = net.profile()
    net.set_profiling(0)
    breakdown, breakdown_steps = None, 3
    # the per-kernel breakdown pass is rank 0's - except that a train step with world > 1 contains collectives (the gradient
    # all-reduce), so then every rank has to take the same steps (only rank 0 records events)
    if args.train == 'bcq' and rank == 0:
        out_bcq = {"updates_per_step": args.bcq_updates, "last_losses": dict((k, float(v[-1])) for k, v in (trainer.last_losses or {}).items() if len(v))}
    if rank == 0 or (trainer is not None and world > 1):
        if rank == 0:
            net.set_profiling(1)
            net.profile_reset()
        for _ in range(breakdown_steps):
            run_step()
        torch.cuda.synchronize()
        if rank == 0:
            breakdown = net.profile()
            net.set_profiling(0)

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
            # the peak is quoted at the 2.4 GHz boost clock; under this kernel the part runs slower (power management)
            mhz = sclk.mean_mhz()
            roofline["sclk_mhz_measured"] = mhz
            roofline["sclk_source"] = ("mean of %d sysfs pp_dpm_sclk samples over the timed region (this device's card, matched by PCI address)" % sclk.n_samples()) if mhz else \
                "no plausible pp_dpm_sclk reading on this box; see the GRBM_GUI_ACTIVE / duration figure in profiles/r05q_pmc.md"
            roofline["frac_at_measured_clock"] = (achieved / (peak * mhz / 2400.0)) if mhz else None
            if not seq and B == 4096 and T == 9 and not trainer:
                # HBM bytes per launch from the PMC passes of this exact workload (rocprofv3 FETCH_SIZE, corrected by the factor
                # CALIBRATED on this kernel's own access pattern, + WRITE_SIZE): launch-weighted mean of the 10 obs-sized
                # and the 1 reward-sized launch of an episode-batch
                roofline["traffic"] = TRAFFIC_B_PER_LAUNCH[net.scorer_mode]
                roofline["traffic_source"] = "constant from the PMC passes of " + TRAFFIC_SOURCE_FILE + " (not re-measured in this run): " + TRAFFIC_NOTE
        # per-kernel breakdown of ONE episode-batch (separate pass, event pairs around every kernel class)
        kernels = dict((k, {"ms": round(v[0] / breakdown_steps, 3), "launches": int(v[1] // breakdown_steps)}) for k, v in breakdown.items())
        kernel_sum = sum(v["ms"] for v in kernels.values())
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
        # the same launch IN SITU: once per episode-batch, right behind an episode's scorer kernels (caches hold the scorer's
        # data, not the previous gather's output) - one launch between two events, averaged over 5 episodes
        situ = []
        for _ in range(5):
            episode(env, T)         # (never run_step here: this block is rank 0's alone, a train step would enter a collective)
            ev0.record()
            samples._env.build_complete()
            ev1.record()
            torch.cuda.synchronize()
            situ.append(ev0.elapsed_time(ev1))
        g_situ_ms = sum(situ) / len(situ)
        g_rows = B * samples._env.n_complete
        g_bytes = g_rows * (cfg['dense_feature_num'] * 4 + cfg['category_feature_num'] * 4 + 32 * 4 + 10 * 4 + 36)
        g_gbs = g_bytes / (g_ms * 1e-3) / 1e9
        # the figure that LEADS is the in-situ one (a single launch behind a whole episode-batch: cold caches, what the episode pays);
        # the back-to-back figure (50 launches onto the same buffers: Infinity-Cache warm) is kept beside it as `burst`
        g_situ_gbs = g_bytes / (g_situ_ms * 1e-3) / 1e9
        gather = {"bound": "hbm", "kernel": "k_env_rows<complete>", "achieved": g_situ_gbs, "peak": HBM_PEAK_GBS,
                  "unit": "GB/s", "frac": g_situ_gbs / HBM_PEAK_GBS, "traffic": None, "avg_launch_ms": g_situ_ms,
                  "rows": g_rows,
                  "what": "one launch behind a whole episode-batch (event pair around the single launch, mean of 5)",
                  "burst": {"avg_launch_ms": g_ms, "achieved": g_gbs, "frac": g_gbs / HBM_PEAK_GBS,
                            "what": "50 back-to-back launches onto the same buffers (Infinity-Cache warm)"}}
        if not seq and B == 4096 and T == 9:
            # rocprofv3 WRITE_SIZE 64.15 MB + FETCH_SIZE 0.65 MB per launch of this exact shape against 66.8 MB of algorithmic
            # writes + 7.5 MB of (L2-resident) reads: no wasted traffic
            gather["traffic"] = GATHER_TRAFFIC_B
            gather["traffic_source"] = "constant from " + TRAFFIC_SOURCE_FILE + " (WRITE_SIZE + FETCH_SIZE per launch), not re-measured in this run"
        out = {
            "metric": "env-steps/s (batch=%d, %d-slot slate)" % (B, 9),
            "value": env_steps / elapsed,
            "unit": "env-steps/s",
            "n_gpus": world,
            "ranks_seen": ranks_seen,
            "per_rank_env_steps_per_s": [round(v, 1) for v in per_rank],
            "param_digest_per_rank": param_digests,
            "replicas_in_step": (None if param_digests is None else bool(max(param_digests) == min(param_digests))),
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32" if (not is_dien or net.scorer_mode == 'fp32') else "f32 (scorer matrix operands as fp16 hi+lo pairs, three f16 MFMAs per product, fp32 accumulate)",
            "data": "synthetic",
            "config": {"workload": "%s batch=%d per GPU, 284-item catalogue, 9-slot slate, %d-step horizon, "
                                   "%s simulator scorer, offline_action replay"
                                   % ('SeqSlateRecEnv-v0' if seq else 'SlateRecEnv-v0', B, T, cfg.get('algo', 'dien').upper() if is_dien else cfg['algo']),
                       "step": "one episode-batch = reset + %d env.step incl. reward forward" % T +
                               ("" if not trainer or args.train == 'bcq' else " with %s policy sampling + update (gradient all-reduce)" % args.train.upper()) +
                               ("" if args.train != 'bcq' else " driven by the BCQ policy (predict: 100 sampled actions per env), after %d BCQ "
                                "updates of 256 transitions (3 gradient all-reduces each)" % args.bcq_updates) +
                               ("" if not args.conti else "; continuous actions -> masked K-NN over the catalogue"),
                       "parallelism": "independent env batches per GPU (no data-path collective)"},
            # dien: the AUGRU recurrence; the GEMM-only families (dnn / widedeep / lstm) have no dominant matrix kernel,
            # their line carries the HBM roofline of the feature-gather kernel
            "roofline": roofline if roofline is not None else gather,
            "roofline_gather": gather,
            "kernels": kernels,
            # everything in an episode-batch that is not one of the scorer's kernels: env-state kernels, torch glue,
            # launch gaps, host (ms_per_step of the timed region - the scorer kernels' sum from the breakdown pass)
            "non_scorer_ms_per_step": round(elapsed / args.steps * 1e3 - kernel_sum, 3),
        }
        hu = getattr(env.samples, '_hist_unique', None)
        # RecDataBase.sample draws WITH replacement from the 2048-line cache window (base.py:92-100): distinct user histories
        # of the last batch (the first GRU / projections run once per distinct history; the AUGRU runs per env row)
        out["config"]["distinct_histories"] = int(hu[0].shape[0]) if hu is not None else B
        default_run = world == 1 and is_dien and not trainer and not seq and not args.conti
        if default_run and not args.no_extra_legs:
            out["extra"] = dict((name, extra_leg(args, workdir, rank, name))
                                for name in ('seq_t32', 'seq_t32_ppo', 'seq_t32_a2c', 'conti', 'bcq_conti', 'all_distinct', 'compat_numpy',
                                             'compat_rllib_mask', 'compat_d3rl_mask'))
            # the two figures VERDICT r5 asked to see beside `value`: the same workload through the API the reference's callers use
            # (rl4rs/env/base.py:256-263: ndarray / list returns, PCIe-inclusive) and with 4096 DISTINCT user histories
            def _v(leg):
                x = out["extra"].get(leg)
                return x.get('value') if isinstance(x, dict) else x
            out["reference_api_env_steps_per_s"] = _v('compat_numpy')
            out["reference_api_rllib_mask_env_steps_per_s"] = _v('compat_rllib_mask')
            out["reference_api_d3rl_mask_env_steps_per_s"] = _v('compat_d3rl_mask')
            out["all_distinct_env_steps_per_s"] = _v('all_distinct')
        if default_run and net.scorer_mode == 'fp16x2' and not args.no_fp32_leg:
            out["exact_fp32_scorer"] = extra_leg(args, workdir, rank, 'fp32')
        if args.train == 'bcq':
            out["bcq"] = out_bcq
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(cfg, records, seq, args.cpu_batch)
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
