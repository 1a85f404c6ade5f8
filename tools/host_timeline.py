"""Host timeline of the reference-shaped API per episode-batch (bench workload, no return_tensors): python tools/host_timeline.py [plain|rllib] [episodes]
-> ms per episode-batch spent in reset, in the library call + launches, in the work done while the kernels run, in the wait."""
import sys, time, os, tempfile, argparse
sys.path.insert(0, os.getcwd())
import torch
import bench
from rl4rs_amd import device as D

mode = sys.argv[1] if len(sys.argv) > 1 else 'rllib'
a = argparse.Namespace(batch=4096, env='slate', horizon=9, log_records=8193, scorer='auto', algo='dien', conti=False)
cfg, _ = bench.make_config(a, tempfile.mkdtemp(), 0)
cfg['return_tensors'] = False
if mode == 'rllib':
    cfg['support_rllib_mask'] = True
env = bench.build_env(cfg, False)
env.seed(1000)
env.sim._recData.store.preload(torch.device('cuda', 0))
T = 9
acc = {}
def tick(name, t0):
    acc[name] = acc.get(name, 0.0) + time.perf_counter() - t0

orig_sr = D.DeviceStepper.step_record
orig_wait = D.wait_stream
def wait_stream(*x, **k):
    t0 = time.perf_counter(); r = orig_wait(*x, **k); tick('wait', t0); return r
D.wait_stream = wait_stream
def step_record(self, actions, conti=False, want=(), shadow=None):
    def sh(rec):
        tick('call+launch', self._t0)
        t0 = time.perf_counter(); shadow(rec); tick('shadow', t0)
    self._t0 = time.perf_counter()
    r = orig_sr(self, actions, conti=conti, want=want, shadow=sh)
    return r
D.DeviceStepper.step_record = step_record
# finer marks: wait return -> env.step return (post), env.step entry -> library call entry (pre), the library call itself (launches)
_lib_mod = D._lib.load()
_orig_call = _lib_mod.rl4rs_env_step_record_host
marks = {}
def _call(*x):
    t0 = time.perf_counter()
    if 'step_entry' in marks:
        acc['pre (step entry -> library call)'] = acc.get('pre (step entry -> library call)', 0.0) + t0 - marks['step_entry']
    r = _orig_call(*x)
    tick('library call (launches)', t0)
    return r
class _Shim(object):
    def __getattr__(self, k):
        return _call if k == 'rl4rs_env_step_record_host' else getattr(_lib_mod, k)
_wait2 = D.wait_stream
def wait_stream2(*x, **k):
    r = _wait2(*x, **k)
    marks['wait_done'] = time.perf_counter()
    return r
D.wait_stream = wait_stream2

EPISODES = int(sys.argv[2]) if len(sys.argv) > 2 else 4
per_episode = []
for ep in range(EPISODES + 2):
    if ep == 2:
        acc.clear(); torch.cuda.synchronize(); T0 = time.perf_counter()
    te = time.perf_counter()
    t0 = time.perf_counter(); env.reset(); tick('reset', t0)
    for _ in range(T):
        t0 = time.perf_counter(); act = env.offline_action; tick('offline_action', t0)
        t0 = time.perf_counter(); marks['step_entry'] = t0
        if getattr(env.sim, '_stepper', None) is not None and not isinstance(env.sim._stepper.lib, _Shim):
            env.sim._stepper.lib = _Shim()
        env.step(act); tick('step_total', t0)
        if 'wait_done' in marks:
            acc['post (wait done -> step returns)'] = acc.get('post (wait done -> step returns)', 0.0) + time.perf_counter() - marks['wait_done']
    per_episode.append(round((time.perf_counter() - te) * 1e3, 2))
torch.cuda.synchronize()
tot = (time.perf_counter() - T0) / EPISODES * 1e3
print(mode, 'episode ms', round(tot, 2), {k: round(v / EPISODES * 1e3, 2) for k, v in acc.items()})
if EPISODES > 4:
    print('per episode:', per_episode[2:])
