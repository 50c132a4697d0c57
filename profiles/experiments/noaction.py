"""Does the S-wave's per-step action store (and the vmcnt(0) the compiler puts at the action
fetch join) cost time?  rollout with and without the action output buffer."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, 'gym-soccer-2d-env_amd')); sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch
from soccer2d_amd.engine import Engine, make_config
from ablate import KW

def timeit(eng, ro, T=64, reps=20):
    for _ in range(3): eng.rollout(T, out=ro)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): eng.rollout(T, out=ro)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / (reps * T)

for name, kw, sp in (('baseline', KW, None), ('never-done', dict(KW, max_steps=1000000, min_distance_to_ball=0.0), dict(pitch_half_length=1e6, pitch_half_width=1e6))):
    eng = Engine(65536, 'cuda:0', cfg=make_config(server_params=sp, **kw)); eng.reset()
    ro = eng.alloc_rollout(64)
    a = timeit(eng, ro)
    ro2 = dict(ro); ro2['action'] = None
    b = timeit(eng, ro2)
    ro3 = dict(ro2); ro3['obs'] = None; ro3['reward'] = None; ro3['done'] = None; ro3['result'] = None
    c = timeit(eng, ro3)
    print(f'{name:12s} all outputs {a:.3f}  no action store {b:.3f}  no outputs at all {c:.3f} us/cycle')
