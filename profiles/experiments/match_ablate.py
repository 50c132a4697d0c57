"""Where does the 11v11 cycle spend its time?  Fused rollouts with fixed caller actions."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, 'gym-soccer-2d-env_amd'))
import torch
from soccer2d_amd.match import MatchEngine

n, T, reps = 8192, 64, 8


def timeit(eng, actions):
    ro = eng.alloc_rollout(T, with_obs=True)
    for _ in range(2):
        eng.rollout(T, actions=actions, out=ro)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        eng.rollout(T, actions=actions, out=ro)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / (reps * T)


def const_actions(cmd, a=50.0, b=10.0):
    act = torch.zeros((T, n, 22, 3), device='cuda:0')
    act[..., 0] = cmd; act[..., 1] = a; act[..., 2] = b
    return act


eng = MatchEngine(n)
print(f'random policy (in-kernel Philox)     {timeit(eng, None):7.2f} us/cycle')
for name, cmd in (('all NONE', 0), ('all DASH', 1), ('all TURN', 2), ('all KICK', 3), ('all TACKLE', 4)):
    eng.reset()
    print(f'{name:36s} {timeit(eng, const_actions(cmd)):7.2f} us/cycle')
g = torch.Generator(device='cuda:0').manual_seed(0)
act = const_actions(0)
act[..., 0] = torch.randint(1, 5, (T, n, 22), device='cuda:0', generator=g).float()
eng.reset()
print(f'{"mixed commands (caller tensor)":36s} {timeit(eng, act):7.2f} us/cycle')
ro = eng.alloc_rollout(T, with_obs=False)
eng.reset()
torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(reps):
    eng.rollout(T, out=ro)
e1.record(); torch.cuda.synchronize()
print(f'{"random policy, no obs stream":36s} {e0.elapsed_time(e1) * 1e3 / (reps * T):7.2f} us/cycle')
