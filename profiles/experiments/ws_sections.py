"""Section timers of the wave-specialised rollout kernel (-DS2D_PROFILE=1/2/3: policy / simulate / observe wave).
Run with S2D_LIB pointing at the instrumented build; prints ticks per wave-step per section."""
import os, sys
sys.path.insert(0, 'gym-soccer-2d-env_amd'); sys.path.insert(0, 'profiles/experiments')
import torch
from soccer2d_amd.engine import Engine, make_config
from ablate import KW
N, T, L = 65536, 64, 16
eng = Engine(N, 'cuda:0', cfg=make_config(noise=bool(int(os.environ.get('S2D_NOISE', '0'))), **KW)); eng.reset()
out = eng.alloc_rollout(T)
for _ in range(8): eng.rollout(T, out=out)
eng.stats_reset()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(L): eng.rollout(T, out=out)
e1.record(); torch.cuda.synchronize()
s = eng.stats.cpu().tolist()
ws = (N // 64) * T * L
names = {'1': ['policy+decode', '-', '-', 'barrier'],
         '2': ['refill check + command read', 'simulator cycle', 'done test+snapshot+reset', 'barrier'],
         '3': ['read+player half+reward', 'reset+stores+tile words', '-', 'barrier']}[os.environ.get('S2D_SECTIONS', '2')]
tot = sum(s[4:8])
print(eng.kernel_name(), 'launch us', e0.elapsed_time(e1) * 1e3 / L, 'per cycle ns', e0.elapsed_time(e1) * 1e6 / L / T)
for nm, v in zip(names, s[4:8]):
    print(f'  {nm:28s} {v / ws:9.1f} ticks/wave-step  {100.0 * v / tot:5.1f} %')
print('  total ticks/wave-step', tot / ws)
