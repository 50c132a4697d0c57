"""Duration of every single launch of the 11v11 rollout kernel (8 192 matches x 64 cycles, matches out of phase) over 600 launches,
HIP events around each launch, no profiler: are the 400-970 us launches that rocprofv3 shows (15 of 850) a property of the kernel?
With the slow ones: what the batch looked like before them (matches in each mode, matches that ended in the launch).
  python profiles/experiments/match_launch_distribution.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (os.path.join(ROOT, 'gym-soccer-2d-env_amd'), ROOT):
    sys.path.insert(0, p)
import numpy as np
import torch
from soccer2d_amd.match import MatchEngine, make_match_config

n, T, L = 8192, 64, 600
eng = MatchEngine(n, 'cuda:0', cfg=make_match_config())
eng.reset()
g = torch.Generator(device='cpu').manual_seed(1234)
eng.cycle += (2 * torch.randint(0, 1500, (n,), generator=g, dtype=torch.int32)).to('cuda:0')
ro = eng.alloc_rollout(T)
for _ in range(30):
    eng.rollout(T, out=ro)
torch.cuda.synchronize()
ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(L)]
modes, piles = [], []


def pile_ups():
    """per match: number of player pairs closer than the sum of their radii + 5 cm (an overlap the collision passes have to resolve)"""
    xy = torch.stack([eng.x[:, :22], eng.y[:, :22]], dim=2)
    d = torch.cdist(xy, xy) + torch.eye(22, device=xy.device) * 10.0
    return (d < 0.65).sum(dim=(1, 2)) // 2


for k in range(L):
    modes.append(torch.bincount(eng.mode.clamp(0, 31), minlength=32).cpu())
    pu = pile_ups()
    piles.append((int((pu > 0).sum()), int(pu.max()), int((pu >= 3).sum())))
    ev[k][0].record(); eng.rollout(T, out=ro); ev[k][1].record()
torch.cuda.synchronize()
us = np.array([a.elapsed_time(b) * 1e3 for a, b in ev])
print(f'{L} launches: median {np.median(us):.1f} us, p90 {np.percentile(us, 90):.1f}, p99 {np.percentile(us, 99):.1f}, max {us.max():.1f}; above 1.5 x median: {(us > 1.5 * np.median(us)).sum()}')
print('(each launch is preceded by a host-side bincount + copy, i.e. starts on an idle GPU)')
slow = np.argsort(us)[-5:]
for k in slow:
    m = modes[k].numpy()
    print(f'  launch {k}: {us[k]:.1f} us; matches with touching players before it {piles[k][0]}, most pairs in one match {piles[k][1]}, matches with >= 3 pairs {piles[k][2]}; '
          'by mode: ' + ', '.join(f'{i}:{c}' for i, c in enumerate(m) if c))
pl = np.array(piles)
fast = us < np.percentile(us, 50)
print(f'launches below the median: most touching pairs in one match {pl[fast, 1].mean():.1f} on average, matches with >= 3 pairs {pl[fast, 2].mean():.2f}; '
      f'the 8 slowest: {pl[np.argsort(us)[-8:], 1].mean():.1f} and {pl[np.argsort(us)[-8:], 2].mean():.2f}')
print('corr(duration, most pairs in one match) =', round(float(np.corrcoef(us, pl[:, 1])[0, 1]), 3), ' corr(duration, matches with >= 3 pairs) =', round(float(np.corrcoef(us, pl[:, 2])[0, 1]), 3))
ev2 = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(L)]
for k in range(L):                                        # back to back, no host work in between
    ev2[k][0].record(); eng.rollout(T, out=ro); ev2[k][1].record()
torch.cuda.synchronize()
us2 = np.array([a.elapsed_time(b) * 1e3 for a, b in ev2])
print(f'back to back: median {np.median(us2):.1f} us, p90 {np.percentile(us2, 90):.1f}, p99 {np.percentile(us2, 99):.1f}, max {us2.max():.1f}; above 1.5 x median: {(us2 > 1.5 * np.median(us2)).sum()}')
