"""roll_driver.py: N launches of the reach_ball rollout (T fused cycles) for profiler runs -- no timing, no child processes.
  python3 profiles/experiments/roll_driver.py --fuse 256 --launches 24 --record full|noobs|none [--noise] [--rotate 2]"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import bench

ap = argparse.ArgumentParser()
ap.add_argument('--envs', type=int, default=65536)
ap.add_argument('--fuse', type=int, default=256)
ap.add_argument('--launches', type=int, default=24)
ap.add_argument('--record', default='full')
ap.add_argument('--noise', action='store_true')
ap.add_argument('--rotate', type=int, default=2)
ap.add_argument('--variant', default='dqn')
a = ap.parse_args()
dev = torch.device('cuda', 0)
eng = bench.reach_engine(a.envs, dev, 0, a.noise, a.variant)
bufs = []
for _ in range(max(1, a.rotate)):
    o = eng.alloc_rollout(a.fuse, with_obs=(a.record == 'full'))
    if a.record == 'none':
        o = {k: None for k in o}
    bufs.append(o)
import time
t0 = time.perf_counter()
k = 0
while time.perf_counter() - t0 < 0.3:
    eng.rollout(a.fuse, out=bufs[k % len(bufs)]); k += 1
    torch.cuda.synchronize()
for i in range(a.launches):
    eng.rollout(a.fuse, out=bufs[i % len(bufs)])
torch.cuda.synchronize()
print('done', eng.kernel_name())
