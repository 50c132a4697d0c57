"""drift.py: does the rollout's rate move with how long the chip has been under load?  Back-to-back timed regions of one graph
replay each (32 launches of 256 cycles, rotating buffers), printed with the time since the first one; then the same after idling.
  python3 profiles/experiments/drift.py [--seconds 6] [--idle 2]"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import bench

ap = argparse.ArgumentParser()
ap.add_argument('--seconds', type=float, default=6.0)
ap.add_argument('--idle', type=float, default=2.0)
a = ap.parse_args()
dev = torch.device('cuda', 0)
n, T, L = 65536, 256, 32
eng = bench.reach_engine(n, dev, 0, False)
bufs = [eng.alloc_rollout(T) for _ in range(2)]
k = [0]


def issue(cnt):
    for _ in range(cnt):
        eng.rollout(T, out=bufs[k[0] % 2]); k[0] += 1


issue(4)
torch.cuda.synchronize()
g = bench.graph_of(lambda: issue(L))


def phase(name, seconds):
    t_start, rows = time.perf_counter(), []
    while time.perf_counter() - t_start < seconds:
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        g.replay()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        rows.append((t0 - t_start, n * T * L / dt / 1e9))
    step = max(1, len(rows) // 24)
    print(name, ' '.join(f'{t:.2f}s:{v:.1f}' for t, v in rows[::step]), flush=True)


phase('from-cold', a.seconds)
time.sleep(a.idle)
phase(f'after-{a.idle:.0f}s-idle', a.seconds)
phase('back-to-back', a.seconds)
