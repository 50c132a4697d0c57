"""The headline rollout under SUSTAINED load: regions of 20 graph-replayed launches (65 536 envs x 256 cycles, two rotating buffers)
back to back for SECONDS seconds, median G env-steps/s per 5-second bin -- does the rate a short bench reads hold for minutes?
  python profiles/experiments/sustained_load.py [SECONDS]"""
import sys, time, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import torch
import bench

secs = float(sys.argv[1]) if len(sys.argv) > 1 else 90.0
noise = '--noise' in sys.argv
dev = torch.device('cuda:0')
eng = bench.reach_engine(65536, dev, 0, noise)
T, K = 256, 20
bufs = [eng.alloc_rollout(T) for _ in range(2)]
k = [0]


def issue(cnt):
    for _ in range(cnt):
        eng.rollout(T, out=bufs[k[0] % 2]); k[0] += 1


issue(4); torch.cuda.synchronize()
g = bench.graph_of(lambda: issue(K))
t0 = time.perf_counter()
cur, b0 = [], t0
print(f'noise={int(noise)}', flush=True)
while True:
    torch.cuda.synchronize(); a = time.perf_counter(); g.replay(); torch.cuda.synchronize(); b = time.perf_counter()
    cur.append(65536 * T * K / (b - a) / 1e9)
    if b - b0 >= 5.0:
        cur.sort()
        print(f'  +{b0 - t0:6.1f} s  n={len(cur):5d}  p10 {cur[len(cur) // 10]:6.1f}  median {cur[len(cur) // 2]:6.1f}  p90 {cur[len(cur) * 9 // 10]:6.1f}', flush=True)
        cur, b0 = [], b
    if b - t0 >= secs:
        break
