"""Per-role busy clocks of the two-envs-per-lane rollout pipeline (experiment build: make -C gym-soccer-2d-env_amd/csrc
OUT=../lib/exp/stamps EXTRA=-DS2D_STAMPS): clocks between a barrier's release and the wave's arrival at the next barrier, and
the clocks of the whole loop, per iteration; median / max over the workgroups.
  S2D_LIB=gym-soccer-2d-env_amd/lib/exp/stamps/libs2d_hip.so python profiles/experiments/ws2_stamps.py [--noise] [--envs N]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import bench

dev = torch.device('cuda', 0)
T = 256
n = int(sys.argv[sys.argv.index('--envs') + 1]) if '--envs' in sys.argv else 65536
NOISE = '--noise' in sys.argv
eng = bench.reach_engine(n, dev, 0, NOISE)
bufs = [eng.alloc_rollout(T) for _ in range(2)]
for i in range(300):
    eng.rollout(T, out=bufs[i & 1])
torch.cuda.synchronize()
print(eng.kernel_name())
t = eng.terminal_obs.view(n // 128, 128, 10)
names = ('policy', 'simulate', 'agent', 'ball')
for r, name in enumerate(names):
    busy = t[:, 2 * r, 0].double() / (T + 3)
    total = t[:, 2 * r, 1].double() / (T + 3)
    print(f'{name:9s} busy per iteration: median {busy.median():8.1f}  max {busy.max():8.1f}   loop per iteration: median {total.median():8.1f} max {total.max():8.1f}')
