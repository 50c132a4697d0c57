"""Per-role busy clocks of the 64-env rollout pipeline (experiment builds with -DS2D_STAMPS): clocks between a barrier's release and
the wave's arrival at the next one, clocks of the whole loop, and when each workgroup began / ended (100 MHz real-time counter) --
is every workgroup resident from the start?   S2D_LIB=.../libs2d_hip.so python profiles/experiments/ws_stamps4.py [--noise]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import bench

dev = torch.device('cuda', 0)
T, n = 256, 65536
NOISE = '--noise' in sys.argv
eng = bench.reach_engine(n, dev, 0, NOISE)
bufs = [eng.alloc_rollout(T) for _ in range(2)]
for i in range(300):
    eng.rollout(T, out=bufs[i & 1])
torch.cuda.synchronize()
print(eng.kernel_name())
t = eng.terminal_obs.view(n // 64, 64, 10)
names = ('policy', 'simulate', 'agent', 'ball', 'store')
for r, name in enumerate(names):
    busy = t[:, 2 * r, 0].double() / (T + 3)
    total = t[:, 2 * r, 1].double() / (T + 3)
    if float(total.max()) == 0:
        continue
    print(f'{name:9s} busy per iteration: median {busy.median():8.1f}  max {busy.max():8.1f}   loop per iteration: median {total.median():8.1f} max {total.max():8.1f}')
beg, end = t[:, 4, 2].double(), t[:, 4, 3].double()       # agent wave's real-time stamps (10 ns ticks, low 24 bits)
b0 = beg.min()
print(f'workgroups begin: median +{(beg.median() - b0) / 100:.1f} us, p99 +{(beg.quantile(0.99) - b0) / 100:.1f} us, max +{(beg.max() - b0) / 100:.1f} us after the first; '
      f'end: median +{(end.median() - b0) / 100:.1f} us, max +{(end.max() - b0) / 100:.1f} us; late starters (> 20 us): {int((beg - b0 > 2000).sum())} of {beg.numel()}')
