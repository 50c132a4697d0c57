"""Which workgroups of the 64-env rollout pipeline are the slow ones?  (-DS2D_STAMPS build.)  Loop clocks per iteration of every group
by XCD, by CU and by the number of episodes the group ended in the launch.   S2D_LIB=.../libs2d_hip.so python profiles/experiments/ws_slowest_groups.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import bench

dev = torch.device('cuda', 0)
T, n = 256, 65536
eng = bench.reach_engine(n, dev, 0, False)
bufs = [eng.alloc_rollout(T) for _ in range(2)]
for i in range(200):
    out = eng.rollout(T, out=bufs[i & 1])
torch.cuda.synchronize()
t = eng.terminal_obs.view(n // 64, 64, 10).cpu()
loop = t[:, 4, 1].double() / (T + 3)                      # agent wave's loop clocks per iteration
hw = t[:, 4, 4].contiguous().view(torch.int32)
xcc = t[:, 4, 5].contiguous().view(torch.int32) & 0xf
cu, sh, se = (hw >> 8) & 0xf, (hw >> 12) & 1, (hw >> 13) & 7
ends = out['done'].view(T, n // 64, 64).sum(dim=(0, 2)).cpu().double()
print(f'all groups: median {loop.median():.1f}  p90 {loop.quantile(0.9):.1f}  p99 {loop.quantile(0.99):.1f}  max {loop.max():.1f}')
for x in range(8):
    m = xcc == x
    if int(m.sum()):
        print(f'  XCD {x}: n={int(m.sum()):4d}  median {loop[m].median():.1f}  max {loop[m].max():.1f}')
key = (xcc * 64 + se * 16 + sh * 16 + cu).long()
per_cu = {}
for k in key.unique().tolist():
    per_cu[k] = loop[key == k]
spread = sorted((float(v.max() - v.min()), float(v.median())) for v in per_cu.values())
print(f'within a CU (its groups): spread of loop clocks median {spread[len(spread) // 2][0]:.1f}, max {spread[-1][0]:.1f}; CU medians from {min(s[1] for s in spread):.1f} to {max(s[1] for s in spread):.1f}')
c = torch.corrcoef(torch.stack([loop, ends]))[0, 1]
print(f'episodes ended per group in the launch: {ends.min():.0f} .. {ends.max():.0f}; correlation with the loop clocks {c:.2f}')
slow = loop >= loop.quantile(0.99)
print('slowest 1 %: XCDs', sorted(set(xcc[slow].tolist())), ' episodes ended', ends[slow].mean().item(), 'vs', ends.mean().item())
