"""Sustained load of the OTHER kernels for the power probe: `match` = 8 192 matches x 64-cycle launches of the 11v11 engine,
`step` = the per-step API at 65 536 envs, `rollout2` = T = 64 rollouts into ONE buffer (record in the Infinity Cache);
hipGraph replays back to back for SECONDS seconds, median rate per 5 s.   python profiles/experiments/sustained_other.py KIND [SECONDS]"""
import sys, time, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import torch
import bench

kind = sys.argv[1]
secs = float(sys.argv[2]) if len(sys.argv) > 2 else 12.0
dev = torch.device('cuda:0')
if kind == 'match':
    from soccer2d_amd.match import MatchEngine, make_match_config
    n, T, K = 8192, 64, 16
    eng = MatchEngine(n, dev, cfg=make_match_config()); eng.reset()
    g0 = torch.Generator(device='cpu').manual_seed(1234)
    eng.cycle += (2 * torch.randint(0, 1500, (n,), generator=g0, dtype=torch.int32)).to(dev)
    ro = eng.alloc_rollout(T)
    issue = lambda: [eng.rollout(T, out=ro) for _ in range(K)]
    units, name = n * T * K, 'G match-steps/s'
elif kind == 'step':
    n, K = 65536, 2048
    eng = bench.reach_engine(n, dev, 0, False)
    issue = lambda: [eng.step(None) for _ in range(K)]
    units, name = n * K, 'G env-steps/s (per-step API)'
else:
    n, T, K = 65536, 64, 64
    eng = bench.reach_engine(n, dev, 0, False)
    ro = eng.alloc_rollout(T)
    issue = lambda: [eng.rollout(T, out=ro) for _ in range(K)]
    units, name = n * T * K, 'G env-steps/s (T = 64, one buffer)'
issue(); torch.cuda.synchronize()
g = bench.graph_of(issue)
t0 = time.perf_counter(); cur, b0 = [], t0
print(kind, flush=True)
while True:
    torch.cuda.synchronize(); a = time.perf_counter(); g.replay(); torch.cuda.synchronize(); b = time.perf_counter()
    cur.append(units / (b - a) / 1e9)
    if b - b0 >= 5.0 or b - t0 >= secs:
        cur.sort()
        print(f'  +{b0 - t0:6.1f} s  n={len(cur):5d}  median {cur[len(cur) // 2]:7.2f} {name}', flush=True)
        cur, b0 = [], b
    if b - t0 >= secs:
        break
