"""Per-launch device time (one HIP event pair per launch, no profiler attached) of the 11v11 and the reach_ball rollout
kernels over a few hundred launches: is the spread rocprofv3 reports (match: 354 .. 847 us) a property of the workload, of the
device, or of the profiler?   python profiles/experiments/launch_spread.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, 'gym-soccer-2d-env_amd'))
import torch
from soccer2d_amd.match import MatchEngine, make_match_config
from soccer2d_amd.engine import Engine, make_config


def spread(run, n, name):
    for _ in range(64):
        run()
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
    for a, b in ev:
        a.record(); run(); b.record()
    torch.cuda.synchronize()
    d = [a.elapsed_time(b) * 1e3 for a, b in ev]
    s = sorted(d)
    print(f'{name}: {n} launches  min {s[0]:.0f}  median {s[n // 2]:.0f}  p95 {s[int(n * .95)]:.0f}  max {s[-1]:.0f} us', flush=True)
    for i in range(0, n, 20):
        print('   ', i, ' '.join(f'{x:5.0f}' for x in d[i:i + 20]), flush=True)
    return d


m = MatchEngine(8192, 'cuda:0', cfg=make_match_config())
ro = m.alloc_rollout(64)
spread(lambda: m.rollout(64, out=ro), 400, 'match 8192 x 64, matches in phase (all started together)')
# the same with the matches spread over all phases of a game (cycle counters staggered by a quarter game per quarter of the batch)
m2 = MatchEngine(8192, 'cuda:0', cfg=make_match_config(half_time_cycles=300))
ro2 = m2.alloc_rollout(64)
spread(lambda: m2.rollout(64, out=ro2), 200, 'match 8192 x 64, 600-cycle games (a time-over every 9.4 launches)')
e = Engine(65536, 'cuda:0', cfg=make_config(noise=False, change_ball_velocity=True, use_continuous_action=False)); e.reset()
r = e.alloc_rollout(64)
spread(lambda: e.rollout(64, out=r), 400, 'reach_ball 65536 x 64')
