"""Which cycles around a restart of all matches are expensive?  Per-launch HIP-event time of 4-cycle launches through two
half-times and a time-over of 600-cycle games, for the stock configuration and for drop_ball_time = 5 (the kick-off set play
ends after 5 cycles instead of 100).   python profiles/experiments/match_restart_cost.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, 'gym-soccer-2d-env_amd'))
import torch
from soccer2d_amd.match import MatchEngine, make_match_config

T = 4
for name, kw in (('stock (drop_ball_time 100)', {}), ('drop_ball_time 5', dict(drop_ball_time=5)), ('noise on', dict(noise=True))):
    m = MatchEngine(8192, 'cuda:0', cfg=make_match_config(half_time_cycles=300, **kw))
    ro = m.alloc_rollout(T)
    for _ in range(40):
        m.rollout(T, out=ro)            # cycles 0 .. 159
    torch.cuda.synchronize()
    n = 200                             # cycles 160 .. 959: half-time at 300, time-over at 600, half-time at 900
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
    for a, b in ev:
        a.record(); m.rollout(T, out=ro); b.record()
    torch.cuda.synchronize()
    d = [a.elapsed_time(b) * 1e3 / T for a, b in ev]
    print(f'{name}: us per cycle, one value per {T}-cycle launch, first launch starts at cycle 160', flush=True)
    for i in range(0, n, 25):
        print(f'  cycle {160 + i * T:4d}:', ' '.join(f'{x:5.1f}' for x in d[i:i + 25]), flush=True)
    print('  modes now:', torch.bincount(m.mode.flatten().long(), minlength=10).tolist(), flush=True)
