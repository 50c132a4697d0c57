"""match_quiet_rate.py: what the 11v11 kernel does when nothing happens -- every match in play_on, the ball at rest far from every
player (nobody can reach it within the timed launches) -- against the benchmark's steady state (set plays, ball-outs, goals).
  python3 profiles/experiments/match_quiet_rate.py"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import bench
from soccer2d_amd.match import MatchEngine, make_match_config
from soccer2d_amd._capi_match import GM_PLAY_ON

dev = torch.device('cuda', 0)
n, T = 8192, 64
eng = MatchEngine(n, dev, cfg=make_match_config())
ro = eng.alloc_rollout(T)
g = None
for trial in range(4):
    eng.reset()
    eng.mode.fill_(GM_PLAY_ON); eng.mode_side.fill_(0)
    eng.x[:, 22] = 0.0; eng.y[:, 22] = 33.0                # the ball at rest beside the side line, 20+ m from the nearest player
    eng.vx[:, 22] = 0.0; eng.vy[:, 22] = 0.0
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); eng.rollout(T, out=ro); eng.rollout(T, out=ro); e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 2
    modes = torch.bincount(eng.mode.long(), minlength=32).tolist()
    print(f'quiet trial {trial}: {us:7.1f} us per 64-cycle launch = {n * T / us / 1e3:5.3f} G match-steps/s; modes after: play_on {modes[GM_PLAY_ON]} of {n}', flush=True)
from soccer2d_amd._capi_match import GM_KICK_IN, GM_AFTER_GOAL
for name, mode in (('kick-in nobody takes', GM_KICK_IN), ('after-goal wait', GM_AFTER_GOAL)):
    for trial in range(2):
        eng.reset()
        eng.mode.fill_(mode); eng.mode_side.fill_(1)
        eng.x[:, 22] = 0.0; eng.y[:, 22] = 34.0; eng.vx[:, 22] = 0.0; eng.vy[:, 22] = 0.0
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); eng.rollout(40, out=ro); e1.record()      # 40 cycles: inside the 50-cycle after-goal wait / the 100-cycle drop-ball time
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 * 64 / 40
        modes = torch.bincount(eng.mode.long(), minlength=32).tolist()
        print(f'{name:22s} trial {trial}: {us:7.1f} us per 64 cycles = {n * T / us / 1e3:5.3f} G; still in that mode: {modes[mode]} of {n}', flush=True)
m = bench.measure_match(n, dev, 0, T, 16, 5, torch.cuda.current_stream(dev), 200.0, phase='spread')
print(f'benchmark steady state: {m["roofline"]["launch_us"]:7.1f} us = {m["value"] / 1e9:5.3f} G', flush=True)
# the benchmark's own engine state (players spread by a long random walk, stamina spent, tackle timers running) with every match
# forced quiet: is what is left the players' state or the modes?
from soccer2d_amd.match import MatchEngine as ME
eng2 = ME(n, dev, cfg=make_match_config())
eng2.reset()
g = torch.Generator(device='cpu').manual_seed(1234)
eng2.cycle += (2 * torch.randint(0, 1500, (n,), generator=g, dtype=torch.int32)).to(dev)
for _ in range(40):
    eng2.rollout(T, out=ro)
torch.cuda.synchronize()
modes = torch.bincount(eng2.mode.long(), minlength=32).tolist()
print('steady state after 2 560 cycles: modes', {k: v for k, v in enumerate(modes) if v}, flush=True)
for label, force in (('as it is', False), ('forced quiet', True), ('as it is', False)):
    if force:
        eng2.mode.fill_(GM_PLAY_ON); eng2.mode_side.fill_(0)
        eng2.x[:, 22] = 0.0; eng2.y[:, 22] = 33.0; eng2.vx[:, 22] = 0.0; eng2.vy[:, 22] = 0.0
        eng2.offside_mask.fill_(0)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); eng2.rollout(T, out=ro); e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3
    print(f'steady-state players, {label:12s}: {us:7.1f} us per 64-cycle launch = {n * T / us / 1e3:5.3f} G', flush=True)

