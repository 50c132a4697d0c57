"""Which match makes launch 202 of the 11v11 benchmark sequence take 960 us instead of 270?  The sequence is deterministic, so: run to the
launch, snapshot the arena, then replay that ONE launch with every match outside a subset replaced by a copy of a quiet donor match,
and bisect on the subset.  Prints the culprit's modes over the 64 cycles and its state before the launch.
  python profiles/experiments/match_slow_launch_bisect.py [launch index]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (os.path.join(ROOT, 'gym-soccer-2d-env_amd'), os.path.join(ROOT, 'tests'), ROOT):
    sys.path.insert(0, p)
import numpy as np
import torch
import match_oracle as MO
from soccer2d_amd.match import MatchEngine, make_match_config

TARGET = int(sys.argv[1]) if len(sys.argv) > 1 else 202
n, T = 8192, 64
eng = MatchEngine(n, 'cuda:0', cfg=make_match_config())
eng.reset()
g = torch.Generator(device='cpu').manual_seed(1234)
eng.cycle += (2 * torch.randint(0, 1500, (n,), generator=g, dtype=torch.int32)).to('cuda:0')
ro = eng.alloc_rollout(T)
for _ in range(30 + TARGET):
    eng.rollout(T, out=ro)
torch.cuda.synchronize()
snap = eng.arena.clone()
FIELDS = MO.OBJ_FIELDS + ('catch_ban', 'card') + MO.ENV_FIELDS + ('ball_holder', 'goalie_moves', 'set_play_taker', 'last_kicker', 'stopped_cycle', 'tick')


def run(keep):
    eng.arena.copy_(snap)
    if keep is not None:
        drop = torch.ones(n, dtype=torch.bool, device='cuda:0'); drop[keep] = False
        donor = int(torch.nonzero((eng.mode == 2) & drop)[0])      # a play_on match outside the subset
        for f in FIELDS:
            t = getattr(eng, f)
            t[drop] = t[donor].clone()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); eng.rollout(T, out=ro); b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3


full = run(None)
none = run(torch.tensor([], dtype=torch.long, device='cuda:0'))
print(f'launch {TARGET}: {full:.1f} us with every match; {none:.1f} us with 8 192 copies of a quiet match')
lo, hi = 0, n
while hi - lo > 1:
    mid = (lo + hi) // 2
    t_lo = run(torch.arange(lo, mid, device='cuda:0'))
    t_hi = run(torch.arange(mid, hi, device='cuda:0'))
    print(f'  matches [{lo}, {mid}): {t_lo:.1f} us   [{mid}, {hi}): {t_hi:.1f} us')
    if t_lo >= t_hi: hi = mid
    else: lo = mid
print('culprit: match', lo, f'alone: {run(torch.tensor([lo], device="cuda:0")):.1f} us')
eng.arena.copy_(snap); torch.cuda.synchronize()
print('before the launch: mode', int(eng.mode[lo]), 'side', int(eng.mode_side[lo]), 'cycle', int(eng.cycle[lo]), 'timer', int(eng.setplay_timer[lo]), 'ball', float(eng.x[lo, 22]), float(eng.y[lo, 22]),
      'ball v', float(eng.vx[lo, 22]), float(eng.vy[lo, 22]), 'holder', int(eng.ball_holder[lo]), 'offside mask', int(eng.offside_mask[lo]))
print('players x:', [round(float(v), 2) for v in eng.x[lo, :22]])
print('players y:', [round(float(v), 2) for v in eng.y[lo, :22]])
print('tackle cycles:', [int(v) for v in eng.tackle_cycles[lo, :22]], 'cards', [int(v) for v in eng.card[lo, :22]])
eng.rollout(T, out=ro); torch.cuda.synchronize()
print('modes over the 64 cycles:', ro['mode'][:, lo].cpu().tolist())
ob = ro['obs'][:, lo].cpu().numpy()
print('ball x over the cycles:', [round(float(v), 1) for v in ob[:, 22, 0]])
print('ball y over the cycles:', [round(float(v), 1) for v in ob[:, 22, 1]])

# which state word of the culprit is it?  Replay with one family of its words replaced by the donor's
keep = torch.tensor([lo], device='cuda:0')


def run_with(patch):
    eng.arena.copy_(snap)
    drop = torch.ones(n, dtype=torch.bool, device='cuda:0'); drop[keep] = False
    donor = int(torch.nonzero((eng.mode == 2) & drop)[0])
    for f in FIELDS:
        t = getattr(eng, f)
        t[drop] = t[donor].clone()
    patch(donor)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); eng.rollout(T, out=ro); b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3


def bits(t):
    return t.view(torch.int32)


eng.arena.copy_(snap); torch.cuda.synchronize()
for f in ('x', 'y', 'vx', 'vy'):
    v = getattr(eng, f)[lo]
    print(f, 'of the 23 objects, bit patterns of the small ones:', [(j, hex(int(bits(v)[j]) & 0xffffffff), float(v[j])) for j in range(23) if abs(float(v[j])) < 1e-30])
for name, fields in (('ball velocity', ('vx', 'vy')), ('ball position', ('x', 'y'))):
    def patch(donor, fields=fields):
        for f in fields:
            getattr(eng, f)[lo, 22] = getattr(eng, f)[donor, 22]
    print(f'culprit with the donor\'s {name}: {run_with(patch):.1f} us')
for name, fields in (('player velocities', ('vx', 'vy')), ('player positions', ('x', 'y', 'body')), ('stamina block', ('stamina', 'effort', 'recovery', 'stamina_capacity')), ('tackle cycles', ('tackle_cycles',))):
    def patch(donor, fields=fields):
        for f in fields:
            getattr(eng, f)[lo, :22] = getattr(eng, f)[donor, :22]
    print(f'culprit with the donor\'s {name}: {run_with(patch):.1f} us')


def patch_env(donor):
    for f in MO.ENV_FIELDS + ('ball_holder', 'goalie_moves', 'set_play_taker', 'last_kicker', 'stopped_cycle'):
        getattr(eng, f)[lo] = getattr(eng, f)[donor]
print(f'culprit with the donor\'s match words (mode, timers, scores, cycle ...; tick kept): {run_with(patch_env):.1f} us')
