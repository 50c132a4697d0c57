"""What a 64-cycle launch of the 11v11 kernel costs when every match is in one scene (stock rules, 8 192 matches, random policy):
quiet play, a kick-in nobody takes, and the shoot-out's waits (PenaltyTaken_ with the ball at rest far from everybody, PenaltyReady_,
PenaltyMiss_); then the share of match-cycles a long random-policy run spends in the shoot-out."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import bench
from soccer2d_amd.match import MatchEngine, make_match_config
from soccer2d_amd._capi_match import (GM_KICK_IN, GM_PLAY_ON, GM_PENALTY_TAKEN, GM_PENALTY_READY, GM_PENALTY_MISS, GM_PENALTY_SETUP)

dev = torch.device('cuda', 0)
n, T = 8192, 64
eng = MatchEngine(n, dev, cfg=make_match_config())
print(eng.kernel_name())
ro = eng.alloc_rollout(T)


def scene(name, mode, side, taker_word, ball):
    ts = []
    for rep in range(12):
        eng.reset()
        if mode == GM_PENALTY_SETUP:                      # let the referee place everybody, then force the scene
            eng.mode.fill_(GM_PENALTY_SETUP); eng.mode_side.fill_(1); eng.set_play_taker.fill_(11)
        else:
            eng.mode.fill_(mode); eng.mode_side.fill_(side); eng.set_play_taker.fill_(taker_word)
            eng.x[:, 22] = ball[0]; eng.y[:, 22] = ball[1]; eng.vx[:, 22] = 0.0; eng.vy[:, 22] = 0.0
        torch.cuda.synchronize(); a = time.perf_counter(); eng.rollout(T, out=ro); torch.cuda.synchronize(); ts.append(time.perf_counter() - a)
    ts.sort()
    print(f'{name:34s} {ts[len(ts) // 2] * 1e6:8.1f} us per {T}-cycle launch (median of 12 single launches from a reset)')


scene('quiet play', GM_PLAY_ON, 0, 0, (0.0, 33.0))
scene('kick-in nobody takes', GM_KICK_IN, 1, 0, (0.0, 34.0))
scene('PenaltyTaken_, ball at rest', GM_PENALTY_TAKEN, 1, 11, (30.0, 20.0))
scene('PenaltyReady_', GM_PENALTY_READY, 1, 11, (10.0, 0.0))
scene('PenaltyMiss_ (verdict)', GM_PENALTY_MISS, 1, 11 | (1 << 12), (10.0, 0.0))
# share of the shoot-out in a long run
eng.reset()
cnt = torch.zeros(32, dtype=torch.int64, device=dev)
for k in range(400):
    out = eng.rollout(T, out=ro)
    cnt += torch.bincount(out['mode'].flatten().to(torch.int64), minlength=32)[:32]
tot = int(cnt.sum())
pen = int(cnt[22:30].sum())
print(f'{400 * T} cycles from a reset: {100.0 * pen / tot:.1f} % of match-cycles in the shoot-out modes; by mode:', {i: round(100.0 * int(c) / tot, 1) for i, c in enumerate(cnt.tolist()) if c})
