"""Round-4 soak of the 11v11 engine against its C oracle: a closed-loop caller policy that crowds the ball, kicks, tackles (half
of the tackles intentional fouls), catches and moves the goalies, over short matches so that every match goes through half time,
extra time, the penalty shoot-out and the automatic restart many times -- per rule variant of round 4 (stock rules on the engine's
own schedule, general kernel, PenaltyFoul_, pen_random_winner, IllegalDefense_, golden goal, heterogeneous types, noise) and per
seed.  Every word of the state is compared after every chunk of 50 cycles (the device replays the checker's commands fused).
Usage (GPU box, repo root): python profiles/experiments/soak_round4_rules.py [seconds]"""
import os, sys, time
sys.path.insert(0, 'gym-soccer-2d-env_amd'); sys.path.insert(0, 'tests')
import numpy as np
import torch
import test_gpu_match as T

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 420.0
n, chunk = 256, 50
sched = dict(half_time_cycles=120, nr_extra_halfs=2, extra_half_cycles=40, pen_before_setup_wait=3, pen_ready_wait=5, pen_taken_wait=25,
             pen_nr_kicks=2, pen_max_extra_kicks=2, after_goal_wait=10, announce_wait=8, drop_ball_time=40)
ids = [0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 0, 11, 12, 13, 14, 15, 16, 17, 1, 2, 3]
variants = [
    ('stock rules, own schedule', dict(sched)),
    ('stock rules, own schedule, noise', dict(sched, noise=True)),
    ('general: PenaltyFoul_ + coin', dict(sched, pen_allow_mult_kicks=0, pen_random_winner=1, noise=True)),
    ('general: illegal defense', dict(sched, illegal_defense_number=3, illegal_defense_duration=12, noise=True)),
    ('general: golden goal, no offside', dict(sched, golden_goal=1, use_offside=0)),
    ('general: heterogeneous types', dict(sched, hetero_seed=11, player_type_id=ids, noise=True)),
    ('general: no shoot-out, free kick faults off', dict(sched, penalty_shoot_outs=0, free_kick_faults=0, back_passes=0)),
]


def policy(orc, rs):
    x, y, body = orc.get('x'), orc.get('y'), orc.get('body')
    dx, dy = x[:, 22:23] - x[:, :22], y[:, 22:23] - y[:, :22]
    dist = np.hypot(dx, dy)
    rel = (np.degrees(np.arctan2(dy, dx)) - body[:, :22] + 540.0) % 360.0 - 180.0
    a = np.zeros((n, 22, 3), dtype=np.float32)
    u = rs.rand(n, 22)
    cmd = np.where(dist < 2.0, rs.choice([3, 4, 4, 3, 2], size=(n, 22)), np.where(u < 0.8, 1, rs.choice([1, 2, 4], size=(n, 22))))
    goalie = np.zeros((n, 22), bool); goalie[:, [0, 11]] = True
    cmd = np.where(goalie & (dist < 2.5), rs.choice([5, 5, 6, 3], size=(n, 22)), cmd)
    a[..., 0] = cmd
    a[..., 1] = np.where(cmd == 1, 100.0, np.where(cmd == 2, rel, np.where(cmd == 5, rel, rs.uniform(-100, 100, (n, 22)))))
    # kicks aim at the goal the team attacks, roughly: relative direction of the far goal's centre
    gx = np.where(np.arange(22)[None, :] < 11, 52.5, -52.5)
    goal_rel = (np.degrees(np.arctan2(-y[:, :22], gx - x[:, :22])) - body[:, :22] + 540.0) % 360.0 - 180.0
    a[..., 2] = np.where(cmd == 1, rel, np.where(cmd == 4, (rs.rand(n, 22) < 0.5).astype(np.float32),
                         np.where(cmd == 3, goal_rel + rs.uniform(-15, 15, (n, 22)), rs.uniform(-60, 60, (n, 22)))))
    return a


t_start, seed, done_runs = time.time(), 0, 0
while time.time() - t_start < budget:
    name, kw = variants[seed % len(variants)]
    eng, orc = T._pair(n, seed=0x5EED + 977 * seed, **dict(kw))
    rs = np.random.RandomState(100 + seed)
    modes, t0, cycles = set(), time.time(), 0
    while cycles < 1500 and time.time() - t_start < budget:
        acts = np.zeros((chunk, n, 22, 3), dtype=np.float32)
        for t in range(chunk):
            acts[t] = policy(orc, rs)
            orc.step(acts[t])
        out = eng.rollout(chunk, torch.as_tensor(acts, device='cuda:0'), with_obs=False)
        T.assert_match_same(eng, orc, f'{name} seed {seed} cycle {cycles + chunk}')
        modes |= set(np.unique(out['mode'].cpu().numpy()).tolist())
        cycles += chunk
    assert list(eng.stats.cpu().numpy()) == list(orc.stats())
    print(f'ok  {name:48s} seed {seed:2d}  {eng.kernel_name().split("<", 1)[1][:-1]:28s} {n} matches x {cycles} cycles, '
          f'{len(modes)} play modes seen {sorted(modes)}, stats {list(orc.stats())[:4]}  ({time.time() - t0:.0f} s)', flush=True)
    seed += 1; done_runs += 1
print(f'round-4 rules soak ok: {done_runs} runs, every state word equal after every {chunk}-cycle chunk')
