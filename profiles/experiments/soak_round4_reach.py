"""Round-4 fuzz soak of the reach-ball engine against its C oracle: random ServerParam / task settings per seed (the generator of
tests/test_gpu_parity.py::test_random_server_parameters_parity, more seeds, more envs, longer runs), every kernel family in turn --
the four-wave pipeline, the unified kernel, s2d_step, s2d_step_k, masked s2d_reset -- with the whole record and every state word compared.
Usage (GPU box, repo root): [SOAK_N=65536,600000 SOAK_T=64] python profiles/experiments/soak_round4_reach.py [seconds]"""
import os, sys, time
sys.path.insert(0, 'gym-soccer-2d-env_amd'); sys.path.insert(0, 'tests')
import numpy as np
import torch
import test_gpu_parity as P

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 300.0
t_start, seed, runs = time.time(), 1000, 0
while time.time() - t_start < budget:
    rs = np.random.RandomState(seed)
    server = dict(
        player_decay=float(rs.uniform(0.2, 0.7)), ball_decay=float(rs.uniform(0.85, 0.99)),
        player_speed_max=float(rs.uniform(0.3, 1.2)), player_accel_max=float(rs.uniform(0.2, 1.0)),
        ball_speed_max=float(rs.uniform(1.0, 3.0)), player_size=float(rs.uniform(0.2, 2.5)), ball_size=float(rs.uniform(0.05, 0.5)),
        dash_power_rate=float(rs.uniform(0.003, 0.012)), side_dash_rate=float(rs.uniform(0.2, 0.6)),
        back_dash_rate=float(rs.uniform(0.4, 0.8)), dash_angle_step=float(rs.choice([0.0, 1.0, 22.5, 45.0])),
        min_dash_power=float(rs.choice([0.0, -100.0])), max_dash_power=float(rs.choice([100.0, 60.0])),
        stamina_max=float(rs.uniform(2000, 8000)), stamina_inc_max=float(rs.uniform(10, 60)),
        stamina_capacity=float(rs.choice([-1.0, 5000.0, 130600.0])), extra_stamina=float(rs.uniform(0, 100)),
        effort_min=float(rs.uniform(0.3, 0.8)), recover_min=float(rs.uniform(0.3, 0.7)),
        collision_vel_rate=float(rs.uniform(-0.5, -0.05)), player_rand=float(rs.uniform(0, 0.2)), ball_rand=float(rs.uniform(0, 0.1)))
    if rs.rand() < 0.4:
        server = {}                                             # the stock parameters: the table-driven fast paths
    mode = int(rs.randint(3))
    kw = dict(server=server, max_steps=int(rs.choice([3, 17, 60, 200])), min_distance_to_ball=float(rs.uniform(0.5, 8.0)),
              change_ball_velocity=bool(rs.randint(2)), change_ball_position=bool(rs.randint(2)),
              ball_position_x=float(rs.uniform(-20, 20)), ball_position_y=float(rs.uniform(-10, 10)),
              ball_speed=float(rs.uniform(0, 2.5)), ball_direction=float(rs.uniform(-180, 180)),
              use_continuous_action=mode != 0, use_turning=mode == 2, action_space_size=int(rs.choice([3, 8, 16, 36])),
              noise=bool(rs.randint(2)), seed=int(rs.randint(1, 2 ** 31)))
    n = int(rs.choice([int(v) for v in os.environ.get('SOAK_N', '63,257,1000,4097').split(',')]))
    ws = str(int(rs.randint(2)))
    os.environ['S2D_ROLLOUT_WS'] = ws
    eng, orc = P._engine(n, **dict(kw)), P._oracle(n, **dict(kw))
    eng.reset(); orc.reset()
    P.assert_state_same(eng, orc, f'seed {seed} reset')
    names = set()
    for T in (int(rs.choice([1, 2, 5, 64])), int(os.environ.get('SOAK_T', '256')), int(rs.choice([3, 100]))):
        out, ref = eng.rollout(T), orc.rollout(T)
        P._compare_rollout(out, ref, f'seed {seed} rollout T={T}')
        names.add(eng.kernel_name().split('<')[0])
    P.assert_state_same(eng, orc, f'seed {seed} after rollouts')
    for t in range(40):                                         # the per-step API with caller actions
        a = P._random_actions(rs, kw, n)
        obs, rew, done, res = eng.step(torch.as_tensor(a, device='cuda:0'))
        o_obs, o_rew, o_done, o_res = orc.step(a)
        P.assert_same(obs, o_obs, f'seed {seed} step {t} obs'); P.assert_same(rew, o_rew, f'seed {seed} step {t} reward')
        P.assert_same(done, o_done, f'seed {seed} step {t} done'); P.assert_same(res, o_res, f'seed {seed} step {t} result')
    for k in (int(rs.choice([1, 2, 3])), int(rs.choice([4, 7, 16])), int(rs.choice([2, 5, 33]))):   # s2d_step_k, caller actions and in-kernel policy
        if rs.rand() < 0.5:
            a = np.stack([P._random_actions(rs, kw, n) for _ in range(k)])
            out, ref = eng.step_k(k, torch.as_tensor(a, device='cuda:0')), orc.rollout(k, a)
        else:
            out, ref = eng.step_k(k), orc.rollout(k)
        P._compare_rollout(out, ref, f'seed {seed} step_k k={k}')
        P.assert_state_same(eng, orc, f'seed {seed} after step_k k={k}')
    for t in range(10):                                         # masked resets between steps
        eng.step(None); orc.step(None)
        m = (rs.rand(n) < 0.3).astype(np.uint8)
        P.assert_same(eng.reset(torch.as_tensor(m, device='cuda:0')), orc.reset(m), f'seed {seed} masked reset {t}')
    for t in range(10):
        eng.step(None); orc.step(None)
    P.assert_state_same(eng, orc, f'seed {seed} after steps')
    P.assert_same(eng.obs, orc.obs(), f'seed {seed} obs')
    print(f'ok seed {seed} n={n:5d} ws={ws} mode={mode} noise={int(kw["noise"])} stock_server={int(not server)} max_steps={kw["max_steps"]:3d} '
          f'kernels {sorted(names)} episodes {eng.stats[1:4].tolist()}', flush=True)
    seed += 1; runs += 1
print(f'round-4 reach-ball fuzz soak ok: {runs} random configurations, every record and state word equal')
