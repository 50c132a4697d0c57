"""Pin the CPU oracle against the golden vectors produced by RUNNING the reference's own
ReachBallEnv methods (tests/golden/make_golden.py): rows A2-A5 of SURVEY.md 8(a).

fp64 build: must reproduce the reference's float64 results to ~1e-12.
fp32 build (the HIP parity target): within the stated fp32 tolerance (2e-6 obs, 1e-5 reward
relative to magnitude) when fed the same float32-rounded inputs.
"""
import json
import math
import os

import numpy as np
import pytest

import oracle as O

G = os.path.join(os.path.dirname(__file__), 'golden')


def load(name):
    with open(os.path.join(G, name)) as f:
        return json.load(f)


CMD = {'dash': 1, 'turn': 2}


# ------------------------------------------------------------------ A2 action map
@pytest.mark.parametrize('prec,tol', [('f64', 0.0), ('f32', 2e-5)])
def test_action_map_discrete(prec, tol):
    g = load('action_map.json')
    for blk in g['discrete']:
        cfg = O.make_config(use_continuous_action=False, action_space_size=blk['n'])
        for row in blk['rows']:
            cmd, power, d = O.action_map(cfg, [row['a']], precision=prec)
            assert cmd == CMD[row['type']] and power == row['power']
            # the fixture's dir went through the protobuf float32 field Dash.relative_direction
            assert abs(float(np.float32(d)) - row['dir']) <= tol, (blk['n'], row, d)


def test_action_map_discrete_n16_exact():
    # reach_ball_env.py:84 with n=16: a=0 -> -180, a=8 -> 0, a=15 -> 157.5 (exact in fp32 too)
    cfg = O.make_config(use_continuous_action=False, action_space_size=16)
    for prec in ('f32', 'f64'):
        for a in range(16):
            assert O.action_map(cfg, [a], precision=prec)[2] == 22.5 * a - 180.0


@pytest.mark.parametrize('prec,tol', [('f64', 0.0), ('f32', 3e-5)])
def test_action_map_continuous(prec, tol):
    g = load('action_map.json')
    cfg = O.make_config(use_continuous_action=True, use_turning=False)
    for row in g['continuous']:
        cmd, power, d = O.action_map(cfg, [row['a']], precision=prec)
        assert cmd == 1 and power == 100.0
        assert abs(float(np.float32(d)) - row['dir']) <= tol
    # not clipped (reach_ball_env.py:81): a = 1.5 -> 270
    assert O.action_map(cfg, [1.5])[2] == 270.0


@pytest.mark.parametrize('prec,tol', [('f64', 0.0), ('f32', 3e-5)])
def test_action_map_turning(prec, tol):
    g = load('action_map.json')
    cfg = O.make_config(use_continuous_action=True, use_turning=True)
    n_turn = 0
    for row in g['turning']:
        a = np.asarray(row['a'])
        # skip rows whose uniform draw sits within fp32 rounding of the softmax threshold
        c = np.clip(a, -1, 1)
        p0 = math.exp(c[2]) / (math.exp(c[2]) + math.exp(c[0]))
        if prec == 'f32' and abs(row['u'] - p0) < 1e-6:
            continue
        cmd, power, d = O.action_map(cfg, a, u=row['u'], precision=prec)
        assert cmd == CMD[row['type']], row
        assert power == row['power']
        assert abs(float(np.float32(d)) - row['dir']) <= tol
        n_turn += cmd == 2
    assert 10 < n_turn < len(g['turning']) - 10   # both branches exercised


# ------------------------------------------------------------------ A3 observation
@pytest.mark.parametrize('prec,tol', [('f64', 1e-12), ('f32', 2e-6)])
def test_observation(prec, tol):
    g = load('obs.json')
    cfg = O.make_config()
    worst = 0.0
    for row in g['rows']:
        obs = O.observation(cfg, row['in'], precision=prec)
        ref = np.asarray(row['obs'])
        err = np.abs(obs - ref)
        # obs[0]/obs[7] are angles: +-180 deg wrap means +-1.0 / +-0.5 are the same direction
        if err[0] > 1.0:
            err[0] = abs(err[0] - 2.0)
        if err[7] > 0.5:
            err[7] = abs(err[7] - 1.0)
        worst = max(worst, err.max())
        assert err.max() <= tol, (row, obs)
    assert worst <= tol


def test_observation_survey_kat():
    cfg = O.make_config()
    obs = O.observation(cfg, [10, -5, 1, -1, -20, 12, 170])
    exp = [0.89145121, 0.94444444, -0.38095238, 0.35294118, 0.19047619, -0.14705882, 0.47140452, -0.125,
           0.33333333, -0.33333333]
    assert np.allclose(obs, exp, atol=5e-9)


# ------------------------------------------------------------------ A4 reward / done / result
RES = {None: 0, 'Goal': 1, 'Out': 2, 'Timeout': 3}


@pytest.mark.parametrize('prec,tol', [('f64', 1e-11), ('f32', 1e-4)])
def test_reward_sequences(prec, tol):
    seqs = load('reward.json')
    assert len(seqs) >= 64
    seen = set()
    for s in seqs:
        cfg = O.make_config(use_continuous_action=False, min_distance_to_ball=s['min_distance_to_ball'],
                            max_steps=s['max_steps'])
        carry = [0.0, 0.0]
        for row in s['rows']:
            done, rw, res = O.check_trainer(cfg, row['in'], row['step_number'], carry, precision=prec)
            assert done == row['done'], (s['note'], row)
            assert res == RES[row['result']], (s['note'], row)
            assert abs(rw - row['reward']) <= tol * max(1.0, abs(row['reward'])), (s['note'], row, rw)
            assert abs(carry[0] - row['carry_dist']) <= tol
            assert abs(carry[1] - row['carry_angle']) <= tol * 10
            seen.add((row['done'], row['result']))
    assert {(True, 'Goal'), (True, 'Out'), (True, 'Timeout'), (False, None)} <= seen


def test_reward_quirks():
    """'Out' ADDS 10 (reward -= -10.0, reach_ball_env.py:144); labels overwrite in order
    Goal < Out < Timeout while rewards accumulate; timeout is strict '>' (line 147)."""
    cfg = O.make_config(use_continuous_action=False)
    carry = [0.0, 0.0]
    O.check_trainer(cfg, [0, 0, 53, 0, 180], 0, carry)
    d, r, res = O.check_trainer(cfg, [0, 0, 53, 0, 180], 1, carry)
    assert d and res == 2 and abs(r - 10.0) < 1e-12
    d, r, res = O.check_trainer(cfg, [51, 0, 53, 0, 180], 201, carry)     # goal + out + timeout
    assert d and res == 3 and abs(r - ((53 - 2) + 10 + 10 - 5)) < 1e-9
    carry = [30.0, 0.0]
    assert O.check_trainer(cfg, [0, 0, 30, 0, 180], 200, carry) == (False, 0.0, 0)
    d, r, res = O.check_trainer(cfg, [0, 0, 30, 0, 180], 201, carry)
    assert d and res == 3 and r == -5.0


# ------------------------------------------------------------------ A5 reset sampler
@pytest.mark.parametrize('prec,tol', [('f64', 1e-7), ('f32', 2e-6)])
def test_reset_from_recorded_draws(prec, tol):
    rows = load('reset.json')
    n_multi = 0
    for row in rows:
        cfg = O.make_config(use_continuous_action=False, **row['cfg'])
        draws = [d['v'] for d in row['draws']]
        if prec == 'f32':
            # recorded random() values are float64; the fp32 engine draws 24-bit uniforms.
            draws = [float(np.float32(v)) if isinstance(v, float) else v for v in draws]
        rc, out, used = O.reset_from_draws(cfg, draws, precision=prec)
        if prec == 'f32' and used != len(draws):
            continue     # an accept/reject decision within fp32 rounding of the boundary
        assert rc == 0 and used == len(draws), row['cfg']
        n_multi += len(draws) > 7
        assert out[0] == row['ball_pos'][0] and out[1] == row['ball_pos'][1]
        assert out[4] == row['player_pos'][0] and out[5] == row['player_pos'][1]
        assert out[6] == row['player_body']
        # protobuf stored the velocity as float32: compare at float32 resolution
        assert abs(out[2] - row['ball_vel'][0]) <= tol * 3 and abs(out[3] - row['ball_vel'][1]) <= tol * 3
    assert n_multi > 10    # the rejection loop was exercised


def test_reset_grids_and_order():
    """Inclusive integer grids (reach_ball_env.py:173-178) and draw order player -> ball."""
    rows = load('reset.json')
    for row in rows:
        d = row['draws']
        assert [(x['a'], x['b']) for x in d[:3]] == [(-50, 50), (-30, 30), (0, 360)]
        if row['cfg'].get('change_ball_position', True):
            assert [(x['a'], x['b']) for x in d[3:5]] == [(-50, 50), (-30, 30)]
