"""The NumPy-vectorised CPU baseline (oracle/s2d_oracle_numpy.py) against
the float64 build of the C oracle: same algorithm, both libm/float64, written independently.
Integers bit-exact; floats within 1e-9 (libm sin/cos/atan2 are called on identical arguments,
the only freedom is the order of a few commutative additions)."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'oracle'))
import oracle as O  # noqa: E402
from s2d_oracle_numpy import NumpyReachBall, philox4x32_10  # noqa: E402

TOL = 1e-9
CASES = {
    'dqn-discrete16': dict(O.DQN_KWARGS),
    'continuous-1d': dict(change_ball_position=True, change_ball_velocity=True, use_continuous_action=True,
                          use_turning=False, max_steps=50),
    'fixed-ball': dict(change_ball_position=False, change_ball_velocity=False, ball_position_x=10, ball_position_y=-5,
                       ball_speed=1.5, ball_direction=30, use_continuous_action=False, action_space_size=8,
                       min_distance_to_ball=2.0, max_steps=40),
}


def _pair(n, kw, **extra):
    cfg = O.make_config(noise=0, **extra, **kw)        # the NumPy restatement has no noise model
    task = dict(O.TASK_DEFAULTS); task.update(kw)
    nb = NumpyReachBall(n, O.SERVER_DEFAULTS, task, seed=cfg.seed, env_id_offset=cfg.env_id_offset,
                        auto_reset=bool(cfg.auto_reset))
    return nb, O.OracleEngine(cfg, n, 'f64')


def _assert_state(nb, orc, tag):
    for f, arr in (('player_x', nb.px), ('player_y', nb.py), ('player_vx', nb.vx), ('player_vy', nb.vy),
                   ('player_body', nb.body), ('stamina', nb.stamina), ('effort', nb.effort), ('recovery', nb.recovery),
                   ('stamina_capacity', nb.capacity), ('ball_x', nb.bx), ('ball_y', nb.by), ('ball_vx', nb.bvx),
                   ('ball_vy', nb.bvy), ('prev_dist', nb.prev_dist), ('prev_angle', nb.prev_angle)):
        np.testing.assert_allclose(arr, orc.state(f), rtol=0, atol=TOL * max(1.0, float(np.abs(arr).max())), err_msg=f'{tag} {f}')
    for f, arr in (('step_number', nb.step_number), ('cycle', nb.cycle), ('policy_step', nb.policy_step)):
        assert (np.asarray(arr, dtype=np.int64) == orc.state(f).astype(np.int64)).all(), f'{tag} {f}'


def test_philox_matches_known_answer():
    # Random123 known-answer vector: counter = key = 0xffffffff...
    w = philox4x32_10(np.array([0xFFFFFFFF]), np.array([0xFFFFFFFF]), np.array([0xFFFFFFFF]), np.array([0xFFFFFFFF]),
                      0xFFFFFFFF, 0xFFFFFFFF)
    assert [int(x[0]) for x in w] == [0x408F276D, 0x41C83B0E, 0xA20BC7C6, 0x6D5451FD]


@pytest.mark.parametrize('name', list(CASES))
def test_numpy_port_matches_c_oracle(name):
    kw = CASES[name]
    n = 300
    nb, orc = _pair(n, kw, seed=1234, env_id_offset=77)
    np.testing.assert_allclose(nb.reset(), orc.reset(), atol=TOL)
    _assert_state(nb, orc, f'{name} reset')
    rs = np.random.RandomState(3)
    for t in range(120):
        if t % 3 == 0:
            a = None                                             # in-engine random policy
        elif not kw.get('use_continuous_action', True):
            a = rs.randint(0, kw.get('action_space_size', 16), n)
        else:
            a = rs.uniform(-1, 1, n).astype(np.float32)
        o1, r1, d1, res1 = nb.step(a)
        o2, r2, d2, res2 = orc.step(a if a is None or a.dtype != np.float32 else a.reshape(n, 1))
        assert (d1 == d2).all() and (res1 == res2).all(), f'{name} t={t}'
        np.testing.assert_allclose(o1, o2, atol=TOL, err_msg=f'{name} obs t={t}')
        np.testing.assert_allclose(r1, r2, atol=1e-8, err_msg=f'{name} reward t={t}')
    _assert_state(nb, orc, name)
    assert (nb.stats == orc.stats()[:4].astype(np.int64)).all()
    assert nb.stats[1:].sum() > 0


def test_numpy_port_masked_reset_and_no_auto_reset():
    kw = dict(O.DQN_KWARGS)
    n = 200
    nb, orc = _pair(n, kw, auto_reset=0)
    nb.reset(); orc.reset()
    rs = np.random.RandomState(5)
    for t in range(30):
        nb.step(None); orc.step(None)
        m = (rs.rand(n) < 0.3).astype(np.uint8)
        np.testing.assert_allclose(nb.reset(m), orc.reset(m), atol=TOL)
    _assert_state(nb, orc, 'masked')
