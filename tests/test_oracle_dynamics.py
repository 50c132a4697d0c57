"""Hand-derived known answers for the rcssserver dynamics restatement (SURVEY.md appendix B;
rows S/P -- EXT, parity-unpinned against a real rcssserver) and engine-level semantics of
reset / step / auto-reset (rows A1, A6, A7)."""
import numpy as np
import pytest

import oracle as O


def fresh(n=1, prec='f64', **kw):
    kw.setdefault('use_continuous_action', False)
    cfg = O.make_config(auto_reset=kw.pop('auto_reset', 0), noise=kw.pop('noise', 0),
                        seed=kw.pop('seed', 0x5EED), server=kw.pop('server', None), **kw)
    return O.OracleEngine(cfg, n, prec)


def place(e, i=0, **kw):
    base = dict(player_x=0, player_y=0, player_vx=0, player_vy=0, player_body=0, ball_x=40, ball_y=0,
                ball_vx=0, ball_vy=0, stamina=8000, effort=1, recovery=1, stamina_capacity=130600,
                prev_dist=0, prev_angle=0, step_number=0, cycle=0)
    base.update(kw)
    e.set_env(i, **base)


@pytest.mark.parametrize('prec,tol', [('f64', 1e-12), ('f32', 1e-6)])
def test_forward_dash_from_rest(prec, tol):
    # action 8 of 16 -> Dash(100, 0): accel .6; x = .6, 1.44, 2.376; v after decay .24, .336, .3744
    e = fresh(prec=prec, server=dict(dash_angle_step=0.0))
    place(e)
    xs, vs = [], []
    for _ in range(3):
        e.step(np.array([8], dtype=np.int32))
        xs.append(e.state('player_x')[0])
        vs.append(e.state('player_vx')[0])
    assert np.allclose(xs, [0.6, 1.44, 2.376], atol=tol)
    assert np.allclose(vs, [0.24, 0.336, 0.3744], atol=tol)
    assert e.state('player_y')[0] == 0 and e.state('cycle')[0] == 3 and e.state('step_number')[0] == 3
    for _ in range(60):
        e.step(np.array([8], dtype=np.int32))
    # fixed point: pre-decay speed 0.6/(1-0.4) = 1.0 < player_speed_max
    assert abs(e.state('player_vx')[0] - 0.4) < 1e-4 * (1 if prec == 'f64' else 10)


@pytest.mark.parametrize('a,acc', [(12, 0.24), (4, 0.24), (0, 0.36), (10, 0.42)])
def test_dash_direction_rates(a, acc):
    # side dash +-90 -> rate .4; back dash 180 -> .6; 45 deg -> .7   (dash_angle_step off)
    e = fresh(server=dict(dash_angle_step=0.0))
    place(e)
    e.step(np.array([a], dtype=np.int32))
    v = np.hypot(e.state('player_vx')[0], e.state('player_vy')[0]) / 0.4
    assert abs(v - acc) < 1e-12
    d = 22.5 * a - 180
    assert abs(e.action_dir()[0] - d) < 1e-12 and e.action_cmd()[0] == 1
    ang = np.degrees(np.arctan2(e.state('player_vy')[0], e.state('player_vx')[0]))
    assert abs(((ang - d) + 180) % 360 - 180) < 1e-9


def test_dash_angle_step_discretises():
    # default dash_angle_step = 1: 22.5 -> rint -> 22 (half-to-even), -157.5 -> -158
    e = fresh()
    place(e)
    e.step(np.array([9], dtype=np.int32))
    ang = np.degrees(np.arctan2(e.state('player_vy')[0], e.state('player_vx')[0]))
    assert abs(ang - 22.0) < 1e-9
    place(e)
    e.step(np.array([1], dtype=np.int32))
    ang = np.degrees(np.arctan2(e.state('player_vy')[0], e.state('player_vx')[0]))
    assert abs(ang + 158.0) < 1e-9


def test_stamina_one_cycle_and_thresholds():
    e = fresh()
    place(e)
    e.step(np.array([8], dtype=np.int32))
    assert e.state('stamina')[0] == 7945.0 and e.state('stamina_capacity')[0] == 130555.0
    assert e.state('effort')[0] == 1.0 and e.state('recovery')[0] == 1.0
    # run down: 55 net per cycle; thresholds at 2400 (recover_dec / effort_dec)
    n = 1
    while e.state('stamina')[0] > 2400.0:
        e.step(np.array([8], dtype=np.int32))
        n += 1
    assert n == 102           # crosses the 2400 thresholds at ~cycle 102
    # updateStamina tests the POST-dash stamina (7900 - 55 (n-1) <= 2400 from n = 101): two decrements
    assert abs(e.state('recovery')[0] - 0.996) < 1e-12 and abs(e.state('effort')[0] - 0.99) < 1e-12
    for _ in range(120):
        e.step(np.array([8], dtype=np.int32))
    st = e.state('stamina')[0]
    assert 0 <= st < 100      # exhausted: only extra_stamina + per-cycle recovery feed the dash
    assert e.state('effort')[0] == pytest.approx(0.6) and 0.5 <= e.state('recovery')[0] < 1


def test_ball_free_roll_and_speed_cap():
    e = fresh()
    place(e, ball_x=0, ball_y=0, ball_vx=2.0, ball_vy=-1.0, player_x=-40)
    for k in range(1, 6):
        e.step(np.array([0], dtype=np.int32))
        f = (1 - 0.94 ** k) / 0.06
        assert abs(e.state('ball_x')[0] - 2.0 * f) < 1e-12 and abs(e.state('ball_y')[0] + 1.0 * f) < 1e-12
    place(e, ball_x=0, ball_y=0, ball_vx=4.0, ball_vy=3.0, player_x=-40)     # |v| = 5 > ball_speed_max
    e.step(np.array([0], dtype=np.int32))
    assert abs(e.state('ball_x')[0] - 2.4) < 1e-12 and abs(e.state('ball_y')[0] - 1.8) < 1e-12


def test_turn_inertia():
    e = fresh(use_continuous_action=True, use_turning=True)
    # a = [turn_p, turn_a, dash_p, dash_a]; "turn" is selected when u < softmax(dash_p) (quirk):
    # dash_p = +1, turn_p = -1 -> p0 = .88; find the branch from action_cmd
    place(e, player_vx=0.4, player_body=10)
    got = set()
    for s in range(40):
        place(e, player_vx=0.4, player_body=10, cycle=s)
        e.step(np.array([[-1.0, 0.5, 1.0, 0.25]], dtype=np.float32))
        cmd = e.action_cmd()[0]
        got.add(cmd)
        if cmd == 2:   # Turn(90): body += 90 / (1 + 5 * 0.4) = 30
            assert abs(e.state('player_body')[0] - 40.0) < 1e-9 and e.action_dir()[0] == 90.0
        else:          # Dash(100, 45)
            assert e.action_dir()[0] == 45.0 and e.state('player_body')[0] == 10.0
    assert got == {1, 2}


def test_collision_player_ball():
    e = fresh(min_distance_to_ball=0.0)
    place(e, player_x=0, ball_x=0.5, ball_vx=-0.1)       # ball rolls towards the player
    e.step(np.array([8], dtype=np.int32))                # player dashes +x: v=.6 -> x=.6 ; ball x=.4 -> overlap
    px, bx = e.state('player_x')[0], e.state('ball_x')[0]
    assert abs(abs(bx - px) - 0.385) < 1e-12             # separated to exact contact
    assert abs((px + bx) / 2 - 0.5) < 1e-12              # about the midpoint of (.6, .4)
    assert bx < px                                       # order along the line preserved
    assert abs(e.state('player_vx')[0] - 0.6 * -0.1 * 0.4) < 1e-12
    assert abs(e.state('ball_vx')[0] - -0.1 * -0.1 * 0.94) < 1e-12


def test_reset_consumes_one_cycle():
    """A6: the first obs after reset is the POST-cycle state (ball moved by v, decayed once)."""
    e = fresh(n=64, change_ball_velocity=True)
    e.reset()
    assert (e.state('cycle') == 1).all() and (e.state('step_number') == 0).all()
    assert (e.state('stamina') == 8000).all() and (e.state('stamina_capacity') == 130600).all()
    px, py, body = e.state('player_x'), e.state('player_y'), e.state('player_body')
    assert (px == np.round(px)).all() and (np.abs(px) <= 50).all() and (np.abs(py) <= 30).all()
    assert (body == np.round(body)).all() and (body >= -180).all() and (body <= 180).all()
    # ball: p0 integer grid, first returned state p0 + v0, velocity 0.94 v0
    bx, bvx = e.state('ball_x'), e.state('ball_vx')
    p0 = bx - bvx / 0.94
    assert np.allclose(p0, np.round(p0), atol=1e-9)
    sp = np.hypot(e.state('ball_vx'), e.state('ball_vy')) / 0.94
    assert (sp < 3.0).all() and sp.max() > 0.5
    # carry seeded from the post-cycle state
    d = np.hypot(e.state('ball_x') - px, e.state('ball_y') - py)
    assert np.allclose(e.state('prev_dist'), d, atol=1e-12)
    # reset ball-velocity acceptance (reach_ball_env.py:207-211) with the literal 0.96
    tf = (1 - 0.96 ** 200) / 0.04
    v0x, v0y = e.state('ball_vx') / 0.94, e.state('ball_vy') / 0.94
    assert (np.abs(p0 + v0x * tf) <= 52.5 + 1e-9).all()
    assert (np.abs((e.state('ball_y') - v0y) + v0y * tf) <= 34 + 1e-9).all()


def test_autoreset_equals_step_then_reset():
    """Vectorised auto-reset == reference flow `step(); if done: reset()` (A1 + A6)."""
    n = 256
    a = fresh(n=n, prec='f32', auto_reset=1, change_ball_velocity=True, max_steps=25)
    b = fresh(n=n, prec='f32', auto_reset=0, change_ball_velocity=True, max_steps=25)
    a.reset(); b.reset()
    rs = np.random.RandomState(3)
    n_done = 0
    for t in range(80):
        act = rs.randint(0, 16, n).astype(np.int32)
        oa, ra, da, resa = a.step(act)
        ob, rb, db, resb = b.step(act)
        assert (da == db).all() and (resa == resb).all() and (ra == rb).all()
        term = ob.copy()
        if db.any():
            ob2 = b.reset(db)
            n_done += int(db.sum())
            assert (a.terminal_obs()[db.astype(bool)] == term[db.astype(bool)]).all()
            ob = ob2
        assert (oa == ob).all()
        for f in O.STATE_FIELDS:
            assert (a.state(f) == b.state(f)).all(), f
    assert n_done > n      # every env finished at least once (max_steps=25)
    assert (a.stats()[1:4].sum() == n_done)


def test_shard_invariance():
    """Philox counters use the GLOBAL env id: cutting the env range into shards does not
    change any trajectory."""
    n = 96
    kw = dict(prec='f32', auto_reset=1, change_ball_velocity=True, max_steps=30, noise=1)
    whole = fresh(n=n, **kw)
    whole.reset()
    parts = []
    for lo, hi in ((0, 32), (32, 96)):
        cfg = O.make_config(auto_reset=1, noise=1, env_id_offset=lo, use_continuous_action=False,
                            change_ball_velocity=True, max_steps=30)
        p = O.OracleEngine(cfg, hi - lo, 'f32')
        p.reset()
        parts.append(p)
    for t in range(70):
        whole.step(None)
        for p in parts:
            p.step(None)
    for f in O.STATE_FIELDS:
        assert (np.concatenate([p.state(f) for p in parts]) == whole.state(f)).all(), f
    assert (np.concatenate([p.obs() for p in parts]) == whole.obs()).all()


def test_noise_is_bounded_and_seeded():
    e1 = fresh(n=128, prec='f32', noise=1, change_ball_velocity=True, seed=1)
    e2 = fresh(n=128, prec='f32', noise=1, change_ball_velocity=True, seed=1)
    e3 = fresh(n=128, prec='f32', noise=1, change_ball_velocity=True, seed=2)
    for e in (e1, e2, e3):
        e.reset()
        for _ in range(5):
            e.step(None)
    assert (e1.state('player_x') == e2.state('player_x')).all()
    assert (e1.state('player_x') != e3.state('player_x')).any()
    sp = np.hypot(e1.state('player_vx'), e1.state('player_vy')) / 0.4
    assert (sp <= 1.05 * 1.1 + 1e-6).all()


def test_f32_tracks_f64_over_an_episode():
    """Stated tolerance: fp32 engine vs fp64 engine, same formulas: <= 1e-3 m over
    a 201-step episode."""
    n = 128
    kw = dict(auto_reset=0, change_ball_velocity=True, server=dict(dash_angle_step=0.0))
    a, b = fresh(n=n, prec='f32', **kw), fresh(n=n, prec='f64', **kw)
    a.reset(); b.reset()
    rs = np.random.RandomState(5)
    for t in range(201):
        act = rs.randint(0, 16, n).astype(np.int32)
        a.step(act); b.step(act)
    for f in ('player_x', 'player_y', 'ball_x', 'ball_y'):
        assert np.abs(a.state(f) - b.state(f)).max() < 1e-3, f
    assert (a.state('cycle') == b.state('cycle')).all() and (a.state('cycle') == 202).all()
