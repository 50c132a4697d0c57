#!/usr/bin/env python3
"""Generate the golden fixtures in tests/golden/*.json by RUNNING the reference's own
Python (`/root/reference/sample_environments/reach_ball_env.py`) in this container.

Only the four pure hooks of ``ReachBallEnv`` are executed (an instance is made with
``__new__`` so that ``Soccer2DEnv.__init__`` -- which would spawn rcssserver / proxy /
gRPC processes, soccer_2d_env.py:71-95 -- never runs):

  A2  action_to_rpc_actions      reach_ball_env.py:53-85
  A3  state_to_observation       reach_ball_env.py:87-111
  A4  check_trainer_observation  reach_ball_env.py:113-161
  A5  trainer_reset_actions / get_ball_velocity  reach_ball_env.py:170-218

``gym`` and ``pyrusgeom`` are absent here; tests/golden/_standins.py restates the tiny
part of their published behaviour the path touches (see its docstring).  The fixtures
are DATA ONLY (inputs + expected outputs); no reference source is copied.

  GTC GoToCenterEnv.reset / step / _get_obs      python_sample_soccer_env.py:115-255
      (the module is imported with sys.argv = ['x'] and cwd = a temp dir: it runs argparse and creates
      a logs/ directory at import, :355-372; stable_baselines3 is a stand-in, only imported, never used)
  reset_dist.json: histograms of >= 100 000 runs of the reference's trainer_reset_actions (distributional
      fixture for the engine's Philox samplers)

Run (in the build container, where /root/reference exists):
    python tests/golden/make_golden.py
The GPU box never runs this; it only reads the committed JSON.
"""
import json
import logging
import math
import os
import random
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = os.environ.get("S2D_REFERENCE", "/root/reference")

sys.path.insert(0, HERE)
import _standins  # noqa: E402

_standins.install()
sys.path.insert(0, REF)

import service_pb2 as pb2  # noqa: E402  (reference's generated protobuf module)
from sample_environments.reach_ball_env import ReachBallEnv  # noqa: E402


def make_env(**kwargs):
    """ReachBallEnv without process spawning; attributes as __init__ would set them
    (reach_ball_env.py:26-51)."""
    env = ReachBallEnv.__new__(ReachBallEnv)
    lg = logging.getLogger("golden-null")
    lg.addHandler(logging.NullHandler())
    lg.propagate = False
    lg.setLevel(logging.CRITICAL)
    env.logger = lg
    env.change_ball_position = kwargs.get('change_ball_position', True)
    env.change_ball_velocity = kwargs.get('change_ball_velocity', False)
    env.ball_position_x = kwargs.get('ball_position_x', 0)
    env.ball_position_y = kwargs.get('ball_position_y', 0)
    env.ball_speed = kwargs.get('ball_speed', 0)
    env.ball_direction = kwargs.get('ball_direction', 0)
    env.min_distance_to_ball = kwargs.get('min_distance_to_ball', 5.0)
    env.max_steps = kwargs.get('max_steps', 200)
    env.use_continuous_action = kwargs.get('use_continuous_action', True)
    env.action_space_size = kwargs.get('action_space_size', 16)
    env.use_turning = kwargs.get('use_turning', False)
    if env.use_continuous_action:
        env.action_space = _standins.Box(-1, 1, shape=(4,) if env.use_turning else (1,), dtype=np.float32)
    else:
        env.action_space = _standins.Discrete(env.action_space_size)
    env.distance_to_ball = 0.0
    env.body_ball_angle_diff = 0.0
    env.step_number = 0
    return env


def player_state(bx, by, bvx, bvy, px, py, body):
    s = pb2.State()
    wm = s.world_model
    wm.ball.position.x, wm.ball.position.y = bx, by
    wm.ball.velocity.x, wm.ball.velocity.y = bvx, bvy
    wm.self.position.x, wm.self.position.y = px, py
    wm.self.body_direction = body
    return s


def trainer_state(bx, by, px, py, body):
    s = pb2.State()
    wm = s.world_model
    wm.ball.position.x, wm.ball.position.y = bx, by
    p = wm.teammates.add()
    p.position.x, p.position.y = px, py
    p.body_direction = body
    return s


def action_fields(a):
    kind = a.WhichOneof('action')
    if kind == 'dash':
        return {'type': 'dash', 'power': a.dash.power, 'dir': a.dash.relative_direction}
    if kind == 'turn':
        return {'type': 'turn', 'power': 0.0, 'dir': a.turn.relative_direction}
    raise AssertionError(kind)


# ----------------------------------------------------------------------------- A2
def gen_action_map():
    out = {'discrete': [], 'continuous': [], 'turning': []}
    for n in (4, 8, 16, 32, 7):
        env = make_env(use_continuous_action=False, action_space_size=n)
        rows = []
        for a in range(n):
            for form in ('int', 'np0', 'np1'):
                arg = a if form == 'int' else (np.array(a) if form == 'np0' else np.array([a]))
                before = env.step_number
                act = env.action_to_rpc_actions(arg, None)
                assert env.step_number == before + 1
                f = action_fields(act)
                if form == 'int':
                    rows.append({'a': a, **f})
                else:
                    assert rows[-1]['dir'] == f['dir']
        out['discrete'].append({'n': n, 'rows': rows})
    env = make_env(use_continuous_action=True, use_turning=False)
    rng = random.Random(1234)
    vals = [-1.0, -0.5, 0.0, 0.25, 1.0, 1.5, -2.0] + [rng.uniform(-1, 1) for _ in range(40)]
    for v in vals:
        v32 = float(np.float32(v))
        act = env.action_to_rpc_actions(np.array([v32], dtype=np.float32), None)
        out['continuous'].append({'a': v32, **action_fields(act)})
    env = make_env(use_continuous_action=True, use_turning=True)
    rs = np.random.RandomState(99)
    for i in range(96):
        a = rs.uniform(-1.6, 1.6, size=4).astype(np.float32)   # beyond [-1,1]: clip is pinned
        u = float(rs.uniform())
        real_rand = np.random.rand
        np.random.rand = lambda u=u: u                         # inject the single uniform draw (line 71)
        try:
            act = env.action_to_rpc_actions(a, None)
        finally:
            np.random.rand = real_rand
        out['turning'].append({'a': [float(x) for x in a], 'u': u, **action_fields(act)})
    return out


# ----------------------------------------------------------------------------- A3
def gen_obs():
    env = make_env()
    rs = np.random.RandomState(7)
    rows = []
    specials = [
        (10.0, -5.0, 1.0, -1.0, -20.0, 12.0, 170.0),     # SURVEY appendix-B KAT
        (0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0),             # everything zero (th() of zero vector)
        (5.0, 5.0, 0.0, 0.0, 5.0, 5.0, 90.0),            # ball on player
        (1.0, 0.0, 0.0, 0.0, 0.0, 0.0, 270.0),           # body beyond 180 -> normalised
        (1.0, 0.0, 0.0, 0.0, 0.0, 0.0, -200.0),
        (-3.0, 4.0, -2.0, 0.5, 52.0, -33.5, 45.0),
        (-52.5, 34.0, 3.0, 0.0, 52.5, -34.0, -179.0),
        (0.0, 10.0, 0.0, -3.0, 0.0, -10.0, 179.0),
        (20.0, 0.0, 0.1, 0.1, -20.0, 0.0, 400.0),        # fmod branch
        (20.0, 1.0, -0.1, 0.0, 25.0, 1.0, -725.0),
    ]
    for i in range(256):
        if i < len(specials):
            v = specials[i]
        else:
            v = (rs.uniform(-55, 55), rs.uniform(-36, 36), rs.uniform(-3, 3), rs.uniform(-3, 3),
                 rs.uniform(-55, 55), rs.uniform(-36, 36), rs.uniform(-180, 180))
            if i % 17 == 0:
                v = v[:2] + (0.0, 0.0) + v[4:]
        s = player_state(*v)
        wm = s.world_model
        inp = [wm.ball.position.x, wm.ball.position.y, wm.ball.velocity.x, wm.ball.velocity.y,
               wm.self.position.x, wm.self.position.y, wm.self.body_direction]   # float32-rounded by protobuf
        obs = env.state_to_observation(s)
        assert obs.dtype == np.float64 and obs.shape == (10,)
        rows.append({'in': [float(x) for x in inp], 'obs': [float(x) for x in obs]})
    return {'layout': ['bx', 'by', 'bvx', 'bvy', 'px', 'py', 'body'], 'rows': rows}


# ----------------------------------------------------------------------------- A4
def gen_reward():
    rs = np.random.RandomState(11)
    seqs = []

    def run(env, frames, note):
        """frames: list of (bx,by,px,py,body, force_step_number or None)"""
        rows = []
        # abs_reset path (reach_ball_env.py:163-168): first call seeds the carry, output discarded
        for k, fr in enumerate(frames):
            bx, by, px, py, body, force = fr
            if k > 0:
                if force is not None:
                    env.step_number = force - 1
                env.action_to_rpc_actions(0, None)          # increments step_number (line 55)
            s = trainer_state(bx, by, px, py, body)
            t = s.world_model.teammates[0]
            inp = [s.world_model.ball.position.x, s.world_model.ball.position.y,
                   t.position.x, t.position.y, t.body_direction]
            done, reward, info = env.check_trainer_observation(s)
            rows.append({'in': [float(x) for x in inp], 'step_number': env.step_number,
                         'done': bool(done), 'reward': float(reward), 'result': info['result'],
                         'carry_dist': float(env.distance_to_ball),
                         'carry_angle': float(env.body_ball_angle_diff)})
        seqs.append({'note': note, 'min_distance_to_ball': env.min_distance_to_ball,
                     'max_steps': env.max_steps, 'rows': rows})

    # SURVEY appendix-B KAT
    run(make_env(use_continuous_action=False), [(10, -5, -20, 12, 170, None), (10, -5, -19.1, 12, 170, None)], 'survey-kat')
    # all 2^3 combinations of Goal / Out / Timeout on the second frame
    for goal in (0, 1):
        for out_ in (0, 1):
            for tmo in (0, 1):
                if out_:
                    px, py = 53.0, 10.0
                else:
                    px, py = 40.0, 10.0
                if goal:
                    bx, by = px + 1.5, py - 2.0
                else:
                    bx, by = px - 20.0, py + 3.0
                step = 201 if tmo else 57
                run(make_env(use_continuous_action=False),
                    [(bx - 0.3, by, px - 0.8, py, 30.0, None), (bx, by, px, py, 35.0, step)],
                    f'combo goal={goal} out={out_} timeout={tmo}')
    # strict '>' boundary of the timeout (step 200 is not a timeout, 201 is)
    for step in (199, 200, 201, 202):
        run(make_env(use_continuous_action=False),
            [(0, 0, 30, 0, 180, None), (0, 0, 29, 0, 180, step)], f'timeout-boundary step={step}')
    # exact-threshold cases: d == min_distance is NOT a goal; |x| == 52.5 is NOT out
    run(make_env(use_continuous_action=False), [(0, 0, 6, 0, 180, None), (0, 0, 5, 0, 180, None), (0, 0, 4.75, 0, 180, None)], 'goal-threshold')
    run(make_env(use_continuous_action=False), [(0, 0, 52, 0, 0, None), (0, 0, 52.5, 0, 0, None), (0, 0, 52.75, 0, 0, None)], 'out-threshold-x')
    run(make_env(use_continuous_action=False), [(0, 0, 0, -33.5, -90, None), (0, 0, 0, -34.0, -90, None), (0, 0, 0, -34.25, -90, None)], 'out-threshold-y')
    run(make_env(use_continuous_action=False, min_distance_to_ball=1.0, max_steps=5),
        [(0, 0, 3, 0, 180, None), (0, 0, 2, 0, 180, None), (0, 0, 1.5, 0, 170, None), (0, 0, 0.5, 0, 170, None)], 'kwargs min_dist=1 max_steps=5')
    run(make_env(use_continuous_action=False, max_steps=3),
        [(0, 0, 30, 0, 0, None)] + [(0, 0, 30 - k, 0, 10 * k, None) for k in range(1, 6)], 'kwargs max_steps=3 natural timeout')
    # random walks
    for i in range(48):
        env = make_env(use_continuous_action=False)
        bx, by = rs.uniform(-50, 50), rs.uniform(-30, 30)
        px, py, body = rs.uniform(-50, 50), rs.uniform(-30, 30), rs.uniform(-180, 180)
        frames = [(bx, by, px, py, body, None)]
        for k in range(int(rs.randint(3, 12))):
            bx += rs.uniform(-1, 1)
            by += rs.uniform(-1, 1)
            px += rs.uniform(-1.05, 1.05)
            py += rs.uniform(-1.05, 1.05)
            body = rs.uniform(-180, 180) if rs.rand() < 0.5 else body
            frames.append((bx, by, px, py, body, None))
        run(env, frames, f'random-walk {i}')
    return seqs


# ----------------------------------------------------------------------------- A5
class _Recorder:
    """Wraps a seeded random.Random; records every draw the reference makes."""

    def __init__(self, seed):
        self.r = random.Random(seed)
        self.draws = []

    def randint(self, a, b):
        v = self.r.randint(a, b)
        self.draws.append({'f': 'randint', 'a': a, 'b': b, 'v': v})
        return v

    def random(self):
        v = self.r.random()
        self.draws.append({'f': 'random', 'v': v})
        return v


def gen_reset():
    import sample_environments.reach_ball_env as mod
    rows = []
    configs = [
        dict(change_ball_position=True, change_ball_velocity=True),
        dict(change_ball_position=True, change_ball_velocity=False),
        dict(change_ball_position=False, change_ball_velocity=True, ball_position_x=40, ball_position_y=-25),
        dict(change_ball_position=False, change_ball_velocity=False, ball_position_x=10, ball_position_y=5,
             ball_speed=1.5, ball_direction=30),
        dict(change_ball_position=True, change_ball_velocity=True, max_steps=50),
    ]
    real_random = mod.random
    try:
        for ci, cfg in enumerate(configs):
            for seed in range(24):
                env = make_env(use_continuous_action=False, **cfg)
                env.step_number = 17
                rec = _Recorder(1000 * ci + seed)
                mod.random = rec
                acts = env.trainer_reset_actions()
                assert env.step_number == 0
                kinds = [a.WhichOneof('action') for a in acts]
                assert kinds == ['do_move_ball', 'do_move_player', 'do_recover'], kinds
                mb, mp = acts[0].do_move_ball, acts[1].do_move_player
                assert mp.our_side is True and mp.uniform_number == 1
                rows.append({'cfg': cfg, 'max_steps': env.max_steps, 'draws': rec.draws,
                             'ball_pos': [mb.position.x, mb.position.y],
                             'ball_vel': [mb.velocity.x, mb.velocity.y],       # float32 (protobuf field)
                             'player_pos': [mp.position.x, mp.position.y],
                             'player_body': mp.body_direction})
    finally:
        mod.random = real_random
    return rows


# ----------------------------------------------------------------------------- wire format
def gen_wire():
    """Bytes produced by the reference's generated protobuf classes (service_pb2) for messages on the
    path: a player `State`, `PlayerAction`s and the trainer reset actions.  Pins soccer2d_amd/wire.py."""
    out = {'states': [], 'player_actions': [], 'trainer_actions': []}
    rs = np.random.RandomState(5)
    for k in range(6):
        s = pb2.State()
        wm = s.world_model
        vals = dict(our_side=1, cycle=int(rs.randint(1, 6000)), game_mode_type=int(rs.choice([2, 3, 4, 7])),
                    left_team_score=int(rs.randint(0, 4)), right_team_score=int(rs.randint(0, 4)), stoped_cycle=0)
        wm.our_side = vals['our_side']; wm.cycle = vals['cycle']; wm.game_mode_type = vals['game_mode_type']
        wm.left_team_score = vals['left_team_score']; wm.right_team_score = vals['right_team_score']
        wm.our_team_score = vals['left_team_score']; wm.their_team_score = vals['right_team_score']
        wm.game_mode_side = 2 if k % 2 else 1
        f32 = lambda v: float(np.float32(v))
        me = dict(x=f32(rs.uniform(-50, 50)), y=f32(rs.uniform(-30, 30)), vx=f32(rs.uniform(-1, 1)), vy=f32(rs.uniform(-1, 1)),
                  side=1, uniform_number=int(rs.randint(1, 12)), body_direction=f32(rs.uniform(-180, 180)),
                  stamina=f32(rs.uniform(1000, 8000)), effort=f32(rs.uniform(0.6, 1)), recovery=f32(rs.uniform(0.5, 1)),
                  stamina_capacity=f32(rs.uniform(1e4, 1.3e5)), dist_from_ball=f32(rs.uniform(1, 60)),
                  angle_from_ball=f32(rs.uniform(-180, 180)))
        sf = wm.self
        sf.position.x, sf.position.y, sf.velocity.x, sf.velocity.y = me['x'], me['y'], me['vx'], me['vy']
        sf.side, sf.uniform_number, sf.body_direction = me['side'], me['uniform_number'], me['body_direction']
        sf.stamina, sf.effort, sf.recovery, sf.stamina_capacity = me['stamina'], me['effort'], me['recovery'], me['stamina_capacity']
        sf.dist_from_ball, sf.angle_from_ball = me['dist_from_ball'], me['angle_from_ball']
        ball = dict(x=f32(rs.uniform(-50, 50)), y=f32(rs.uniform(-30, 30)), vx=f32(rs.uniform(-3, 3)), vy=f32(rs.uniform(-3, 3)),
                    rel_x=f32(rs.uniform(-9, 9)), rel_y=f32(rs.uniform(-9, 9)), dist_from_self=f32(rs.uniform(1, 60)),
                    angle_from_self=f32(rs.uniform(-180, 180)))
        b = wm.ball
        b.position.x, b.position.y, b.velocity.x, b.velocity.y = ball['x'], ball['y'], ball['vx'], ball['vy']
        b.relative_position.x, b.relative_position.y = ball['rel_x'], ball['rel_y']
        b.dist_from_self, b.angle_from_self = ball['dist_from_self'], ball['angle_from_self']
        mates, opps = [], []
        for team, lst, side in ((wm.teammates, mates, 1), (wm.opponents, opps, 2)):
            for u in range(1, 1 + int(rs.randint(1, 4))):
                d = dict(x=f32(rs.uniform(-50, 50)), y=f32(rs.uniform(-30, 30)), vx=f32(rs.uniform(-1, 1)), vy=f32(rs.uniform(-1, 1)),
                         side=side, uniform_number=u, body_direction=f32(rs.uniform(-180, 180)), is_tackling=bool(u == 2),
                         is_goalie=bool(u == 1), type_id=int(0 if u == 1 else (3 * u + k) % 18))
                p = team.add()
                p.position.x, p.position.y, p.velocity.x, p.velocity.y = d['x'], d['y'], d['vx'], d['vy']
                p.side, p.uniform_number, p.body_direction, p.is_tackling = d['side'], d['uniform_number'], d['body_direction'], d['is_tackling']
                p.is_goalie, p.type_id = d['is_goalie'], d['type_id']
                lst.append(d)
        vals.update(self=me, ball=ball, teammates=mates, opponents=opps, our_team_score=vals['left_team_score'],
                    their_team_score=vals['right_team_score'], game_mode_side=2 if k % 2 else 1)
        out['states'].append({'fields': vals, 'hex': s.SerializeToString().hex()})
    for cmd, a, b in (('dash', 100.0, -22.5), ('dash', 55.5, 180.0), ('turn', 90.0, 0.0), ('turn', -12.25, 0.0),
                      ('kick', 80.0, 45.0), ('tackle', -30.0, 0.0), ('catch', 0.0, 0.0), ('move', -10.5, 3.25)):
        if cmd == 'dash':
            m = pb2.PlayerAction(dash=pb2.Dash(power=a, relative_direction=b))
        elif cmd == 'turn':
            m = pb2.PlayerAction(turn=pb2.Turn(relative_direction=a))
        elif cmd == 'kick':
            m = pb2.PlayerAction(kick=pb2.Kick(power=a, relative_direction=b))
        elif cmd == 'catch':
            m = pb2.PlayerAction(catch=pb2.Catch())
        elif cmd == 'move':
            m = pb2.PlayerAction(move=pb2.Move(x=a, y=b))
        else:
            m = pb2.PlayerAction(tackle=pb2.Tackle(power_or_dir=a, foul=False))
        out['player_actions'].append({'cmd': cmd, 'a': a, 'b': b, 'hex': m.SerializeToString().hex(),
                                      'list_hex': pb2.PlayerActions(actions=[m], ignore_preprocess=True).SerializeToString().hex()})
    env = make_env(use_continuous_action=False, change_ball_velocity=True)
    import sample_environments.reach_ball_env as mod
    real = mod.random
    mod.random = _Recorder(4242)
    try:
        for a in env.trainer_reset_actions():
            out['trainer_actions'].append({'kind': a.WhichOneof('action'), 'hex': a.SerializeToString().hex(),
                                           'fields': {'ball_pos': [a.do_move_ball.position.x, a.do_move_ball.position.y],
                                                      'ball_vel': [a.do_move_ball.velocity.x, a.do_move_ball.velocity.y],
                                                      'player_pos': [a.do_move_player.position.x, a.do_move_player.position.y],
                                                      'body': a.do_move_player.body_direction, 'unum': a.do_move_player.uniform_number,
                                                      'our_side': a.do_move_player.our_side}})
    finally:
        mod.random = real
    m = pb2.TrainerAction(do_change_mode=pb2.DoChangeMode(game_mode_type=pb2.GameModeType.PlayOn, side=pb2.Side.LEFT))   # soccer_2d_env.py:242
    out['trainer_actions'].append({'kind': 'do_change_mode', 'hex': m.SerializeToString().hex(), 'fields': {'mode': 2, 'side': 1}})
    return out


# ----------------------------------------------------------------------------- GoToCenterEnv (python_sample_soccer_env.py)
def import_gtc_module():
    """Import the reference's python_sample_soccer_env.py.  It parses sys.argv and creates ./logs/<stamp>/ at
    import (:355-372), so argv is patched and the cwd is a temp dir for the duration of the import."""
    import tempfile
    _standins.install_sb3()
    old_argv, old_cwd = sys.argv, os.getcwd()
    tmp = tempfile.mkdtemp(prefix='gtc_golden_')
    try:
        sys.argv = ['x']
        os.chdir(tmp)
        import python_sample_soccer_env as gm
    finally:
        sys.argv = old_argv
        os.chdir(old_cwd)
    for name in ('SampleRL', 'Train', 'Test'):
        lg = logging.getLogger(name)
        lg.handlers[:] = []
        lg.addHandler(logging.NullHandler())
        lg.propagate = False
        lg.setLevel(logging.CRITICAL)
    return gm


class _UniformRecorder:
    """np.random.uniform of a seeded RandomState, recording what the reference asked for and got."""

    def __init__(self, seed):
        self.rs, self.draws = np.random.RandomState(seed), []

    def __call__(self, low, high):
        v = float(self.rs.uniform(low, high))
        self.draws.append({'low': float(low), 'high': float(high), 'v': v})
        return v


def gen_gtc():
    gm = import_gtc_module()
    GoToCenterEnv = gm.GoToCenterEnv
    real_uniform, real_rand = np.random.uniform, np.random.rand
    out = {'numpy': np.__version__, 'resets': [], 'sequences': []}

    def snapshot(env):
        return {'x': float(env.x), 'y': float(env.y), 'body': float(env.body_angle_deg), 'step_count': int(env.step_count),
                'prev_distance': float(env.prev_distance), 'prev_angle_diff': float(env.prev_angle_diff)}

    def do_reset(env, seed):
        rec = _UniformRecorder(seed)
        np.random.uniform = rec
        try:
            obs, info = env.reset()
        finally:
            np.random.uniform = real_uniform
        assert info == {} and obs.dtype == np.float32 and obs.shape == (4,)
        return rec.draws, obs

    # ---- resets :115-134
    for i in range(96):
        env = GoToCenterEnv()
        draws, obs = do_reset(env, 7000 + i)
        assert [d['low'] for d in draws] == [-52.5, -34.0, -180.0]
        out['resets'].append({'draws': draws, 'state': snapshot(env), 'obs': [float(v) for v in obs]})

    # ---- step sequences :136-233
    MODES = {
        'discrete': dict(continuous=False, turn=False, actor_out_size=1, use_turn=False),
        'continuous': dict(continuous=True, turn=False, actor_out_size=1, use_turn=False),
        'turn1': dict(continuous=True, turn=True, actor_out_size=1, use_turn=False),
        'turn4': dict(continuous=True, turn=True, actor_out_size=4, use_turn=False),
        'turn4_useturn': dict(continuous=True, turn=True, actor_out_size=4, use_turn=True),   # the script's default (:356-359)
    }

    def run_sequence(mode, start, actions, us=None, note='', attrs=None, f32=False):
        """start = (x, y, body, step_count); returns the fixture row or None if a decision is too close to call in fp32."""
        kw = MODES[mode]
        env = GoToCenterEnv(**kw)
        do_reset(env, 1)
        for k, v in (attrs or {}).items():
            setattr(env, k, v)
        env.x, env.y, env.body_angle_deg, env.step_count = float(start[0]), float(start[1]), float(start[2]), int(start[3])
        env.prev_distance = np.sqrt(env.x ** 2 + env.y ** 2)                       # as reset does, :127-129
        env.prev_angle_diff = gm.diff_angle_deg_abs(env.body_angle_deg, gm.angle_to_point_deg(env.x, env.y, 0.0, 0.0))
        row = {'mode': mode, 'note': note, 'kwargs': kw, 'attrs': attrs or {}, 'f32_actions': bool(f32),
               'start': snapshot(env), 'obs0': [float(v) for v in env._get_obs()], 'steps': []}
        safe = True
        for t, a in enumerate(actions):
            u = None if us is None else us[t]
            if kw['continuous']:
                arg = np.asarray(a, dtype=np.float32 if f32 else np.float64).reshape(-1)
            else:
                arg = int(a)
            if u is not None:
                np.random.rand = lambda u=u: u
            try:
                obs, reward, done, done2, info = env.step(arg)
            finally:
                np.random.rand = real_rand
            assert done == done2 and obs.dtype == np.float32
            if mode == 'turn4_useturn':
                cl = np.clip(np.asarray(a, dtype=np.float64), -1, 1)
                p0 = np.exp(cl[3]) / (np.exp(cl[3]) + np.exp(cl[2]))
                safe &= abs(u - p0) > 1e-3
            d = float(np.hypot(env.x, env.y))
            edge = min(abs(d - env.min_distance_to_center), abs(env.x - env.x_min), abs(env.x - env.x_max),
                       abs(env.y - env.y_min), abs(env.y - env.y_max))
            if 'exact' not in note:
                safe &= edge > 2e-3
            row['steps'].append({'a': [float(v) for v in np.asarray(a, dtype=np.float64).reshape(-1)], 'u': u,
                                 'obs': [float(v) for v in obs], 'reward': float(reward), 'done': bool(done),
                                 'result': info['result'], **snapshot(env)})
            if done:
                break
        return row if safe else None

    def rand_action(rs, mode):
        if mode == 'discrete':
            return int(rs.randint(0, 16))
        n = MODES[mode]['actor_out_size']
        return [float(v) for v in rs.uniform(-1.3, 1.3, size=n)]          # beyond [-1, 1]: the clip is pinned

    rs = np.random.RandomState(31)
    for mode in MODES:
        got = 0
        while got < 16:
            start = (rs.uniform(-50, 50), rs.uniform(-32, 32), rs.uniform(-180, 180), 0)
            L = int(rs.randint(5, 60))
            acts = [rand_action(rs, mode) for _ in range(L)]
            us = [float(rs.uniform()) for _ in range(L)] if mode == 'turn4_useturn' else None
            row = run_sequence(mode, start, acts, us, note=f'random {got}', f32=(got % 4 == 3 and mode != 'discrete'))
            if row is not None:
                out['sequences'].append(row)
                got += 1
    seq = out['sequences']
    # terminal cases and the if / elif priority of :203-217  (Out > Goal > Timeout)
    seq.append(run_sequence('discrete', (5.9, 0.0, 180.0, 0), [8], note='goal'))
    seq.append(run_sequence('discrete', (6.0, 0.0, 180.0, 0), [8, 8], note='exact: d == 5 is not a goal, then goal'))
    seq.append(run_sequence('discrete', (52.0, 0.0, 0.0, 0), [8], note='out +x'))
    seq.append(run_sequence('discrete', (51.5, 0.0, 0.0, 0), [8, 8], note='exact: x == 52.5 is not out, then out'))
    seq.append(run_sequence('discrete', (0.0, -33.5, -90.0, 50), [8], attrs={'min_distance_to_center': 1.0}, note='out -y'))
    seq.append(run_sequence('discrete', (30.0, 0.0, 180.0, 198), [8, 8], note='timeout: step_count 199 no, 200 yes (>=)'))
    seq.append(run_sequence('discrete', (30.0, 0.0, 180.0, 199), [8], note='timeout at 200'))
    seq.append(run_sequence('discrete', (30.0, 0.0, 180.0, 0), [8] * 6, attrs={'max_steps': 4}, note='max_steps attr = 4'))
    seq.append(run_sequence('discrete', (5.9, 0.0, 180.0, 199), [8], note='priority: goal beats timeout'))
    seq.append(run_sequence('discrete', (52.0, 0.0, 0.0, 199), [8], note='priority: out beats timeout'))
    seq.append(run_sequence('discrete', (3.5, 0.0, 0.0, 199), [8], attrs={'x_max': 4.0}, note='priority: out beats goal and timeout'))
    seq.append(run_sequence('continuous', (20.0, 10.0, 45.0, 0), [[2.0], [-3.0], [1.0], [-1.0], [0.0]], note='clip'))
    # the use_turn mode: forced turns (u = 0) never move -> a natural 200-step timeout; forced dashes (u -> 1)
    seq.append(run_sequence('turn4_useturn', (20.0, 10.0, 45.0, 0), [[0.3, 0.05, -0.5, 0.5]] * 200, [0.0] * 200, note='turn only: natural timeout'))
    seq.append(run_sequence('turn4_useturn', (20.0, 10.0, 45.0, 0), [[0.0, 0.9, 0.5, -0.5]] * 40, [0.999] * 40, note='dash only'))
    seq.append(run_sequence('turn4_useturn', (-30.0, -20.0, -120.0, 0),
                            [[0.1 * k - 1.0, 0.7 - 0.1 * k, 1.0, 1.0] for k in range(20)], [0.25, 0.75] * 10, note='p0 = 0.5, alternating'))
    assert all(r is not None for r in seq)
    res = {r['steps'][-1]['result'] for r in seq}
    assert res >= {'', 'Goal', 'Out', 'Timeout'}, res
    assert any(st['u'] is not None and st['body'] != r['start']['body'] for r in seq for st in r['steps'])
    return out


# ----------------------------------------------------------------------------- distribution of the reset sampler
def gen_reset_dist(n_runs=120000):
    """Run the reference's trainer_reset_actions n_runs times with its own (seeded) `random` module and keep
    histograms: the integer grids of :173-181 and the accepted ball velocity of :202-212 (speed in 30 bins of 0.1,
    direction in 36 bins of 10 degrees, number of tries).  Pins the DISTRIBUTION of the engine's Philox samplers."""
    import sample_environments.reach_ball_env as mod

    class _Counting:
        def __init__(self, seed):
            self.r, self.tries = random.Random(seed), 0

        def randint(self, a, b):
            return self.r.randint(a, b)

        def random(self):
            self.tries += 1
            return self.r.random()

    env = make_env(use_continuous_action=False, change_ball_position=True, change_ball_velocity=True)
    rec = _Counting(20261004)
    real = mod.random
    mod.random = rec
    h = {'px': np.zeros(101, np.int64), 'py': np.zeros(61, np.int64), 'body': np.zeros(361, np.int64),
         'bx': np.zeros(101, np.int64), 'by': np.zeros(61, np.int64), 'speed': np.zeros(30, np.int64),
         'dir': np.zeros(36, np.int64), 'tries': np.zeros(16, np.int64)}
    try:
        for _ in range(n_runs):
            rec.tries = 0
            acts = env.trainer_reset_actions()
            mb, mp = acts[0].do_move_ball, acts[1].do_move_player
            h['px'][int(round(mp.position.x)) + 50] += 1
            h['py'][int(round(mp.position.y)) + 30] += 1
            h['body'][int(round(mp.body_direction))] += 1            # randint(0, 360), sent un-normalised (:175, :193)
            h['bx'][int(round(mb.position.x)) + 50] += 1
            h['by'][int(round(mb.position.y)) + 30] += 1
            vx, vy = float(mb.velocity.x), float(mb.velocity.y)
            sp = math.hypot(vx, vy)
            h['speed'][min(29, int(sp / 0.1))] += 1
            if sp > 0.0:
                deg = int(round(math.degrees(math.atan2(vy, vx)))) % 360          # the draw was a whole degree (:205)
                h['dir'][((deg + 5) % 360) // 10] += 1                            # bins centred on multiples of 10
            h['tries'][min(15, rec.tries - 1)] += 1
    finally:
        mod.random = real
    return {'runs': n_runs, 'seed': 20261004, 'kwargs': {'change_ball_position': True, 'change_ball_velocity': True, 'max_steps': 200},
            'bins': {'speed': '30 bins of 0.1 over [0, 3)', 'dir': '36 bins of 10 deg centred on 0, 10, ..., 350: whole degrees d with (d + 5) % 360 // 10 == bin (direction of the accepted velocity, rounded to the whole degree it was drawn as)',
                     'tries': 'number of candidates drawn, 1..15, last bin = 16 and more', 'body': 'randint(0, 360) as sent'},
            'hist': {k: v.tolist() for k, v in h.items()}}


def main():
    out = {
        'wire.json': gen_wire(),
        'action_map.json': gen_action_map(),
        'obs.json': gen_obs(),
        'reward.json': gen_reward(),
        'reset.json': gen_reset(),
        'reset_dist.json': gen_reset_dist(),
        'gtc.json': gen_gtc(),
    }
    for name, data in out.items():
        with open(os.path.join(HERE, name), 'w') as f:
            json.dump(data, f, indent=None, separators=(',', ':'))
            f.write('\n')
        print(name, os.path.getsize(os.path.join(HERE, name)), 'bytes')


if __name__ == '__main__':
    main()
