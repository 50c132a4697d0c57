"""GoToCenter surrogate task.  The oracle is PINNED by tests/golden/gtc.json, produced by
running the reference's own GoToCenterEnv (python_sample_soccer_env.py:46-255; tests/golden/make_golden.py):
fp64 build <= 1e-12, fp32 spec within the stated fp32 tolerances; the GPU is checked against the same fixture
and against the fp32 oracle bit for bit.  Hand-derived known answers are kept as a second, independent check."""
import ctypes as C
import json
import math
import os

import numpy as np
import pytest

import oracle as O
from soccer2d_amd import _capi
from soccer2d_amd.gtc import S2DGtcConfig


def gtc_cfg(**kw):
    c = S2DGtcConfig()
    c.abi_version, c.struct_bytes = _capi.S2D_ABI_VERSION, C.sizeof(S2DGtcConfig)
    c.x_min, c.x_max, c.y_min, c.y_max = -52.5, 52.5, -34.0, 34.0
    c.min_distance_to_center, c.max_steps, c.continuous = 5.0, 200, 0
    c.seed, c.env_id_offset, c.auto_reset = 0x5EED, 0, 1
    c.turn, c.use_turn, c.actor_out_size = 0, 0, 1
    for k, v in kw.items():
        setattr(c, k, v)
    return c


class GtcOracle:
    FIELDS = ('x', 'y', 'body', 'prev_distance', 'prev_angle_diff', 'step_count', 'episode', 'reward', 'done', 'result')

    def __init__(self, cfg, n, prec='f32'):
        O.build_oracle()
        path = os.path.join(O.ORACLE_DIR, '_build', f'libs2d_gtc_oracle_{prec}.so')
        if not os.path.exists(path):
            import subprocess
            subprocess.run(['make', '-C', O.ORACLE_DIR], check=True, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
        L = self.L = C.CDLL(path)
        L.s2dgo_create.argtypes = [C.POINTER(S2DGtcConfig), C.c_int64]; L.s2dgo_create.restype = C.c_void_p
        L.s2dgo_reset.argtypes = [C.c_void_p, C.c_void_p]; L.s2dgo_step.argtypes = [C.c_void_p, C.c_void_p]
        L.s2dgo_step_u.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.s2dgo_set.argtypes = [C.c_void_p, C.c_int64, C.c_double, C.c_double, C.c_double, C.c_int]
        L.s2dgo_get.argtypes = [C.c_void_p, C.c_int, C.c_void_p]; L.s2dgo_obs.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        L.s2dgo_stats.argtypes = [C.c_void_p]; L.s2dgo_stats.restype = C.POINTER(C.c_ulonglong)
        L.s2dgo_destroy.argtypes = [C.c_void_p]
        self.n, self.prec = n, prec
        self.h = L.s2dgo_create(C.byref(cfg), n)

    def reset(self):
        self.L.s2dgo_reset(self.h, None)

    def step(self, a=None, u=None):
        a = None if a is None else np.ascontiguousarray(a, dtype=np.float64)
        u = None if u is None else np.ascontiguousarray(u, dtype=np.float64)
        self.L.s2dgo_step_u(self.h, None if a is None else a.ctypes.data, None if u is None else u.ctypes.data)

    def set(self, i, x, y, body, step_count=0):
        self.L.s2dgo_set(self.h, i, x, y, body, step_count)

    def get(self, f):
        out = np.zeros(self.n)
        self.L.s2dgo_get(self.h, self.FIELDS.index(f), out.ctypes.data)
        if f in ('step_count', 'episode'):
            return out.astype(np.int32)
        if f in ('done', 'result'):
            return out.astype(np.uint8)
        return out.astype(np.float32 if self.prec == 'f32' else np.float64)

    def obs(self, terminal=False):
        out = np.zeros((self.n, 4))
        self.L.s2dgo_obs(self.h, int(terminal), out.ctypes.data)
        return out.astype(np.float32 if self.prec == 'f32' else np.float64)

    def stats(self):
        return np.ctypeslib.as_array(self.L.s2dgo_stats(self.h), shape=(8,)).astype(np.int64)


@pytest.mark.parametrize('prec,tol', [('f64', 1e-12), ('f32', 2e-6)])
def test_known_answers(prec, tol):
    o = GtcOracle(gtc_cfg(auto_reset=0), 1, prec)
    # (10, 0) facing the centre (180): action 8 -> dash_r = 0 -> one metre towards the centre
    o.set(0, 10.0, 0.0, 180.0)
    assert np.allclose(o.obs()[0], [0.0, 1.0, 10 / 52.5, 0.0], atol=tol)
    o.step([8])
    assert abs(o.get('x')[0] - 9.0) < tol * 10 and abs(o.get('y')[0]) < tol * 10
    assert abs(o.get('reward')[0] - 1.0) < tol * 10 and o.get('done')[0] == 0 and o.get('step_count')[0] == 1
    # action 0 -> dash_r = -1 -> direction body - 180 = 0 -> away from the centre; angle term 0
    o.step([0])
    assert abs(o.get('x')[0] - 10.0) < tol * 10 and abs(o.get('reward')[0] + 1.0) < tol * 10
    # action 12 -> dash_r = .5 -> direction wrap(180 + 90) = -90 -> y -= 1
    o.step([12])
    assert abs(o.get('y')[0] + 1.0) < tol * 10
    d0, d1 = 10.0, math.hypot(10, 1)
    a1 = abs(((180 - math.degrees(math.atan2(1, -10))) + 180) % 360 - 180)
    assert abs(o.get('reward')[0] - ((d0 - d1) + (0 - a1) / 180)) < tol * 20
    # Goal: strictly inside 5 m ; Out is a PENALTY here (unlike reach_ball) ; Timeout at step_count >= 200
    o.set(0, 5.9, 0.0, 180.0); o.step([8])
    assert o.get('result')[0] == 1 and abs(o.get('reward')[0] - 11.0) < tol * 20
    o.set(0, 52.0, 0.0, 0.0); o.step([8])
    assert o.get('result')[0] == 2 and abs(o.get('reward')[0] - (-1.0 - 10.0)) < tol * 20
    o.set(0, 30.0, 0.0, 180.0, step_count=199); o.step([8])
    assert o.get('result')[0] == 3 and abs(o.get('reward')[0] - (1.0 - 5.0)) < tol * 20
    o.set(0, 30.0, 0.0, 180.0, step_count=198); o.step([8])
    assert o.get('result')[0] == 0 and o.get('done')[0] == 0


def test_reset_distribution_and_f32_tracks_f64():
    a, b = GtcOracle(gtc_cfg(), 4096, 'f32'), GtcOracle(gtc_cfg(), 4096, 'f64')
    a.reset(); b.reset()
    x, y, body = a.get('x'), a.get('y'), a.get('body')
    assert -52.5 <= x.min() < -50 and 50 < x.max() <= 52.5 and -34 <= y.min() and y.max() <= 34 and -180 <= body.min() and body.max() < 180
    assert np.abs(a.obs() - b.obs()).max() < 2e-6
    rs = np.random.RandomState(0)
    for _ in range(60):
        act = rs.randint(0, 16, 4096)
        a.step(act); b.step(act)
        same = a.get('episode') == b.get('episode')
        assert same.mean() > 0.999
    assert np.abs(a.get('x') - b.get('x'))[same].max() < 1e-3


# ---------------------------------------------------------------------------------------------------------
# the reference-run fixture (tests/golden/gtc.json)
# ---------------------------------------------------------------------------------------------------------
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'gtc.json')
RESULT_CODE = {'': 0, 'Goal': 1, 'Out': 2, 'Timeout': 3}
# tolerances.  fp64 oracle, fp64 actions: everything the reference computes is float64 -> 1e-12.  fp64 oracle, float32
# actions: the reference's `dash_r * 180.0` is then a float32 product under the NumPy 2 that produced the fixture (a float64
# one under NumPy 1.x, which is what the oracle restates) -> 1e-4.  fp32 spec (oracle build and GPU): fp32 rounding
# accumulated over up to 200 steps.  Observations are float32 in the reference (`_get_obs`, :255).
TOL = {'f64': dict(pos=1e-12, ang=1e-12, reward=1e-12, obs=1.2e-7),
       'f64_f32act': dict(pos=1e-4, ang=1e-3, reward=1e-4, obs=2e-5),
       'f32': dict(pos=3e-4, ang=5e-3, reward=2e-4, obs=5e-5)}


def load_gold():
    with open(GOLD) as f:
        return json.load(f)


def group_sequences(gold):
    groups = {}
    for r in gold['sequences']:
        key = (r['mode'], json.dumps(r['attrs'], sort_keys=True), r['f32_actions'])
        groups.setdefault(key, []).append(r)
    return groups


def cfg_for(rows):
    kw, attrs = rows[0]['kwargs'], rows[0]['attrs']
    turn_mode = kw['turn'] and kw['continuous']
    return dict(continuous=int(kw['continuous']), turn=int(kw['turn']), use_turn=int(kw['use_turn']),
                actor_out_size=int(kw['actor_out_size']) if turn_mode else 1, auto_reset=0,
                x_max=attrs.get('x_max', 52.5), min_distance_to_center=attrs.get('min_distance_to_center', 5.0),
                max_steps=attrs.get('max_steps', 200))


def replay_group(rows, make, tol):
    """make(cfg_kwargs, n) -> object with set_states(list of start dicts), step(a[n][adim], u[n] or None) and
    read() -> dict of arrays (x, y, body, step_count, prev_distance, prev_angle_diff, reward, done, result, obs)."""
    kw = cfg_for(rows)
    n, adim = len(rows), kw['actor_out_size']
    eng = make(kw, n)
    eng.set_states([r['start'] for r in rows])
    got0 = eng.read()
    for i, r in enumerate(rows):
        if got0['obs'] is not None:                    # the observation of the injected start state (the oracle computes it in set())
            assert np.abs(got0['obs'][i] - np.array(r['obs0'])).max() <= tol['obs'], (r['mode'], r['note'], 'obs0')
    for t in range(max(len(r['steps']) for r in rows)):
        a = np.zeros((n, adim)); u = np.zeros(n)
        for i, r in enumerate(rows):
            if t < len(r['steps']):
                a[i, :] = r['steps'][t]['a'][:adim]
                u[i] = r['steps'][t]['u'] or 0.0
        eng.step(a, u if kw['use_turn'] else None)
        got = eng.read()
        for i, r in enumerate(rows):
            if t >= len(r['steps']):
                continue
            st, where = r['steps'][t], (r['mode'], r['note'], t)
            assert int(got['done'][i]) == int(st['done']) and int(got['result'][i]) == RESULT_CODE[st['result']], where
            assert int(got['step_count'][i]) == st['step_count'], where
            assert abs(got['x'][i] - st['x']) <= tol['pos'] and abs(got['y'][i] - st['y']) <= tol['pos'], where
            assert abs(got['body'][i] - st['body']) <= tol['ang'], where
            assert abs(got['prev_distance'][i] - st['prev_distance']) <= tol['pos'], where
            assert abs(got['prev_angle_diff'][i] - st['prev_angle_diff']) <= tol['ang'], where
            assert abs(got['reward'][i] - st['reward']) <= tol['reward'], where
            assert np.abs(got['obs'][i] - np.array(st['obs'])).max() <= tol['obs'], where


class OracleReplay:
    def __init__(self, kw, n, prec):
        self.o = GtcOracle(gtc_cfg(**kw), n, prec)

    def set_states(self, starts):
        for i, s in enumerate(starts):
            self.o.set(i, s['x'], s['y'], s['body'], s['step_count'])

    def step(self, a, u):
        self.o.step(a, u)

    def read(self):
        d = {f: self.o.get(f).astype(np.float64) for f in GtcOracle.FIELDS}
        d['obs'] = self.o.obs().astype(np.float64)
        return d


@pytest.mark.parametrize('prec', ['f64', 'f32'])
def test_oracle_pinned_by_reference_fixture(prec):
    gold = load_gold()
    assert len(gold['resets']) >= 64 and len(gold['sequences']) >= 64
    modes = {r['mode'] for r in gold['sequences']}
    assert modes == {'discrete', 'continuous', 'turn1', 'turn4', 'turn4_useturn'}
    for (mode, attrs, f32act), rows in group_sequences(gold).items():
        tol = TOL['f32'] if prec == 'f32' else (TOL['f64_f32act'] if f32act else TOL['f64'])
        replay_group(rows, lambda kw, n: OracleReplay(kw, n, prec), tol)
    # resets (:115-134): the reference's own draws are injected as the state; carry and observation must follow
    tol = TOL[prec]
    o = GtcOracle(gtc_cfg(auto_reset=0), len(gold['resets']), prec)
    for i, r in enumerate(gold['resets']):
        lo, hi, v = r['draws'][0]['low'], r['draws'][0]['high'], r['draws'][0]['v']
        assert (lo, hi) == (-52.5, 52.5) and lo <= v < hi and r['state']['x'] == v        # uniform(x_min, x_max) IS the state
        o.set(i, r['state']['x'], r['state']['y'], r['state']['body'], 0)
    assert np.abs(o.get('prev_distance') - np.array([r['state']['prev_distance'] for r in gold['resets']])).max() <= tol['pos']
    assert np.abs(o.get('prev_angle_diff') - np.array([r['state']['prev_angle_diff'] for r in gold['resets']])).max() <= tol['ang']
    assert np.abs(o.obs() - np.array([r['obs'] for r in gold['resets']])).max() <= tol['obs']


def test_turn_mode_random_policy_f32_tracks_f64():
    """The engine's own Philox policy / selection streams in the script's default mode (turn, use_turn, 4 outputs)."""
    kw = dict(continuous=1, turn=1, use_turn=1, actor_out_size=4, max_steps=60)
    a, b = GtcOracle(gtc_cfg(**kw), 2048, 'f32'), GtcOracle(gtc_cfg(**kw), 2048, 'f64')
    a.reset(); b.reset()
    body0 = a.get('body').copy()
    turned = np.zeros(2048, bool)
    for _ in range(40):
        a.step(None); b.step(None)
        turned |= (a.get('body') != body0) & (a.get('episode') == 1)
    same = a.get('episode') == b.get('episode')
    assert same.mean() > 0.995 and 0.9 < turned.mean() <= 1.0          # p(turn) ~ 0.5 per step
    assert np.abs(a.get('x') - b.get('x'))[same].max() < 1e-3


@pytest.mark.gpu
def test_gpu_matches_reference_fixture():
    """The DEVICE code against the reference-run fixture (through the C ABI, selection uniforms injected via s2d_gtc_step_u)."""
    torch = pytest.importorskip('torch')
    from soccer2d_amd.gtc import GoToCenterVecEnv

    class GpuReplay:
        def __init__(self, kw, n):
            self.env = GoToCenterVecEnv(n, 'cuda:0', **kw)

        def set_states(self, starts):
            e = self.env
            for name, key in (('x', 'x'), ('y', 'y'), ('body', 'body'), ('prev_distance', 'prev_distance'),
                              ('prev_angle_diff', 'prev_angle_diff')):
                getattr(e, name).copy_(torch.tensor([s[key] for s in starts], dtype=torch.float32))
            e.step_count.copy_(torch.tensor([s['step_count'] for s in starts], dtype=torch.int32))

        def step(self, a, u):
            e = self.env
            act = torch.as_tensor(a[:, 0] if not e.cfg.continuous else a, device='cuda:0')
            e.step(act, select_u=None if u is None else torch.as_tensor(u, device='cuda:0'))
            self.stepped = True

        def read(self):
            e = self.env
            torch.cuda.synchronize()
            d = {f: getattr(e, f).cpu().numpy().astype(np.float64) for f in
                 ('x', 'y', 'body', 'step_count', 'prev_distance', 'prev_angle_diff', 'reward', 'done', 'result')}
            d['obs'] = e.obs.cpu().numpy().astype(np.float64)
            if not getattr(self, 'stepped', False):
                d['obs'] = None                                     # injected start state: no kernel has produced its row yet
            return d

    gold = load_gold()
    for (mode, attrs, f32act), rows in group_sequences(gold).items():
        replay_group(rows, GpuReplay, TOL['f32'])


@pytest.mark.gpu
@pytest.mark.parametrize('mode', ['discrete', 'continuous', 'turn1', 'turn4', 'turn4_useturn'])
def test_gpu_matches_oracle_bit_for_bit(mode):
    torch = pytest.importorskip('torch')
    from soccer2d_amd.gtc import GoToCenterVecEnv
    n = 5000
    kw = {'discrete': dict(continuous=0), 'continuous': dict(continuous=1), 'turn1': dict(continuous=1, turn=1, actor_out_size=1),
          'turn4': dict(continuous=1, turn=1, actor_out_size=4),
          'turn4_useturn': dict(continuous=1, turn=1, use_turn=1, actor_out_size=4)}[mode]
    continuous, adim = kw['continuous'], kw.get('actor_out_size', 1)
    env = GoToCenterVecEnv(n, 'cuda:0', max_steps=50, **kw)
    orc = GtcOracle(gtc_cfg(max_steps=50, **kw), n, 'f32')
    env.reset(); orc.reset()
    rs = np.random.RandomState(1)
    for t in range(120):
        if t % 2:
            act = rs.uniform(-1.2, 1.2, (n, adim)).astype(np.float32) if continuous else rs.randint(0, 16, n)
            u = rs.uniform(0, 1, n).astype(np.float32) if (t % 4 == 1 and kw.get('use_turn')) else None
            env.step(torch.as_tensor(act, device='cuda:0'), select_u=None if u is None else torch.as_tensor(u, device='cuda:0'))
            orc.step(act, u)
        else:
            env.step(None); orc.step(None)
        torch.cuda.synchronize()
        assert np.array_equal(env.obs.cpu().numpy().view(np.int32), orc.obs().view(np.int32)), t
        assert np.array_equal(env.reward.cpu().numpy().view(np.int32), orc.get('reward').view(np.int32))
        assert np.array_equal(env.result.cpu().numpy(), orc.get('result'))
    for f in ('x', 'y', 'body', 'prev_distance', 'prev_angle_diff'):
        assert np.array_equal(getattr(env, f).cpu().numpy().view(np.int32), orc.get(f).view(np.int32)), f
    assert np.array_equal(env.episode.cpu().numpy(), orc.get('episode'))
    assert list(env.stats.cpu().numpy()[:4]) == list(orc.stats()[:4])
    ro = env.rollout(64)
    assert ro['obs'].shape == (64, n, 4) and bool(torch.isfinite(ro['obs']).all())
    assert ro['action'].shape == ((64, n, adim) if env.turn_mode else (64, n))


@pytest.mark.gpu
@pytest.mark.parametrize('kw', [dict(), dict(continuous=True, turn=True, actor_out_size=4, use_turn=True)])   # class / script defaults
def test_gtc_drop_in_single_env(kw):
    from python_sample_soccer_env import GoToCenterEnv
    env = GoToCenterEnv(**kw)
    obs, info = env.reset()
    assert obs.shape == (4,) and obs.dtype == np.float32 and info == {}
    total = 0
    for t in range(400):
        obs, r, term, trunc, info = env.step(env.action_space.sample())
        assert isinstance(r, float) and term == trunc and info['result'] in ('', 'Goal', 'Out', 'Timeout')
        if term:
            total += 1
            assert info['result']
            env.reset()
    assert total >= 1
    env.close()
