"""GoToCenter surrogate task: hand-derived known answers from the text of
python_sample_soccer_env.py:17-255 (the module cannot be imported: it needs stable_baselines3
and runs argparse at import), oracle fp32 vs fp64, and GPU == oracle bit for bit."""
import ctypes as C
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
        L.s2dgo_set.argtypes = [C.c_void_p, C.c_int64, C.c_double, C.c_double, C.c_double, C.c_int]
        L.s2dgo_get.argtypes = [C.c_void_p, C.c_int, C.c_void_p]; L.s2dgo_obs.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        L.s2dgo_stats.argtypes = [C.c_void_p]; L.s2dgo_stats.restype = C.POINTER(C.c_ulonglong)
        L.s2dgo_destroy.argtypes = [C.c_void_p]
        self.n, self.prec = n, prec
        self.h = L.s2dgo_create(C.byref(cfg), n)

    def reset(self):
        self.L.s2dgo_reset(self.h, None)

    def step(self, a=None):
        if a is None:
            self.L.s2dgo_step(self.h, None)
        else:
            a = np.ascontiguousarray(a, dtype=np.float64)
            self.L.s2dgo_step(self.h, a.ctypes.data)

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


@pytest.mark.gpu
@pytest.mark.parametrize('continuous', [0, 1])
def test_gpu_matches_oracle_bit_for_bit(continuous):
    torch = pytest.importorskip('torch')
    from soccer2d_amd.gtc import GoToCenterVecEnv
    n = 5000
    env = GoToCenterVecEnv(n, 'cuda:0', continuous=continuous, max_steps=50)
    orc = GtcOracle(gtc_cfg(continuous=continuous, max_steps=50), n, 'f32')
    env.reset(); orc.reset()
    rs = np.random.RandomState(1)
    for t in range(120):
        if t % 2:
            act = rs.uniform(-1.2, 1.2, n).astype(np.float32) if continuous else rs.randint(0, 16, n)
            env.step(torch.as_tensor(act, device='cuda:0')); orc.step(act)
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


@pytest.mark.gpu
def test_gtc_drop_in_single_env():
    from python_sample_soccer_env import GoToCenterEnv
    env = GoToCenterEnv()
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
