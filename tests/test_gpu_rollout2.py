"""Parity tests of the two-envs-per-lane rollout pipeline (csrc/s2d_rollout2.hip, round 4) through the C ABI: the HIP engine
against the CPU oracle on the same seeded inputs, BIT-EXACT for every integer and every fp32 word.  The kernel is chosen by
s2d_rollout for batches that are a multiple of 128 envs with a complete record; every test asserts that it is the one that ran
(s2d_kernel_name).  What one step is: Soccer2DEnv.step, /root/reference/soccer_2d_env.py:226-269, with the ReachBallEnv hooks
(sample_environments/reach_ball_env.py:53-161).
"""
import numpy as np
import pytest

import oracle as O
from test_gpu_parity import CONFIGS, _compare_rollout, _engine, _oracle, assert_same, assert_state_same

pytestmark = pytest.mark.gpu

torch = pytest.importorskip('torch')

WS2 = 's2d_reach_rollout_ws2_kernel'


@pytest.fixture(autouse=True)
def _two_envs_per_lane(monkeypatch):
    monkeypatch.setenv('S2D_ROLLOUT_E', '2')               # read by s2d_create: opt-in for the 128-env pipeline


def _ran_ws2(eng):
    assert eng.kernel_name().startswith(WS2), eng.kernel_name()


@pytest.mark.parametrize('name', list(CONFIGS))
def test_ws2_every_config_random_policy(name):
    """Every task configuration of the parity table (discrete / continuous / turning actions, fixed ball, noise on, no auto-reset
    with collisions, free dash angle): in-kernel Philox policy, a long launch, then launches shorter than the pipeline."""
    kw = CONFIGS[name]
    n = 128 * 5
    eng, orc = _engine(n, **dict(kw)), _oracle(n, **dict(kw))
    eng.reset(); orc.reset()
    for T in (130, 3, 1, 2, 40):
        out, ref = eng.rollout(T), orc.rollout(T)
        _ran_ws2(eng)
        _compare_rollout(out, ref, f'{name} T={T}')
        assert_state_same(eng, orc, f'{name} T={T}')
        assert_same(eng.obs, orc.obs(), f'{name} T={T} last obs')
        assert_same(eng.reward, orc.reward(), f'{name} T={T} last reward')
        assert_same(eng.done, orc.done(), f'{name} T={T} last done')
        assert_same(eng.result, orc.result(), f'{name} T={T} last result')
        assert_same(eng.action_dir, orc.action_dir(), f'{name} T={T} action_dir')
        assert_same(eng.action_cmd, orc.action_cmd(), f'{name} T={T} action_cmd')
        if kw.get('auto_reset', True):
            assert_same(eng.terminal_obs, orc.terminal_obs(), f'{name} T={T} terminal obs')
    assert 'nt=0' in eng.kernel_name() and ('noise=1' in eng.kernel_name()) == bool(kw.get('noise', False))
    o1, r1, d1, s1 = eng.step(None); o2, r2, d2, s2 = orc.step(None)    # the per-step API continues from the pipeline's state
    assert_same(o1, o2, f'{name} step after'); assert_same(r1, r2, f'{name} step after reward')
    st = eng.stats.cpu().numpy()
    assert list(st[:4]) == list(orc.stats()[:4].astype(np.int64))


@pytest.mark.parametrize('name,dtype', [('dqn-discrete16', np.int32), ('dqn-discrete16', np.int64), ('continuous1', np.float32),
                                        ('turning4', np.float32), ('noise-on', np.int64)])
def test_ws2_caller_actions(name, dtype):
    """[T][N] caller actions in every layout of include/s2d.h (loaded as pairs by the policy wave)."""
    kw = CONFIGS[name]
    n, T = 256, 64
    eng, orc = _engine(n, **dict(kw)), _oracle(n, **dict(kw))
    eng.reset(); orc.reset()
    rs = np.random.RandomState(5)
    for rep in range(2):
        if kw.get('use_continuous_action', True):
            a = rs.uniform(-1.3, 1.3, (T, n, 4 if kw.get('use_turning') else 1)).astype(np.float32)
        else:
            a = rs.randint(0, kw.get('action_space_size', 16), (T, n)).astype(dtype)
        out, ref = eng.rollout(T, torch.as_tensor(a, device='cuda:0')), orc.rollout(T, a)
        _ran_ws2(eng)
        _compare_rollout(out, ref, f'{name} {dtype.__name__} rep={rep}')
        assert_state_same(eng, orc, f'{name} {dtype.__name__} rep={rep}')


@pytest.mark.parametrize('noise', [False, True])
@pytest.mark.parametrize('case', ['every-step-ends', 'very-short', 'mixed-256', 'long-512'])
def test_ws2_many_episodes_per_launch(case, noise):
    """Three episodes per env are prepared before the loop; from the fourth on the simulating wave prepares inline and publishes
    through the slot used longest ago.  Same cases as test_many_episodes_per_launch_in_long_launches of the 64-env pipeline."""
    base = dict(use_continuous_action=False, action_space_size=16, change_ball_velocity=True, noise=noise)
    kw, n, Ts = {'every-step-ends': (dict(base, min_distance_to_ball=500.0, max_steps=200), 128 * 2, (70, 33)),
                 'very-short': (dict(base, max_steps=2), 128 * 3, (150, 64)),
                 'mixed-256': (dict(base, max_steps=200), 128 * 5, (256, 256)),
                 'long-512': (dict(base, max_steps=40), 128 * 2, (512, 200))}[case]
    eng, orc = _engine(n, **dict(kw)), _oracle(n, **dict(kw))
    eng.reset(); orc.reset()
    for T in Ts:
        out, ref = eng.rollout(T), orc.rollout(T)
        _ran_ws2(eng)
        _compare_rollout(out, ref, f'{case} T={T}')
        assert_state_same(eng, orc, f'{case} T={T}')
        assert_same(eng.obs, orc.obs(), f'{case} T={T} last obs')
        assert_same(eng.terminal_obs, orc.terminal_obs(), f'{case} T={T} terminal obs')
    o1, r1, d1, s1 = eng.step(None); o2, r2, d2, s2 = orc.step(None)
    assert_same(o1, o2, f'{case} step after'); assert_same(r1, r2, f'{case} step after reward')
    assert int(eng.episode.min()) >= (sum(Ts) if case == 'every-step-ends' else 2)


@pytest.mark.parametrize('noise', [False, True])
def test_ws2_dash_fast_path_and_its_fallback(noise):
    """A group takes the dash-only fast path when all its 128 envs sit on the stamina table with whole-degree body angles; a foreign
    state in either env of a lane sends the whole group through the generic loop -- both bit-equal to the oracle, in one launch."""
    kw = dict(use_continuous_action=False, action_space_size=16, change_ball_velocity=True, max_steps=4, noise=noise)
    n = 128 * 6
    eng, orc = _engine(n, **dict(kw)), _oracle(n, **dict(kw))
    eng.reset(); orc.reset()
    _compare_rollout(eng.rollout(24), orc.rollout(24), 'many episodes per launch')
    assert_state_same(eng, orc, 'after the short-episode launch')
    torch.cuda.synchronize()
    # foreign states in groups 1 (even env of a lane), 3 (odd env) and 5 (both envs of the last lane); groups 0, 2, 4 stay on the table
    edits = {128 + 6: dict(stamina=7777.0), 128 * 3 + 5: dict(player_body=33.5), 128 * 5 + 126: dict(effort=0.75, recovery=0.9),
             128 * 5 + 127: dict(stamina=1234.0)}
    for i, kv in edits.items():
        for k, v in kv.items():
            getattr(eng, k)[i] = v
        orc.set_env(i, **kv)
    for T in (7, 64):
        _compare_rollout(eng.rollout(T), orc.rollout(T), f'foreign states T={T}')
        _ran_ws2(eng)
        assert_state_same(eng, orc, f'foreign states T={T}')


@pytest.mark.parametrize('seed', list(range(6)))
def test_ws2_random_server_parameters(seed):
    """Random ServerParam / task settings (clamps that fire, big collision radius, no stamina capacity, free dash angles ...)."""
    rs = np.random.RandomState(700 + seed)
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
    mode = seed % 3
    kw = dict(server=server, max_steps=int(rs.randint(5, 40)), min_distance_to_ball=float(rs.uniform(0.5, 8.0)),
              change_ball_velocity=bool(rs.randint(2)), change_ball_position=bool(rs.randint(2)),
              ball_position_x=float(rs.uniform(-20, 20)), ball_position_y=float(rs.uniform(-10, 10)),
              ball_speed=float(rs.uniform(0, 2.5)), ball_direction=float(rs.uniform(-180, 180)),
              use_continuous_action=mode != 0, use_turning=mode == 2, action_space_size=int(rs.choice([3, 8, 16, 36])),
              noise=bool(seed & 1), seed=int(rs.randint(1, 2 ** 31)))
    n = 256
    eng, orc = _engine(n, **dict(kw)), _oracle(n, **dict(kw))
    eng.reset(); orc.reset()
    out, ref = eng.rollout(70), orc.rollout(70)
    _ran_ws2(eng)
    _compare_rollout(out, ref, f'cfg{seed}')
    for t in range(10):
        eng.step(None); orc.step(None)
    out, ref = eng.rollout(31), orc.rollout(31)
    _compare_rollout(out, ref, f'cfg{seed} second launch')
    assert_state_same(eng, orc, f'cfg{seed}')
    assert_same(eng.obs, orc.obs(), f'cfg{seed} obs')


def test_ws2_equals_the_64_env_pipeline_and_falls_back(monkeypatch):
    """The two pipelines are two schedules of the same arithmetic: records, state, counters identical over 300 cycles of 4 096
    envs (noise on).  And the 128-env pipeline is not chosen where it does not apply: a batch that is no multiple of 128, a
    record without observations."""
    from soccer2d_amd.engine import Engine, make_config
    kw = dict(CONFIGS['noise-on'])
    res = []
    for e in ('1', '2'):
        monkeypatch.setenv('S2D_ROLLOUT_E', e)
        eng = Engine(4096, 'cuda:0', cfg=make_config(**dict(kw)))
        eng.reset()
        out = [eng.rollout(T) for T in (200, 100)]
        torch.cuda.synchronize()
        res.append((eng, out))
    (a, oa), (b, ob) = res
    assert a.kernel_name().startswith('s2d_reach_rollout_ws_kernel<') and b.kernel_name().startswith(WS2)
    for x, y in zip(oa, ob):
        for k in ('obs', 'action', 'reward', 'done', 'result'):
            assert torch.equal(x[k], y[k]), k
    for f in O.STATE_FIELDS:
        assert torch.equal(getattr(a, f), getattr(b, f)), f
    assert torch.equal(a.obs, b.obs) and torch.equal(a.terminal_obs, b.terminal_obs) and torch.equal(a.stats, b.stats)
    monkeypatch.setenv('S2D_ROLLOUT_E', '2')
    odd = Engine(130, 'cuda:0', cfg=make_config(**dict(kw)))
    odd.reset(); odd.rollout(5)
    assert odd.kernel_name().startswith('s2d_reach_rollout_ws_kernel<')
    part = Engine(256, 'cuda:0', cfg=make_config(**dict(kw)))
    part.reset(); part.rollout(5, with_obs=False)
    assert part.kernel_name().startswith('s2d_reach_rollout_ws_kernel<')
    part.rollout(5)
    assert part.kernel_name().startswith(WS2)


@pytest.mark.parametrize('name', ['noise-on', 'continuous1', 'reference-default'])
def test_ws2_full_size_parity(name):
    """BASELINE.json configs[2] size (65 536 envs x 256 cycles = bench.py's launch) for the configurations the headline does not
    cover: noise on (the drop-in default of make_config), 1-D continuous actions, and the reference's own default kwargs
    (use_continuous_action=True, change_ball_velocity=False: reach_ball_env.py:26-36) -- full bit-exact comparison."""
    kw = dict(use_continuous_action=True, use_turning=False) if name == 'reference-default' else dict(CONFIGS[name])
    kw.pop('max_steps', None)                              # the reference's 200
    n, T = 65536, 256
    eng, orc = _engine(n, **dict(kw)), _oracle(n, **dict(kw))
    eng.reset(); orc.reset()
    out = eng.rollout(T)
    ref = orc.rollout(T)
    _ran_ws2(eng)
    assert 'nt=1' in eng.kernel_name()                     # 838 MB of record: streamed
    _compare_rollout(out, ref, f'full-size {name}')
    assert_state_same(eng, orc, f'full-size {name}')
    done = out['done'].cpu().numpy()
    assert (eng.cycle.cpu().numpy() == T + 1 + done.sum(axis=0)).all()
    st = eng.stats.cpu().numpy()
    assert st[0] == n * T and st[1:4].sum() == done.sum()
