"""Parity tests proper: the HIP engine (through the C ABI of libs2d_hip.so) against the CPU
oracle on the same seeded inputs.  Bar: BIT-EXACT for every integer AND every fp32 word
(the engine and the fp32 oracle implement the same deterministic math spec), plus
size-independent properties at the BASELINE.json sizes.
"""
import numpy as np
import pytest

import oracle as O

pytestmark = pytest.mark.gpu

torch = pytest.importorskip('torch')


def _engine(n, **kw):
    from soccer2d_amd.engine import Engine, make_config
    server = kw.pop('server', None)
    kw.setdefault('noise', False)           # the product default is noise ON; these parity cases name it explicitly
    cfg = make_config(server_params=server, **kw)
    return Engine(n, 'cuda:0', cfg=cfg)


def _oracle(n, **kw):
    server = kw.pop('server', None)
    cfg = O.make_config(seed=kw.pop('seed', 0x5EED), env_id_offset=kw.pop('env_id_offset', 0),
                        auto_reset=int(kw.pop('auto_reset', True)), noise=int(kw.pop('noise', False)),
                        server=server, **kw)
    return O.OracleEngine(cfg, n, 'f32')


def bits(a):
    a = np.ascontiguousarray(a)
    if a.dtype == np.float32:
        return a.view(np.int32)
    return a


def assert_same(gpu_t, cpu_a, what):
    g = gpu_t.detach().cpu().numpy()
    assert g.shape == cpu_a.shape, (what, g.shape, cpu_a.shape)
    if not np.array_equal(bits(g), bits(cpu_a)):
        bad = np.argwhere(bits(g) != bits(cpu_a))
        i = tuple(bad[0])
        raise AssertionError(f"{what}: {len(bad)} of {g.size} words differ; first at {i}: gpu={g[i]!r} cpu={cpu_a[i]!r}")


def _compare_rollout(out, ref, tag):
    torch.cuda.synchronize()
    for k in ('obs', 'action', 'reward', 'done', 'result'):
        assert_same(out[k], ref[k], f'{tag} rollout.{k}')


def assert_state_same(eng, orc, tag=''):
    for f in O.STATE_FIELDS:
        assert_same(getattr(eng, f), orc.state(f), f'{tag} state.{f}')


def test_default_config_matches_test_table():
    import ctypes as C
    from soccer2d_amd import _capi
    lib = _capi.load_library()
    cfg = _capi.S2DConfig()
    lib.s2d_default_config(C.byref(cfg))
    ref = O.make_config()
    assert bytes(memoryview(cfg)) == bytes(memoryview(ref))


def test_device_math_bit_exact():
    """sincos/atan2/exp/norm/hypot/philox evaluated on the GPU == the oracle's fp32 spec."""
    import ctypes as C
    from soccer2d_amd import _capi
    lib = _capi.load_library()
    rs = np.random.RandomState(0)
    dev = 'cuda:0'

    def run(op, inp, out_shape, out_dtype=torch.float32):
        x = torch.as_tensor(inp, device=dev).contiguous()
        y = torch.zeros(out_shape, dtype=out_dtype, device=dev)
        n = out_shape[0]
        _capi.check(lib, lib.s2d_debug_eval(op, x.data_ptr(), y.data_ptr(), n, None), 's2d_debug_eval')
        torch.cuda.synchronize()
        return y.cpu().numpy()

    deg = np.concatenate([rs.uniform(-720, 720, 20000), np.arange(-720, 721, 0.5), 22.5 * np.arange(-32, 33)]).astype(np.float32)
    got = run(0, deg, (len(deg), 2))
    exp = np.array([O.sincos_deg(float(d)) for d in deg], dtype=np.float32)
    assert np.array_equal(bits(got), bits(exp))
    yx = rs.uniform(-110, 110, (20000, 2)).astype(np.float32)
    yx[:50] = np.round(yx[:50])
    yx[50:60] = 0
    got = run(1, yx, (len(yx),))
    exp = np.array([O.atan2_deg(float(y), float(x)) for y, x in yx], dtype=np.float32)
    assert np.array_equal(bits(got), bits(exp))
    x = rs.uniform(-3, 3, 5000).astype(np.float32)
    assert np.array_equal(bits(run(2, x, (len(x),))), bits(np.array([O.exp(float(v)) for v in x], dtype=np.float32)))
    x = rs.uniform(-1000, 1000, 5000).astype(np.float32)
    assert np.array_equal(bits(run(3, x, (len(x),))),
                          bits(np.array([O.lib('f32').s2do_norm_deg(float(v)) for v in x], dtype=np.float32)))
    got = run(5, yx, (len(yx),))
    exp = np.array([O.hypot(float(a), float(b)) for a, b in yx], dtype=np.float32)
    assert np.array_equal(bits(got), bits(exp))   # hypot spec = sqrtf(fmaf(x,x,y*y))
    true = np.sqrt(yx[:, 0].astype(np.float64) ** 2 + yx[:, 1].astype(np.float64) ** 2)
    assert (np.abs(got - true) <= 1.2e-7 * np.maximum(true, 1e-30)).all()
    # the device takes its square roots with a lean correctly-rounded sequence (sqrt_cr, s2d_device.h): every magnitude,
    # squares that are exact / just off a representable root, the 2^-96 hand-over to the general sequence, denormals, zero
    mag = 10.0 ** rs.uniform(-24, 18, 60000)
    wide = np.stack([mag * rs.uniform(-1, 1, 60000), mag * rs.uniform(-1, 1, 60000)], axis=1).astype(np.float32)
    k = np.arange(1, 2001, dtype=np.float32)
    exact = np.stack([k * 0.25, np.zeros_like(k)], axis=1)
    off = np.stack([np.nextafter(k, np.float32(np.inf)), np.zeros_like(k)], axis=1)
    edge = np.array([[0, 0], [2.0 ** -48, 0], [2.0 ** -48, 2.0 ** -48], [2.0 ** -49, 2.0 ** -49], [2.0 ** -50, 0], [1e-30, 1e-31],
                     [1e-20, 0], [3e-20, 4e-20], [1e-22, 0], [1e19, 1e19], [5, 0], [3, 4], [52.5, 34]], dtype=np.float32)
    for arr in (wide, exact, off, edge):
        got = run(5, arr, (len(arr),))
        exp = np.array([O.hypot(float(a), float(b)) for a, b in arr], dtype=np.float32)
        assert np.array_equal(bits(got), bits(exp)), arr[np.argwhere(bits(got) != bits(exp))[0, 0]]
    ck = rs.randint(0, 2 ** 32, (512, 6), dtype=np.uint64).astype(np.uint32)
    ck[0] = 0
    ck[1] = 0xffffffff
    ck[2] = [0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344, 0xa4093822, 0x299f31d0]
    got = run(4, ck.view(np.int32), (512, 4), torch.int32).view(np.uint32)
    exp = np.array([O.philox(list(map(int, r[:4])), list(map(int, r[4:]))) for r in ck], dtype=np.uint32)
    assert np.array_equal(got, exp)
    assert list(got[0]) == [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]


@pytest.mark.parametrize('n', [1, 63, 1000, 4096])
def test_reset_parity(n):
    kw = dict(use_continuous_action=False, change_ball_velocity=True)
    eng, orc = _engine(n, **kw), _oracle(n, **kw)
    obs = eng.reset()
    torch.cuda.synchronize()
    orc.reset()
    assert_same(obs, orc.obs(), 'reset obs')
    assert_state_same(eng, orc, 'reset')
    assert int(eng.cycle.min()) == 1 and int(eng.step_number.max()) == 0


def test_cooperative_reset_draw_equals_the_sequential_one():
    """reset_sample_coop (the whole wave shares the rejection loop of get_ball_velocity, reach_ball_env.py:202-212) must return
    what reset_sample returns, bit for bit: for full waves, sparse `need` patterns (the refill of a few lanes) and acceptance
    rates from ~1 (short travel) to a few per cent (a travel factor that puts most candidates outside the pitch)."""
    from soccer2d_amd import _capi
    lib = _capi.load_library()
    rs = np.random.RandomState(7)
    n = 256 * 64
    for travel, p_need in ((16.0, 1.0), (16.0, 0.15), (1.0, 1.0), (60.0, 1.0), (60.0, 0.05), (200.0, 0.5)):
        u = np.zeros((n, 4), np.uint32)
        u[:, 0] = rs.randint(1, 1 << 30, n)
        u[:, 1] = rs.uniform(size=n) < p_need
        u[:, 2] = rs.randint(0, 1 << 31)
        u[:, 3] = np.float32(travel).view(np.uint32)
        x = torch.as_tensor(u.view(np.float32), device='cuda:0').contiguous()
        y = torch.zeros((n, 14), dtype=torch.float32, device='cuda:0')
        _capi.check(lib, lib.s2d_debug_eval(9, x.data_ptr(), y.data_ptr(), n, None), 's2d_debug_eval')
        torch.cuda.synchronize()
        got = y.cpu().numpy()
        need = u[:, 1] != 0
        assert need.sum() > 0
        coop, seq = bits(got[need, :7]), bits(got[need, 7:])
        assert np.array_equal(coop, seq), (travel, p_need, int((coop != seq).any(axis=1).sum()), np.argwhere((coop != seq).any(axis=1))[:5].ravel())
        assert (got[need, 12] != 0).mean() > 0.5 or travel > 100      # velocities were actually drawn


CONFIGS = {
    'dqn-discrete16': dict(use_continuous_action=False, action_space_size=16, change_ball_velocity=True),
    'discrete7-fixed-ball': dict(use_continuous_action=False, action_space_size=7, change_ball_position=False,
                                 ball_position_x=10, ball_position_y=-5, ball_speed=1.5, ball_direction=30, max_steps=40),
    'continuous1': dict(use_continuous_action=True, use_turning=False, change_ball_velocity=True, max_steps=60),
    'turning4': dict(use_continuous_action=True, use_turning=True, change_ball_velocity=True, max_steps=60),
    'noise-on': dict(use_continuous_action=False, change_ball_velocity=True, noise=True, max_steps=50),
    'no-autoreset-collide': dict(use_continuous_action=False, auto_reset=False, min_distance_to_ball=0.0, max_steps=30,
                                 change_ball_velocity=True),
    'free-dash-angle': dict(use_continuous_action=True, use_turning=False, server=dict(dash_angle_step=0.0), max_steps=80),
}


def _random_actions(rs, cfg_kw, n):
    if not cfg_kw.get('use_continuous_action', True):
        return rs.randint(0, cfg_kw.get('action_space_size', 16), n).astype(np.int64)
    w = 4 if cfg_kw.get('use_turning', False) else 1
    return rs.uniform(-1.3, 1.3, (n, w)).astype(np.float32)


@pytest.mark.parametrize('name', list(CONFIGS))
def test_step_parity(name):
    """Per-step API, 250 steps with caller actions: every output and every state word equal."""
    kw = CONFIGS[name]
    n = 777
    eng, orc = _engine(n, **dict(kw)), _oracle(n, **dict(kw))
    eng.reset(); orc.reset()
    rs = np.random.RandomState(42)
    dones = 0
    for t in range(250):
        a = _random_actions(rs, kw, n)
        obs, rew, done, res = eng.step(torch.as_tensor(a, device='cuda:0'))
        o_obs, o_rew, o_done, o_res = orc.step(a)
        torch.cuda.synchronize()
        assert_same(done, o_done, f'{name} t={t} done')
        assert_same(res, o_res, f'{name} t={t} result')
        assert_same(rew, o_rew, f'{name} t={t} reward')
        assert_same(obs, o_obs, f'{name} t={t} obs')
        assert_same(eng.action_dir, orc.action_dir(), f'{name} t={t} action_dir')
        assert_same(eng.action_cmd, orc.action_cmd(), f'{name} t={t} action_cmd')
        if o_done.any() and kw.get('auto_reset', True):
            m = o_done.astype(bool)
            assert_same(eng.terminal_obs[torch.as_tensor(m, device='cuda:0')], orc.terminal_obs()[m], 'terminal_obs')
        dones += int(o_done.sum())
        if t % 50 == 49:
            assert_state_same(eng, orc, f'{name} t={t}')
    assert_state_same(eng, orc, name)
    assert dones > 0
    st = eng.stats.cpu().numpy()
    assert list(st[:4]) == list(orc.stats()[:4].astype(np.int64))
    assert st[0] == 250 * n and st[1:4].sum() == dones


@pytest.mark.parametrize('ws', ['0', '1'])
@pytest.mark.parametrize('n', [1, 63, 65, 130])
def test_rollout_parity_at_ragged_sizes(n, ws, monkeypatch):
    """Waves with one env, one missing env, one env more than a wave: the prepared episodes are drawn by the whole wave together
    (reset_sample_coop), lanes without an env included, and must not depend on how many lanes have one."""
    monkeypatch.setenv('S2D_ROLLOUT_WS', ws)
    kw = dict(use_continuous_action=False, change_ball_velocity=True, max_steps=9)
    eng, orc = _engine(n, **dict(kw)), _oracle(n, **dict(kw))
    eng.reset(); orc.reset()
    for T in (70, 3):                                      # several episodes per env and launch, then a short launch
        _compare_rollout(eng.rollout(T), orc.rollout(T), f'n={n} ws={ws} T={T}')
    assert_state_same(eng, orc, f'n={n} ws={ws}')


@pytest.mark.parametrize('ws', ['0', '1'])
@pytest.mark.parametrize('name', ['dqn-discrete16', 'turning4', 'noise-on'])
def test_random_policy_and_rollout_parity(name, ws, monkeypatch):
    """In-kernel Philox policy (S2D_ACT_RANDOM): per-step launches == one fused rollout launch
    (unified kernel ws=0 and wave-specialised kernel ws=1) == the oracle, bit for bit, including
    the emitted actions."""
    monkeypatch.setenv('S2D_ROLLOUT_WS', ws)
    kw = CONFIGS[name]
    n, T = 1000, 130
    a, b, orc = _engine(n, **dict(kw)), _engine(n, **dict(kw)), _oracle(n, **dict(kw))
    a.reset(); b.reset(); orc.reset()
    ref = orc.rollout(T)
    out = b.rollout(T)
    torch.cuda.synchronize()
    for k in ('obs', 'action', 'reward', 'done', 'result'):
        assert_same(out[k], ref[k], f'{name} rollout.{k}')
    assert_state_same(b, orc, f'{name} rollout')
    assert_same(b.obs, orc.obs(), 'rollout last obs')
    for t in range(T):
        obs, rew, done, res = a.step(None)
        assert_same(obs, ref['obs'][t], f'{name} step t={t} obs')
        assert_same(done, ref['done'][t], f'{name} step t={t} done')
    assert_state_same(a, orc, f'{name} stepwise')
    assert (a.stats.cpu().numpy() == b.stats.cpu().numpy()).all()
    assert b.kernel_name().startswith('s2d_reach_rollout_ws_kernel<' if ws == '1' else 's2d_reach_rollout_kernel<')


@pytest.mark.parametrize('ws', ['0', '1'])
def test_rollout_with_caller_actions_and_odd_n(ws, monkeypatch):
    """[T][N] caller actions; odd N exercises the unaligned observation-store path."""
    monkeypatch.setenv('S2D_ROLLOUT_WS', ws)
    kw = CONFIGS['dqn-discrete16']
    n, T = 333, 64
    eng, orc = _engine(n, **dict(kw)), _oracle(n, **dict(kw))
    eng.reset(); orc.reset()
    rs = np.random.RandomState(9)
    acts = rs.randint(0, 16, (T, n)).astype(np.int32)
    ref = orc.rollout(T, acts)
    out = eng.rollout(T, torch.as_tensor(acts, device='cuda:0'))
    torch.cuda.synchronize()
    for k in ('obs', 'action', 'reward', 'done', 'result'):
        assert_same(out[k], ref[k], f'rollout.{k}')
    assert_state_same(eng, orc, 'rollout caller actions')


def test_masked_reset_parity():
    kw = CONFIGS['dqn-discrete16']
    n = 500
    eng, orc = _engine(n, **dict(kw)), _oracle(n, **dict(kw))
    eng.reset(); orc.reset()
    rs = np.random.RandomState(1)
    for t in range(20):
        eng.step(None); orc.step(None)
        m = (rs.rand(n) < 0.2).astype(np.uint8)
        obs = eng.reset(torch.as_tensor(m, device='cuda:0'))
        o = orc.reset(m)
        assert_same(obs, o, f'masked reset t={t}')
    assert_state_same(eng, orc, 'masked reset')


def test_shard_invariance_on_device():
    """Two engines over [0,600) and [600,1000) == one engine over [0,1000) (global env ids)."""
    kw = dict(CONFIGS['noise-on'])
    whole = _engine(1000, **dict(kw))
    lo, hi = _engine(600, env_id_offset=0, **dict(kw)), _engine(400, env_id_offset=600, **dict(kw))
    for e in (whole, lo, hi):
        e.reset()
        e.rollout(90, with_obs=False)
    torch.cuda.synchronize()
    for f in O.STATE_FIELDS:
        assert torch.equal(torch.cat([getattr(lo, f), getattr(hi, f)]), getattr(whole, f)), f
    assert torch.equal(torch.cat([lo.obs, hi.obs]), whole.obs)


@pytest.mark.parametrize('ws', ['0', '1'])
def test_full_size_parity_and_properties(ws, monkeypatch):
    """BASELINE.json sizes: 65 536 envs, dqn kwargs, 256 random-policy steps (> one full episode).
    Full bit-exact comparison with the oracle plus size-independent invariants."""
    monkeypatch.setenv('S2D_ROLLOUT_WS', ws)
    kw = CONFIGS['dqn-discrete16']
    n, T = 65536, 256
    eng, orc = _engine(n, **dict(kw)), _oracle(n, **dict(kw))
    eng.reset(); orc.reset()
    out = eng.rollout(T)
    ref = orc.rollout(T)
    torch.cuda.synchronize()
    for k in ('action', 'reward', 'done', 'result', 'obs'):
        assert_same(out[k], ref[k], f'full-size rollout.{k}')
    assert_state_same(eng, orc, 'full-size')
    done, res = out['done'].cpu().numpy(), out['result'].cpu().numpy()
    assert ((res != 0) == (done != 0)).all()
    # cycle bookkeeping (A7): one cycle per step, plus one per reset (initial + each auto-reset)
    cyc = eng.cycle.cpu().numpy()
    assert (cyc == T + 1 + done.sum(axis=0)).all()
    # an episode never exceeds max_steps + 1 steps
    assert int(eng.step_number.max()) <= 201
    st = eng.stats.cpu().numpy()
    assert st[0] == n * T and st[1] == (res == 1).sum() and st[2] == (res == 2).sum() and st[3] == (res == 3).sum()
    assert st[3] > 0 and st[1] > 0
    obs = out['obs']
    assert bool(torch.isfinite(obs).all())
    assert float(obs[..., 0].abs().max()) <= 1.0 and float(obs[..., 7].abs().max()) <= 0.5


@pytest.mark.parametrize('name', ['noise-on', 'continuous1', 'turning4', 'reference-default'])
def test_full_size_parity_other_configurations(name):
    """BASELINE.json configs[2] size (65 536 envs x 256 cycles = bench.py's launch) for what the headline configuration does not
    cover: noise on (the drop-in default of make_config), 1-D continuous actions (the DDPG script's), the 4-D turning mode
    (reach_ball_env.py:59-79; bench.py's `turning4` line), and the reference's own default
    kwargs (use_continuous_action=True, change_ball_velocity=False: reach_ball_env.py:26-36) -- full bit-exact comparison through
    the kernel s2d_rollout picks by default."""
    kw = dict(use_continuous_action=True, use_turning=False) if name == 'reference-default' else dict(CONFIGS[name])
    kw.pop('max_steps', None)                              # the reference's 200
    n, T = 65536, 256
    eng, orc = _engine(n, **dict(kw)), _oracle(n, **dict(kw))
    eng.reset(); orc.reset()
    out = eng.rollout(T)
    ref = orc.rollout(T)
    assert eng.kernel_name().startswith('s2d_reach_rollout_ws_kernel<') and 'nt=1' in eng.kernel_name()
    _compare_rollout(out, ref, f'full-size {name}')
    assert_state_same(eng, orc, f'full-size {name}')
    done = out['done'].cpu().numpy()
    assert (eng.cycle.cpu().numpy() == T + 1 + done.sum(axis=0)).all()
    st = eng.stats.cpu().numpy()
    assert st[0] == n * T and st[1:4].sum() == done.sum()


@pytest.mark.parametrize('noise', [False, True])
def test_colliding_resets_keep_the_sign_of_zero(noise):
    """A reset whose command-less cycle ends in a player-ball collision multiplies the resting player's +0 velocity by
    collision_vel_rate < 0 and leaves -0 in the state (54 of 650 000 resets of the stock task; here nearly all: a player as large as
    half the pitch).  The per-step API serves resets from prepared slots that keep seven words of the post-reset state -- and the
    two signs in the slot's tag (round 3 restored +0 and differed from the checker in that sign bit).  Every state word, step by
    step, through the per-step API and through the rollout pipelines."""
    kw = dict(use_continuous_action=False, action_space_size=16, change_ball_velocity=True, max_steps=3, noise=noise,
              server=dict(player_size=25.0))
    n = 777
    eng, orc = _engine(n, **dict(kw)), _oracle(n, **dict(kw))
    eng.reset(); orc.reset()
    minus_zero = 0
    for t in range(40):
        eng.step(None); orc.step(None)
        assert_state_same(eng, orc, f't={t}')
        vx = orc.state('player_vx')
        minus_zero += int(((vx == 0) & np.signbit(vx)).sum())
    assert minus_zero > 1000                               # the case really occurs
    for T in (30, 5):
        _compare_rollout(eng.rollout(T), orc.rollout(T), f'rollout T={T}')
        assert_state_same(eng, orc, f'rollout T={T}')
    for t in range(6):                                      # and again through slots that a rollout left stale
        eng.step(None); orc.step(None)
        assert_state_same(eng, orc, f'after rollouts t={t}')


@pytest.mark.parametrize('name', ['dqn-discrete16', 'turning4', 'noise-on', 'no-autoreset-collide'])
def test_step_k_equals_k_steps(name):
    """s2d_step_k: k cycles of the per-step API in one launch == k calls of s2d_step == the oracle (every record word, the state,
    the arena's last-step outputs, the episode counters), with caller actions and with the in-kernel policy, k = 1, 2, 4, 7;
    episodes short enough that an env ends more than one inside a launch (the second reset is drawn inline)."""
    kw = dict(CONFIGS[name]); kw['max_steps'] = min(kw.get('max_steps', 200), 5)
    n = 777
    eng, one, orc = _engine(n, **dict(kw)), _engine(n, **dict(kw)), _oracle(n, **dict(kw))
    eng.reset(); one.reset(); orc.reset()
    rs = np.random.RandomState(3)
    for rep, k in enumerate([1, 2, 4, 7, 2, 4, 1, 7]):
        if rep % 2:
            a = np.stack([_random_actions(rs, kw, n) for _ in range(k)])
            out, ref = eng.step_k(k, torch.as_tensor(a, device='cuda:0')), orc.rollout(k, a)
            for t in range(k):
                one.step(torch.as_tensor(a[t], device='cuda:0'))
        else:
            out, ref = eng.step_k(k), orc.rollout(k)
            for t in range(k):
                one.step(None)
        assert eng.kernel_name() == 's2d_reach_step_k_kernel'
        _compare_rollout(out, ref, f'{name} k={k} rep={rep}')
        assert_state_same(eng, orc, f'{name} k={k} rep={rep}')
        assert_same(eng.obs, orc.obs(), f'{name} k={k} last obs'); assert_same(eng.reward, orc.reward(), f'{name} k={k} last reward')
        assert_same(eng.done, orc.done(), 'last done'); assert_same(eng.result, orc.result(), 'last result')
        for f in O.STATE_FIELDS:
            assert torch.equal(getattr(eng, f), getattr(one, f)), (name, k, f)
        assert torch.equal(eng.obs, one.obs) and torch.equal(eng.stats, one.stats)
    assert int(eng.stats[1:4].sum()) > n
    with pytest.raises(ValueError):
        eng.step_k(65)


def test_errors_are_loud():
    from soccer2d_amd.engine import Engine, make_config
    with pytest.raises(ValueError):
        make_config(no_such_kwarg=1)
    with pytest.raises(ValueError):
        make_config(use_continuous_action=False, action_space_size=0)
    eng = _engine(8, use_continuous_action=False)
    with pytest.raises(ValueError):
        eng.step(torch.zeros(9, dtype=torch.int64, device='cuda:0'))
    with pytest.raises(ValueError):
        Engine(0, 'cuda:0')


def test_edge_inputs_parity():
    """Edge cases: discrete actions outside [0, n) (floor-mod wrap, reach_ball_env.py:84), huge and tiny
    continuous actions (not clipped, :81; dash angle clamps to +-180), a single env, reset with an
    all-zero and an all-one mask, zero-step rollout."""
    kw = dict(use_continuous_action=False, action_space_size=16, change_ball_velocity=True, max_steps=30)
    n = 130
    eng, orc = _engine(n, **dict(kw)), _oracle(n, **dict(kw))
    eng.reset(); orc.reset()
    rs = np.random.RandomState(7)
    for t in range(40):
        a = rs.randint(-40, 60, n).astype(np.int32)
        obs, rew, done, res = eng.step(torch.as_tensor(a, device='cuda:0'))
        o_obs, o_rew, o_done, o_res = orc.step(a)
        assert_same(obs, o_obs, f'oob t={t} obs'); assert_same(rew, o_rew, f'oob t={t} reward')
        assert_same(eng.action_dir, orc.action_dir(), f'oob t={t} dir')
    assert_state_same(eng, orc, 'out-of-range discrete')
    kw = dict(use_continuous_action=True, use_turning=False, max_steps=25)
    eng, orc = _engine(n, **dict(kw)), _oracle(n, **dict(kw))
    eng.reset(); orc.reset()
    for t in range(30):
        a = (rs.uniform(-1, 1, (n, 1)) * 10.0 ** rs.randint(-8, 9, (n, 1))).astype(np.float32)
        obs, rew, done, res = eng.step(torch.as_tensor(a, device='cuda:0'))
        o = orc.step(a)
        assert_same(obs, o[0], f'huge t={t} obs')
    assert_state_same(eng, orc, 'huge continuous')
    one, orc1 = _engine(1, **dict(kw)), _oracle(1, **dict(kw))
    one.reset(); orc1.reset()
    one.rollout(70); orc1.rollout(70)
    assert_state_same(one, orc1, 'single env')
    before = one.arena.clone()
    one.reset(torch.zeros(1, dtype=torch.uint8, device='cuda:0'))     # empty mask: nothing changes
    one.rollout(0)
    torch.cuda.synchronize()
    assert torch.equal(before, one.arena)
    one.reset(torch.ones(1, dtype=torch.uint8, device='cuda:0')); orc1.reset(np.ones(1, dtype=np.uint8))
    assert_state_same(one, orc1, 'full mask')


@pytest.mark.parametrize('ws', ['0', '1'])
@pytest.mark.parametrize('name', ['dqn-discrete16', 'turning4'])
def test_short_rollouts_pipeline_fill_and_drain(name, ws, monkeypatch):
    """Rollouts of 1, 2, 3 and 5 cycles (shorter than / equal to the depth of the three-stage pipeline of
    the wave-specialised kernel), interleaved with per-step launches and with caller actions, odd N:
    state, outputs and policy_step stay equal to the oracle's, launch after launch."""
    monkeypatch.setenv('S2D_ROLLOUT_WS', ws)
    kw = dict(CONFIGS[name]); kw['max_steps'] = 12
    n = 197
    eng, orc = _engine(n, **dict(kw)), _oracle(n, **dict(kw))
    eng.reset(); orc.reset()
    rs = np.random.RandomState(11)
    for rep, T in enumerate([1, 2, 3, 5, 1, 2, 3, 5, 4, 1]):
        if rep % 3 == 2:                                   # caller actions
            if kw.get('use_continuous_action', True):
                a = rs.uniform(-1, 1, (T, n, 4 if kw.get('use_turning') else 1)).astype(np.float32)
            else:
                a = rs.randint(0, kw['action_space_size'], (T, n)).astype(np.int32)
            out, ref = eng.rollout(T, torch.as_tensor(a, device='cuda:0')), orc.rollout(T, a)
        else:
            out, ref = eng.rollout(T), orc.rollout(T)
        torch.cuda.synchronize()
        for k in ('obs', 'action', 'reward', 'done', 'result'):
            assert_same(out[k], ref[k], f'{name} T={T} rep={rep} {k}')
        assert_state_same(eng, orc, f'{name} T={T} rep={rep}')
        assert_same(eng.obs, orc.obs(), f'{name} T={T} rep={rep} last obs')
        eng.step(None); orc.step(None)
        assert_state_same(eng, orc, f'{name} step after T={T}')
    assert int(eng.policy_step.min()) == int(eng.policy_step.max()) > 0


@pytest.mark.parametrize('ws', ['1', '0'])
@pytest.mark.parametrize('noise', [False, True])
def test_dash_fast_path_and_its_fallback(noise, ws, monkeypatch):
    """The four-wave pipeline's dash-only fast path (stamina table + whole-degree sine table) is taken by groups whose envs
    sit on the table; a group holding a foreign state (another stamina word, a fractional body angle, a step number beyond
    the table) must run the generic loop -- both bit-equal to the oracle, in the same launch.  Also: more than kSlots = 3
    episodes ending inside one launch (the inline-prepare path), forced by max_steps = 4."""
    monkeypatch.setenv('S2D_ROLLOUT_WS', ws)                # the four-wave pipeline and the unified kernel both have the fast path
    kw = dict(use_continuous_action=False, action_space_size=16, change_ball_velocity=True, max_steps=4, noise=noise)
    n = 64 * 6 + 17
    eng, orc = _engine(n, **dict(kw)), _oracle(n, **dict(kw))
    eng.reset(); orc.reset()
    _compare_rollout(eng.rollout(24), orc.rollout(24), 'many episodes per launch')
    assert_state_same(eng, orc, 'after the short-episode launch')
    # foreign states in groups 1, 3 and 5 (groups 0, 2, 4 and the ragged last one stay on the table)
    torch.cuda.synchronize()
    edits = {70: dict(stamina=7777.0), 64 * 3 + 5: dict(player_body=33.5), 64 * 5 + 63: dict(effort=0.75, recovery=0.9)}
    for i, kv in edits.items():
        for k, v in kv.items():
            getattr(eng, k)[i] = v
        orc.set_env(i, **kv)
    for T in (7, 64):
        _compare_rollout(eng.rollout(T), orc.rollout(T), f'foreign states T={T}')
        assert_state_same(eng, orc, f'foreign states T={T}')
    o1, r1, d1, s1 = eng.step(None); o2, r2, d2, s2 = orc.step(None)
    assert_same(o1, o2, 'step after'); assert_same(r1, r2, 'step after reward')


def test_cycle_counter_wraps_past_int32():
    """`cycle` is an int32 like WorldModel.cycle; a long-running engine passes 2^31 (47 minutes of fused
    rollouts): the counter wraps, the Philox counters derived from it keep matching the oracle."""
    kw = dict(CONFIGS['dqn-discrete16']); kw['max_steps'] = 15
    n = 96
    eng, orc = _engine(n, **dict(kw)), _oracle(n, **dict(kw))
    eng.reset(); orc.reset()
    for i in range(0, n, 3):
        c = 2 ** 31 - 1 - (i % 7)
        eng.cycle[i] = c
        orc.set_env(i, cycle=c)
    out, ref = eng.rollout(40), orc.rollout(40)
    torch.cuda.synchronize()
    for k in ('obs', 'reward', 'done', 'result'):
        assert_same(out[k], ref[k], f'wrap {k}')
    assert_state_same(eng, orc, 'wrap')
    assert int(eng.cycle.min()) < -2 ** 31 + 200


@pytest.mark.parametrize('ws', ['0', '1'])
@pytest.mark.parametrize('seed', list(range(6)))
def test_random_server_parameters_parity(seed, ws, monkeypatch):
    """Parity must not hinge on the stock parameter values: random ServerParam / task settings (speeds and
    accelerations that make the clamps fire, a player as large as the ball-collision radius needs, no stamina
    capacity, continuous dash directions, short episodes, ...) -- rollouts, per-step launches and resets stay
    bit-identical to the oracle."""
    monkeypatch.setenv('S2D_ROLLOUT_WS', ws)
    rs = np.random.RandomState(100 + seed)
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
    n = 257
    eng, orc = _engine(n, **dict(kw)), _oracle(n, **dict(kw))
    eng.reset(); orc.reset()
    assert_state_same(eng, orc, f'cfg{seed} reset')
    out, ref = eng.rollout(70), orc.rollout(70)
    torch.cuda.synchronize()
    for k in ('obs', 'action', 'reward', 'done', 'result'):
        assert_same(out[k], ref[k], f'cfg{seed} rollout.{k}')
    for t in range(25):
        eng.step(None); orc.step(None)
    assert_state_same(eng, orc, f'cfg{seed} after steps')
    assert_same(eng.obs, orc.obs(), f'cfg{seed} obs')
    assert int(eng.stats[1:4].sum()) > 0


def test_long_run_pipeline_vs_unified_kernels(monkeypatch):
    """65 536 envs x 4 096 cycles (64 launches), noise on: the four-wave pipeline and the unified kernel are
    two independent schedules of the same arithmetic -- every state word, the last observations, the episode
    counters and running checksums of the rollout records stay identical."""
    from soccer2d_amd.engine import Engine, make_config
    kw = dict(CONFIGS['noise-on'])
    res = []
    for ws in ('0', '1'):
        monkeypatch.setenv('S2D_ROLLOUT_WS', ws)
        eng = Engine(65536, 'cuda:0', cfg=make_config(**{k: v for k, v in kw.items()}))
        eng.reset()
        out = eng.alloc_rollout(64)
        acc = torch.zeros(4, dtype=torch.float64, device='cuda:0')
        for i in range(64):
            eng.rollout(64, out=out)
            if i % 8 == 7:
                acc += torch.stack([out['obs'].double().sum(), out['reward'].double().sum(), out['done'].double().sum(),
                                    out['action'].double().sum()])
        torch.cuda.synchronize()
        res.append((eng, acc))
    (a, ca), (b, cb) = res
    assert a.kernel_name().startswith('s2d_reach_rollout_kernel<') and b.kernel_name().startswith('s2d_reach_rollout_ws_kernel<')
    assert torch.equal(ca, cb) and torch.equal(a.stats, b.stats) and int(a.stats[0]) == 65536 * 4096
    for f in O.STATE_FIELDS:
        assert torch.equal(getattr(a, f), getattr(b, f)), f
    assert torch.equal(a.obs, b.obs) and torch.equal(a.terminal_obs, b.terminal_obs)


@pytest.mark.parametrize('noise', [False, True])
@pytest.mark.parametrize('case', ['every-step-ends', 'very-short', 'mixed-256', 'long-512'])
def test_many_episodes_per_launch_in_long_launches(case, noise, monkeypatch):
    """The four-wave pipeline prepares three episodes per env before its loop and prepares inline from the fourth on.  Launches
    far longer than the prepared slots last, against the oracle bit for bit: (a) min_distance_to_ball beyond the pitch -> EVERY
    step ends an episode; (b) max_steps = 2 (every env consumes a slot every third cycle); (c) the benchmark task at T = 256
    (bench.py's default launch) with ragged N; (d) T = 512, episodes of <= 40 cycles.  (Round 3 also built an in-loop top-up of the
    slots by the policy wave -- profiles/experiments/ws_slot_topup.patch -- which passes these same cases and was rejected on time.)"""
    monkeypatch.setenv('S2D_ROLLOUT_WS', '1')
    base = dict(use_continuous_action=False, action_space_size=16, change_ball_velocity=True, noise=noise)
    kw, n, Ts = {'every-step-ends': (dict(base, min_distance_to_ball=500.0, max_steps=200), 64 * 3 + 9, (70, 33)),
                 'very-short': (dict(base, max_steps=2), 64 * 2 + 40, (150, 64)),
                 'mixed-256': (dict(base, max_steps=200), 64 * 9 + 1, (256, 256)),
                 'long-512': (dict(base, max_steps=40), 64 * 4, (512, 200))}[case]
    eng, orc = _engine(n, **dict(kw)), _oracle(n, **dict(kw))
    eng.reset(); orc.reset()
    for T in Ts:
        out, ref = eng.rollout(T), orc.rollout(T)
        assert eng.kernel_name().startswith('s2d_reach_rollout_ws_kernel<')
        _compare_rollout(out, ref, f'{case} T={T}')
        assert_state_same(eng, orc, f'{case} T={T}')
        assert_same(eng.obs, orc.obs(), f'{case} T={T} last obs')
        assert_same(eng.terminal_obs, orc.terminal_obs(), f'{case} T={T} terminal obs')
    o1, r1, d1, s1 = eng.step(None); o2, r2, d2, s2 = orc.step(None)
    assert_same(o1, o2, f'{case} step after'); assert_same(r1, r2, f'{case} step after reward')
    assert int(eng.episode.min()) >= (sum(Ts) if case == 'every-step-ends' else 2)
