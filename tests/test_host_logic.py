"""CPU tests of the host-side logic around the C ABI: spaces, factory error behaviour, lazy
infos, the InfoCollectorCallback mirror, shard arithmetic, and the N>1 path (world_size-2
gloo): shard invariance + the rollout all-gather, with the oracle standing in for the GPU
engine (the collective plumbing is what is under test here)."""
import os
import sys

import numpy as np
import pytest
import torch

import oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_spaces_match_reference_shapes():
    from soccer2d_amd.spaces import reach_ball_spaces
    act, obs = reach_ball_spaces(False, False, 16)
    assert act.n == 16 and obs.shape == (10,) and obs.dtype == np.float32
    act, _ = reach_ball_spaces(True, False)
    assert act.shape == (1,) and float(act.low.min()) == -1 and float(act.high.max()) == 1
    act, _ = reach_ball_spaces(True, True)
    assert act.shape == (4,)


def test_factory_errors_like_reference():
    from sample_environments.environment_factory import EnvironmentFactory
    with pytest.raises(ValueError, match='not found'):
        EnvironmentFactory().create('no-such-env', None, None, None)
    with pytest.raises(ValueError, match='not found'):
        EnvironmentFactory().create_vec('no-such-env', 4)
    if not torch.cuda.is_available():
        with pytest.raises(RuntimeError, match='no CPU fallback'):
            EnvironmentFactory().create('ReachBall', None, None, None)


def test_lazy_infos_and_callback():
    from soccer2d_amd.sb3_vec_env import LazyInfos
    from utils.info_collector_callback import InfoCollectorCallback
    codes = np.array([0, 1, 0, 3, 2, 0], dtype=np.uint8)
    term = np.arange(60, dtype=np.float32).reshape(6, 10)
    infos = LazyInfos(codes, term)
    assert len(infos) == 6 and infos[0] == {'result': None}
    assert infos[1]['result'] == 'Goal' and infos[3]['TimeLimit.truncated'] is True and infos[4]['TimeLimit.truncated'] is False
    assert (infos[3]['terminal_observation'] == term[3]).all()
    assert [d['result'] for d in infos] == [None, 'Goal', None, 'Timeout', 'Out', None]
    cb = InfoCollectorCallback()
    cb.locals = {'infos': list(infos)}
    assert cb._on_step() is True                        # reference contract: info['result'] falsy -> skipped
    assert [i['result'] for i in cb.infos] == ['Goal', 'Timeout', 'Out']
    cb.on_infos(infos)                                  # lazy path: only finished envs are touched
    cb.on_result_codes(torch.tensor([[0, 1], [1, 3]], dtype=torch.uint8))
    res = cb.update_results_dict()
    assert res['Goal'] == [pytest.approx(4 / 9 * 100)] and res['Timeout'] == [pytest.approx(3 / 9 * 100)]
    cb.reset()
    assert cb.infos == []


def test_shard_range():
    from soccer2d_amd.dist import shard_range
    for n, w in ((65536 * 8, 8), (10, 3), (7, 7)):
        parts = [shard_range(n, r, w) for r in range(w)]
        assert parts[0][0] == 0 and sum(c for _, c in parts) == n
        for (o0, c0), (o1, _c1) in zip(parts, parts[1:]):
            assert o0 + c0 == o1
        assert max(c for _, c in parts) - min(c for _, c in parts) <= 1
    with pytest.raises(ValueError):
        shard_range(4, 4, 4)


def _worker(rank, world, port, n_global, T, q):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    import torch.distributed as dist
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    sys.path.insert(0, os.path.join(ROOT, 'gym-soccer-2d-env_amd'))
    import oracle as O2
    from soccer2d_amd.dist import all_gather_rollout, all_reduce_stats, shard_range
    dist.init_process_group('gloo', rank=rank, world_size=world)
    off, cnt = shard_range(n_global, rank, world)
    cfg = O2.make_config(env_id_offset=off, noise=1, **O2.DQN_KWARGS)
    eng = O2.OracleEngine(cfg, cnt, 'f32')
    eng.reset()
    ro = eng.rollout(T)
    local = {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in ro.items()}
    calls, real = [], dist.all_gather_into_tensor
    dist.all_gather_into_tensor = lambda *a, **k: (calls.append(1), real(*a, **k))[1]
    full = all_gather_rollout(local, time_major=True)
    assert len(calls) == 1, 'one collective per exchange (all five fields travel in one slab)'
    raw = all_gather_rollout(local, time_major=False)
    from soccer2d_amd.dist import LEAGUE_FIELDS
    calls.clear()
    small = all_gather_rollout(local, time_major=True, fields=LEAGUE_FIELDS)       # the league payload: 10 of the record's 50 bytes
    assert len(calls) == 1 and sorted(small) == sorted(LEAGUE_FIELDS)
    dist.all_gather_into_tensor = real
    stats = all_reduce_stats(torch.from_numpy(eng.stats().astype(np.int64)))
    if rank == 0:
        full.update({'small_' + k: v for k, v in small.items()})
        q.put(({k: v.numpy() for k, v in full.items()}, {k: tuple(v.shape) for k, v in raw.items()}, stats.numpy()))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gloo_shards_and_all_gather():
    """world_size 2: each rank simulates its shard, rollouts are all-gathered; the result is
    identical to one process simulating the whole env range (no data-path collective needed
    for the simulation itself)."""
    import torch.multiprocessing as mp
    n_global, T, world = 64, 40, 2
    port = 29500 + (os.getpid() % 2000)
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_global, T, q)) for r in range(world)]
    for p in procs:
        p.start()
    full, raw_shapes, stats = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    cfg = O.make_config(noise=1, **O.DQN_KWARGS)
    whole = O.OracleEngine(cfg, n_global, 'f32')
    whole.reset()
    ref = whole.rollout(T)
    for k in ('obs', 'action', 'reward', 'done', 'result'):
        assert full[k].shape == ref[k].shape
        assert np.array_equal(full[k].view(np.uint8), np.ascontiguousarray(ref[k]).view(np.uint8)), k
    for k in ('action', 'reward', 'done', 'result'):       # the subset slab (LeagueRolloutExchange(fields=LEAGUE_FIELDS)) == the same oracle
        assert np.array_equal(full['small_' + k].view(np.uint8), np.ascontiguousarray(ref[k]).view(np.uint8)), k
    assert 'small_obs' not in full
    assert raw_shapes['obs'] == (2, T, n_global // 2, 10)
    assert list(stats[:4]) == list(whole.stats()[:4].astype(np.int64))


def _league_worker(rank, world, port, q):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    import torch.distributed as dist
    sys.path.insert(0, os.path.join(ROOT, 'gym-soccer-2d-env_amd'))
    from soccer2d_amd.league import League, exchange_results
    dist.init_process_group('gloo', rank=rank, world_size=world)
    lg = League(5, seed=3)
    n_local = 12
    for rnd in range(3):
        left, right = lg.pairing(rnd, rank * n_local, n_local)
        g = torch.Generator().manual_seed(100 * rnd + rank)
        gl, gr = torch.randint(0, 4, (n_local,), generator=g), torch.randint(0, 4, (n_local,), generator=g)
        L, R, GL, GR = exchange_results(left, right, gl, gr)
        lg.update(L, R, GL, GR)
    q.put((rank, lg.elo.clone(), lg.games.clone()))
    dist.barrier()
    dist.destroy_process_group()


def test_league_table_stays_replicated_over_two_ranks():
    """configs[4] plumbing on CPU: results all-gathered with gloo, identical Elo tables on both ranks,
    equal to a single process that saw every match."""
    import torch.multiprocessing as mp
    from soccer2d_amd.league import League
    world, port = 2, 29800 + (os.getpid() % 1000)
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_league_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = sorted([q.get(timeout=120) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert torch.equal(got[0][1], got[1][1]) and torch.equal(got[0][2], got[1][2])
    ref = League(5, seed=3)
    for rnd in range(3):
        parts = []
        for rank in range(world):
            left, right = ref.pairing(rnd, rank * 12, 12)
            g = torch.Generator().manual_seed(100 * rnd + rank)
            parts.append((left, right, torch.randint(0, 4, (12,), generator=g), torch.randint(0, 4, (12,), generator=g)))
        ref.update(*(torch.cat([p[k] for p in parts]) for k in range(4)))
    assert torch.allclose(ref.elo, got[0][1]) and int(got[0][2].sum()) == 2 * 3 * 24
    left, right = ref.pairing(0, 0, 1000)
    assert bool((left != right).all()) and int(left.min()) == 0 and int(left.max()) == 4


def test_player_type_generator_is_deterministic_and_respects_the_trade_offs():
    """s2d_match_generate_player_types is host code (no GPU): 17 heterogeneous types from the stock
    PlayerParam ranges, the trade-off pairs move together, type 0 stays the default."""
    import ctypes as C
    from soccer2d_amd import _capi, _capi_match as M
    lib = M.bind(_capi.load_library())
    cfgs = []
    for seed in (1, 1, 2):
        cfg = M.S2DMatchConfig()
        lib.s2d_match_default_config(C.byref(cfg))
        assert lib.s2d_match_generate_player_types(C.byref(cfg), None, seed) == 0
        cfgs.append(cfg)
    a, b, c = cfgs
    as_rows = lambda cfg: [[getattr(cfg.player_types[t], f) for f in M.PLAYER_TYPE_FIELDS] for t in range(18)]
    assert as_rows(a) == as_rows(b) and as_rows(a) != as_rows(c)
    t0 = a.player_types[0]
    assert (t0.player_decay, t0.dash_power_rate, t0.kickable_margin, t0.effort_max) == (0.4, 0.006, 0.7, 1.0)
    kinds = set()
    for t in range(1, 18):
        y = a.player_types[t]
        assert 0.3 <= y.player_decay <= 0.5 and abs((y.inertia_moment - 5.0) - 25.0 * (y.player_decay - 0.4)) < 1e-9
        assert 0.0048 <= y.dash_power_rate <= 0.0068
        assert abs((y.stamina_inc_max - 45.0) + 6000.0 * (y.dash_power_rate - 0.006)) < 1e-9
        assert 0.6 <= y.kickable_margin <= 0.8 and abs((y.kick_rand - 0.1) - (y.kickable_margin - 0.7)) < 1e-9
        assert 50.0 <= y.extra_stamina <= 100.0 and abs((y.effort_max - 1.0) + 0.004 * (y.extra_stamina - 50.0)) < 1e-9
        assert abs((y.effort_min - 0.6) + 0.004 * (y.extra_stamina - 50.0)) < 1e-9
        assert 1.0 <= y.catchable_area_l_stretch <= 1.3 and y.player_speed_max == 1.05 and y.player_size == 0.3
        top = y.effort_max * y.dash_power_rate * 100.0 / (1.0 - y.player_decay)
        assert 0.75 < top <= 1.05
        kinds.add(round(y.player_decay, 6))
    assert len(kinds) == 17
    assert lib.s2d_match_validate_config(C.byref(a)) == 0
    a.player_type_id[3] = 18
    assert lib.s2d_match_validate_config(C.byref(a)) != 0


def test_bench_step_units(monkeypatch):
    """bench.py: a step is one launch of `--fuse` cycles in rollout / graph mode, one cycle in step mode."""
    import importlib
    import sys as _sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    monkeypatch.syspath_prepend(root)
    bench = importlib.import_module('bench')
    for argv, want in ((['bench.py'], (32, 4, 256)), (['bench.py', '--mode', 'step'], (4096, 256, 1)),
                       (['bench.py', '--steps', '5', '--warmup', '1', '--fuse', '32'], (5, 1, 32)),
                       (['bench.py', '--mode', 'graph', '--steps', '7'], (7, 4, 256))):
        monkeypatch.setattr(_sys, 'argv', argv)
        a = bench.parse()
        assert (a.steps, a.warmup, a.cycles_per_step) == want
    monkeypatch.setattr(_sys, 'argv', ['bench.py', '--steps', '0'])
    with pytest.raises(SystemExit):
        bench.parse()
    # the untimed settle phase (steady clocks before the W warm-up steps): on by default, 0 switches it off
    monkeypatch.setattr(_sys, 'argv', ['bench.py'])
    assert bench.parse().settle_ms == 200.0
    monkeypatch.setattr(_sys, 'argv', ['bench.py', '--settle-ms', '0'])
    assert bench.parse().settle_ms == 0.0
    calls = []
    assert bench.settle(lambda k: calls.append(k), 64, 0) == 0 and calls == []
    # defaults per task, the rotating-buffer headline, five timed regions
    a = bench.parse(['--task', 'match'])
    assert a.envs == 8192 and bench.parse([]).envs == 65536 and bench.parse([]).repeats == 5 and bench.parse([]).rotate_buffers == 0
    assert bench.n_rotating(64 * 65536 * 50) == 3 and bench.n_rotating(64 * 65536 * 50) * 64 * 65536 * 50 > (512 << 20)


def test_bench_names_the_workload_it_runs():
    """metric / config.workload follow --envs / --task / --league-exchange (BASELINE.json configs[1..4]); other sizes are 'custom'."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import bench
    m, w = bench.workload_of('reach_ball', 65536, 1, False)
    assert '65536' in m and 'configs[2]' in w
    m, w = bench.workload_of('reach_ball', 4096, 1, False)
    assert '4096' in m and 'configs[1]' in w and '65536' not in m + w
    m, w = bench.workload_of('match', 8192, 1, False)
    assert '11v11' in m and 'configs[3]' in w
    m, w = bench.workload_of('reach_ball', 65536, 8, True)
    assert 'all-gather' in m and 'configs[4]' in w and '65536x8' in w
    m, w = bench.workload_of('reach_ball', 1000, 1, False)
    assert 'custom' in w and 'configs[' not in w


def test_bench_self_launch_builds_the_torchrun_command(monkeypatch, capsys):
    """`python bench.py --gpus N` without WORLD_SIZE: the parent counts GPUs in a child, starts
    `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py <same args>` as a child
    process and relays exactly rank 0's JSON line; fewer GPUs than ranks -> the shared-GPU gloo rehearsal environment."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import bench
    seen = []

    def fake_run(cmd, **kw):
        seen.append((cmd, kw.get('env')))
        if '-c' in cmd:
            return subprocess.CompletedProcess(cmd, 0, stdout=f'{ndev}\n', stderr='')
        return subprocess.CompletedProcess(cmd, 0, stdout='noise\n{"metric": "m", "value": 1}\n', stderr='')
    monkeypatch.setattr(subprocess, 'run', fake_run)
    monkeypatch.delenv('WORLD_SIZE', raising=False)
    monkeypatch.delenv('S2D_DIST_BACKEND', raising=False)
    ndev = 8
    bench.main(['--gpus', '4', '--steps', '3'])
    cmd, env = seen[-1]
    assert cmd[1:4] == ['-m', 'torch.distributed.run', '--nnodes=1'] and '--nproc-per-node=4' in cmd
    assert cmd[cmd.index('--master-addr') + 1] == '127.0.0.1' and cmd[-4:] == ['--gpus', '4', '--steps', '3']
    assert cmd[cmd.index('--master-port') + 2].endswith('bench.py') and 'S2D_DIST_BACKEND' not in env
    assert env['HSA_ENABLE_IPC_MODE_LEGACY'] == '0'
    assert capsys.readouterr().out.strip() == '{"metric": "m", "value": 1}'
    ndev = 1
    bench.main(['--gpus', '2'])
    assert seen[-1][1]['S2D_DIST_BACKEND'] == 'gloo' and seen[-1][1]['S2D_BENCH_SHARE_GPU'] == '1'
    with pytest.raises(SystemExit):
        bench.main(['--gpus', '8'])            # 8 ranks on one card: refused (at most 6 processes may share a GPU)
    ndev = 0
    with pytest.raises(SystemExit):
        bench.main(['--gpus', '2'])


def test_logger_utils_mirror(tmp_path, monkeypatch):
    """utils.logger_utils.setup_logger as the reference's scripts use it (dqn_stable_baselines3.py:12-14)."""
    import logging
    from utils.logger_utils import setup_logger
    # like the reference, "already configured" is Logger.hasHandlers(), which also looks at the root logger:
    # pytest hangs its capture handlers there, a training script does not
    monkeypatch.setattr(logging.getLogger(), 'handlers', [])
    d = tmp_path / 'logs' / 'run1'
    lg = setup_logger('S2DTestLogger', str(d), console_level=logging.DEBUG, file_level=logging.DEBUG)
    assert lg is logging.getLogger('S2DTestLogger') and len(lg.handlers) == 2 and d.is_dir()
    lg.info('hello %d', 7)
    for h in lg.handlers:
        h.flush()
    text = (d / 'S2DTestLogger.log').read_text()
    assert 'S2DTestLogger - INFO - hello 7' in text
    assert setup_logger('S2DTestLogger', str(d)) is lg and len(lg.handlers) == 2      # configured once
    only_file = setup_logger('S2DTestLogger2', str(d), console_level=None)
    assert len(only_file.handlers) == 1 and isinstance(only_file.handlers[0], logging.FileHandler)
    for name in ('S2DTestLogger', 'S2DTestLogger2'):
        for h in list(logging.getLogger(name).handlers):
            h.close(); logging.getLogger(name).removeHandler(h)


def test_service_pb2_messages_and_command_translation():
    """The light service_pb2 the hook path ships: keyword constructors, proto3 defaults, oneof bookkeeping, enum values of
    idl/service.proto:267-301; and what the mirror makes of a hook's PlayerAction (one body command per cycle)."""
    sys.path.insert(0, os.path.join(ROOT, 'gym-soccer-2d-env_amd'))
    import service_pb2 as pb2
    from soccer2d_amd import _capi
    from soccer2d_amd.hook_env import command_of
    a = pb2.PlayerAction(dash=pb2.Dash(power=100, relative_direction=22.5))
    assert a.WhichOneof('action') == 'dash' and a.HasField('dash') and not a.HasField('turn') and a.turn.relative_direction == 0.0
    a.turn = pb2.Turn(relative_direction=-30.0)                          # a oneof keeps one member
    assert a.WhichOneof('action') == 'turn' and not a.HasField('dash')
    assert command_of(a) == (_capi.CMD_TURN, 0.0, -30.0)
    assert command_of(pb2.PlayerAction(dash=pb2.Dash(power=55, relative_direction=90))) == (_capi.CMD_DASH, 55.0, 90.0)
    assert command_of(pb2.PlayerAction(body_hold_ball=pb2.Body_HoldBall())) == (_capi.CMD_NONE, 0.0, 0.0)
    assert command_of(pb2.PlayerAction()) == (_capi.CMD_NONE, 0.0, 0.0) and command_of(None)[0] == _capi.CMD_NONE
    assert command_of([pb2.PlayerAction(turn_neck=pb2.TurnNeck(moment=10)), a]) == (_capi.CMD_TURN, 0.0, -30.0)
    with pytest.raises(NotImplementedError):
        command_of(pb2.PlayerAction(kick=pb2.Kick(power=100, relative_direction=0)))
    with pytest.raises(NotImplementedError):
        command_of(pb2.PlayerAction(body_go_to_point={}))                # a helios behaviour: accepted as a message, not executable
    with pytest.raises(AttributeError):
        pb2.Dash(powr=1)
    t = pb2.TrainerAction(do_move_player=pb2.DoMovePlayer(our_side=True, uniform_number=1, position=pb2.RpcVector2D(x=3, y=-4), body_direction=270))
    assert t.WhichOneof('action') == 'do_move_player' and t.do_move_player.position.y == -4 and t.do_move_ball.velocity.x == 0.0
    g = pb2.GameModeType
    assert (g.BeforeKickOff, g.PlayOn, g.KickOff_, g.OffSide_, g.FirstHalfOver, g.BackPass_, g.FreeKickFault_, g.CatchFault_, g.IndFreeKick_,
            g.PenaltySetup_, g.GoalieCatch_, g.MODE_MAX) == (0, 2, 3, 9, 11, 18, 19, 20, 21, 22, 30, 32)
    assert (pb2.Side.UNKNOWN, pb2.Side.LEFT, pb2.Side.RIGHT) == (0, 1, 2)
    import soccer_2d_env
    from sample_environments.reach_ball_env import ReachBallEnv
    assert not soccer_2d_env.Soccer2DEnv.overrides_task_hooks() and not ReachBallEnv.overrides_task_hooks()   # the fused tasks
