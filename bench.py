#!/usr/bin/env python3
"""bench.py -- env-steps/sec of the reach_ball hot path on N MI355X (BASELINE.json metric).

One simulator CYCLE for every env of the batch = action decode -> dash/turn -> stamina ->
integrate -> collide -> decay -> observation -> reward/done/result -> auto-reset, with the
rollout record of that cycle (obs[10], action, reward, done, result per env) written to HBM.
A bench "step" (--steps K / --warmup W) is one pass of the hot path over the batch as the mode
issues it: ONE LAUNCH of T = 64 fused cycles in the default rollout mode (and one replay of a
T-cycle hipGraph in graph mode), one cycle in step mode.  `value` is env-steps (env-cycles) per
second in every mode: N envs x cycles / seconds.  Workload = BASELINE.json configs[2]: 65 536 reach_ball envs per GPU, kwargs of
dqn_stable_baselines3.py:18-31, uniform random policy drawn in-kernel (Philox), synthetic
reset distribution of reach_ball_env.py:170-218.  Inputs are resident in HBM before the
timed region; nothing is copied to the host inside it.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--mode rollout|step|graph] [--fuse T]
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
         --master-port P bench.py --gpus N --steps K --warmup W

Modes (all run the same arithmetic, bit-identical trajectories):
  rollout  (default) T cycles fused per launch (s2d_rollout): state stays in registers
  step     one launch per cycle (s2d_step), eager
  graph    one launch per cycle, T of them captured in a hipGraph and replayed
Multi-GPU: one process per GPU, contiguous global env-id ranges, NO data-path collective
(envs are independent) -> weak scaling; only the timing max is all-reduced.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for _p in (os.path.join(ROOT, 'gym-soccer-2d-env_amd'), os.path.join(ROOT, 'tests'), ROOT):
    if _p not in sys.path:
        sys.path.insert(0, _p)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)
STATE_BYTES = 68               # 17 words per env
RECORD_BYTES = 50              # obs 40 + action 4 + reward 4 + done 1 + result 1 per env-step
DQN_KWARGS = dict(change_ball_position=True, change_ball_velocity=True, min_distance_to_ball=5.0,
                  max_steps=200, use_continuous_action=False, action_space_size=16, use_turning=False)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=None, help='timed bench steps (default 64 launches; 4096 in step mode)')
    ap.add_argument('--warmup', type=int, default=None, help='untimed bench steps (default 4 launches; 256 in step mode)')
    ap.add_argument('--envs', type=int, default=65536, help='envs per GPU')
    ap.add_argument('--mode', choices=('rollout', 'step', 'graph'), default='rollout')
    ap.add_argument('--fuse', type=int, default=64, help='cycles per launch (rollout) / per graph (graph)')
    ap.add_argument('--noise', action='store_true', help='player_rand/ball_rand Philox noise on')
    ap.add_argument('--task', choices=('reach_ball', 'match'), default='reach_ball',
                    help='reach_ball = the BASELINE.json metric (default); match = 11v11 engine, configs[3] (8 192 matches)')
    ap.add_argument('--variant', choices=('dqn', 'no-auto-reset', 'never-done'), default='dqn',
                    help='experiments only: dqn = the benchmark workload; the others switch episode ends off')
    ap.add_argument('--settle-ms', type=float, default=200.0,
                    help='untimed load before the W warm-up steps so that the chip\'s clock has settled (it takes tens of ms '
                         'of sustained load; a 4 ms run measures the ramp: 66 G instead of 80 G env-steps/s); 0 = off')
    ap.add_argument('--rotate-buffers', type=int, default=1,
                    help='rollout mode: cycle through this many record buffers (K x 218.6 MB > 512 MiB makes every launch write cold '
                         'lines: no hit in the 256 MiB Infinity Cache); the default line re-writes one buffer and carries the rotating '
                         'figure in `secondary`')
    ap.add_argument('--no-secondary', action='store_true', help='skip the secondary measurements (per-step API, cold, rotating, noise)')
    ap.add_argument('--eager', action='store_true', help='issue the timed launches eagerly from Python instead of replaying one hipGraph')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--cpu-sample-steps', type=int, default=0, help='0 = auto (about 10-20 s of CPU work)')
    args = ap.parse_args()
    per_cycle = args.mode == 'step'
    if args.steps is None:
        args.steps = 4096 if per_cycle else 64
    if args.warmup is None:
        args.warmup = 256 if per_cycle else 4
    if args.steps < 1 or args.warmup < 0 or args.fuse < 1:
        ap.error('--steps must be >= 1, --warmup >= 0, --fuse >= 1')
    if args.mode == 'graph':
        args.eager = True          # that mode replays its own graph of single-cycle launches
    # cycles per bench step: a whole launch (or graph replay) of `fuse` cycles, or one cycle in step mode
    args.cycles_per_step = 1 if per_cycle else args.fuse
    return args


def cpu_baseline(n_envs, sample_steps, noise=False):
    """Time the CPU oracle (plain-C scalar port of the same algorithm, fp32 build) on the
    host cores of this box, on a bounded sample of the same workload.  kind = "port"."""
    import ctypes as C
    import numpy as np  # noqa: F401
    import oracle as O
    cores_avail = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1)
    threads = max(1, min(cores_avail, 16))
    try:
        gomp = C.CDLL('libgomp.so.1')
    except OSError:
        gomp, threads = None, 1
    cfg = O.make_config(noise=int(noise), **DQN_KWARGS)
    eng = O.OracleEngine(cfg, n_envs, 'f32')

    def run(nthreads, steps):
        if gomp is not None:
            gomp.omp_set_num_threads(int(nthreads))
        eng.reset()
        eng.L.s2do_rollout(eng.h, 2, None, 4, None, None, None, None, None)   # touch memory
        t0 = time.perf_counter()
        eng.L.s2do_rollout(eng.h, steps, None, 4, None, None, None, None, None)
        return time.perf_counter() - t0

    probe = run(1, 4)
    per_step_1t = probe / 4
    s1 = sample_steps or max(8, min(256, int(6.0 / per_step_1t)))
    t1 = run(1, s1)
    v1 = n_envs * s1 / t1
    sN = sample_steps or max(8, min(2048, int(8.0 / (per_step_1t / threads))))
    tN = run(threads, sN) if threads > 1 else t1
    vN = n_envs * sN / tN if threads > 1 else v1
    return {'value': vN, 'unit': 'env-steps/s', 'cores': threads, 'kind': 'port',
            'value_1thread': v1, 'numpy': numpy_baseline(n_envs),
            'sample': f'{n_envs} envs x {sN} steps ({threads} OpenMP threads, {tN:.2f} s) and x {s1} steps '
                      f'(1 thread, {t1:.2f} s); oracle/s2d_oracle.c fp32 build, gcc -O2, same kwargs/seed; '
                      f'reference rcssserver+proxy+gRPC chain not measurable (binaries absent offline)'}


def numpy_baseline(n_envs):
    """the NumPy-vectorised restatement (oracle/s2d_oracle_numpy.py,
    float64) -- what a Python user could do on the host -- on a ~3 s sample of the same workload."""
    sys.path.insert(0, os.path.join(ROOT, 'oracle'))
    import oracle as O
    from s2d_oracle_numpy import NumpyReachBall
    task = dict(O.TASK_DEFAULTS); task.update(DQN_KWARGS)
    eng = NumpyReachBall(n_envs, O.SERVER_DEFAULTS, task)
    eng.reset()
    eng.step(None)
    t0 = time.perf_counter(); eng.step(None); eng.step(None); per = (time.perf_counter() - t0) / 2
    steps = max(4, min(512, int(3.0 / per)))
    t0 = time.perf_counter()
    for _ in range(steps):
        eng.step(None)
    dt = time.perf_counter() - t0
    return {'value': n_envs * steps / dt, 'unit': 'env-steps/s', 'cores': 1, 'kind': 'port-numpy',
            'sample': f'{n_envs} envs x {steps} steps ({dt:.2f} s), float64 NumPy, same kwargs/seed'}


def settle(run_cycles, chunk_cycles, ms):
    """Keep the device under the bench's own load for `ms` milliseconds (untimed) -- see --settle-ms."""
    import torch
    if ms <= 0:
        return 0
    t0, n = time.perf_counter(), 0
    while (time.perf_counter() - t0) * 1e3 < ms:
        run_cycles(chunk_cycles)
        torch.cuda.synchronize()
        n += chunk_cycles
    return n


def init_distributed(rank, local_rank, world):
    """One process per GPU (RCCL = backend 'nccl').  S2D_DIST_BACKEND=gloo + S2D_BENCH_SHARE_GPU=1 is a
    rehearsal mode for a one-GPU box: all ranks use cuda:0 and only the timing max travels over gloo."""
    import torch
    backend = os.environ.get('S2D_DIST_BACKEND', 'nccl')
    share = os.environ.get('S2D_BENCH_SHARE_GPU', '0') == '1'
    idx = 0 if share else local_rank
    torch.cuda.set_device(idx)
    dev = torch.device('cuda', idx)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        if backend == 'nccl':
            dist.init_process_group('nccl', rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    return dev, dist


def max_over_ranks(dist, dev, elapsed):
    if dist is None:
        return elapsed
    import torch
    on = dev if dist.get_backend() == 'nccl' else torch.device('cpu')
    t = torch.tensor([elapsed], dtype=torch.float64, device=on)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def match_cpu_baseline(n_envs):
    """11v11 oracle (oracle/s2d_match_oracle.c, OpenMP over matches) timed on the host cores."""
    import ctypes as C
    import match_oracle as MO
    cores_avail = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1)
    threads = max(1, min(cores_avail, 16))
    try:
        C.CDLL('libgomp.so.1').omp_set_num_threads(threads)
    except OSError:
        threads = 1
    orc = MO.MatchOracle(MO.make_match_config(), n_envs)
    for _ in range(2):
        orc.step(None)
    t0 = time.perf_counter()
    orc.step(None)
    per = time.perf_counter() - t0
    steps = max(4, min(512, int(8.0 / per)))
    t0 = time.perf_counter()
    for _ in range(steps):
        orc.step(None)
    dt = time.perf_counter() - t0
    return {'value': n_envs * steps / dt, 'unit': 'env-steps/s', 'cores': threads, 'kind': 'port',
            'sample': f'{n_envs} matches x {steps} cycles, {threads} OpenMP threads, {dt:.2f} s; oracle/s2d_match_oracle.c fp32'}


def bench_match(args):
    """BASELINE.json configs[3]: 11v11 full-match engine, 8 192 matches per GPU, random policy."""
    import torch
    from soccer2d_amd.match import MatchEngine, make_match_config
    rank = int(os.environ.get('RANK', '0')); local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    dev, dist = init_distributed(rank, local_rank, world)
    n = args.envs if args.envs != 65536 else 8192
    eng = MatchEngine(n, dev, cfg=make_match_config(env_id_offset=rank * n, noise=args.noise))
    T = max(1, args.fuse)
    K, W = args.steps * args.cycles_per_step, args.warmup * args.cycles_per_step     # in cycles
    ro = eng.alloc_rollout(T) if args.mode == 'rollout' else None
    stream = torch.cuda.current_stream(dev)

    def run(k):
        if args.mode == 'rollout':
            full, rem = divmod(k, T)
            for _ in range(full):
                eng.rollout(T, out=ro)
            if rem:
                eng.rollout(rem, out=ro)
            return full + (1 if rem else 0)
        for _ in range(k):
            eng.step(None)
        return k
    settle(run, 8 * T if args.mode == 'rollout' else 512, args.settle_ms)
    run(W)
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record(stream)
    launches = run(K)
    e1.record(stream)
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    elapsed = max_over_ranks(dist, dev, time.perf_counter() - t0)
    state_b, rec_b = 23 * 11 * 4 + 15 * 4, 24 * 5 * 4 + 4 + 4 + 1   # 23 objects x 11 words + 15 game ints; rollout record
    per_launch_steps = T if args.mode == 'rollout' else 1
    alg = n * (2 * state_b + (per_launch_steps * rec_b if args.mode == 'rollout' else 5))
    launch_s = e0.elapsed_time(e1) * 1e-3 / launches
    achieved = alg / launch_s / 1e9
    st = eng.stats.cpu().tolist()
    if rank == 0:
        line = {'metric': 'env-steps/sec, 11v11 full-match engine (22 players, kick/tackle/catch/offside/stamina, player types)',
                'value': world * n * K / elapsed, 'unit': 'env-steps/s', 'n_gpus': world, 'steps': args.steps,
                'warmup': args.warmup, 'ms_per_step': elapsed / args.steps * 1e3, 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
                'dtype': 'f32', 'data': 'synthetic',
                'config': {'workload': f'11v11 match, {n} matches per GPU, random policy (BASELINE.json configs[3])',
                           'envs_per_gpu': n, 'mode': args.mode, 'cycles_per_launch': per_launch_steps, 'settle_ms': args.settle_ms,
                           'player_steps_per_s': world * n * K * 22 / elapsed},
                'roofline': {'bound': 'hbm', 'achieved': achieved, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                             'frac': achieved / HBM_PEAK_GBS, 'traffic': load_traffic('match-' + args.mode, T, n), 'kernel': 's2d_match_rollout_kernel',
                             'launch_us': launch_s * 1e6, 'algorithmic_bytes_per_launch': alg,
                             'algorithmic_bytes_per_env_step': alg / (n * per_launch_steps)},
                'events': {'goals_left': st[1], 'goals_right': st[2], 'matches': st[3], 'kicks': st[4], 'tackles': st[5],
                           'offsides': st[6], 'ball_outs': st[7]}}
        if world == 1 and not args.no_cpu_baseline:
            try:
                line['cpu_baseline'] = match_cpu_baseline(n)
            except Exception as ex:
                line['cpu_baseline'] = {'value': None, 'unit': 'env-steps/s', 'cores': 0, 'kind': 'port', 'sample': f'failed: {ex!r}'}
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def load_traffic(mode, fuse, n_envs):
    """HBM bytes per launch from committed rocprofv3 --pmc passes (profiles/traffic_*.json)."""
    best = None
    pdir = os.path.join(ROOT, 'profiles')
    if not os.path.isdir(pdir):
        return None
    for f in sorted(os.listdir(pdir)):
        if f.startswith('traffic_') and f.endswith('.json'):
            try:
                d = json.load(open(os.path.join(pdir, f)))
            except Exception:
                continue
            for row in d.get('rows', []):
                if row.get('mode') == mode and row.get('fuse') == fuse and row.get('envs') == n_envs:
                    best = row.get('hbm_bytes_per_launch')
    return best


def graph_of(issue):
    """Capture what `issue()` launches into a hipGraph (setup, outside every timed region).  A launch of 64 fused cycles
    takes ~48 us on the device; issued eagerly from Python (ctypes call + stream lookup + hipLaunchKernel, 30-70 us per call
    depending on the host) the HOST would be the slower side and the figure would measure it.  Replaying a graph hands
    the whole sequence to the device at once."""
    import torch
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        issue()
    torch.cuda.synchronize()
    return g


def time_graph(g, stream):
    """seconds of one replay, from one HIP event pair on the launch stream (a pair around every launch would cost ~4 us per
    launch in marker packets)"""
    import torch
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(stream)
    g.replay()
    e1.record(stream)
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e-3


def time_rollout(eng, T, launches, bufs, stream):
    """`launches` rollout launches of T cycles, cycling through the buffers in `bufs`.  Returns seconds (device time)."""
    nb = len(bufs)
    g = graph_of(lambda: [eng.rollout(T, out=bufs[i % nb]) for i in range(launches)])
    return time_graph(g, stream)


def time_steps(eng, k, stream):
    g = graph_of(lambda: [eng.step(None) for _ in range(k)])
    return time_graph(g, stream)


def roofline_of(alg_bytes_launch, launch_s, kernel, traffic, n, steps_per_launch):
    achieved = alg_bytes_launch / launch_s / 1e9
    return {'bound': 'hbm', 'achieved': achieved, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': achieved / HBM_PEAK_GBS,
            'traffic': traffic, 'kernel': kernel, 'launch_us': launch_s * 1e6,
            'algorithmic_bytes_per_launch': alg_bytes_launch,
            'algorithmic_bytes_per_env_step': alg_bytes_launch / (n * steps_per_launch),
            # what the PMC profiles say limits the kernel (profiles/r02/pmc_rollout_instmix.txt): HBM traffic equals the
            # algorithmic bytes; the SIMDs' instruction issue is what is saturated
            'limiter': 'valu-issue' if steps_per_launch > 1 else 'launch-latency'}


def secondary_measurements(args, dev, stream, rank, n, T):
    """More figures for the same workload, carried in the same JSON line (`secondary`): the per-step API, the rollout into
    rotating buffers (> 512 MiB in flight, so that no line of the record can be served by the 256 MiB Infinity Cache) and
    the noise-on rollout (the drop-in default).  Fresh engines, a few seconds in all.  (`rollout_cold`, the figure without the
    settle phase, is measured by main() as the first GPU work of the process.)"""
    import torch
    from soccer2d_amd.engine import Engine, make_config
    out = {}

    def fresh(noise=False):
        e = Engine(n, dev, cfg=make_config(seed=0x5EED, env_id_offset=rank * n, auto_reset=True, noise=noise, **DQN_KWARGS))
        e.reset()
        return e
    alg_roll = n * (2 * STATE_BYTES + T * RECORD_BYTES)
    eng = fresh()
    ro = eng.alloc_rollout(T)
    # (b) rotating buffers, steady clocks
    per_buf = T * n * RECORD_BYTES
    nb = max(2, -(-(600 << 20) // per_buf))
    bufs = [ro] + [eng.alloc_rollout(T) for _ in range(nb - 1)]
    settle(lambda k: [eng.rollout(T, out=bufs[j % nb]) for j in range(k // T)], 16 * T, args.settle_ms)
    dt = time_rollout(eng, T, 8 * nb, bufs, stream)
    out['rollout_rotating_buffers'] = {'value': n * T * 8 * nb / dt, 'unit': 'env-steps/s', 'buffers': nb,
                                       'bytes_in_flight': nb * per_buf, 'launches': 8 * nb,
                                       'roofline': roofline_of(alg_roll, dt / (8 * nb), eng.kernel_name(),
                                                               load_traffic('rollout-rotate', T, n), n, T)}
    del bufs
    # (c) per-step API (what an SB3-style learner drives), steady clocks, 2 048 launches
    settle(lambda k: [eng.step(None) for _ in range(k)], 2048, args.settle_ms)
    dt = time_steps(eng, 2048, stream)
    alg_step = n * (2 * STATE_BYTES + 4 + RECORD_BYTES - 4)
    out['step_api'] = {'value': n * 2048 / dt, 'unit': 'env-steps/s', 'launches': 2048,
                       'roofline': roofline_of(alg_step, dt / 2048, eng.kernel_name(), load_traffic('step', T, n), n, 1)}
    # (c') the same with the caller's actions (what dqn_stable_baselines3.py does: the learner hands a tensor of 65 536 discrete
    # actions to every step; (c) lets the engine draw them, which costs a Philox block per env and step)
    acts = torch.randint(0, DQN_KWARGS['action_space_size'], (n,), device=dev, dtype=torch.int32)
    settle(lambda k: [eng.step(acts) for _ in range(k)], 2048, args.settle_ms)
    dt = time_graph(graph_of(lambda: [eng.step(acts) for _ in range(2048)]), stream)
    out['step_api_caller_actions'] = {'value': n * 2048 / dt, 'unit': 'env-steps/s', 'launches': 2048, 'actions': 'int32[N] device tensor',
                                      'roofline': roofline_of(alg_step, dt / 2048, eng.kernel_name(), None, n, 1)}
    del eng
    # (d) noise on (the drop-in default: rcssserver's stock player_rand / ball_rand)
    eng = fresh(noise=True)
    ro = eng.alloc_rollout(T)
    settle(lambda k: [eng.rollout(T, out=ro) for _ in range(k // T)], 16 * T, args.settle_ms)
    dt = time_rollout(eng, T, 64, [ro], stream)
    out['rollout_noise_on'] = {'value': n * T * 64 / dt, 'unit': 'env-steps/s', 'launches': 64,
                               'roofline': roofline_of(alg_roll, dt / 64, eng.kernel_name(), None, n, T)}
    return out


def main():
    args = parse()
    if args.task == 'match':
        return bench_match(args)
    import torch
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit('bench.py --gpus N>1 must be launched with torch.distributed.run (one process per GPU)')
        args.gpus = world
    assert torch.cuda.is_available(), 'bench.py needs MI355X GPUs'
    dev, dist = init_distributed(rank, local_rank, world)

    from soccer2d_amd.engine import Engine, make_config
    n = args.envs
    kw, sp, auto = dict(DQN_KWARGS), None, True
    if args.variant == 'no-auto-reset':
        auto = False
    elif args.variant == 'never-done':
        kw.update(max_steps=1000000, min_distance_to_ball=0.0)
        sp = dict(pitch_half_length=1e6, pitch_half_width=1e6)
    cfg = make_config(seed=0x5EED, env_id_offset=rank * n, auto_reset=auto, noise=args.noise, server_params=sp, **kw)
    eng = Engine(n, dev, cfg=cfg)
    eng.reset()
    T = max(1, args.fuse)
    K, W = args.steps * args.cycles_per_step, args.warmup * args.cycles_per_step     # in cycles
    nbuf = max(1, args.rotate_buffers)
    bufs = [eng.alloc_rollout(T) for _ in range(nbuf)] if args.mode == 'rollout' else None
    stream = torch.cuda.current_stream(dev)
    issued = [0]
    cold = None
    if world == 1 and not args.no_secondary and args.mode == 'rollout' and args.variant == 'dqn':
        # the cold figure: the first GPU work of this process -- 4 warm-up launches, 20 timed, no settle phase (the chip's
        # clock has not settled: profiles/r01/duration_sweep.txt)
        for _ in range(4):
            eng.rollout(T, out=bufs[0])
        dt = time_rollout(eng, T, 20, bufs[:1], stream)
        cold = {'value': n * T * 20 / dt, 'unit': 'env-steps/s', 'launches': 20, 'settle_ms': 0,
                'roofline': roofline_of(n * (2 * STATE_BYTES + T * RECORD_BYTES), dt / 20, eng.kernel_name(), None, n, T)}

    graph = None
    if args.mode == 'graph':
        side = torch.cuda.Stream(dev)
        with torch.cuda.stream(side):
            for _ in range(3):
                eng.step(None)
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            for _ in range(T):
                eng.step(None)

    def launch(k):
        """run k cycles; returns number of launches issued"""
        if args.mode == 'rollout':
            full, rem = divmod(k, T)
            for _ in range(full):
                eng.rollout(T, out=bufs[issued[0] % nbuf]); issued[0] += 1
            if rem:
                eng.rollout(rem, out=bufs[issued[0] % nbuf]); issued[0] += 1
            return full + (1 if rem else 0)
        if args.mode == 'graph':
            full, rem = divmod(k, T)
            for _ in range(full):
                graph.replay()
            for _ in range(rem):
                eng.step(None)
            return full * T + rem
        for _ in range(k):
            eng.step(None)
        return k

    settle(launch, 16 * T if args.mode != 'step' else 2048, args.settle_ms)
    launch(W)
    torch.cuda.synchronize()
    # the K timed cycles as ONE hipGraph (captured here, outside the timed region; see graph_of) unless --eager
    timed = None
    n_launches = [0]
    if not args.eager:
        timed = graph_of(lambda: n_launches.__setitem__(0, launch(K)))
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()

    # ---- timed region: EXACTLY K cycles ----
    t0 = time.perf_counter()
    # ONE event pair over the timed region, recorded on the launch stream
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(stream)
    if timed is not None:
        timed.replay()
    else:
        n_launches[0] = launch(K)
    e1.record(stream)
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    n_launches = n_launches[0]

    elapsed = max_over_ranks(dist, dev, elapsed)

    # dominant-kernel launch duration from HIP events on the launch stream
    if args.mode == 'rollout':
        steps_per_launch = T if K >= T else K
        # average duration of a full launch; a trailing short launch (K not a multiple of T) counts by its share of cycles
        launch_s = e0.elapsed_time(e1) * 1e-3 * steps_per_launch / K
        alg_bytes_launch = n * (2 * STATE_BYTES + steps_per_launch * RECORD_BYTES)
        kernel = eng.kernel_name() or 's2d_reach_rollout_kernel'
    else:
        launch_s = e0.elapsed_time(e1) * 1e-3 / K
        steps_per_launch = 1
        alg_bytes_launch = n * (2 * STATE_BYTES + 4 + RECORD_BYTES - 4)      # 186 B per env-step
        kernel = eng.kernel_name() or 's2d_reach_step_kernel'
    traffic_key = args.mode if nbuf == 1 else args.mode + '-rotate'

    total_steps = world * n * K
    stats = eng.stats.cpu().tolist()
    if rank == 0:
        line = {
            'metric': 'env-steps/sec at 65 536 parallel reach_ball envs per MI355X',
            'value': total_steps / elapsed,
            'unit': 'env-steps/s',
            'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': elapsed / args.steps * 1e3,
            'higher_is_better': True,
            'scaling': 'weak',
            'vs_baseline': None,
            'dtype': 'f32',
            'data': 'synthetic',
            'config': {'workload': f'reach_ball_env, {n} envs per GPU, random-policy rollouts '
                                   f'(BASELINE.json configs[2]; kwargs of dqn_stable_baselines3.py:18-31)',
                       'envs_per_gpu': n, 'global_envs': world * n, 'mode': args.mode, 'variant': args.variant,
                       'cycles_per_launch': steps_per_launch if args.mode != 'graph' else f'1 ({T} per graph replay)',
                       'settle_ms': args.settle_ms, 'rollout_buffers': nbuf, 'issue': 'eager' if args.eager else 'one hipGraph replay',
                       'noise': bool(args.noise), 'parallelism': f'env-shard x{world} (no collective)',
                       'launches': n_launches},
            'roofline': roofline_of(alg_bytes_launch, launch_s, kernel, load_traffic(traffic_key, T, n), n, steps_per_launch),
            'episodes': {'goal': stats[1], 'out': stats[2], 'timeout': stats[3]},
        }
        if world == 1 and not args.no_secondary and args.mode == 'rollout' and args.variant == 'dqn':
            del eng, bufs
            try:
                line['secondary'] = secondary_measurements(args, dev, stream, rank, n, T)
                line['secondary']['rollout_cold'] = cold
            except Exception as ex:
                line['secondary'] = {'error': repr(ex), 'rollout_cold': cold}
        if world == 1 and not args.no_cpu_baseline:
            try:
                line['cpu_baseline'] = cpu_baseline(n, args.cpu_sample_steps, bool(args.noise))
            except Exception as ex:     # the baseline is a report, never the product
                line['cpu_baseline'] = {'value': None, 'unit': 'env-steps/s', 'cores': 0, 'kind': 'port',
                                        'sample': f'failed: {ex!r}'}
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
