#!/usr/bin/env python3
"""bench.py -- env-steps/sec of the reach_ball hot path on N MI355X (BASELINE.json metric).

One simulator CYCLE for every env of the batch = action decode -> dash/turn -> stamina ->
integrate -> collide -> decay -> observation -> reward/done/result -> auto-reset, with the
rollout record of that cycle (obs[10], action, reward, done, result per env) written to HBM.
A bench "step" (--steps K / --warmup W) is one pass of the hot path over the batch as the mode
issues it: ONE LAUNCH of T = 256 fused cycles (--fuse) in the default rollout mode, one cycle in step
mode.  `value` is env-steps (env-cycles) per second in every mode: N envs x cycles / seconds.
Inputs are resident in HBM before the timed region; nothing is copied to the host inside it.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--envs E] [--task reach_ball|match]
                  [--mode rollout|step|graph] [--fuse T] [--noise] [--league-exchange]

Workload naming (BASELINE.json `configs`): --envs 65536 = configs[2] (default; kwargs of
dqn_stable_baselines3.py:18-31, uniform random policy drawn in-kernel, reset distribution of
reach_ball_env.py:170-218), --envs 4096 = configs[1], --task match (8 192 matches) = configs[3],
--gpus N --league-exchange = configs[4].  Every other size is named "custom".

Protocol: settle phase, W untimed steps, graph capture, >= 3 and >= 100 ms of untimed replays
(the first capture of a process leaves the device idle for ~25 ms and its clock takes ~25 ms of load to come
back: profiles/r04/clock_trajectory.txt), then R = 5 timed regions of EXACTLY
K steps each (one hipGraph replay per region, bracketed by barrier + synchronize on both sides, max
over ranks); `value` is the MEDIAN region, the five figures are kept in `repeats`.  The default record
is written into ROTATING buffers (> 512 MiB in flight, so no line of it can live in the 256 MiB
Infinity Cache): the headline is the pure-HBM figure; the one-buffer figure is in `secondary`.

Multi-GPU: one process per GPU, contiguous global env-id ranges, NO data-path collective (envs
are independent) -> weak scaling; only the timing max is all-reduced.  `python bench.py --gpus N`
without a torch.distributed environment launches its own N ranks (a parent that never touches the
GPU starts `python -m torch.distributed.run ... bench.py` as a child and relays rank 0's line); on
a box with fewer than N GPUs the same entry point runs a REHEARSAL (ranks share cuda:0, gloo) and
says so in `config.rehearsal`.  --league-exchange adds the one collective the path has: the
all-gather of every rank's rollout slab (RCCL over xGMI) on a side stream, overlapped with the
next rollout (BASELINE.json configs[4]).
"""
import argparse
import json
import os
import statistics
import subprocess
import sys
import threading
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for _p in (os.path.join(ROOT, 'gym-soccer-2d-env_amd'), os.path.join(ROOT, 'tests'), ROOT):
    if _p not in sys.path:
        sys.path.insert(0, _p)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)
XGMI_LINK_GBS = 153.0          # per link and direction; 7 links per GPU
STATE_BYTES = 68               # 17 words per env
RECORD_BYTES = 50              # obs 40 + action 4 + reward 4 + done 1 + result 1 per env-step
# 11v11: 23 objects x 11 words + 15 game ints of state; rollout record 24 x 5 words + reward + mode + done
MATCH_STATE_BYTES, MATCH_RECORD_BYTES = 23 * 11 * 4 + 15 * 4, 24 * 5 * 4 + 4 + 4 + 1
# VALU issue peak of one SIMD with >= 2 resident waves: 1.08 ns per wave-instruction (profiles/r01/instr_rate_gfx950.txt,
# v_fma_f32 at 2 and 4 waves per SIMD) -> 256 CUs x 4 SIMDs / 1.08 ns
VALU_PEAK_GINSTR = 256 * 4 / 1.08
DQN_KWARGS = dict(change_ball_position=True, change_ball_velocity=True, min_distance_to_ball=5.0,
                  max_steps=200, use_continuous_action=False, action_space_size=16, use_turning=False)
# ddpg_stable_baselines3.py:18-31 (1-D continuous actions, ball at rest at the centre spot); the 4-D turning mode = the same with use_turning
DDPG_KWARGS = dict(change_ball_position=False, change_ball_velocity=False, ball_position_x=0, ball_position_y=0, ball_speed=0,
                   ball_direction=0, min_distance_to_ball=5.0, max_steps=200, action_space_size=16, use_continuous_action=True,
                   use_turning=False)
REPEATS = 5


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=None, help='timed bench steps per region (default 32 launches; 4096 in step mode)')
    ap.add_argument('--warmup', type=int, default=None, help='untimed bench steps (default 4 launches; 256 in step mode)')
    ap.add_argument('--envs', type=int, default=None, help='envs per GPU (default 65 536; 8 192 matches for --task match)')
    ap.add_argument('--mode', choices=('rollout', 'step', 'graph'), default='rollout')
    ap.add_argument('--fuse', type=int, default=256, help='cycles per launch (rollout) / per graph (graph); 256 = a 256-step rollout per launch')
    ap.add_argument('--noise', action='store_true', help='player_rand/ball_rand Philox noise on (the drop-in default of the product)')
    ap.add_argument('--task', choices=('reach_ball', 'match'), default='reach_ball',
                    help='reach_ball = the BASELINE.json metric (default); match = 11v11 engine, configs[3] (8 192 matches)')
    ap.add_argument('--variant', choices=('dqn', 'no-auto-reset', 'never-done'), default='dqn',
                    help='experiments only: dqn = the benchmark workload; the others switch episode ends off')
    ap.add_argument('--settle-ms', type=float, default=200.0,
                    help='untimed load before the W warm-up steps so that the chip\'s clock has settled; 0 = off')
    ap.add_argument('--rotate-buffers', type=int, default=0,
                    help='rollout mode: cycle through this many record buffers; 0 = as many as put > 600 MiB in flight (the '
                         'default: every launch writes lines the 256 MiB Infinity Cache cannot hold), 1 = re-write one buffer')
    ap.add_argument('--repeats', type=int, default=REPEATS, help='timed regions of K steps each; the median is reported')
    ap.add_argument('--match-phase', choices=('spread', 'lockstep'), default='spread',
                    help='match task: spread = every match starts at its own match time (restarts do not coincide); lockstep = all at 0')
    ap.add_argument('--league-exchange', action='store_true',
                    help='configs[4]: every launch writes a slab that is all-gathered over the ranks (RCCL) on a side stream')
    ap.add_argument('--league-payload', choices=('record', 'league'), default='record',
                    help='what the league all-gather carries: record = all five arrays (50 B per env-step), league = action, reward, '
                         'done, result (10 B per env-step; observations stay on the rank that trains on them)')
    ap.add_argument('--no-secondary', action='store_true', help='skip the secondary measurements')
    ap.add_argument('--eager', action='store_true', help='issue the timed launches eagerly from Python instead of replaying one hipGraph')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--cpu-sample-steps', type=int, default=0, help='0 = auto (about 10-20 s of CPU work)')
    args = ap.parse_args(argv)
    per_cycle = args.mode == 'step'
    if args.steps is None:
        args.steps = 4096 if per_cycle else 32
    if args.warmup is None:
        args.warmup = 256 if per_cycle else 4
    if args.envs is None:
        args.envs = 8192 if args.task == 'match' else 65536
    if args.steps < 1 or args.warmup < 0 or args.fuse < 1 or args.repeats < 1 or args.envs < 1:
        ap.error('--steps, --fuse, --repeats, --envs must be >= 1, --warmup >= 0')
    if args.mode == 'graph':
        args.eager = True          # that mode replays its own graph of single-cycle launches
    args.cycles_per_step = 1 if per_cycle else args.fuse
    return args


# --------------------------------------------------------------------------------------------------------------------
# naming
# --------------------------------------------------------------------------------------------------------------------
def workload_of(task, n, world, league):
    """(metric, config.workload) named after BASELINE.json's configs; sizes BASELINE does not list are 'custom'."""
    if task == 'match':
        tag = 'BASELINE.json configs[3]' if n == 8192 else 'custom size'
        return (f'env-steps/sec, 11v11 full-match engine (22 players, kick/tackle/catch/offside/stamina, player types), {n} matches per MI355X',
                f'11v11 full-match env, {n} matches per GPU, random policy ({tag})')
    if league:
        tag = 'BASELINE.json configs[4]' if (n == 65536 and world == 8) else 'configs[4] shape at another size'
        return (f'env-steps/sec at {n} parallel reach_ball envs per MI355X with the league all-gather of rollouts, {world} MI355X',
                f'reach_ball_env, {n}x{world} envs sharded across {world} GPUs, all-gather of rollout slabs for a self-play league ({tag})')
    tag = {65536: 'BASELINE.json configs[2]', 4096: 'BASELINE.json configs[1]'}.get(n, 'custom size')
    return (f'env-steps/sec at {n} parallel reach_ball envs per MI355X',
            f'reach_ball_env, {n} envs per GPU, random-policy rollouts ({tag}; kwargs of dqn_stable_baselines3.py:18-31)')


# --------------------------------------------------------------------------------------------------------------------
# CPU baselines (the oracle is the checker / the reported baseline here, never the product)
# --------------------------------------------------------------------------------------------------------------------
def cpu_baseline(n_envs, sample_steps, noise=False):
    """Time the CPU oracle (plain-C scalar port of the same algorithm, fp32 build) on the
    host cores of this box, on a bounded sample of the same workload.  kind = "port"."""
    import ctypes as C
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
    s1 = sample_steps or max(8, min(1024, int(6.0 / per_step_1t)))
    t1 = run(1, s1)
    v1 = n_envs * s1 / t1
    sN = sample_steps or max(8, min(16384, int(8.0 / (per_step_1t / threads))))
    tN = run(threads, sN) if threads > 1 else t1
    vN = n_envs * sN / tN if threads > 1 else v1
    return {'value': vN, 'unit': 'env-steps/s', 'cores': threads, 'kind': 'port',
            'value_1thread': v1, 'numpy': numpy_baseline(n_envs),
            'sample': f'{n_envs} envs x {sN} steps ({threads} OpenMP threads, {tN:.2f} s) and x {s1} steps '
                      f'(1 thread, {t1:.2f} s); oracle/s2d_oracle.c fp32 build, gcc -O2, same kwargs/seed, noise {"on" if noise else "off"}; '
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


# --------------------------------------------------------------------------------------------------------------------
# process / device plumbing
# --------------------------------------------------------------------------------------------------------------------
def self_launch(args, argv):
    """`python bench.py --gpus N` (N > 1) outside a torch.distributed environment: start the N ranks ourselves.
    This parent makes NO GPU call (it does not even import torch): the ranks are CHILD processes of
    `python -m torch.distributed.run`, and rank 0's JSON line is relayed.  The visible GPUs are counted in a child too."""
    import socket
    probe = subprocess.run([sys.executable, '-c', 'import torch; print(torch.cuda.device_count())'],
                           stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True)
    try:
        ndev = int(probe.stdout.strip().splitlines()[-1])
    except Exception:
        ndev = 0
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    if ndev < args.gpus:
        if ndev < 1 or args.gpus > 6:
            sys.exit(f'bench.py --gpus {args.gpus}: {ndev} GPU(s) visible; the shared-GPU rehearsal runs at most 6 ranks on one card')
        # rehearsal on a smaller box: every rank uses cuda:0, collectives travel over gloo (staged through the host)
        env['S2D_DIST_BACKEND'], env['S2D_BENCH_SHARE_GPU'] = 'gloo', '1'
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={args.gpus}',
           '--master-addr', '127.0.0.1', '--master-port', str(port), os.path.abspath(__file__)] + list(argv)
    r = subprocess.run(cmd, stdout=subprocess.PIPE, text=True, env=env)
    line = None
    for ln in r.stdout.splitlines():
        if ln.startswith('{') and '"metric"' in ln:
            line = ln
        elif ln.strip():
            print(ln, file=sys.stderr)
    if line is not None:
        print(line, flush=True)
    if r.returncode != 0 or line is None:
        sys.exit(r.returncode or 1)


def init_distributed(rank, local_rank, world):
    """One process per GPU (RCCL = backend 'nccl').  S2D_DIST_BACKEND=gloo + S2D_BENCH_SHARE_GPU=1 is the
    rehearsal mode for a box with fewer GPUs than ranks: all ranks use cuda:0, collectives go over gloo."""
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


def max_over_ranks(dist, dev, values):
    """element-wise max over ranks of a list of floats (the timing all-reduce; the only collective of the default mode)"""
    if dist is None:
        return list(values)
    import torch
    on = dev if dist.get_backend() == 'nccl' else torch.device('cpu')
    t = torch.tensor(list(values), dtype=torch.float64, device=on)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return [float(v) for v in t.cpu().tolist()]


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


def graph_of(issue):
    """Capture what `issue()` launches into a hipGraph (setup, outside every timed region).  A launch of 64 fused cycles
    takes ~45 us on the device; issued eagerly from Python (ctypes call + stream lookup + hipLaunchKernel, 30-70 us per call
    depending on the host) the HOST would be the slower side and the figure would measure it.  Replaying a graph hands
    the whole sequence to the device at once."""
    import torch
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        issue()
    torch.cuda.synchronize()
    return g


WARM_REGIONS = 3
WARM_MS = 100.0


def timed_regions(run, repeats, stream, dist, dev, warm_regions=WARM_REGIONS):
    """`repeats` timed regions of one `run()` each: barrier + synchronize on both sides, wall clock and ONE HIP event pair on the
    launch stream per region.  Returns (wall seconds per region, max over ranks; event seconds per region, this rank).
    `warm_regions` untimed runs come first: capturing the graph leaves the device idle for tens of milliseconds and its clock ramps
    up again over the next ~10 ms of load -- round 3's five timed regions of every line rose monotonically (87 -> 102 G) because the
    first two were that ramp, not the kernel."""
    import torch
    wall, evs = [], []
    if warm_regions > 0:
        # ... and with short regions (the driver's --steps 20 = 3.3 ms) three of them are not enough: the ramp lasts ~25 ms
        # (profiles/r04/clock_trajectory.txt: 88 78 88 92 95 100 100 101 then 104 +- 3 G for seconds, the same after every idle
        # gap of >= 50 ms; the first graph capture of a process is such a gap, 24 ms).  So: at least WARM_MS of untimed replays.
        t0, done = time.perf_counter(), 0
        while done < warm_regions or (time.perf_counter() - t0) * 1e3 < WARM_MS:
            run()
            torch.cuda.synchronize()
            done += 1
    for _ in range(repeats):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        e0.record(stream)
        run()
        e1.record(stream)
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()
        wall.append(time.perf_counter() - t0)
        evs.append(e0.elapsed_time(e1) * 1e-3)
    return max_over_ranks(dist, dev, wall), evs


def median_of(xs):
    return float(statistics.median(xs))


def load_traffic(mode, fuse, n_envs):
    """(HBM bytes per launch, source file) from the committed rocprofv3 --pmc passes (profiles/traffic_*.json): the PMC
    counters need their own profiler runs (MI355X_MICROARCH.md), so this run cannot measure them itself."""
    best, src = None, None
    pdir = os.path.join(ROOT, 'profiles')
    if not os.path.isdir(pdir):
        return None, None
    for f in sorted(os.listdir(pdir)):
        if f.startswith('traffic_') and f.endswith('.json'):
            try:
                d = json.load(open(os.path.join(pdir, f)))
            except Exception:
                continue
            for row in d.get('rows', []):
                if row.get('mode') == mode and row.get('fuse') == fuse and row.get('envs') == n_envs:
                    best, src = row.get('hbm_bytes_per_launch'), 'profiles/' + f
    return best, src


def roofline_of(alg_bytes_launch, launch_s, launch_s_events, kernel, traffic_key, fuse, n, steps_per_launch):
    """HBM roofline of the dominant kernel.  `frac` uses the SAME clock as `value` (wall time of the timed region / launches);
    `frac_events` the HIP-event duration of the same region."""
    achieved = alg_bytes_launch / launch_s / 1e9
    traffic, src = load_traffic(traffic_key, fuse, n) if traffic_key else (None, None)
    r = {'bound': 'hbm', 'achieved': achieved, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': achieved / HBM_PEAK_GBS,
         'traffic': traffic, 'traffic_source': src if src else 'none for this configuration (PMC passes are separate rocprofv3 runs)',
         'kernel': kernel, 'launch_us': launch_s * 1e6,
         'algorithmic_bytes_per_launch': alg_bytes_launch,
         'algorithmic_bytes_per_env_step': alg_bytes_launch / (n * steps_per_launch),
         # what limits the kernel (HBM traffic equals the algorithmic bytes): in the fused kernels the slowest workgroup's dependent
         # chain at the clock the chip grants, launch + memory latency in the one-cycle kernel
         'limiter': ("slowest workgroup's instruction chain x the shader clock granted at the socket's 1400 W power cap, which this kernel reaches "
                     "(about 2.0 of 2.4 GHz on the faster boxes, profiles/r04/power_probe.txt; DESIGN section 7)"
                     if steps_per_launch > 1 else 'launch-latency')}
    if launch_s_events:
        r['launch_us_events'] = launch_s_events * 1e6
        r['frac_events'] = alg_bytes_launch / launch_s_events / 1e9 / HBM_PEAK_GBS
    return r


# --------------------------------------------------------------------------------------------------------------------
# reach_ball measurements
# --------------------------------------------------------------------------------------------------------------------
def reach_engine(n, dev, rank, noise, variant='dqn', task_kwargs=None):
    from soccer2d_amd.engine import Engine, make_config
    kw, sp, auto = dict(task_kwargs or DQN_KWARGS), None, True
    if variant == 'no-auto-reset':
        auto = False
    elif variant == 'never-done':
        kw.update(max_steps=1000000, min_distance_to_ball=0.0)
        sp = dict(pitch_half_length=1e6, pitch_half_width=1e6)
    eng = Engine(n, dev, cfg=make_config(seed=0x5EED, env_id_offset=rank * n, auto_reset=auto, noise=noise, server_params=sp, **kw))
    eng.reset()
    return eng


def n_rotating(per_buf_bytes):
    return max(2, -(-(600 << 20) // per_buf_bytes))


def measure_rollout(eng, T, launches, nbuf, repeats, stream, settle_ms, dist=None, dev=None, warm=4, warm_regions=WARM_REGIONS):
    """`repeats` regions of `launches` rollout launches of T cycles each, cycling through nbuf record buffers.
    Returns dict(value-free raw figures): median wall / event seconds per launch, the per-region figures."""
    n = eng.num_envs
    bufs = [eng.alloc_rollout(T) for _ in range(nbuf)]
    k = [0]

    def issue(cnt):
        for _ in range(cnt):
            eng.rollout(T, out=bufs[k[0] % nbuf]); k[0] += 1
    settle(lambda c: issue(max(1, c // T)), 16 * T, settle_ms)
    issue(warm)
    import torch
    torch.cuda.synchronize()
    g = graph_of(lambda: issue(launches))
    wall, evs = timed_regions(g.replay, repeats, stream, dist, dev, warm_regions)
    rec = RECORD_BYTES + (12 if (eng.cfg.task.use_continuous_action and eng.cfg.task.use_turning) else 0)   # float[4] actions
    alg = n * (2 * STATE_BYTES + T * rec)
    return {'wall': wall, 'events': evs, 'launch_s': median_of(wall) / launches, 'launch_s_events': median_of(evs) / launches,
            'alg_bytes_launch': alg, 'launches': launches, 'buffers': nbuf, 'bytes_in_flight': nbuf * T * n * rec}


def rollout_entry(m, n, T, kernel, traffic_key):
    """secondary-style entry from a measure_rollout() result"""
    return {'value': n * T / m['launch_s'], 'unit': 'env-steps/s', 'launches_per_region': m['launches'], 'buffers': m['buffers'],
            'bytes_in_flight': m['bytes_in_flight'],
            'repeats': [n * T * m['launches'] / w for w in m['wall']],
            'roofline': roofline_of(m['alg_bytes_launch'], m['launch_s'], m['launch_s_events'], kernel, traffic_key, T, n, T)}


def measure_steps(eng, launches, repeats, stream, settle_ms, actions=None):
    import torch
    n = eng.num_envs
    settle(lambda c: [eng.step(actions) for _ in range(c)], 2048, settle_ms)
    g = graph_of(lambda: [eng.step(actions) for _ in range(launches)])
    wall, evs = timed_regions(g.replay, repeats, stream, None, None)
    alg = n * (2 * STATE_BYTES + 4 + RECORD_BYTES - 4)      # 186 B per env-step
    ls, le = median_of(wall) / launches, median_of(evs) / launches
    return {'value': n / ls, 'unit': 'env-steps/s', 'launches_per_region': launches,
            'repeats': [n * launches / w for w in wall],
            'roofline': roofline_of(alg, ls, le, eng.kernel_name(), 'step', 64, n, 1)}


def measure_steps_with_policy(eng, launches, repeats, stream, settle_ms):
    """BASELINE configs[2] read literally -- "obs as PyTorch-ROCm tensors into dqn_stable_baselines3": every step a Q-network of
    SB3's DQN MlpPolicy shape (10-64-64-16, ReLU; random weights, fp32, plain torch ops) reads the engine's observation tensor,
    its greedy action (int64 [N]) goes back into s2d_step; one hipGraph of `launches` such steps per region.  The network is torch's,
    not this repo's: the figure says what the per-step API costs NEXT TO a learner's forward pass."""
    import torch
    n = eng.num_envs
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Linear(10, 64), torch.nn.ReLU(), torch.nn.Linear(64, 64), torch.nn.ReLU(),
                              torch.nn.Linear(64, DQN_KWARGS['action_space_size'])).to(eng.device)

    def one():
        with torch.no_grad():
            eng.step(net(eng.obs).argmax(dim=1))
    settle(lambda c: [one() for _ in range(c)], 256, settle_ms)
    g = graph_of(lambda: [one() for _ in range(launches)])
    wall, evs = timed_regions(g.replay, repeats, stream, None, None)
    per = median_of(wall) / launches
    return {'value': n / per, 'unit': 'env-steps/s', 'us_per_step': per * 1e6, 'launches_per_region': launches,
            'repeats': [n * launches / w for w in wall],
            'policy': 'greedy Q-network 10-64-64-16 (SB3 DQN MlpPolicy shape), torch fp32 ops in the same hipGraph'}


def measure_step_k(eng, k, launches, repeats, stream, settle_ms):
    """s2d_step_k: k cycles of the per-step API per launch, caller actions [k][N] (int32, resident), record written into one [k][N]
    buffer set.  Returns us per launch and per cycle."""
    import torch
    n = eng.num_envs
    acts = torch.randint(0, DQN_KWARGS['action_space_size'], (k, n), device=eng.device, dtype=torch.int32)
    out = eng.alloc_rollout(k)
    settle(lambda c: [eng.step_k(k, acts, out=out) for _ in range(max(1, c // k))], 2048, settle_ms)
    g = graph_of(lambda: [eng.step_k(k, acts, out=out) for _ in range(launches)])
    wall, evs = timed_regions(g.replay, repeats, stream, None, None)
    ls = median_of(wall) / launches
    return {'k': k, 'us_per_launch': ls * 1e6, 'us_per_cycle': ls * 1e6 / k, 'value': n * k / ls, 'unit': 'env-steps/s',
            'repeats_us_per_cycle': [w / launches / k * 1e6 for w in wall], 'kernel': eng.kernel_name(),
            'record': 'obs, action, reward, done, result of every cycle ([k][N]); caller actions int32 [k][N]'}


def measure_match(n, dev, rank, T, launches, repeats, stream, settle_ms, noise=False, phase='spread', mode='rollout', dist=None, match_kw=None):
    """11v11 engine: `launches` rollout launches of T cycles (or single-cycle launches in step mode) per region."""
    import torch
    from soccer2d_amd.match import MatchEngine, make_match_config
    extra = dict(match_kw or {})
    extra.update(json.loads(os.environ.get('S2D_MATCH_KW', '{}')))      # experiments: match parameter overrides
    eng = MatchEngine(n, dev, cfg=make_match_config(env_id_offset=rank * n, noise=noise, **extra))
    eng.reset()
    if phase == 'spread':
        # every match at its own (even) match time in the first half: half-time and time-over restarts -- and the ~120 expensive
        # cycles after each -- do not coincide across the batch, as in any batch that has run for a while
        g = torch.Generator(device='cpu').manual_seed(1234 + rank)
        eng.cycle += (2 * torch.randint(0, 1500, (n,), generator=g, dtype=torch.int32)).to(dev)
    ro = eng.alloc_rollout(T) if mode == 'rollout' else None
    per_launch = T if mode == 'rollout' else 1

    def issue(cnt):
        for _ in range(cnt):
            if mode == 'rollout':
                eng.rollout(T, out=ro)
            else:
                eng.step(None)
    settle(lambda c: issue(max(1, c // per_launch)), 8 * T if mode == 'rollout' else 512, settle_ms)
    issue(2)
    torch.cuda.synchronize()
    gr = graph_of(lambda: issue(launches))
    wall, evs = timed_regions(gr.replay, repeats, stream, dist, dev)
    ls, le = median_of(wall) / launches, median_of(evs) / launches
    alg = n * (2 * MATCH_STATE_BYTES + (per_launch * MATCH_RECORD_BYTES if mode == 'rollout' else 5))
    st = eng.stats.cpu().tolist()
    # SURVEY 8(d) prices every kernel of the path against the HBM roofline, so that is `roofline` here too: algorithmic bytes (state
    # round trip + the 489 B record per match-step) over the launch time against 8 TB/s.  The kernel is far from it (0.12) because
    # it is bound by instruction issue; that is kept as a SECONDARY model (`issue_model`), counting only the VALU instructions of
    # the committed PMC mix against the VALU issue peak of the SIMDs -- read it with `value`: the same work in fewer instructions
    # lowers it while the rate rises.
    valu_per_wave_cycle, src, mix = match_valu_per_wave_cycle()
    waves = -(-n // 2)
    tr, tsrc = load_traffic('match-' + mode, T, n)
    roof = {'bound': 'hbm', 'achieved': alg / ls / 1e9, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': alg / ls / 1e9 / HBM_PEAK_GBS,
            'traffic': tr, 'traffic_source': tsrc or 'none for this configuration',
            'kernel': 's2d_match_rollout_kernel' if mode == 'rollout' else 's2d_match_step_kernel', 'kernel_variant': eng.kernel_name(),
            'launch_us': ls * 1e6, 'launch_us_events': le * 1e6,
            'algorithmic_bytes_per_launch': alg, 'algorithmic_bytes_per_env_step': alg / (n * per_launch),
            'limiter': 'instruction issue (see issue_model): 22 players + ball per match on a half wave, ~450 VALU instructions per wave-cycle',
            'issue_model': {'bound': 'valu-issue', 'unit': 'G VALU wave-instr/s', 'peak': VALU_PEAK_GINSTR,
                            'peak_source': 'profiles/r01/instr_rate_gfx950.txt (1.08 ns per VALU instruction and SIMD at >= 2 waves per SIMD; 1024 SIMDs)',
                            'valu_per_wave_cycle': valu_per_wave_cycle, 'instructions_source': src, 'instruction_mix': mix}}
    if valu_per_wave_cycle:
        roof['issue_model']['achieved'] = waves * per_launch * valu_per_wave_cycle / ls / 1e9
        roof['issue_model']['frac'] = roof['issue_model']['achieved'] / VALU_PEAK_GINSTR
    return {'value': n * per_launch / ls, 'unit': 'env-steps/s', 'launches_per_region': launches, 'cycles_per_launch': per_launch,
            'phase': phase, 'repeats': [n * per_launch * launches / w for w in wall], 'wall': wall,
            'player_steps_per_s': 22 * n * per_launch / ls, 'roofline': roof,
            'events': {'goals_left': st[1], 'goals_right': st[2], 'matches': st[3], 'kicks': st[4], 'tackles': st[5],
                       'offsides': st[6], 'ball_outs': st[7]}}


def match_valu_per_wave_cycle():
    """VALU instructions per wave and cycle of the 11v11 kernel from the committed PMC mix (the newest round that has one)."""
    for rnd in ('r04', 'r03', 'r02', 'r01'):
        f = os.path.join(ROOT, 'profiles', rnd, 'pmc_match_instmix.json')
        if os.path.exists(f):
            try:
                d = json.load(open(f))
                return float(d['valu']), f'profiles/{rnd}/pmc_match_instmix.json', {k: d[k] for k in ('valu', 'salu', 'lds') if k in d}
            except Exception:
                pass
    return None, 'no committed PMC mix', None


class _EnvShim:
    """what LeagueRolloutExchange needs of a vec env"""
    def __init__(self, eng):
        self.engine, self.device = eng, eng.device


def xgmi_bound(world, n, T, bytes_sent):
    """What the links allow: a direct all-gather sends this rank's slab once over each of its world - 1 xGMI links (all at the same
    time), so an exchange cannot take less than bytes_sent / 153 GB/s, and the job's env-steps/s with one exchange per launch cannot
    exceed world x N x T / that -- whatever the simulation rate."""
    return world * n * T / (bytes_sent / (XGMI_LINK_GBS * 1e9))


def measure_league_exchange(eng, T, steps, warm, dist, dev, world, payload='record'):
    """configs[4]: `steps` rounds of {rollout of T cycles into a slab; all-gather of that slab on the side stream, overlapped with
    the next rollout}.  Returns env-steps/s with the exchange, the exchange's own duration (side-stream HIP events) and what
    that is against the xGMI links."""
    import torch
    from soccer2d_amd.dist import LEAGUE_FIELDS, LeagueRolloutExchange
    ex = LeagueRolloutExchange(_EnvShim(eng), T, timing=True, fields=LEAGUE_FIELDS if payload == 'league' else None)
    for _ in range(warm):
        ex.step()
    ex.flush()
    torch.cuda.synchronize()
    ex.timings.clear()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        ex.step()
    ex.flush()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
        torch.cuda.synchronize()
    elapsed = max_over_ranks(dist, dev, [time.perf_counter() - t0])[0]
    gather_s = median_of([a.elapsed_time(b) * 1e-3 for a, b in ex.timings]) if ex.timings else None
    gather_s = max_over_ranks(dist, dev, [gather_s])[0] if gather_s else None
    sent = ex.bytes_per_exchange['sent']
    backend = dist.get_backend() if dist is not None else 'none (single process: local copy)'
    out = {'value': world * eng.num_envs * T * steps / elapsed, 'unit': 'env-steps/s', 'exchanges': steps, 'cycles_per_exchange': T,
           'seconds_per_round': elapsed / steps, 'collectives_per_exchange': 1, 'backend': backend,
           'rccl_ranks': world if backend == 'nccl' else 0,
           'bytes_sent_per_rank': sent, 'bytes_received_per_rank': ex.bytes_per_exchange['received'],
           'gather_seconds': gather_s,
           'payload': 'action, reward, done, result (10 B per env-step)' if payload == 'league' else 'the whole record (50 B per env-step)',
           # the link bound of the WHOLE JOB for this payload (one exchange per launch), next to the measured value; and for both
           # payloads at this batch, so that the choice can be read off the line
           'xgmi_bound_env_steps_s': xgmi_bound(max(world, 2), eng.num_envs, T, sent),
           'xgmi_bound_by_payload': {'record_50B': xgmi_bound(max(world, 2), eng.num_envs, T, eng.num_envs * T * RECORD_BYTES),
                                     'league_10B': xgmi_bound(max(world, 2), eng.num_envs, T, eng.num_envs * T * 10)}}
    if gather_s and world > 1:
        # a direct all-gather sends this rank's slab once over each of its world-1 links and receives one slab over each
        per_peer = sent / gather_s / 1e9
        out['xgmi'] = {'achieved_GBps_per_peer_link': per_peer, 'peak_GBps_per_link': XGMI_LINK_GBS, 'frac_of_link': per_peer / XGMI_LINK_GBS,
                       'aggregate_ingress_GBps': (world - 1) * per_peer, 'aggregate_peak_GBps': 7 * XGMI_LINK_GBS,
                       'note': 'measured over gloo through host memory, not xGMI' if backend != 'nccl' else 'RCCL all_gather_into_tensor'}
    del ex
    return out


def with_deadline(fn, seconds, fallback):
    """Run fn() on this thread; if it has not returned after `seconds`, call fallback() from a timer thread (it prints the
    line without this measurement and ends the process): a collective that never completes must not cost the whole line."""
    done = threading.Event()

    def watchdog():
        if not done.wait(seconds):
            fallback()
    th = threading.Thread(target=watchdog, daemon=True)
    th.start()
    try:
        return fn()
    finally:
        done.set()


def run_reach(args, dev, dist, rank, world):
    import torch
    n, T = args.envs, max(1, args.fuse)
    stream = torch.cuda.current_stream(dev)
    metric, workload = workload_of('reach_ball', n, world, args.league_exchange)
    default_line = world == 1 and not args.no_secondary and args.mode == 'rollout' and args.variant == 'dqn' and not args.league_exchange
    eng = reach_engine(n, dev, rank, args.noise, args.variant)
    per_buf = T * n * RECORD_BYTES
    nbuf = args.rotate_buffers if args.rotate_buffers > 0 else n_rotating(per_buf)
    K = args.steps

    cold = None
    if default_line:
        # the cold figure: the first GPU work of this process -- 4 warm-up launches, one region of 20, no settle phase
        m = measure_rollout(eng, T, 20, 1, 1, stream, 0.0, warm_regions=0)
        cold = rollout_entry(m, n, T, eng.kernel_name(), None)
        cold['settle_ms'] = 0

    league = None
    if args.mode == 'rollout' and not args.league_exchange:
        m = measure_rollout(eng, T, K, nbuf, args.repeats, stream, args.settle_ms, dist, dev, warm=args.warmup)
        wall, launch_s, launch_ev = m['wall'], m['launch_s'], m['launch_s_events']
        steps_per_launch, alg = T, m['alg_bytes_launch']
        traffic_key = ('rollout' if nbuf == 1 else 'rollout-rotate') + ('-noise' if args.noise else '')
        n_launches = K
    elif args.mode == 'rollout':
        # configs[4]: a bench step = one rollout launch + the all-gather of its slab (side stream, overlapped with the next)
        settle(lambda c: [eng.rollout(T) for _ in range(max(1, c // T))], 16 * T, args.settle_ms)
        league = measure_league_exchange(eng, T, K, args.warmup, dist, dev, world, args.league_payload)
        wall = [league['seconds_per_round'] * K]
        m = measure_rollout(eng, T, K, 1, args.repeats, stream, 0.0, dist, dev, warm=1)     # the same launches without the exchange
        launch_s, launch_ev = league['seconds_per_round'], m['launch_s_events']
        league['value_without_exchange'] = world * n * T / m['launch_s']
        steps_per_launch, alg, traffic_key, n_launches = T, m['alg_bytes_launch'], 'rollout', K
        nbuf = 2
    else:
        # one launch per cycle: eager (step) or one hipGraph of T single-cycle launches replayed (graph)
        k_cycles = K * args.cycles_per_step
        if args.mode == 'graph':
            for _ in range(3):
                eng.step(None)
            torch.cuda.synchronize()
            g = graph_of(lambda: [eng.step(None) for _ in range(T)])

            def run():
                for _ in range(K):
                    g.replay()
        else:
            if not args.eager:
                g = graph_of(lambda: [eng.step(None) for _ in range(k_cycles)])
                run = g.replay
            else:
                def run():
                    for _ in range(k_cycles):
                        eng.step(None)
        settle(lambda c: [eng.step(None) for _ in range(c)], 2048, args.settle_ms)
        for _ in range(args.warmup * args.cycles_per_step):
            eng.step(None)
        wall, evs = timed_regions(run, args.repeats, stream, dist, dev)
        launch_s, launch_ev = median_of(wall) / k_cycles, median_of(evs) / k_cycles
        steps_per_launch, alg, traffic_key, n_launches = 1, n * (2 * STATE_BYTES + 4 + RECORD_BYTES - 4), 'step', k_cycles
    kernel = eng.kernel_name()
    elapsed = median_of(wall)
    cycles_per_region = K * args.cycles_per_step
    stats = eng.stats.cpu().tolist()
    line = None
    if rank == 0:
        share = os.environ.get('S2D_BENCH_SHARE_GPU', '0') == '1'
        line = {
            'metric': metric,
            'value': world * n * cycles_per_region / elapsed,
            'unit': 'env-steps/s',
            'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': elapsed / args.steps * 1e3,
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
            'config': {'workload': workload + ('; noise on (player_rand 0.1, ball_rand 0.05)' if args.noise else '; noise off (SURVEY 8d)'),
                       'envs_per_gpu': n, 'global_envs': world * n, 'mode': args.mode, 'variant': args.variant,
                       'cycles_per_launch': steps_per_launch if args.mode != 'graph' else f'1 ({T} per graph replay)',
                       'settle_ms': args.settle_ms, 'rollout_buffers': nbuf,
                       'record_bytes_in_flight': nbuf * per_buf if args.mode == 'rollout' else 0,
                       'issue': 'eager' if (args.eager or args.league_exchange) else 'one hipGraph replay per timed region',
                       'noise': bool(args.noise),
                       'parallelism': (f'env-shard x{world} + one all-gather of rollout slabs per launch' if args.league_exchange
                                       else f'env-shard x{world} (no collective)'),
                       'launches_per_region': n_launches, 'timed_regions': len(wall), 'value_is': 'median region'},
            'repeats': [world * n * cycles_per_region / w for w in wall],
            'roofline': roofline_of(alg, launch_s, launch_ev, kernel, traffic_key, T, n, steps_per_launch),
            'episodes': {'goal': stats[1], 'out': stats[2], 'timeout': stats[3]},
        }
        if share:
            line['config']['rehearsal'] = f'{world} ranks share cuda:0, collectives over gloo: NOT a multi-GPU measurement'
        if league is not None:
            line['league_exchange'] = league
    # world > 1, default mode: the league all-gather measured beside the metric (short; RCCL on a real node), under a deadline
    if world > 1 and not args.league_exchange and not args.no_secondary and args.mode == 'rollout':
        def give_up():
            # a collective that never completes is a finding, not a success: the line (without this measurement) is on stdout,
            # the exit code says that the run did not finish
            if rank == 0:
                line.setdefault('secondary', {})['league_exchange'] = {'error': 'no completion within 120 s'}
                print(json.dumps(line), flush=True)
            os._exit(3)
        le = {}
        for payload in ('league', 'record'):                  # both payloads: 10 B and 50 B per env-step
            try:
                le[payload] = with_deadline(lambda: measure_league_exchange(eng, T, 6, 2, dist, dev, world, payload), 120.0, give_up)
            except Exception as ex:
                le[payload] = {'error': repr(ex)}
        if rank == 0:
            line.setdefault('secondary', {})['league_exchange'] = le
    if rank != 0:
        return None
    if default_line:
        del eng
        sec = line.setdefault('secondary', {})
        try:
            secondary_measurements(args, dev, stream, rank, n, T, line, sec)
        except Exception as ex:
            sec['error'] = repr(ex)
        sec['rollout_cold'] = cold
    if world == 1 and not args.no_cpu_baseline:
        try:
            line['cpu_baseline'] = cpu_baseline(n, args.cpu_sample_steps, bool(args.noise))
        except Exception as ex:     # the baseline is a report, never the product
            line['cpu_baseline'] = {'value': None, 'unit': 'env-steps/s', 'cores': 0, 'kind': 'port', 'sample': f'failed: {ex!r}'}
    return line


def box_write_rate(dev, stream, nbytes=1 << 30, reps=8):
    """What this box's memory system takes from a plain streaming write: torch's fill_ of 1 GiB, GB/s (median of `reps`).  The
    pool's boxes differ (the same rollout build measured 71-95 G env-steps/s on different ones); this figure travels with the line
    so that a reader can tell the box from the kernel.  Not a roofline peak: `roofline.peak` stays the 8 TB/s of the data sheet."""
    import torch
    buf = torch.empty(nbytes // 4, dtype=torch.float32, device=dev)
    buf.fill_(1.0); torch.cuda.synchronize()
    rates = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream); buf.fill_(2.0); e1.record(stream)
        torch.cuda.synchronize()
        rates.append(nbytes / (e0.elapsed_time(e1) * 1e-3) / 1e9)
    del buf
    return median_of(rates)


def secondary_measurements(args, dev, stream, rank, n, T, line, out):
    """More figures of the same run, in the same JSON line: the noise-on rollout (the product's drop-in default) as a
    first-class entry `noise_on`; in `secondary` the one-buffer rollout (what rounds 1-2 reported), the per-step API, BASELINE
    configs[1] (4 096 envs) and configs[3] (11v11, 8 192 matches).  Fresh engines, a few seconds in all."""
    import torch
    R = max(3, min(args.repeats, 5))
    nb = n_rotating(T * n * RECORD_BYTES)
    line['roofline']['box_fill_gbs'] = box_write_rate(dev, stream)     # see box_write_rate()
    # noise on (rcssserver's stock player_rand / ball_rand), rotating buffers like the headline
    eng = reach_engine(n, dev, rank, True)
    m = measure_rollout(eng, T, 32, nb, R, stream, args.settle_ms)
    line['noise_on'] = rollout_entry(m, n, T, eng.kernel_name(), 'rollout-rotate-noise')
    line['noise_on']['config'] = 'same workload with player_rand 0.1 / ball_rand 0.05 (make_config default)'
    del eng
    # the other action modes (the reference's default is use_continuous_action=True, reach_ball_env.py:34): the DDPG script's kwargs
    # with 1-D continuous actions, and the same with the 4-D turning mode; uniform random policy, noise off like the headline
    for key, over in (('continuous', {}), ('turning4', {'use_turning': True})):
        eng = reach_engine(n, dev, rank, False, task_kwargs=dict(DDPG_KWARGS, **over))
        m = measure_rollout(eng, T, 32, nb, R, stream, args.settle_ms)
        out[key] = rollout_entry(m, n, T, eng.kernel_name(), None)
        out[key]['config'] = ('kwargs of ddpg_stable_baselines3.py:18-31' + (' with use_turning=True (4-D actions, reach_ball_env.py:59-79)' if over else '') +
                              '; record 50 B per env-step' + (' + 12 B of action' if over else ''))
        del eng
    eng = reach_engine(n, dev, rank, False)
    # the rounds 1-2 launch shape: 64 cycles per launch, (a) into rotating buffers, (b) re-writing ONE 218.6 MB buffer, which the
    # 256 MiB Infinity Cache can hold (the rounds 1-2 headline)
    m = measure_rollout(eng, 64, 64, n_rotating(64 * n * RECORD_BYTES), R, stream, args.settle_ms)
    out['rollout_T64_rotating'] = rollout_entry(m, n, 64, eng.kernel_name(), 'rollout-rotate')
    m = measure_rollout(eng, 64, 64, 1, R, stream, args.settle_ms)
    out['rollout_T64_one_buffer'] = rollout_entry(m, n, 64, eng.kernel_name(), 'rollout')
    out['rollout_T64_one_buffer']['note'] = 'the record re-writes ONE 218.6 MB buffer, which the 256 MiB Infinity Cache can hold (rounds 1-2 headline)'
    # per-step API (what an SB3-style learner drives), 2 048 launches per region
    out['step_api'] = measure_steps(eng, 2048, R, stream, args.settle_ms)
    acts = torch.randint(0, DQN_KWARGS['action_space_size'], (n,), device=dev, dtype=torch.int32)
    out['step_api_caller_actions'] = measure_steps(eng, 2048, R, stream, args.settle_ms, actions=acts)
    out['step_api_caller_actions']['actions'] = 'int32[N] device tensor (what dqn_stable_baselines3.py hands to step())'
    out['step_api_dqn_policy'] = measure_steps_with_policy(eng, 256, R, stream, args.settle_ms)
    # s2d_step_k: the per-step API with k cycles per launch (a learner that holds its actions for k steps)
    out['step_k'] = [measure_step_k(eng, k, 2048 // k, R, stream, args.settle_ms) for k in (1, 2, 4, 8)]
    del eng
    # BASELINE configs[1]: 4 096 envs (64 workgroups on 256 CUs: chain-latency-bound)
    eng = reach_engine(4096, dev, rank, False)
    m = measure_rollout(eng, T, 32, 2, R, stream, args.settle_ms)
    out['reach_ball_4096'] = rollout_entry(m, 4096, T, eng.kernel_name(), 'rollout-rotate')
    out['reach_ball_4096']['workload'] = workload_of('reach_ball', 4096, 1, False)[1]
    out['reach_ball_4096']['step_api'] = measure_steps(eng, 2048, R, stream, args.settle_ms)
    del eng
    # BASELINE configs[3]: 11v11, 8 192 matches
    mm = measure_match(8192, dev, rank, 64, 16, R, stream, args.settle_ms, phase='spread')
    mm.pop('wall', None)
    mm['workload'] = workload_of('match', 8192, 1, False)[1]
    mm['rules'] = ('rcssserver stock: 2 x 3 000 cycles, a draw is extended by 2 x 1 000 and then goes to the penalty shoot-out -- under the '
                   'uniform random policy nearly every match is a 0-0 draw and a fifth of all match-cycles are shoot-out cycles')
    out['match_8192'] = mm
    # the same without the shoot-out (a per-engine word of the stock instantiation): what rounds 1-4 measured before the shoot-out existed
    m2 = measure_match(8192, dev, rank, 64, 16, R, stream, args.settle_ms, phase='spread', match_kw={'penalty_shoot_outs': 0})
    m2.pop('wall', None)
    m2['rules'] = 'penalty_shoot_outs = 0: a draw after extra time stands (TimeOver at cycle 8 000)'
    out['match_8192_no_shoot_out'] = m2


def run_match(args, dev, dist, rank, world):
    import torch
    n, T = args.envs, max(1, args.fuse if args.fuse != 256 else 64)     # the 11v11 rollout record is 489 B per match-step: 64 cycles per launch
    stream = torch.cuda.current_stream(dev)
    metric, workload = workload_of('match', n, world, False)
    mm = measure_match(n, dev, rank, T, args.steps, args.repeats, stream, args.settle_ms, noise=args.noise,
                       phase=args.match_phase, mode='rollout' if args.mode == 'rollout' else 'step', dist=dist)
    if rank != 0:
        return None
    elapsed = median_of(mm['wall'])
    per_launch = mm['cycles_per_launch']
    line = {'metric': metric, 'value': world * n * per_launch * args.steps / elapsed, 'unit': 'env-steps/s', 'n_gpus': world,
            'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': elapsed / args.steps * 1e3, 'higher_is_better': True,
            'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
            'config': {'workload': workload, 'envs_per_gpu': n, 'mode': args.mode, 'cycles_per_launch': per_launch,
                       'settle_ms': args.settle_ms, 'match_phase': args.match_phase, 'noise': bool(args.noise),
                       'player_steps_per_s': world * mm['player_steps_per_s'], 'timed_regions': len(mm['wall']), 'value_is': 'median region',
                       'parallelism': f'match-shard x{world} (no collective)'},
            'repeats': [world * v for v in mm['repeats']], 'roofline': mm['roofline'], 'events': mm['events']}
    if world == 1 and not args.no_cpu_baseline:
        try:
            line['cpu_baseline'] = match_cpu_baseline(n)
        except Exception as ex:
            line['cpu_baseline'] = {'value': None, 'unit': 'env-steps/s', 'cores': 0, 'kind': 'port', 'sample': f'failed: {ex!r}'}
    return line


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    args = parse(argv)
    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        return self_launch(args, argv)
    import torch
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    args.gpus = world
    assert torch.cuda.is_available(), 'bench.py needs MI355X GPUs'
    dev, dist = init_distributed(rank, local_rank, world)
    line = run_match(args, dev, dist, rank, world) if args.task == 'match' else run_reach(args, dev, dist, rank, world)
    if rank == 0:
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
