"""Interleaved A/B of libs2d_hip.so builds, timed like bench.py times the headline: hipGraph replays of `launches` rollout
launches into rotating buffers, `reps` regions per child process, variants interleaved over `rounds` rounds in one gpurun
call (same device, same thermal state).  Prints median us per launch and a checksum per (lib, T, noise).

  python profiles/experiments/ab_graph.py [--fuse 64,256] [--noise 0,1] [--rounds 3] a.so b.so ...
"""
import argparse
import json
import os
import statistics
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def child(a):
    sys.path.insert(0, ROOT)
    import torch
    import bench
    dev = torch.device('cuda', 0)
    stream = torch.cuda.current_stream(dev)
    eng = bench.reach_engine(a.envs, dev, 0, bool(a.noise_one), a.variant)
    T = a.fuse_one
    nb = bench.n_rotating(T * a.envs * bench.RECORD_BYTES) if a.rotate else 1
    launches = max(4, a.cycles // T)
    if a.record != 'full':                                 # store ablation: no observations / no record at all
        real = eng.alloc_rollout
        def alloc(T_, with_obs=True, slab=False):
            o = real(T_, with_obs=a.record in ('nodr', 'onlyobs'))
            if a.record == 'none':
                o = {k: None for k in o}
            elif a.record == 'nodr':
                o['done'] = None; o['result'] = None
            elif a.record == 'onlyobs':
                o = {k: (v if k == 'obs' else None) for k, v in o.items()}
            return o
        eng.alloc_rollout = alloc
    m = bench.measure_rollout(eng, T, launches, nb, a.reps, stream, a.settle_ms)
    if a.record != 'full':
        print(json.dumps({'us': m['launch_s_events'] * 1e6, 'us_all': [], 'checksum': [a.record], 'frac': 0.0}), flush=True)
        return
    ro = eng.alloc_rollout(T)
    eng2 = bench.reach_engine(a.envs, dev, 0, bool(a.noise_one), a.variant)
    eng2.rollout(T, out=ro); eng2.rollout(T, out=ro)
    torch.cuda.synchronize()
    cs = [float(ro['obs'].double().sum()), float(ro['reward'].double().sum()), int(ro['done'].long().sum()), int(eng2.cycle.long().sum())]
    print(json.dumps({'us': m['launch_s_events'] * 1e6, 'us_all': [e / launches * 1e6 for e in m['events']], 'checksum': cs,
                      'frac': m['alg_bytes_launch'] / m['launch_s_events'] / 8e12}), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('libs', nargs='*')
    ap.add_argument('--envs', type=int, default=65536)
    ap.add_argument('--fuse', default='64')
    ap.add_argument('--noise', default='0')
    ap.add_argument('--variant', default='dqn')
    ap.add_argument('--rounds', type=int, default=3)
    ap.add_argument('--reps', type=int, default=5)
    ap.add_argument('--cycles', type=int, default=4096, help='cycles per timed region')
    ap.add_argument('--settle-ms', type=float, default=200.0)
    ap.add_argument('--rotate', type=int, default=1)
    ap.add_argument('--record', choices=('full', 'noobs', 'none', 'nodr', 'onlyobs'), default='full')
    ap.add_argument('--child', action='store_true')
    ap.add_argument('--fuse-one', type=int, default=64)
    ap.add_argument('--noise-one', type=int, default=0)
    a = ap.parse_args()
    if a.child:
        return child(a)
    res = {}

    def label(lib):                                        # lib/exp/<name>/libs2d_hip.so -> <name>; lib/libs2d_hip.so -> product
        b, d = os.path.basename(lib), os.path.basename(os.path.dirname(os.path.abspath(lib)))
        return b if b != 'libs2d_hip.so' else ('product' if d == 'lib' else d)
    for r in range(a.rounds):
        for T in [int(x) for x in a.fuse.split(',')]:
            for nz in [int(x) for x in a.noise.split(',')]:
                for lib in a.libs:
                    env = dict(os.environ, S2D_LIB=os.path.abspath(lib))
                    cmd = [sys.executable, os.path.abspath(__file__), '--child', '--envs', str(a.envs), '--fuse-one', str(T), '--noise-one', str(nz),
                           '--variant', a.variant, '--reps', str(a.reps), '--cycles', str(a.cycles), '--settle-ms', str(a.settle_ms), '--rotate', str(a.rotate), '--record', a.record]
                    p = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env)
                    line = [ln for ln in p.stdout.splitlines() if ln.startswith('{')]
                    if not line:
                        print(f'{label(lib)} T={T} noise={nz}: FAILED\n{p.stderr[-600:]}', flush=True)
                        continue
                    d = json.loads(line[-1])
                    res.setdefault((label(lib), T, nz), []).append(d)
                    print(f'round {r} {label(lib):28s} T={T:4d} noise={nz} {d["us"]:8.2f} us/launch frac={d["frac"]:.3f} checksum={d["checksum"]}', flush=True)
    print('---- medians over rounds')
    for (lib, T, nz), ds in sorted(res.items(), key=lambda kv: (kv[0][1], kv[0][2], kv[0][0])):
        us = statistics.median(d['us'] for d in ds)
        print(f'{lib:28s} T={T:4d} noise={nz} {us:8.2f} us/launch  {us / T:6.3f} us/cycle  frac={statistics.median(d["frac"] for d in ds):.3f}')


if __name__ == '__main__':
    main()
