"""A/B of rollout-kernel builds at steady clocks.  Each library variant runs in its own process (the library is
loaded once per process); every variant reports us per launch (T fused cycles, N envs) over `reps` repeats and a
checksum of the rollout record, so that result-preserving variants can be told from spec-changing ones.

  python profiles/experiments/ab_rollout.py [--envs 65536] [--fuse 64] [--noise] lib/exp/a.so lib/exp/b.so ...
"""
import argparse
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
KW = dict(change_ball_position=True, change_ball_velocity=True, min_distance_to_ball=5.0, max_steps=200,
          use_continuous_action=False, action_space_size=16, use_turning=False)


def child(args):
    sys.path.insert(0, os.path.join(ROOT, 'gym-soccer-2d-env_amd'))
    import time
    import torch
    from soccer2d_amd.engine import Engine, make_config
    n, T = args.envs, args.fuse
    kw, sp = dict(KW), None
    if args.variant != 'dqn':                              # 'timeouts': episodes end by max_steps only; 'never': they never end
        kw.update(min_distance_to_ball=0.0)
        sp = dict(pitch_half_length=1e6, pitch_half_width=1e6)
        if args.variant == 'never':
            kw.update(max_steps=1000000000)
    make = lambda: Engine(n, 'cuda:0', cfg=make_config(noise=args.noise, server_params=sp, **kw))
    eng = make()
    eng.reset()
    ro = eng.alloc_rollout(T)
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < args.settle:
        for _ in range(16):
            eng.rollout(T, out=ro)
        torch.cuda.synchronize()
    res = []
    for _ in range(args.reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(args.launches):
            eng.rollout(T, out=ro)
        e1.record()
        torch.cuda.synchronize()
        res.append(e0.elapsed_time(e1) * 1e3 / args.launches)
    # checksum of a fresh engine's first two launches (comparable across variants)
    eng2 = make()
    eng2.reset()
    eng2.rollout(T, out=ro)
    eng2.rollout(T, out=ro)
    torch.cuda.synchronize()
    cs = [float(ro['obs'].double().sum()), float(ro['reward'].double().sum()), int(ro['done'].long().sum()),
          int(ro['action'].long().sum()), int(eng2.cycle.long().sum())]
    print(json.dumps({'lib': os.environ.get('S2D_LIB'), 'us': res, 'kernel': eng.kernel_name(), 'checksum': cs,
                      'G': [n * T / u / 1e3 for u in res]}), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('libs', nargs='*')
    ap.add_argument('--envs', type=int, default=65536)
    ap.add_argument('--fuse', type=int, default=64)
    ap.add_argument('--noise', action='store_true')
    ap.add_argument('--variant', default='dqn', choices=('dqn', 'timeouts', 'never'))
    ap.add_argument('--launches', type=int, default=512)
    ap.add_argument('--reps', type=int, default=3)
    ap.add_argument('--settle', type=float, default=0.3)
    ap.add_argument('--child', action='store_true')
    args = ap.parse_args()
    if args.child:
        return child(args)
    base = None
    for lib in args.libs:
        env = dict(os.environ, S2D_LIB=os.path.abspath(lib))
        cmd = [sys.executable, os.path.abspath(__file__), '--child', '--envs', str(args.envs), '--fuse', str(args.fuse),
               '--launches', str(args.launches), '--reps', str(args.reps), '--settle', str(args.settle), '--variant', args.variant]
        if args.noise:
            cmd.append('--noise')
        r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
        line = [l for l in r.stdout.splitlines() if l.startswith('{')]
        if r.returncode != 0 or not line:
            print(f'{lib}: FAILED rc={r.returncode}\n{r.stderr[-2000:]}', flush=True)
            continue
        d = json.loads(line[-1])
        us = sorted(d['us'])[len(d['us']) // 2]
        if base is None:
            base = (us, d['checksum'])
        same = 'same' if d['checksum'] == base[1] else 'DIFFERENT'
        print(f"{os.path.basename(lib):28s} {us:8.2f} us/launch  {args.envs * args.fuse / us / 1e3:7.2f} G/s  x{base[0] / us:5.3f}  "
              f"[{' '.join(f'{u:.2f}' for u in d['us'])}]  results {same}  {d['kernel']}", flush=True)


if __name__ == '__main__':
    main()
