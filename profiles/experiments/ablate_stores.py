"""Runtime ablation at steady clocks: how much of a rollout launch is the record store stream?
  python profiles/experiments/ablate_stores.py [lib.so ...]   (each library in its own process)"""
import json, os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
KW = dict(change_ball_position=True, change_ball_velocity=True, min_distance_to_ball=5.0, max_steps=200,
          use_continuous_action=False, action_space_size=16, use_turning=False)


def child():
    sys.path.insert(0, os.path.join(ROOT, 'gym-soccer-2d-env_amd'))
    import torch
    from soccer2d_amd.engine import Engine, make_config
    n, T = 65536, 64
    eng = Engine(n, 'cuda:0', cfg=make_config(noise=False, **KW)); eng.reset()
    full = eng.alloc_rollout(T)
    cases = {'all five record fields': full,
             'no obs': {k: (None if k == 'obs' else v) for k, v in full.items()},
             'obs only': {k: (v if k == 'obs' else None) for k, v in full.items()},
             'no record at all': {k: None for k in full}}
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.3:
        for _ in range(16):
            eng.rollout(T, out=full)
        torch.cuda.synchronize()
    for name, out in cases.items():
        for _ in range(64):
            eng.rollout(T, out=out)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(512):
            eng.rollout(T, out=out)
        e1.record(); torch.cuda.synchronize()
        print(f'  {name:28s} {e0.elapsed_time(e1) * 1e3 / 512:8.2f} us/launch', flush=True)


if __name__ == '__main__':
    if len(sys.argv) > 1 and sys.argv[1] == '--child':
        child()
    else:
        for lib in sys.argv[1:] or [os.path.join(ROOT, 'gym-soccer-2d-env_amd', 'lib', 'libs2d_hip.so')]:
            print(os.path.basename(lib), flush=True)
            r = subprocess.run([sys.executable, os.path.abspath(__file__), '--child'], env=dict(os.environ, S2D_LIB=os.path.abspath(lib)),
                               stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
            print(r.stdout if r.returncode == 0 else r.stdout[-3000:], flush=True)
