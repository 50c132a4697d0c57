"""A/B of per-step-API builds at steady clocks: us per s2d_step launch (device time of a hipGraph of 2 048 launches) for the
benchmark workload and for a never-done variant (no episode ever ends: what the reset path costs).
  python profiles/experiments/ab_step.py lib/exp/a.so lib/exp/b.so ..."""
import json, os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
KW = dict(change_ball_position=True, change_ball_velocity=True, min_distance_to_ball=5.0, max_steps=200,
          use_continuous_action=False, action_space_size=16, use_turning=False)


def child(n):
    sys.path.insert(0, os.path.join(ROOT, 'gym-soccer-2d-env_amd'))
    import torch
    from soccer2d_amd.engine import Engine, make_config
    out = {}
    for name, kw, sp in (('dqn', KW, None), ('never-done', dict(KW, max_steps=1000000, min_distance_to_ball=0.0),
                                             dict(pitch_half_length=1e6, pitch_half_width=1e6))):
        eng = Engine(n, 'cuda:0', cfg=make_config(noise=False, server_params=sp, **kw)); eng.reset()
        eng.rollout(256, with_obs=False)                      # spread the episode phases
        t0 = time.perf_counter()
        while time.perf_counter() - t0 < 0.25:
            for _ in range(512):
                eng.step(None)
            torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for _ in range(2048):
                eng.step(None)
        torch.cuda.synchronize()
        res = []
        for _ in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
            res.append(e0.elapsed_time(e1) * 1e3 / 2048)
        out[name] = sorted(res)[1]
        out[name + '_kernel'] = eng.kernel_name()
    print(json.dumps(out), flush=True)


if __name__ == '__main__':
    if sys.argv[1] == '--child':
        child(int(sys.argv[2]))
    else:
        n = 65536
        for lib in sys.argv[1:]:
            r = subprocess.run([sys.executable, os.path.abspath(__file__), '--child', str(n)], env=dict(os.environ, S2D_LIB=os.path.abspath(lib)),
                               stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
            line = [l for l in r.stdout.splitlines() if l.startswith('{')]
            if r.returncode != 0 or not line:
                print(f'{lib}: FAILED\n{r.stderr[-1500:]}', flush=True); continue
            d = json.loads(line[-1])
            print(f"{os.path.basename(lib):24s} dqn {d['dqn']:6.2f} us/step   never-done {d['never-done']:6.2f} us/step   {d['dqn_kernel']}", flush=True)
