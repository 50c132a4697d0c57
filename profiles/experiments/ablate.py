"""Runtime ablations of the rollout kernel (timing only; no rebuild): which part of the
cycle dominates?  Usage on the GPU box: python profiles/experiments/ablate.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, 'gym-soccer-2d-env_amd'))
import torch
from soccer2d_amd.engine import Engine, make_config

KW = dict(change_ball_position=True, change_ball_velocity=True, min_distance_to_ball=5.0, max_steps=200,
          use_continuous_action=False, action_space_size=16, use_turning=False)


def timeit(eng, T, reps, actions=None, with_obs=True):
    ro = eng.alloc_rollout(T, with_obs=with_obs)
    for _ in range(3):
        eng.rollout(T, actions=actions, out=ro)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        eng.rollout(T, actions=actions, out=ro)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / (reps * T)     # us per cycle


def main():
    T, reps = 64, 20
    for n in (65536, 1048576):
        rows = []
        eng = Engine(n, 'cuda:0', cfg=make_config(**KW)); eng.reset()
        rows.append(('baseline (random policy, auto-reset, obs)', timeit(eng, T, reps)))
        rows.append(('no obs stream', timeit(eng, T, reps, with_obs=False)))
        acts = torch.randint(0, 16, (T, n), dtype=torch.int32, device='cuda:0')
        rows.append(('caller actions (no Philox policy)', timeit(eng, T, reps, actions=acts)))
        kw = dict(KW); kw['max_steps'] = 1000000; kw['min_distance_to_ball'] = 0.0
        eng2 = Engine(n, 'cuda:0', cfg=make_config(server_params=dict(pitch_half_length=1e6, pitch_half_width=1e6), **kw)); eng2.reset()
        rows.append(('never done (no reset path taken)', timeit(eng2, T, reps)))
        rows.append(('never done + caller actions', timeit(eng2, T, reps, actions=acts)))
        eng4 = Engine(n, 'cuda:0', cfg=make_config(auto_reset=False, **KW)); eng4.reset()
        rows.append(('auto_reset off (dones flagged, no reset work)', timeit(eng4, T, reps)))
        eng3 = Engine(n, 'cuda:0', cfg=make_config(noise=True, **KW)); eng3.reset()
        rows.append(('noise on', timeit(eng3, T, reps)))
        for name, us in rows:
            print(f'N={n:8d}  {name:45s} {us:8.3f} us/cycle  {n / us / 1e3:8.2f} G env-steps/s', flush=True)


if __name__ == '__main__':
    main()
