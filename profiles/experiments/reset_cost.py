"""Cost of one reset event in the rollout kernels: synchronous resets every max_steps+1 cycles
(all lanes of a wave finish together) against the never-done configuration."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, 'gym-soccer-2d-env_amd'))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch
from soccer2d_amd.engine import Engine, make_config
from ablate import KW, timeit

n, T, reps = 65536, 64, 20
big = dict(pitch_half_length=1e6, pitch_half_width=1e6)
for ms in (1000000, 63, 31, 15, 7, 3):
    kw = dict(KW); kw['max_steps'] = ms; kw['min_distance_to_ball'] = 0.0
    for cbv in (True, False):
        kw['change_ball_velocity'] = cbv
        eng = Engine(n, 'cuda:0', cfg=make_config(server_params=big, **kw)); eng.reset()
        us = timeit(eng, T, reps)
        print(f'max_steps={ms:8d} change_ball_velocity={cbv!s:5s} {us:7.3f} us/cycle   kernel={eng.kernel_name()}', flush=True)
