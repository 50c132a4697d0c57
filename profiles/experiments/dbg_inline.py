"""Debug build (-DS2D_DEBUG_INLINE): how many prepared-episode slots run out inside a launch (inline draws on the simulating wave)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), 'gym-soccer-2d-env_amd'))
import torch
from soccer2d_amd.engine import Engine, make_config
KW = dict(change_ball_position=True, change_ball_velocity=True, min_distance_to_ball=5.0, max_steps=200,
          use_continuous_action=False, action_space_size=16, use_turning=False)
T = int(sys.argv[1]) if len(sys.argv) > 1 else 64
eng = Engine(65536, 'cuda:0', cfg=make_config(noise=False, **KW)); eng.reset()
ro = eng.alloc_rollout(T)
prev = eng.stats.clone()
tot = [0, 0, 0]
for it in range(64):
    eng.rollout(T, out=ro)
    torch.cuda.synchronize()
    st = eng.stats
    d = (st - prev).tolist(); prev = st.clone()
    if it >= 16:
        tot[0] += d[1] + d[2] + d[3]; tot[1] += d[4]; tot[2] += d[5]
    if it < 8:
        print(it, 'episodes', d[1] + d[2] + d[3], 'inline draws (lanes)', d[4], '(wave events)', d[5])
print(f'T={T}: per launch over 48 launches: episodes {tot[0] / 48:.0f}, inline draws {tot[1] / 48:.1f} lanes in {tot[2] / 48:.1f} wave events')
