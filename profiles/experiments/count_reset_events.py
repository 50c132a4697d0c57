import os, sys
sys.path.insert(0, 'gym-soccer-2d-env_amd'); sys.path.insert(0, 'profiles/experiments')
import torch
from soccer2d_amd.engine import Engine, make_config
from ablate import KW
eng = Engine(65536, 'cuda:0', cfg=make_config(**KW)); eng.reset()
for _ in range(8): eng.rollout(64, with_obs=False)
eng.stats_reset()
for _ in range(16): eng.rollout(64, with_obs=False)
torch.cuda.synchronize()
s = eng.stats.cpu().tolist()
ws = 1024 * 64 * 16
print('env-steps', s[0], 'resets', s[1]+s[2]+s[3], 'refill events/wave-step', s[4]/ws, 'fallback fills per reset', s[5]/max(1,s[6]), 'resets(lanes)/wave-step', s[6]/ws)
