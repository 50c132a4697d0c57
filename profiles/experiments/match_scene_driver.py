"""match_scene_driver.py quiet|kickin|aftergoal: N launches of 40 cycles of the 11v11 kernel with every match forced into one scene
(for counter runs: SQ_INSTS_* per wave-cycle of a quiet match against a waiting one)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import bench  # (puts the package directory on sys.path)
from soccer2d_amd.match import MatchEngine, make_match_config
from soccer2d_amd._capi_match import GM_AFTER_GOAL, GM_KICK_IN, GM_PLAY_ON

scene = sys.argv[1]
dev = torch.device('cuda', 0)
n = 8192
eng = MatchEngine(n, dev, cfg=make_match_config())
ro = eng.alloc_rollout(40)
for _ in range(6):
    eng.reset()
    eng.mode.fill_({'quiet': GM_PLAY_ON, 'kickin': GM_KICK_IN, 'aftergoal': GM_AFTER_GOAL}[scene]); eng.mode_side.fill_(0 if scene == 'quiet' else 1)
    eng.x[:, 22] = 0.0; eng.y[:, 22] = 33.0 if scene == 'quiet' else 34.0; eng.vx[:, 22] = 0.0; eng.vy[:, 22] = 0.0
    eng.rollout(40, out=ro)
torch.cuda.synchronize()
print('done', scene)
