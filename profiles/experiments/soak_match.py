"""Long random-policy run of the 11v11 engine against its oracle: 1 024 matches x 6 000 cycles with short halves
(half_time_cycles = 700, so every match goes through half times, time over and the automatic restart several
times), noise on, heterogeneous types, after-goal pause; every word compared every 250 cycles."""
import os, sys
sys.path.insert(0, 'gym-soccer-2d-env_amd'); sys.path.insert(0, 'tests')
import numpy as np
import torch
import test_gpu_match as T

n, cycles = 1024, 6000
ids = [0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 0, 11, 12, 13, 14, 15, 16, 17, 1, 2, 3]
eng, orc = T._pair(n, hetero_seed=11, player_type_id=ids, half_time_cycles=700, noise=True, after_goal_wait=50)
for t in range(cycles):
    eng.step(None); orc.step(None)
    if (t + 1) % 250 == 0:
        T.assert_match_same(eng, orc, f't={t + 1}')
T.assert_match_same(eng, orc, 'final')
print('match soak ok:', n, 'matches x', cycles, 'cycles; stats', list(eng.stats.cpu().numpy()), '== oracle', list(orc.stats()))
assert list(eng.stats.cpu().numpy()) == list(orc.stats())
