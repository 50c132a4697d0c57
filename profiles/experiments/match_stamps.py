"""match_stamps.py <stamps.so>: per-section cycle counts of the 11v11 rollout kernel (a build with match_stamps.patch applied):
lane 0 of every wave adds the s_memtime difference across each section to a counter; printed as the share of the loop's time.
  S2D_LIB=gym-soccer-2d-env_amd/lib/exp/stamps.so python3 profiles/experiments/match_stamps.py"""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import bench

NAMES = {0: 'policy + entry broadcasts', 1: '1 commands (dash/turn/kick/tackle)', 2: '1b player movement', 3: '1c catch / foul tests',
         4: '2 ball impulses + move', 5: '3 overlap detection (11 pairs)', 6: '3b Jacobi passes', 7: '3c collision aftermath',
         8: '4 set-play distance', 9: '5 referee', 10: 'parking', 11: '6 decay / timers / stamina', 12: 'event counters / auto reset',
         13: 'observation tile + store', 14: 'record words', 15: 'record words + loop back'}
dev = torch.device('cuda', 0)
stream = torch.cuda.current_stream(dev)
m = bench.measure_match(8192, dev, 0, 64, 16, 3, stream, 200.0, phase='spread')
lib = ctypes.CDLL(os.environ['S2D_LIB'])
out = (ctypes.c_ulonglong * 32)()
lib.s2d_match_debug_stamps(out, 1)
bench_again = bench.measure_match(8192, dev, 0, 64, 16, 3, stream, 0.0, phase='spread')
lib.s2d_match_debug_stamps(out, 0)
tot = sum(out)
print('match-steps/s (stamped build):', round(bench_again['value'] / 1e9, 3), 'G')
for k in range(16):
    print(f'{NAMES[k]:40s} {out[k] / tot * 100:5.1f} %   {out[k]:>14d}')
