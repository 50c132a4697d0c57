"""match_event_share.py <counting build>: what share of the 11v11 benchmark's wave-cycles takes which path (a build with counters
in match_cycle: all wave-cycles, eventful copy, some match not in play_on, a kick connected, an offside flag up, overlap scan, full
referee pass).   S2D_LIB=.../cnt.so python3 profiles/experiments/match_event_share.py"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import bench

lib = C.CDLL(os.environ['S2D_LIB'])
out = (C.c_ulonglong * 16)()
dev = torch.device('cuda', 0)
m = bench.measure_match(8192, dev, 0, 64, 16, 3, torch.cuda.current_stream(dev), 200.0, phase='spread')
lib.s2d_match_debug_counts(out, 1)
m = bench.measure_match(8192, dev, 0, 64, 16, 3, torch.cuda.current_stream(dev), 0.0, phase='spread')
lib.s2d_match_debug_counts(out, 0)
tot = max(1, out[0])
for k, name in enumerate(('wave-cycles', 'eventful copy', 'a match not in play_on', 'a kick / tackle connected', 'an offside flag up',
                          'overlap scan (Jacobi passes)', 'full referee pass')):
    print(f'{name:32s} {out[k]:>12d}  {100.0 * out[k] / tot:6.2f} %')
