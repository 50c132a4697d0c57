"""Per-role busy cycles of the four-wave rollout (experiment build with s_memtime stamps, profiles/experiments/ws_stamps.patch):
cycles between a barrier's release and the wave's arrival at the next one, per iteration, for the record fully written, without
observations and with no record -- which role is it that the HBM write stream slows down?
  S2D_LIB=.../stamp.so python profiles/experiments/ws_stamps.py [--noise]"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import bench
from soccer2d_amd import _capi

lib = _capi.load_library()
lib.s2d_debug_stamps.argtypes = [C.POINTER(C.c_ulonglong), C.c_int]
dev = torch.device('cuda', 0)
T, n = 256, 65536
NOISE = '--noise' in sys.argv
for record in ('full', 'noobs', 'none'):
    eng = bench.reach_engine(n, dev, 0, NOISE)
    bufs = []
    for _ in range(2):
        o = eng.alloc_rollout(T, with_obs=(record == 'full'))
        if record == 'none':
            o = {k: None for k in o}
        bufs.append(o)
    for i in range(200):
        eng.rollout(T, out=bufs[i & 1])
    out = (C.c_ulonglong * 16)()
    lib.s2d_debug_stamps(out, 1)
    L = 40
    for i in range(L):
        eng.rollout(T, out=bufs[i & 1])
    lib.s2d_debug_stamps(out, 1)
    v = list(out)
    names = ('policy', 'simulate', 'agent', 'ball')
    per = [v[r] / max(1, v[8 + r]) / (T + 3) for r in range(4)]
    print(f'record {record:5s}: busy cycles per iteration (100 MHz ticks x?) ' + ', '.join(f'{names[r]} {per[r]:8.1f}' for r in range(4)) +
          f', ball wave tile flush {v[4] / max(1, v[8 + 3]) / (T + 3):8.1f}', flush=True)
