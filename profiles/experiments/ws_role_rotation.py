"""ws_role_rotation.py: which SIMD do the four role waves of the rollout pipeline land on, and does rotating the role assignment
per workgroup (so that a SIMD hosts a mix of roles instead of four waves of one role) change the rate?  Needs a build with the
experiment hook s2d_debug_rot (see profiles/experiments/ws_role_rotation.patch).
  S2D_LIB=.../rot.so python3 profiles/experiments/ws_role_rotation.py"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import bench
from soccer2d_amd import _capi

lib = _capi.load_library()
lib.s2d_debug_rot.argtypes = [C.c_int, C.POINTER(C.c_uint)]
dev = torch.device('cuda', 0)
stream = torch.cuda.current_stream(dev)
T, n = 256, 65536
hist = (C.c_uint * 16)()
for noise in (False, True):
    for mode in (0, 1, 2, 3, 4, 0):
        lib.s2d_debug_rot(mode, hist)
        eng = bench.reach_engine(n, dev, 0, noise)
        m = bench.measure_rollout(eng, T, 16, 2, 5, stream, 200.0)
        lib.s2d_debug_rot(mode, hist)
        h = list(hist); tot = max(1, sum(h[:4]))
        rows = ' | '.join('role %d: ' % r + ' '.join('%3.0f%%' % (100.0 * h[r * 4 + s] / tot) for s in range(4)) for r in range(4))
        print(f'noise={int(noise)} rotation mode {mode}: {m["launch_s"] * 1e6:7.1f} us/launch   SIMD share per role: {rows}', flush=True)
