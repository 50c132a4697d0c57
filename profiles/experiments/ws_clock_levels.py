"""ws_clock_levels.py <debug build>: wall time of a rollout launch against the shader clocks its workgroups count in their loop
(s_memtime): when the same launch takes 165 us on one occasion and 225 us on another, do the workgroups count the same clocks (the
clock changed) or more (they waited)?   S2D_LIB=.../wg.so python3 profiles/experiments/ws_clock_levels.py"""
import ctypes as C
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import bench

lib = C.CDLL(os.environ['S2D_LIB'])
dev = torch.device('cuda', 0)
n, T = 65536, 256
buf = np.zeros(2048 * 2, dtype=np.uint32)
for noise in (False, True, False):
    eng = bench.reach_engine(n, dev, 0, noise)
    rows = []
    for rep in range(6):
        bufs = [eng.alloc_rollout(T) for _ in range(2)]
        t_end = time.perf_counter() + 0.4
        k = 0
        while time.perf_counter() < t_end:                  # back to back, as in the bench
            eng.rollout(T, out=bufs[k & 1]); k += 1
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for j in range(8):
            eng.rollout(T, out=bufs[j & 1])
        e1.record(); torch.cuda.synchronize()
        lib.s2d_debug_wg(buf.ctypes.data_as(C.POINTER(C.c_uint)))
        dur = buf.reshape(2048, 2)[:1024, 0].astype(np.int64)
        us = e0.elapsed_time(e1) * 1e3 / 8
        rows.append((us, np.median(dur), dur.max()))
    for us, med, mx in rows:
        print(f'noise={int(noise)}  {us:7.1f} us per launch   loop clocks median {med:8.0f} max {mx:8d}   -> {mx / (us - 8.0):7.1f} clocks per us (MHz)', flush=True)
