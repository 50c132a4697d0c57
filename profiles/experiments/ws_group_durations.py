"""ws_group_durations.py <debug build>: clocks each of the 1 024 workgroups of ONE launch of the four-wave rollout spends in its loop
(read by the agent wave), by the order in which its CU received it -- does the oldest-first issue arbiter favour early workgroups
here as it did in the 11v11 kernel?   S2D_LIB=.../wg.so python3 profiles/experiments/ws_group_durations.py [--noise]"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import bench

lib = C.CDLL(os.environ['S2D_LIB'])
dev = torch.device('cuda', 0)
n, T = 65536, 256
for noise in (False, True):
    eng = bench.reach_engine(n, dev, 0, noise)
    bufs = [eng.alloc_rollout(T) for _ in range(2)]
    for i in range(40):
        eng.rollout(T, out=bufs[i & 1])
    buf = np.zeros(2048 * 2, dtype=np.uint32)
    for launch in range(2):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); eng.rollout(T, out=bufs[launch & 1]); e1.record(); torch.cuda.synchronize()
        lib.s2d_debug_wg(buf.ctypes.data_as(C.POINTER(C.c_uint)))
        w = buf.reshape(2048, 2)[:1024].astype(np.int64)
        dur = w[:, 0]; blk = np.arange(1024)
        print(f'noise={int(noise)} launch {launch}: {e0.elapsed_time(e1) * 1e3:.1f} us; loop clocks p10 {np.percentile(dur, 10):.0f} p50 {np.percentile(dur, 50):.0f} '
              f'p90 {np.percentile(dur, 90):.0f} max {dur.max()}')
        print('   median by the turn in which its CU received the workgroup (block // 8 // 32): ' + ' '.join(f'{np.median(dur[(blk // 8) // 32 == k]):.0f}' for k in range(4)))
        print('   by XCD: ' + ' '.join(f'{np.median(dur[blk % 8 == x]):.0f}' for x in range(8)))
