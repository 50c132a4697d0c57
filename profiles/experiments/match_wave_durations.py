"""match_wave_durations.py <debug build>: how long does each of the 4 096 waves of ONE steady-state launch of the 11v11 kernel take
(s_memtime around its 64-cycle loop), and what did the slowest ones do (cycles with command events / in a non-play_on mode /
with an overlap scan / through the referee's full pass)?   S2D_LIB=.../wd.so python3 profiles/experiments/match_wave_durations.py"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import bench
from soccer2d_amd.match import MatchEngine, make_match_config

lib = C.CDLL(os.environ['S2D_LIB'])
dev = torch.device('cuda', 0)
n, T = 8192, 64
eng = MatchEngine(n, dev, cfg=make_match_config())
eng.reset()
g = torch.Generator(device='cpu').manual_seed(1234)
eng.cycle += (2 * torch.randint(0, 1500, (n,), generator=g, dtype=torch.int32)).to(dev)
ro = eng.alloc_rollout(T)
for _ in range(60):
    eng.rollout(T, out=ro)
buf = np.zeros(4096 * 8, dtype=np.uint32)
for launch in range(3):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); eng.rollout(T, out=ro); e1.record(); torch.cuda.synchronize()
    lib.s2d_match_debug_waves(buf.ctypes.data_as(C.POINTER(C.c_uint)))
    w = buf.reshape(4096, 8).astype(np.int64)
    dur = w[:, 0]
    start = (w[:, 6] - w[:, 6].min()) & 0xffffffff
    end = start + dur
    order = np.argsort(end)
    q = lambda p: np.percentile(dur, p)
    print(f'launch {launch}: {e0.elapsed_time(e1) * 1e3:.1f} us; loop cycles (100 MHz ticks?) p50 {q(50):.0f} p90 {q(90):.0f} p99 {q(99):.0f} max {dur.max()}; '
          f'last start {start.max()}, span {end.max()}')
    calm = (w[:, 1] == 0) & (w[:, 2] == 0)
    print(f'   waves with no event cycle at all: {calm.sum()} median {np.median(dur[calm]):.0f}; with mode cycles only: {((w[:, 1] == 0) & (w[:, 2] > 0)).sum()} '
          f'median {np.median(dur[(w[:, 1] == 0) & (w[:, 2] > 0)]):.0f}; with command events: {(w[:, 1] > 0).sum()} median {np.median(dur[w[:, 1] > 0]) if (w[:, 1] > 0).any() else 0:.0f}')
    print('   the 12 waves that END last: ' + '; '.join(f'end {end[i]} dur {dur[i]} cmd {w[i, 1]} mode {w[i, 2]} ovl {w[i, 3]} ref {w[i, 4]} m{w[i, 5]}' for i in order[-12:]))
    blk = np.arange(4096) // 4
    print('   median loop cycles by XCD (block % 8): ' + ' '.join(f'{np.median(dur[blk % 8 == x]):.0f}' for x in range(8)))
    print('   by position of the block inside its XCD (block // 8, sixteenths): ' + ' '.join(f'{np.median(dur[(blk // 8) // 8 == k]):.0f}' for k in range(16)))
    print('   by wave of the block: ' + ' '.join(f'{np.median(dur[np.arange(4096) % 4 == k]):.0f}' for k in range(4)))
    hist, edges = np.histogram(dur, bins=12)
    print('   histogram: ' + ' '.join(f'{int(e / 1000)}k:{h}' for h, e in zip(hist, edges)))

