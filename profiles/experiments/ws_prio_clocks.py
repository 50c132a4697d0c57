"""ws_prio_clocks.py lib.so ...: clocks (s_memtime) the workgroups of the rollout count in their loop, per build -- the launch lasts
as long as the slowest workgroup, and clocks do not depend on the box's clock level.  Debug builds with s2d_debug_wg."""
import ctypes as C
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 2 or (len(sys.argv) == 2 and not os.environ.get('S2D_CHILD')):
    for lib in sys.argv[1:]:
        r = subprocess.run([sys.executable, __file__, lib], env=dict(os.environ, S2D_LIB=os.path.abspath(lib), S2D_CHILD='1'), stdout=subprocess.PIPE, text=True)
        print(r.stdout.strip(), flush=True)
    sys.exit(0)
sys.path.insert(0, ROOT)
import numpy as np
import torch
import bench

lib = C.CDLL(os.environ['S2D_LIB'])
dev = torch.device('cuda', 0)
n, T = 65536, 256
buf = np.zeros(2048 * 2, dtype=np.uint32)
out = []
for noise in (False, True):
    eng = bench.reach_engine(n, dev, 0, noise)
    bufs = [eng.alloc_rollout(T) for _ in range(2)]
    for i in range(60):
        eng.rollout(T, out=bufs[i & 1])
    meds, maxs, uss = [], [], []
    for rep in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for j in range(8):
            eng.rollout(T, out=bufs[j & 1])
        e1.record(); torch.cuda.synchronize()
        lib.s2d_debug_wg(buf.ctypes.data_as(C.POINTER(C.c_uint)))
        dur = buf.reshape(2048, 2)[:1024, 0].astype(np.int64)
        meds.append(np.median(dur)); maxs.append(dur.max()); uss.append(e0.elapsed_time(e1) * 1e3 / 8)
    out.append(f'noise={int(noise)}: loop clocks median {np.median(meds):.0f} max {np.median(maxs):.0f}  ({np.median(uss):.1f} us per launch)')
print(f'{os.path.basename(os.environ["S2D_LIB"]):18s} ' + ' | '.join(out))
