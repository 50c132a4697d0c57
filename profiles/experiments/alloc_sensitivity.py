"""alloc_sensitivity.py: is the rollout's rate a property of the box or of where its record buffers happen to lie?  The same
engine, noise off, T = 256, timed with fresh pairs of record buffers (the earlier ones stay allocated, so every pair has new
addresses), then noise on with the LAST pair's addresses, then noise off again.
  python3 profiles/experiments/alloc_sensitivity.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import bench

dev = torch.device('cuda', 0)
stream = torch.cuda.current_stream(dev)
n, T, L = 65536, 256, 32
keep = []


def timed(eng, bufs, tag):
    k = [0]

    def issue(cnt):
        for _ in range(cnt):
            eng.rollout(T, out=bufs[k[0] % len(bufs)]); k[0] += 1
    bench.settle(lambda c: issue(max(1, c // T)), 16 * T, 200.0)
    issue(4); torch.cuda.synchronize()
    g = bench.graph_of(lambda: issue(L))
    wall, evs = bench.timed_regions(g.replay, 5, stream, None, None)
    us = sorted(evs)[2] / L * 1e6
    ptrs = ' '.join('%s@%x' % (f[0], bufs[0][f].data_ptr() % (1 << 30)) for f in ('obs', 'reward', 'done'))
    print(f'{tag:28s} {us:7.1f} us/launch  {n * T / us / 1e3:6.1f} G   regions {[round(e / L * 1e6, 1) for e in evs]}   {ptrs}', flush=True)


eng = bench.reach_engine(n, dev, 0, False)
engn = bench.reach_engine(n, dev, 0, True)
if '--slabs' in sys.argv:                                  # separate tensors vs one slab per record, alternating, fresh addresses each time
    for a in range(4):
        sep = [eng.alloc_rollout(T) for _ in range(2)]
        keep.append(sep)
        timed(eng, sep, f'noise off, separate tensors {a}')
        slab = [eng.alloc_rollout(T, slab=True) for _ in range(2)]
        keep.append(slab)
        timed(eng, slab, f'noise off, slab {a}')
        timed(engn, slab, f'noise ON,  slab {a}')
        timed(engn, sep, f'noise ON,  separate tensors {a}')
    sys.exit(0)
for a in range(5):
    bufs = [eng.alloc_rollout(T) for _ in range(2)]
    keep.append(bufs)
    timed(eng, bufs, f'noise off, allocation {a}')
timed(engn, keep[-1], 'noise ON, allocation 4')
timed(engn, keep[0], 'noise ON, allocation 0')
timed(eng, keep[0], 'noise off, allocation 0 again')
timed(eng, keep[-1], 'noise off, allocation 4 again')
# one big slab, fields at 256-byte-aligned offsets
slab = [eng.alloc_rollout(T, slab=True) for _ in range(2)]
timed(eng, slab, 'noise off, one slab per record')
