"""alloc_carve.py: the rollout's record buffers carved out of ONE 8 GiB allocation at chosen offsets (same pages, same fragments
every time): does the rate still move with the addresses?  Fields in slab order (obs, action, reward, done, result), each field
`pad` bytes after the end of the previous one; the second rotating buffer follows the first.
  python3 profiles/experiments/alloc_carve.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import bench

dev = torch.device('cuda', 0)
stream = torch.cuda.current_stream(dev)
n, T, L = 65536, 256, 32
big = torch.empty(8 << 30, dtype=torch.uint8, device=dev)
base = (-big.data_ptr()) % (2 << 20)                       # 2 MiB-aligned origin
FIELDS = (('obs', torch.float32, (10,), 40), ('action', torch.int32, (), 4), ('reward', torch.float32, (), 4), ('done', torch.uint8, (), 1),
          ('result', torch.uint8, (), 1))


def carve(off, pad):
    out = {}
    for name, dt, trail, b in FIELDS:
        nb = T * n * b
        out[name] = big[base + off: base + off + nb].view(dt).view((T, n) + trail)
        off += nb + pad
        off = (off + 255) // 256 * 256
    return out, off


def timed(eng, bufs, tag):
    k = [0]

    def issue(cnt):
        for _ in range(cnt):
            eng.rollout(T, out=bufs[k[0] % len(bufs)]); k[0] += 1
    bench.settle(lambda c: issue(max(1, c // T)), 16 * T, 100.0)
    issue(4); torch.cuda.synchronize()
    g = bench.graph_of(lambda: issue(L))
    wall, evs = bench.timed_regions(g.replay, 5, stream, None, None)
    us = sorted(evs)[2] / L * 1e6
    print(f'{tag:44s} {us:7.1f} us/launch  {n * T / us / 1e3:6.1f} G   regions {[round(e / L * 1e6, 1) for e in evs]}', flush=True)


eng = bench.reach_engine(n, dev, 0, False)
for rep in range(2):
    for start, pad in ((0, 0), (0, 4096), (0, 65536 + 4096), (0, (2 << 20) + 8192), (1 << 30, 0), ((1 << 30) + (1 << 20), 0), (3 << 30, 0),
                       ((3 << 30) + 12288, 0), (0, 256 * 37), ((5 << 29), 1 << 20)):
        a, end = carve(start, pad)
        b, _ = carve(end + pad, pad)
        timed(eng, [a, b], f'start {start:#x} pad {pad:#x} (pass {rep})')
