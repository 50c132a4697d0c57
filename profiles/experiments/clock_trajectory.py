"""How the headline figure moves with what the device did just before (round 4): regions of 20 graph-replayed launches
(65 536 envs x 256 cycles, two rotating buffers), timed one by one from the first GPU work of the process, then after idle gaps
of 50 ms / 500 ms / 3 s, then after capturing a new graph.  Prints G env-steps/s per region, binned."""
import sys, time, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..', '..'))
import torch
import bench

dev = torch.device('cuda:0')
eng = bench.reach_engine(65536, dev, 0, False)
T, K = 256, 20
bufs = [eng.alloc_rollout(T) for _ in range(2)]
k = [0]


def issue(cnt):
    for _ in range(cnt):
        eng.rollout(T, out=bufs[k[0] % 2]); k[0] += 1


t_start = time.perf_counter()
issue(4); torch.cuda.synchronize()
t0 = time.perf_counter(); g = bench.graph_of(lambda: issue(K)); print(f'first capture {1e3 * (time.perf_counter() - t0):.1f} ms')


def regions(cnt, tag, bin_ms=100.0):
    rows = []
    for _ in range(cnt):
        torch.cuda.synchronize(); a = time.perf_counter(); g.replay(); torch.cuda.synchronize(); b = time.perf_counter()
        rows.append((a - t_start, 65536 * T * K / (b - a) / 1e9))
    print(tag)
    first = rows[0][0]
    print('   first eight regions:', ' '.join(f'{v:.1f}' for _, v in rows[:8]))
    b0, cur = first, []
    for t, v in rows + [(1e9, 0)]:
        if (t - b0) * 1e3 >= bin_ms and cur:
            print(f'   +{(b0 - first) * 1e3:7.0f} ms  n={len(cur):3d}  min {min(cur):6.1f}  median {sorted(cur)[len(cur) // 2]:6.1f}  max {max(cur):6.1f}')
            b0, cur = t, []
        cur.append(v)


regions(800, 'from the first GPU work of the process (no settle phase)', 250.0)
for gap in (0.05, 0.5, 3.0):
    time.sleep(gap); regions(100, f'after {gap * 1e3:.0f} ms idle', 50.0)
t0 = time.perf_counter(); g = bench.graph_of(lambda: issue(K)); print(f'second capture {1e3 * (time.perf_counter() - t0):.1f} ms')
regions(100, 'after the second capture', 50.0)
regions(1500, 'five more seconds', 500.0)
