"""Where do the four role waves of the 64-env rollout pipeline run?  (-DS2D_STAMPS build: lane 0 of every role wave records HW_ID and
XCC_ID.)  Prints, per SIMD of a CU, which roles it holds.   S2D_LIB=.../libs2d_hip.so python profiles/experiments/ws_placement.py"""
import os, sys, collections
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import bench

dev = torch.device('cuda', 0)
T, n = 256, int(sys.argv[1]) if len(sys.argv) > 1 else 65536
eng = bench.reach_engine(n, dev, 0, '--noise' in sys.argv)
buf = eng.alloc_rollout(T)
for i in range(3):
    eng.rollout(T, out=buf)
torch.cuda.synchronize()
print(eng.kernel_name())
t = eng.terminal_obs.view(n // 64, 64, 10).cpu()
hw = t[:, 0:8:2, 4].contiguous().view(torch.int32)         # [group, role]
xcc = t[:, 0:8:2, 5].contiguous().view(torch.int32) & 0xf
wave_id, simd, cu, sh, se = hw & 0xf, (hw >> 4) & 3, (hw >> 8) & 0xf, (hw >> 12) & 1, (hw >> 13) & 7
cus = collections.defaultdict(lambda: collections.defaultdict(list))
for g in range(n // 64):
    for r in range(4):
        cus[(int(xcc[g, r]), int(se[g, r]), int(sh[g, r]), int(cu[g, r]))][int(simd[g, r])].append('PSAB'[r])
print('CUs used:', len(cus))
pat = collections.Counter()
for key, simds in cus.items():
    pat[' | '.join(''.join(sorted(simds.get(s, []))) for s in range(4))] += 1
for k, v in pat.most_common(12):
    print(f'{v:4d} CUs: SIMD0..3 = {k}')
same = sum(1 for g in range(n // 64) if len(set(int(simd[g, r]) for r in range(4))) == 4)
print(f'groups whose four waves sit on four different SIMDs: {same} of {n // 64}; groups on one CU: {sum(1 for g in range(n // 64) if len(set((int(xcc[g, r]), int(se[g, r]), int(sh[g, r]), int(cu[g, r])) for r in range(4))) == 1)}')
print('simd of role P/S/A/B, first 8 groups:', [[int(simd[g, r]) for r in range(4)] for g in range(8)])
print('wave slot of role P/S/A/B, first 8 groups:', [[int(wave_id[g, r]) for r in range(4)] for g in range(8)])
