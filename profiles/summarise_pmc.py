"""summarise_pmc.py <dir>: turn the rocprofv3 --pmc passes of profiles/run_profile.sh into traffic.json (HBM bytes per launch of the
dominant kernel; bench.py reads it as roofline.traffic), pmc_*_instmix.txt and pmc_match_instmix.json.
FETCH_SIZE / WRITE_SIZE are KiB.  MI355X_MICROARCH.md: on gfx950 FETCH_SIZE reports HALF of the bytes of wide (16 B per lane)
coalesced reads; these kernels read dword-per-lane planes (uncalibrated width), so the raw value is kept and the x2 bound is
stated next to it; WRITE_SIZE is exact for 16-byte-per-lane streaming stores."""
import collections
import csv
import glob
import json
import os
import sys

out = sys.argv[1]


def per_dispatch(tag, kern):
    acc = collections.defaultdict(lambda: [0.0, 0])
    for f in glob.glob(os.path.join(out, tag, '**', '*counter_collection.csv'), recursive=True):
        for r in csv.DictReader(open(f)):
            if kern in r['Kernel_Name']:
                a = acc[r['Counter_Name']]
                a[0] += float(r['Counter_Value']); a[1] += 1
    return {c: (v / n, n) for c, (v, n) in acc.items()}


rows = []
for name, mode, fuse, envs, kern in (('rollout256', 'rollout-rotate', 256, 65536, 'rollout_'), ('rollout64rot', 'rollout-rotate', 64, 65536, 'rollout_'),
                                     ('rollout64one', 'rollout', 64, 65536, 'rollout_'), ('step', 'step', 64, 65536, 'step_kernel'),
                                     ('noise256', 'rollout-rotate-noise', 256, 65536, 'rollout_'), ('r4096', 'rollout-rotate', 256, 4096, 'rollout_'),
                                     ('r1m', 'rollout-rotate', 64, 1048576, 'rollout_'),
                                     ('match', 'match-rollout', 64, 8192, 'match_rollout')):
    f = per_dispatch('pmc_%s_fetch' % name, kern).get('FETCH_SIZE')
    w = per_dispatch('pmc_%s_write' % name, kern).get('WRITE_SIZE')
    if not f or not w:
        continue
    rows.append({'name': name, 'mode': mode, 'fuse': fuse, 'envs': envs, 'kernel': kern, 'fetch_kib_per_launch': f[0],
                 'write_kib_per_launch': w[0], 'dispatches': [f[1], w[1]], 'hbm_bytes_per_launch': (f[0] + w[0]) * 1024.0,
                 'hbm_bytes_per_launch_if_fetch_x2': (2 * f[0] + w[0]) * 1024.0,
                 'note': 'FETCH_SIZE raw (dword-per-lane reads: width uncalibrated, x2 bound alongside); WRITE_SIZE exact'})
json.dump({'source': 'rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE, separate passes (profiles/run_profile.sh)', 'rows': rows},
          open(os.path.join(out, 'traffic.json'), 'w'), indent=1)
print(json.dumps(rows, indent=1))
for tag, kern, waves, cycles, fn in (('rollout', 'rollout_', 4096, 256, 'pmc_rollout_instmix.txt'), ('noise', 'rollout_', 4096, 256, 'pmc_rollout_noise_instmix.txt'),
                                      ('match', 'match_rollout', 4096, 64, 'pmc_match_instmix.txt')):
    lines = []
    tot = {}
    for p in ('mix1', 'mix2'):
        for c, (v, n) in sorted(per_dispatch('pmc_%s_%s' % (tag, p), kern).items()):
            unit = v / (waves * cycles) if tag == 'match' else v / (1024 * cycles)
            lines.append('%-26s per dispatch %16.1f   per %s %10.2f   (n=%d)' % (c, v, 'wave-cycle' if tag == 'match' else 'group-cycle', unit, n))
            tot[c] = unit
    open(os.path.join(out, fn), 'w').write('\n'.join(lines) + '\n')
    print(fn); print('\n'.join(lines))
    if tag == 'match' and 'SQ_INSTS_VALU' in tot:
        json.dump({'instructions_per_wave_cycle': tot['SQ_INSTS_VALU'] + tot['SQ_INSTS_SALU'] + tot['SQ_INSTS_LDS'] + tot.get('SQ_INSTS_VMEM_WR', 0) + tot.get('SQ_INSTS_VMEM_RD', 0),
                   'valu': tot['SQ_INSTS_VALU'], 'salu': tot['SQ_INSTS_SALU'], 'lds': tot['SQ_INSTS_LDS'],
                   'source': 'rocprofv3 --pmc SQ_INSTS_* of bench.py --task match (8 192 matches x 64 cycles per launch, 4 096 waves)'},
                  open(os.path.join(out, 'pmc_match_instmix.json'), 'w'), indent=1)
