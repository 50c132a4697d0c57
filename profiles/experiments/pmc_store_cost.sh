#!/bin/bash
# What the record stores cost the four-wave rollout in the pure-HBM regime: clock (GRBM_GUI_ACTIVE), busy / wait cycles with the
# full record, without the observations and with no record at all.  Separate --pmc passes, kernel-trace only.
TAG=${1:-pmc_store}; T=${2:-1024}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
for R in full none; do
  rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM_WR --output-format csv -d $OUT/$R -- python3 profiles/experiments/roll_driver.py --fuse $T --launches 12 --record $R > $OUT/$R.log 2>&1
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${R}_t -- python3 profiles/experiments/roll_driver.py --fuse $T --launches 12 --record $R > $OUT/${R}_t.log 2>&1
done
python3 - <<PY
import csv, glob, collections
for R in ('full', 'none'):
    for f in glob.glob('$OUT/'+R+'/**/*counter_collection.csv', recursive=True):
        acc = collections.defaultdict(lambda: [0.0,0])
        for r in csv.DictReader(open(f)):
            if 'rollout' in r['Kernel_Name']:
                a = acc[r['Counter_Name']]; a[0] += float(r['Counter_Value']); a[1] += 1
        for c,(v,n) in sorted(acc.items()):
            print(f'{R:5s} {c:26s} per-dispatch {v/n:16.1f}  (n={n})')
    for f in glob.glob('$OUT/'+R+'_t/**/*kernel_stats.csv', recursive=True):
        for r in csv.DictReader(open(f)):
            if 'rollout' in r['Name']:
                print(f"{R:5s} kernel_stats avg_ns {r['AverageNs']} calls {r['Calls']} min {r['MinNs']} max {r['MaxNs']}")
PY
