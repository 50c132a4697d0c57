#!/bin/bash
# Instruction-mix counters of the 11v11 rollout kernel (separate --pmc pass, no tracing domains).
set -e
TAG=${1:-pmc_match}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES --output-format csv -d $OUT/p1 -- python3 bench.py --task match --steps 8 --warmup 1 --no-cpu-baseline > $OUT/p1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_THREAD_CYCLES_VALU SQ_LDS_BANK_CONFLICT --output-format csv -d $OUT/p2 -- python3 bench.py --task match --steps 8 --warmup 1 --no-cpu-baseline > $OUT/p2.log 2>&1
python3 - <<PY
import csv, glob, collections
for p in ('p1','p2'):
    for f in glob.glob('$OUT/'+p+'/**/*counter_collection.csv', recursive=True):
        acc = collections.defaultdict(lambda: [0.0,0])
        for r in csv.DictReader(open(f)):
            if 'match_rollout' in r['Kernel_Name']:
                a = acc[r['Counter_Name']]; a[0] += float(r['Counter_Value']); a[1] += 1
        for c,(v,n) in sorted(acc.items()):
            print(f'{c:24s} per-dispatch {v/n:16.1f} per wave-cycle(4096 waves x 64) {v/n/(4096*64):10.2f}')
PY
