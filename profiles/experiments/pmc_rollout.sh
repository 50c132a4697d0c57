#!/bin/bash
# Instruction-mix / stall counters of the rollout kernel (separate --pmc passes, no tracing domains).
# Usage on the GPU box: bash profiles/experiments/pmc_rollout.sh <tag> [bench args]
set -e
TAG=${1:-pmc}; shift || true
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
ARGS="--steps 16 --warmup 1 --no-cpu-baseline --no-secondary $@"
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES --output-format csv -d $OUT/p1 -- python3 bench.py $ARGS > $OUT/p1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU --output-format csv -d $OUT/p2 -- python3 bench.py $ARGS > $OUT/p2.log 2>&1
python3 - <<PY
import csv, glob, collections
for p in ('p1','p2'):
    for f in glob.glob('$OUT/'+p+'/**/*counter_collection.csv', recursive=True):
        acc = collections.defaultdict(lambda: [0.0,0])
        for r in csv.DictReader(open(f)):
            if 'rollout' in r['Kernel_Name'] or 'step_kernel' in r['Kernel_Name']:
                a = acc[(r['Kernel_Name'][:40], r['Counter_Name'])]; a[0] += float(r['Counter_Value']); a[1] += 1
        for (k,c),(v,n) in sorted(acc.items()):
            print(f'{k:42s} {c:24s} per-dispatch {v/n:16.1f}  (n={n})')
PY
