#!/bin/bash
# Instruction-fetch / LDS / VMEM-queue counters of the rollout kernel (separate --pmc passes, no tracing domains).
# Usage on the GPU box: bash profiles/experiments/pmc_frontend.sh <tag> [bench args]
set -e
TAG=${1:-pmcfe}; shift || true
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
ARGS="--steps 16 --warmup 1 --no-cpu-baseline --no-secondary $@"
rocprofv3 --kernel-trace --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $OUT/p1 -- python3 bench.py $ARGS > $OUT/p1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INST_LEVEL_VMEM SQ_INST_CYCLES_VMEM_WR SQ_INST_LEVEL_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL --output-format csv -d $OUT/p2 -- python3 bench.py $ARGS > $OUT/p2.log 2>&1
python3 - <<PY
import csv, glob, collections
for p in ('p1','p2'):
    for f in glob.glob('$OUT/'+p+'/**/*counter_collection.csv', recursive=True):
        acc = collections.defaultdict(lambda: [0.0,0])
        for r in csv.DictReader(open(f)):
            if 'rollout' in r['Kernel_Name'] or 'step_kernel' in r['Kernel_Name']:
                a = acc[(r['Kernel_Name'][:40], r['Counter_Name'])]; a[0] += float(r['Counter_Value']); a[1] += 1
        for (k,c),(v,n) in sorted(acc.items()):
            print(f'{k:42s} {c:28s} per-dispatch {v/n:16.1f}  (n={n})')
PY
