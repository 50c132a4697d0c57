#!/bin/bash
# VALU instructions by type (rocprofv3 PMC, own passes): gpurun_out/valu_types/{rollout,noise,match}.  Usage: bash profiles/experiments/pmc_valu_types.sh
set -o pipefail
export TMPDIR=/tmp
OUT=gpurun_out/valu_types
mkdir -p $OUT
A="SQ_INSTS_VALU SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32"
B="SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT"
rocprofv3 --kernel-trace --pmc $A --output-format csv -d $OUT/rollout_a -- python3 bench.py --no-cpu-baseline --no-secondary --steps 8 --warmup 1 > $OUT/rollout_a.log 2>&1 || exit 1
rocprofv3 --kernel-trace --pmc $B --output-format csv -d $OUT/rollout_b -- python3 bench.py --no-cpu-baseline --no-secondary --steps 8 --warmup 1 > $OUT/rollout_b.log 2>&1 || exit 1
rocprofv3 --kernel-trace --pmc $A --output-format csv -d $OUT/noise_a -- python3 bench.py --noise --no-cpu-baseline --no-secondary --steps 8 --warmup 1 > $OUT/noise_a.log 2>&1 || exit 1
rocprofv3 --kernel-trace --pmc $B --output-format csv -d $OUT/noise_b -- python3 bench.py --noise --no-cpu-baseline --no-secondary --steps 8 --warmup 1 > $OUT/noise_b.log 2>&1 || exit 1
rocprofv3 --kernel-trace --pmc $A --output-format csv -d $OUT/match_a -- python3 bench.py --task match --steps 8 --warmup 1 --no-cpu-baseline > $OUT/match_a.log 2>&1 || exit 1
rocprofv3 --kernel-trace --pmc $B --output-format csv -d $OUT/match_b -- python3 bench.py --task match --steps 8 --warmup 1 --no-cpu-baseline > $OUT/match_b.log 2>&1 || exit 1
python3 - <<'PY' | tee gpurun_out/valu_types/summary.txt
import csv, glob, collections
for tag, kern, units, what in (('rollout', 'rollout_', 1024 * 256, 'group-cycle'), ('noise', 'rollout_', 1024 * 256, 'group-cycle'), ('match', 'match_rollout', 4096 * 64, 'wave-cycle')):
    acc = collections.defaultdict(list)
    for part in 'ab':
        for f in glob.glob(f'gpurun_out/valu_types/{tag}_{part}/**/*counter_collection.csv', recursive=True):
            for r in csv.DictReader(open(f)):
                if kern in r['Kernel_Name']:
                    acc[r['Counter_Name']].append(float(r['Counter_Value']))
    print(tag)
    for k in sorted(acc):
        v = acc[k]; v = v[len(v) // 4:]          # skip the first launches (warm-up shapes)
        print(f'  {k:28s} {sum(v) / len(v) / units:9.2f} per {what}  (n={len(v)})')
PY
