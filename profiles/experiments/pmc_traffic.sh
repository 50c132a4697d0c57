#!/bin/bash
# HBM traffic of the bench's dominant kernel from PMC counters, as MI355X_MICROARCH.md prescribes:
# FETCH_SIZE and WRITE_SIZE in SEPARATE --pmc passes (they do not fit one pass), no tracing
# domains besides --kernel-trace; values are KiB; on gfx950 FETCH_SIZE reports half of the bytes of
# wide coalesced reads (doubled below for 16 B/lane streams only -- the kernels here read 4 B/lane
# dword planes, so the raw value is also kept).
# Usage on the GPU box: bash profiles/experiments/pmc_traffic.sh <tag>
set -e
TAG=${1:-traffic}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
run() { # name, bench args...
  NAME=$1; shift
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/${NAME}_fetch -- python3 bench.py --no-cpu-baseline --no-secondary "$@" > $OUT/${NAME}_fetch.log 2>&1
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/${NAME}_write -- python3 bench.py --no-cpu-baseline --no-secondary "$@" > $OUT/${NAME}_write.log 2>&1
}
run rollout64k --steps 16 --warmup 1
run rollout64k_rot --steps 18 --warmup 3 --rotate-buffers 3
run rollout1m --steps 16 --warmup 1 --envs 1048576
run step64k --steps 1024 --warmup 64 --mode step
run match8k --task match --steps 8 --warmup 1
python3 - <<PY
import csv, glob, json, collections
def avg(tag, counter, kern):
    tot, n = 0.0, 0
    for f in glob.glob('$OUT/%s/**/*counter_collection.csv' % tag, recursive=True):
        for r in csv.DictReader(open(f)):
            if kern in r['Kernel_Name'] and r['Counter_Name'] == counter:
                tot += float(r['Counter_Value']); n += 1
    return (tot / n if n else None), n
rows = []
for name, mode, fuse, envs, kern in (('rollout64k', 'rollout', 64, 65536, 'rollout_'),
                                     ('rollout64k_rot', 'rollout-rotate', 64, 65536, 'rollout_'),
                                     ('rollout1m', 'rollout', 64, 1048576, 'rollout_'),
                                     ('step64k', 'step', 64, 65536, 'step_kernel'),
                                     ('match8k', 'match-rollout', 64, 8192, 'match_rollout')):
    f, nf = avg(name + '_fetch', 'FETCH_SIZE', kern)
    w, nw = avg(name + '_write', 'WRITE_SIZE', kern)
    if f is None or w is None: continue
    rows.append({'name': name, 'mode': mode, 'fuse': fuse, 'envs': envs, 'kernel': kern,
                 'fetch_kib_per_launch': f, 'write_kib_per_launch': w, 'dispatches': [nf, nw],
                 'hbm_bytes_per_launch': (f + w) * 1024.0,
                 'note': 'FETCH_SIZE raw (dword-per-lane reads, no x2 correction applied); WRITE_SIZE exact'})
json.dump({'source': 'rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes', 'rows': rows}, open('$OUT/traffic.json', 'w'), indent=1)
print(json.dumps(rows, indent=1))
PY
