#!/bin/bash
# All bench configurations with the default settle phase (steady clocks) -> DESIGN.md section 7 table
run() { python bench.py --no-cpu-baseline "$@" 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$*', '|', round(d['value']/1e9,3), 'G |', round(r['launch_us'],1), 'us |', round(r['achieved']), 'GB/s | frac', round(r['frac'],3), '|', r['kernel'])"; }
run
run --settle-ms 0
run --envs 4096
run --envs 32768
run --envs 131072
run --envs 262144
run --envs 1048576
run --noise
run --noise --envs 1048576
run --variant no-auto-reset
run --mode step
run --mode graph --steps 64 --warmup 4
run --task match
run --task match --mode step --steps 1024 --warmup 64
for F in 16 256 1024; do run --fuse $F; done
