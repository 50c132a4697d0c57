#!/bin/bash
# fuse_sweep.sh <tag>: the headline launch at T = 64 / 128 / 256 / 512 fused cycles, rotating buffers, noise off and on
TAG=${1:-r03}
OUT=gpurun_out/$TAG
mkdir -p $OUT
for T in 64 128 256 512; do
  K=$((4096 / T))
  for NZ in "" "--noise"; do
    python bench.py --fuse $T --steps $K --warmup 2 --no-secondary --no-cpu-baseline $NZ 2>/dev/null | \
      python -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('T=$T noise=%s value=%.2f G launch_us=%.1f frac=%.3f frac_events=%.3f buffers=%d spread=%.1f..%.1f' % (d['config']['noise'], d['value']/1e9, r['launch_us'], r['frac'], r['frac_events'], d['config']['rollout_buffers'], min(d['repeats'])/1e9, max(d['repeats'])/1e9))"
  done
done | tee $OUT/fuse_sweep.txt
