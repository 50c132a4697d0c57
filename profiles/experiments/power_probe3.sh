#!/bin/bash
# rocm-smi socket power / shader clock once a second under the other kernels' sustained load (11v11, per-step API, cached T = 64 rollout)
OUT=gpurun_out/power3; mkdir -p $OUT
for KIND in match step rollout2; do
  python profiles/experiments/sustained_other.py $KIND 11 > $OUT/load_$KIND.txt 2>&1 &
  BP=$!
  for i in $(seq 30); do
    if ! kill -0 $BP 2>/dev/null; then break; fi
    echo "$KIND t=$i $(rocm-smi --showpower --showclocks --showuse 2>/dev/null | grep -E 'Power \(W\)|sclk|GPU use' | sed 's/GPU\[0\]\s*: //' | tr -s ' \t' ' ' | tr '\n' ';')"
    sleep 1
  done
  wait $BP
  grep -v amdgpu.ids $OUT/load_$KIND.txt
done
