#!/bin/bash
# rocm-smi socket power / shader clock, sampled once a second, while the headline rollout runs back to back (sustained_load.py):
# noise off, then noise on.  Is the rate the shader clock of a power-limited chip?   bash profiles/experiments/power_probe2.sh
OUT=gpurun_out/power2; mkdir -p $OUT
probe() {
  python profiles/experiments/sustained_load.py 14 $1 > $OUT/load_$2.txt 2>&1 &
  BP=$!
  for i in $(seq 30); do
    if ! kill -0 $BP 2>/dev/null; then break; fi
    echo "$2 t=$i $(rocm-smi --showpower --showclocks --showuse --showtemp 2>/dev/null | grep -E 'Power \(W\)|sclk|mclk|fclk|GPU use|junction|memory\)' | sed 's/GPU\[0\]\s*: //' | tr -s ' \t' ' ' | tr '\n' ';')"
    sleep 1
  done
  wait $BP
  grep -E "^noise|\+" $OUT/load_$2.txt
}
probe "" off
probe "--noise" on
