#!/bin/bash
# Throughput vs length of the timed region (the chip's clock settles only after tens of ms of load).
for K in 64 256 1024 4096 16384; do for W in 4 1024; do
  python bench.py --steps $K --warmup $W --no-cpu-baseline $EXTRA 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('K=$K W=$W', round(d['value']/1e9,2), 'G', round(d['roofline']['launch_us'],1), 'us/launch frac', round(d['roofline']['frac'],3))"
done; done
